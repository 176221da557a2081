// Instantiations and launch table of the fused resblock-pair kernel.
#include "resblock_pair.hpp"

#define VK_PAIR_CASE(G, CH, EPI, VEC)                                                            \
    if (channels == (CH) && epi == (EPI) && vec == (VEC)) {                                    \
        auto k = resblock_pair_kernel<G.b_rep, G.wm, G.wn, G.ck, (EPI), (VEC) != 0>;             \
        static std::atomic<unsigned long long> lds_raised{0};   /* bit d: limit raised on device d */\
        if (hipError_t e = vk_allow_lds(k, lds, lds_raised); e != hipSuccess) return (int)e;     \
        hipLaunchKernelGGL(k, grid, dim3(64 * G.wm * G.wn), lds, s, q);                          \
        return (int)hipGetLastError();                                                           \
    }

#define VK_PAIR_ALL(G, CH)                       \
    VK_PAIR_CASE(G, CH, VK_EPI_RESID, 0)         \
    VK_PAIR_CASE(G, CH, VK_EPI_RESID, 1)         \
    VK_PAIR_CASE(G, CH, VK_EPI_RESID_ACC, 0)     \
    VK_PAIR_CASE(G, CH, VK_EPI_RESID_ACC, 1)     \
    VK_PAIR_CASE(G, CH, VK_EPI_RESID_AVG, 0)     \
    VK_PAIR_CASE(G, CH, VK_EPI_RESID_AVG, 1)

int vk_launch_pair(int channels, int epi, int vec, const PairDev& q, dim3 grid, size_t lds, hipStream_t s)
{
    VK_PAIR_ALL(kPair32, 32)
    VK_PAIR_ALL(kPair64, 64)
    return -1;
}
