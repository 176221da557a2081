// Instantiations and launch table of the sliced-fp32 fused resblock-pair kernel.
#include "resblock_pair_split.hpp"

#define VK_PAIRS_CASE(G, CH, EPI)                                                                \
    if (channels == (CH) && epi == (EPI)) {                                                      \
        auto k = resblock_pair_split_kernel<G.b_rep, G.wm, G.wn, (EPI)>;                         \
        static std::atomic<unsigned long long> lds_raised{0};   /* bit d: limit raised on device d */\
        if (hipError_t e = vk_allow_lds(k, lds, lds_raised); e != hipSuccess) return (int)e;     \
        hipLaunchKernelGGL(k, grid, dim3(64 * G.wm * G.wn), lds, s, q);                          \
        return (int)hipGetLastError();                                                           \
    }

#define VK_PAIRS_ALL(G, CH)                     \
    VK_PAIRS_CASE(G, CH, VK_EPI_RESID)          \
    VK_PAIRS_CASE(G, CH, VK_EPI_RESID_ACC)      \
    VK_PAIRS_CASE(G, CH, VK_EPI_RESID_AVG)

int vk_launch_pair_split(int channels, int epi, const PairDev& q, dim3 grid, size_t lds, hipStream_t s)
{
    VK_PAIRS_ALL(kPairSplit32, 32)
    VK_PAIRS_ALL(kPairSplit64, 64)
    return -1;
}
