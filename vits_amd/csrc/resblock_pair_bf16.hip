// Instantiations and launch table of the bf16 fused resblock-pair kernel.
#include "resblock_pair_bf16.hpp"

#define VK_PAIR16_CASE(G, CH, EPI)                                                               \
    if (channels == (CH) && epi == (EPI)) {                                                      \
        auto k = resblock_pair_bf16_kernel<G.a_rep, G.b_rep, G.wm, G.wn, (EPI)>;                          \
        static std::atomic<unsigned long long> lds_raised{0};   /* bit d: limit raised on device d */\
        if (hipError_t e = vk_allow_lds(k, lds, lds_raised); e != hipSuccess) return (int)e;     \
        hipLaunchKernelGGL(k, grid, dim3(64 * G.wm * G.wn), lds, s, q);                          \
        return (int)hipGetLastError();                                                           \
    }

#define VK_PAIR16_ALL(G, CH)                     \
    VK_PAIR16_CASE(G, CH, VK_EPI_RESID)          \
    VK_PAIR16_CASE(G, CH, VK_EPI_RESID_ACC)      \
    VK_PAIR16_CASE(G, CH, VK_EPI_RESID_AVG)

int vk_launch_pair_bf16(int channels, int epi, const PairDev& q, dim3 grid, size_t lds, hipStream_t s)
{
    VK_PAIR16_ALL(kPairBf16_32, 32)
    VK_PAIR16_ALL(kPairBf16_64, 64)
    VK_PAIR16_ALL(kPairBf16_128, 128)
    VK_PAIR16_ALL(kPairBf16_256, 256)
    return -1;
}
