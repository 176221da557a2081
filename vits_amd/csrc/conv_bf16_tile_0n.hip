// bf16-input instantiations of tile family 0, narrow variant (half the time tile; launches too small to fill the GPU).
#include "conv_launch.hpp"
#include "conv_bf16.hpp"

#define VK_BF16_CASE(EPI, ACT)                                                                   \
    if (epi == (EPI) && act == (ACT)) {                                                          \
        auto k = conv_bf16_kernel<G.a_rep, G.b_rep, G.wm, G.wn, CKB, (EPI), (ACT) != 0>;         \
        static std::atomic<unsigned long long> lds_raised{0};   /* bit d: limit raised on device d */\
        if (hipError_t e = vk_allow_lds(k, lds, lds_raised); e != hipSuccess) return (int)e;     \
        hipLaunchKernelGGL(k, grid, dim3(64 * G.wm * G.wn), lds, s, d);                          \
        return (int)hipGetLastError();                                                           \
    }

int vk_launch_bf16_tile0n(int epi, int act, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s)
{
    constexpr TileGeom G = kTilesNarrow[0];
    constexpr int CKB = kBf16Ck[0];
    VK_BF16_CASE(VK_EPI_STORE, 0)
    VK_BF16_CASE(VK_EPI_STORE, 1)
    VK_BF16_CASE(VK_EPI_RESID, 1)
    VK_BF16_CASE(VK_EPI_RESID_ACC, 1)
    VK_BF16_CASE(VK_EPI_RESID_AVG, 1)
    VK_BF16_CASE(VK_EPI_UPSAMPLE, 1)
    VK_BF16_CASE(VK_EPI_WN_MID, 0)
    VK_BF16_CASE(VK_EPI_WN_LAST, 0)
    VK_BF16_CASE(VK_EPI_COUPLE_REV, 0)
    VK_BF16_CASE(VK_EPI_COUPLE_FWD, 0)
    return -1;
}
