// Instantiations and launch table of the fused multi-receptive-field stage kernel.
#include "mrf_stage.hpp"

#define VK_MRF_CASE(G)                                                                           \
    if (channels == 32 * G.wm) {                                                                 \
        auto k = mrf_stage_bf16_kernel<G.wm, G.wn, G.nt, G.two != 0, G.sum != 0>;                \
        static std::atomic<unsigned long long> lds_raised{0};   /* bit d: limit raised on device d */\
        if (hipError_t e = vk_allow_lds(k, lds, lds_raised); e != hipSuccess) return (int)e;     \
        hipLaunchKernelGGL(k, grid, dim3(64 * G.wm * G.wn), lds, s, q);                          \
        return (int)hipGetLastError();                                                           \
    }

int vk_launch_mrf_bf16(int channels, bool single, const MrfDev& q, dim3 grid, size_t lds, hipStream_t s)
{
    if (single) {
        VK_MRF_CASE(kMrf32s)
        VK_MRF_CASE(kMrf64s)
    }
    VK_MRF_CASE(kMrf32)
    VK_MRF_CASE(kMrf64)
    VK_MRF_CASE(kMrf128)
    VK_MRF_CASE(kMrf256)
    return -1;
}
