// Instantiations and launch table of the fp32 multi-receptive-field stage / resblock-chain kernel.
#include "mrf_stage_f32.hpp"

#define VK_MRF_F32_CASE(G)                                                                       \
    if (channels == 32 * G.wm) {                                                                 \
        auto k = mrf_stage_f32_kernel<G.wm, G.wn, G.nt, G.sum != 0>;                             \
        static std::atomic<unsigned long long> lds_raised{0};   /* bit d: limit raised on device d */\
        if (hipError_t e = vk_allow_lds(k, lds, lds_raised); e != hipSuccess) return (int)e;     \
        hipLaunchKernelGGL(k, grid, dim3(64 * G.wm * G.wn), lds, s, q);                          \
        return (int)hipGetLastError();                                                           \
    }

int vk_launch_mrf_f32(int channels, bool single, const MrfDev& q, dim3 grid, size_t lds, hipStream_t s)
{
    if (single) {
        VK_MRF_F32_CASE(kMrfF32_32s)
        VK_MRF_F32_CASE(kMrfF32_64s)
    }
    VK_MRF_F32_CASE(kMrfF32_32)
    VK_MRF_F32_CASE(kMrfF32_64)
    return -1;
}
