// Instantiations and launch table of the fused WN layer kernel with bf16 matrix-core inputs.
#include "wn_layer_bf16.hpp"

#define VK_WN16_CASE(BR, LAST)                                                                   \
    if (b_rep == (BR) && last == (LAST)) {                                                       \
        auto k = wn_layer_bf16_kernel<VK_WN_WAVES, (BR), (LAST) != 0>;                           \
        static std::atomic<unsigned long long> lds_raised{0};   /* bit d: limit raised on device d */\
        if (hipError_t e = vk_allow_lds(k, lds, lds_raised); e != hipSuccess) return (int)e;     \
        hipLaunchKernelGGL(k, grid, dim3(64 * VK_WN_WAVES), lds, s, q);                          \
        return (int)hipGetLastError();                                                           \
    }

int vk_launch_wn_layer_bf16(int b_rep, int last, const WnDev& q, dim3 grid, size_t lds, hipStream_t s)
{
    VK_WN16_CASE(1, 0) VK_WN16_CASE(1, 1)
    VK_WN16_CASE(2, 0) VK_WN16_CASE(2, 1)
    VK_WN16_CASE(4, 0) VK_WN16_CASE(4, 1)
    return -1;
}
