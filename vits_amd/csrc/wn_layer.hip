// Instantiations and launch table of the fused WN layer kernel.
#include "wn_layer.hpp"

#define VK_WN_CASE(BR, LAST, VEC)                                                                \
    if (b_rep == (BR) && last == (LAST) && vec == (VEC)) {                                       \
        auto k = wn_layer_kernel<VK_WN_WAVES, (BR), (LAST) != 0, (VEC) != 0>;                    \
        static std::atomic<unsigned long long> lds_raised{0};   /* bit d: limit raised on device d */\
        if (hipError_t e = vk_allow_lds(k, lds, lds_raised); e != hipSuccess) return (int)e;     \
        hipLaunchKernelGGL(k, grid, dim3(64 * VK_WN_WAVES), lds, s, q);                          \
        return (int)hipGetLastError();                                                           \
    }

int vk_launch_wn_layer(int b_rep, int last, int vec, const WnDev& q, dim3 grid, size_t lds, hipStream_t s)
{
    VK_WN_CASE(1, 0, 0) VK_WN_CASE(1, 0, 1) VK_WN_CASE(1, 1, 0) VK_WN_CASE(1, 1, 1)
    VK_WN_CASE(2, 0, 0) VK_WN_CASE(2, 0, 1) VK_WN_CASE(2, 1, 0) VK_WN_CASE(2, 1, 1)
    return -1;
}
