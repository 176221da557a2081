// Instantiations and launch table of the fused WN layer kernel.
#include "wn_layer.hpp"

#define VK_WN_CASE(BR, LAST, VEC)                                                                \
    if (b_rep == (BR) && last == (LAST) && vec == (VEC)) {                                       \
        auto k = wn_layer_kernel<VK_WN_WAVES, (BR), (LAST) != 0, (VEC) != 0>;                    \
        static size_t lds_allowed = 65536;                                                       \
        if (lds > lds_allowed) {                                                                 \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                 \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            if (e != hipSuccess) return (int)e;                                                  \
            lds_allowed = 160 * 1024;                                                            \
        }                                                                                        \
        hipLaunchKernelGGL(k, grid, dim3(64 * VK_WN_WAVES), lds, s, q);                          \
        return (int)hipGetLastError();                                                           \
    }

int vk_launch_wn_layer(int b_rep, int last, int vec, const WnDev& q, dim3 grid, size_t lds, hipStream_t s)
{
    VK_WN_CASE(1, 0, 0) VK_WN_CASE(1, 0, 1) VK_WN_CASE(1, 1, 0) VK_WN_CASE(1, 1, 1)
    VK_WN_CASE(2, 0, 0) VK_WN_CASE(2, 0, 1) VK_WN_CASE(2, 1, 0) VK_WN_CASE(2, 1, 1)
    return -1;
}
