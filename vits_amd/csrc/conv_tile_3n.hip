#include "conv_launch.hpp"
int vk_launch_tile3n(int epi, int act, int vec, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s)
{
    constexpr TileGeom G = kTilesNarrow[3];
    VK_CONV_CASE(VK_EPI_GATE, 0, 0)
    VK_CONV_CASE(VK_EPI_SAMPLE, 0, 0)
    return -1;
}
