#include "conv_launch.hpp"
int vk_launch_tile1n(int epi, int act, int vec, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s)
{
    constexpr TileGeom G = kTilesNarrow[1];
    VK_PLAIN_TILE_BODY()
}
