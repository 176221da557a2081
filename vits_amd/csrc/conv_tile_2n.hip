#include "conv_launch.hpp"
int vk_launch_tile2n(int epi, int act, int vec, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s)
{
    constexpr TileGeom G = kTilesNarrow[2];
    VK_PLAIN_TILE_BODY()
}
