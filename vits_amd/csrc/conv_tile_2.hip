#include "conv_launch.hpp"
int vk_launch_tile2(int epi, int act, int vec, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s)
{
    constexpr TileGeom G = kTiles[2];
    VK_PLAIN_TILE_BODY()
}
