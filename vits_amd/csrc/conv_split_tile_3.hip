// bf16-sliced fp32 instantiations of tile family 3.
#include "conv_launch.hpp"
#include "conv_split.hpp"

int vk_launch_split_tile3(int epi, int act, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s)
{
    constexpr TileGeom G = kSplitTiles[3];
    constexpr int CKB = VK_SPLIT_CK;
    VK_SPLIT_CASE(VK_EPI_GATE, 0)
    VK_SPLIT_CASE(VK_EPI_SAMPLE, 0)
    return -1;
}
