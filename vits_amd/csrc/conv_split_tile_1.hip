// bf16-sliced fp32 instantiations of tile family 1.
#include "conv_launch.hpp"
#include "conv_split.hpp"

int vk_launch_split_tile1(int epi, int act, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s)
{
    constexpr TileGeom G = kSplitTiles[1];
    constexpr int CKB = VK_SPLIT_CK;
    VK_SPLIT_CASE(VK_EPI_STORE, 0)
    VK_SPLIT_CASE(VK_EPI_STORE, 1)
    VK_SPLIT_CASE(VK_EPI_RESID, 1)
    VK_SPLIT_CASE(VK_EPI_RESID_ACC, 1)
    VK_SPLIT_CASE(VK_EPI_RESID_AVG, 1)
    VK_SPLIT_CASE(VK_EPI_UPSAMPLE, 1)
    VK_SPLIT_CASE(VK_EPI_WN_MID, 0)
    VK_SPLIT_CASE(VK_EPI_WN_LAST, 0)
    VK_SPLIT_CASE(VK_EPI_COUPLE_REV, 0)
    VK_SPLIT_CASE(VK_EPI_COUPLE_FWD, 0)
    return -1;
}
