// Per-tile launch tables for conv_mfma_kernel.  Each tile family is instantiated in a
// translation unit of its own (conv_tile_*.hip) so the library builds in parallel.
#pragma once
#include "conv_mfma.hpp"

struct TileGeom { int a_rep, b_rep, wm, wn, ck; };

// VK_TILE_M128, VK_TILE_M64, VK_TILE_M32, VK_TILE_PAIR
static constexpr TileGeom kTiles[VK_TILE_COUNT] = {
    {1, 4, 4, 1, 16},
    {1, 4, 2, 2, 16},
    {1, 4, 1, 4, 16},
    {2, 2, 2, 2, 16},
};

// The same families with half the time tile (same rows, same K-chunk, so the same packed
// weights): twice the workgroups for launches that would otherwise leave CUs without work.
static constexpr TileGeom kTilesNarrow[VK_TILE_COUNT] = {
    {1, 2, 4, 1, 16},
    {1, 2, 2, 2, 16},
    {1, 2, 1, 4, 16},
    {2, 1, 2, 2, 16},
};

// Returns hipSuccess / a HIP error, or -1 when (epilogue, act_in) is not built for the tile.
int vk_launch_tile0(int epi, int act, int vec, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s);
int vk_launch_tile1(int epi, int act, int vec, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s);
int vk_launch_tile2(int epi, int act, int vec, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s);
int vk_launch_tile3(int epi, int act, int vec, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s);
int vk_launch_tile0n(int epi, int act, int vec, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s);
int vk_launch_tile1n(int epi, int act, int vec, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s);
int vk_launch_tile2n(int epi, int act, int vec, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s);
int vk_launch_tile3n(int epi, int act, int vec, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s);

#define VK_CONV_CASE(EPI, ACT, VEC)                                                              \
    if (epi == (EPI) && act == (ACT) && vec == (VEC)) {                                          \
        auto k = conv_mfma_kernel<G.a_rep, G.b_rep, G.wm, G.wn, G.ck, (EPI), (ACT) != 0, (VEC) != 0>; \
        static std::atomic<unsigned long long> lds_raised{0};   /* bit d: limit raised on device d */\
        if (hipError_t e = vk_allow_lds(k, lds, lds_raised); e != hipSuccess) return (int)e;     \
        hipLaunchKernelGGL(k, grid, dim3(64 * G.wm * G.wn), lds, s, d);                          \
        return (int)hipGetLastError();                                                           \
    }

// (epilogue, act_in, vector staging) combinations that are built for the plain tiles
#define VK_PLAIN_TILE_BODY()                     \
    VK_CONV_CASE(VK_EPI_STORE, 0, 0)             \
    VK_CONV_CASE(VK_EPI_STORE, 1, 0)             \
    VK_CONV_CASE(VK_EPI_STORE, 1, 1)             \
    VK_CONV_CASE(VK_EPI_RESID, 1, 0)             \
    VK_CONV_CASE(VK_EPI_RESID, 1, 1)             \
    VK_CONV_CASE(VK_EPI_RESID_ACC, 1, 0)         \
    VK_CONV_CASE(VK_EPI_RESID_ACC, 1, 1)         \
    VK_CONV_CASE(VK_EPI_RESID_AVG, 1, 0)         \
    VK_CONV_CASE(VK_EPI_RESID_AVG, 1, 1)         \
    VK_CONV_CASE(VK_EPI_UPSAMPLE, 1, 0)          \
    VK_CONV_CASE(VK_EPI_UPSAMPLE, 1, 1)          \
    VK_CONV_CASE(VK_EPI_WN_MID, 0, 0)            \
    VK_CONV_CASE(VK_EPI_WN_LAST, 0, 0)           \
    VK_CONV_CASE(VK_EPI_COUPLE_REV, 0, 0)        \
    VK_CONV_CASE(VK_EPI_COUPLE_FWD, 0, 0)        \
    return -1;
