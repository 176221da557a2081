// fp32-accurate convolution on the bf16 matrix cores.
//
// Every fp32 operand is cut into three bf16 slices,
//   v = hi + mid + lo,   hi = bf16(v), mid = bf16(v - hi), lo = v - hi - mid
// (round to nearest even; the remainders are exact, |mid| <= 2^-9 |v|, |lo| <= 2^-18 |v|, and lo
// fits bf16 exactly), and a product is evaluated as the six largest slice products
//   x*w ~= hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid      (each exact in fp32),
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The dropped terms (mid*lo, lo*mid, lo*lo)
// are below 2^-26 of the product, less than the rounding of an fp32 multiply.
// Six bf16 matrix instructions contract 16 channels in 192 cycles; the fp32 instruction
// (v_mfma_f32_32x32x2_f32, 64 cycles for 2 channels) needs 512.
//
// Activations are split while they are staged into LDS: channel-minor like conv_bf16.hpp, the
// three slices of a column's chunk side by side (PITCH = 3 * CK * 2 + 16 bytes per column);
// weights are split and packed on the host.  Epilogues are the shared fp32
// ones.
#pragma once
#include "conv_bf16.hpp"
#include "conv_launch.hpp"

// Two values -> their slices, each slice pair already packed as one dword of two bf16 (first
// value in the low half).  A slice is what is left, rounded to nearest even in bf16; the
// remainders are exact in fp32, and the third slice is exact (at most 8 significant bits left).
__device__ __forceinline__ void vk_split3_pair(float v0, float v1, unsigned& h, unsigned& m, unsigned& l)
{
    const f32x2 p0 = {v0, v1};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(p0, bf16x2));
    const float r0 = v0 - __builtin_bit_cast(float, h << 16);
    const float r1 = v1 - __builtin_bit_cast(float, h & 0xffff0000u);
    const f32x2 p1 = {r0, r1};
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(p1, bf16x2));
    const float q0 = r0 - __builtin_bit_cast(float, m << 16);
    const float q1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
    const f32x2 p2 = {q0, q1};
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(p2, bf16x2));
}

template <int NW, int NJ, int CK, int PITCH, int PLANE, bool ACT>
struct SplitStager {
    static constexpr int CG = CK / 8;
    static constexpr int NP = NJ * CG;
    static constexpr int NI = (NP + NW - 1) / NW;
    int  u_voff[NI], u_lds[NI], u_cg[NI];
    bool u_in[NI], u_on[NI];
    float st[NI][8];
    __amdgpu_buffer_rsrc_t rx;

    __device__ __forceinline__ void init(const ConvDev& d, const float* xb, int t0, int lane, int w)
    {
        const int L = d.a.length;
        rx = vk_rsrc(xb, d.x_bytes);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int pid = w + NW * i;
            const int cg = pid / NJ, piece = pid - cg * NJ;
            const int col = piece * 64 + lane;
            const int t = t0 - d.a.pad + (col < d.xw ? col : d.xw - 1);
            u_cg[i] = cg;
            u_on[i] = (pid < NP) && (piece * 64 < d.xw);
            u_voff[i] = 4 * (t < 0 ? 0 : (t >= L ? L - 1 : t));
            u_in[i] = (t >= 0) && (t < L);
            u_lds[i] = col * PITCH + cg * 16;
        }
    }
    __device__ __forceinline__ void issue(const ConvDev& d, int chunk)
    {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (u_on[i]) {
#pragma unroll
                for (int c8 = 0; c8 < 8; ++c8) {
                    const int c = chunk * CK + u_cg[i] * 8 + c8;
                    const int soff = (int)((c < d.a.c_in ? c : 0) * d.a.x_cstride * 4);
                    st[i][c8] = vk_bld(rx, u_voff[i], soff);
                }
            }
        }
    }
    __device__ __forceinline__ void store(const ConvDev& d, int chunk, char* buf)
    {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (u_on[i]) {
                u32x4 oh, om, ol;
#pragma unroll
                for (int c8 = 0; c8 < 8; c8 += 2) {
                    const int c = chunk * CK + u_cg[i] * 8 + c8;
                    float v0 = (u_in[i] && c < d.a.c_in) ? st[i][c8] : 0.0f;
                    float v1 = (u_in[i] && c + 1 < d.a.c_in) ? st[i][c8 + 1] : 0.0f;
                    if (ACT) { v0 = vk_lrelu(v0, d.a.in_slope); v1 = vk_lrelu(v1, d.a.in_slope); }
                    unsigned h, m, l;
                    vk_split3_pair(v0, v1, h, m, l);
                    oh[c8 >> 1] = h; om[c8 >> 1] = m; ol[c8 >> 1] = l;
                }
                *reinterpret_cast<u32x4*>(buf + u_lds[i]) = oh;
                *reinterpret_cast<u32x4*>(buf + PLANE + u_lds[i]) = om;
                *reinterpret_cast<u32x4*>(buf + 2 * PLANE + u_lds[i]) = ol;
            }
        }
    }
};

template <int A_REP, int B_REP, int WM, int WN, int CK, int EPI, bool ACT>
__global__ void __launch_bounds__(64 * WM * WN, 2)
conv_split_kernel(const ConvDev d)
{
    constexpr int NW  = WM * WN;
    constexpr int TNW = 32 * B_REP;
    constexpr int TN  = TNW * WN;
    constexpr int NJ  = TN / 64 + 1;
    constexpr int KS  = CK / 16;
    constexpr int PLANE = CK * 2;               // bytes between the slices of one column
    constexpr int PITCH = 3 * PLANE + 16;       // bytes between staged columns (odd multiple of 16: conflict-free)
    constexpr int BUF = (TN + 64) * PITCH;
    static_assert(CK % 16 == 0, "a matrix instruction contracts 16 channels");
    constexpr bool PAIR = (EPI == VK_EPI_GATE || EPI == VK_EPI_SAMPLE);
    static_assert(!PAIR || (A_REP % 2 == 0), "paired epilogues need an even number of row tiles per wave");

    extern __shared__ __attribute__((aligned(16))) char lds8[];

    const vk_conv_args& p = d.a;
    const int lane = threadIdx.x & 63;
    const int w    = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm   = w % WM;
    const int wn   = w / WM;
    const int half = lane >> 5;
    const int l31  = lane & 31;
    const int b    = blockIdx.z;
    int tile = blockIdx.x, rowblk = blockIdx.y;
    if (d.xcd_map) {                                   // see conv_mfma_kernel: row blocks of a time tile share an XCD
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        rowblk = slot % d.n_rowblk;
        tile = (slot / d.n_rowblk) * 8 + xcd;
        if (tile >= d.n_tiles) return;
    }
    const int t0   = tile * TN;
    const int mt0  = (rowblk * WM + wm) * A_REP;

    const float* xb = p.x + (int64_t)b * p.x_bstride;
    SplitStager<NW, NJ, CK, PITCH, PLANE, ACT> stg;
    stg.init(d, xb, t0, lane, w);
    stg.issue(d, 0);

    // ---- weight stream: per (tap, k-step) three 16-byte fragments per lane: hi, mid, lo ----
    const bf16x8* aptr[A_REP];
#pragma unroll
    for (int m = 0; m < A_REP; ++m)
        aptr[m] = reinterpret_cast<const bf16x8*>(p.w) + ((int64_t)(mt0 + m) * d.groups_per_mt) * 64 + lane;
    bf16x8 a[KS][3][A_REP];
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int m = 0; m < A_REP; ++m) a[k][s][m] = aptr[m][64 * (3 * k + s)];

    f32x16 acc[A_REP][B_REP];
    {
        // accumulators start at the bias of their row: register r of a lane is row
        // (r & 3) + 8 * (r >> 2) + 4 * half of its 32-row tile
        const float* biasb = p.bias ? p.bias + (int64_t)b * p.bias_bstride : nullptr;
        const int rows_total_b = PAIR ? 2 * p.hidden : p.c_out;
#pragma unroll
        for (int m = 0; m < A_REP; ++m) {
            float bv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int prow = (mt0 + m) * 32 + i;                                     // packed row
                int brow = prow;                                                         // bias index
                bool ok = prow < rows_total_b;
                if (PAIR) brow = ((m & 1) ? p.hidden : 0) + 32 * ((mt0 + m) >> 1) + i;
                if (EPI == VK_EPI_UPSAMPLE) {
                    int phase;
                    vk_ups_row(p, mt0 + m, i, brow, phase);
                    ok = brow < p.c_out && phase < p.up_stride;
                }
                bv[r] = 0.0f;
                if (biasb != nullptr) {                      // wave-uniform
                    const float raw = biasb[ok ? brow : 0];
                    bv[r] = ok ? raw : 0.0f;
                }
            }
#pragma unroll
            for (int n = 0; n < B_REP; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = bv[r];
        }
    }
    const int shift = (EPI == VK_EPI_UPSAMPLE) ? vk_ups_shift(p, mt0) : 0;     // upper phases read one column later
    const int bcol = (wn * TNW + l31 + shift) * PITCH + half * 16;
    const int tap_step = p.dilation * PITCH;

    for (int chunk = 0; chunk < d.nchunk; ++chunk) {
        char* buf = lds8 + (chunk & 1) * BUF;
        stg.store(d, chunk, buf);
        __syncthreads();
        if (chunk + 1 < d.nchunk) stg.issue(d, chunk + 1);

        // B fragments rotate in place: as soon as the six products of a column tile have been
        // issued, its three fragment registers are reloaded for the next k-step, so every LDS
        // read has the other column tiles' MFMAs to land behind.  Weights, which come from L2,
        // are fetched one tap ahead.
        const char* bt = buf + bcol;
        bf16x8 bf[3][B_REP];
#pragma unroll
        for (int n = 0; n < B_REP; ++n)
#pragma unroll
            for (int s = 0; s < 3; ++s) bf[s][n] = *reinterpret_cast<const bf16x8*>(bt + s * PLANE + n * 32 * PITCH);
#pragma unroll 1
        for (int tap = 0; tap < p.taps; ++tap) {
            const char* bt_next = (tap + 1 < p.taps) ? bt + tap_step : buf + bcol;   // chunk end: harmless re-read
            bf16x8 a_nxt[KS][3][A_REP];
#pragma unroll
            for (int k = 0; k < KS; ++k)
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int m = 0; m < A_REP; ++m) a_nxt[k][s][m] = aptr[m][64 * (3 * (KS + k) + s)];
#pragma unroll
            for (int k = 0; k < KS; ++k) {
#pragma unroll
                for (int n = 0; n < B_REP; ++n) {
                    // smallest terms first
#pragma unroll
                    for (int m = 0; m < A_REP; ++m) {
                        f32x16 c = acc[m][n];
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k][2][m], bf[0][n], c, 0, 0, 0);   // lo  * hi
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k][0][m], bf[2][n], c, 0, 0, 0);   // hi  * lo
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k][1][m], bf[1][n], c, 0, 0, 0);   // mid * mid
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k][1][m], bf[0][n], c, 0, 0, 0);   // mid * hi
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k][0][m], bf[1][n], c, 0, 0, 0);   // hi  * mid
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k][0][m], bf[0][n], c, 0, 0, 0);   // hi  * hi
                        acc[m][n] = c;
                    }
                    const char* nx = (k + 1 < KS) ? bt + (k + 1) * 32 : bt_next;
#pragma unroll
                    for (int s = 0; s < 3; ++s)
                        bf[s][n] = *reinterpret_cast<const bf16x8*>(nx + s * PLANE + n * 32 * PITCH);
                    __builtin_amdgcn_sched_group_barrier(0x008, 6 * A_REP, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                }
            }
            bt = bt_next;
#pragma unroll
            for (int m = 0; m < A_REP; ++m) aptr[m] += 64 * 3 * KS;
#pragma unroll
            for (int k = 0; k < KS; ++k)
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int m = 0; m < A_REP; ++m) a[k][s][m] = a_nxt[k][s][m];
        }
    }

    conv_epilogue<A_REP, B_REP, EPI>(d, acc, b, t0, mt0, wn, half, l31);
}

// channels per staged chunk of the split kernels, and their tile geometry: as kTiles except
// that the 32-row family takes a 256-column window (three slices per column in LDS)
constexpr int VK_SPLIT_CK = 16;
static constexpr TileGeom kSplitTiles[VK_TILE_COUNT] = {
    {1, 4, 4, 1, VK_SPLIT_CK},
    {1, 4, 2, 2, VK_SPLIT_CK},
    {1, 2, 1, 4, VK_SPLIT_CK},
    {2, 2, 2, 2, VK_SPLIT_CK},
};

int vk_launch_split_tile0(int epi, int act, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s);
int vk_launch_split_tile1(int epi, int act, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s);
int vk_launch_split_tile2(int epi, int act, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s);
int vk_launch_split_tile3(int epi, int act, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s);
