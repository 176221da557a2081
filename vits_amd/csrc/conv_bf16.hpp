// The same implicit-GEMM convolution with bf16 matrix-core inputs and fp32 accumulation
// (v_mfma_f32_32x32x16_bf16).  Activations stay fp32 in HBM; they are activated (leaky-ReLU)
// and rounded to bf16 (round to nearest even) while the tile is staged into LDS, weights are
// pre-rounded and pre-packed on the host.  Everything after the accumulators - bias, residuals,
// gates, coupling updates, stores - is the fp32 epilogue of conv_mfma.hpp.
//
// One matrix instruction contracts 16 input channels: lane l supplies 8 consecutive channels
// (8 * (l >> 5) ..) of row / column l & 31.  The staged tile is therefore channel-minor,
//   lds[column][channel within chunk]   (bf16, PITCH bytes per column),
// so a B fragment is one 16-byte LDS read per lane and a tap is a column offset.
// Staging transposes on the way in: a lane owns one column, loads 8 channels of it (each load
// is contiguous across the wave) and writes them with one 16-byte LDS store.
#pragma once
#include "conv_mfma.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Staging of one K-chunk for the bf16 kernels.  Work unit = (group of 8 channels, 64-column
// piece); unit id = w + NW * i for wave w.  A lane owns one column of its unit: 8 coalesced
// 4-byte loads (one per channel), activation, rounding, one 16-byte LDS store.
template <int NW, int NJ, int CK, int PITCH, bool ACT>
struct Bf16Stager {
    static constexpr int CG = CK / 8;
    static constexpr int NP = NJ * CG;
    static constexpr int NI = (NP + NW - 1) / NW;
    int  u_voff[NI];        // clamped byte offset of this lane's column within a row of x
    int  u_lds[NI];         // byte offset of this lane's 16-byte slot within a staging buffer
    int  u_cg[NI];          // channel group (wave-uniform)
    bool u_in[NI];          // this lane's column lies inside [0, L)
    bool u_on[NI];          // wave-uniform: the unit exists and its piece overlaps the staged width
    float st[NI][8];
    float u_mk[NI];         // input mask of this lane's column (only read when the launch has one)
    __amdgpu_buffer_rsrc_t rx;

    __device__ __forceinline__ void init(const ConvDev& d, const float* xb, int t0, int lane, int w, int b = 0)
    {
        const int L = d.a.length;
        rx = vk_rsrc(xb, d.x_bytes);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int pid = w + NW * i;
            const int cg = pid / NJ, piece = pid - cg * NJ;
            const int col = piece * 64 + lane;
            const int t = t0 - d.a.pad + (col < d.xw ? col : d.xw - 1);   // lanes past the staged width re-read its last column
            u_cg[i] = cg;
            u_on[i] = (pid < NP) && (piece * 64 < d.xw);
            u_voff[i] = 4 * (t < 0 ? 0 : (t >= L ? L - 1 : t));
            u_in[i] = (t >= 0) && (t < L);
            u_lds[i] = col * PITCH + cg * 16;
            if (d.a.in_mask != nullptr) u_mk[i] = d.a.in_mask[(int64_t)b * L + (u_voff[i] >> 2)];     // wave-uniform branch
        }
    }
    __device__ __forceinline__ void issue(const ConvDev& d, int chunk)
    {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (u_on[i]) {                                           // wave-uniform
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
                bf16x8 o;
#pragma unroll
                for (int c8 = 0; c8 < 8; c8 += 2) {
                    const int c = chunk * CK + u_cg[i] * 8 + c8;
                    float v0 = (u_in[i] && c < d.a.c_in) ? st[i][c8] : 0.0f;
                    float v1 = (u_in[i] && c + 1 < d.a.c_in) ? st[i][c8 + 1] : 0.0f;
                    if (d.a.in_mask != nullptr) { v0 *= u_mk[i]; v1 *= u_mk[i]; }
                    if (ACT) { v0 = vk_lrelu(v0, d.a.in_slope); v1 = vk_lrelu(v1, d.a.in_slope); }
                    const f32x2 pr = {v0, v1};
                    const bf16x2 q = __builtin_convertvector(pr, bf16x2);
                    o[c8] = q[0]; o[c8 + 1] = q[1];
                }
                *reinterpret_cast<bf16x8*>(buf + u_lds[i]) = o;
            }
        }
    }
};

// Taps of weight fragments in flight ahead of the matrix instructions that consume them.  A tap
// of a 32-channel chunk is only 2 * A_REP * B_REP matrix instructions (250-500 cycles), about what
// an L2 hit takes on a busy chip, so fragments are requested two taps ahead.
#ifndef VK_BF16_PF
#define VK_BF16_PF 2
#endif
constexpr int VK_PF = VK_BF16_PF;
// Diagnostic build only (-DVK_DIAG -DVK_BF16_ABL=n, results invalid): 1 = the tap loop fetches no
// weight fragments, 2 = it reads no B fragments from LDS.
#if defined(VK_DIAG) && defined(VK_BF16_ABL)
constexpr int kBf16Abl = VK_BF16_ABL;
#else
constexpr int kBf16Abl = 0;
#endif

// The weight fragments of a wave: a[i] holds tap i of the stream `aptr` points into (i < PF).
template <int KS, int A_REP, int PF = VK_PF>
__device__ __forceinline__ void bf16_wload(bf16x8 (&a)[PF][KS][A_REP], const bf16x8* (&aptr)[A_REP])
{
#pragma unroll
    for (int i = 0; i < PF; ++i)
#pragma unroll
        for (int k = 0; k < KS; ++k)
#pragma unroll
            for (int m = 0; m < A_REP; ++m) a[i][k][m] = aptr[m][64 * (KS * i + k)];
}

// The k-loop over one resident chunk (KS matrix instructions per tap).  B fragments of the next
// k-step are read before the current k-step's MFMAs are issued; `base` is this lane's fragment
// address for tap 0 / k-step 0 / column tile 0.  Weights: the loop runs PF taps per iteration,
// tap i out of a[i]; as soon as the matrix instructions of a k-step have been issued its registers
// are refilled in place with the same k-step PF taps further on, so every fragment has PF whole
// taps to arrive and the loop copies no register.  A remainder of taps % PF taps follows, after
// which the fragments are moved back so that a[0] is again the next tap (once per chunk).  The
// stream runs on linearly past the chunk (next chunk, next row tile or the zero groups that end a
// packed tensor); with `use_cont` (set for a convolution's last chunk) it continues in the stream
// `cont` instead (the next convolution's weights), whose first PF taps a[] then holds on return.
template <int A_REP, int B_REP, int KS, int PITCH, int PF = VK_PF>
__device__ __forceinline__ void bf16_taps(f32x16 (&acc)[A_REP][B_REP], bf16x8 (&a)[PF][KS][A_REP],
                                          const bf16x8* (&aptr)[A_REP], const char* base, int taps, int tap_step,
                                          const bf16x8* (&cont)[A_REP], bool use_cont)
{
    const char* bt = base;
    bf16x8 bf[B_REP];
#pragma unroll
    for (int n = 0; n < B_REP; ++n) bf[n] = *reinterpret_cast<const bf16x8*>(bt + n * 32 * PITCH);

    // one tap out of ring slot S; aptr points at this tap's fragments
    auto one_tap = [&](auto slot_tag, int tap) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_tag)::value;
        const char* bt_next = (tap + 1 < taps) ? bt + tap_step : base;     // chunk end: harmless re-read
        const int ahead = tap + PF - taps;              // >= 0: what this slot holds next lies past the chunk
        const bf16x8* src[A_REP];
#pragma unroll
        for (int m = 0; m < A_REP; ++m)
            src[m] = (use_cont && ahead >= 0) ? cont[m] + 64 * KS * ahead : aptr[m] + 64 * KS * PF;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            bf16x8 bn[B_REP];
#pragma unroll
            for (int n = 0; n < B_REP; ++n)
                bn[n] = (kBf16Abl & 2) ? bf[n]
                      : (k + 1 < KS) ? *reinterpret_cast<const bf16x8*>(bt + n * 32 * PITCH + (k + 1) * 32)
                                     : *reinterpret_cast<const bf16x8*>(bt_next + n * 32 * PITCH);
#pragma unroll
            for (int m = 0; m < A_REP; ++m)
#pragma unroll
                for (int n = 0; n < B_REP; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[S][k][m], bf[n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < B_REP; ++n) bf[n] = bn[n];
            if (!(kBf16Abl & 1)) {
#pragma unroll
                for (int m = 0; m < A_REP; ++m) a[S][k][m] = src[m][64 * k];
            }
        }
        bt = bt_next;
#pragma unroll
        for (int m = 0; m < A_REP; ++m) aptr[m] += 64 * KS;
    };

    int tap = 0;
#pragma unroll 1
    for (; tap + PF <= taps; tap += PF) {
        one_tap(std::integral_constant<int, 0>{}, tap);
        if constexpr (PF > 1) one_tap(std::integral_constant<int, 1>{}, tap + 1);
        if constexpr (PF > 2) one_tap(std::integral_constant<int, 2>{}, tap + 2);
    }
    if constexpr (PF > 1) {
        static_assert(PF == 2, "the remainder below is written for a ring of two");
        if (tap < taps) {                               // one tap left over: use slot 0, then swap the slots
            one_tap(std::integral_constant<int, 0>{}, tap);
#pragma unroll
            for (int k = 0; k < KS; ++k)
#pragma unroll
                for (int m = 0; m < A_REP; ++m) {
                    const bf16x8 t = a[0][k][m];
                    a[0][k][m] = a[1][k][m];
                    a[1][k][m] = t;
                }
        }
    }
}

template <int A_REP, int B_REP, int KS, int PITCH, int PF = VK_PF>
__device__ __forceinline__ void bf16_taps(f32x16 (&acc)[A_REP][B_REP], bf16x8 (&a)[PF][KS][A_REP],
                                          const bf16x8* (&aptr)[A_REP], const char* base, int taps, int tap_step)
{
    bf16_taps<A_REP, B_REP, KS, PITCH, PF>(acc, a, aptr, base, taps, tap_step, aptr, false);
}

template <int A_REP, int B_REP, int WM, int WN, int CK, int EPI, bool ACT>
__global__ void __launch_bounds__(64 * WM * WN, 2)
conv_bf16_kernel(const ConvDev d)
{
    constexpr int NW  = WM * WN;
    constexpr int TNW = 32 * B_REP;
    constexpr int TN  = TNW * WN;
    constexpr int NJ  = TN / 64 + 1;            // 64-column pieces per staged tile (halo <= 64)
    constexpr int KS  = CK / 16;                // matrix instructions per tap per chunk
    constexpr int PITCH = CK * 2 + 16;          // bytes between staged columns (odd multiple of 16: conflict-free)
    constexpr int BUF = (TN + 64) * PITCH;      // bytes per staging buffer
    static_assert(CK % 16 == 0, "a matrix instruction contracts 16 channels");
    static_assert(TN % 64 == 0, "time tile must be a multiple of the wave width");
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
    int b = blockIdx.z;
    int tile = blockIdx.x, rowblk = blockIdx.y;
    if (d.xcd_map) {                                   // see conv_mfma_kernel: row blocks of a time tile share an XCD
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        rowblk = slot % d.n_rowblk;
        const int gt = (slot / d.n_rowblk) * 8 + xcd;
        if (gt >= d.n_tiles * p.batch) return;
        b = gt / d.n_tiles;
        tile = gt - b * d.n_tiles;
    }
    const int t0   = tile * TN;
    const int mt0  = (rowblk * WM + wm) * A_REP;
    const int L    = p.length;

    const float* xb = p.x + (int64_t)b * p.x_bstride;
    Bf16Stager<NW, NJ, CK, PITCH, ACT> stg;
    stg.init(d, xb, t0, lane, w, b);
    stg.issue(d, 0);

    // ---- weight stream: one 16-byte fragment per lane per matrix instruction ----
    const bf16x8* aptr[A_REP];
#pragma unroll
    for (int m = 0; m < A_REP; ++m)
        aptr[m] = reinterpret_cast<const bf16x8*>(p.w) + ((int64_t)(mt0 + m) * d.groups_per_mt) * 64 + lane;
    bf16x8 a[VK_PF][KS][A_REP];
    bf16_wload<KS, A_REP>(a, aptr);

    f32x16 acc[A_REP][B_REP];
    {
        // accumulators start at the bias of their row: register r of a lane is row
        // (r & 3) + 8 * (r >> 2) + 4 * half of its 32-row tile
        const float* biasb = p.bias ? vk_bias_of(p, b) : nullptr;
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
        bf16_taps<A_REP, B_REP, KS, PITCH>(acc, a, aptr, buf + bcol, p.taps, tap_step);
    }

    conv_epilogue<A_REP, B_REP, EPI>(d, acc, b, t0, mt0, wn, half, l31);
}

// channels per staged chunk of the bf16 kernels, by tile family
static constexpr int kBf16Ck[VK_TILE_COUNT] = {32, 32, 16, 32};

int vk_launch_bf16_tile0(int epi, int act, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s);
int vk_launch_bf16_tile1(int epi, int act, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s);
int vk_launch_bf16_tile2(int epi, int act, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s);
int vk_launch_bf16_tile3(int epi, int act, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s);
int vk_launch_bf16_tile0n(int epi, int act, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s);
int vk_launch_bf16_tile1n(int epi, int act, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s);
int vk_launch_bf16_tile2n(int epi, int act, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s);
int vk_launch_bf16_tile3n(int epi, int act, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s);
