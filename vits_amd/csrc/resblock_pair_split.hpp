// The fused resblock pair (see resblock_pair.hpp) in sliced-fp32 arithmetic (conv_split.hpp):
//
//   h   = conv1(lrelu(x))        taps k, dilation d
//   out = conv2(lrelu(h)) + x    taps k, dilation 1      (/root/reference/module/model_component/decoder.py:49-58)
//
// Every operand of a matrix instruction is an fp32 value cut into three bf16 slices, a product is
// the six largest slice products (conv_split.hpp has the error bound).  Built for 32 and 64
// channels, where two launches of the sliced convolution kernel move 20 bytes per tensor element
// (x, h twice, the residual, out) and at 32 channels are bound by that: here x is cut once while
// it is staged, h is cut once when it is handed over in LDS, and h never touches HBM.
//
// A workgroup owns every channel of a window of TN columns.  The image in LDS is channel-minor,
// the three slices of a column side by side: [column][slice][channel] in bf16, PITCH = 3 * 2 * C + 16
// bytes per column (an odd multiple of 16: conflict-free 16-byte fragment reads).  h is written
// over the x image once every wave has left the first convolution.  Weights are packed by
// packing.pack_conv_weight_split with 16-channel chunks: per 32-row tile and (chunk, tap) step three
// 16-byte fragments per lane (hi, mid, lo), streamed from L2 four steps ahead.  The arithmetic of
// an output element is that of two conv_split_kernel launches, operation for operation, so the
// results are the same bits.
#pragma once
#include "conv_split.hpp"
#include "resblock_pair.hpp"

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Diagnostic build only (-DVK_DIAG -DVK_PS_ABL=n, results invalid): 1 = the step loop fetches no weight
// fragments, 2 = it reads no B fragments from LDS.  (With both, a 32-channel 11-tap launch still takes
// 1090 of 1250 us: the loops are bound by the matrix pipe, at the clock it holds under bf16 load.)
#if defined(VK_DIAG) && defined(VK_PS_ABL)
constexpr int kPairSplitAbl = VK_PS_ABL;
#else
constexpr int kPairSplitAbl = 0;
#endif

template <int B_REP, int WM, int WN, int EPI>
__global__ void __launch_bounds__(64 * WM * WN, 2)
resblock_pair_split_kernel(const PairDev q)
{
    constexpr int NW  = WM * WN;
    constexpr int TNW = 32 * B_REP;
    constexpr int TN  = TNW * WN;
    constexpr int TNO = TN - VK_PAIR_OVERLAP;
    constexpr int NJ  = TN / 64 + 1;            // 64-column pieces of the staged window (halo <= 64)
    constexpr int C   = 32 * WM;
    constexpr int NKS = C / 16;                 // 16-channel chunks: one matrix instruction per slice product each
    constexpr int PLANE = C * 2;                // bytes of one slice of a column
    constexpr int PITCH = 3 * PLANE + 16;
    constexpr int CG  = C / 8;                  // staging: groups of 8 channels
    constexpr int NP  = NJ * CG;                // staging units (8 channels x 64 columns)
    constexpr int NI  = (NP + NW - 1) / NW;
    static_assert(TN % 64 == 0, "time tile must be a multiple of the wave width");

    extern __shared__ __attribute__((aligned(16))) char lds8[];

    const ConvDev& d = q.c;
    const vk_conv_args& p = d.a;
    const int lane = threadIdx.x & 63;
    const int w    = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm   = w % WM;
    const int wn   = w / WM;
    const int half = lane >> 5;
    const int l31  = lane & 31;
    const int b    = blockIdx.z;
    const int t0   = blockIdx.x * TNO;
    const int L    = p.length;
    const int h2   = (p.taps - 1) >> 1;

    vk_stamp(d, 0);         // diagnostic build: phase time stamps (tools/pair_stamps.py)
    constexpr int RING = 4;
    f32x16 acc[B_REP];
    const bf16x8* ap;
    bf16x8 a[RING][3];
    auto fill = [&](const void* wpacked, int groups) {
        ap = reinterpret_cast<const bf16x8*>(wpacked) + ((int64_t)wm * groups) * 64 + lane;
#pragma unroll
        for (int i = 0; i < RING; ++i)
#pragma unroll
            for (int s = 0; s < 3; ++s) a[i][s] = ap[64 * (3 * i + s)];
    };
    fill(p.w, d.groups_per_mt);

    // ---- the x window: loaded, activated, cut into slices, stored channel-minor ----
    {
        const __amdgpu_buffer_rsrc_t rx = vk_rsrc(p.x + (int64_t)b * p.x_bstride, d.x_bytes);
        float st[NI][8];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int pid = w + NW * i;                         // wave-uniform
            const int cg = pid / NJ, piece = pid - cg * NJ;
            const int col = piece * 64 + lane;
            const int t = t0 - p.pad + (col < d.xw ? col : d.xw - 1);
            const int voff = 4 * (t < 0 ? 0 : (t >= L ? L - 1 : t));
            if (pid < NP) {
#pragma unroll
                for (int c8 = 0; c8 < 8; ++c8) st[i][c8] = vk_bld(rx, voff, (cg * 8 + c8) * (int)p.x_cstride * 4);
            }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int pid = w + NW * i;
            const int cg = pid / NJ, piece = pid - cg * NJ;
            const int col = piece * 64 + lane;
            const int t = t0 - p.pad + col;
            const unsigned keep = ((t >= 0) && (t < L) && (col < d.xw)) ? 0xffffffffu : 0u;
            if (pid < NP) {
                u32x4 oh, om, ol;
#pragma unroll
                for (int c8 = 0; c8 < 8; c8 += 2) {
                    // (a mask, not a select on the loaded value: no branch around the activation)
                    const float v0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, vk_lrelu(st[i][c8], p.in_slope)) & keep);
                    const float v1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, vk_lrelu(st[i][c8 + 1], p.in_slope)) & keep);
                    unsigned h, m, l;
                    vk_split3_pair(v0, v1, h, m, l);
                    oh[c8 >> 1] = h; om[c8 >> 1] = m; ol[c8 >> 1] = l;
                }
                char* dst = lds8 + col * PITCH + cg * 16;
                *reinterpret_cast<u32x4*>(dst) = oh;
                *reinterpret_cast<u32x4*>(dst + PLANE) = om;
                *reinterpret_cast<u32x4*>(dst + 2 * PLANE) = ol;
            }
        }
    }

    // One convolution of this wave's 32 rows x TNW columns out of the image: (chunk, tap) steps in
    // the order of the weight stream, three fragments (hi, mid, lo) per step.  The fragments of the
    // next RING steps are in flight: `fill` requests the first ones (before the phase that precedes
    // the convolution, so that they arrive under it), and a slot is refilled with the step RING
    // further on as soon as its matrix instructions have been issued.
    auto run_conv = [&](const float* bias, const char* base, int tap_step) {
        {
            float bv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bv[r] = bias ? bias[wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] : 0.0f;
#pragma unroll
            for (int n = 0; n < B_REP; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[n][r] = bv[r];
        }
        const char* bt = base;
        int tap = 0, ks = 0;
        bf16x8 bf[3][B_REP];                    // the B fragments of the step about to run (read a step ahead)
        auto read_b = [&](bf16x8 (&dst)[3][B_REP], const char* at) __attribute__((always_inline)) {
#pragma unroll
            for (int n = 0; n < B_REP; ++n)
#pragma unroll
                for (int s = 0; s < 3; ++s) dst[s][n] = *reinterpret_cast<const bf16x8*>(at + s * PLANE + n * 32 * PITCH);
        };
        read_b(bf, bt);
        auto step = [&](auto slot_tag) __attribute__((always_inline)) {
            constexpr int S = decltype(slot_tag)::value;
            if (++tap == p.taps) {
                tap = 0;
                ++ks;
                bt = base + (ks < NKS ? ks : 0) * 32;      // after the last step: a harmless re-read
            } else {
                bt += tap_step;
            }
            bf16x8 bn[3][B_REP];
            if (kPairSplitAbl & 2) {
#pragma unroll
                for (int n = 0; n < B_REP; ++n)
#pragma unroll
                    for (int s = 0; s < 3; ++s) bn[s][n] = bf[s][n];
            } else {
                read_b(bn, bt);
            }
            // The six slice products of a column tile form a dependent chain (smallest terms first, as
            // conv_split_kernel); the chains of the B_REP column tiles are interleaved, so that a matrix
            // instruction never follows the one whose result it accumulates onto.
            constexpr int kA[6] = {2, 0, 1, 1, 0, 0};      // lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
            constexpr int kB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int n = 0; n < B_REP; ++n)
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[S][kA[t]], bf[kB[t]][n], acc[n], 0, 0, 0);
            // (past the end of the stream: the next tile's, or the zero groups that end a packed tensor)
            if (!(kPairSplitAbl & 1)) {
#pragma unroll
                for (int s = 0; s < 3; ++s) a[S][s] = ap[64 * (3 * (RING + S) + s)];
            }
#pragma unroll
            for (int n = 0; n < B_REP; ++n)
#pragma unroll
                for (int s = 0; s < 3; ++s) bf[s][n] = bn[s][n];
            // issue order: the next step's LDS reads, this step's matrix instructions, the weight refill
            __builtin_amdgcn_sched_group_barrier(0x100, 3 * B_REP, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 6 * B_REP, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);
        };
        // NKS is even, so the steps come in pairs: whole rings first, then at most one pair
        static_assert(NKS % 2 == 0 && RING == 4, "the step loop below");
        const int steps = NKS * p.taps;
        int i = 0;
#pragma unroll 1
        for (; i + RING <= steps; i += RING) {
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{});
            ap += 64 * 3 * RING;
        }
        if (i < steps) {
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
        }
    };

    __syncthreads();
    vk_stamp(d, 1);
    // ------------------------------ first convolution ------------------------------
    const int lcol = (wn * TNW + l31) * PITCH + half * 16;
    run_conv(p.bias, lds8 + lcol, p.dilation * PITCH);

    vk_stamp(d, 2);
    fill(q.w2, q.groups2);
    __syncthreads();                    // every wave is done reading the x image
    {
        // register r of a lane is channel 32*wm + (r & 3) + 8*(r >> 2) + 4*half of its column: four
        // consecutive registers are four consecutive channels, 8 bytes of each slice
        char* hcol = lds8 + (wn * TNW + l31) * PITCH + (wm * 32 + 4 * half) * 2;
#pragma unroll
        for (int n = 0; n < B_REP; ++n) {
            const int t = t0 - h2 + wn * TNW + 32 * n + l31;
            const unsigned keep = ((t >= 0) && (t < L)) ? 0xffffffffu : 0u;     // zero outside the sequence: conv2's padding
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned hh[2], mm[2], ll[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float v0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, vk_lrelu(acc[n][4 * g + 2 * e], p.in_slope)) & keep);
                    const float v1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, vk_lrelu(acc[n][4 * g + 2 * e + 1], p.in_slope)) & keep);
                    vk_split3_pair(v0, v1, hh[e], mm[e], ll[e]);
                }
                char* dst = hcol + n * 32 * PITCH + g * 16;
                *reinterpret_cast<u32x2*>(dst) = u32x2{hh[0], hh[1]};
                *reinterpret_cast<u32x2*>(dst + PLANE) = u32x2{mm[0], mm[1]};
                *reinterpret_cast<u32x2*>(dst + 2 * PLANE) = u32x2{ll[0], ll[1]};
            }
        }
    }
    __syncthreads();
    vk_stamp(d, 3);

    // ------------------------------ second convolution ------------------------------
    run_conv(q.bias2, lds8 + lcol, PITCH);

    vk_stamp(d, 5);
    pair_epilogue<B_REP, EPI, TNO>(d, acc, b, t0, wm, wn, half, l31);
    vk_stamp(d, 6);
}

// Four waves of 32 rows x 64 columns, two workgroups per CU.  Measured against these (16 x 862 frames,
// 3 / 7 / 11 taps): eight waves of 32 x 32 (twice the weight fragments per matrix instruction)
// 523 / 845 / 1146 us against 467 / 754 / 1014 at 32 channels; 128-column windows at 32 channels, four
// workgroups per CU: 514 / 821 / 1094 us.
struct PairGeomSplit { int b_rep, wm, wn; };
static constexpr PairGeomSplit kPairSplit32 = {2, 1, 4};        // 256-column window, 66.5 KB of LDS
static constexpr PairGeomSplit kPairSplit64 = {2, 2, 2};        // 128-column window, 76.8 KB

static inline size_t pair_split_lds(const PairGeomSplit& G)
{
    return (size_t)(32 * G.b_rep * G.wn + 64) * (3 * 2 * 32 * G.wm + 16);
}

// Returns hipSuccess / a HIP error, or -1 when the combination is not built.
int vk_launch_pair_split(int channels, int epi, const PairDev& q, dim3 grid, size_t lds, hipStream_t s);
