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
// weights are split and packed on the host.  Epilogues are the shared fp32 ones.
//
// Schedule.  At six matrix instructions per k-step a tile's matrix work is short next to the
// memory latency and to the cutting of its input, and workgroups that run the same program fall
// into step with each other, so none of them covers another's gaps.  Hence:
//   * a workgroup is persistent: the launch has as many workgroups as the GPU holds at once and
//     each walks a contiguous run of (batch item, time tile, row block) jobs;
//   * the work items of a run, (job, channel chunk), form one software pipeline that does not
//     drain at job boundaries: while item i is multiplied out of one LDS buffer, item i+1 is cut
//     into slices and written to the other, and the loads of item i+3 are issued into the
//     registers that item i+1 has just left (two items of loads in flight);
//   * the cutting (about 75 vector instructions per 8 channels x 64 columns) sits in the same
//     basic block as a tap's matrix instructions and is interleaved with them, so a wave keeps
//     the matrix pipe fed by itself.
#pragma once
#include <utility>
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

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <int N, typename F>
__device__ __forceinline__ void vk_static_for(F&& f)
{
    if constexpr (N > 0) {
        vk_static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// Position of a pipeline stage in the workgroup's run: which job, which chunk of it.
struct SplitCursor {
    int b, tile, split, rowblk, chunk;      // split: which share of the channel chunks (launches that split them)
};

// vector instructions of the cut scheduled behind each matrix instruction of a covering tap
#define VK_SPLIT_VALU_PER_MFMA 4

// Staging work unit = (group of 8 channels, 64-column piece); unit id = w + NW * i for wave w.
// A lane owns one column of its unit: 8 coalesced 4-byte loads (one per channel), then the
// activation, the three-way cut and three 16-byte LDS stores.
template <int NW, int NJ, int CK, int PITCH, int PLANE, int BUF, int TN, bool ACT>
struct SplitStager {
    static constexpr int CG = CK / 8;
    static constexpr int NP = NJ * CG;
    static constexpr int NI = (NP + NW - 1) / NW;
    int  u_col[NI];         // this lane's staged column in unit i (clamped to the staged width)
    int  u_lds[NI];         // byte offset of its 16-byte slot in a buffer (a unit that does not exist: a dump slot)
    int  u_cg[NI];          // channel group (wave-uniform)
    bool u_on[NI];          // wave-uniform: the unit exists and its piece overlaps the staged width
    float st[2][NI][8];     // two items in flight; set and unit are compile-time indices after unrolling
    float mk[2][NI];        // input mask of this lane's column of the item (only with an input mask)

    __device__ __forceinline__ void init(const ConvDev& d, int lane, int w)
    {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int pid = w + NW * i;
            const int cg = pid / NJ, piece = pid - cg * NJ;
            const int col = piece * 64 + lane;
            u_on[i] = (pid < NP) && (piece * 64 < d.xw);
            u_cg[i] = u_on[i] ? cg : 0;
            u_col[i] = col < d.xw ? col : d.xw - 1;            // lanes past the staged width re-read its last column
            u_lds[i] = u_on[i] ? col * PITCH + cg * 16 : 2 * BUF + lane * 16;
        }
    }
    template <int SET>
    __device__ __forceinline__ void issue(const ConvDev& d, const SplitCursor& c, int i)
    {
        if (!u_on[i]) return;                                   // wave-uniform
        if (kDiag && (d.dbg & 128)) return;                     // timing experiment: no staging loads
        const int L = d.a.length;
        const __amdgpu_buffer_rsrc_t rx = vk_rsrc(d.a.x + (int64_t)c.b * d.a.x_bstride, d.x_bytes);
        const int t = c.tile * TN - d.a.pad + u_col[i];
        const int voff = 4 * (t < 0 ? 0 : (t >= L ? L - 1 : t));
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
            const int ch = c.chunk * CK + u_cg[i] * 8 + c8;
            const int soff = (int)((ch < d.a.c_in ? ch : 0) * d.a.x_cstride * 4);
            st[SET][i][c8] = vk_bld(rx, voff, soff);
        }
        if (d.a.in_mask != nullptr) mk[SET][i] = d.a.in_mask[(int64_t)c.b * L + (voff >> 2)];     // wave-uniform branch
    }
    // Straight-line code (no branch), so that it can be scheduled between the matrix
    // instructions of a tap.  A unit that does not exist cuts whatever its registers hold and
    // stores the result to the dump slot.
    template <int SET>
    __device__ __forceinline__ void convert(const ConvDev& d, const SplitCursor& c, int i, char* lds8)
    {
        if (kDiag && (d.dbg & 64)) return;                      // timing experiment: no cut, no LDS stores
        const int t = c.tile * TN - d.a.pad + u_col[i];
        const bool in = (t >= 0) && (t < d.a.length);
        u32x4 oh, om, ol;
#pragma unroll
        for (int c8 = 0; c8 < 8; c8 += 2) {
            const int ch = c.chunk * CK + u_cg[i] * 8 + c8;
            float v0 = (in && ch < d.a.c_in) ? st[SET][i][c8] : 0.0f;
            float v1 = (in && ch + 1 < d.a.c_in) ? st[SET][i][c8 + 1] : 0.0f;
            if (d.a.in_mask != nullptr) { v0 *= mk[SET][i]; v1 *= mk[SET][i]; }
            if (ACT) { v0 = vk_lrelu(v0, d.a.in_slope); v1 = vk_lrelu(v1, d.a.in_slope); }
            unsigned h, m, l;
            vk_split3_pair(v0, v1, h, m, l);
            oh[c8 >> 1] = h; om[c8 >> 1] = m; ol[c8 >> 1] = l;
        }
        char* dst = lds8 + u_lds[i] + (u_on[i] ? SET * BUF : 0);
        *reinterpret_cast<u32x4*>(dst) = oh;
        *reinterpret_cast<u32x4*>(dst + PLANE) = om;
        *reinterpret_cast<u32x4*>(dst + 2 * PLANE) = ol;
    }
};

// Dynamic LDS of a launch: two staging buffers and the dump slots of units that do not exist.
// (and one word for the arrival count of a split channel loop)
constexpr size_t vk_split_lds_bytes(int tn, int ck) { return (size_t)2 * (tn + 64) * (3 * ck * 2 + 16) + 64 * 16 + 4 * ck + 16; }

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
    using Stager = SplitStager<NW, NJ, CK, PITCH, PLANE, BUF, TN, ACT>;
    constexpr int NI = Stager::NI;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    extern __shared__ __attribute__((aligned(16))) char lds8[];

    const vk_conv_args& p = d.a;
    const int lane = threadIdx.x & 63;
    const int w    = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm   = w % WM;
    const int wn   = w / WM;
    const int half = lane >> 5;
    const int l31  = lane & 31;

    // this workgroup's run of jobs; job = (batch item, time tile, share of the channel chunks, row
    // block), row block fastest, so that the row blocks of a time tile, which stage the same x
    // tile, follow each other.  A launch too small for the GPU splits its channel chunks into
    // ksplit equal shares (the host picks a divisor of nchunk); a share's partial sums meet the
    // others' in scratch memory (vk_splitk_exchange).
    const int cpj = d.nchunk / d.ksplit;            // chunks per job
    const int per_tile = d.ksplit * d.n_rowblk;
    const int per = d.n_tiles * per_tile;
    const int64_t total = (int64_t)per * p.batch;
    const int job0 = (int)(total * blockIdx.x / gridDim.x);
    const int njobs = (int)(total * (blockIdx.x + 1) / gridDim.x) - job0;
    if (njobs <= 0) return;
    const int items = njobs * cpj;

    auto advance = [&](SplitCursor& c) {
        if (++c.chunk < (c.split + 1) * cpj) return;
        if (++c.rowblk == d.n_rowblk) {
            c.rowblk = 0;
            if (++c.split == d.ksplit) {
                c.split = 0;
                if (++c.tile == d.n_tiles) {
                    c.tile = 0;
                    ++c.b;
                }
            }
        }
        c.chunk = c.split * cpj;
    };

    vk_stamp(d, 0);
    Stager stg;
    stg.init(d, lane, w);
    SplitCursor cur;                      // the item being multiplied
    cur.b = job0 / per;
    {
        const int in_item = job0 - cur.b * per;
        cur.tile = in_item / per_tile;
        const int in_tile = in_item - cur.tile * per_tile;
        cur.split = in_tile / d.n_rowblk;
        cur.rowblk = in_tile - cur.split * d.n_rowblk;
    }
    cur.chunk = cur.split * cpj;
    SplitCursor cvt = cur;                // the item being cut into slices (one ahead)
    SplitCursor ldc = cur;                // the item whose loads are issued next (three ahead)
    int to_issue = items;

    auto issue_item = [&](auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
        if (to_issue <= 0) return;
        --to_issue;
#pragma unroll
        for (int i = 0; i < NI; ++i) stg.template issue<SET>(d, ldc, i);
        advance(ldc);
    };

    // ---- weights: per (tap, k-step) three 16-byte fragments per lane: hi, mid, lo ----
    auto wbase = [&](const SplitCursor& c, int m) {     // where job c's weight stream starts
        const int mt = (c.rowblk * WM + wm) * A_REP + m;
        return reinterpret_cast<const bf16x8*>(p.w)
             + ((int64_t)mt * d.groups_per_mt + (int64_t)c.split * cpj * p.taps * (3 * KS)) * 64 + lane;
    };

    f32x16 acc[A_REP][B_REP];
    auto init_acc = [&](const SplitCursor& c) {
        // accumulators start at the bias of their row: register r of a lane is row
        // (r & 3) + 8 * (r >> 2) + 4 * half of its 32-row tile
        const int mt0 = (c.rowblk * WM + wm) * A_REP;
        // (of a split channel loop only the first share carries the bias)
        const float* biasb = (p.bias && c.split == 0) ? vk_bias_of(p, c.b) : nullptr;
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
    };

    // ---- prologue: items 0 and 1 loaded, item 0 cut and stored, item 2 on its way ----
    issue_item(S0{});
    issue_item(S1{});

    const bf16x8* aptr[A_REP];
    bf16x8 a[KS][3][A_REP];
#pragma unroll
    for (int m = 0; m < A_REP; ++m) aptr[m] = wbase(cur, m);
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int m = 0; m < A_REP; ++m) a[k][s][m] = aptr[m][64 * (3 * k + s)];
    init_acc(cur);

#pragma unroll
    for (int i = 0; i < NI; ++i) stg.template convert<0>(d, cvt, i, lds8);
    advance(cvt);
    issue_item(S0{});
    __syncthreads();
    vk_stamp(d, 1);

    const int tap_step = p.dilation * PITCH;
    unsigned long long barrier_ticks = 0;       // diagnostic build: time this wave spent at the item barriers

    // One item: its taps run out of LDS buffer SET; during the first NI taps the next item is
    // cut, unit by unit, into the other buffer, and then the loads of the item three ahead go
    // out into the registers the cut has left.  `more`: another item follows in this run.
    auto run_item = [&](auto set_tag, bool more) {
        constexpr int SET = decltype(set_tag)::value;
        constexpr int OTHER = SET ^ 1;
        const int mt0 = (cur.rowblk * WM + wm) * A_REP;
        // transposed conv: row tiles of the upper phases read the staged tile one column later
        const int shift = (EPI == VK_EPI_UPSAMPLE) ? vk_ups_shift(p, mt0) : 0;
        const char* base = lds8 + SET * BUF + (wn * TNW + l31 + shift) * PITCH + half * 16;
        const bool last_chunk = cur.chunk + 1 == (cur.split + 1) * cpj;
        SplitCursor nxt = cur;            // the job after this one (where its weight stream starts)
        if (last_chunk) advance(nxt);

        const char* bt = base;
        bf16x8 bf[3][B_REP];
#pragma unroll
        for (int n = 0; n < B_REP; ++n)
#pragma unroll
            for (int s = 0; s < 3; ++s) bf[s][n] = *reinterpret_cast<const bf16x8*>(bt + s * PLANE + n * 32 * PITCH);

        // B fragments rotate in place: as soon as the six products of a column tile have been
        // issued, its three fragment registers are reloaded for the next k-step, so every LDS
        // read has the other column tiles' MFMAs to land behind.  Weights, which come from L2,
        // are fetched one tap ahead (at a job's last tap: the first tap of the next job).
        auto tap_body = [&](int tap, auto unit_tag) {
            constexpr int U = decltype(unit_tag)::value;         // staging unit cut in this tap, or -1
            const bool last_tap = tap + 1 == p.taps;
            const char* bt_next = last_tap ? base : bt + tap_step;   // chunk end: harmless re-read
            const bf16x8* an[A_REP];
#pragma unroll
            for (int m = 0; m < A_REP; ++m)
                an[m] = (last_tap && last_chunk) ? wbase(nxt, m) : aptr[m] + 64 * 3 * KS;
            bf16x8 a_nxt[KS][3][A_REP];
#pragma unroll
            for (int k = 0; k < KS; ++k)
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int m = 0; m < A_REP; ++m)
                        a_nxt[k][s][m] = (kDiag && (d.dbg & 512)) ? a[k][s][m] : an[m][64 * (3 * k + s)];
            if constexpr (U >= 0) stg.template convert<OTHER>(d, cvt, U, lds8);
            // the next tap's weights leave for L2 before anything else of this tap is issued
            __builtin_amdgcn_sched_group_barrier(0x020, 3 * KS * A_REP, 0);
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
                        if (!(kDiag && (d.dbg & 256)))
                            bf[s][n] = *reinterpret_cast<const bf16x8*>(nx + s * PLANE + n * 32 * PITCH);
                    if constexpr (U >= 0) {
                        // a few vector instructions of the cut behind every matrix instruction
#pragma unroll
                        for (int q = 0; q < 6 * A_REP; ++q) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, VK_SPLIT_VALU_PER_MFMA, 0);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    } else {
                        __builtin_amdgcn_sched_group_barrier(0x008, 6 * A_REP, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                    }
                }
            }
            bt = bt_next;
#pragma unroll
            for (int m = 0; m < A_REP; ++m) aptr[m] = an[m];
#pragma unroll
            for (int k = 0; k < KS; ++k)
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int m = 0; m < A_REP; ++m) a[k][s][m] = a_nxt[k][s][m];
        };

        int tap = 0;
        // units 0 .. NI-1 ride on taps 0 .. NI-1; with fewer taps the rest is cut without cover
        vk_static_for<NI>([&](auto u) {
            if (tap < p.taps) {
                tap_body(tap, u);
                ++tap;
            } else {
                stg.template convert<OTHER>(d, cvt, decltype(u)::value, lds8);
            }
        });
        advance(cvt);
        issue_item(std::integral_constant<int, OTHER>{});
#pragma unroll 1
        for (; tap < p.taps; ++tap) tap_body(tap, std::integral_constant<int, -1>{});
        if (kDiag && d.stamps != nullptr) {
            const unsigned long long tb = __builtin_amdgcn_s_memrealtime();
            __syncthreads();
            barrier_ticks += __builtin_amdgcn_s_memrealtime() - tb;
        } else
        __syncthreads();
        if (last_chunk) {
            // (the arrival word sits behind the staging buffers and the dump slots)
            const bool finish = d.ksplit == 1 ||
                vk_splitk_exchange<A_REP, B_REP, NW>(d, acc, cur.b, cur.tile, cur.rowblk, cur.split, w, lane,
                                                    reinterpret_cast<unsigned*>(lds8 + 2 * BUF + 64 * 16 + 4 * CK));
            if (finish) conv_epilogue<A_REP, B_REP, EPI>(d, acc, cur.b, cur.tile * TN, mt0, wn, half, l31);
            if (more) init_acc(nxt);
        }
        advance(cur);
    };

    for (int it = 0; it < items; it += 2) {
        run_item(S0{}, it + 1 < items);
        if (it + 1 < items) run_item(S1{}, it + 2 < items);
    }
    vk_stamp(d, 2);
    if (kDiag && d.stamps != nullptr && lane == 0 && w < 4)
        d.stamps[(size_t)blockIdx.x * 8 + 3 + w] = barrier_ticks;       // slots 3..6: waves 0..3
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

// Workgroups that a launch of `kernel` keeps resident on the whole GPU (asked once per
// instantiation: every GPU of a node is the same part).
template <typename K>
inline int vk_resident_workgroups(K kernel, int threads, size_t lds, std::atomic<int>& cached)
{
    int n = cached.load(std::memory_order_acquire);
    if (n > 0) return n;
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds) != hipSuccess ||
        cus <= 0 || per_cu <= 0)
        return 512;                     // 256 CUs x 2: what the register budget of these kernels allows
    n = cus * per_cu;
    cached.store(n, std::memory_order_release);
    return n;
}

// One launch-table entry.  The grid is the resident set, or one workgroup per job when there
// are fewer jobs than that.
#define VK_SPLIT_CASE(EPI, ACT)                                                                   \
    if (epi == (EPI) && act == (ACT)) {                                                          \
        auto k = conv_split_kernel<G.a_rep, G.b_rep, G.wm, G.wn, CKB, (EPI), (ACT) != 0>;         \
        static std::atomic<unsigned long long> lds_raised{0};   /* bit d: limit raised on device d */\
        if (hipError_t e = vk_allow_lds(k, lds, lds_raised); e != hipSuccess) return (int)e;     \
        static std::atomic<int> resident{0};                                                     \
        const int64_t jobs = (int64_t)d.n_tiles * d.n_rowblk * d.a.batch * d.ksplit;             \
        const int slots = vk_resident_workgroups(k, 64 * G.wm * G.wn, lds, resident);            \
        const dim3 pgrid((unsigned)(jobs < slots ? jobs : slots), 1, 1);                         \
        hipLaunchKernelGGL(k, pgrid, dim3(64 * G.wm * G.wn), lds, s, d);                         \
        return (int)hipGetLastError();                                                           \
    }

int vk_launch_split_tile0(int epi, int act, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s);
int vk_launch_split_tile1(int epi, int act, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s);
int vk_launch_split_tile2(int epi, int act, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s);
int vk_launch_split_tile3(int epi, int act, const ConvDev& d, dim3 grid, size_t lds, hipStream_t s);
