// Implicit-GEMM Conv1d / polyphase ConvTranspose1d on the gfx950 fp32 matrix cores.
//
//   out[row, t] = sum_{c, j} W[row, c, j] * act(x[c, t + j*dilation - pad])   (+ fused epilogue)
//
// GEMM view: M = output rows (channels, or phase*channels for the transposed conv),
// N = time, K = (input channel, tap).  One workgroup owns 32*A_REP*WM rows by
// 32*B_REP*WN time steps.  The activation tile (CK input channels by time tile + halo) is
// staged through LDS once per K-chunk, double buffered, with the leaky-ReLU applied while
// staging; every tap of the convolution then reads it at a shifted column, so nothing is
// re-fetched from HBM per tap and no im2col matrix exists.  Weights arrive pre-packed in
// fragment order (see packing.py) and stream straight from L2 into registers, one 16-byte
// load per lane per four k-steps; a group's registers are refilled for the next tap right
// after their last use.
//
// Matrix instruction: v_mfma_f32_32x32x2_f32 (exact fp32 fma chain).  Operand maps:
//   A: lane l holds A[row = l & 31][k = l >> 5]        (k = channel within the pair)
//   B: lane l holds B[k = l >> 5][col = l & 31]        (col = time)
//   D: register r of lane l is D[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31]
// Time is the lane dimension everywhere, so LDS reads, global loads and stores are
// contiguous in time.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <type_traits>
#include "../../include/vits_amd.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// A kernel may use more than 64 KiB of dynamic LDS only after the limit has been raised for it
// on the device it is launched on.  `raised` belongs to one kernel instantiation: bit d says
// that device d has been done (devices 64 apart share a bit, which costs one redundant call).
template <typename K>
inline hipError_t vk_allow_lds(K kernel, size_t lds, std::atomic<unsigned long long>& raised)
{
    if (lds <= 65536) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (raised.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) raised.fetch_or(bit, std::memory_order_release);
    return e;
}

struct ConvDev {
    vk_conv_args a;
    int nchunk;         // K-chunks: ceil(c_in / CK)
    int xw;             // staged columns per row: time tile + halo
    int lds_stride;     // floats between staged rows
    int groups_per_mt;  // 64-lane x float4 groups per 32-row tile: nchunk * taps * CK / 8
    int col0;           // staged column of time t0 - pad (vector staging starts at a multiple of 4 columns)
    unsigned x_bytes, out_bytes, res_bytes, out2_bytes;   // byte spans of one batch item's views (buffer bounds)
    unsigned long long* stamps;   // diagnostic build only: 8 x u64 per workgroup (times, placement) or null
    int n_tiles;        // time tiles per batch item
    int n_rowblk;       // row blocks; > 1 with xcd_map: blockIdx.x enumerates (XCD, row block, tile group)
    int xcd_map;        // 1: workgroups that share a staged x tile (all row blocks of a time tile) are
                        // placed on the same XCD in consecutive dispatch slots, so the tile is fetched
                        // from HBM once per XCD instead of once per row block
    int dbg;            // diagnostic build only (VK_DEBUG_SKIP): 1 = no epilogue, 2 = no MFMA loop,
                        // 4 = stage only the first chunk, 8 = barrier only on the first chunk
    // Split of the channel loop (fp32 and sliced-fp32 kernels; 1 elsewhere).  With ksplit > 1, blockIdx.y
    // enumerates (split, row block); workgroup `split` runs chunks [split * nchunk / ksplit,
    // (split + 1) * nchunk / ksplit) and leaves its accumulators in `ws`; the last workgroup of an
    // output tile to arrive adds the partials in split order and runs the epilogue.
    int ksplit;
    float* ws;          // VK_WS_COUNTER_BYTES of arrival counters (one per output tile), then the partials
};

// tanh(a) * sigmoid(b) on the hardware exp2 / reciprocal units (each about 1 ulp): ten vector
// instructions instead of the several dozen of the library tanhf / expf / division, which matter
// because every one of them waits behind the other waves' fp32 matrix instructions.
//   sigmoid(b) = 1 / (1 + 2^(-b log2 e)),   tanh(a) = 1 - 2 / (1 + 2^(2 a log2 e))
// Saturation is exact (2^x -> inf or 0 gives 0, 1 or -1); the absolute error is about 2e-7.
__device__ __forceinline__ float vk_gate(float a, float b)
{
    constexpr float LOG2E = 1.4426950408889634f;
    const float eb = __builtin_amdgcn_exp2f(-LOG2E * b);
    const float ea = __builtin_amdgcn_exp2f(2.0f * LOG2E * a);
    const float sg = __builtin_amdgcn_rcpf(1.0f + eb);
    const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + ea);
    return th * sg;
}


typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Buffer addressing: a wave-uniform descriptor per tensor view, the row offset in a scalar
// register and the per-lane column offset in one VGPR, so loads and stores need no per-access
// vector address arithmetic (every vector instruction issued in a load/store phase competes
// with the other waves' matrix instructions for the SIMD).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t vk_rsrc(const void* base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float vk_bld(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void vk_bst(float v, __amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), r, voff, soff, 0);
}
// Bias vector of batch item b: row b of [B, *], or with a bias index (a per-speaker table) the
// row that the item's index names, clamped into the table.
__device__ __forceinline__ const float* vk_bias_of(const vk_conv_args& p, int b)
{
    long long row = b;
    if (p.bias_index != nullptr) {
        row = p.bias_index[b];
        row = row < 0 ? 0 : (row >= p.bias_rows ? p.bias_rows - 1 : row);
    }
    return p.bias + row * p.bias_bstride;
}
// leaky-ReLU with 0 <= slope <= 1 is max(v, slope * v); one v_max_f32 without the compiler's
// canonicalising pre-max (NaN in gives NaN out either way).
__device__ __forceinline__ float vk_lrelu(float v, float slope)
{
    float r;
    const float sv = v * slope;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(sv));
    return r;
}

// Staging of one K-chunk, 4 bytes per lane per load: any length and alignment.
template <int RPW, int NJ, int NW, int CK, bool ACT>
struct ScalarStager {
    int toff[NJ];               // clamped byte offset of column j*64 + lane within a row
    unsigned inrange;           // bit j: that column lies inside [0, L)
    bool last_col_ok;
    float st[RPW][NJ];
    float mk[NJ];               // input mask of column j*64 + lane (only read when the launch has one)
    __amdgpu_buffer_rsrc_t rx;

    __device__ __forceinline__ void init(const ConvDev& d, const float* xb, int t0, int lane, int b)
    {
        const int L = d.a.length;
        rx = vk_rsrc(xb, d.x_bytes);
        inrange = 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            int col = lane + 64 * j;
            col = col < d.xw ? col : d.xw - 1;                // lanes past the staged width re-read its last column
            const int t = t0 - d.a.pad + col;
            toff[j] = 4 * (t < 0 ? 0 : (t >= L ? L - 1 : t));
            if (t >= 0 && t < L) inrange |= 1u << j;
            if (d.a.in_mask != nullptr) mk[j] = d.a.in_mask[(int64_t)b * L + (toff[j] >> 2)];     // wave-uniform branch
        }
        last_col_ok = (lane + 64 * (NJ - 1)) < d.xw;          // only the last piece can overhang
    }
    __device__ __forceinline__ void issue(const ConvDev& d, int chunk, int w)
    {
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int c = chunk * CK + w + r * NW;                                  // wave-uniform
            const int soff = (int)((c < d.a.c_in ? c : 0) * d.a.x_cstride * 4);
#pragma unroll
            for (int j = 0; j < NJ; ++j) st[r][j] = vk_bld(rx, toff[j], soff);      // the select happens at store time
        }
    }
    __device__ __forceinline__ void store(const ConvDev& d, int chunk, float* buf, int w, int lane)
    {
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int cl = w + r * NW;
            const bool cok = (chunk * CK + cl) < d.a.c_in;
            float* dst = buf + cl * d.lds_stride + lane;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float v = (cok && ((inrange >> j) & 1u)) ? st[r][j] : 0.0f;
                if (d.a.in_mask != nullptr) v *= mk[j];
                if (ACT) v = vk_lrelu(v, d.a.in_slope);
                if (j < NJ - 1 || last_col_ok) dst[64 * j] = v;
            }
        }
    }
};

// Staging of one K-chunk, 16 bytes per lane per load: rows 16-byte aligned and L % 4 == 0.  The
// staged tile starts at a column that is a multiple of 4, so each float4 lies wholly inside or
// wholly outside [0, L); interior tiles skip the validity selects altogether.
template <int RPW, int NJ4, int NW, int CK, bool ACT>
struct VectorStager {
    int voff[NJ4];              // clamped byte offset of float4 number j*64 + lane within a row
    unsigned inrange, wanted;   // bit j: float4 inside [0, L) / inside the staged width
    bool edge;                  // wave-uniform: some float4 of this tile is out of range
    float4 st[RPW][NJ4];
    __amdgpu_buffer_rsrc_t rx;

    __device__ __forceinline__ void init(const ConvDev& d, const float* xb, int t0, int lane, int /*b*/)
    {
        const int L = d.a.length;
        const int ts = t0 - d.a.pad - d.col0;                   // multiple of 4
        rx = vk_rsrc(xb, d.x_bytes);
        inrange = 0; wanted = 0;
#pragma unroll
        for (int j = 0; j < NJ4; ++j) {
            const int c4 = lane + 64 * j;
            if (4 * c4 < d.xw) wanted |= 1u << j;
            const int t = ts + (4 * c4 < d.xw ? 4 * c4 : d.xw - 4);   // lanes past the staged width re-read its last float4
            voff[j] = 4 * (t < 0 ? 0 : (t > L - 4 ? L - 4 : t));
            if (t >= 0 && t + 4 <= L) inrange |= 1u << j;
        }
        edge = (ts < 0) || (ts + d.xw > L);
    }
    __device__ __forceinline__ void issue(const ConvDev& d, int chunk, int w)
    {
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int c = chunk * CK + w + r * NW;
            const int soff = (int)((c < d.a.c_in ? c : 0) * d.a.x_cstride * 4);
#pragma unroll
            for (int j = 0; j < NJ4; ++j)
                st[r][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff[j], soff, 0));
        }
    }
    __device__ __forceinline__ void store(const ConvDev& d, int chunk, float* buf, int w, int lane)
    {
        const bool chan_edge = (chunk + 1) * CK > d.a.c_in;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int cl = w + r * NW;
            float4* dst = reinterpret_cast<float4*>(buf + cl * d.lds_stride) + lane;
#pragma unroll
            for (int j = 0; j < NJ4; ++j) {
                float4 v = st[r][j];
                if (edge || chan_edge) {
                    const bool ok = ((chunk * CK + cl) < d.a.c_in) && ((inrange >> j) & 1u);
                    v.x = ok ? v.x : 0.0f; v.y = ok ? v.y : 0.0f; v.z = ok ? v.z : 0.0f; v.w = ok ? v.w : 0.0f;
                }
                if (ACT) {
                    v.x = vk_lrelu(v.x, d.a.in_slope); v.y = vk_lrelu(v.y, d.a.in_slope);
                    v.z = vk_lrelu(v.z, d.a.in_slope); v.w = vk_lrelu(v.w, d.a.in_slope);
                }
                if ((wanted >> j) & 1u) dst[64 * j] = v;
            }
        }
    }
};

// Timing experiments (time stamps per workgroup, kernels with parts left out) exist only in the
// diagnostic build of the library (-DVK_DIAG, libvits_amd_diag.so); in the product build the
// conditions below are compile-time false and the code is gone.
#ifdef VK_DIAG
constexpr bool kDiag = true;
#else
constexpr bool kDiag = false;
#endif

// Diagnostic time stamps (100 MHz constant clock); written only when a stamp buffer is set.
__device__ __forceinline__ void vk_stamp(const ConvDev& d, int slot)
{
    if (kDiag && d.stamps != nullptr && threadIdx.x == 0) {
        const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        d.stamps[(size_t)lin * 8 + slot] = __builtin_amdgcn_s_memrealtime();
        if (slot == 0) d.stamps[(size_t)lin * 8 + 7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11))
                                                      | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32);
    }
}

// Row layout of the polyphase transposed convolution (VK_EPI_UPSAMPLE).
//  - stride 8 ("grouped"): a 32-row tile holds 8 channels x 4 phases of one half of the phases,
//    tile mt = (channel group mt >> 1, half mt & 1), row-in-tile i = 4 * channel + phase-in-half.
//    A lane's four consecutive registers are then four consecutive output samples of one channel
//    (one 16-byte store), and a whole tile shares the staged-column shift of its half.
//  - other strides: packed row = phase * c_out + channel (a tile is 32 channels of one phase).
__device__ __forceinline__ bool vk_ups_grouped(const vk_conv_args& p) { return p.up_stride == 8; }
__device__ __forceinline__ void vk_ups_row(const vk_conv_args& p, int mt, int i, int& co, int& phase)
{
    if (vk_ups_grouped(p)) {
        co = 8 * (mt >> 1) + (i >> 2);
        phase = 4 * (mt & 1) + (i & 3);
    } else {
        const int prow = mt * 32 + i;
        phase = prow / p.c_out;
        co = prow - phase * p.c_out;
    }
}
// upper phases read the staged tile one column later
__device__ __forceinline__ int vk_ups_shift(const vk_conv_args& p, int mt)
{
    int co, phase;
    vk_ups_row(p, mt, 0, co, phase);
    return phase >= (p.up_stride >> 1) ? 1 : 0;
}

// Fused epilogue of one wave's accumulators (A_REP x B_REP tiles of 32 x 32, rows starting at
// packed tile mt0, columns at t0 + wn * 32 * B_REP); shared by every main-loop variant.
template <int A_REP, int B_REP, int EPI>
__device__ __forceinline__ void conv_epilogue(const ConvDev& d, f32x16 (&acc)[A_REP][B_REP], int b, int t0, int mt0,
                                              int wn, int half, int l31)
{
    constexpr int TNW = 32 * B_REP;
    constexpr bool PAIR = (EPI == VK_EPI_GATE || EPI == VK_EPI_SAMPLE);
    const vk_conv_args& p = d.a;
    const int L = p.length;
    // ---------------------------------- epilogue ----------------------------------------
    // Interior tiles (every row and column of the wave's tile in range) take a branch-free
    // path: wave-uniform row pointers plus one per-lane offset, the epilogue's loads issued
    // sixteen at a time ahead of the arithmetic.  Edge tiles take the predicated path below.
    const int tw = t0 + wn * TNW + l31;                 // this lane's time for n = 0
    const int64_t ob  = (int64_t)b * p.out_bstride;
    const int64_t rb  = (int64_t)b * p.res_bstride;
    const int64_t o2b = (int64_t)b * p.out2_bstride;
    const int rows_total = PAIR ? 2 * p.hidden : (EPI == VK_EPI_UPSAMPLE ? p.c_out * p.up_stride : p.c_out);
    const bool interior = (t0 + wn * TNW + TNW <= L) && ((mt0 + A_REP) * 32 <= rows_total);

    if (interior) {
        const float* maskb = p.mask ? p.mask + (int64_t)b * L : nullptr;
        if constexpr (PAIR) {
#pragma unroll
            for (int m = 0; m < A_REP; m += 2) {
                const int R0 = 16 * (mt0 + m);            // first output row of the pair (32 * q)
#pragma unroll
                for (int n = 0; n < B_REP; ++n) {
                    const int t = tw + 32 * n;
                    const float mkv = maskb ? maskb[t] : 1.0f;
                    float ev[16];
                    if constexpr (EPI == VK_EPI_SAMPLE) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float* ap = p.aux + ob + (int64_t)(R0 + (r & 3) + 8 * (r >> 2)) * p.out_cstride;
                            ev[r] = ap[4 * half * p.out_cstride + t];
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = R0 + (r & 3) + 8 * (r >> 2);
                        const int loff = 4 * half * (int)p.out_cstride + t;
                        const float va = acc[m][n][r];
                        const float vb = acc[m + 1][n][r];
                        float* op = p.out + ob + (int64_t)rr * p.out_cstride;
                        if constexpr (EPI == VK_EPI_GATE) {
                            op[loff] = vk_gate(va, vb);
                        } else {
                            const float mean = va * mkv, logs = vb * mkv;
                            op[loff] = mean;
                            (p.out3 + ob + (int64_t)rr * p.out_cstride)[loff] = logs;
                            (p.out2 + o2b + (int64_t)rr * p.out2_cstride)[4 * half * (int)p.out2_cstride + t] =
                                (mean + ev[r] * expf(logs)) * mkv;
                        }
                    }
                }
            }
        } else if constexpr (EPI == VK_EPI_UPSAMPLE) {
            const int S = p.up_stride;
            const __amdgpu_buffer_rsrc_t r_out = vk_rsrc(p.out + ob, d.out_bytes);
            if (vk_ups_grouped(p)) {
                // registers 4j..4j+3 of a lane: channel 8*cg + 2j + half, output samples 8t + 4*half_of_phases + 0..3
#pragma unroll
                for (int m = 0; m < A_REP; ++m) {
                    const int cg = (mt0 + m) >> 1, cls = (mt0 + m) & 1;
                    const int vo = 4 * (half * (int)p.out_cstride + tw * 8 + 4 * cls);
#pragma unroll
                    for (int n = 0; n < B_REP; ++n) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            u32x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float f = acc[m][n][4 * j + e];     // by value: bit_cast of a vector element lvalue reads element 0
                                v[e] = __builtin_bit_cast(unsigned int, f);
                            }
                            __builtin_amdgcn_raw_buffer_store_b128(v, r_out, vo + 128 * n * 8,
                                                                   (8 * cg + 2 * j) * (int)p.out_cstride * 4, 0);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int m = 0; m < A_REP; ++m) {
                    const int prow0 = (mt0 + m) * 32;
                    const int phase = prow0 / p.c_out;
                    const int co0 = prow0 - phase * p.c_out;
                    const int vo = 4 * (4 * half * (int)p.out_cstride + tw * S + phase);
#pragma unroll
                    for (int n = 0; n < B_REP; ++n) {
                        const int von = vo + 128 * n * S;
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            vk_bst(acc[m][n][r], r_out, von, (co0 + (r & 3) + 8 * (r >> 2)) * (int)p.out_cstride * 4);
                    }
                }
            }
        } else {
            // buffer descriptors of this batch item's views; row offsets are scalar, the lane's
            // column offset is one VGPR per view, the column tile n is an immediate
            const __amdgpu_buffer_rsrc_t r_out  = vk_rsrc(p.out + ob, d.out_bytes);
            const __amdgpu_buffer_rsrc_t r_res  = vk_rsrc(p.res ? p.res + rb : p.out + ob, p.res ? d.res_bytes : 0u);
            const __amdgpu_buffer_rsrc_t r_out2 = vk_rsrc(p.out2 ? p.out2 + o2b : p.out + ob, p.out2 ? d.out2_bytes : 0u);
            const int vo  = 4 * (4 * half * (int)p.out_cstride + tw);
            const int vr  = 4 * (4 * half * (int)p.res_cstride + tw);
            const int vo2 = 4 * (4 * half * (int)p.out2_cstride + tw);
#pragma unroll
            for (int m = 0; m < A_REP; ++m) {
                const int R0 = (mt0 + m) * 32;
                const bool skip_half = (EPI == VK_EPI_WN_MID) && (R0 >= p.hidden);     // wave-uniform
#pragma unroll
                for (int n = 0; n < B_REP; ++n) {
                    const float mkv = maskb ? maskb[tw + 32 * n] : 1.0f;
                    constexpr bool NEED_RES = (EPI == VK_EPI_RESID || EPI == VK_EPI_RESID_ACC || EPI == VK_EPI_RESID_AVG ||
                                               EPI == VK_EPI_WN_MID || EPI == VK_EPI_COUPLE_REV || EPI == VK_EPI_COUPLE_FWD);
                    constexpr bool NEED_ACC = (EPI == VK_EPI_RESID_ACC || EPI == VK_EPI_RESID_AVG ||
                                               EPI == VK_EPI_WN_MID || EPI == VK_EPI_WN_LAST);
                    float rv[16], sv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = R0 + (r & 3) + 8 * (r >> 2);
                        rv[r] = 0.0f; sv[r] = 0.0f;
                        if constexpr (NEED_RES) {
                            if (!skip_half) rv[r] = vk_bld(r_res, vr + 128 * n, rr * (int)p.res_cstride * 4);
                        }
                        if constexpr (NEED_ACC) {
                            if constexpr (EPI == VK_EPI_WN_MID) {
                                if (skip_half && !p.first)
                                    sv[r] = vk_bld(r_out2, vo2 + 128 * n, (rr - p.hidden) * (int)p.out2_cstride * 4);
                            } else if constexpr (EPI == VK_EPI_WN_LAST) {
                                if (!p.first) sv[r] = vk_bld(r_out2, vo2 + 128 * n, rr * (int)p.out2_cstride * 4);
                            } else {
                                sv[r] = vk_bld(r_out2, vo2 + 128 * n, rr * (int)p.out2_cstride * 4);
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = R0 + (r & 3) + 8 * (r >> 2);
                        const float v = acc[m][n][r];
                        const int so = rr * (int)p.out_cstride * 4;
                        const int voo = vo + 128 * n;
                        if constexpr (EPI == VK_EPI_STORE) {
                            if (maskb) vk_bst(v * mkv, r_out, voo, so);      // wave-uniform: the decoder has no mask
                            else       vk_bst(v, r_out, voo, so);
                        } else if constexpr (EPI == VK_EPI_RESID) {
                            vk_bst(v + rv[r], r_out, voo, so);
                        } else if constexpr (EPI == VK_EPI_RESID_ACC) {
                            vk_bst(sv[r] + (v + rv[r]), r_out, voo, so);
                        } else if constexpr (EPI == VK_EPI_RESID_AVG) {
                            vk_bst((sv[r] + (v + rv[r])) / p.navg, r_out, voo, so);
                        } else if constexpr (EPI == VK_EPI_WN_MID) {
                            if (!skip_half) vk_bst((rv[r] + v) * mkv, r_out, voo, so);
                            else vk_bst(sv[r] + v, r_out2, vo2 + 128 * n, (rr - p.hidden) * (int)p.out2_cstride * 4);
                        } else if constexpr (EPI == VK_EPI_WN_LAST) {
                            vk_bst((sv[r] + v) * mkv, r_out, voo, so);
                        } else if constexpr (EPI == VK_EPI_COUPLE_REV) {
                            vk_bst((rv[r] - v * mkv) * mkv, r_out, voo, so);
                        } else if constexpr (EPI == VK_EPI_COUPLE_FWD) {
                            vk_bst(v * mkv + rv[r] * mkv, r_out, voo, so);
                        }
                    }
                }
            }
        }
        vk_stamp(d, 3);
        return;
    }

    // ---- edge tiles: per-element predication ----
    float mk[B_REP];
#pragma unroll
    for (int n = 0; n < B_REP; ++n) {
        const int t = tw + 32 * n;
        mk[n] = (p.mask != nullptr && t < L) ? p.mask[(int64_t)b * L + t] : 1.0f;
    }

    if constexpr (PAIR) {
        // packed row tiles alternate (a, b): a = rows [32q, 32q+32), b = rows [hidden + 32q, ...)
#pragma unroll
        for (int m = 0; m < A_REP; m += 2) {
            const int q = (mt0 + m) >> 1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * q + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row >= p.hidden) continue;
#pragma unroll
                for (int n = 0; n < B_REP; ++n) {
                    const int t = tw + 32 * n;
                    if (t >= L) continue;
                    const float va = acc[m][n][r];
                    const float vb = acc[m + 1][n][r];
                    const int64_t o = ob + (int64_t)row * p.out_cstride + t;
                    if constexpr (EPI == VK_EPI_GATE) {
                        p.out[o] = vk_gate(va, vb);
                    } else {   // VK_EPI_SAMPLE: out = mean, out3 = log-std, out2 = sampled latent
                        const float mean = va * mk[n];
                        const float logs = vb * mk[n];
                        p.out[o]  = mean;
                        p.out3[o] = logs;
                        p.out2[o2b + (int64_t)row * p.out2_cstride + t] = (mean + p.aux[o] * expf(logs)) * mk[n];
                    }
                }
            }
        }
    } else if constexpr (EPI == VK_EPI_UPSAMPLE) {
        const int S = p.up_stride;
#pragma unroll
        for (int m = 0; m < A_REP; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int co, phase;
                vk_ups_row(p, mt0 + m, (r & 3) + 8 * (r >> 2) + 4 * half, co, phase);
                if (co >= p.c_out || phase >= S) continue;
#pragma unroll
                for (int n = 0; n < B_REP; ++n) {
                    const int t = tw + 32 * n;
                    if (t >= L) continue;
                    p.out[(int64_t)b * p.out_bstride + (int64_t)co * p.out_cstride + (int64_t)t * S + phase] = acc[m][n][r];
                }
            }
        }
    } else {
#pragma unroll
        for (int m = 0; m < A_REP; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (mt0 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row >= p.c_out) continue;
#pragma unroll
                for (int n = 0; n < B_REP; ++n) {
                    const int t = tw + 32 * n;
                    if (t >= L) continue;
                    const float v = acc[m][n][r];
                    if constexpr (EPI == VK_EPI_STORE) {
                        p.out[ob + (int64_t)row * p.out_cstride + t] = v * mk[n];
                    } else if constexpr (EPI == VK_EPI_RESID) {
                        p.out[ob + (int64_t)row * p.out_cstride + t] = v + p.res[rb + (int64_t)row * p.res_cstride + t];
                    } else if constexpr (EPI == VK_EPI_RESID_ACC || EPI == VK_EPI_RESID_AVG) {
                        const float y = v + p.res[rb + (int64_t)row * p.res_cstride + t];
                        const int64_t o2 = o2b + (int64_t)row * p.out2_cstride + t;
                        float s = p.out2[o2] + y;
                        if constexpr (EPI == VK_EPI_RESID_AVG) s = s / p.navg;
                        p.out[ob + (int64_t)row * p.out_cstride + t] = s;
                    } else if constexpr (EPI == VK_EPI_WN_MID) {
                        if (row < p.hidden) {
                            const int64_t o = ob + (int64_t)row * p.out_cstride + t;
                            p.out[o] = (p.res[rb + (int64_t)row * p.res_cstride + t] + v) * mk[n];
                        } else {
                            const int64_t o2 = o2b + (int64_t)(row - p.hidden) * p.out2_cstride + t;
                            p.out2[o2] = p.first ? v : p.out2[o2] + v;
                        }
                    } else if constexpr (EPI == VK_EPI_WN_LAST) {
                        const int64_t o2 = o2b + (int64_t)row * p.out2_cstride + t;
                        const float s = p.first ? v : p.out2[o2] + v;
                        p.out[ob + (int64_t)row * p.out_cstride + t] = s * mk[n];
                    } else if constexpr (EPI == VK_EPI_COUPLE_REV) {
                        const int64_t o = ob + (int64_t)row * p.out_cstride + t;
                        p.out[o] = (p.res[rb + (int64_t)row * p.res_cstride + t] - v * mk[n]) * mk[n];
                    } else if constexpr (EPI == VK_EPI_COUPLE_FWD) {
                        const int64_t o = ob + (int64_t)row * p.out_cstride + t;
                        p.out[o] = v * mk[n] + p.res[rb + (int64_t)row * p.res_cstride + t] * mk[n];
                    }
                }
            }
        }
    }
}

// Split channel loop, end of a workgroup's share: partial accumulators go to the workspace in
// register order (a wave's register is 256 contiguous bytes); the workgroup that finds the tile's
// counter at ksplit - 1 is the last, adds the partials in split order (its own included, so the
// sum does not depend on who arrives last), clears the counter for the next launch and returns
// true: it runs the epilogue.  The others return false.
// The partials travel as device-scope accesses (written through to memory, read past the
// caches) and a workgroup counts itself in only once its own stores have completed, so no
// cache-wide write-back or invalidate is needed: a device-scope fence costs a small launch
// about 40 us on the 8-XCD part, more than the split saves.
// `arrived`: one LDS word nobody else uses at this point; every wave of the workgroup calls this.
// The ordering of the partial stores before the arrival count rests on two properties of the gfx9
// family: vmcnt counts stores, and device-scope stores are written through.  A target with a
// separate store counter would compile this and race.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "vk_splitk_exchange: written for gfx942 / gfx950 (stores counted by vmcnt, write-through device-scope stores)"
#endif
template <int A_REP, int B_REP, int NW>
__device__ __forceinline__ bool vk_splitk_exchange(const ConvDev& d, f32x16 (&acc)[A_REP][B_REP], int b, int tile,
                                                   int rowblk, int split, int w, int lane, unsigned* arrived)
{
    constexpr int PER_WAVE = A_REP * B_REP * 16 * 64;
    const int tile_id = (b * d.n_tiles + tile) * d.n_rowblk + rowblk;
    unsigned* counter = reinterpret_cast<unsigned*>(d.ws) + tile_id;
    unsigned* part = reinterpret_cast<unsigned*>(d.ws) + VK_WS_COUNTER_BYTES / 4
                   + ((size_t)tile_id * d.ksplit * NW + w) * PER_WAVE + lane;
    unsigned* mine = part + (size_t)split * NW * PER_WAVE;
#pragma unroll
    for (int m = 0; m < A_REP; ++m)
#pragma unroll
        for (int n = 0; n < B_REP; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[m][n][r];       // (a bit_cast straight from the vector element stores element 0)
                __hip_atomic_store(mine + ((m * B_REP + n) * 16 + r) * 64, __builtin_bit_cast(unsigned, v),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
    // this wave's write-through stores have completed (a workgroup-scope fence does not wait for them)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                               // ... every wave's (and every wave is done with the staged tiles)
    if (threadIdx.x == 0)
        *arrived = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*arrived != (unsigned)(d.ksplit - 1)) return false;
#pragma unroll
    for (int m = 0; m < A_REP; ++m)
#pragma unroll
        for (int n = 0; n < B_REP; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
    for (int sp = 0; sp < d.ksplit; ++sp) {
        const unsigned* src = part + (size_t)sp * NW * PER_WAVE;
#pragma unroll
        for (int m = 0; m < A_REP; ++m)
#pragma unroll
            for (int n = 0; n < B_REP; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[m][n][r] += __builtin_bit_cast(float, __hip_atomic_load(src + ((m * B_REP + n) * 16 + r) * 64,
                                                                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

template <int A_REP, int B_REP, int WM, int WN, int CK, int EPI, bool ACT, bool VEC>
__global__ void __launch_bounds__(64 * WM * WN, (EPI == VK_EPI_SAMPLE) ? 2 : ((WM == 1 && WN == 4) ? 3 : 4))   // second figure: waves per SIMD the register budget must allow
conv_mfma_kernel(const ConvDev d)
{
    constexpr int NW  = WM * WN;
    constexpr int TNW = 32 * B_REP;          // time steps per wave
    constexpr int TN  = TNW * WN;            // time steps per workgroup
    constexpr int NJ  = TN / 64 + 1;         // 64-column pieces per staged row (halo <= 64)
    constexpr int NJ4 = (TN + 70 + 255) / 256;   // 64-lane x float4 pieces per staged row (vector staging)
    constexpr int RPW = CK / NW;             // staged rows per wave per chunk
    constexpr int GPC = CK / 8;              // weight groups per (chunk, tap)
    static_assert(CK % NW == 0 && CK % 8 == 0, "chunk must split over waves and k-groups");
    static_assert(TN % 64 == 0, "time tile must be a multiple of the wave width");
    constexpr bool PAIR = (EPI == VK_EPI_GATE || EPI == VK_EPI_SAMPLE);
    static_assert(!PAIR || (A_REP % 2 == 0), "paired epilogues need an even number of row tiles per wave");

    extern __shared__ __attribute__((aligned(16))) float lds[];

    const vk_conv_args& p = d.a;
    const int lane = threadIdx.x & 63;
    const int w    = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm   = w % WM;
    const int wn   = w / WM;
    const int half = lane >> 5;
    const int l31  = lane & 31;
    int b = blockIdx.z;
    int tile = blockIdx.x, rowblk = blockIdx.y, split = 0;
    if (d.xcd_map) {
        // hardware deals consecutive workgroup ids round-robin over the 8 XCDs; the time tiles of
        // all batch items form one sequence that is dealt over the XCDs eight at a time, so the XCDs'
        // shares differ by one tile at most however few tiles an item has
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        rowblk = slot % d.n_rowblk;
        const int gt = (slot / d.n_rowblk) * 8 + xcd;
        if (gt >= d.n_tiles * p.batch) return;         // padding of the last tile group (whole workgroup leaves)
        b = gt / d.n_tiles;
        tile = gt - b * d.n_tiles;
    } else if (d.ksplit > 1) {
        split = blockIdx.y / d.n_rowblk;
        rowblk = blockIdx.y - split * d.n_rowblk;
    }
    const int c_begin = split * d.nchunk / d.ksplit;       // this workgroup's share of the channel chunks
    const int c_end   = (split + 1) * d.nchunk / d.ksplit;
    const int t0   = tile * TN;
    const int mt0  = (rowblk * WM + wm) * A_REP;       // first packed 32-row tile of this wave
    const int L    = p.length;
    constexpr int STRIDE = TN + 64;          // floats between staged rows (the host checks that tile + halo fits)

    const float* xb = p.x + (int64_t)b * p.x_bstride;
    vk_stamp(d, 0);

    // ---- staging: global -> registers (issued early) -> LDS (written after the barrier) ----
    // The column layout of the staged tile is the same for every row and chunk of the tile, so
    // load offsets and in-range bits are computed once (see the stager structs above).
    using Stager = typename std::conditional<VEC, VectorStager<RPW, NJ4, NW, CK, ACT>,
                                                  ScalarStager<RPW, NJ, NW, CK, ACT>>::type;
    Stager stg;
    stg.init(d, xb, t0, lane, b);
    auto issue_loads = [&](int chunk) { stg.issue(d, chunk, w); };
    auto store_chunk = [&](int chunk, float* buf) { stg.store(d, chunk, buf, w, lane); };
    issue_loads(c_begin);    // first: the longest latency of the prologue; weights and bias follow it

    // ---- weight stream: one float4 per lane per 4 k-steps per row tile ----
    const float4* aptr[A_REP];
#pragma unroll
    for (int m = 0; m < A_REP; ++m)
        aptr[m] = reinterpret_cast<const float4*>(p.w)
                + ((int64_t)(mt0 + m) * d.groups_per_mt + (int64_t)c_begin * p.taps * GPC) * 64 + lane;

    // Weight fragments of a tap live in GPC register groups; in the tap loop each group is
    // reloaded for the next tap as soon as its four k-steps have been issued, so the other
    // groups' MFMAs cover the L2 latency and no register is copied at the loop's back edge.
    float4 a[GPC][A_REP];
#pragma unroll
    for (int g = 0; g < GPC; ++g)
#pragma unroll
        for (int m = 0; m < A_REP; ++m) a[g][m] = aptr[m][64 * g];

    f32x16 acc[A_REP][B_REP];
    {
        // accumulators start at the bias of their row: register r of a lane is row
        // (r & 3) + 8 * (r >> 2) + 4 * half of its 32-row tile
        // (of a split channel loop only the first share carries the bias)
        const float* biasb = (p.bias && split == 0) ? vk_bias_of(p, b) : nullptr;
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

    // transposed conv: row tiles of the upper phases read the staged tile one column later
    const int shift = (EPI == VK_EPI_UPSAMPLE) ? vk_ups_shift(p, mt0) : 0;
    const int bcol = half * STRIDE + wn * TNW + l31 + shift + d.col0;
    
    unsigned long long boundary_ticks = 0;      // diagnostic: time spent handing over chunks 1..n-1
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
        float* buf = lds + ((chunk - c_begin) & 1) * (CK * STRIDE);
        unsigned long long tb0 = 0;
        const int dbg = kDiag ? d.dbg : 0;
        const bool stamping = kDiag && d.stamps != nullptr;
        if (stamping) tb0 = __builtin_amdgcn_s_memrealtime();
        if (!(dbg & 4) || chunk == c_begin) store_chunk(chunk, buf);
        if (!(dbg & 8) || chunk == c_begin) __syncthreads();
        if (chunk == c_begin) vk_stamp(d, 1);
        else if (stamping) boundary_ticks += __builtin_amdgcn_s_memrealtime() - tb0;
        if (chunk + 1 < c_end && !(dbg & 4)) issue_loads(chunk + 1);
        if (dbg & 2) continue;

        // k-steps walk channel pairs within a tap, then taps; the B fragments of the next
        // k-step are read from LDS before the current k-step's MFMAs are issued.  Within a tap
        // every read is the tap's base address plus a compile-time offset.
        const float* bt = buf + bcol;
        float bf[B_REP];
#pragma unroll
        for (int n = 0; n < B_REP; ++n) bf[n] = bt[32 * n];
#pragma unroll 1
        for (int tap = 0; tap < p.taps; ++tap) {
            const float* bt_next = (tap + 1 < p.taps) ? bt + p.dilation : buf + bcol;   // chunk end: harmless re-read
#pragma unroll
            for (int g = 0; g < GPC; ++g) {
                float4 a_cur[A_REP];
#pragma unroll
                for (int m = 0; m < A_REP; ++m) a_cur[m] = a[g][m];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    constexpr int KSTEPS = 4 * GPC;
                    const int ks = 4 * g + e;
                    float bn[B_REP];
#pragma unroll
                    for (int n = 0; n < B_REP; ++n)
                        bn[n] = (ks + 1 < KSTEPS) ? bt[2 * (ks + 1) * STRIDE + 32 * n] : bt_next[32 * n];
#pragma unroll
                    for (int m = 0; m < A_REP; ++m) {
                        const float av = (e == 0) ? a_cur[m].x : (e == 1) ? a_cur[m].y
                                       : (e == 2) ? a_cur[m].z : a_cur[m].w;
#pragma unroll
                        for (int n = 0; n < B_REP; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf[n], acc[m][n], 0, 0, 0);
                    }
#pragma unroll
                    for (int n = 0; n < B_REP; ++n) bf[n] = bn[n];
                    __builtin_amdgcn_sched_group_barrier(0x100, (B_REP + 1) / 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, A_REP * B_REP, 0);
                }
                // this group's registers are dead now: refill them in place with the next tap's
                // group; the other groups' MFMAs of this tap and the next cover the L2 latency
#pragma unroll
                for (int m = 0; m < A_REP; ++m) a[g][m] = aptr[m][64 * (GPC + g)];
                __builtin_amdgcn_sched_group_barrier(0x020, A_REP, 0);
            }
            bt = bt_next;
#pragma unroll
            for (int m = 0; m < A_REP; ++m) aptr[m] += 64 * GPC;
        }
    }

    vk_stamp(d, 2);
    if (kDiag && d.stamps != nullptr && threadIdx.x == 0)
        d.stamps[(size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + 4] = boundary_ticks;
    if (kDiag && (d.dbg & 1) && acc[0][0][0] != 12345.678f) return;

    if (d.ksplit > 1 &&
        !vk_splitk_exchange<A_REP, B_REP, NW>(d, acc, b, tile, rowblk, split, w, lane, reinterpret_cast<unsigned*>(lds)))
        return;
    conv_epilogue<A_REP, B_REP, EPI>(d, acc, b, t0, mt0, wn, half, l31);
}
