// One launch for a whole layer of the WN gated stack
// (/root/reference/module/model_component/wn.py:68-85):
//
//   a    = in_conv(x) + cond                 k taps, H -> 2H, per-utterance conditioning bias
//   acts = tanh(a[:H]) * sigmoid(a[H:])      wn.py:15-22
//   rs   = res_skip(acts)                    1x1, H -> 2H (H on the last layer)
//   x'   = (x + rs[:H]) * mask               skip (+)= rs[H:]          wn.py:80-83
//   last layer:  out = (skip + rs) * mask                               wn.py:85-86
//
// A workgroup owns all 2H gate rows of a window of TN time steps: wave q computes the
// (a, b) row tiles of channels 32q..32q+31, gates them in registers and writes its 32 x TN
// slab of acts into LDS (over the space of the staged x chunks); after a barrier the same
// waves run the 1x1 convolution with the acts tile as the B operand, wave q producing the
// residual tile q and the skip tile q.  acts never reaches HBM, and because the 1x1 conv
// needs no halo nothing is recomputed.  x' goes to a second buffer (neighbouring workgroups
// still read halo columns of x).
#pragma once
#include "conv_mfma.hpp"

struct WnDev {
    ConvDev c;              // in-conv: x view, packed w (pair order), bias = conditioning (+ in-conv bias), taps,
                            // dilation, pad, staging geometry; out = x', out2 = skip, res = x, mask; a.first
    const float* w_rs;      // 1x1 conv, packed with the same chunk size
    const float* bias_rs;
    int groups_rs;          // 64-lane x float4 groups per 32-row tile of the 1x1 conv
    int last;               // last layer: H output rows, out = (skip + rs) * mask
};

// k-loop over one resident chunk of CK channels: taps outermost, channel pairs inside; each
// weight group is refetched for the next tap right after its last use.
template <int A_REP, int B_REP, int GPC, int STRIDE>
__device__ __forceinline__ void wn_taps(f32x16 (&acc)[A_REP][B_REP], float4 (&a)[GPC][A_REP], const float4* (&aptr)[A_REP],
                                        const float* base, int taps, int dilation)
{
    const float* bt = base;
    float bf[B_REP];
#pragma unroll
    for (int n = 0; n < B_REP; ++n) bf[n] = bt[32 * n];
#pragma unroll 1
    for (int tap = 0; tap < taps; ++tap) {
        const float* bt_next = (tap + 1 < taps) ? bt + dilation : base;     // chunk end: harmless re-read
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
                    const float av = (e == 0) ? a_cur[m].x : (e == 1) ? a_cur[m].y : (e == 2) ? a_cur[m].z : a_cur[m].w;
#pragma unroll
                    for (int n = 0; n < B_REP; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf[n], acc[m][n], 0, 0, 0);
                }
#pragma unroll
                for (int n = 0; n < B_REP; ++n) bf[n] = bn[n];
                __builtin_amdgcn_sched_group_barrier(0x100, (B_REP + 1) / 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, A_REP * B_REP, 0);
            }
#pragma unroll
            for (int m = 0; m < A_REP; ++m) a[g][m] = aptr[m][64 * (GPC + g)];
            __builtin_amdgcn_sched_group_barrier(0x020, A_REP, 0);
        }
        bt = bt_next;
#pragma unroll
        for (int m = 0; m < A_REP; ++m) aptr[m] += 64 * GPC;
    }
}

// NWAVES = H / 32 waves; B_REP column tiles of 32 time steps per workgroup.
template <int NWAVES, int B_REP, bool LAST, bool VEC>
__global__ void __launch_bounds__(64 * NWAVES, 2)
wn_layer_kernel(const WnDev q)
{
    constexpr int H   = 32 * NWAVES;
    constexpr int TN  = 32 * B_REP;
    constexpr int CK  = 4 * NWAVES;             // staged channels per chunk: 4 rows per wave
    constexpr int RPW = 4;
    constexpr int GPC = CK / 8;
    constexpr int NCH = H / CK;                 // 8 chunks
    constexpr int NJ  = (TN + 64 + 63) / 64;
    constexpr int NJ4 = (TN + 70 + 255) / 256;
    constexpr int STRIDE = TN + 64;             // staged x rows
    constexpr int ASTRIDE = TN + 4;             // acts rows
    constexpr int RS_REP = LAST ? 1 : 2;        // row tiles per wave in the 1x1 conv
    static_assert(CK % 8 == 0, "chunk must be whole k-groups");

    extern __shared__ __attribute__((aligned(16))) float lds[];

    const ConvDev& d = q.c;
    const vk_conv_args& p = d.a;
    const int lane = threadIdx.x & 63;
    const int w    = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;
    const int l31  = lane & 31;
    const int b    = blockIdx.z;
    const int t0   = blockIdx.x * TN;
    const int L    = p.length;

    const float* xb = p.x + (int64_t)b * p.x_bstride;
    using Stager = typename std::conditional<VEC, VectorStager<RPW, NJ4, NWAVES, CK, false>,
                                                  ScalarStager<RPW, NJ, NWAVES, CK, false>>::type;
    Stager stg;
    stg.init(d, xb, t0, lane, b);
    stg.issue(d, 0, w);

    // ---------------- in-conv: row tiles 2w (tanh half) and 2w + 1 (sigmoid half) ----------------
    const float4* aptr[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
        aptr[m] = reinterpret_cast<const float4*>(p.w) + ((int64_t)(2 * w + m) * d.groups_per_mt) * 64 + lane;
    float4 a[GPC][2];
#pragma unroll
    for (int g = 0; g < GPC; ++g)
#pragma unroll
        for (int m = 0; m < 2; ++m) a[g][m] = aptr[m][64 * g];

    f32x16 acc[2][B_REP];
    {
        const float* biasb = vk_bias_of(p, b);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            float bv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bv[r] = biasb[m * H + 32 * w + (r & 3) + 8 * (r >> 2) + 4 * half];
#pragma unroll
            for (int n = 0; n < B_REP; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = bv[r];
        }
    }

    const int bcol = half * STRIDE + l31 + d.col0;
    for (int chunk = 0; chunk < NCH; ++chunk) {
        float* buf = lds + (chunk & 1) * (CK * STRIDE);
        stg.store(d, chunk, buf, w, lane);
        __syncthreads();
        if (chunk + 1 < NCH) stg.issue(d, chunk + 1, w);
        wn_taps<2, B_REP, GPC, STRIDE>(acc, a, aptr, buf + bcol, p.taps, p.dilation);
    }

    // ---- 1x1 conv: its first weights and its bias are requested before acts is handed over ----
    const float4* rptr[RS_REP];
#pragma unroll
    for (int m = 0; m < RS_REP; ++m)
        rptr[m] = reinterpret_cast<const float4*>(q.w_rs) + ((int64_t)(m * NWAVES + w) * q.groups_rs) * 64 + lane;
    float4 ra[GPC][RS_REP];
#pragma unroll
    for (int g = 0; g < GPC; ++g)
#pragma unroll
        for (int m = 0; m < RS_REP; ++m) ra[g][m] = rptr[m][64 * g];

    __syncthreads();                    // every wave is done reading the staged x chunks
    {
        float* arow = lds + (32 * w + 4 * half) * ASTRIDE + l31;
#pragma unroll
        for (int n = 0; n < B_REP; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                arow[((r & 3) + 8 * (r >> 2)) * ASTRIDE + 32 * n] = vk_gate(acc[0][n][r], acc[1][n][r]);
    }
    f32x16 rs[RS_REP][B_REP];
#pragma unroll
    for (int m = 0; m < RS_REP; ++m) {
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            bv[r] = q.bias_rs ? q.bias_rs[m * H + 32 * w + (r & 3) + 8 * (r >> 2) + 4 * half] : 0.0f;
#pragma unroll
        for (int n = 0; n < B_REP; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) rs[m][n][r] = bv[r];
    }
    __syncthreads();

    const int acol = half * ASTRIDE + l31;
    for (int chunk = 0; chunk < NCH; ++chunk)
        wn_taps<RS_REP, B_REP, GPC, ASTRIDE>(rs, ra, rptr, lds + chunk * (CK * ASTRIDE) + acol, 1, 0);

    // ---------------- epilogue: residual / skip / mask ----------------
    const int64_t ob  = (int64_t)b * p.out_bstride;
    const int64_t rb  = (int64_t)b * p.res_bstride;
    const int64_t o2b = (int64_t)b * p.out2_bstride;
    const __amdgpu_buffer_rsrc_t r_out  = vk_rsrc(p.out + ob, d.out_bytes);
    const __amdgpu_buffer_rsrc_t r_res  = vk_rsrc(p.res + rb, d.res_bytes);
    const __amdgpu_buffer_rsrc_t r_skip = vk_rsrc(p.out2 + o2b, d.out2_bytes);
    const float* maskb = p.mask + (int64_t)b * L;
    const int tw = t0 + l31;
    const int R0 = 32 * w;
#pragma unroll
    for (int n = 0; n < B_REP; ++n) {
        const int t = tw + 32 * n;
        const bool ok = t < L;
        const float mk = ok ? maskb[t] : 0.0f;
        const int vo = 4 * (4 * half * (int)p.out_cstride + t);
        const int vr = 4 * (4 * half * (int)p.res_cstride + t);
        const int vs = 4 * (4 * half * (int)p.out2_cstride + t);
        // loads are unconditional (past the view's end the buffer returns zero); stores are not
        float xv[16], sv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = R0 + (r & 3) + 8 * (r >> 2);
            xv[r] = LAST ? 0.0f : vk_bld(r_res, vr, rr * (int)p.res_cstride * 4);
            sv[r] = p.first ? 0.0f : vk_bld(r_skip, vs, rr * (int)p.out2_cstride * 4);
        }
        if (ok) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = R0 + (r & 3) + 8 * (r >> 2);
                if constexpr (LAST) {
                    vk_bst((sv[r] + rs[0][n][r]) * mk, r_out, vo, rr * (int)p.out_cstride * 4);
                } else {
                    vk_bst((xv[r] + rs[0][n][r]) * mk, r_out, vo, rr * (int)p.out_cstride * 4);
                    vk_bst(sv[r] + rs[RS_REP - 1][n][r], r_skip, vs, rr * (int)p.out2_cstride * 4);
                }
            }
        }
    }
}

constexpr int VK_WN_WAVES = 6;          // hidden = 192
constexpr int VK_WN_CK = 4 * VK_WN_WAVES;

// b_rep 1 or 2 (32 / 64 time steps per workgroup).  Returns hipSuccess / a HIP error, or -1.
int vk_launch_wn_layer(int b_rep, int last, int vec, const WnDev& q, dim3 grid, size_t lds, hipStream_t s);
