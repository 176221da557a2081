// The one-launch WN layer of wn_layer.hpp with bf16 matrix-core inputs
// (/root/reference/module/model_component/wn.py:68-85):
//
//   a    = in_conv(bf16(x)) + cond           k taps, H -> 2H, per-utterance conditioning bias
//   acts = tanh(a[:H]) * sigmoid(a[H:])      wn.py:15-22 (fp32, on the accumulators)
//   rs   = res_skip(bf16(acts))              1x1, H -> 2H (H on the last layer)
//   x'   = (x + rs[:H]) * mask               skip (+)= rs[H:]          wn.py:80-83
//   last layer:  out = (skip + rs) * mask                               wn.py:85-86
//
// The roundings are those of two conv_bf16_kernel launches (GATE epilogue, then the 1x1 conv with
// the WN_MID / WN_LAST epilogue): x and the gated activations are rounded to bf16 where they enter
// a matrix instruction, everything else is fp32.  Six waves: wave q owns the (tanh, sigmoid) row
// tiles of channels 32q.. through the k-tap conv (x staged channel-minor in bf16 like
// conv_bf16.hpp, 32 channels per chunk), gates them in registers, writes its slab of the acts
// image [column][H] to LDS (over the staged x chunks), and after a barrier runs the 1x1 conv from
// that image, producing the residual tile q and the skip tile q.  The 1x1 conv needs no halo, so
// nothing is recomputed and acts never reaches HBM.
#pragma once
#include "conv_bf16.hpp"
#include "resblock_pair_bf16.hpp"
#include "wn_layer.hpp"

template <int NWAVES, int B_REP, bool LAST>
__global__ void __launch_bounds__(64 * NWAVES, 2)
wn_layer_bf16_kernel(const WnDev q)
{
    constexpr int H   = 32 * NWAVES;
    constexpr int TN  = 32 * B_REP;
    constexpr int CK  = 32;
    constexpr int KS  = CK / 16;
    constexpr int NCH = H / CK;
    constexpr int NJ  = (TN + 64 + 63) / 64;
    constexpr int XP  = CK * 2 + 16;            // bytes per staged x column (one chunk)
    constexpr int XBUF = NJ * 64 * XP;          // every lane of a staged 64-column piece writes its slot
    constexpr int AP  = H * 2 + 16;             // bytes per acts column (all channels): odd multiple of 16
    constexpr int RS_REP = LAST ? 1 : 2;        // row tiles per wave in the 1x1 conv

    extern __shared__ __attribute__((aligned(16))) char lds8[];

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
    Bf16Stager<NWAVES, NJ, CK, XP, false> stg;
    stg.init(d, xb, t0, lane, w);
    stg.issue(d, 0);

    // ---------------- in-conv: row tiles 2w (tanh half) and 2w + 1 (sigmoid half) ----------------
    const bf16x8* aptr[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
        aptr[m] = reinterpret_cast<const bf16x8*>(p.w) + ((int64_t)(2 * w + m) * d.groups_per_mt) * 64 + lane;
    bf16x8 a[VK_PF][KS][2];
    bf16_wload<KS, 2>(a, aptr);

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

    const int bcol = l31 * XP + half * 16;
    for (int chunk = 0; chunk < NCH; ++chunk) {
        char* buf = lds8 + (chunk & 1) * XBUF;
        stg.store(d, chunk, buf);
        __syncthreads();
        if (chunk + 1 < NCH) stg.issue(d, chunk + 1);
        bf16_taps<2, B_REP, KS, XP>(acc, a, aptr, buf + bcol, p.taps, p.dilation * XP);
    }

    // ---- 1x1 conv: its first weights and its bias are requested before acts is handed over ----
    const bf16x8* rptr[RS_REP];
#pragma unroll
    for (int m = 0; m < RS_REP; ++m)
        rptr[m] = reinterpret_cast<const bf16x8*>(q.w_rs) + ((int64_t)(m * NWAVES + w) * q.groups_rs) * 64 + lane;
    bf16x8 ra[VK_PF][KS][RS_REP];
    bf16_wload<KS, RS_REP>(ra, rptr);

    __syncthreads();                    // every wave is done reading the staged x chunks
    {
        // register r of a lane is channel 32w + (r & 3) + 8 * (r >> 2) + 4 * half of its column:
        // four consecutive channels per group of four registers -> one 8-byte LDS store
        char* acol = lds8 + l31 * AP + (32 * w + 4 * half) * 2;
#pragma unroll
        for (int n = 0; n < B_REP; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const f32x2 pr = {vk_gate(acc[0][n][4 * g + e], acc[1][n][4 * g + e]),
                                      vk_gate(acc[0][n][4 * g + e + 1], acc[1][n][4 * g + e + 1])};
                    const bf16x2 c2 = __builtin_convertvector(pr, bf16x2);
                    o[e] = c2[0]; o[e + 1] = c2[1];
                }
                *reinterpret_cast<bf16x4*>(acol + n * 32 * AP + g * 16) = o;
            }
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

    const int acol2 = l31 * AP + half * 16;
    for (int chunk = 0; chunk < NCH; ++chunk)
        bf16_taps<RS_REP, B_REP, KS, AP>(rs, ra, rptr, lds8 + acol2 + chunk * (CK * 2), 1, 0);

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

// LDS bytes of a workgroup: the larger of the two x staging buffers and the acts image
static inline size_t wn_bf16_lds(int b_rep)
{
    const int TN = 32 * b_rep, NJ = (TN + 64 + 63) / 64;
    const size_t x = (size_t)2 * NJ * 64 * (32 * 2 + 16);
    const size_t a = (size_t)TN * (32 * VK_WN_WAVES * 2 + 16);
    return x > a ? x : a;
}

// b_rep 1, 2 or 4 (32 / 64 / 128 time steps per workgroup).  Returns hipSuccess / a HIP error, or -1.
int vk_launch_wn_layer_bf16(int b_rep, int last, const WnDev& q, dim3 grid, size_t lds, hipStream_t s);
