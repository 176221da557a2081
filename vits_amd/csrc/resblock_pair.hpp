// One launch for a residual pair of the HiFi-GAN resblock:
//
//   h   = conv1(lrelu(x))   (taps k, dilation d)          /root/reference/module/model_component/decoder.py:49-52
//   out = conv2(lrelu(h)) + x   (taps k, dilation 1)      decoder.py:54-58  (+ the sum over resblocks, decoder.py:124-125)
//
// Both convolutions run on the fp32 matrix cores exactly as conv_mfma_kernel does; the
// intermediate h never leaves the compute unit.  A workgroup owns every channel (C = 32*WM)
// of a window of TN time steps of h, computed from an x tile with the first conv's halo; h is
// written (activated, zeroed outside [0, L) like conv2's zero padding) into the LDS space the x
// tile occupied, and the second conv reads its B fragments from there.  Of the TN columns the
// second conv produces, the first TN - 12 are valid (taps <= 13), so consecutive workgroups
// advance by TN - 12 time steps and recompute a 12-column overlap of h.
#pragma once
#include "conv_mfma.hpp"

struct PairDev {
    ConvDev c;              // first conv: x view, packed w, bias, taps, dilation, pad = halo/2 + (taps-1)/2,
                            // staging geometry; out / res (= x) / out2 (running sum) views, epilogue, navg
    const float* w2;        // second conv, packed like the first
    const float* bias2;
    int groups2;            // 64-lane x float4 groups per 32-row tile of the second conv
};

constexpr int VK_PAIR_OVERLAP = 12;

// out = acc + x (+ running sum) for the first TNO columns of the workgroup's window.
// RES_IN_ACC: the accumulators were started at bias + x, so x is not read again here.
template <int B_REP, int EPI, int TNO, bool RES_IN_ACC = false>
__device__ __forceinline__ void pair_epilogue(const ConvDev& d, f32x16 (&acc)[B_REP], int b, int t0, int wm, int wn,
                                              int half, int l31)
{
    constexpr int TNW = 32 * B_REP;
    const vk_conv_args& p = d.a;
    const int L = p.length;
    // ---- epilogue: out = acc + x (+ running sum), valid columns only ----
    const int64_t ob  = (int64_t)b * p.out_bstride;
    const int64_t rb  = (int64_t)b * p.res_bstride;
    const int64_t o2b = (int64_t)b * p.out2_bstride;
    const __amdgpu_buffer_rsrc_t r_out  = vk_rsrc(p.out + ob, d.out_bytes);
    const __amdgpu_buffer_rsrc_t r_res  = vk_rsrc(p.res + rb, d.res_bytes);
    const __amdgpu_buffer_rsrc_t r_out2 = vk_rsrc(p.out2 ? p.out2 + o2b : p.out + ob, p.out2 ? d.out2_bytes : 0u);
    constexpr bool NEED_ACC = (EPI == VK_EPI_RESID_ACC || EPI == VK_EPI_RESID_AVG);
    const int cw = wn * TNW + l31;                      // this lane's column for n = 0
    const int tw = t0 + cw;
    const int vo  = 4 * (4 * half * (int)p.out_cstride + tw);
    const int vr  = 4 * (4 * half * (int)p.res_cstride + tw);
    const int vo2 = 4 * (4 * half * (int)p.out2_cstride + tw);
    const int R0 = wm * 32;
#pragma unroll
    for (int n = 0; n < B_REP; ++n) {
        const bool ok = (cw + 32 * n < TNO) && (tw + 32 * n < L);
        // loads are unconditional: past the view's end the buffer returns zero, and a column
        // that is not stored may read a neighbouring row
        float rv[16], sv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = R0 + (r & 3) + 8 * (r >> 2);
            rv[r] = RES_IN_ACC ? 0.0f : vk_bld(r_res, vr + 128 * n, rr * (int)p.res_cstride * 4);
            sv[r] = 0.0f;
            if constexpr (NEED_ACC) sv[r] = vk_bld(r_out2, vo2 + 128 * n, rr * (int)p.out2_cstride * 4);
        }
        if (ok) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = R0 + (r & 3) + 8 * (r >> 2);
                const float v = RES_IN_ACC ? acc[n][r] : acc[n][r] + rv[r];
                const int so = rr * (int)p.out_cstride * 4;
                if constexpr (EPI == VK_EPI_RESID)          vk_bst(v, r_out, vo + 128 * n, so);
                else if constexpr (EPI == VK_EPI_RESID_ACC) vk_bst(sv[r] + v, r_out, vo + 128 * n, so);
                else                                        vk_bst((sv[r] + v) / p.navg, r_out, vo + 128 * n, so);
            }
        }
    }
}

template <int B_REP, int WM, int WN, int CK, int EPI, bool VEC>
__global__ void __launch_bounds__(64 * WM * WN, 2)
resblock_pair_kernel(const PairDev q)
{
    constexpr int NW  = WM * WN;
    constexpr int TNW = 32 * B_REP;
    constexpr int TN  = TNW * WN;
    constexpr int TNO = TN - VK_PAIR_OVERLAP;     // valid output columns per workgroup
    constexpr int NJ  = TN / 64 + 1;
    constexpr int NJ4 = (TN + 70 + 255) / 256;
    constexpr int RPW = CK / NW;
    constexpr int GPC = CK / 8;
    constexpr int C   = 32 * WM;                  // channels
    constexpr int NCH = C / CK;                   // K-chunks of either conv
    constexpr int STRIDE = TN + 64;
    static_assert(C % CK == 0 && CK % NW == 0 && CK % 8 == 0, "chunking");
    static_assert(TN % 64 == 0, "time tile must be a multiple of the wave width");
    static_assert(NCH <= 2, "every chunk of x keeps an LDS buffer of its own");

    extern __shared__ __attribute__((aligned(16))) float lds[];

    const ConvDev& d = q.c;
    const vk_conv_args& p = d.a;
    const int lane = threadIdx.x & 63;
    const int w    = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm   = w % WM;
    const int wn   = w / WM;
    const int half = lane >> 5;
    const int l31  = lane & 31;
    const int b    = blockIdx.z;
    const int t0   = blockIdx.x * TNO;            // first output time of this workgroup
    const int L    = p.length;
    const int h2   = (p.taps - 1) >> 1;           // second conv's padding; h column j is time t0 - h2 + j

    const float* xb = p.x + (int64_t)b * p.x_bstride;

    using Stager = typename std::conditional<VEC, VectorStager<RPW, NJ4, NW, CK, true>,
                                                  ScalarStager<RPW, NJ, NW, CK, true>>::type;
    Stager stg;
    stg.init(d, xb, t0, lane, b);
    stg.issue(d, 0, w);

    // ------------------------------ first convolution ------------------------------
    const float4* aptr = reinterpret_cast<const float4*>(p.w) + ((int64_t)wm * d.groups_per_mt) * 64 + lane;
    float4 a[GPC];
#pragma unroll
    for (int g = 0; g < GPC; ++g) a[g] = aptr[64 * g];

    f32x16 acc[B_REP];
    {
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = p.bias ? p.bias[wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] : 0.0f;
#pragma unroll
        for (int n = 0; n < B_REP; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = bv[r];
    }

    // The k-loop of one staged chunk: taps outermost, channel pairs inside, each weight group
    // refetched for the next tap right after its last use, B fragments of the next k-step read before the current k-step's MFMAs.
    auto run_chunk = [&](const float* base, int taps, int dilation) {
        const float* bt = base;
        float bf[B_REP];
#pragma unroll
        for (int n = 0; n < B_REP; ++n) bf[n] = bt[32 * n];
#pragma unroll 1
        for (int tap = 0; tap < taps; ++tap) {
            const float* bt_next = (tap + 1 < taps) ? bt + dilation : base;
#pragma unroll
            for (int g = 0; g < GPC; ++g) {
                const float4 a_cur = a[g];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    constexpr int KSTEPS = 4 * GPC;
                    const int ks = 4 * g + e;
                    float bn[B_REP];
#pragma unroll
                    for (int n = 0; n < B_REP; ++n)
                        bn[n] = (ks + 1 < KSTEPS) ? bt[2 * (ks + 1) * STRIDE + 32 * n] : bt_next[32 * n];
                    const float av = (e == 0) ? a_cur.x : (e == 1) ? a_cur.y : (e == 2) ? a_cur.z : a_cur.w;
#pragma unroll
                    for (int n = 0; n < B_REP; ++n)
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf[n], acc[n], 0, 0, 0);
#pragma unroll
                    for (int n = 0; n < B_REP; ++n) bf[n] = bn[n];
                    __builtin_amdgcn_sched_group_barrier(0x100, (B_REP + 1) / 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, B_REP, 0);
                }
                a[g] = aptr[64 * (GPC + g)];        // refilled in place for the next tap
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            bt = bt_next;
            aptr += 64 * GPC;
        }
    };

    const int bcol1 = half * STRIDE + wn * TNW + l31 + d.col0;
#pragma unroll
    for (int chunk = 0; chunk < NCH; ++chunk) {
        float* buf = lds + chunk * (CK * STRIDE);
        stg.store(d, chunk, buf, w, lane);
        __syncthreads();
        if (chunk + 1 < NCH) stg.issue(d, chunk + 1, w);
        run_chunk(buf + bcol1, p.taps, p.dilation);
    }

    // ---- second conv's weights and bias are requested before h is handed over ----
    aptr = reinterpret_cast<const float4*>(q.w2) + ((int64_t)wm * q.groups2) * 64 + lane;
#pragma unroll
    for (int g = 0; g < GPC; ++g) a[g] = aptr[64 * g];
    float bv2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bv2[r] = q.bias2 ? q.bias2[wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] : 0.0f;

    __syncthreads();                    // every wave is done reading the x tile
    {
        float* hrow = lds + (wm * 32 + 4 * half) * STRIDE + wn * TNW + l31;
#pragma unroll
        for (int n = 0; n < B_REP; ++n) {
            const int t = t0 - h2 + wn * TNW + 32 * n + l31;
            const bool inside = (t >= 0) && (t < L);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = vk_lrelu(acc[n][r], p.in_slope);
                hrow[((r & 3) + 8 * (r >> 2)) * STRIDE + 32 * n] = inside ? v : 0.0f;
            }
        }
    }
#pragma unroll
    for (int n = 0; n < B_REP; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = bv2[r];
    __syncthreads();

    // ------------------------------ second convolution ------------------------------
    const int bcol2 = half * STRIDE + wn * TNW + l31;
#pragma unroll
    for (int chunk = 0; chunk < NCH; ++chunk)
        run_chunk(lds + chunk * (CK * STRIDE) + bcol2, p.taps, 1);

    pair_epilogue<B_REP, EPI, TNO>(d, acc, b, t0, wm, wn, half, l31);
}

struct PairGeom { int b_rep, wm, wn, ck; };
// fused kernels exist for 32 and 64 channels
static constexpr PairGeom kPair32 = {4, 1, 4, 32};      // 512-column window, 74 KB of LDS: two workgroups per CU
static constexpr PairGeom kPair64 = {3, 2, 2, 32};      // 192-column window, 66.5 KB

// Returns hipSuccess / a HIP error, or -1 when the combination is not built.
int vk_launch_pair(int channels, int epi, int vec, const PairDev& q, dim3 grid, size_t lds, hipStream_t s);
