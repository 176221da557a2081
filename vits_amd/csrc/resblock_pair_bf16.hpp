// The fused resblock pair (see resblock_pair.hpp) with bf16 matrix-core inputs:
//
//   h   = conv1(bf16(lrelu(x)))        taps k, dilation d
//   out = conv2(bf16(lrelu(h))) + x    taps k, dilation 1      (fp32 accumulation, fp32 x / out in memory)
//
// The x window is staged channel-minor in bf16 exactly as in conv_bf16_kernel; h is activated,
// rounded and written channel-minor into the same LDS space (zero outside [0, L)), from where
// the second convolution takes its B fragments.  In bf16 the whole-channel h tile of a 256
// column window fits LDS up to 256 channels.  The same roundings as two separate conv_bf16_kernel
// launches; the residual enters the sum first instead of last (bias + x + products).
#pragma once
#include "conv_bf16.hpp"
#include "resblock_pair.hpp"

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// Diagnostic build with -DVK_PAIR_FINE only (tools/pair_stamps.py): wave 0's time inside the first
// convolution's chunk loop, split into waiting for a chunk's loads, converting and storing it,
// the barrier, and the tap loops.  Empty otherwise.  Two more timing switches of that build
// (results invalid): -DVK_PAIR_NOLOAD loads only the first chunk, -DVK_PAIR_ONE_WG asks for enough
// LDS that one workgroup fits a CU.
#if defined(VK_DIAG) && defined(VK_PAIR_FINE)
struct PairFine {
    unsigned long long t[4] = {0, 0, 0, 0}, m = 0;
    __device__ __forceinline__ void mark() { m = __builtin_amdgcn_s_memrealtime(); }
    __device__ __forceinline__ void lap(int i, bool count = true)
    {
        const unsigned long long n = __builtin_amdgcn_s_memrealtime();
        if (count) t[i] += n - m;
        m = n;
    }
    __device__ __forceinline__ void wait_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __device__ __forceinline__ void flush(const ConvDev& d)
    {
        if (d.stamps != nullptr && threadIdx.x == 0) {      // four 16-bit fields of 10 ns ticks in stamp slot 4
            const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            d.stamps[(size_t)lin * 8 + 4] = (t[0] & 0xffff) | ((t[1] & 0xffff) << 16) | ((t[2] & 0xffff) << 32) | ((t[3] & 0xffff) << 48);
        }
    }
};
#else
struct PairFine {
    __device__ __forceinline__ void mark() {}
    __device__ __forceinline__ void lap(int, bool = true) {}
    __device__ __forceinline__ void wait_loads() {}
    __device__ __forceinline__ void flush(const ConvDev&) {}
};
#endif
#if defined(VK_DIAG) && defined(VK_PAIR_NOLOAD)
constexpr bool kPairLoadAllChunks = false;
#else
constexpr bool kPairLoadAllChunks = true;
#endif

template <int A_REP, int B_REP, int WM, int WN, int EPI>
__global__ void __launch_bounds__(64 * WM * WN, 2)
resblock_pair_bf16_kernel(const PairDev q)
{
    constexpr int CK  = 32;
    constexpr int KS  = CK / 16;
    constexpr int NW  = WM * WN;
    constexpr int TNW = 32 * B_REP;
    constexpr int TN  = TNW * WN;
    constexpr int TNO = TN - VK_PAIR_OVERLAP;
    constexpr int NJ  = TN / 64 + 1;
    constexpr int C   = 32 * A_REP * WM;
    constexpr int NCH = C / CK;
    constexpr int XP  = CK * 2 + 16;            // bytes per staged x column (one chunk)
    constexpr int XBUF = (TN + 64) * XP;
    constexpr int HP  = C * 2 + 16;             // bytes per h column (all channels)

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

    const float* xb = p.x + (int64_t)b * p.x_bstride;
    vk_stamp(d, 0);         // diagnostic build: phase time stamps (tools/pair_stamps.py)
    Bf16Stager<NW, NJ, CK, XP, true> stg;
    stg.init(d, xb, t0, lane, w);
    stg.issue(d, 0);

    // ------------------------------ first convolution ------------------------------
    const int mt0 = wm * A_REP;                 // first 32-channel tile of this wave
    const bf16x8* aptr[A_REP];
#pragma unroll
    for (int m = 0; m < A_REP; ++m)
        aptr[m] = reinterpret_cast<const bf16x8*>(p.w) + ((int64_t)(mt0 + m) * d.groups_per_mt) * 64 + lane;
    bf16x8 a[VK_PF][KS][A_REP];
    bf16_wload<KS, A_REP>(a, aptr);
    // the second conv's weight stream: the first one's last taps already fetch its first ones
    const bf16x8* w2ptr[A_REP];
#pragma unroll
    for (int m = 0; m < A_REP; ++m)
        w2ptr[m] = reinterpret_cast<const bf16x8*>(q.w2) + ((int64_t)(mt0 + m) * q.groups2) * 64 + lane;
    const bool chained = p.taps >= VK_PF;       // (a shorter kernel's ring would still hold taps of the first stream)

    f32x16 acc[A_REP][B_REP];
#pragma unroll
    for (int m = 0; m < A_REP; ++m) {
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = p.bias ? p.bias[(mt0 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] : 0.0f;
#pragma unroll
        for (int n = 0; n < B_REP; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = bv[r];
    }

    const int bcol1 = (wn * TNW + l31) * XP + half * 16;
    for (int chunk = 0; chunk < NCH; ++chunk) {
        char* buf = lds8 + (chunk & 1) * XBUF;
        stg.store(d, chunk, buf);
        __syncthreads();
        if (chunk == 0) vk_stamp(d, 1);
        if (chunk + 1 < NCH) stg.issue(d, chunk + 1);
        bf16_taps<A_REP, B_REP, KS, XP>(acc, a, aptr, buf + bcol1, p.taps, p.dilation * XP,
                                        w2ptr, chained && chunk + 1 == NCH);
    }
    vk_stamp(d, 2);

    // ---- second conv's weights: in the ring already, or requested before h is handed over ----
#pragma unroll
    for (int m = 0; m < A_REP; ++m) aptr[m] = w2ptr[m];
    if (!chained) bf16_wload<KS, A_REP>(a, aptr);

    __syncthreads();                    // every wave is done reading the x tile
    {
        // register r of a lane is channel 32*tile + (r & 3) + 8*(r >> 2) + 4*half of its column:
        // four consecutive channels per group of four registers -> one 8-byte LDS store
        char* hcol = lds8 + (wn * TNW + l31) * HP + (mt0 * 32 + 4 * half) * 2;
#pragma unroll
        for (int n = 0; n < B_REP; ++n) {
            const int t = t0 - h2 + wn * TNW + 32 * n + l31;
            const bool inside = (t >= 0) && (t < L);
#pragma unroll
            for (int m = 0; m < A_REP; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        float v0 = vk_lrelu(acc[m][n][4 * g + e], p.in_slope);
                        float v1 = vk_lrelu(acc[m][n][4 * g + e + 1], p.in_slope);
                        v0 = inside ? v0 : 0.0f;
                        v1 = inside ? v1 : 0.0f;
                        const f32x2 pr = {v0, v1};
                        const bf16x2 c2 = __builtin_convertvector(pr, bf16x2);
                        o[e] = c2[0]; o[e + 1] = c2[1];
                    }
                    *reinterpret_cast<bf16x4*>(hcol + n * 32 * HP + m * 64 + g * 16) = o;
                }
        }
    }
    // The second conv's accumulators start at bias + x, the residual: its loads are in flight while
    // the workgroup meets at the barrier below, and the epilogue is left with stores only (read
    // there, x stood between the last matrix instruction and the first store of every workgroup).
    {
        const __amdgpu_buffer_rsrc_t r_res = vk_rsrc(p.res + (int64_t)b * p.res_bstride, d.res_bytes);
        const int vr = 4 * (4 * half * (int)p.res_cstride + t0 + wn * TNW + l31);
#pragma unroll
        for (int m = 0; m < A_REP; ++m) {
            float bv2[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bv2[r] = q.bias2 ? q.bias2[(mt0 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] : 0.0f;
#pragma unroll
            for (int n = 0; n < B_REP; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = (mt0 + m) * 32 + (r & 3) + 8 * (r >> 2);
                    acc[m][n][r] = bv2[r] + vk_bld(r_res, vr + 128 * n, rr * (int)p.res_cstride * 4);
                }
        }
    }
    __syncthreads();
    vk_stamp(d, 3);

    // ------------------------------ second convolution ------------------------------
    const int bcol2 = (wn * TNW + l31) * HP + half * 16;
    for (int chunk = 0; chunk < NCH; ++chunk)
        bf16_taps<A_REP, B_REP, KS, HP>(acc, a, aptr, lds8 + bcol2 + chunk * (CK * 2), p.taps, HP);
    vk_stamp(d, 5);

#pragma unroll
    for (int m = 0; m < A_REP; ++m) pair_epilogue<B_REP, EPI, TNO, true>(d, acc[m], b, t0, mt0 + m, wn, half, l31);
    vk_stamp(d, 6);
}

struct PairGeomBf16 { int a_rep, b_rep, wm, wn; };
static constexpr PairGeomBf16 kPairBf16_32  = {1, 4, 1, 4};     // 512-column window, 4 waves
static constexpr PairGeomBf16 kPairBf16_64  = {1, 4, 2, 2};     // 256-column window, 4 waves
static constexpr PairGeomBf16 kPairBf16_128 = {2, 4, 2, 2};     // 256-column window, 4 waves of 64 x 128 (8-wave splits measured slower)
static constexpr PairGeomBf16 kPairBf16_256 = {2, 4, 4, 2};     // 256-column window, 8 waves of 64 x 128 (h tile 142 KB)

// LDS bytes of a workgroup: the larger of the x staging buffers and the h tile (+ the columns
// the second conv's invalid outputs read past the window)
static inline size_t pair_bf16_lds(const PairGeomBf16& G)
{
    const int TN = 32 * G.b_rep * G.wn, C = 32 * G.a_rep * G.wm, nch = C / 32;
    const size_t x = (size_t)(nch < 2 ? nch : 2) * (TN + 64) * (32 * 2 + 16);
    const size_t h = (size_t)(TN + VK_PAIR_OVERLAP) * (C * 2 + 16);
#if defined(VK_DIAG) && defined(VK_PAIR_ONE_WG)
    return 100 * 1024;
#endif
    return x > h ? x : h;
}

int vk_launch_pair_bf16(int channels, int epi, const PairDev& q, dim3 grid, size_t lds, hipStream_t s);
