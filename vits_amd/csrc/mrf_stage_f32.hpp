// The one-launch multi-receptive-field stage / resblock chain of mrf_stage.hpp on the exact fp32
// matrix instruction (v_mfma_f32_32x32x2_f32), for the decoder's 32- and 64-channel stages
// (/root/reference/module/model_component/decoder.py:119-125 around :46-61):
//
//   out = ( acc + sum over resblocks r of resblock_r(x) ) / navg
//   resblock_r:  for each pair j:  h = conv1_rj(lrelu(x_r));  x_r = conv2_rj(lrelu(h)) + x_r
//
// Same ownership as the bf16 kernel: a workgroup owns every channel of a window of
// W = 32 * NT * WN time steps, of which the inner W - 2 * halo are outputs (halo = the longest
// receptive half-width among the launch's resblocks); the residual stream x_r and the sum over
// the resblocks live in registers in accumulator layout; what the convolutions read through the
// matrix cores - lrelu(x_r), lrelu(h) - lives in ONE fp32 LDS image [channel][column] (the layout
// of conv_mfma.hpp's staged tiles: a B fragment is a 4-byte read per lane, a tap is a column
// offset), rewritten in place after every convolution between two barriers.  Columns outside
// [0, L) are forced to zero after every convolution; window columns a later convolution cannot
// reach any more hold don't-care values that never reach an output column.
// Every product and sum is the fp32 fma chain of conv_mfma.hpp in the same order (channel chunk,
// tap, channel pair), so a chain of kernel-size-3 pairs is bit-identical to the pair launches.
// Weights are packed as for resblock_pair_kernel (32-channel chunks, fragment order).
#pragma once
#include "mrf_stage.hpp"

// geometry by channel count (waves along channels x waves along time, tiles per wave)
static constexpr MrfGeom kMrfF32_32  = {1, 4, 4, 0, 1};     // 512-column window, 4 waves, 74 KB image: two workgroups per CU
static constexpr MrfGeom kMrfF32_64  = {2, 4, 4, 0, 1};     // 512-column window, 8 waves, 147 KB image
static constexpr MrfGeom kMrfF32_32s = {1, 4, 4, 0, 0};     // one resblock per launch (no register copy for the sum)
static constexpr MrfGeom kMrfF32_64s = {2, 4, 4, 0, 0};
static inline const MrfGeom* vk_mrf_f32_geom(int channels, bool single = false)
{
    return channels == 32 ? (single ? &kMrfF32_32s : &kMrfF32_32) :
           channels == 64 ? (single ? &kMrfF32_64s : &kMrfF32_64) : nullptr;
}
static inline size_t vk_mrf_f32_lds_bytes(const MrfGeom& g)
{
    const int channels = 32 * g.wm;
    return ((size_t)channels * (vk_mrf_window(g) + 2 * VK_MRF_GUARD)
            + (size_t)VK_MRF_MAX_BLOCKS * VK_MRF_MAX_PAIRS * 2 * channels) * sizeof(float);
}

template <int WM, int WN, int NT, bool SUM>
__global__ void __launch_bounds__(64 * WM * WN, 2)
mrf_stage_f32_kernel(const MrfDev q)
{
    constexpr int WIN = 32 * NT * WN;               // window columns
    constexpr int C   = 32 * WM;
    constexpr int CK  = 32;                         // channels per packed chunk
    constexpr int NCH = C / CK;
    constexpr int GPC = CK / 8;                     // weight groups (4 channel pairs each) per chunk and tap
    constexpr int STRIDE = WIN + 2 * VK_MRF_GUARD;  // floats between image rows

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* img = lds;                               // lrelu(x_r) / lrelu(h): what the next convolution reads
    float* bias_tab = lds + C * STRIDE;

    const vk_mrf_args& p = q.a;
    const int lane = threadIdx.x & 63;
    const int w    = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm   = w % WM;
    const int wn   = w / WM;
    const int half = lane >> 5;
    const int l31  = lane & 31;
    const int b    = blockIdx.z;
    const int t0   = blockIdx.x * q.out_cols;
    const int L    = p.length;
    const int col0 = wn * (32 * NT) + l31;          // this lane's window column in its first tile
    const int tw   = t0 - q.halo + col0;            // ... and its time step

    // every conv's bias, once, into LDS: entry ((r * n_pairs + j) * 2 + which) * C + channel
    for (int i = threadIdx.x; i < p.n_blocks * p.n_pairs * 2 * C; i += 64 * WM * WN) {
        const int conv = i / C, ch = i - conv * C;
        const int rj = conv >> 1, r = rj / p.n_pairs, j = rj - r * p.n_pairs;
        const float* bp = (conv & 1) ? p.bias2[r][j] : p.bias1[r][j];
        bias_tab[i] = bp ? bp[ch] : 0.0f;
    }

    bool inside[NT];                                // this lane's column of tile n lies in [0, L)
#pragma unroll
    for (int n = 0; n < NT; ++n) inside[n] = (tw + 32 * n >= 0) && (tw + 32 * n < L);
    const bool edge = (t0 - q.halo < 0) || (t0 - q.halo + WIN > L);     // workgroup-uniform

    // This wave's weight stream: one float4 per lane per four channel pairs.  `a` holds the groups
    // of the next tap; each is refilled right after its last use, and the last tap of a
    // convolution already fetches the first tap of the next one.
    auto wfrag = [&](const void* wp, int groups) {
        return reinterpret_cast<const float4*>(wp) + ((int64_t)wm * groups) * 64 + lane;
    };
    const float4* aptr = wfrag(p.w1[0][0], q.groups[0]);
    float4 a[GPC];
#pragma unroll
    for (int g = 0; g < GPC; ++g) a[g] = aptr[64 * g];

    // accumulator-layout register r of a lane is channel 32*wm + (r & 3) + 8*(r >> 2) + 4*half
    auto put_image_as = [&](const f32x16 (&v)[NT], auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        float* dst = img + (wm * 32 + 4 * half) * STRIDE + VK_MRF_GUARD + col0;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float val = vk_lrelu(v[n][r], p.slope);
                if (MASKED) val = inside[n] ? val : 0.0f;
                dst[((r & 3) + 8 * (r >> 2)) * STRIDE + 32 * n] = val;
            }
    };
    auto put_image = [&](const f32x16 (&v)[NT], bool act_only_inside) {
        if (act_only_inside && edge) put_image_as(v, std::true_type{});
        else                         put_image_as(v, std::false_type{});
    };

    // one convolution of this wave's tiles: accumulators start at the bias, then every chunk's
    // taps out of the image
    f32x16 acc[NT];
    auto conv = [&](f32x16 (&dst)[NT], int conv_index, int taps, int dilation,
                    const void* w_next, int groups_next) {
        const float* bt0 = bias_tab + conv_index * C + wm * 32 + 4 * half;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float bv = bt0[(r & 3) + 8 * (r >> 2)];
#pragma unroll
            for (int n = 0; n < NT; ++n) dst[n][r] = bv;
        }
        const float4* next = w_next != nullptr ? wfrag(w_next, groups_next) : aptr;      // wave-uniform
        const int pad = ((taps - 1) >> 1) * dilation;
        const float* base0 = img + half * STRIDE + VK_MRF_GUARD + col0 - pad;
        for (int chunk = 0; chunk < NCH; ++chunk) {
            const float* base = base0 + chunk * (CK * STRIDE);
            const float* bt = base;
            float bf[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) bf[n] = bt[32 * n];
#pragma unroll 1
            for (int tap = 0; tap < taps; ++tap) {
                const bool last_tap = tap + 1 == taps;
                const float* bt_next = last_tap ? (chunk + 1 < NCH ? base + CK * STRIDE : base0) : bt + dilation;
                // where the groups of the tap after this one come from: this stream, or the next convolution's
                const float4* refill = (last_tap && chunk + 1 == NCH && w_next != nullptr) ? next - 64 * GPC : aptr;
#pragma unroll
                for (int g = 0; g < GPC; ++g) {
                    const float4 a_cur = a[g];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        constexpr int KSTEPS = 4 * GPC;
                        const int ks = 4 * g + e;
                        float bn[NT];
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            bn[n] = (ks + 1 < KSTEPS) ? bt[2 * (ks + 1) * STRIDE + 32 * n] : bt_next[32 * n];
                        const float av = (e == 0) ? a_cur.x : (e == 1) ? a_cur.y : (e == 2) ? a_cur.z : a_cur.w;
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            dst[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf[n], dst[n], 0, 0, 0);
#pragma unroll
                        for (int n = 0; n < NT; ++n) bf[n] = bn[n];
                        __builtin_amdgcn_sched_group_barrier(0x100, (NT + 1) / 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
                    }
                    a[g] = refill[64 * (GPC + g)];      // refilled in place for the next tap
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                bt = bt_next;
                aptr += 64 * GPC;
            }
        }
        if (w_next != nullptr) aptr = next;
    };

    const __amdgpu_buffer_rsrc_t r_x = vk_rsrc(p.x + (int64_t)b * p.x_bstride, q.x_bytes);
    f32x16 xr[NT], tot[SUM ? NT : 1];
    if (SUM) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) tot[n][r] = 0.0f;
    }

    for (int blk = 0; blk < p.n_blocks; ++blk) {
        // the stage input, straight into accumulator layout
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int t = tw + 32 * n;
            const int tc = t < 0 ? 0 : (t >= L ? L - 1 : t);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 + (r & 3) + 8 * (r >> 2);                  // wave-uniform part
                const int voff = 4 * (tc + 4 * half * (int)p.x_cstride);
                const float v = vk_bld(r_x, voff, row * (int)p.x_cstride * 4);
                xr[n][r] = inside[n] ? v : 0.0f;
            }
        }
        if (blk > 0) __syncthreads();               // every wave is done reading the previous resblock's last image
        put_image(xr, false);
        __syncthreads();

        const int taps = p.taps[blk];
        for (int j = 0; j < p.n_pairs; ++j) {
            const int ci = (blk * p.n_pairs + j) * 2;
            // ---- h = conv1(lrelu(x_r)), dilated ----
            conv(acc, ci, taps, p.dilation[blk][j], p.w2[blk][j], q.groups[blk]);
            __syncthreads();                        // every wave has read what the next lines overwrite
            put_image(acc, true);
            __syncthreads();
            // ---- x_r = conv2(lrelu(h)) + x_r ----
            const bool last_pair = j + 1 == p.n_pairs;
            const bool last_conv = last_pair && blk + 1 == p.n_blocks;
            const void* wn1 = last_conv ? nullptr : (last_pair ? p.w1[blk + 1][0] : p.w1[blk][j + 1]);
            const int gn1 = last_conv ? 0 : (last_pair ? q.groups[blk + 1] : q.groups[blk]);
            conv(acc, ci + 1, taps, 1, wn1, gn1);
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float y = acc[n][r] + xr[n][r];           // the order of the pair kernel's epilogue
                    xr[n][r] = (edge && !inside[n]) ? 0.0f : y;
                }
            if (!last_pair) {
                __syncthreads();
                put_image(xr, false);
                __syncthreads();
            }
        }
        if (SUM) {
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) tot[n][r] += xr[n][r];
        }
    }

    // ---- out = (acc + sum) / navg for the window's output columns ----
    const __amdgpu_buffer_rsrc_t r_out = vk_rsrc(p.out + (int64_t)b * p.out_bstride, q.out_bytes);
    const __amdgpu_buffer_rsrc_t r_acc = vk_rsrc(p.acc ? p.acc + (int64_t)b * p.acc_bstride : p.x, p.acc ? q.acc_bytes : 0u);
    const float nb = p.navg;
    const bool scale = nb != 1.0f;                  // launch-uniform
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int c = col0 + 32 * n;
        const int t = tw + 32 * n;
        if (c >= q.halo && c < q.halo + q.out_cols && t < L) {
            float av[16];
            if (p.acc != nullptr) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    av[r] = vk_bld(r_acc, 4 * (t + 4 * half * (int)p.acc_cstride),
                                   (wm * 32 + (r & 3) + 8 * (r >> 2)) * (int)p.acc_cstride * 4);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 + (r & 3) + 8 * (r >> 2);
                float v = SUM ? tot[n][r] : xr[n][r];
                if (p.acc != nullptr) v = av[r] + v;
                if (scale) v = v / nb;
                vk_bst(v, r_out, 4 * (t + 4 * half * (int)p.out_cstride), row * (int)p.out_cstride * 4);
            }
        }
    }
}

// Returns hipSuccess / a HIP error, or -1 when there is no kernel for this channel count.
int vk_launch_mrf_f32(int channels, bool single, const MrfDev& q, dim3 grid, size_t lds, hipStream_t s);
