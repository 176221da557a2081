// One launch for a whole multi-receptive-field stage of the HiFi-GAN decoder
// (/root/reference/module/model_component/decoder.py:119-125 around :46-61):
//
//   out = ( sum over resblocks r of  resblock_r(x) ) / n_blocks
//   resblock_r:  for each pair j:  h = conv1_rj(lrelu(x_r));  x_r = conv2_rj(lrelu(h)) + x_r
//
// with bf16 matrix-core inputs and fp32 accumulation (the arithmetic of conv_bf16.hpp and
// resblock_pair_bf16.hpp: activations are leaky-ReLU'd and rounded to bf16 where they enter a
// matrix instruction; bias, residual chain and the sum over resblocks stay fp32).
//
// A workgroup owns every channel of a window of W = 32 * NT * WN time steps (WN waves along time,
// NT adjacent 32-column tiles per wave), of which the inner W - 2 * halo are outputs; halo is the
// longest receptive half-width among the launch's resblocks (12 / 36 / 60 samples for kernel
// sizes 3 / 7 / 11 with dilations 1, 3, 5).  Nothing but the stage input and the stage output touches HBM:
//   * the residual stream x_r lives in REGISTERS, in accumulator layout: a wave owns 32 channels
//     of its NT tiles, so `x_r = conv2(...) + x_r` is a register add and the sum over the
//     resblocks (kernel sizes) is accumulated in registers too;
//   * what the convolutions read through the matrix cores - bf16(lrelu(x_r)) and bf16(lrelu(h)) -
//     lives in LDS, channel-minor [column][channel] like conv_bf16.hpp, so a B fragment is one
//     16-byte read and a tap is a column offset; a wave writes its own tiles after each
//     convolution and reads its neighbours' columns as halo after a barrier.  With two images
//     (TWO) a convolution reads one and its result goes to the other: one barrier per
//     convolution.  With one image the result replaces what was read (a second barrier in
//     between) and the window can be twice as wide for the same LDS;
//   * weights stream from L2 in fragment order VK_PF taps ahead, and the first fragments of the
//     next convolution are requested before the current one starts (a convolution of a
//     32-channel stage is only a few dozen matrix instructions per wave).
// Columns outside [0, L) are forced to zero after every convolution (the zero padding each
// reference conv sees); columns of the window that a later convolution cannot reach any more
// hold don't-care values that never reach an output column.
#pragma once
#include "conv_bf16.hpp"
#include "resblock_pair_bf16.hpp"

// Diagnostic build only (-DVK_DIAG -DVK_MRF_ABL=n, results invalid): timing experiments that take a
// part of the kernel away.  1: no activation / rounding / LDS store of a conv's result, 2: no
// barriers, 4: no matrix instructions.
#if defined(VK_DIAG) && defined(VK_MRF_ABL)
constexpr int kMrfAbl = VK_MRF_ABL;
#else
constexpr int kMrfAbl = 0;
#endif
#define VK_MRF_SYNC() do { if (!(kMrfAbl & 2)) __syncthreads(); } while (0)

constexpr int VK_MRF_HALO   = 64;
constexpr int VK_MRF_GUARD  = 32;                           // addressable columns on either side of an image

// geometry by channel count: waves along channels (32 each) x waves along time, tiles per wave
// sum = 1: the launch sums several resblocks (a third register copy of the window);
// sum = 0: one resblock per launch, out = resblock(x)
struct MrfGeom { int wm, wn, nt, two, sum; };
static constexpr MrfGeom kMrf32  = {1, 4, 4, 0, 1};         // 512-column window, 4 waves, one image (46 KB)
static constexpr MrfGeom kMrf64  = {2, 4, 4, 0, 1};         // 512-column window, 8 waves, one image (83 KB)
static constexpr MrfGeom kMrf32s = {1, 4, 4, 0, 0};         // the same windows for one resblock per launch: no
static constexpr MrfGeom kMrf64s = {2, 4, 4, 0, 0};         // register copy for the sum, weights two taps ahead
static constexpr MrfGeom kMrf128 = {4, 2, 4, 0, 0};         // 256-column window, 8 waves, one image (87 KB)
static constexpr MrfGeom kMrf256 = {8, 1, 4, 0, 0};         // 128-column window, 8 waves, one image (101 KB)
// the widest launch a channel count has (`single`: the kernel for one resblock per launch)
static inline const MrfGeom* vk_mrf_geom(int channels, bool single = false)
{
    return channels == 32 ? (single ? &kMrf32s : &kMrf32) : channels == 64 ? (single ? &kMrf64s : &kMrf64) :
           channels == 128 ? &kMrf128 : channels == 256 ? &kMrf256 : nullptr;
}
static inline int vk_mrf_window(const MrfGeom& g) { return 32 * g.nt * g.wn; }

struct MrfDev {
    vk_mrf_args a;
    unsigned x_bytes, out_bytes, acc_bytes;    // byte spans of one batch item's views
    int halo, out_cols;             // recomputed columns on either side of a window; output columns per workgroup
    int groups[VK_MRF_MAX_BLOCKS];  // 64-lane x 16-byte groups per 32-row tile of a conv of block r: chunks * taps * 2
};

static inline size_t vk_mrf_lds_bytes(const MrfGeom& g)
{
    const int channels = 32 * g.wm;
    return (size_t)(g.two ? 2 : 1) * (vk_mrf_window(g) + 2 * VK_MRF_GUARD) * (channels * 2 + 16)
         + (size_t)VK_MRF_MAX_BLOCKS * VK_MRF_MAX_PAIRS * 2 * channels * sizeof(float);
}

template <int WM, int WN, int NT, bool TWO, bool SUM>
__global__ void __launch_bounds__(64 * WM * WN, 2)
mrf_stage_bf16_kernel(const MrfDev q)
{
    constexpr int WIN = 32 * NT * WN;           // window columns
    constexpr int C   = 32 * WM;
    constexpr int NCH = C / 32;                 // K-chunks of 32 channels (two matrix instructions per tap each)
    constexpr int KS  = 2;
    constexpr int PITCH = C * 2 + 16;           // bytes per column (odd multiple of 16: conflict-free 16-byte reads)
    constexpr int IMG = (WIN + 2 * VK_MRF_GUARD) * PITCH;
    // taps of weight fragments in flight (registers refilled in place); three register copies of a
    // four-tile window leave room for one (with two the compiler spills, and spills are HBM traffic)
    constexpr int PF  = (SUM && NT >= 4) ? 1 : VK_PF;

    extern __shared__ __attribute__((aligned(16))) char lds8[];
    char* img_a = lds8;                         // bf16(lrelu(x_r)): input of the dilated convs
    char* img_h = TWO ? lds8 + IMG : lds8;      // bf16(lrelu(h)):   input of the plain convs
    float* bias_tab = reinterpret_cast<float*>(lds8 + (TWO ? 2 : 1) * IMG);

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
    const int col0 = wn * (32 * NT) + l31;      // this lane's window column in its first tile
    const int tw   = t0 - q.halo + col0;        // ... and its time step

    // every conv's bias, once, into LDS: entry ((r * n_pairs + j) * 2 + which) * C + channel
    for (int i = threadIdx.x; i < p.n_blocks * p.n_pairs * 2 * C; i += 64 * WM * WN) {
        const int conv = i / C, ch = i - conv * C;
        const int rj = conv >> 1, r = rj / p.n_pairs, j = rj - r * p.n_pairs;
        const float* bp = (conv & 1) ? p.bias2[r][j] : p.bias1[r][j];
        bias_tab[i] = bp ? bp[ch] : 0.0f;
    }

    bool inside[NT];                            // this lane's column of tile n lies in [0, L)
#pragma unroll
    for (int n = 0; n < NT; ++n) inside[n] = (tw + 32 * n >= 0) && (tw + 32 * n < L);
    // most windows lie wholly inside the sequence: their workgroups skip the per-element selects
    const bool edge = (t0 - q.halo < 0) || (t0 - q.halo + WIN > L);     // workgroup-uniform

    // This wave's weight stream.  The ring `a` holds the next PF taps; the last taps of a convolution
    // already fetch the first ones of the next (every kernel has at least PF taps: checked on the host).
    auto wfrag = [&](const void* wp, int groups) {
        return reinterpret_cast<const bf16x8*>(wp) + ((int64_t)wm * groups) * 64 + lane;
    };
    bf16x8 a[PF][KS][1];
    const bf16x8* aptr[1] = {wfrag(p.w1[0][0], q.groups[0])};
    bf16_wload<KS, 1, PF>(a, aptr);

    // accumulator-layout register r of a lane is channel 32*wm + (r & 3) + 8*(r >> 2) + 4*half;
    // four consecutive registers are four consecutive channels: one 8-byte store into an image
    auto put_image_as = [&](char* img, const f32x16 (&v)[NT], auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        if (kMrfAbl & 1) return;
        char* dst = img + (VK_MRF_GUARD + col0) * PITCH + (wm * 32 + 4 * half) * 2;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    float v0 = vk_lrelu(v[n][4 * g + e], p.slope);
                    float v1 = vk_lrelu(v[n][4 * g + e + 1], p.slope);
                    if (MASKED) { v0 = inside[n] ? v0 : 0.0f; v1 = inside[n] ? v1 : 0.0f; }
                    const f32x2 pr = {v0, v1};
                    const bf16x2 c2 = __builtin_convertvector(pr, bf16x2);
                    o[e] = c2[0]; o[e + 1] = c2[1];
                }
                *reinterpret_cast<bf16x4*>(dst + n * 32 * PITCH + g * 16) = o;
            }
    };
    // two copies of the code, chosen by a workgroup-uniform branch: the selects cost a vector
    // instruction per element in a phase that is all vector instructions
    auto put_image = [&](char* img, const f32x16 (&v)[NT], bool act_only_inside) {
        if (act_only_inside && edge) put_image_as(img, v, std::true_type{});
        else                         put_image_as(img, v, std::false_type{});
    };

    // one convolution of this wave's three tiles: bias, then every chunk's taps out of an image
    // `onto`: the products are added to what `dst` holds (plus the bias) instead of starting at the bias
    f32x16 acc[1][NT];
    auto conv = [&](f32x16 (&dst)[1][NT], bool onto, const char* img, int conv_index, int taps, int dilation,
                    const void* w_next, int groups_next) {
        const float* bt = bias_tab + conv_index * C + wm * 32 + 4 * half;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float bv = bt[(r & 3) + 8 * (r >> 2)];
#pragma unroll
            for (int n = 0; n < NT; ++n) dst[0][n][r] = onto ? dst[0][n][r] + bv : bv;
        }
        const bf16x8* next[1] = {w_next != nullptr ? wfrag(w_next, groups_next) : aptr[0]};      // wave-uniform
        const int pad = ((taps - 1) >> 1) * dilation;
        const char* base = img + (VK_MRF_GUARD + col0 - pad) * PITCH + half * 16;
        if (kMrfAbl & 4) return;
        for (int chunk = 0; chunk < NCH; ++chunk)
            bf16_taps<1, NT, KS, PITCH, PF>(dst, a, aptr, base + chunk * 64, taps, dilation * PITCH,
                                            next, w_next != nullptr && chunk + 1 == NCH);
        if (w_next != nullptr) aptr[0] = next[0];
    };

    const __amdgpu_buffer_rsrc_t r_x = vk_rsrc(p.x + (int64_t)b * p.x_bstride, q.x_bytes);
    f32x16 xr[1][NT], tot[SUM ? NT : 1];
    if (SUM) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) tot[n][r] = 0.0f;
    }

    for (int blk = 0; blk < p.n_blocks; ++blk) {
        // the stage input, straight into accumulator layout (32 lanes = 128 contiguous bytes of a row;
        // the row of a register differs between the two halves of the wave, hence a per-lane row term)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int t = tw + 32 * n;
            const int tc = t < 0 ? 0 : (t >= L ? L - 1 : t);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 + (r & 3) + 8 * (r >> 2);                  // wave-uniform part
                const int voff = 4 * (tc + 4 * half * (int)p.x_cstride);
                const float v = vk_bld(r_x, voff, row * (int)p.x_cstride * 4);
                xr[0][n][r] = inside[n] ? v : 0.0f;
            }
        }
        put_image(img_a, xr[0], false);
        VK_MRF_SYNC();

        const int taps = p.taps[blk];
        for (int j = 0; j < p.n_pairs; ++j) {
            const int ci = (blk * p.n_pairs + j) * 2;
            // ---- h = conv1(lrelu(x_r)), dilated ----
            conv(acc, false, img_a, ci, taps, p.dilation[blk][j], p.w2[blk][j], q.groups[blk]);
            if (!TWO) VK_MRF_SYNC();          // every wave has read what the next lines overwrite
            put_image(img_h, acc[0], true);
            VK_MRF_SYNC();
            // ---- x_r = conv2(lrelu(h)) + x_r: the matrix instructions accumulate onto x_r + bias ----
            const bool last_pair = j + 1 == p.n_pairs;
            const bool last_conv = last_pair && blk + 1 == p.n_blocks;
            const void* wn1 = last_conv ? nullptr : (last_pair ? p.w1[blk + 1][0] : p.w1[blk][j + 1]);
            const int gn1 = last_conv ? 0 : (last_pair ? q.groups[blk + 1] : q.groups[blk]);
            conv(xr, true, img_h, ci + 1, taps, 1, wn1, gn1);     // accumulated straight onto the residual stream
            if (edge) {
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) xr[0][n][r] = inside[n] ? xr[0][n][r] : 0.0f;
            }
            if (!last_pair) {
                if (!TWO) VK_MRF_SYNC();
                put_image(img_a, xr[0], false);
                VK_MRF_SYNC();
            }
        }
        if (SUM) {
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) tot[n][r] += xr[0][n][r];
        }
        if (!TWO && blk + 1 < p.n_blocks) VK_MRF_SYNC();     // the next resblock's input replaces h
    }

    // ---- out = (acc + sum) / navg for the window's output columns ----
    const __amdgpu_buffer_rsrc_t r_out = vk_rsrc(p.out + (int64_t)b * p.out_bstride, q.out_bytes);
    const __amdgpu_buffer_rsrc_t r_acc = vk_rsrc(p.acc ? p.acc + (int64_t)b * p.acc_bstride : p.x, p.acc ? q.acc_bytes : 0u);
    const float nb = p.navg;
    const bool scale = nb != 1.0f;                          // launch-uniform
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
                float v = SUM ? tot[n][r] : xr[0][n][r];
                if (p.acc != nullptr) v = av[r] + v;
                if (scale) v = v / nb;
                vk_bst(v, r_out, 4 * (t + 4 * half * (int)p.out_cstride), row * (int)p.out_cstride * 4);
            }
        }
    }
}

// Returns hipSuccess / a HIP error, or -1 when there is no kernel for this channel count.
int vk_launch_mrf_bf16(int channels, bool single, const MrfDev& q, dim3 grid, size_t lds, hipStream_t s);
