// Token-level kernels of the stochastic duration predictor (reverse direction): one launch per
// layer of a dilated depth-separable stack, with the 1x1 convolutions before and after the
// stack and, for a coupling flow, the inverse rational-quadratic spline riding in the first and
// last layer's launch.  The tensors are a few dozen to a few hundred token columns of 192
// channels, so the launches are latency-bound: a workgroup owns eight columns, which spreads even
// one short sentence over several CUs, and has 768 threads - (channel, quarter): a thread does
// the pointwise work of two columns of its channel and, in the 1x1 convs, a quarter of the input
// channels for all eight columns (partials meet in LDS) - so that no thread runs a long serial
// instruction stream.  Everything between a layer's input and output (depthwise conv, two
// LayerNorms, two GELUs, the 1x1 conv, the residual) stays in registers and LDS.  fp32 on the vector ALUs: the matrix cores have no better
// fp32 rate than vector FMAs at these sizes, and the durations downstream are rounded up to
// integers, so no reduced precision here.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <type_traits>
#include "../../include/vits_amd.h"
#include "api_internal.hpp"
#include "conv_mfma.hpp"      // vk_allow_lds

namespace {

constexpr int DC = VK_DDS_CHANNELS;   // channels
constexpr int NC = VK_DDS_COLUMNS;    // token columns per workgroup
constexpr int NQ = 4;                 // threads per channel: column pairs / quarters of the input channels
constexpr int NT = DC * NQ;           // threads of a workgroup
constexpr int KQ = DC / NQ;           // input channels per quarter
constexpr int MAXDIL = 9;             // taps 3, up to 3 layers: dilations 1, 3, 9
constexpr int XW = NC + 2 * MAXDIL;   // widest window of layer input a workgroup reads
constexpr int NB = 10;                // spline bins
constexpr int NR = 3 * NB - 1;        // spline parameters per element
constexpr int XP = NR;                // row pitch of the window (odd; the buffer later holds the NR x DC projection)
constexpr int YP = 12;                // row pitch of the 1x1 input tile (up to 10 columns, 16-byte rows)
constexpr int NWAVE = NT / 64;
constexpr int WPQ = NWAVE / NQ;       // wavefronts per quarter
static_assert(XP >= XW && (XP & 1), "window pitch");

struct LayerDev {
    vk_dds_args a;
    const float* src;   // this layer's input [B, C, T] (dense); unused when first != 0
    float* dst;         // this layer's output [B, C, T] (dense); unused when last != 0
    const float* L;     // this layer's parameters
    int dil;
    int first;          // 0: read src   1: x = pre(x0) + g (coupling)   2: x = pre(x) + add_vec (conditioning)
    int last;           // 0: store x    1: out = proj(x) * mask         2: proj, spline, write z
    int final_layer;    // the stack's last layer: x * mask
};

__device__ inline float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }

// Sum over the 64 lanes: rotations inside each row of 16 lanes (DPP, no LDS traffic), then the
// four row totals through scalar registers.
__device__ inline float wave_sum(float v)
{
    auto ror = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    v += ror(v, std::integral_constant<int, 0x128>{});     // row_ror:8
    v += ror(v, std::integral_constant<int, 0x124>{});     // row_ror:4
    v += ror(v, std::integral_constant<int, 0x122>{});     // row_ror:2
    v += ror(v, std::integral_constant<int, 0x121>{});     // row_ror:1
    const int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16))
         + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48));
}

// Mean and 1/std over the 192 channels of the thread's two columns (the three wavefronts of
// its quarter hold them).
__device__ inline void column_norm(const float (&v)[2], float (&mean)[2], float (&rstd)[2], float (*red)[NWAVE][2])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, w0 = wave / WPQ * WPQ;
    float s[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) s[j] = wave_sum(v[j]);
    if (lane == 0) { red[0][wave][0] = s[0]; red[0][wave][1] = s[1]; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < WPQ; ++w) t += red[0][w0 + w][j];
        mean[j] = t * (1.0f / DC);
        const float d = v[j] - mean[j];
        s[j] = wave_sum(d * d);
    }
    if (lane == 0) { red[1][wave][0] = s[0]; red[1][wave][1] = s[1]; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < WPQ; ++w) t += red[1][w0 + w][j];
        rstd[j] = 1.0f / sqrtf(t * (1.0f / DC) + 1e-5f);
    }
}

typedef float v2f __attribute__((ext_vector_type(2)));

// This thread's weights of output row r for its quarter of the input channels (weights are
// input-major: the lanes of a wavefront read 256 consecutive bytes per channel).  All loads are
// issued before any is used, so the caller can put them in flight early.
__device__ inline void weights_fetch(float (&w)[KQ], const float* __restrict__ wt, int r, int q)
{
    const float* wp = wt + (int64_t)q * KQ * DC + r;
#pragma unroll
    for (int i = 0; i < KQ; ++i) w[i] = wp[i * DC];
}

// part[q][r][0 .. N) = sum over the quarter's channels of w * tile[c][0 .. N)
template <int N>
__device__ inline void quarter_product(const float (&w)[KQ], const float* tile, float* part, int r, int q)
{
    v2f acc[N / 2];
#pragma unroll
    for (int k = 0; k < N / 2; ++k) acc[k] = (v2f){0.0f, 0.0f};
    const float* tq = tile + q * KQ * YP;
#pragma unroll
    for (int i = 0; i < KQ; ++i) {
        const v2f w2 = {w[i], w[i]};
        const float4 y0 = *reinterpret_cast<const float4*>(tq + i * YP);
        const float4 y1 = *reinterpret_cast<const float4*>(tq + i * YP + 4);
        acc[0] += w2 * (v2f){y0.x, y0.y};
        acc[1] += w2 * (v2f){y0.z, y0.w};
        acc[2] += w2 * (v2f){y1.x, y1.y};
        acc[3] += w2 * (v2f){y1.z, y1.w};
        if (N > 8) {
            const float2 y2 = *reinterpret_cast<const float2*>(tq + i * YP + 8);
            acc[4] += w2 * (v2f){y2.x, y2.y};
        }
    }
    float* pp = part + (q * DC + r) * YP;
#pragma unroll
    for (int k = 0; k < N / 2; ++k) *reinterpret_cast<float2*>(pp + 2 * k) = make_float2(acc[k].x, acc[k].y);
}

__device__ inline float part_sum(const float* part, int r, int col)
{
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < NQ; ++k) s += part[(k * DC + r) * YP + col];
    return s;
}

__device__ inline float softplus(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

// Bin edges of one spline axis: softmax-normalised sizes with a floor, cumulated over [-tail, tail].
__device__ inline void spline_knots(const float (&u)[NB], float tail, float floor_, float (&edge)[NB + 1])
{
    float mx = u[0];
#pragma unroll
    for (int k = 1; k < NB; ++k) mx = fmaxf(mx, u[k]);
    float e[NB], sum = 0.0f;
#pragma unroll
    for (int k = 0; k < NB; ++k) { e[k] = expf(u[k] - mx); sum += e[k]; }
    float cs = 0.0f;
    edge[0] = -tail;
#pragma unroll
    for (int k = 0; k < NB - 1; ++k) {
        cs += floor_ + (1.0f - floor_ * NB) * (e[k] / sum);
        edge[k + 1] = 2.0f * tail * cs - tail;
    }
    edge[NB] = tail;
}

// x with spline(x) = y for the monotone rational-quadratic spline whose 3 * NB - 1 parameters
// (widths, heights, inner derivatives) are h[r * stride]; identity outside [-tail, tail].
__device__ inline float inverse_spline(float y, const float* h, int stride, float tail)
{
    const float scale = 1.0f / sqrtf((float)DC);
    float u[NB], xk[NB + 1], yk[NB + 1], d[NB + 1];
#pragma unroll
    for (int k = 0; k < NB; ++k) u[k] = h[k * stride] * scale;
    spline_knots(u, tail, 1e-3f, xk);
#pragma unroll
    for (int k = 0; k < NB; ++k) u[k] = h[(NB + k) * stride] * scale;
    spline_knots(u, tail, 1e-3f, yk);
    const float edge_d = 0.5397424172369522f;            // log(exp(1 - 1e-3) - 1): derivative 1 at the tails
    d[0] = d[NB] = 1e-3f + softplus(edge_d);
#pragma unroll
    for (int k = 1; k < NB; ++k) d[k] = 1e-3f + softplus(h[(2 * NB + k - 1) * stride]);
    const float yc = fminf(fmaxf(y, -tail), tail);
    int idx = -1;
#pragma unroll
    for (int k = 0; k <= NB; ++k) idx += (yc >= (k == NB ? yk[k] + 1e-6f : yk[k])) ? 1 : 0;
    idx = idx < 0 ? 0 : (idx > NB - 1 ? NB - 1 : idx);
    float x0 = 0, bw = 1, y0 = 0, bh = 1, d0 = 1, d1 = 1;
#pragma unroll
    for (int k = 0; k < NB; ++k)
        if (k == idx) { x0 = xk[k]; bw = xk[k + 1] - xk[k]; y0 = yk[k]; bh = yk[k + 1] - yk[k]; d0 = d[k]; d1 = d[k + 1]; }
    const float delta = bh / bw;
    const float dy = yc - y0;
    const float s = d0 + d1 - 2.0f * delta;
    const float qa = dy * s + bh * (delta - d0);
    const float qb = bh * d0 - dy * s;
    const float qc = -delta * dy;
    const float root = (2.0f * qc) / (-qb - sqrtf(qb * qb - 4.0f * qa * qc));
    const bool inside = (y >= -tail) && (y <= tail);
    return inside ? root * bw + x0 : y;
}


constexpr size_t DDS_LDS_FLOATS = DC * XP + DC * YP + NQ * DC * YP + XW + 2 * NWAVE * 2 + 3 * NR * NC + NR * NC;

template <int DIL>
__global__ void __launch_bounds__(NT)
dds_layer_kernel(LayerDev p)
{
    extern __shared__ __align__(16) float lds[];
    float* xw = lds;                                  // window of the layer's input [channel][XP]; later the spline projection
    float* yb = xw + DC * XP;                         // input tile of a 1x1 conv [channel][YP]
    float* part = yb + DC * YP;                       // per-quarter partial sums of a 1x1 conv [quarter][channel][YP]
    float* ms = part + NQ * DC * YP;                  // mask of the window's columns
    float (*red)[NWAVE][2] = reinterpret_cast<float (*)[NWAVE][2]>(ms + XW);
    float* hp = ms + XW + 2 * NWAVE * 2;              // spline parameters: three partial sums, then the sum
    float* hb = hp + 3 * NR * NC;
    const vk_dds_args& a = p.a;
    const int tid = threadIdx.x;
    const int q = tid / DC, r = tid - q * DC;         // quarter, channel
    const int b = blockIdx.y;
    const int t0 = (int)blockIdx.x * NC;
    const int T = a.length;
    constexpr int W = NC + 2 * DIL;                   // window column i is token t0 - DIL + i
    const int j0 = 2 * q;                             // the thread's two columns

    const float* sep_w = p.L;
    const float* sep_b = sep_w + DC * 3;
    const float* g1 = sep_b + DC;
    const float* b1 = g1 + DC;
    const float* pw_wt = b1 + DC;
    const float* pw_b = pw_wt + DC * DC;
    const float* g2 = pw_b + DC;
    const float* b2 = g2 + DC;

    // this thread's weights of the first 1x1 conv: in flight under everything before it
    float wq[KQ];
    weights_fetch(wq, p.first == 2 ? a.pre_wt : pw_wt, r, q);
    constexpr int PJN = (DC * NR + NT - 1) / NT;
    float pjr[PJN];
    if (p.last == 2) {
#pragma unroll
        for (int i = 0; i < PJN; ++i) pjr[i] = (tid + i * NT < DC * NR) ? a.proj_wt[tid + i * NT] : 0.0f;
    }
    if (tid < W) {
        const int t = t0 - DIL + tid;
        ms[tid] = (t >= 0 && t < T) ? a.mask[(int64_t)b * T + t] : 0.0f;
    }
    const float* xb = a.x + (int64_t)b * a.x_bstride;
    constexpr int WN = (DC * W + NT - 1) / NT;        // window elements per thread
    if (p.first == 0) {
        const float* sb = p.src + (int64_t)b * DC * T;
        float tmp[WN];
#pragma unroll
        for (int i = 0; i < WN; ++i) {
            const int e = tid + i * NT, c = e / W, col = e - c * W, t = t0 - DIL + col;
            tmp[i] = (e < DC * W && t >= 0 && t < T) ? sb[(int64_t)c * T + t] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < WN; ++i) {
            const int e = tid + i * NT, c = e / W, col = e - c * W;
            if (e < DC * W) xw[c * XP + col] = tmp[i];
        }
    } else if (p.first == 1) {
        // x = pre(x0) + g, x0 the conditioning channel of z
        const float* gb = a.add_map + (int64_t)b * a.add_map_bstride;
        const float* x0 = xb + (int64_t)a.pass_channel * a.x_cstride;
        float tg[WN], tx[WN];
#pragma unroll
        for (int i = 0; i < WN; ++i) {
            const int e = tid + i * NT, c = e / W, col = e - c * W, t = t0 - DIL + col;
            const bool ok = e < DC * W && t >= 0 && t < T;
            tg[i] = ok ? gb[(int64_t)c * T + t] : 0.0f;
            tx[i] = ok ? x0[t] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < WN; ++i) {
            const int e = tid + i * NT, c = e / W, col = e - c * W, t = t0 - DIL + col;
            if (e < DC * W)
                xw[c * XP + col] = (t >= 0 && t < T) ? fmaf(a.pre_wt[c], tx[i] * a.in_scale, a.pre_b[c]) + tg[i] : 0.0f;
        }
    } else {
        // x = pre(x) + speaker vector on the ten columns of the first layer's window
        float tmp[WN];
#pragma unroll
        for (int i = 0; i < WN; ++i) {
            const int e = tid + i * NT, c = e / W, col = e - c * W, t = t0 - DIL + col;
            tmp[i] = (e < DC * W && t >= 0 && t < T) ? xb[(int64_t)c * a.x_cstride + t] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < WN; ++i) {
            const int e = tid + i * NT, c = e / W, col = e - c * W;
            if (e < DC * W) yb[c * YP + col] = tmp[i];
        }
        __syncthreads();
        quarter_product<NC + 2>(wq, yb, part, r, q);
        weights_fetch(wq, pw_wt, r, q);
        __syncthreads();
        long long add_row = b;                  // row of the speaker vector: the item's own, or its row of a per-speaker table
        if (a.add_index != nullptr) {
            add_row = a.add_index[b];
            add_row = add_row < 0 ? 0 : (add_row >= a.add_rows ? a.add_rows - 1 : add_row);
        }
        const float add = a.pre_b[r] + (a.add_vec ? a.add_vec[add_row * DC + r] : 0.0f);
        for (int col = q; col < NC + 2; col += NQ) xw[r * XP + col] = part_sum(part, r, col) + add;
    }
    __syncthreads();

    // depthwise conv of x * mask (columns beyond the sequence are zero), LayerNorm, GELU
    float v[2], xc[2], mean[2], rstd[2];
    {
        const float w0 = sep_w[r * 3], w1 = sep_w[r * 3 + 1], w2 = sep_w[r * 3 + 2], bs = sep_b[r];
        const float* row = xw + r * XP;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int j = j0 + k;
            xc[k] = row[j + DIL];
            float s = bs;
            s = fmaf(w0, row[j] * ms[j], s);
            s = fmaf(w1, xc[k] * ms[j + DIL], s);
            s = fmaf(w2, row[j + 2 * DIL] * ms[j + 2 * DIL], s);
            v[k] = s;
        }
    }
    column_norm(v, mean, rstd, red);
    {
        const float g = g1[r], bt = b1[r];
        *reinterpret_cast<float2*>(yb + r * YP + j0) =
            make_float2(gelu_erf((v[0] - mean[0]) * rstd[0] * g + bt), gelu_erf((v[1] - mean[1]) * rstd[1] * g + bt));
    }
    __syncthreads();                                  // yb complete; nobody reads the window any more
    if (p.last == 2) {
#pragma unroll
        for (int i = 0; i < PJN; ++i)
            if (tid + i * NT < DC * NR) xw[tid + i * NT] = pjr[i];
    }

    // 1x1 conv, LayerNorm, GELU, residual
    quarter_product<NC>(wq, yb, part, r, q);
    if (p.last == 1) weights_fetch(wq, a.proj_wt, r, q);            // in flight under the second LayerNorm
    __syncthreads();
    {
        const float bias = pw_b[r];
        v[0] = part_sum(part, r, j0) + bias;
        v[1] = part_sum(part, r, j0 + 1) + bias;
    }
    column_norm(v, mean, rstd, red);
    float xn[2];
    {
        const float g = g2[r], bt = b2[r];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            xn[k] = xc[k] + gelu_erf((v[k] - mean[k]) * rstd[k] * g + bt);
            if (p.final_layer) xn[k] *= ms[j0 + k + DIL];
        }
    }

    if (p.last == 0) {
        float* db = p.dst + (int64_t)b * DC * T + (int64_t)r * T;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (t0 + j0 + k < T) db[t0 + j0 + k] = xn[k];
        return;
    }

    // every thread is past its reads of yb and part (column_norm synchronises)
    *reinterpret_cast<float2*>(yb + r * YP + j0) = make_float2(xn[0], xn[1]);
    __syncthreads();
    if (p.last == 1) {
        quarter_product<NC>(wq, yb, part, r, q);
        __syncthreads();
        const float bias = a.proj_b[r];
        float* ob = a.out + (int64_t)b * a.out_bstride + (int64_t)r * a.out_cstride;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (t0 + j0 + k < T) ob[t0 + j0 + k] = (part_sum(part, r, j0 + k) + bias) * ms[j0 + k + DIL];
        return;
    }

    // spline parameters of the eight columns: NR x NC outputs, the input channels in three
    // thirds (696 threads), then one lane per column inverts the spline
    const float* pj = xw;
    if (tid < 3 * NR * NC) {
        const int third = tid / (NR * NC), o = tid - third * (NR * NC), rr = o / NC, j = o - rr * NC;
        float s = 0.0f;
#pragma unroll 8
        for (int c = third * (DC / 3); c < (third + 1) * (DC / 3); ++c) s = fmaf(pj[c * NR + rr], yb[c * YP + j], s);
        hp[tid] = s;
    }
    __syncthreads();
    if (tid < NR * NC) {
        const int rr = tid / NC, j = tid - rr * NC;
        hb[tid] = (hp[tid] + hp[NR * NC + tid] + hp[2 * NR * NC + tid] + a.proj_b[rr]) * ms[j + DIL];
    }
    __syncthreads();
    if (tid < NC && t0 + tid < T) {
        const int t = t0 + tid;
        const float mk = ms[tid + DIL];
        const int other = 1 - a.pass_channel;
        const float x0 = xb[(int64_t)a.pass_channel * a.x_cstride + t] * a.in_scale;
        const float x1 = xb[(int64_t)other * a.x_cstride + t] * a.in_scale;
        float r0 = x0 * mk;
        float r1 = inverse_spline(x1, hb + tid, NC, a.tail_bound) * mk;
        if (a.affine_m) {
            r0 = (r0 - a.affine_m[a.pass_channel]) * expf(-a.affine_logs[a.pass_channel]) * mk;
            r1 = (r1 - a.affine_m[other]) * expf(-a.affine_logs[other]) * mk;
        }
        float* ob = a.out + (int64_t)b * a.out_bstride;
        ob[(int64_t)a.pass_channel * a.out_cstride + t] = r0;
        ob[(int64_t)other * a.out_cstride + t] = r1;
    }
}

// ---- text encoder: LayerNorm over channels, windowed relative-position attention ----------

struct NormDev {
    const float* x; const float* gamma; const float* beta; const float* mask; float* out;
    int length;
};

// out = LayerNorm_channels(x) * mask for [B, 192, T] (dense): a workgroup owns eight columns, a
// thread two columns of one channel.
__global__ void __launch_bounds__(NT)
channel_norm_kernel(NormDev p)
{
    __shared__ float red[2][NWAVE][2];
    const int tid = threadIdx.x, q = tid / DC, r = tid - q * DC;
    const int T = p.length, t = (int)blockIdx.x * NC + 2 * q;
    const int64_t row = ((int64_t)blockIdx.y * DC + r) * T;
    float v[2], mean[2], rstd[2];
    v[0] = t < T ? p.x[row + t] : 0.0f;
    v[1] = t + 1 < T ? p.x[row + t + 1] : 0.0f;
    column_norm(v, mean, rstd, red);
    const float g = p.gamma[r], bt = p.beta[r];
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (t + k < T) {
            const float mk = p.mask ? p.mask[(int64_t)blockIdx.y * T + t + k] : 1.0f;
            p.out[row + t + k] = ((v[k] - mean[k]) * rstd[k] * g + bt) * mk;
        }
}

constexpr int AH = VK_ATT_HEAD_DIM;    // channels per head
constexpr int AQ = 8;                  // queries per workgroup
constexpr int AKB = 32;                // keys per staged block
constexpr int AKP = AKB + 1;           // pitch of a staged key / value block
constexpr int AWIN = VK_ATT_WINDOW;    // relative-position window
constexpr int ANR = 2 * AWIN + 1;
constexpr int ATHREADS = 256;

struct AttDev {
    const float* qkv; const float* emb_k; const float* emb_v; const float* mask; float* out;
    int length, heads, s_pitch;
};

// One head's attention for eight queries: scores against all keys (the keys staged 32 at a
// time), learned relative-position logits inside the window, key/query masking, softmax, then
// the weighted sum of the values plus the relative-position values.
__global__ void __launch_bounds__(ATHREADS)
rel_attention_kernel(AttDev p)
{
    extern __shared__ __align__(16) float lds[];
    float* qs = lds;                       // [AH][AQ] queries, scaled
    float* kt = qs + AH * AQ;              // [AH][AKP] staged keys, then values
    float* ek = kt + AH * AKP;             // [ANR][AH]
    float* ev = ek + ANR * AH;             // [ANR][AH]
    float* rq = ev + ANR * AH;             // [AQ][ANR] relative logits of the queries (padded to 80)
    float* S = rq + 80;                    // [AQ][s_pitch] scores, then probabilities
    const int tid = threadIdx.x;
    const int T = p.length, H = p.heads, C = H * AH;
    const int b = blockIdx.z, h = blockIdx.y, t0 = (int)blockIdx.x * AQ;
    const float* qg = p.qkv + ((int64_t)b * 3 * C + (int64_t)h * AH) * T;
    const float* kg = qg + (int64_t)C * T;
    const float* vg = kg + (int64_t)C * T;
    const float* mb = p.mask + (int64_t)b * T;
    const float scale = 1.0f / sqrtf((float)AH);

    for (int e = tid; e < AH * AQ; e += ATHREADS) {
        const int d = e / AQ, i = e - d * AQ;
        qs[e] = (t0 + i < T) ? qg[(int64_t)d * T + t0 + i] * scale : 0.0f;
    }
    for (int e = tid; e < ANR * AH; e += ATHREADS) { ek[e] = p.emb_k[e]; ev[e] = p.emb_v[e]; }
    __syncthreads();
    if (tid < AQ * ANR) {
        const int i = tid / ANR, rr = tid - i * ANR;
        float s = 0.0f;
#pragma unroll 8
        for (int d = 0; d < AH; ++d) s = fmaf(qs[d * AQ + i], ek[rr * AH + d], s);
        rq[tid] = s;
    }

    // scores: thread (query i, key lane jl)
    const int i = tid >> 5, jl = tid & 31;
    const int ti = t0 + i;
    const float mi = ti < T ? mb[ti] : 0.0f;
    constexpr int LN_ = AH * AKB / ATHREADS;            // staged elements per thread and block
    for (int jb = 0; jb < T; jb += AKB) {
        float tmp[LN_];
#pragma unroll
        for (int n = 0; n < LN_; ++n) {
            const int e = tid + n * ATHREADS, d = e / AKB, j = e - d * AKB;
            tmp[n] = (jb + j < T) ? kg[(int64_t)d * T + jb + j] : 0.0f;
        }
        __syncthreads();                                // the previous block's readers are done (and rq is written)
#pragma unroll
        for (int n = 0; n < LN_; ++n) {
            const int e = tid + n * ATHREADS, d = e / AKB, j = e - d * AKB;
            kt[d * AKP + j] = tmp[n];
        }
        __syncthreads();
        const int j = jb + jl;
        float s = 0.0f;
#pragma unroll 8
        for (int d = 0; d < AH; ++d) s = fmaf(qs[d * AQ + i], kt[d * AKP + jl], s);
        if (j < T) {
            const int rel = j - ti;
            if (rel >= -AWIN && rel <= AWIN) s += rq[i * ANR + rel + AWIN];
            if (mi * mb[j] == 0.0f) s = -1e4f;
            S[i * p.s_pitch + j] = s;
        }
    }
    __syncthreads();

    // softmax over the keys: 32 lanes per query
    {
        float* row = S + i * p.s_pitch;
        float mx = -3.0e38f;
        for (int j = jl; j < T; j += 32) mx = fmaxf(mx, row[j]);
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 32));
        float sum = 0.0f;
        for (int j = jl; j < T; j += 32) { const float e = expf(row[j] - mx); row[j] = e; sum += e; }
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 32);
        const float inv = 1.0f / sum;
        for (int j = jl; j < T; j += 32) row[j] *= inv;
    }
    __syncthreads();

    // values: thread (channel d, four of the queries)
    const int d = tid % AH, half = tid / AH;            // threads 192.. only help staging
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int jb = 0; jb < T; jb += AKB) {
        float tmp[LN_];
#pragma unroll
        for (int n = 0; n < LN_; ++n) {
            const int e = tid + n * ATHREADS, dd = e / AKB, j = e - dd * AKB;
            tmp[n] = (jb + j < T) ? vg[(int64_t)dd * T + jb + j] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int n = 0; n < LN_; ++n) {
            const int e = tid + n * ATHREADS, dd = e / AKB, j = e - dd * AKB;
            kt[dd * AKP + j] = tmp[n];
        }
        __syncthreads();
        if (half < 2) {
            const int jn = min(AKB, T - jb);
            for (int j = 0; j < jn; ++j) {
                const float vv = kt[d * AKP + j];
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = fmaf(S[(4 * half + k) * p.s_pitch + jb + j], vv, acc[k]);
            }
        }
    }
    if (half < 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int tq = t0 + 4 * half + k;
            if (tq >= T) continue;
            float o = acc[k];
#pragma unroll
            for (int rr = 0; rr < ANR; ++rr) {
                const int j = tq + rr - AWIN;
                if (j >= 0 && j < T) o = fmaf(S[(4 * half + k) * p.s_pitch + j], ev[rr * AH + d], o);
            }
            p.out[((int64_t)b * C + (int64_t)h * AH + d) * T + tq] = o;
        }
    }
}

// ---- token level -> frame level: the hard monotonic alignment as a gather ------------------

constexpr int EX_THREADS = 256;
constexpr int EX_CH = 8;                       // channels per workgroup
constexpr int EX_PER = VK_EXPAND_MAX_TOKENS / EX_THREADS;

struct ExpandDev {
    const float* dur; const float* m; const float* logs; float* m_out; float* logs_out; float* frame_mask;
    int channels, tokens, frames;
};

// Frame y belongs to token x iff cum[x - 1] <= y < cum[x] (cum = running sum of the durations).
// Every workgroup scans the durations of its batch item (a few hundred values), each thread
// finds the token of its frame by bisection and copies EX_CH channels of both statistics.
__global__ void __launch_bounds__(EX_THREADS)
expand_prior_kernel(ExpandDev p)
{
    __shared__ float cum[VK_EXPAND_MAX_TOKENS];
    __shared__ float partial[EX_THREADS];
    const int tid = threadIdx.x, b = blockIdx.z, c0 = (int)blockIdx.y * EX_CH;
    const int Tx = p.tokens, Ty = p.frames;
    const int per = (Tx + EX_THREADS - 1) / EX_THREADS;
    float loc[EX_PER], s = 0.0f;
#pragma unroll
    for (int k = 0; k < EX_PER; ++k) {
        const int idx = tid * per + k;
        s += (k < per && idx < Tx) ? p.dur[(int64_t)b * Tx + idx] : 0.0f;
        loc[k] = s;
    }
    partial[tid] = s;
    __syncthreads();
    float base = 0.0f;
    for (int i = 0; i < tid; ++i) base += partial[i];
#pragma unroll
    for (int k = 0; k < EX_PER; ++k) {
        const int idx = tid * per + k;
        if (k < per && idx < Tx) cum[idx] = base + loc[k];
    }
    __syncthreads();
    const float total = cum[Tx - 1];
    const int y = (int)blockIdx.x * EX_THREADS + tid;
    if (y >= Ty) return;
    const float fy = (float)y;
    int lo = 0, hi = Tx;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cum[mid] > fy) hi = mid; else lo = mid + 1;
    }
    const bool valid = fy < total;
    const int x = valid ? lo : 0;
    if (c0 == 0) p.frame_mask[(int64_t)b * Ty + y] = fy < fmaxf(total, 1.0f) ? 1.0f : 0.0f;
    float vm[EX_CH], vl[EX_CH];
#pragma unroll
    for (int c = 0; c < EX_CH; ++c) {
        const bool ok = valid && c0 + c < p.channels;
        const int64_t src = ((int64_t)b * p.channels + c0 + c) * Tx + x;
        vm[c] = ok ? p.m[src] : 0.0f;
        vl[c] = ok ? p.logs[src] : 0.0f;
    }
#pragma unroll
    for (int c = 0; c < EX_CH; ++c)
        if (c0 + c < p.channels) {
            const int64_t dst = ((int64_t)b * p.channels + c0 + c) * Ty + y;
            p.m_out[dst] = vm[c];
            p.logs_out[dst] = vl[c];
        }
}

bool overlap(const void* p, int64_t pn, const void* q, int64_t qn)
{
    const char* a = (const char*)p; const char* b = (const char*)q;
    return a < b + qn && b < a + pn;
}

int64_t dds_workspace_bytes(int batch, int length) { return 2ll * batch * DC * length * (int64_t)sizeof(float); }

int dds_launch(const vk_dds_args* a, void* stream, bool coupling, const char* who)
{
    char msg[160];
    auto fail = [&](int code, const char* what) { snprintf(msg, sizeof msg, "%s: %s", who, what); return vk_fail(code, msg); };
    if (!a) return fail(VK_E_ARG, "null argument struct");
    if (!a->x || !a->pre_wt || !a->pre_b || !a->layers || !a->proj_wt || !a->proj_b || !a->mask || !a->out)
        return fail(VK_E_ARG, "null pointer");
    if (a->batch <= 0 || a->length <= 0) return fail(VK_E_ARG, "non-positive size");
    if (a->batch > 65535) return fail(VK_E_UNSUPPORTED, "batch > 65535");
    if (a->taps != 3 || a->n_layers < 1 || a->n_layers > 3) return fail(VK_E_UNSUPPORTED, "taps must be 3 and n_layers 1..3");
    if (a->x_cstride < a->length || a->out_cstride < a->length) return fail(VK_E_ARG, "channel stride shorter than the sequence");
    const int in_ch = coupling ? 2 : DC, out_ch = coupling ? 2 : DC;
    if (a->x_bstride < (int64_t)in_ch * a->x_cstride && a->batch > 1) return fail(VK_E_ARG, "x batch stride too small");
    if (a->out_bstride < (int64_t)out_ch * a->out_cstride && a->batch > 1) return fail(VK_E_ARG, "out batch stride too small");
    const int64_t xspan = ((int64_t)(a->batch - 1) * a->x_bstride + (int64_t)(in_ch - 1) * a->x_cstride + a->length) * 4;
    const int64_t ospan = ((int64_t)(a->batch - 1) * a->out_bstride + (int64_t)(out_ch - 1) * a->out_cstride + a->length) * 4;
    if (overlap(a->x, xspan, a->out, ospan)) return fail(VK_E_ARG, "out overlaps x");
    const int64_t need = a->n_layers > 1 ? dds_workspace_bytes(a->batch, a->length) : 0;
    if (need && (!a->workspace || a->workspace_bytes < need)) return fail(VK_E_WORKSPACE, "workspace missing or too small");
    if (need && (overlap(a->workspace, need, a->out, ospan) || overlap(a->workspace, need, a->x, xspan)))
        return fail(VK_E_ARG, "workspace overlaps x or out");
    if (coupling) {
        if (!a->add_map) return fail(VK_E_ARG, "coupling needs add_map");
        if (a->add_map_bstride < (int64_t)DC * a->length && a->batch > 1) return fail(VK_E_ARG, "add_map batch stride too small");
        if (a->pass_channel != 0 && a->pass_channel != 1) return fail(VK_E_ARG, "pass_channel must be 0 or 1");
        if (a->num_bins != NB) return fail(VK_E_UNSUPPORTED, "num_bins must be 10");
        if (!(a->tail_bound > 0.0f)) return fail(VK_E_ARG, "tail_bound must be positive");
        if ((a->affine_m == nullptr) != (a->affine_logs == nullptr)) return fail(VK_E_ARG, "affine_m and affine_logs go together");
    }
    LayerDev d;
    d.a = *a;
    const int layer_floats = DC * a->taps + 3 * DC + DC * DC + 3 * DC;
    float* ws[2] = {(float*)a->workspace, (float*)a->workspace + (int64_t)a->batch * DC * a->length};
    dim3 grid((unsigned)((a->length + NC - 1) / NC), (unsigned)a->batch);
    for (int l = 0, dil = 1; l < a->n_layers; ++l, dil *= a->taps) {
        const bool fin = l == a->n_layers - 1;
        d.src = l == 0 ? nullptr : ws[(l - 1) & 1];
        d.dst = fin ? nullptr : ws[l & 1];
        d.L = a->layers + (int64_t)l * layer_floats;
        d.dil = dil;
        d.first = l == 0 ? (coupling ? 1 : 2) : 0;
        d.last = fin ? (coupling ? 2 : 1) : 0;
        d.final_layer = fin;
        const size_t ldsb = DDS_LDS_FLOATS * sizeof(float);
        hipError_t e = hipSuccess;
        if (dil == 1) {
            static std::atomic<unsigned long long> raised{0};
            if ((e = vk_allow_lds(dds_layer_kernel<1>, ldsb, raised)) == hipSuccess)
                hipLaunchKernelGGL(dds_layer_kernel<1>, grid, dim3(NT), ldsb, (hipStream_t)stream, d);
        } else if (dil == 3) {
            static std::atomic<unsigned long long> raised{0};
            if ((e = vk_allow_lds(dds_layer_kernel<3>, ldsb, raised)) == hipSuccess)
                hipLaunchKernelGGL(dds_layer_kernel<3>, grid, dim3(NT), ldsb, (hipStream_t)stream, d);
        } else {
            static std::atomic<unsigned long long> raised{0};
            if ((e = vk_allow_lds(dds_layer_kernel<9>, ldsb, raised)) == hipSuccess)
                hipLaunchKernelGGL(dds_layer_kernel<9>, grid, dim3(NT), ldsb, (hipStream_t)stream, d);
        }
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) return vk_fail(VK_E_LAUNCH, hipGetErrorString(e));
    }
    return VK_OK;
}

}  // namespace

extern "C" int vk_channel_norm(const float* x, const float* gamma, const float* beta, const float* mask, float* out,
                               int batch, int channels, int length, void* stream)
{
    if (!x || !gamma || !beta || !out || batch <= 0 || length <= 0)
        return vk_fail(VK_E_ARG, "vk_channel_norm: null pointer or non-positive size");
    if (channels != DC) return vk_fail(VK_E_UNSUPPORTED, "vk_channel_norm: channels must be 192");
    if (batch > 65535) return vk_fail(VK_E_UNSUPPORTED, "vk_channel_norm: batch > 65535");
    NormDev d{x, gamma, beta, mask, out, length};
    dim3 grid((unsigned)((length + NC - 1) / NC), (unsigned)batch);
    hipLaunchKernelGGL(channel_norm_kernel, grid, dim3(NT), 0, (hipStream_t)stream, d);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vk_fail(VK_E_LAUNCH, hipGetErrorString(e));
    return VK_OK;
}

extern "C" int vk_rel_attention(const float* qkv, const float* emb_k, const float* emb_v, const float* mask, float* out,
                                int batch, int heads, int head_dim, int window, int length, void* stream)
{
    if (!qkv || !emb_k || !emb_v || !mask || !out || batch <= 0 || heads <= 0 || length <= 0)
        return vk_fail(VK_E_ARG, "vk_rel_attention: null pointer or non-positive size");
    if (head_dim != AH || window != AWIN)
        return vk_fail(VK_E_UNSUPPORTED, "vk_rel_attention: head_dim must be 96 and window 4");
    if (length > VK_ATT_MAX_LENGTH) return vk_fail(VK_E_UNSUPPORTED, "vk_rel_attention: sequence longer than VK_ATT_MAX_LENGTH");
    if (batch > 65535 || heads > 65535) return vk_fail(VK_E_UNSUPPORTED, "vk_rel_attention: batch or heads > 65535");
    AttDev d{qkv, emb_k, emb_v, mask, out, length, heads, (length + 3) & ~3};
    const size_t ldsb = (size_t)(AH * AQ + AH * AKP + 2 * ANR * AH + 80 + AQ * d.s_pitch) * sizeof(float);
    static std::atomic<unsigned long long> raised{0};
    hipError_t e = vk_allow_lds(rel_attention_kernel, ldsb, raised);
    if (e != hipSuccess) return vk_fail(VK_E_LAUNCH, hipGetErrorString(e));
    dim3 grid((unsigned)((length + AQ - 1) / AQ), (unsigned)heads, (unsigned)batch);
    hipLaunchKernelGGL(rel_attention_kernel, grid, dim3(ATHREADS), ldsb, (hipStream_t)stream, d);
    e = hipGetLastError();
    if (e != hipSuccess) return vk_fail(VK_E_LAUNCH, hipGetErrorString(e));
    return VK_OK;
}

extern "C" int vk_expand_prior(const float* durations, const float* m, const float* logs, float* m_out, float* logs_out,
                               float* frame_mask, int batch, int channels, int tokens, int frames, void* stream)
{
    if (!durations || !m || !logs || !m_out || !logs_out || !frame_mask || batch <= 0 || channels <= 0 || tokens <= 0 || frames <= 0)
        return vk_fail(VK_E_ARG, "vk_expand_prior: null pointer or non-positive size");
    if (tokens > VK_EXPAND_MAX_TOKENS) return vk_fail(VK_E_UNSUPPORTED, "vk_expand_prior: more than VK_EXPAND_MAX_TOKENS tokens");
    if (batch > 65535) return vk_fail(VK_E_UNSUPPORTED, "vk_expand_prior: batch > 65535");
    ExpandDev d{durations, m, logs, m_out, logs_out, frame_mask, channels, tokens, frames};
    dim3 grid((unsigned)((frames + EX_THREADS - 1) / EX_THREADS), (unsigned)((channels + EX_CH - 1) / EX_CH), (unsigned)batch);
    hipLaunchKernelGGL(expand_prior_kernel, grid, dim3(EX_THREADS), 0, (hipStream_t)stream, d);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vk_fail(VK_E_LAUNCH, hipGetErrorString(e));
    return VK_OK;
}

extern "C" int64_t vk_dds_workspace_bytes(int batch, int length)
{
    if (batch <= 0 || length <= 0) return vk_fail(VK_E_ARG, "vk_dds_workspace_bytes: non-positive size");
    return dds_workspace_bytes(batch, length);
}
extern "C" int vk_dds_condition(const vk_dds_args* a, void* stream) { return dds_launch(a, stream, false, "vk_dds_condition"); }
extern "C" int vk_dds_coupling(const vk_dds_args* a, void* stream)  { return dds_launch(a, stream, true, "vk_dds_coupling"); }
