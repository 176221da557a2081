// Small kernels beside the MFMA convolution: the speaker-conditioning linear maps
// (a wavefront-reduced GEMV) and the decoder's output convolution (HBM-bound).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vits_amd.h"
#include "api_internal.hpp"

// One wavefront per (batch item, output): lanes stride over the input vector, then a
// butterfly reduction across the 64 lanes.
__global__ void __launch_bounds__(256)
speaker_linear_kernel(const float* __restrict__ g, const float* __restrict__ w,
                      const float* __restrict__ bias, float* __restrict__ out,
                      int batch, int n_in, int n_out)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= (int64_t)batch * n_out) return;
    const int b = (int)(wave / n_out);
    const int o = (int)(wave - (int64_t)b * n_out);
    const float* wr = w + (int64_t)o * n_in;
    const float* gb = g + (int64_t)b * n_in;
    float s = 0.0f;
    for (int i = lane; i < n_in; i += 64) s = fmaf(wr[i], gb[i], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) out[wave] = s + (bias ? bias[o] : 0.0f);
}

extern "C" int vk_speaker_linear(const float* g, const float* w, const float* bias, float* out,
                      int batch, int n_in, int n_out, void* stream)
{
    if (!g || !w || !out || batch <= 0 || n_in <= 0 || n_out <= 0)
        return vk_fail(VK_E_ARG, "vk_speaker_linear: null pointer or non-positive size");
    const int64_t waves = (int64_t)batch * n_out;
    dim3 grid((unsigned)((waves + 3) / 4));
    hipLaunchKernelGGL(speaker_linear_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                       g, w, bias, out, batch, n_in, n_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vk_fail(VK_E_LAUNCH, hipGetErrorString(e));
    return VK_OK;
}

// Output convolution: C channels -> 1, `taps` wide, leaky-ReLU on the input, tanh on the
// output.  A workgroup produces 1024 consecutive samples; channels are staged through LDS
// eight at a time (with the activation applied once per element; the next eight are already
// being loaded) and each thread keeps four outputs in registers.
constexpr int POST_TILE = 1024;
constexpr int POST_CH   = 8;
constexpr int POST_MAXTAPS = 16;

__global__ void __launch_bounds__(256)
post_conv_tanh_kernel(const float* __restrict__ x, const float* __restrict__ w,
                      float* __restrict__ wav, int channels, int length, int taps, float slope)
{
    __shared__ float tile[POST_CH][POST_TILE + POST_MAXTAPS];
    __shared__ float wl[POST_CH][POST_MAXTAPS];
    constexpr int NCOL = (POST_TILE + POST_MAXTAPS + 255) / 256;     // staged columns per thread per row
    const int b  = blockIdx.y;
    const int t0 = blockIdx.x * POST_TILE;
    const int pad = (taps - 1) / 2;
    const int width = POST_TILE + taps - 1;
    const float* xb = x + (int64_t)b * channels * length;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    // the next chunk's loads are in flight while the current chunk is consumed
    float st[POST_CH][NCOL];
    auto issue = [&](int c0) {
#pragma unroll
        for (int c = 0; c < POST_CH; ++c) {
            const bool cok = (c0 + c) < channels;
            const float* row = xb + (int64_t)(cok ? c0 + c : 0) * length;
#pragma unroll
            for (int i = 0; i < NCOL; ++i) {
                const int col = threadIdx.x + 256 * i;
                const int t = t0 - pad + col;
                st[c][i] = (cok && col < width && t >= 0 && t < length) ? row[t] : 0.0f;
            }
        }
    };
    issue(0);
    for (int c0 = 0; c0 < channels; c0 += POST_CH) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < POST_CH; ++c)
#pragma unroll
            for (int i = 0; i < NCOL; ++i) {
                const int col = threadIdx.x + 256 * i;
                const float v = st[c][i];
                if (col < POST_TILE + POST_MAXTAPS) tile[c][col] = v > 0.0f ? v : v * slope;
            }
        if (threadIdx.x < POST_CH * POST_MAXTAPS) {
            const int c = threadIdx.x / POST_MAXTAPS, j = threadIdx.x % POST_MAXTAPS;
            wl[c][j] = ((c0 + c) < channels && j < taps) ? w[(c0 + c) * taps + j] : 0.0f;
        }
        __syncthreads();
        if (c0 + POST_CH < channels) issue(c0 + POST_CH);
        for (int c = 0; c < POST_CH; ++c) {
            for (int j = 0; j < taps; ++j) {
                const float wv = wl[c][j];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[q] = fmaf(wv, tile[c][threadIdx.x + 256 * q + j], acc[q]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int t = t0 + threadIdx.x + 256 * q;
        if (t < length) wav[(int64_t)b * length + t] = tanhf(acc[q]);
    }
}

extern "C" int vk_post_conv_tanh(const float* x, const float* w, float* wav, int batch, int channels,
                      int length, int taps, float slope, void* stream)
{
    if (!x || !w || !wav || batch <= 0 || channels <= 0 || length <= 0)
        return vk_fail(VK_E_ARG, "vk_post_conv_tanh: null pointer or non-positive size");
    if (taps <= 0 || taps > POST_MAXTAPS || (taps & 1) == 0)
        return vk_fail(VK_E_UNSUPPORTED, "vk_post_conv_tanh: taps must be odd and <= 16");
    dim3 grid((unsigned)((length + POST_TILE - 1) / POST_TILE), (unsigned)batch);
    hipLaunchKernelGGL(post_conv_tanh_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                       x, w, wav, channels, length, taps, slope);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vk_fail(VK_E_LAUNCH, hipGetErrorString(e));
    return VK_OK;
}
