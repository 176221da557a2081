/*
 * vits_amd.h - C ABI of the MI355X (gfx950) VITS generator-inference kernels.
 *
 * The reference (zassou65535/VITS) has no FFI or operator registry: its hot path is a
 * set of Python nn.Module.forward methods that call PyTorch ATen ops.  Each entry point
 * below therefore names the reference forward (file:line under /root/reference) whose
 * arithmetic it replaces; the Python classes in vits_amd/module mirror those forwards
 * and call these functions through ctypes.
 *
 * Conventions
 *  - every data pointer is a DEVICE pointer owned by the caller (fp32, time-contiguous
 *    [batch, channel, time] views described by element strides);
 *  - functions only enqueue work on `stream` (a hipStream_t passed as void*); they never
 *    allocate, free or synchronise, so they may be captured into a HIP graph;
 *  - return value 0 = success, negative = error (VK_E_*); vk_last_error() returns a
 *    thread-local message for the last failing call on this thread;
 *  - no mutable global state beyond that thread-local message and once-per-device kernel
 *    attributes: calls are re-entrant, and the library reads no environment variable.
 */
#ifndef VITS_AMD_H
#define VITS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VK_ABI_VERSION 10

#define VK_OK            0
#define VK_E_ARG        -1   /* null pointer / non-positive size / inconsistent strides */
#define VK_E_UNSUPPORTED -2  /* shape or option outside what the kernels implement */
#define VK_E_LAUNCH     -3   /* hipLaunchKernel reported an error */
#define VK_E_WORKSPACE  -4   /* workspace too small */

/* ---- weight packing geometry -------------------------------------------------------
 * Convolution weights are consumed in "MFMA fragment order": for every 32-row tile of
 * output channels the K dimension (input-channel chunk, tap, channel pair) is laid out
 * as a linear stream of 64-lane x 4-float groups, so a wavefront fetches its A operands
 * with one coalesced 1 KiB load per 4 k-steps.  vits_amd/packing.py produces the layout;
 * vk_tile_query reports the sizes a packer must honour for a tile family.           */
typedef struct {
    int32_t ck;            /* input channels per LDS chunk (K-chunk)                  */
    int32_t rows_per_block;/* output rows (32 * A_REP * WM) handled by one workgroup  */
    int32_t time_tile;     /* output columns (time steps) handled by one workgroup    */
    int32_t a_rep;         /* consecutive 32-row tiles owned by one wavefront         */
    int32_t ck_bf16;       /* K-chunk of the bf16-input kernels (VK_PREC_BF16)         */
    int32_t ck_split;      /* K-chunk of the sliced-fp32 kernels (VK_PREC_F32_SPLIT)   */
    int32_t max_blocks;    /* vk_mrf_query: resblocks one launch can sum; 0 elsewhere  */
} vk_tile_info;

/* Matrix-core input precision (vk_conv_args.precision).  Activations, bias, residuals and
 * outputs are fp32 in memory either way and accumulation is fp32.
 *   VK_PREC_F32   v_mfma_f32_32x32x2_f32: exact fp32 products (the <= 1e-4 parity path);
 *                 weights packed as fp32 groups [mt][chunk][tap][g][64 lanes][4].
 *   VK_PREC_BF16  v_mfma_f32_32x32x16_bf16: activations (after the leaky-ReLU) and weights are
 *                 rounded to bf16 (nearest even); weights packed as bf16
 *                 [mt][chunk][tap][kstep][64 lanes][8], lane l holding row l & 31, channels
 *                 chunk*ck_bf16 + 16*kstep + 8*(l >> 5) .. +7.                                */
#define VK_PREC_F32        0
#define VK_PREC_BF16       1
/*   VK_PREC_F32_SPLIT  fp32-accurate products on the bf16 matrix cores: every operand is cut
 *                 exactly into three bf16 slices hi + mid + lo (each the rounded remainder) and a product
 *                 is the six slice products hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid
 *                 accumulated in fp32; the dropped terms are < 2^-26 of the product.  Weights
 *                 packed as bf16 [mt][chunk][tap][kstep][slice][64 lanes][8], chunk = 16
 *                 channels (ck_split), lane l holding row l & 31, channels
 *                 chunk*16 + 8*(l >> 5) .. +7 of slice hi / mid / lo.                          */
#define VK_PREC_F32_SPLIT  2

/* Tile families (vk_conv_args.tile). */
#define VK_TILE_M128  0   /* 128 rows x 128 cols, 4 waves split rows        (C_out >= 128) */
#define VK_TILE_M64   1   /*  64 rows x 256 cols, 2x2 waves                 (C_out 64/192) */
#define VK_TILE_M32   2   /*  32 rows x 512 cols, 4 waves split time        (C_out 32/96)  */
#define VK_TILE_PAIR  3   /* 128 rows x 128 cols, row tiles in (a,b) pairs  (gated / split outputs) */
#define VK_TILE_COUNT 4

/* Epilogues (vk_conv_args.epilogue); v = accumulator + bias[row].
 * Cited lines say which reference statement the epilogue evaluates.                   */
#define VK_EPI_STORE      0 /* out = v (* mask)                        decoder.py:52,112 flow.py:54 */
#define VK_EPI_RESID      1 /* out = v + res                           decoder.py:57-58            */
#define VK_EPI_RESID_ACC  2 /* out = accin + (v + res)                 decoder.py:124              */
#define VK_EPI_RESID_AVG  3 /* out = (accin + (v + res)) / navg        decoder.py:124-125          */
#define VK_EPI_GATE       4 /* out = tanh(v_a) * sigmoid(v_b)          wn.py:15-22,73-75           */
#define VK_EPI_WN_MID     5 /* rows<H: out=(res+v)*mask; rows>=H: out2 (+)= v   wn.py:80-83        */
#define VK_EPI_WN_LAST    6 /* out2 = (out2 + v) * mask                wn.py:85-86                 */
#define VK_EPI_COUPLE_REV 7 /* out = (out - v*mask) * mask             flow.py:58,67               */
#define VK_EPI_COUPLE_FWD 8 /* out = v*mask + out*mask                 flow.py:58,63               */
#define VK_EPI_UPSAMPLE   9 /* polyphase ConvTranspose1d scatter store decoder.py:117  (rows: phase*c_out + channel;
                               stride 8: tile = 8 channels x 4 phases, see packing.polyphase_row_order) */
#define VK_EPI_SAMPLE    10 /* (m,logs) pairs: m, logs, z=(m+eps*exp(logs))*mask  posterior_encoder.py:56-59 */
#define VK_EPI_COUNT     11

typedef struct {
    /* input activation view [B, c_in, L] */
    const float* x;       int64_t x_bstride;   int64_t x_cstride;
    /* packed weights (fragment order) and bias indexed by ORIGINAL output row */
    const float* w;
    const float* bias;    int64_t bias_bstride;      /* 0 = one bias vector for all batch items */
    /* primary output view */
    float* out;           int64_t out_bstride; int64_t out_cstride;
    /* residual / auxiliary input view (same indexing as out unless stated) */
    const float* res;     int64_t res_bstride; int64_t res_cstride;
    /* secondary output view (skip accumulator, MRF accumulator, z of VK_EPI_SAMPLE) */
    float* out2;          int64_t out2_bstride; int64_t out2_cstride;
    /* third output view (logs of VK_EPI_SAMPLE) and noise input (eps) */
    float* out3;          const float* aux;
    const float* mask;    /* [B, L] 0/1 floats or NULL */
    int32_t batch, c_in, c_out, length;
    int32_t taps, dilation, pad;  /* x index of tap j for output t: t + j*dilation - pad */
    int32_t tile, epilogue;
    int32_t act_in;       /* 1: leaky-ReLU(in_slope) applied to x while staging (decoder.py:49,54,115) */
    float   in_slope;
    int32_t first;        /* VK_EPI_WN_MID/LAST: 1 = skip accumulator starts at zero */
    int32_t hidden;       /* VK_EPI_WN_MID: rows < hidden are the residual half */
    int32_t up_stride;    /* VK_EPI_UPSAMPLE: stride S (kernel 2S, padding S/2) */
    float   navg;         /* VK_EPI_RESID_AVG divisor */
    int32_t precision;    /* VK_PREC_* (selects the packed weight format too) */
    int32_t narrow;       /* 1: half the family's time tile (twice the workgroups, same packed weights);
                             for launches too small to fill the GPU.  VK_PREC_F32 and VK_PREC_BF16; ignored otherwise */
    /* Scratch for launches too small to fill the GPU (VK_PREC_F32 and VK_PREC_F32_SPLIT): with it, such a launch
     * splits the channel loop over up to 8 workgroups per output tile; each writes its partial
     * accumulators here and the last one to arrive adds them in a fixed order and runs the
     * epilogue.  NULL: no split.  vk_conv1d_workspace_bytes says how much a launch uses.  The
     * first VK_WS_COUNTER_BYTES hold arrival counters: zero before the first use, and every
     * launch leaves them zero again.  Launches that may overlap in time (other streams) need
     * workspaces of their own. */
    void*   workspace;    int64_t workspace_bytes;
    /* Per-speaker bias table (decoder.py:91,112 `cond`; wn.py:41-42,66 `condition_layer` applied to
     * an embedding that is a row of vits_generator.py:88-91's table): with bias_index, batch item b
     * takes bias row bias_index[b] (clamped to [0, bias_rows)) of a [bias_rows, *] table whose row
     * stride is bias_bstride, instead of row b.  NULL: row b. */
    const int64_t* bias_index;  int32_t bias_rows;
    /* x is read as x[b, c, t] * in_mask[b, t] ([B, L] floats; vits_generator.py:186 `z * spec_mask`
     * in front of the decoder's first convolution).  NULL: x as it is.  Not with 16-byte staging
     * of the fp32 kernel's activated inputs (the launch falls back to 4-byte staging). */
    const float* in_mask;
} vk_conv_args;

#define VK_WS_COUNTER_BYTES 65536

int         vk_abi_version(void);
const char* vk_last_error(void);
/* sizeof of a struct as the library was compiled, so that a binding can check its own mirror:
 * which = 0 vk_conv_args, 1 vk_pair_args, 2 vk_wn_args, 3 vk_tile_info, 4 vk_mrf_args,
 * 5 vk_dds_args; 0 for an unknown id. */
size_t      vk_sizeof_args(int which);
int         vk_tile_query(int tile, vk_tile_info* info);

/* Dynamic LDS bytes one workgroup of this launch uses (diagnostic; <= 160 KiB). */
int64_t     vk_conv1d_lds_bytes(const vk_conv_args* a);

/* Dilated Conv1d / polyphase ConvTranspose1d as implicit GEMM on fp32 MFMA
 * (v_mfma_f32_32x32x2_f32) with a fused epilogue.
 * Replaces nn.Conv1d / nn.ConvTranspose1d + surrounding pointwise ops at
 * decoder.py:49-58,112,115-117 ; wn.py:70-85 ; flow.py:54,58-67 ;
 * posterior_encoder.py:52,56-59.                                                     */
int vk_conv1d(const vk_conv_args* a, void* stream);

/* Bytes of `workspace` this launch would use (counters included), 0 if it does not split, or a
 * negative VK_E_* code.  A launch splits when it has fewer than 512 workgroups (VK_PREC_F32_SPLIT: 128)
 * and at least 4 channel chunks; never more than VK_WS_MAX_BYTES. */
#define VK_WS_MAX_BYTES (VK_WS_COUNTER_BYTES + (64 << 20))
int64_t vk_conv1d_workspace_bytes(const vk_conv_args* a);

/* vk_conv1d with the split's scratch in a device array that is part of the library
 * (VK_STATIC_SCRATCH_BYTES per GPU, held until the process ends) instead of a caller's
 * workspace: `workspace` / `workspace_bytes` are ignored and the call never allocates or frees
 * device memory.  No counterpart in the reference; it is for callers that cannot allocate:
 * code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.
 * The array is a fixed VK_STATIC_SCRATCH_BYTES: a launch whose widest split would need more takes
 * the widest split that fits (none at all in the end), so its result can differ from vk_conv1d's
 * with a large workspace in the last bits, never in validity; a launch whose split fits is
 * bit-identical to vk_conv1d's.  All calls on one GPU share that GPU's array: calls that
 * overlap in time on one GPU (different streams) race; calls on one stream do not. */
#define VK_STATIC_SCRATCH_BYTES (VK_WS_COUNTER_BYTES + (32 << 20))
int vk_conv1d_static(const vk_conv_args* a, void* stream);

/* ---- fused residual pair of a HiFi-GAN resblock ------------------------------------
 *   h   = conv1(lrelu(x, slope))            taps k, dilation d, padding (k-1)*d/2
 *   out = conv2(lrelu(h, slope)) + x        taps k, dilation 1, padding (k-1)/2
 * (decoder.py:49-58), optionally folded into the sum over a stage's resblocks
 * (decoder.py:119-125) through VK_EPI_RESID_ACC / VK_EPI_RESID_AVG.  The intermediate h
 * stays in LDS.  Built for the channel widths vk_pair_query accepts; other
 * widths use two vk_conv1d launches.  `out` must not overlap `x` (neighbouring
 * workgroups read halo columns of x); `acc` may be the same view as `out`.            */
typedef struct {
    const float* x;       int64_t x_bstride;   int64_t x_cstride;
    const float* w1;      const float* bias1;   /* packed for vk_pair_query's geometry */
    const float* w2;      const float* bias2;
    float* out;           int64_t out_bstride; int64_t out_cstride;
    const float* acc;     int64_t acc_bstride; int64_t acc_cstride;   /* running sum or NULL */
    int32_t batch, channels, length;
    int32_t taps, dilation;                     /* taps odd, <= 13; (taps-1)*dilation <= 52 */
    int32_t epilogue;                           /* VK_EPI_RESID, VK_EPI_RESID_ACC, VK_EPI_RESID_AVG */
    float   slope, navg;
    int32_t precision;                          /* VK_PREC_F32 / VK_PREC_F32_SPLIT (32/64 channels) or VK_PREC_BF16 (32..256) */
} vk_pair_args;

/* Packing geometry of the fused kernel for this channel width and precision
 * (VK_E_UNSUPPORTED if none): weights are packed with `ck` (fp32), `ck_bf16` (bf16) or
 * `ck_split` (sliced fp32: the three-slice format of the sliced convolution kernel) and
 * `rows_per_block`; time_tile is the number of output columns a workgroup keeps.  In
 * sliced fp32 the result is bit-identical to the two vk_conv1d launches (VK_EPI_STORE, then
 * the residual epilogue) it replaces.                                                  */
int vk_pair_query(int channels, int precision, vk_tile_info* info);
int vk_resblock_pair(const vk_pair_args* a, void* stream);

/* ---- one whole layer of the WN gated stack in one launch ---------------------------
 *   a = in_conv(x) + cond ; acts = tanh(a[:H]) * sigmoid(a[H:]) ; rs = res_skip(acts)
 *   x_out = (x + rs[:H]) * mask ; skip (+)= rs[H:]         (wn.py:68-83)
 *   last layer: out = (skip + rs) * mask                    (wn.py:85-86)
 * acts stays in LDS.  hidden = 192 (vk_wn_query reports the packing geometry: ck and
 * rows_per_block for VK_PREC_F32, ck_bf16 and rows_per_block for VK_PREC_BF16, whose roundings
 * are those of two vk_conv1d launches; other widths use two vk_conv1d launches).
 * x_out must not overlap x.                                                            */
typedef struct {
    const float* x;       int64_t x_bstride;    int64_t x_cstride;     /* layer input [B, H, L] */
    const float* w_in;                                                 /* in-conv, rows in (a_q, b_q) pair order */
    const float* cond;    int64_t cond_bstride;                        /* [B, 2H] conditioning + in-conv bias */
    const float* w_rs;    const float* bias_rs;                        /* 1x1 conv (2H rows, H on the last layer) */
    float* x_out;         int64_t xo_bstride;   int64_t xo_cstride;    /* (x + res) * mask; last layer: the output */
    float* skip;          int64_t skip_bstride; int64_t skip_cstride;  /* running skip sum */
    const float* mask;                                                 /* [B, L] */
    int32_t batch, hidden, length, taps, dilation;
    int32_t first, last;                                               /* first: skip starts at zero */
    const int64_t* cond_index;  int32_t cond_rows;                     /* as vk_conv_args.bias_index: row of a per-speaker table */
    int32_t precision;                                                 /* VK_PREC_F32 or VK_PREC_BF16 (packed weight format too) */
} vk_wn_args;

int vk_wn_query(int hidden, vk_tile_info* info);
int vk_wn_layer(const vk_wn_args* a, void* stream);

/* ---- a whole multi-receptive-field stage in one launch ------------------------------
 *   out = ( acc + sum_r resblock_r(x) ) / navg                     (decoder.py:119-125)
 *   resblock_r: for each pair j: h = conv1_rj(lrelu(x_r)); x_r = conv2_rj(lrelu(h)) + x_r
 *                                                                  (decoder.py:46-61)
 * conv1_rj has taps[r] taps and dilation[r][j], conv2_rj taps[r] taps and dilation 1, both
 * "same" padded.  The residual stream and the sum over the resblocks (kernel sizes) stay in
 * registers, the activated conv inputs in LDS; only x is read and out written.  Built for
 * VK_PREC_BF16 (bf16 matrix-core inputs, fp32 accumulation) and VK_PREC_F32 (the exact fp32
 * matrix instruction; 32 and 64 channels) and the channel widths
 * vk_mrf_query accepts: 32 and 64 channels take up to VK_MRF_MAX_BLOCKS resblocks per launch,
 * 128 and 256 channels one (vk_tile_info.max_blocks; a single resblock is its own mean, so
 * out = resblock(x): the three pairs of a resblock chained without touching HBM).  The
 * receptive half-width of every resblock, sum_j (taps[r]-1)/2 * (dilation[r][j] + 1), must not
 * exceed 64; a workgroup recomputes that many columns on either side of its outputs, so the
 * launch pays off for short kernels.  Weights are packed like vk_resblock_pair's for the same
 * precision (vk_mrf_query reports ck or ck_bf16 and rows_per_block).  `out` must not overlap `x`. */
#define VK_MRF_MAX_BLOCKS 4
#define VK_MRF_MAX_PAIRS  4
typedef struct {
    const float* x;       int64_t x_bstride;   int64_t x_cstride;     /* stage input [B, C, L] */
    float* out;           int64_t out_bstride; int64_t out_cstride;
    int32_t batch, channels, length;
    int32_t n_blocks, n_pairs;
    int32_t taps[VK_MRF_MAX_BLOCKS];                                   /* odd, >= 3 */
    int32_t dilation[VK_MRF_MAX_BLOCKS][VK_MRF_MAX_PAIRS];
    const void*  w1[VK_MRF_MAX_BLOCKS][VK_MRF_MAX_PAIRS];
    const float* bias1[VK_MRF_MAX_BLOCKS][VK_MRF_MAX_PAIRS];
    const void*  w2[VK_MRF_MAX_BLOCKS][VK_MRF_MAX_PAIRS];
    const float* bias2[VK_MRF_MAX_BLOCKS][VK_MRF_MAX_PAIRS];
    float   slope;
    int32_t precision;                                                 /* VK_PREC_BF16 or VK_PREC_F32 */
    /* Sum over resblocks carried from launch to launch (decoder.py:124): the resblocks of a stage
     * may be spread over several launches, each with the halo of its own resblocks only; a later
     * launch adds the earlier ones' sum `acc` (same indexing as out; may be the same view as out;
     * NULL: none) and the last one divides by the stage's resblock count `navg` (1: no division). */
    const float* acc;     int64_t acc_bstride; int64_t acc_cstride;
    float   navg;
} vk_mrf_args;

int vk_mrf_query(int channels, int precision, vk_tile_info* info);
int vk_mrf_stage(const vk_mrf_args* a, void* stream);

/* ---- token-level blocks of the stochastic duration predictor (reverse direction) -----
 * A dilated depth-separable stack (stochastic_duration_predictor.py:285-323 `DDSConv`): per
 * layer i a depthwise `taps`-tap conv of x * mask with dilation taps**i, LayerNorm over
 * channels, GELU, 1x1 conv, LayerNorm, GELU, added onto x; at the end x * mask.  One launch
 * per layer; a workgroup owns VK_DDS_COLUMNS token columns of all VK_DDS_CHANNELS channels, so
 * a sentence of a few dozen tokens still spreads over several CUs, and everything between the
 * layer's input and output stays in registers and LDS.  The 1x1 convs before and after the
 * stack, and for a coupling flow the spline, ride in the first and last layer's launch.
 * fp32 on the vector ALUs.
 *
 *   vk_dds_condition  out = proj( DDSConv( pre(x) + add_vec ) ) * mask
 *                     (stochastic_duration_predictor.py:397-402: `pre`, `cond`, `convs`, `proj`)
 *   vk_dds_coupling   one ConvFlow in reverse (stochastic_duration_predictor.py:230-274) with the
 *                     Flip before it folded into `pass_channel`: x0 = in_scale * x[:, pass_channel],
 *                     h = proj( DDSConv( pre(x0) + add_map ) ) * mask, the other channel goes
 *                     through the inverse rational-quadratic spline with parameters h (:122-201),
 *                     both channels are written to `out` times mask; with `affine_m` the reversed
 *                     ElementwiseAffine (:213-228) is applied to the result as well.
 *
 * 1x1 weights are passed input-major (w_t[i * rows + o] = weight[o, i]) so that the lanes of
 * a wavefront read consecutive output rows.  `layers` holds, per layer and in this order:
 * depthwise weight [C, taps], depthwise bias [C], first LayerNorm gamma [C], beta [C], 1x1
 * weight [C, C] input-major, 1x1 bias [C], second LayerNorm gamma [C], beta [C].
 * `workspace` holds the stream between the layers' launches: vk_dds_workspace_bytes(batch,
 * length) bytes, contents irrelevant before and after the call.  `out` must not overlap `x`. */
#define VK_DDS_CHANNELS 192
#define VK_DDS_COLUMNS  8
typedef struct {
    const float* x;        int64_t x_bstride;   int64_t x_cstride;   /* condition: [B, C, T]; coupling: z [B, 2, T] */
    const float* pre_wt;   /* condition: [C, C] input-major; coupling: [C] (weight of the 1 -> C conv) */
    const float* pre_b;    /* [C] */
    const float* add_vec;  /* condition: [B, C] added after `pre` (speaker conditioning) or NULL */
    const float* add_map;  int64_t add_map_bstride;                  /* coupling: [B, C, T] added after `pre` */
    const float* layers;
    const float* proj_wt;  /* condition: [C, C] input-major; coupling: [C, 3 * num_bins - 1] input-major */
    const float* proj_b;
    const float* mask;     /* [B, T] 0/1 floats */
    float* out;            int64_t out_bstride; int64_t out_cstride; /* condition: [B, C, T]; coupling: [B, 2, T] */
    const float* affine_m; const float* affine_logs;                 /* coupling: [2] each, or both NULL */
    void*   workspace;     int64_t workspace_bytes;
    int32_t batch, length, n_layers, taps;                           /* taps 3, up to 3 layers */
    int32_t pass_channel;  /* coupling: channel of x that conditions (0 or 1); the other one is transformed */
    int32_t num_bins;      /* coupling: 10 */
    float   tail_bound;    /* coupling: identity outside [-tail_bound, tail_bound] */
    float   in_scale;      /* coupling: x is read as in_scale * x (noise scale of the first flow) */
    const int64_t* add_index;  int32_t add_rows;   /* condition: add_vec row of item b is add_index[b] (clamped) of an
                                                      [add_rows, C] per-speaker table; NULL: row b */
} vk_dds_args;

int64_t vk_dds_workspace_bytes(int batch, int length);
int vk_dds_condition(const vk_dds_args* a, void* stream);
int vk_dds_coupling(const vk_dds_args* a, void* stream);

/* ---- token-level blocks of the text encoder ------------------------------------------
 * The encoder's 1x1 and 3-tap convolutions go through vk_conv1d; these two are what is left
 * between them.  Tensors are dense [B, C, T] fp32.
 *
 * vk_channel_norm: out = LayerNorm over the channel axis (eps 1e-5, gamma / beta [C]) times
 * mask [B, T] (NULL: no mask): `norm_layers_1/2` at text_encoder.py:233-243 (the reference
 * transposes to [B, T, C] around nn.LayerNorm).  C = 192.
 *
 * vk_rel_attention: multi-head attention with learned relative-position keys and values inside
 * a window (text_encoder.py:58-99, 101-174): qkv [B, 3 * heads * head_dim, T] holds the
 * projected queries, keys and values (rows h * head_dim .. of each third belong to head h),
 * scores = q k^T / sqrt(head_dim) + q . emb_k[j - i + window] for |j - i| <= window, positions
 * with mask_i * mask_j == 0 set to -1e4, softmax over keys, out = p v + sum_r p[i, i + r -
 * window] emb_v[r]; out [B, heads * head_dim, T].  emb_k, emb_v are [2 * window + 1, head_dim].
 * head_dim = 96, window = 4, T <= VK_ATT_MAX_LENGTH.                                      */
#define VK_ATT_HEAD_DIM   96
#define VK_ATT_WINDOW     4
#define VK_ATT_MAX_LENGTH 1024
int vk_channel_norm(const float* x, const float* gamma, const float* beta, const float* mask, float* out,
                    int batch, int channels, int length, void* stream);
int vk_rel_attention(const float* qkv, const float* emb_k, const float* emb_v, const float* mask, float* out,
                     int batch, int heads, int head_dim, int window, int length, void* stream);

/* Token level -> frame level (vits_generator.py:45-64 `generate_path`, :177-182): frame y of item
 * b takes the statistics of the token x with cum[x - 1] <= y < cum[x], cum the running sum of
 * durations[b, :] (whole numbers, zero for padded tokens); frames at or beyond the total are
 * zero.  m, logs [B, C, tokens] -> m_out, logs_out [B, C, frames] (dense); frame_mask [B, frames]
 * = 1 for y < max(total, 1).  The reference builds the 0/1 alignment matrix and multiplies by
 * it; the gather gives the same values.  tokens <= VK_EXPAND_MAX_TOKENS.                  */
#define VK_EXPAND_MAX_TOKENS 2048
int vk_expand_prior(const float* durations, const float* m, const float* logs, float* m_out, float* logs_out,
                    float* frame_mask, int batch, int channels, int tokens, int frames, void* stream);

/* out[b, o] = bias[o] + sum_i W[o, i] * g[b, i]   (speaker-conditioning 1x1 convs on
 * [B, G, 1] inputs: decoder.py:91,112 `cond`; wn.py:41-42,66 `condition_layer`).
 * W is [n_out, n_in] row-major (weight norm already folded). One wavefront per output. */
int vk_speaker_linear(const float* g, const float* w, const float* bias, float* out,
                      int batch, int n_in, int n_out, void* stream);

/* wav[b, t] = tanh( sum_{c,j} w[c, j] * lrelu(x[b, c, t + j - pad], slope) )
 * (decoder.py:126-129: leaky_relu(0.01) -> conv1d_post (C->1, k=7, no bias) -> tanh). */
int vk_post_conv_tanh(const float* x, const float* w, float* wav, int batch, int channels,
                      int length, int taps, float slope, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VITS_AMD_H */
