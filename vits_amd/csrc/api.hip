// Host side of the C ABI: argument validation, tile geometry, launches.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "conv_launch.hpp"
#include "resblock_pair.hpp"
#include "conv_bf16.hpp"
#include "resblock_pair_bf16.hpp"
#include "conv_split.hpp"
#include "wn_layer.hpp"
#include "mrf_stage.hpp"
#include "mrf_stage_f32.hpp"
#include "wn_layer_bf16.hpp"
#include "resblock_pair_split.hpp"
#include "api_internal.hpp"

static thread_local char g_err[512] = "";

int vk_fail(int code, const char* msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "");
    return code;
}

// The diagnostic build (-DVK_DIAG, libvits_amd_diag.so, loaded only by tools/) adds switches for
// timing experiments: environment variables that drop parts of a kernel or change the launch
// geometry, and per-workgroup time stamps.  The product library has none of them: it reads no
// environment variable and exports nothing the header does not declare.
#ifdef VK_DIAG
static unsigned long long* g_stamps = nullptr;
extern "C" void vk_debug_set_stamps(void* dev_buffer) { g_stamps = (unsigned long long*)dev_buffer; }
static int diag_env(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#else
static constexpr unsigned long long* g_stamps = nullptr;
static int diag_env(const char*, int dflt) { return dflt; }
#endif

extern "C" int vk_abi_version(void) { return VK_ABI_VERSION; }
extern "C" const char* vk_last_error(void) { return g_err; }
extern "C" size_t vk_sizeof_args(int which)
{
    switch (which) {
    case 0: return sizeof(vk_conv_args);
    case 1: return sizeof(vk_pair_args);
    case 2: return sizeof(vk_wn_args);
    case 3: return sizeof(vk_tile_info);
    case 4: return sizeof(vk_mrf_args);
    case 5: return sizeof(vk_dds_args);
    default: return 0;
    }
}

extern "C" int vk_tile_query(int tile, vk_tile_info* info)
{
    if (tile < 0 || tile >= VK_TILE_COUNT || !info) return vk_fail(VK_E_ARG, "vk_tile_query: bad tile id");
    const TileGeom& G = kTiles[tile];
    info->ck = G.ck;
    info->rows_per_block = 32 * G.a_rep * G.wm;
    info->time_tile = 32 * G.b_rep * G.wn;
    info->a_rep = G.a_rep;
    info->ck_bf16 = kBf16Ck[tile];
    info->ck_split = VK_SPLIT_CK;
    info->max_blocks = 0;
    return VK_OK;
}

// Rows of the implicit GEMM (before padding to the tile's row block).
static int64_t gemm_rows(const vk_conv_args* a)
{
    if (a->epilogue == VK_EPI_UPSAMPLE) return (int64_t)a->c_out * a->up_stride;
    if (a->epilogue == VK_EPI_GATE || a->epilogue == VK_EPI_SAMPLE) return 2 * (int64_t)a->hidden;
    return a->c_out;
}

// `ws` / `ws_bytes`: the scratch a split channel loop may use (null: the launch does not split);
// `query`: only work out how much it would use (*ws_need), whatever scratch there is.
static int derive(const vk_conv_args* a, ConvDev* d, dim3* grid, size_t* lds, int* vec,
                  void* ws, int64_t ws_bytes, int64_t* ws_need, bool query, const char* who = "vk_conv1d",
                  bool fit_split = false)
{
    if (!a) return vk_fail(VK_E_ARG, "vk_conv1d: null argument block");
    if (!a->x || !a->w || !a->out) return vk_fail(VK_E_ARG, "vk_conv1d: null x / w / out pointer");
    if (a->batch <= 0 || a->c_in <= 0 || a->c_out <= 0 || a->length <= 0 || a->taps <= 0 || a->dilation <= 0)
        return vk_fail(VK_E_ARG, "vk_conv1d: non-positive size");
    if (a->tile < 0 || a->tile >= VK_TILE_COUNT) return vk_fail(VK_E_ARG, "vk_conv1d: bad tile id");
    if (a->epilogue < 0 || a->epilogue >= VK_EPI_COUNT) return vk_fail(VK_E_ARG, "vk_conv1d: bad epilogue id");
    if (a->batch > 65535) return vk_fail(VK_E_UNSUPPORTED, "vk_conv1d: batch > 65535");
    if (a->precision < VK_PREC_F32 || a->precision > VK_PREC_F32_SPLIT) return vk_fail(VK_E_ARG, "vk_conv1d: bad precision id");
    const bool split = a->precision == VK_PREC_F32_SPLIT;
    const bool bf16 = a->precision == VK_PREC_BF16 || split;      // bf16 matrix instructions, channel-minor LDS tile
    const bool narrow = a->narrow && !split;
    const TileGeom& G = split ? kSplitTiles[a->tile] : (narrow ? kTilesNarrow[a->tile] : kTiles[a->tile]);
    const int TN = 32 * G.b_rep * G.wn;
    const bool ups = a->epilogue == VK_EPI_UPSAMPLE;
    const int halo = (a->taps - 1) * a->dilation + (ups ? 1 : 0);
    if (halo > 64) return vk_fail(VK_E_UNSUPPORTED, "vk_conv1d: (taps-1)*dilation exceeds the 64-column halo of the LDS tile");
    if (a->pad < 0 || a->pad > halo) return vk_fail(VK_E_ARG, "vk_conv1d: pad outside [0, (taps-1)*dilation]");
    switch (a->epilogue) {
    case VK_EPI_RESID:
        if (!a->res) return vk_fail(VK_E_ARG, "vk_conv1d: residual epilogue without res"); break;
    case VK_EPI_RESID_ACC: case VK_EPI_RESID_AVG:
        if (!a->res || !a->out2) return vk_fail(VK_E_ARG, "vk_conv1d: accumulate epilogue needs res and out2"); break;
    case VK_EPI_GATE:
        if (a->hidden <= 0 || a->hidden % 32) return vk_fail(VK_E_UNSUPPORTED, "vk_conv1d: gate needs hidden % 32 == 0"); break;
    case VK_EPI_SAMPLE:
        if (a->hidden <= 0 || a->hidden % 32 || !a->out2 || !a->out3 || !a->aux)
            return vk_fail(VK_E_ARG, "vk_conv1d: sample epilogue needs hidden % 32 == 0, out2, out3, aux"); break;
    case VK_EPI_WN_MID:
        if (!a->res || !a->out2 || a->hidden <= 0 || a->hidden % 32)
            return vk_fail(VK_E_ARG, "vk_conv1d: WN epilogue needs res, out2 and hidden % 32 == 0"); break;
    case VK_EPI_WN_LAST:
        if (!a->out2 && !a->first) return vk_fail(VK_E_ARG, "vk_conv1d: WN last epilogue needs out2"); break;
    case VK_EPI_COUPLE_REV: case VK_EPI_COUPLE_FWD:
        if (!a->res || !a->mask) return vk_fail(VK_E_ARG, "vk_conv1d: coupling epilogue needs res and mask"); break;
    case VK_EPI_UPSAMPLE:
        if (a->up_stride < 2 || (a->up_stride & 1) || a->taps != 2 || a->dilation != 1 || a->pad != 1)
            return vk_fail(VK_E_UNSUPPORTED, "vk_conv1d: upsample needs even stride, 2 polyphase taps, pad 1");
        if (a->up_stride == 8) {        // grouped row layout, 16-byte stores
            if (a->c_out % 8 || ((uintptr_t)a->out % 16) || (a->out_bstride % 4) || (a->out_cstride % 4))
                return vk_fail(VK_E_UNSUPPORTED, "vk_conv1d: stride-8 upsample needs c_out % 8 == 0 and a 16-byte aligned output view");
        } else if (a->c_out % (32 * G.a_rep))
            return vk_fail(VK_E_UNSUPPORTED, "vk_conv1d: upsample needs c_out to be a multiple of the wave's row span");
        break;
    default: break;
    }
    if (a->act_in && !(a->in_slope >= 0.0f && a->in_slope <= 1.0f))
        return vk_fail(VK_E_UNSUPPORTED, "vk_conv1d: leaky-ReLU slope must lie in [0, 1]");
    // Buffer addressing: offsets within one batch item's view are 32-bit byte offsets.
    auto span = [&](int64_t rows, int64_t cstride, int64_t len) -> int64_t {
        return ((rows - 1) * (cstride < 0 ? -cstride : cstride) + len) * 4;
    };
    const int64_t out_len = ups ? (int64_t)a->length * a->up_stride : a->length;
    const int64_t out_rows = (a->epilogue == VK_EPI_GATE || a->epilogue == VK_EPI_SAMPLE) ? a->hidden
                           : (a->epilogue == VK_EPI_WN_MID ? a->hidden : a->c_out);
    const int64_t xb = span(a->c_in, a->x_cstride, a->length);
    const int64_t ob = span(out_rows, a->out_cstride, out_len);
    const int64_t rb = a->res ? span(out_rows, a->res_cstride, a->length) : 0;
    const int64_t o2 = a->out2 ? span(a->epilogue == VK_EPI_WN_MID ? a->c_out - a->hidden : out_rows, a->out2_cstride, a->length) : 0;
    if (a->x_cstride < 0 || a->out_cstride < 0 || a->res_cstride < 0 || a->out2_cstride < 0)
        return vk_fail(VK_E_ARG, "vk_conv1d: negative channel stride");
    if (xb > 0x7fffffffLL || ob > 0x7fffffffLL || rb > 0x7fffffffLL || o2 > 0x7fffffffLL)
        return vk_fail(VK_E_UNSUPPORTED, "vk_conv1d: one batch item's view spans more than 2 GiB");
    memset(d, 0, sizeof(*d));
    d->a = *a;
    d->x_bytes = (unsigned)xb; d->out_bytes = (unsigned)ob; d->res_bytes = (unsigned)rb; d->out2_bytes = (unsigned)o2;
    const int ck = split ? VK_SPLIT_CK : (bf16 ? kBf16Ck[a->tile] : G.ck);
    d->nchunk = (a->c_in + ck - 1) / ck;
    // 16-byte staging when every row of x starts 16-byte aligned and the length is a multiple of 4
    static const int no_vec = diag_env("VK_NO_VEC", 0);
    if (a->bias_index && (!a->bias || a->bias_rows <= 0)) return vk_fail(VK_E_ARG, "vk_conv1d: bias_index without a bias table");
    *vec = !no_vec && !bf16 && a->act_in && !a->in_mask &&
           ((uintptr_t)a->x % 16 == 0) && (a->x_bstride % 4 == 0) && (a->x_cstride % 4 == 0) && (a->length % 4 == 0) &&
           (a->epilogue == VK_EPI_STORE || a->epilogue == VK_EPI_RESID || a->epilogue == VK_EPI_RESID_ACC ||
            a->epilogue == VK_EPI_RESID_AVG || a->epilogue == VK_EPI_UPSAMPLE) && a->tile != VK_TILE_PAIR;
    if (*vec) {
        const int padv = (a->pad + 3) & ~3;
        d->col0 = padv - a->pad;
        d->xw = (TN + halo + d->col0 + 3) & ~3;
    } else {
        d->col0 = 0;
        d->xw = TN + halo;
    }
    d->lds_stride = TN + 64;              // compile-time constant of the kernel
    if (*vec && d->xw > d->lds_stride) {  // the 4-column alignment slack does not fit: 4-byte staging does
        *vec = 0;
        d->col0 = 0;
        d->xw = TN + halo;
    }
    if (d->xw > d->lds_stride)
        return vk_fail(VK_E_UNSUPPORTED, "vk_conv1d: time tile + halo does not fit the staged LDS row");
    d->groups_per_mt = d->nchunk * a->taps * (split ? 3 * (ck / 16) : (bf16 ? ck / 16 : ck / 8));
    static const int dbg = diag_env("VK_DEBUG_SKIP", 0);
    d->dbg = dbg;
    d->stamps = g_stamps;
    const int rows_blk = 32 * G.a_rep * G.wm;
    const int64_t rows = gemm_rows(a);
    d->n_tiles = (a->length + TN - 1) / TN;
    d->n_rowblk = (int)((rows + rows_blk - 1) / rows_blk);
    if (halo > 0 || d->n_rowblk > 1) {
        // other workgroups read these columns of x (halo) or the same tile (further row blocks)
        // at any time during the launch, so no output may land on x
        auto whole = [&](const void* p, int64_t bstride, int64_t bytes, const char*& lo, const char*& hi) {
            lo = (const char*)p;
            hi = lo + (int64_t)(a->batch - 1) * (bstride < 0 ? -bstride : bstride) * 4 + bytes;
        };
        const char *x0, *x1, *o0, *o1;
        whole(a->x, a->x_bstride, xb, x0, x1);
        whole(a->out, a->out_bstride, ob, o0, o1);
        if (o0 < x1 && x0 < o1) return vk_fail(VK_E_ARG, "vk_conv1d: out overlaps x (taps > 1 or several row blocks)");
        if (a->out2) {
            whole(a->out2, a->out2_bstride, o2, o0, o1);
            if (o0 < x1 && x0 < o1) return vk_fail(VK_E_ARG, "vk_conv1d: out2 overlaps x (taps > 1 or several row blocks)");
        }
    }
    // Placing the row blocks of a time tile on one XCD cuts the fetched bytes of the multi-row-block
    // launches by 2-4x (one HBM read of the x tile per XCD instead of one per row block).  It pays where
    // the kernel is HBM-bound (bf16 inputs, +2 %); the fp32 kernels are MFMA-bound and their time does
    // not depend on it (256 channels: -0.6 % at 7 taps, 0 at 11, +6 % for the short 3-tap residual
    // launches, 0.1 % of a decoder step in all), so it is on there too and the wasted re-reads go.
    // The sliced-fp32 kernel walks its tiles in an order of its own.
    // (Diagnostic build: VK_XCD_MAP=0 / 1 forces it off / on for every precision.)
    static const int xcd_env = diag_env("VK_XCD_MAP", -1);
    d->xcd_map = d->n_rowblk > 1 && (xcd_env >= 0 ? xcd_env != 0 : !split) &&
                 ((int64_t)d->n_tiles * a->batch + 7) / 8 * 8 * d->n_rowblk <= 0x7fffffffLL;
    // A launch too small for 256 CUs splits its channel loop over several workgroups per output
    // tile (fp32 and sliced-fp32 kernels): as many as bring it to about four workgroups per CU, at least two
    // chunks each, at most 8 (one utterance of 3 s, decoder: 2.07 ms at two per CU, 1.96 ms at four).
    // (Diagnostic build: VK_SPLITK_BELOW / VK_SPLITK_TARGET move the two thresholds.)  The partial accumulators and arrival counters live in scratch memory.
    d->ksplit = 1;
    d->ws = nullptr;
    *ws_need = 0;
    const int64_t wgs = (int64_t)d->n_tiles * d->n_rowblk * a->batch;
    static const int sk_below = diag_env("VK_SPLITK_BELOW", 512), sk_target = diag_env("VK_SPLITK_TARGET", 1024);
    // (the sliced-fp32 kernel has thresholds of its own: VK_SPLITK_BELOW_S / VK_SPLITK_TARGET_S)
    static const int sks_below = diag_env("VK_SPLITK_BELOW_S", 128), sks_target = diag_env("VK_SPLITK_TARGET_S", 256);
    if ((!bf16 || split) && wgs < (split ? sks_below : sk_below) && d->nchunk >= 4 && (query || ws != nullptr)) {
        int k = (int)((split ? sks_target : sk_target) / wgs);
        if (k > d->nchunk / 2) k = d->nchunk / 2;
        if (k > 8) k = 8;
        if (split)                                   // the sliced-fp32 kernel's pipeline wants equal shares
            while (k > 1 && d->nchunk % k) --k;
        auto need = [&](int kk) { return VK_WS_COUNTER_BYTES + wgs * kk * (int64_t)rows_blk * TN * 4; };
        if (k > 1) *ws_need = need(k);
        if (k > 1 && !query) {
            // less scratch than the widest split wants.  A caller's workspace: an error the caller
            // can act on.  The library's own array (`fit_split`): a narrower split (for the sliced
            // kernel the next divisor of the chunk count), down to none at all - the result is the
            // same to rounding either way, and a caller that cannot allocate has no other way out
            if (need(k) > ws_bytes && !fit_split) {
                snprintf(g_err, sizeof(g_err), "%s: the split channel loop of this launch needs %lld bytes of "
                         "scratch, %lld are there", who, (long long)*ws_need, (long long)ws_bytes);
                return VK_E_UNSUPPORTED;
            }
            while (k > 1 && (need(k) > ws_bytes || (split && d->nchunk % k))) --k;
            if (k > 1) {
                if ((uintptr_t)ws % 16) return vk_fail(VK_E_ARG, "vk_conv1d: workspace must be 16-byte aligned");
                if (wgs * (int64_t)sizeof(unsigned) > VK_WS_COUNTER_BYTES) k = 1;      // (cannot happen below 512 tiles)
            }
            if (k > 1) {
                d->ksplit = k;
                d->ws = static_cast<float*>(ws);
            }
        }
    }
    if (d->ksplit > 1) {
        d->xcd_map = 0;
        *grid = dim3((unsigned)d->n_tiles, (unsigned)(d->n_rowblk * d->ksplit), (unsigned)a->batch);
    } else if (d->xcd_map)
        *grid = dim3((unsigned)(8 * (((int64_t)d->n_tiles * a->batch + 7) / 8) * d->n_rowblk), 1, 1);
    else
        *grid = dim3((unsigned)d->n_tiles, (unsigned)d->n_rowblk, (unsigned)a->batch);
    *lds = split ? vk_split_lds_bytes(TN, ck)
         : (bf16 ? (size_t)2 * (TN + 64) * (ck * 2 + 16) : (size_t)2 * G.ck * d->lds_stride * sizeof(float));
    if (split && (int64_t)d->n_tiles * d->n_rowblk * a->batch > 0x7fffffffLL)
        return vk_fail(VK_E_UNSUPPORTED, "vk_conv1d: more than 2^31 tiles in one launch");
    static const int lds_min = diag_env("VK_LDS_MIN", 0);   // occupancy experiments
    if ((int)*lds < lds_min) *lds = lds_min;
    if (*lds > 160 * 1024) return vk_fail(VK_E_UNSUPPORTED, "vk_conv1d: LDS tile exceeds 160 KiB");
    return VK_OK;
}

extern "C" int64_t vk_conv1d_lds_bytes(const vk_conv_args* a)
{
    ConvDev d; dim3 grid; size_t lds = 0; int vec = 0; int64_t need = 0;
    int rc = derive(a, &d, &grid, &lds, &vec, nullptr, 0, &need, true);
    return rc == VK_OK ? (int64_t)lds : (int64_t)rc;
}

extern "C" int64_t vk_conv1d_workspace_bytes(const vk_conv_args* a)
{
    ConvDev d; dim3 grid; size_t lds = 0; int vec = 0; int64_t need = 0;
    int rc = derive(a, &d, &grid, &lds, &vec, nullptr, 0, &need, true);
    return rc == VK_OK ? need : (int64_t)rc;
}

static int conv1d_launch(const vk_conv_args* a, const ConvDev& d, dim3 grid, size_t lds, int vec, void* stream);

extern "C" int vk_conv1d(const vk_conv_args* a, void* stream)
{
    ConvDev d; dim3 grid; size_t lds = 0; int vec = 0; int64_t need = 0;
    int rc = derive(a, &d, &grid, &lds, &vec, a ? a->workspace : nullptr, a ? a->workspace_bytes : 0, &need, false);
    if (rc != VK_OK) return rc;
    return conv1d_launch(a, d, grid, lds, vec, stream);
}

// ---- vk_conv1d_static: the split's scratch is a device array of the library ----
// Zero when the code object is loaded; each GPU holds its own copy at its own address.  In host
// code the array's name is only a handle for hipGetSymbolAddress: kernels, memsets and copies
// take the address that call returns for the current device.
static __device__ __align__(256) unsigned char scratch[VK_STATIC_SCRATCH_BYTES];
static std::atomic<void*> g_scratch_addr[64];          // by device ordinal; null until first used there

static int static_scratch(void** out)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return vk_fail(VK_E_LAUNCH, hipGetErrorString(e));
    if (dev < 0 || dev >= 64) return vk_fail(VK_E_UNSUPPORTED, "vk_conv1d_static: device ordinal beyond 63");
    void* p = g_scratch_addr[dev].load(std::memory_order_acquire);
    if (p == nullptr) {
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));     // this device's copy (the current device)
        if (e != hipSuccess || p == nullptr)
            return vk_fail(VK_E_LAUNCH, e != hipSuccess ? hipGetErrorString(e) : "vk_conv1d_static: no address for the scratch array");
        g_scratch_addr[dev].store(p, std::memory_order_release);
    }
    *out = p;
    return VK_OK;
}

extern "C" int vk_conv1d_static(const vk_conv_args* a, void* stream)
{
    void* sp = nullptr;
    int rc = static_scratch(&sp);
    if (rc != VK_OK) return rc;
    ConvDev d; dim3 grid; size_t lds = 0; int vec = 0; int64_t need = 0;
    rc = derive(a, &d, &grid, &lds, &vec, sp, VK_STATIC_SCRATCH_BYTES, &need, false, "vk_conv1d_static", true);
    if (rc != VK_OK) return rc;            // (a split wider than the array holds has been narrowed to fit)
    if (d.ksplit > 1) {
        // the arrival counters of this launch's output tiles start at zero whatever ran before
        const size_t tiles = (size_t)d.n_tiles * d.n_rowblk * a->batch;
        hipError_t e = hipMemsetAsync(sp, 0, tiles * sizeof(unsigned), (hipStream_t)stream);
        if (e != hipSuccess) return vk_fail(VK_E_LAUNCH, hipGetErrorString(e));
    }
    return conv1d_launch(a, d, grid, lds, vec, stream);
}

static int conv1d_launch(const vk_conv_args* a, const ConvDev& d, dim3 grid, size_t lds, int vec, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    const int act = a->act_in ? 1 : 0;
    int e = -1;
    if (a->precision == VK_PREC_F32_SPLIT) {
        switch (a->tile) {
        case VK_TILE_M128: e = vk_launch_split_tile0(a->epilogue, act, d, grid, lds, s); break;
        case VK_TILE_M64:  e = vk_launch_split_tile1(a->epilogue, act, d, grid, lds, s); break;
        case VK_TILE_M32:  e = vk_launch_split_tile2(a->epilogue, act, d, grid, lds, s); break;
        case VK_TILE_PAIR: e = vk_launch_split_tile3(a->epilogue, act, d, grid, lds, s); break;
        }
    } else if (a->precision == VK_PREC_BF16 && a->narrow) {
        switch (a->tile) {
        case VK_TILE_M128: e = vk_launch_bf16_tile0n(a->epilogue, act, d, grid, lds, s); break;
        case VK_TILE_M64:  e = vk_launch_bf16_tile1n(a->epilogue, act, d, grid, lds, s); break;
        case VK_TILE_M32:  e = vk_launch_bf16_tile2n(a->epilogue, act, d, grid, lds, s); break;
        case VK_TILE_PAIR: e = vk_launch_bf16_tile3n(a->epilogue, act, d, grid, lds, s); break;
        }
    } else if (a->precision == VK_PREC_BF16) {
        switch (a->tile) {
        case VK_TILE_M128: e = vk_launch_bf16_tile0(a->epilogue, act, d, grid, lds, s); break;
        case VK_TILE_M64:  e = vk_launch_bf16_tile1(a->epilogue, act, d, grid, lds, s); break;
        case VK_TILE_M32:  e = vk_launch_bf16_tile2(a->epilogue, act, d, grid, lds, s); break;
        case VK_TILE_PAIR: e = vk_launch_bf16_tile3(a->epilogue, act, d, grid, lds, s); break;
        }
    } else if (a->narrow) {
        switch (a->tile) {
        case VK_TILE_M128: e = vk_launch_tile0n(a->epilogue, act, vec, d, grid, lds, s); break;
        case VK_TILE_M64:  e = vk_launch_tile1n(a->epilogue, act, vec, d, grid, lds, s); break;
        case VK_TILE_M32:  e = vk_launch_tile2n(a->epilogue, act, vec, d, grid, lds, s); break;
        case VK_TILE_PAIR: e = vk_launch_tile3n(a->epilogue, act, vec, d, grid, lds, s); break;
        }
    } else
    switch (a->tile) {
    case VK_TILE_M128: e = vk_launch_tile0(a->epilogue, act, vec, d, grid, lds, s); break;
    case VK_TILE_M64:  e = vk_launch_tile1(a->epilogue, act, vec, d, grid, lds, s); break;
    case VK_TILE_M32:  e = vk_launch_tile2(a->epilogue, act, vec, d, grid, lds, s); break;
    case VK_TILE_PAIR: e = vk_launch_tile3(a->epilogue, act, vec, d, grid, lds, s); break;
    }
    if (e == -1) return vk_fail(VK_E_UNSUPPORTED, "vk_conv1d: this (tile, epilogue, act_in) combination is not built");
    if (e != 0) return vk_fail(VK_E_LAUNCH, hipGetErrorString((hipError_t)e));
    return VK_OK;
}

static const PairGeom* pair_geom(int channels)
{
    if (channels == 32) return &kPair32;
    if (channels == 64) return &kPair64;
    return nullptr;
}

static const PairGeomSplit* pair_geom_split(int channels)
{
    if (channels == 32) return &kPairSplit32;
    if (channels == 64) return &kPairSplit64;
    return nullptr;
}

static const PairGeomBf16* pair_geom_bf16(int channels)
{
    if (channels == 32) return &kPairBf16_32;
    if (channels == 64) return &kPairBf16_64;
    if (channels == 128) return &kPairBf16_128;
    if (channels == 256) return &kPairBf16_256;
    return nullptr;
}

extern "C" int vk_pair_query(int channels, int precision, vk_tile_info* info)
{
    if (!info) return vk_fail(VK_E_ARG, "vk_pair_query: null info");
    if (precision == VK_PREC_BF16) {
        const PairGeomBf16* G = pair_geom_bf16(channels);
        if (!G) return vk_fail(VK_E_UNSUPPORTED, "vk_pair_query: no fused bf16 kernel for this channel count");
        info->ck = 0; info->ck_bf16 = 32; info->ck_split = 0; info->max_blocks = 0;
        info->rows_per_block = 32 * G->a_rep * G->wm;
        info->time_tile = 32 * G->b_rep * G->wn - VK_PAIR_OVERLAP;
        info->a_rep = G->a_rep;
        return VK_OK;
    }
    if (precision == VK_PREC_F32_SPLIT) {
        const PairGeomSplit* G = pair_geom_split(channels);
        if (!G) return vk_fail(VK_E_UNSUPPORTED, "vk_pair_query: no fused sliced-fp32 kernel for this channel count");
        info->ck = 0; info->ck_bf16 = 0; info->ck_split = VK_SPLIT_CK; info->max_blocks = 0;
        info->rows_per_block = 32;
        info->time_tile = 32 * G->b_rep * G->wn - VK_PAIR_OVERLAP;
        info->a_rep = 1;
        return VK_OK;
    }
    const PairGeom* G = pair_geom(channels);
    if (!G || precision != VK_PREC_F32) return vk_fail(VK_E_UNSUPPORTED, "vk_pair_query: no fused kernel for this channel count");
    info->ck = G->ck;
    info->rows_per_block = 32 * G->wm;
    info->time_tile = 32 * G->b_rep * G->wn - VK_PAIR_OVERLAP;
    info->a_rep = 1;
    info->ck_bf16 = 0;
    info->ck_split = 0;
    info->max_blocks = 0;
    return VK_OK;
}

extern "C" int vk_resblock_pair(const vk_pair_args* a, void* stream)
{
    if (!a) return vk_fail(VK_E_ARG, "vk_resblock_pair: null argument block");
    if (!a->x || !a->w1 || !a->w2 || !a->out) return vk_fail(VK_E_ARG, "vk_resblock_pair: null x / w1 / w2 / out pointer");
    if (a->batch <= 0 || a->length <= 0 || a->dilation <= 0) return vk_fail(VK_E_ARG, "vk_resblock_pair: non-positive size");
    if (a->batch > 65535) return vk_fail(VK_E_UNSUPPORTED, "vk_resblock_pair: batch > 65535");
    const bool bf16 = a->precision == VK_PREC_BF16, split = a->precision == VK_PREC_F32_SPLIT;
    if (a->precision != VK_PREC_F32 && !bf16 && !split) return vk_fail(VK_E_ARG, "vk_resblock_pair: bad precision id");
    const PairGeom* G = (bf16 || split) ? nullptr : pair_geom(a->channels);
    const PairGeomBf16* G16 = bf16 ? pair_geom_bf16(a->channels) : nullptr;
    const PairGeomSplit* GS = split ? pair_geom_split(a->channels) : nullptr;
    if (!G && !G16 && !GS) return vk_fail(VK_E_UNSUPPORTED, "vk_resblock_pair: no fused kernel for this channel count and precision");
    if (a->taps < 1 || !(a->taps & 1) || a->taps - 1 > VK_PAIR_OVERLAP)
        return vk_fail(VK_E_UNSUPPORTED, "vk_resblock_pair: taps must be odd and at most 13");
    const int h2 = (a->taps - 1) / 2;
    const int halo = (a->taps - 1) * a->dilation;
    if (halo > 52) return vk_fail(VK_E_UNSUPPORTED, "vk_resblock_pair: (taps-1)*dilation exceeds 52");
    if (a->epilogue != VK_EPI_RESID && a->epilogue != VK_EPI_RESID_ACC && a->epilogue != VK_EPI_RESID_AVG)
        return vk_fail(VK_E_ARG, "vk_resblock_pair: epilogue must be RESID, RESID_ACC or RESID_AVG");
    if (a->epilogue != VK_EPI_RESID && !a->acc) return vk_fail(VK_E_ARG, "vk_resblock_pair: accumulate epilogue without acc");
    if (!(a->slope >= 0.0f && a->slope <= 1.0f)) return vk_fail(VK_E_UNSUPPORTED, "vk_resblock_pair: slope must lie in [0, 1]");
    if (a->x_cstride < 0 || a->out_cstride < 0 || a->acc_cstride < 0) return vk_fail(VK_E_ARG, "vk_resblock_pair: negative channel stride");
    auto span = [&](int64_t cstride) -> int64_t { return ((int64_t)(a->channels - 1) * cstride + a->length) * 4; };
    const int64_t xb = span(a->x_cstride), ob = span(a->out_cstride), ab = a->acc ? span(a->acc_cstride) : 0;
    if (xb > 0x7fffffffLL || ob > 0x7fffffffLL || ab > 0x7fffffffLL)
        return vk_fail(VK_E_UNSUPPORTED, "vk_resblock_pair: one batch item's view spans more than 2 GiB");
    {   // neighbouring workgroups read halo columns of x, so the output may not land on it
        const char* x0 = (const char*)a->x;  const char* x1 = x0 + (int64_t)(a->batch - 1) * a->x_bstride * 4 + xb;
        const char* o0 = (const char*)a->out; const char* o1 = o0 + (int64_t)(a->batch - 1) * a->out_bstride * 4 + ob;
        if (o0 < x1 && x0 < o1) return vk_fail(VK_E_ARG, "vk_resblock_pair: out overlaps x");
    }
    const int TN = bf16 ? 32 * G16->b_rep * G16->wn : split ? 32 * GS->b_rep * GS->wn : 32 * G->b_rep * G->wn;
    const int ck = bf16 ? 32 : split ? VK_SPLIT_CK : G->ck;
    PairDev q;
    memset(&q, 0, sizeof(q));
    vk_conv_args& c = q.c.a;
    c.x = a->x; c.x_bstride = a->x_bstride; c.x_cstride = a->x_cstride;
    c.w = a->w1; c.bias = a->bias1;
    c.out = a->out; c.out_bstride = a->out_bstride; c.out_cstride = a->out_cstride;
    c.res = a->x; c.res_bstride = a->x_bstride; c.res_cstride = a->x_cstride;
    c.out2 = const_cast<float*>(a->acc); c.out2_bstride = a->acc_bstride; c.out2_cstride = a->acc_cstride;
    c.batch = a->batch; c.c_in = a->channels; c.c_out = a->channels; c.length = a->length;
    c.taps = a->taps; c.dilation = a->dilation;
    c.pad = halo / 2 + h2;                       // column j of the first conv's output is time t0 - h2 + j
    c.epilogue = a->epilogue; c.act_in = 1; c.in_slope = a->slope; c.navg = a->navg;
    q.c.x_bytes = (unsigned)xb; q.c.out_bytes = (unsigned)ob; q.c.res_bytes = (unsigned)xb; q.c.out2_bytes = (unsigned)ab;
    q.c.nchunk = a->channels / ck;
    static const int no_vec = diag_env("VK_NO_VEC", 0);
    const int vec = !no_vec && !bf16 && !split && ((uintptr_t)a->x % 16 == 0) && (a->x_bstride % 4 == 0) && (a->x_cstride % 4 == 0) &&
                    (a->length % 4 == 0);
    if (vec) {
        const int padv = (c.pad + 3) & ~3;
        q.c.col0 = padv - c.pad;
        q.c.xw = (TN + halo + q.c.col0 + 3) & ~3;
    } else {
        q.c.col0 = 0;
        q.c.xw = TN + halo;
    }
    q.c.lds_stride = TN + 64;
    q.c.groups_per_mt = q.c.nchunk * a->taps * (bf16 ? ck / 16 : split ? 3 * (ck / 16) : ck / 8);
    q.w2 = a->w2; q.bias2 = a->bias2; q.groups2 = q.c.groups_per_mt;
    q.c.stamps = g_stamps;
    const int TNO = TN - VK_PAIR_OVERLAP;
    dim3 grid((unsigned)((a->length + TNO - 1) / TNO), 1, (unsigned)a->batch);
    const size_t lds = bf16 ? pair_bf16_lds(*G16) : split ? pair_split_lds(*GS)
                            : (size_t)a->channels * q.c.lds_stride * sizeof(float);
    int e = bf16  ? vk_launch_pair_bf16(a->channels, a->epilogue, q, grid, lds, (hipStream_t)stream)
          : split ? vk_launch_pair_split(a->channels, a->epilogue, q, grid, lds, (hipStream_t)stream)
                  : vk_launch_pair(a->channels, a->epilogue, vec, q, grid, lds, (hipStream_t)stream);
    if (e == -1) return vk_fail(VK_E_UNSUPPORTED, "vk_resblock_pair: this combination is not built");
    if (e != 0) return vk_fail(VK_E_LAUNCH, hipGetErrorString((hipError_t)e));
    return VK_OK;
}

extern "C" int vk_wn_query(int hidden, vk_tile_info* info)
{
    if (!info || hidden != 32 * VK_WN_WAVES) return vk_fail(VK_E_UNSUPPORTED, "vk_wn_query: the fused WN layer is built for hidden = 192");
    info->ck = VK_WN_CK;
    info->rows_per_block = 64;
    info->time_tile = 64;
    info->a_rep = 2;
    info->ck_bf16 = 32;
    info->ck_split = 0;
    info->max_blocks = 0;
    return VK_OK;
}

extern "C" int vk_wn_layer(const vk_wn_args* a, void* stream)
{
    if (!a) return vk_fail(VK_E_ARG, "vk_wn_layer: null argument block");
    if (!a->x || !a->w_in || !a->cond || !a->w_rs || !a->x_out || !a->skip || !a->mask)
        return vk_fail(VK_E_ARG, "vk_wn_layer: null pointer");
    if (a->hidden != 32 * VK_WN_WAVES) return vk_fail(VK_E_UNSUPPORTED, "vk_wn_layer: built for hidden = 192");
    if (a->cond_index && a->cond_rows <= 0) return vk_fail(VK_E_ARG, "vk_wn_layer: cond_index without cond_rows");
    if (a->batch <= 0 || a->batch > 65535 || a->length <= 0 || a->taps < 1 || a->dilation < 1)
        return vk_fail(VK_E_ARG, "vk_wn_layer: bad size");
    const int halo = (a->taps - 1) * a->dilation;
    if (halo > 56 || (halo & 1)) return vk_fail(VK_E_UNSUPPORTED, "vk_wn_layer: (taps-1)*dilation must be even and at most 56");
    if (a->x_cstride < 0 || a->xo_cstride < 0 || a->skip_cstride < 0) return vk_fail(VK_E_ARG, "vk_wn_layer: negative channel stride");
    auto span = [&](int64_t cstride) -> int64_t { return ((int64_t)(a->hidden - 1) * cstride + a->length) * 4; };
    const int64_t xb = span(a->x_cstride), ob = span(a->xo_cstride), sb = span(a->skip_cstride);
    if (xb > 0x7fffffffLL || ob > 0x7fffffffLL || sb > 0x7fffffffLL)
        return vk_fail(VK_E_UNSUPPORTED, "vk_wn_layer: one batch item's view spans more than 2 GiB");
    {
        const char* x0 = (const char*)a->x;      const char* x1 = x0 + (int64_t)(a->batch - 1) * a->x_bstride * 4 + xb;
        const char* o0 = (const char*)a->x_out;  const char* o1 = o0 + (int64_t)(a->batch - 1) * a->xo_bstride * 4 + ob;
        if (o0 < x1 && x0 < o1) return vk_fail(VK_E_ARG, "vk_wn_layer: x_out overlaps x");
    }
    const bool bf16 = a->precision == VK_PREC_BF16;
    if (!bf16 && a->precision != VK_PREC_F32) return vk_fail(VK_E_UNSUPPORTED, "vk_wn_layer: built for VK_PREC_F32 and VK_PREC_BF16");
    // 64-column windows unless that leaves most CUs without a workgroup (bf16 inputs: 128 columns
    // once every CU gets several workgroups of them; diagnostic build: VK_WN_BREP forces the choice)
    const int64_t win64 = (int64_t)a->batch * ((a->length + 63) / 64);
    static const int force_brep = diag_env("VK_WN_BREP", 0);
    // measured, reverse flow with bf16 inputs, 32 / 64 / 128 columns: 16 x 862 frames 0.63 / 0.61 / 0.77 ms,
    // 64 x 862 frames 1.99 / 1.93 / 1.67 ms
    const int b_rep = (force_brep && (bf16 || force_brep <= 2)) ? force_brep
                    : bf16 ? (win64 < 128 ? 1 : win64 < 768 ? 2 : 4) : (win64 < 512 ? 1 : 2);
    const int TN = 32 * b_rep;
    const int ck = bf16 ? 32 : VK_WN_CK;
    WnDev q;
    memset(&q, 0, sizeof(q));
    vk_conv_args& c = q.c.a;
    c.x = a->x; c.x_bstride = a->x_bstride; c.x_cstride = a->x_cstride;
    c.w = a->w_in; c.bias = a->cond; c.bias_bstride = a->cond_bstride;
    c.out = a->x_out; c.out_bstride = a->xo_bstride; c.out_cstride = a->xo_cstride;
    c.res = a->x; c.res_bstride = a->x_bstride; c.res_cstride = a->x_cstride;
    c.out2 = a->skip; c.out2_bstride = a->skip_bstride; c.out2_cstride = a->skip_cstride;
    c.mask = a->mask;
    c.batch = a->batch; c.c_in = a->hidden; c.c_out = 2 * a->hidden; c.length = a->length;
    c.taps = a->taps; c.dilation = a->dilation; c.pad = halo / 2;
    c.first = a->first; c.hidden = a->hidden;
    c.bias_index = a->cond_index; c.bias_rows = a->cond_rows;
    q.c.x_bytes = (unsigned)xb; q.c.out_bytes = (unsigned)ob; q.c.res_bytes = (unsigned)xb; q.c.out2_bytes = (unsigned)sb;
    q.c.nchunk = a->hidden / ck;
    static const int no_vec = diag_env("VK_NO_VEC", 0);
    const int vec = !bf16 && !no_vec && ((uintptr_t)a->x % 16 == 0) && (a->x_bstride % 4 == 0) && (a->x_cstride % 4 == 0) &&
                    (a->length % 4 == 0);
    if (vec) {
        const int padv = (c.pad + 3) & ~3;
        q.c.col0 = padv - c.pad;
        q.c.xw = (TN + halo + q.c.col0 + 3) & ~3;
    } else {
        q.c.col0 = 0;
        q.c.xw = TN + halo;
    }
    q.c.lds_stride = TN + 64;
    q.c.groups_per_mt = q.c.nchunk * a->taps * (bf16 ? ck / 16 : ck / 8);
    q.w_rs = a->w_rs; q.bias_rs = a->bias_rs; q.groups_rs = q.c.nchunk * (bf16 ? ck / 16 : ck / 8);
    q.last = a->last;
    dim3 grid((unsigned)((a->length + TN - 1) / TN), 1, (unsigned)a->batch);
    int e;
    if (bf16) {
        e = vk_launch_wn_layer_bf16(b_rep, a->last ? 1 : 0, q, grid, wn_bf16_lds(b_rep), (hipStream_t)stream);
    } else {
        const size_t xs = (size_t)2 * VK_WN_CK * (TN + 64), as = (size_t)a->hidden * (TN + 4);
        const size_t lds = (xs > as ? xs : as) * sizeof(float);
        e = vk_launch_wn_layer(b_rep, a->last ? 1 : 0, vec, q, grid, lds, (hipStream_t)stream);
    }
    if (e == -1) return vk_fail(VK_E_UNSUPPORTED, "vk_wn_layer: this combination is not built");
    if (e != 0) return vk_fail(VK_E_LAUNCH, hipGetErrorString((hipError_t)e));
    return VK_OK;
}

extern "C" int vk_mrf_query(int channels, int precision, vk_tile_info* info)
{
    if (!info) return vk_fail(VK_E_ARG, "vk_mrf_query: null info");
    const MrfGeom* G = precision == VK_PREC_BF16 ? vk_mrf_geom(channels) :
                       precision == VK_PREC_F32 ? vk_mrf_f32_geom(channels) : nullptr;
    if (!G) return vk_fail(VK_E_UNSUPPORTED, "vk_mrf_query: the fused stage is built for bf16 inputs at 32, 64, 128 and 256 "
                                             "channels and for fp32 at 32 and 64");
    info->ck = precision == VK_PREC_F32 ? 32 : 0;
    info->ck_bf16 = precision == VK_PREC_BF16 ? 32 : 0;
    info->ck_split = 0;
    info->rows_per_block = 32;
    info->time_tile = vk_mrf_window(*G) - 2 * VK_MRF_HALO;       // at least this many output columns per workgroup
    info->a_rep = 1;
    info->max_blocks = G->sum ? VK_MRF_MAX_BLOCKS : 1;
    return VK_OK;
}

extern "C" int vk_mrf_stage(const vk_mrf_args* a, void* stream)
{
    if (!a) return vk_fail(VK_E_ARG, "vk_mrf_stage: null argument block");
    if (!a->x || !a->out) return vk_fail(VK_E_ARG, "vk_mrf_stage: null x / out pointer");
    const bool f32 = a->precision == VK_PREC_F32;
    if (!f32 && a->precision != VK_PREC_BF16) return vk_fail(VK_E_UNSUPPORTED, "vk_mrf_stage: built for VK_PREC_BF16 and VK_PREC_F32");
    const bool single = a->n_blocks == 1;
    const MrfGeom* G = f32 ? vk_mrf_f32_geom(a->channels, single) : vk_mrf_geom(a->channels, single);
    if (!G) return vk_fail(VK_E_UNSUPPORTED, "vk_mrf_stage: built for 32, 64, 128 and 256 channels (fp32: 32 and 64)");
    if (a->batch <= 0 || a->batch > 65535 || a->length <= 0) return vk_fail(VK_E_ARG, "vk_mrf_stage: bad batch or length");
    if (a->n_blocks < 1 || a->n_blocks > (G->sum ? VK_MRF_MAX_BLOCKS : 1) || a->n_pairs < 1 || a->n_pairs > VK_MRF_MAX_PAIRS)
        return vk_fail(VK_E_UNSUPPORTED, "vk_mrf_stage: 1..4 resblocks (one at 128 / 256 channels) of 1..4 pairs");
    if (!(a->navg >= 1.0f)) return vk_fail(VK_E_ARG, "vk_mrf_stage: navg must be at least 1");
    if (a->acc && (a->acc_cstride < 0 || a->acc_bstride < 0)) return vk_fail(VK_E_ARG, "vk_mrf_stage: negative stride");
    if (!(a->slope >= 0.0f && a->slope <= 1.0f)) return vk_fail(VK_E_UNSUPPORTED, "vk_mrf_stage: slope must lie in [0, 1]");
    if (a->x_cstride < 0 || a->out_cstride < 0 || a->x_bstride < 0 || a->out_bstride < 0)
        return vk_fail(VK_E_ARG, "vk_mrf_stage: negative stride");
    MrfDev q;
    memset(&q, 0, sizeof(q));
    q.a = *a;
    int halo = 0;
    for (int r = 0; r < a->n_blocks; ++r) {
        const int k = a->taps[r];
        if (k < 3 || !(k & 1)) return vk_fail(VK_E_UNSUPPORTED, "vk_mrf_stage: taps must be odd and at least 3");
        int reach = 0;
        for (int j = 0; j < a->n_pairs; ++j) {
            if (a->dilation[r][j] < 1) return vk_fail(VK_E_ARG, "vk_mrf_stage: dilation < 1");
            if (!a->w1[r][j] || !a->w2[r][j]) return vk_fail(VK_E_ARG, "vk_mrf_stage: null weight pointer");
            if ((k - 1) / 2 * a->dilation[r][j] > VK_MRF_GUARD)
                return vk_fail(VK_E_UNSUPPORTED, "vk_mrf_stage: (taps-1)/2 * dilation exceeds 32");
            reach += (k - 1) / 2 * (a->dilation[r][j] + 1);
        }
        if (reach > VK_MRF_HALO) return vk_fail(VK_E_UNSUPPORTED, "vk_mrf_stage: a resblock reaches further than 64 samples");
        if (reach > halo) halo = reach;
        q.groups[r] = (a->channels / 32) * k * (f32 ? 4 : 2);
    }
    auto span = [&](int64_t cstride) -> int64_t { return ((int64_t)(a->channels - 1) * cstride + a->length) * 4; };
    const int64_t xb = span(a->x_cstride), ob = span(a->out_cstride);
    if (xb > 0x7fffffffLL || ob > 0x7fffffffLL)
        return vk_fail(VK_E_UNSUPPORTED, "vk_mrf_stage: one batch item's view spans more than 2 GiB");
    {   // other workgroups read halo columns of x for as long as the launch runs
        const char* x0 = (const char*)a->x;   const char* x1 = x0 + (int64_t)(a->batch - 1) * a->x_bstride * 4 + xb;
        const char* o0 = (const char*)a->out; const char* o1 = o0 + (int64_t)(a->batch - 1) * a->out_bstride * 4 + ob;
        if (o0 < x1 && x0 < o1) return vk_fail(VK_E_ARG, "vk_mrf_stage: out overlaps x");
    }
    q.x_bytes = (unsigned)xb; q.out_bytes = (unsigned)ob;
    if (a->acc) {
        const int64_t ab = span(a->acc_cstride);
        if (ab > 0x7fffffffLL) return vk_fail(VK_E_UNSUPPORTED, "vk_mrf_stage: one batch item's view spans more than 2 GiB");
        q.acc_bytes = (unsigned)ab;
    }
    q.halo = halo;                                       // columns recomputed on either side: the longest reach
    q.out_cols = vk_mrf_window(*G) - 2 * halo;
    dim3 grid((unsigned)((a->length + q.out_cols - 1) / q.out_cols), 1, (unsigned)a->batch);
    int e = f32 ? vk_launch_mrf_f32(a->channels, single, q, grid, vk_mrf_f32_lds_bytes(*G), (hipStream_t)stream)
                : vk_launch_mrf_bf16(a->channels, single, q, grid, vk_mrf_lds_bytes(*G), (hipStream_t)stream);
    if (e == -1) return vk_fail(VK_E_UNSUPPORTED, "vk_mrf_stage: this channel count is not built");
    if (e != 0) return vk_fail(VK_E_LAUNCH, hipGetErrorString((hipError_t)e));
    return VK_OK;
}
