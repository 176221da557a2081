"""ctypes binding of libvits_amd.so (the C ABI declared in include/vits_amd.h).

There is no CPU fallback: if the library is missing, or a call fails, an exception is
raised.  Build the library with ``python -m vits_amd.build`` (hipcc, gfx950).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvits_amd.so")
DIAG_LIB_PATH = os.path.join(_HERE, "lib", "libvits_amd_diag.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "vits_amd.h")


def header_abi_version() -> int:
    """VK_ABI_VERSION as include/vits_amd.h declares it (the binding below mirrors that header)."""
    import re
    with open(HEADER_PATH) as f:
        return int(re.search(r"#define\s+VK_ABI_VERSION\s+(\d+)", f.read()).group(1))

TILE_M128, TILE_M64, TILE_M32, TILE_PAIR = 0, 1, 2, 3
PREC_F32, PREC_BF16, PREC_F32_SPLIT = 0, 1, 2
(EPI_STORE, EPI_RESID, EPI_RESID_ACC, EPI_RESID_AVG, EPI_GATE, EPI_WN_MID, EPI_WN_LAST,
 EPI_COUPLE_REV, EPI_COUPLE_FWD, EPI_UPSAMPLE, EPI_SAMPLE) = range(11)

_fp = C.c_void_p


class TileInfo(C.Structure):
    _fields_ = [("ck", C.c_int32), ("rows_per_block", C.c_int32),
                ("time_tile", C.c_int32), ("a_rep", C.c_int32), ("ck_bf16", C.c_int32),
                ("ck_split", C.c_int32), ("max_blocks", C.c_int32)]


class ConvArgs(C.Structure):
    """Mirror of ``vk_conv_args`` (field order and types must match the header)."""
    _fields_ = [
        ("x", _fp), ("x_bstride", C.c_int64), ("x_cstride", C.c_int64),
        ("w", _fp),
        ("bias", _fp), ("bias_bstride", C.c_int64),
        ("out", _fp), ("out_bstride", C.c_int64), ("out_cstride", C.c_int64),
        ("res", _fp), ("res_bstride", C.c_int64), ("res_cstride", C.c_int64),
        ("out2", _fp), ("out2_bstride", C.c_int64), ("out2_cstride", C.c_int64),
        ("out3", _fp), ("aux", _fp),
        ("mask", _fp),
        ("batch", C.c_int32), ("c_in", C.c_int32), ("c_out", C.c_int32), ("length", C.c_int32),
        ("taps", C.c_int32), ("dilation", C.c_int32), ("pad", C.c_int32),
        ("tile", C.c_int32), ("epilogue", C.c_int32),
        ("act_in", C.c_int32), ("in_slope", C.c_float),
        ("first", C.c_int32), ("hidden", C.c_int32), ("up_stride", C.c_int32),
        ("navg", C.c_float),
        ("precision", C.c_int32), ("narrow", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("bias_index", _fp), ("bias_rows", C.c_int32),
        ("in_mask", _fp),
    ]


class PairArgs(C.Structure):
    """Mirror of ``vk_pair_args``."""
    _fields_ = [
        ("x", _fp), ("x_bstride", C.c_int64), ("x_cstride", C.c_int64),
        ("w1", _fp), ("bias1", _fp), ("w2", _fp), ("bias2", _fp),
        ("out", _fp), ("out_bstride", C.c_int64), ("out_cstride", C.c_int64),
        ("acc", _fp), ("acc_bstride", C.c_int64), ("acc_cstride", C.c_int64),
        ("batch", C.c_int32), ("channels", C.c_int32), ("length", C.c_int32),
        ("taps", C.c_int32), ("dilation", C.c_int32), ("epilogue", C.c_int32),
        ("slope", C.c_float), ("navg", C.c_float), ("precision", C.c_int32),
    ]


class WnArgs(C.Structure):
    """Mirror of ``vk_wn_args``."""
    _fields_ = [
        ("x", _fp), ("x_bstride", C.c_int64), ("x_cstride", C.c_int64),
        ("w_in", _fp),
        ("cond", _fp), ("cond_bstride", C.c_int64),
        ("w_rs", _fp), ("bias_rs", _fp),
        ("x_out", _fp), ("xo_bstride", C.c_int64), ("xo_cstride", C.c_int64),
        ("skip", _fp), ("skip_bstride", C.c_int64), ("skip_cstride", C.c_int64),
        ("mask", _fp),
        ("batch", C.c_int32), ("hidden", C.c_int32), ("length", C.c_int32), ("taps", C.c_int32),
        ("dilation", C.c_int32), ("first", C.c_int32), ("last", C.c_int32),
        ("cond_index", _fp), ("cond_rows", C.c_int32),
        ("precision", C.c_int32),
    ]


MRF_MAX_BLOCKS, MRF_MAX_PAIRS = 4, 4


class MrfArgs(C.Structure):
    """Mirror of ``vk_mrf_args``."""
    _fields_ = [
        ("x", _fp), ("x_bstride", C.c_int64), ("x_cstride", C.c_int64),
        ("out", _fp), ("out_bstride", C.c_int64), ("out_cstride", C.c_int64),
        ("batch", C.c_int32), ("channels", C.c_int32), ("length", C.c_int32),
        ("n_blocks", C.c_int32), ("n_pairs", C.c_int32),
        ("taps", C.c_int32 * MRF_MAX_BLOCKS),
        ("dilation", (C.c_int32 * MRF_MAX_PAIRS) * MRF_MAX_BLOCKS),
        ("w1", (_fp * MRF_MAX_PAIRS) * MRF_MAX_BLOCKS),
        ("bias1", (_fp * MRF_MAX_PAIRS) * MRF_MAX_BLOCKS),
        ("w2", (_fp * MRF_MAX_PAIRS) * MRF_MAX_BLOCKS),
        ("bias2", (_fp * MRF_MAX_PAIRS) * MRF_MAX_BLOCKS),
        ("slope", C.c_float), ("precision", C.c_int32),
        ("acc", _fp), ("acc_bstride", C.c_int64), ("acc_cstride", C.c_int64),
        ("navg", C.c_float),
    ]


class DdsArgs(C.Structure):
    """Mirror of ``vk_dds_args``."""
    _fields_ = [
        ("x", _fp), ("x_bstride", C.c_int64), ("x_cstride", C.c_int64),
        ("pre_wt", _fp), ("pre_b", _fp),
        ("add_vec", _fp),
        ("add_map", _fp), ("add_map_bstride", C.c_int64),
        ("layers", _fp),
        ("proj_wt", _fp), ("proj_b", _fp),
        ("mask", _fp),
        ("out", _fp), ("out_bstride", C.c_int64), ("out_cstride", C.c_int64),
        ("affine_m", _fp), ("affine_logs", _fp),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("batch", C.c_int32), ("length", C.c_int32), ("n_layers", C.c_int32), ("taps", C.c_int32),
        ("pass_channel", C.c_int32), ("num_bins", C.c_int32),
        ("tail_bound", C.c_float), ("in_scale", C.c_float),
        ("add_index", _fp), ("add_rows", C.c_int32),
    ]


DDS_CHANNELS = 192


class NativeError(RuntimeError):
    pass


_lib = None
_lib_path = LIB_PATH


def use_diagnostic_library():
    """Timing tools only: load the -DVK_DIAG build (environment switches, time stamps) instead
    of the product library.  Must be called before the first kernel call of the process."""
    global _lib_path
    if _lib is not None:
        raise NativeError("the library is already loaded")
    _lib_path = DIAG_LIB_PATH


def lib():
    """The loaded library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_lib_path):
            raise NativeError(f"{_lib_path} not found: run `python -m vits_amd.build` "
                              "(there is no CPU fallback for the HIP path)")
        L = C.CDLL(_lib_path)
        L.vk_abi_version.restype = C.c_int
        L.vk_last_error.restype = C.c_char_p
        L.vk_tile_query.argtypes = [C.c_int, C.POINTER(TileInfo)]
        L.vk_conv1d.argtypes = [C.POINTER(ConvArgs), C.c_void_p]
        L.vk_conv1d_lds_bytes.argtypes = [C.POINTER(ConvArgs)]
        L.vk_conv1d_lds_bytes.restype = C.c_int64
        L.vk_speaker_linear.argtypes = [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.vk_post_conv_tanh.argtypes = [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_float, C.c_void_p]
        L.vk_pair_query.argtypes = [C.c_int, C.c_int, C.POINTER(TileInfo)]
        L.vk_resblock_pair.argtypes = [C.POINTER(PairArgs), C.c_void_p]
        L.vk_wn_query.argtypes = [C.c_int, C.POINTER(TileInfo)]
        L.vk_wn_layer.argtypes = [C.POINTER(WnArgs), C.c_void_p]
        L.vk_mrf_query.argtypes = [C.c_int, C.c_int, C.POINTER(TileInfo)]
        L.vk_mrf_stage.argtypes = [C.POINTER(MrfArgs), C.c_void_p]
        L.vk_conv1d_static.argtypes = [C.POINTER(ConvArgs), C.c_void_p]
        L.vk_conv1d_workspace_bytes.argtypes = [C.POINTER(ConvArgs)]
        L.vk_conv1d_workspace_bytes.restype = C.c_int64
        L.vk_channel_norm.argtypes = [_fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.vk_rel_attention.argtypes = [_fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.vk_expand_prior.argtypes = [_fp, _fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.vk_dds_workspace_bytes.argtypes = [C.c_int, C.c_int]
        L.vk_dds_workspace_bytes.restype = C.c_int64
        L.vk_dds_condition.argtypes = [C.POINTER(DdsArgs), C.c_void_p]
        L.vk_dds_coupling.argtypes = [C.POINTER(DdsArgs), C.c_void_p]
        L.vk_sizeof_args.argtypes = [C.c_int]
        L.vk_sizeof_args.restype = C.c_size_t
        want = header_abi_version()
        if L.vk_abi_version() != want:
            raise NativeError(f"{_lib_path}: ABI version {L.vk_abi_version()}, include/vits_amd.h declares {want}; "
                              "rebuild with `python -m vits_amd.build`")
        for which, mirror in enumerate((ConvArgs, PairArgs, WnArgs, TileInfo, MrfArgs, DdsArgs)):
            if L.vk_sizeof_args(which) != C.sizeof(mirror):
                raise NativeError(f"{mirror.__name__} is {C.sizeof(mirror)} bytes here, "
                                  f"{L.vk_sizeof_args(which)} in the library: native.py is out of step with the header")
        _lib = L
        if torch.cuda.is_initialized():          # take the side streams' hardware queues before anybody captures a graph
            branch_streams(torch.device("cuda", torch.cuda.current_device()))
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        raise NativeError(f"{what} failed ({rc}): {lib().vk_last_error().decode()}")


_tile_cache = {}


def tile_info(tile: int) -> TileInfo:
    if tile not in _tile_cache:
        ti = TileInfo()
        check(lib().vk_tile_query(tile, C.byref(ti)), "vk_tile_query")
        _tile_cache[tile] = ti
    return _tile_cache[tile]


_pair_cache = {}


def pair_info(channels: int, precision: int = 0):
    """Packing geometry of the fused resblock-pair kernel for this width and precision, or
    None if there is no such kernel."""
    key = (channels, precision)
    if key not in _pair_cache:
        ti = TileInfo()
        _pair_cache[key] = ti if lib().vk_pair_query(channels, precision, C.byref(ti)) == 0 else None
    return _pair_cache[key]


def _device_index(device) -> int:
    idx = device.index if isinstance(device, torch.device) else torch.device(device).index
    return torch.cuda.current_device() if idx is None else idx


def stream_ptr(device) -> int:
    """The raw hipStream_t of torch's current stream on ``device`` (the direct query: building a
    torch.cuda.Stream object per launch costs several microseconds of host time)."""
    return torch._C._cuda_getCurrentRawStream(_device_index(device))


def _view(t: torch.Tensor):
    """(pointer, batch stride, channel stride) of a [B, C, L] fp32 view whose time axis is dense."""
    assert t.dtype == torch.float32 and t.dim() == 3 and (t.shape[2] == 1 or t.stride(2) == 1), \
        (t.dtype, t.shape, t.stride())
    return t.data_ptr(), t.stride(0), t.stride(1)


class LaunchTimer:
    """While active, every kernel launch made through this module is bracketed by a pair of
    events on the launch stream and tagged with its kernel family and algorithmic FLOPs, so
    a benchmark can attribute device time per kernel without an external profiler."""

    active = None

    def __init__(self, only=None):
        self.records = []          # (family, flops, bytes, start_event, end_event, detail)
        self.only = only           # optional set of family names: other launches are not bracketed

    def __enter__(self):
        LaunchTimer.active = self
        return self

    def __exit__(self, *exc):
        LaunchTimer.active = None

    def summary(self):
        """family -> dict(launches, ms, flops, bytes); call after a device synchronize."""
        out = {}
        for fam, flops, nbytes, e0, e1, _ in self.records:
            d = out.setdefault(fam, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
            d["bytes"] += nbytes
        return out


def _timed(family, flops, nbytes, device, launch, detail=""):
    t = LaunchTimer.active
    if t is None or (t.only is not None and family not in t.only):
        return launch()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    s = torch.cuda.current_stream(device)
    e0.record(s)
    rc = launch()
    e1.record(s)
    t.records.append((family, flops, nbytes, e0, e1, detail))
    return rc


NARROW_BELOW_BLOCKS = int(os.environ.get("VITS_AMD_NARROW_BELOW", "1024"))     # default: 256 CUs x 4 resident workgroups

_EPI_NAMES = ["store", "resid", "resid_acc", "resid_avg", "gate", "wn_mid", "wn_last",
              "couple_rev", "couple_fwd", "upsample", "sample"]
_TILE_NAMES = ["m128", "m64", "m32", "pair"]


SPLIT_SMALL_LAUNCHES = os.environ.get("VITS_AMD_SPLIT_K", "1") != "0"
WS_COUNTER_BYTES = 65536
WS_BYTES = WS_COUNTER_BYTES + (64 << 20)     # VK_WS_MAX_BYTES: at most 1024 partial tiles of 64 KiB
_workspaces = {}


def workspace(device):
    """Scratch for split channel loops, one per (device, stream): launches on different streams
    may overlap in time and must not share arrival counters.  Zeroed once; every launch leaves the
    counters zero.  The side streams of ``branch_streams`` get theirs when they are made, so they
    find it whether they run eagerly or inside a capture.  Any other stream that is being captured
    into a graph takes the device's capture workspace (made ahead of the capture by
    ``ensure_capture_workspace``): capture streams come and go, and a workspace made during capture
    would live in the graph's private pool.  A capture that forks onto streams of its own, other
    than those of ``branch_streams``, must keep split-loop launches on one of them."""
    index = _device_index(device)
    key = (index, torch._C._cuda_getCurrentRawStream(index))
    ws = _workspaces.get(key)           # the side streams own theirs from birth, also while they are captured
    if ws is None:
        device = torch.device("cuda", index)
        if torch.cuda.is_current_stream_capturing():
            ws = _workspaces.get((index, "capture"))
            if ws is None:
                raise NativeError("no capture workspace: call native.ensure_capture_workspace(device) before the capture")
            return ws
        ws = torch.zeros(WS_BYTES // 4, dtype=torch.float32, device=device)
        _workspaces[key] = ws
    return ws


_branch_streams = {}


def branch_streams(device, n: int = 2):
    """``n`` side streams per device for launches that may overlap with the current stream (the
    resblocks of a decoder stage at small batch).  Made once and bound to hardware queues at once
    (a first tiny launch): the GPU exposes four hardware queues per process and streams take them
    in the order they first launch, so side streams that come to life after other streams (a graph
    capture makes two) can land on the queue of the stream they are meant to overlap with."""
    index = _device_index(device)
    have = _branch_streams.setdefault(index, [])
    while len(have) < n:
        st = torch.cuda.Stream(device=torch.device("cuda", index))
        with torch.cuda.stream(st):
            # its own split-loop scratch from the start: launches on it may overlap with those of
            # the stream it branches from, eagerly and inside a capture alike
            _workspaces[(index, st.cuda_stream)] = torch.zeros(WS_BYTES // 4, dtype=torch.float32,
                                                               device=torch.device("cuda", index))
        have.append(st)
    return have[:n]


def ensure_capture_workspace(device):
    """Make the workspace that launches on a capturing stream use (see ``workspace``).  Call it
    outside any capture; graphs captured on one device share it, so they must not replay
    concurrently with each other."""
    index = _device_index(device)
    key = (index, "capture")
    if key not in _workspaces:
        device = torch.device("cuda", index)
        _workspaces[key] = torch.zeros(WS_BYTES // 4, dtype=torch.float32, device=device)
        torch.cuda.synchronize(device)
    return _workspaces[key]


def conv1d(x, w_packed, out, *, c_in, c_out, taps, dilation, pad, tile, epilogue,
           bias=None, bias_bstride=0, res=None, out2=None, out3=None, aux=None, mask=None,
           act_in=False, in_slope=1.0, first=False, hidden=0, up_stride=0, navg=1.0, static_scratch=False,
           bias_index=None, in_mask=None):
    """Enqueue one fused convolution on the current stream of ``x``'s device.  ``static_scratch``:
    through ``vk_conv1d_static`` (a split channel loop keeps its partials in the library's own device
    array: no workspace tensor, no allocation in the call)."""
    assert x.is_cuda, "the HIP path needs device tensors"
    a = ConvArgs()
    a.x, a.x_bstride, a.x_cstride = _view(x)
    a.w = w_packed.data_ptr()
    a.bias = bias.data_ptr() if bias is not None else None
    a.bias_bstride = bias_bstride
    a.out, a.out_bstride, a.out_cstride = _view(out)
    if res is not None:
        a.res, a.res_bstride, a.res_cstride = _view(res)
    if out2 is not None:
        a.out2, a.out2_bstride, a.out2_cstride = _view(out2)
    if out3 is not None:
        p3, b3, c3 = _view(out3)
        assert (b3, c3) == (a.out_bstride, a.out_cstride)
        a.out3 = p3
    if aux is not None:
        pa, ba, ca = _view(aux)
        assert (ba, ca) == (a.out_bstride, a.out_cstride)
        a.aux = pa
    if mask is not None:
        assert mask.dtype == torch.float32 and mask.is_contiguous()
        a.mask = mask.data_ptr()
    if bias_index is not None:          # bias is a per-speaker table, bias_index[b] the row of item b
        assert bias is not None and bias_index.dtype == torch.int64 and bias_index.is_contiguous() \
            and bias_index.numel() == x.shape[0] and bias_index.device == x.device
        a.bias_index, a.bias_rows = bias_index.data_ptr(), bias.shape[0]
    if in_mask is not None:             # x is read as x * in_mask[b, t]
        assert in_mask.dtype == torch.float32 and in_mask.is_contiguous() and in_mask.numel() == x.shape[0] * x.shape[2]
        a.in_mask = in_mask.data_ptr()
    a.batch, a.c_in, a.c_out, a.length = x.shape[0], c_in, c_out, x.shape[2]
    a.taps, a.dilation, a.pad = taps, dilation, pad
    a.tile, a.epilogue = tile, epilogue
    a.act_in, a.in_slope = int(act_in), in_slope
    a.first, a.hidden, a.up_stride, a.navg = int(first), hidden, up_stride, navg
    # the packed weight format names the matrix-core input precision
    # (packing.py tags sliced-fp32 streams, which are bf16 tensors too)
    assert w_packed.dtype in (torch.float32, torch.bfloat16)
    precision = getattr(w_packed, "vk_precision", PREC_BF16 if w_packed.dtype == torch.bfloat16 else PREC_F32)
    a.precision = precision
    # launches with too few workgroups to fill the GPU take the family's half-width time tile
    rows_all = c_out * up_stride if epilogue == EPI_UPSAMPLE else (2 * hidden if epilogue in (EPI_GATE, EPI_SAMPLE) else c_out)
    ti = tile_info(tile)
    blocks = -(-x.shape[2] // ti.time_tile) * -(-rows_all // ti.rows_per_block) * x.shape[0]
    a.narrow = int(precision in (PREC_F32, PREC_BF16) and blocks < NARROW_BELOW_BLOCKS)
    L = lib()
    if precision in (PREC_F32, PREC_F32_SPLIT) and blocks < 512 and SPLIT_SMALL_LAUNCHES and not static_scratch:
        # far too few workgroups for 256 CUs: with scratch the library splits the channel loop
        # (it decides itself whether to; under capture only launches that will split ask for the
        # capture workspace, so a foreign capture without one keeps working for the others)
        if not torch.cuda.is_current_stream_capturing() or L.vk_conv1d_workspace_bytes(C.byref(a)) > 0:
            ws = workspace(x.device)
            a.workspace, a.workspace_bytes = ws.data_ptr(), WS_BYTES
    rows = c_out * up_stride if epilogue == EPI_UPSAMPLE else c_out
    flops = 2.0 * x.shape[0] * x.shape[2] * rows * c_in * taps
    family = (f"conv_{('mfma', 'bf16', 'split')[precision]}"
              f"[{_TILE_NAMES[tile]}{'n' if a.narrow else ''},{_EPI_NAMES[epilogue]},act={int(act_in)}]")
    sp = stream_ptr(x.device)
    detail = f"B{x.shape[0]} L{x.shape[2]} {c_in}->{rows} k{taps} d{dilation}"
    # algorithmic bytes: every activation view the launch reads or writes, once
    nbytes = 4.0 * (x.shape[0] * c_in * x.shape[2] + out.numel())
    if res is not None and res.data_ptr() != x.data_ptr():
        nbytes += 4.0 * res.numel()
    if out2 is not None and epilogue in (EPI_RESID_ACC, EPI_RESID_AVG, EPI_WN_MID, EPI_WN_LAST, EPI_SAMPLE):
        nbytes += 4.0 * out2.numel()
    entry = L.vk_conv1d_static if static_scratch else L.vk_conv1d
    check(_timed(family, flops, nbytes, x.device, lambda: entry(C.byref(a), sp), detail),
          "vk_conv1d_static" if static_scratch else "vk_conv1d")


def resblock_pair(x, w1, bias1, w2, bias2, out, *, taps, dilation, epilogue, slope, acc=None, navg=1.0):
    """out = conv2(lrelu(conv1(lrelu(x)))) + x (+ acc) in one launch; ``out`` must not alias ``x``."""
    assert x.is_cuda, "the HIP path needs device tensors"
    a = PairArgs()
    a.x, a.x_bstride, a.x_cstride = _view(x)
    a.w1, a.w2 = w1.data_ptr(), w2.data_ptr()
    a.bias1 = bias1.data_ptr() if bias1 is not None else None
    a.bias2 = bias2.data_ptr() if bias2 is not None else None
    a.out, a.out_bstride, a.out_cstride = _view(out)
    if acc is not None:
        a.acc, a.acc_bstride, a.acc_cstride = _view(acc)
    a.batch, a.channels, a.length = x.shape
    a.taps, a.dilation, a.epilogue = taps, dilation, epilogue
    a.slope, a.navg = slope, navg
    assert w1.dtype == w2.dtype and w1.dtype in (torch.float32, torch.bfloat16)
    # the packed weight format names the matrix-core input precision (sliced fp32 is stored as bf16 slices)
    a.precision = getattr(w1, "vk_precision", PREC_BF16 if w1.dtype == torch.bfloat16 else PREC_F32)
    ch = x.shape[1]
    flops = 2.0 * x.shape[0] * x.shape[2] * ch * ch * taps * 2
    family = f"resblock_pair{('', '_bf16', '_split')[a.precision]}[c{ch},{_EPI_NAMES[epilogue]}]"
    L, sp = lib(), stream_ptr(x.device)
    detail = f"B{x.shape[0]} L{x.shape[2]} c{ch} k{taps} d{dilation}"
    nbytes = 4.0 * (x.numel() + out.numel() + (acc.numel() if acc is not None else 0))
    check(_timed(family, flops, nbytes, x.device, lambda: L.vk_resblock_pair(C.byref(a), sp), detail),
          "vk_resblock_pair")


_mrf_cache = {}


def mrf_info(channels: int, precision: int):
    """Packing geometry of the fused multi-receptive-field stage kernel, or None if there is none
    for this width and precision."""
    key = (channels, precision)
    if key not in _mrf_cache:
        ti = TileInfo()
        _mrf_cache[key] = ti if lib().vk_mrf_query(channels, precision, C.byref(ti)) == 0 else None
    return _mrf_cache[key]


def mrf_stage(x, out, blocks, *, slope, acc=None, navg=None):
    """out = (acc + sum over resblocks of resblock(x)) / navg in one launch (navg: the number of
    resblocks unless given; acc: the sum carried from earlier launches of the stage, may be ``out``).
    ``blocks``: per resblock (taps, [(w1, bias1, w2, bias2, dilation) per pair]) with
    fragment-packed weights, bf16 (bf16 matrix inputs) or fp32 (the exact fp32 instruction)."""
    assert x.is_cuda and x.dtype == torch.float32 and out.dtype == torch.float32
    a = MrfArgs()
    a.x, a.x_bstride, a.x_cstride = _view(x)
    a.out, a.out_bstride, a.out_cstride = _view(out)
    if acc is not None:
        assert acc.dtype == torch.float32 and acc.shape == out.shape
        a.acc, a.acc_bstride, a.acc_cstride = _view(acc)
    a.navg = float(len(blocks)) if navg is None else float(navg)
    a.batch, a.channels, a.length = x.shape
    a.n_blocks, a.n_pairs = len(blocks), len(blocks[0][1])
    flops = 0.0
    wdt = blocks[0][1][0][0].dtype
    for r, (taps, pairs) in enumerate(blocks):
        assert len(pairs) == a.n_pairs
        a.taps[r] = taps
        for j, (w1, b1, w2, b2, dil) in enumerate(pairs):
            assert w1.dtype == wdt and w2.dtype == wdt and wdt in (torch.bfloat16, torch.float32)
            a.dilation[r][j] = dil
            a.w1[r][j], a.w2[r][j] = w1.data_ptr(), w2.data_ptr()
            a.bias1[r][j] = b1.data_ptr() if b1 is not None else None
            a.bias2[r][j] = b2.data_ptr() if b2 is not None else None
            flops += 4.0 * x.shape[0] * x.shape[2] * x.shape[1] * x.shape[1] * taps
    a.slope, a.precision = slope, (PREC_BF16 if wdt == torch.bfloat16 else PREC_F32)
    L, sp = lib(), stream_ptr(x.device)
    nbytes = 4.0 * (x.numel() + out.numel() + (acc.numel() if acc is not None else 0))
    family = f"mrf_stage{'_bf16' if wdt == torch.bfloat16 else ''}[c{x.shape[1]}]"
    check(_timed(family, flops, nbytes, x.device,
                 lambda: L.vk_mrf_stage(C.byref(a), sp), f"B{x.shape[0]} L{x.shape[2]} c{x.shape[1]}"),
          "vk_mrf_stage")


_wn_cache = {}


def wn_info(hidden: int):
    """Packing geometry of the fused WN layer kernel for this width, or None if there is none."""
    if hidden not in _wn_cache:
        ti = TileInfo()
        _wn_cache[hidden] = ti if lib().vk_wn_query(hidden, C.byref(ti)) == 0 else None
    return _wn_cache[hidden]


def wn_layer(x, w_in, cond, cond_bstride, w_rs, bias_rs, x_out, skip, mask, *, taps, dilation, first, last,
             cond_index=None):
    """One WN layer: gate conv + conditioning + tanh*sigmoid + 1x1 res/skip conv + residual, skip
    and mask updates in one launch.  ``x_out`` (the output on the last layer) must not alias ``x``."""
    assert x.is_cuda and mask.dtype == torch.float32 and mask.is_contiguous()
    a = WnArgs()
    a.x, a.x_bstride, a.x_cstride = _view(x)
    a.w_in, a.w_rs = w_in.data_ptr(), w_rs.data_ptr()
    a.cond, a.cond_bstride = cond.data_ptr(), cond_bstride
    if cond_index is not None:
        assert cond_index.dtype == torch.int64 and cond_index.is_contiguous() and cond_index.numel() == x.shape[0]
        a.cond_index, a.cond_rows = cond_index.data_ptr(), cond.shape[0]
    a.bias_rs = bias_rs.data_ptr() if bias_rs is not None else None
    a.x_out, a.xo_bstride, a.xo_cstride = _view(x_out)
    a.skip, a.skip_bstride, a.skip_cstride = _view(skip)
    a.mask = mask.data_ptr()
    a.batch, a.hidden, a.length = x.shape
    a.taps, a.dilation, a.first, a.last = taps, dilation, int(first), int(last)
    assert w_in.dtype == w_rs.dtype and w_in.dtype in (torch.float32, torch.bfloat16)
    bf16 = w_in.dtype == torch.bfloat16
    a.precision = PREC_BF16 if bf16 else PREC_F32
    B, H, T = x.shape
    flops = 2.0 * B * T * H * (2 * H * taps + (H if last else 2 * H))
    nbytes = 4.0 * B * H * T * (3 if last else 5)
    L, sp = lib(), stream_ptr(x.device)
    check(_timed(f"wn_layer{'_bf16' if bf16 else ''}[{'last' if last else 'mid'}]", flops, nbytes, x.device,
                 lambda: L.vk_wn_layer(C.byref(a), sp), f"B{B} L{T} h{H} k{taps} d{dilation}"), "vk_wn_layer")


def _dds_common(a, x, pre_wt, pre_b, layers, n_layers, taps, proj_wt, proj_b, mask, out):
    assert x.is_cuda and mask.dtype == torch.float32 and mask.is_contiguous() and mask.dim() == 2
    a.x, a.x_bstride, a.x_cstride = _view(x)
    a.pre_wt, a.pre_b = pre_wt.data_ptr(), pre_b.data_ptr()
    a.layers, a.n_layers, a.taps = layers.data_ptr(), n_layers, taps
    a.proj_wt, a.proj_b = proj_wt.data_ptr(), proj_b.data_ptr()
    a.mask = mask.data_ptr()
    a.out, a.out_bstride, a.out_cstride = _view(out)
    a.batch, a.length = x.shape[0], x.shape[2]
    if n_layers > 1:                      # the stream between the layers' launches
        need = lib().vk_dds_workspace_bytes(a.batch, a.length)
        ws = torch.empty(need // 4, device=x.device, dtype=torch.float32)
        a.workspace, a.workspace_bytes = ws.data_ptr(), need
        return ws
    return None


def dds_condition(x, pre_wt, pre_b, add_vec, layers, n_layers, taps, proj_wt, proj_b, mask, out, add_index=None):
    """out = proj(DDSConv(pre(x) + add_vec)) * mask, one launch per layer: x, out [B, 192, T]; 1x1 weights
    input-major; ``layers`` packed as the header describes; add_vec [B, 192] or None."""
    a = DdsArgs()
    ws = _dds_common(a, x, pre_wt, pre_b, layers, n_layers, taps, proj_wt, proj_b, mask, out)  # noqa: F841 (kept alive)
    a.add_vec = add_vec.data_ptr() if add_vec is not None else None
    if add_index is not None:           # add_vec is a per-speaker table [n_speakers, 192]
        assert add_vec is not None and add_index.dtype == torch.int64 and add_index.is_contiguous()
        a.add_index, a.add_rows = add_index.data_ptr(), add_vec.shape[0]
    B, H, T = x.shape
    flops = 2.0 * B * T * H * H * (n_layers + 2)
    L, sp = lib(), stream_ptr(x.device)
    check(_timed("dds_condition", flops, 8.0 * B * H * T, x.device,
                 lambda: L.vk_dds_condition(C.byref(a), sp), f"B{B} T{T}"), "vk_dds_condition")


def dds_coupling(z, pre_w, pre_b, g, layers, n_layers, taps, proj_wt, proj_b, mask, out, *, pass_channel,
                 num_bins, tail_bound, in_scale=1.0, affine=None):
    """One reversed ConvFlow: out[:, pass] = z[:, pass], out[:, other] = inverse spline of z[:, other]
    with parameters proj(DDSConv(pre(z[:, pass]) + g)), both times mask; ``affine`` = (m, logs) of a
    reversed ElementwiseAffine applied to the result.  z, out [B, 2, T] (distinct), g [B, 192, T]."""
    assert g.is_contiguous() and g.dtype == torch.float32
    a = DdsArgs()
    ws = _dds_common(a, z, pre_w, pre_b, layers, n_layers, taps, proj_wt, proj_b, mask, out)  # noqa: F841 (kept alive)
    a.add_map, a.add_map_bstride = g.data_ptr(), g.stride(0)
    a.pass_channel, a.num_bins, a.tail_bound, a.in_scale = pass_channel, num_bins, tail_bound, in_scale
    if affine is not None:
        a.affine_m, a.affine_logs = affine[0].data_ptr(), affine[1].data_ptr()
    B, _, T = z.shape
    flops = 2.0 * B * T * DDS_CHANNELS * DDS_CHANNELS * n_layers
    L, sp = lib(), stream_ptr(z.device)
    check(_timed("dds_coupling", flops, 4.0 * B * DDS_CHANNELS * T, z.device,
                 lambda: L.vk_dds_coupling(C.byref(a), sp), f"B{B} T{T}"), "vk_dds_coupling")


ATT_MAX_LENGTH = 1024


def channel_norm(x, gamma, beta, mask, out):
    """out = LayerNorm over channels of x [B, 192, T] (dense) times mask [B, T] (or None)."""
    assert x.is_cuda and x.is_contiguous() and out.is_contiguous() and x.dtype == torch.float32
    B, Cn, T = x.shape
    L, sp = lib(), stream_ptr(x.device)
    check(_timed("channel_norm", 8.0 * x.numel(), 8.0 * x.numel(), x.device,
                 lambda: L.vk_channel_norm(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                           mask.data_ptr() if mask is not None else None, out.data_ptr(),
                                           B, Cn, T, sp), f"B{B} T{T}"), "vk_channel_norm")


def rel_attention(qkv, emb_k, emb_v, mask, out, *, heads, window):
    """Windowed relative-position attention: qkv [B, 3 * C, T] (queries, keys, values), out [B, C, T]."""
    assert qkv.is_cuda and qkv.is_contiguous() and out.is_contiguous() and mask.is_contiguous()
    assert emb_k.is_contiguous() and emb_v.is_contiguous()
    B, C3, T = qkv.shape
    head_dim = C3 // 3 // heads
    L, sp = lib(), stream_ptr(qkv.device)
    check(_timed("rel_attention", 4.0 * B * heads * T * T * head_dim, 4.0 * (qkv.numel() + out.numel()), qkv.device,
                 lambda: L.vk_rel_attention(qkv.data_ptr(), emb_k.data_ptr(), emb_v.data_ptr(), mask.data_ptr(),
                                            out.data_ptr(), B, heads, head_dim, window, T, sp), f"B{B} T{T}"),
          "vk_rel_attention")


EXPAND_MAX_TOKENS = 2048


def expand_prior(durations, m, logs, frames):
    """Token-level (m, logs) [B, C, Tx] spread over ``frames`` frames by the whole-number
    ``durations`` [B, Tx]: returns (m_frames, logs_frames [B, C, frames], frame_mask [B, frames])."""
    assert m.is_cuda and m.is_contiguous() and logs.is_contiguous() and durations.is_contiguous()
    assert m.dtype == logs.dtype == durations.dtype == torch.float32
    B, Cn, Tx = m.shape
    m_out = torch.empty(B, Cn, frames, device=m.device, dtype=torch.float32)
    logs_out = torch.empty_like(m_out)
    fmask = torch.empty(B, frames, device=m.device, dtype=torch.float32)
    L, sp = lib(), stream_ptr(m.device)
    check(_timed("expand_prior", 0.0, 16.0 * B * Cn * frames, m.device,
                 lambda: L.vk_expand_prior(durations.data_ptr(), m.data_ptr(), logs.data_ptr(), m_out.data_ptr(),
                                           logs_out.data_ptr(), fmask.data_ptr(), B, Cn, Tx, frames, sp),
                 f"B{B} Tx{Tx} Ty{frames}"), "vk_expand_prior")
    return m_out, logs_out, fmask


def speaker_linear(g, w, bias, out):
    """out[b, :] = w @ g[b, :] + bias for g [B, n_in], w [n_out, n_in]."""
    assert g.is_cuda and g.is_contiguous() and w.is_contiguous() and out.is_contiguous()
    L, sp = lib(), stream_ptr(g.device)
    flops = 2.0 * g.shape[0] * w.shape[0] * w.shape[1]
    check(_timed("speaker_linear", flops, 4.0 * w.numel(), g.device,
                 lambda: L.vk_speaker_linear(g.data_ptr(), w.data_ptr(),
                                             bias.data_ptr() if bias is not None else None,
                                             out.data_ptr(), g.shape[0], w.shape[1], w.shape[0], sp)),
          "vk_speaker_linear")


def post_conv_tanh(x, w, wav, slope):
    assert x.is_cuda and x.is_contiguous() and w.is_contiguous() and wav.is_contiguous()
    L, sp = lib(), stream_ptr(x.device)
    flops = 2.0 * x.numel() * w.shape[-1]
    check(_timed("post_conv_tanh", flops, 4.0 * (x.numel() + wav.numel()), x.device,
                 lambda: L.vk_post_conv_tanh(x.data_ptr(), w.data_ptr(), wav.data_ptr(), x.shape[0],
                                             x.shape[1], x.shape[2], w.shape[-1], slope, sp)),
          "vk_post_conv_tanh")
