"""Shared host-side pieces of the HIP path: parameter holders that keep the reference's
``state_dict`` layout, tile selection, and the per-module cache of packed weights."""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn

from . import native, packing


class PlainConvParams(nn.Module):
    """Parameters of an ``nn.Conv1d`` (keys ``weight``, ``bias``), no compute."""

    def __init__(self, c_in: int, c_out: int, k: int, bias: bool = True):
        super().__init__()
        bound = 1.0 / math.sqrt(c_in * k)
        self.weight = nn.Parameter(torch.empty(c_out, c_in, k).uniform_(-bound, bound))
        if bias:
            self.bias = nn.Parameter(torch.empty(c_out).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)

    def effective_weight(self) -> torch.Tensor:
        return self.weight.detach()


class WeightNormConvParams(nn.Module):
    """Parameters of an old-style weight-normed conv, registered in the order the
    reference's ``nn.utils.weight_norm`` leaves them: ``bias``, ``weight_g``, ``weight_v``.
    ``transposed`` selects the ConvTranspose1d weight shape [c_in, c_out, k] (norm axis =
    input channels)."""

    def __init__(self, c_in: int, c_out: int, k: int, transposed: bool = False):
        super().__init__()
        bound = 1.0 / math.sqrt((c_out if transposed else c_in) * k)
        shape = (c_in, c_out, k) if transposed else (c_out, c_in, k)
        v = torch.empty(shape).uniform_(-bound, bound)
        self.bias = nn.Parameter(torch.empty(c_out).uniform_(-bound, bound))
        self.weight_g = nn.Parameter(v.flatten(1).norm(dim=1).view(-1, 1, 1))
        self.weight_v = nn.Parameter(v)

    def effective_weight(self) -> torch.Tensor:
        return packing.fold_weight_norm(self.weight_v.detach(), self.weight_g.detach())


_PRECISIONS = {"fp32": native.PREC_F32, "bf16": native.PREC_BF16, "fp32-split": native.PREC_F32_SPLIT}
_precision = _PRECISIONS.get(os.environ.get("VITS_AMD_PRECISION", "fp32").lower(), native.PREC_F32)


def set_precision(name: str) -> None:
    """Matrix-core input precision of every convolution launched afterwards: "fp32" (exact fp32
    products, the <= 1e-4 parity path) or "bf16" (activations and weights rounded to bf16 at the
    matrix-core inputs, fp32 accumulation and fp32 tensors in memory) or "fp32-split" (fp32
    operands cut exactly into three bf16 slices, six slice products per product on the bf16
    matrix cores, fp32 accumulation: fp32-level accuracy at about 2.7x the fp32 matrix rate)."""
    global _precision
    if name not in _PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}")
    _precision = _PRECISIONS[name]


def precision() -> int:
    return _precision


def precision_name() -> str:
    return next(k for k, v in _PRECISIONS.items() if v == _precision)


def choose_tile(rows: int) -> int:
    """Tile family for a GEMM with ``rows`` output rows: least row padding, then the
    largest rows-per-workgroup."""
    best = None
    for tile in (native.TILE_M128, native.TILE_M64, native.TILE_M32):
        rpb = native.tile_info(tile).rows_per_block
        waste = (-rows) % rpb
        key = (waste, -rpb)
        if best is None or key < best[0]:
            best = (key, tile)
    return best[1]


class PackedConv:
    """A convolution's weights in fragment order on one device, plus launch constants."""

    def __init__(self, w: torch.Tensor, bias, device, *, tile=None, row_order=None, dilation=1,
                 pad=None, rows=None):
        w = w.to(device=device, dtype=torch.float32)
        self.rows = w.shape[0] if rows is None else rows
        self.c_in, self.taps = w.shape[1], w.shape[2]
        self.dilation = dilation
        self.pad = (self.taps - 1) * dilation // 2 if pad is None else pad
        self.tile = choose_tile(w.shape[0]) if tile is None else tile
        ti = native.tile_info(self.tile)
        self.precision = precision()
        if self.precision == native.PREC_F32_SPLIT:
            self.w = packing.pack_conv_weight_split(w, ti.ck_split, ti.rows_per_block, row_order)
        elif self.precision == native.PREC_BF16:
            self.w = packing.pack_conv_weight_bf16(w, ti.ck_bf16, ti.rows_per_block, row_order)
        else:
            self.w = packing.pack_conv_weight(w, ti.ck, ti.rows_per_block, row_order)
        self.bias = None if bias is None else bias.detach().to(device=device, dtype=torch.float32).contiguous()


# Bumped whenever a Parameter is (re)registered on any module: the caches below keep a flat list of
# their module's parameters (walking the module tree costs several times more than checking the
# parameters themselves) and refresh it when this moves.
_registrations = [0]


def _count_registration(module, name, param):
    _registrations[0] += 1
    return None


torch.nn.modules.module.register_module_parameter_registration_hook(_count_registration)


class PlanCache:
    """Rebuilds a module's packed weights when its parameters or device change."""

    def __init__(self, module: nn.Module, builder):
        self._module = module
        self._builder = builder
        self._key = None
        self._plan = None
        self._params = None
        self._seen = -1

    def get(self, device):
        if self._seen != _registrations[0]:
            self._params = list(self._module.parameters())
            self._seen = _registrations[0]
        key = (str(device), precision()) + tuple((p.data_ptr(), p._version) for p in self._params)
        if key != self._key:
            with torch.no_grad():
                self._plan = self._builder(device)
            self._key = key
        return self._plan


def require_device(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise native.NativeError(
            f"{what}: the HIP path needs tensors on the MI355X (got {t.device}); there is no CPU fallback")
