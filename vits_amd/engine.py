"""Shared host-side pieces of the HIP path: parameter holders that keep the reference's
``state_dict`` layout, tile selection, and the per-module cache of packed weights."""
from __future__ import annotations

import math
import os
import threading

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
_building = threading.local()       # .stack: precisions of the plans being built on this thread, innermost last


def set_precision(name: str) -> None:
    """The process default of the matrix-core input precision: "fp32" (exact fp32 products, the
    <= 1e-4 parity path) or "bf16" (activations and weights rounded to bf16 at the matrix-core
    inputs, fp32 accumulation and fp32 tensors in memory) or "fp32-split" (fp32 operands cut exactly
    into three bf16 slices, six slice products per product on the bf16 matrix cores, fp32
    accumulation: fp32-level accuracy at about 2.7x the fp32 matrix rate).  A module on which
    ``set_module_precision`` has been called keeps its own."""
    global _precision
    if name not in _PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}")
    _precision = _PRECISIONS[name]


def set_module_precision(module: nn.Module, name) -> None:
    """Matrix-core input precision of ``module`` and everything under it (None: back to the process
    default).  It is part of the key of every plan cache below, so two models in one process can
    run in different precisions side by side; a launch takes the precision of its packed weights."""
    if name is not None and name not in _PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(_PRECISIONS)} or None")
    for m in module.modules():
        m.vk_precision = None if name is None else _PRECISIONS[name]


def module_precision(module: nn.Module) -> int:
    """What ``module``'s plans are built for: its own setting, else that of the plan being built
    around it on this thread, else the process default."""
    own = getattr(module, "vk_precision", None)
    return precision() if own is None else own


def precision() -> int:
    """Inside a plan builder: the precision that plan is built for.  Elsewhere: the process default."""
    stack = getattr(_building, "stack", None)
    return stack[-1] if stack else _precision


def precision_name(prec=None) -> str:
    prec = precision() if prec is None else prec
    return next(k for k, v in _PRECISIONS.items() if v == prec)


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
        self.precision = precision()              # of the plan being built (see PlanCache.get)
        if self.precision == native.PREC_F32_SPLIT:
            self.w = packing.pack_conv_weight_split(w, ti.ck_split, ti.rows_per_block, row_order)
        elif self.precision == native.PREC_BF16:
            self.w = packing.pack_conv_weight_bf16(w, ti.ck_bf16, ti.rows_per_block, row_order)
        else:
            self.w = packing.pack_conv_weight(w, ti.ck, ti.rows_per_block, row_order)
        self.bias = None if bias is None else bias.detach().to(device=device, dtype=torch.float32).contiguous()


# Bumped whenever a Parameter or a sub-module is (re)registered on any module: the caches below keep
# a flat list of their module's parameters (walking the module tree costs several times more than
# checking the parameters themselves) and refresh it when this moves.
_registrations = [0]


def _count_registration(module, name, value):
    _registrations[0] += 1
    return None


torch.nn.modules.module.register_module_parameter_registration_hook(_count_registration)
torch.nn.modules.module.register_module_module_registration_hook(_count_registration)


class PlanCache:
    """Rebuilds a module's packed weights when its parameters, device or precision change."""

    def __init__(self, module: nn.Module, builder):
        self._module = module
        self._builder = builder
        self._key = None
        self._plan = None
        self._params = None
        self._slots = None
        self._seen = -1

    def _refresh(self):
        # (module, name) of every parameter slot as well: a parameter written straight into
        # ``_parameters`` (conversions that overwrite module parameters) passes no hook
        slots = [(m, n) for m in self._module.modules() for n in m._parameters]
        self._slots = slots
        self._params = [m._parameters[n] for m, n in slots]
        self._seen = _registrations[0]

    def get(self, device):
        if self._seen != _registrations[0]:
            self._refresh()
        elif any(m._parameters.get(n) is not p for (m, n), p in zip(self._slots, self._params)):
            self._refresh()
        prec = module_precision(self._module)
        key = (str(device), prec) + tuple((p.data_ptr(), p._version) for p in self._params if p is not None)
        if key != self._key:
            stack = getattr(_building, "stack", None)
            if stack is None:
                stack = _building.stack = []
            stack.append(prec)
            try:
                with torch.no_grad():
                    self._plan = self._builder(device)
            finally:
                stack.pop()
            self._key = key
        return self._plan


class SpeakerRef:
    """Speaker ids of a batch together with the embedding table they index
    (/root/reference/module/vits_generator.py:88-91,170,191-192).  Modules given a SpeakerRef where the
    reference passes ``speaker_id_embedded`` read their speaker conditioning as rows of a per-speaker
    table instead of computing it per call; ``embedded()`` is the tensor the reference would pass."""

    def __init__(self, ids: torch.Tensor, embedding: nn.Embedding):
        self.ids = ids.to(torch.int64).contiguous()
        self.embedding = embedding

    def embedded(self) -> torch.Tensor:
        return self.embedding(self.ids).unsqueeze(-1)


class Conditioning:
    """Where a launch finds the speaker-conditioning vector of batch item b: row b of ``rows``
    [B, n], or with ``index`` row index[b] of a per-speaker table [n_speakers, n]."""

    __slots__ = ("rows", "index")

    def __init__(self, rows, index=None):
        self.rows, self.index = rows, index

    @property
    def stride(self) -> int:
        return self.rows.stride(0)

    def columns(self, start: int) -> torch.Tensor:
        return self.rows[:, start:]


class SpeakerTables:
    """Per plan: out = W g + b for every row g of a speaker embedding table, computed once (with the
    GEMV kernel a per-call conditioning uses, so the rows are the values that call would give) and
    kept until the table's storage or version changes.  The plan that owns this goes when the
    module's own weights move."""

    def __init__(self, cond_w: torch.Tensor, cond_b: torch.Tensor):
        self.cond_w, self.cond_b = cond_w, cond_b
        self._key, self._table = None, None

    def conditioning(self, speaker, batch: int) -> Conditioning:
        """``speaker``: a SpeakerRef, or the embedded speakers [B, G, 1] as the reference passes them."""
        dev = self.cond_w.device
        if isinstance(speaker, SpeakerRef):
            w = speaker.embedding.weight
            key = (w.data_ptr(), w._version, str(w.device))
            if key != self._key and not torch.cuda.is_current_stream_capturing():
                table = torch.empty(w.shape[0], self.cond_w.shape[0], device=dev, dtype=torch.float32)
                native.speaker_linear(w.detach().to(device=dev, dtype=torch.float32).contiguous(),
                                      self.cond_w, self.cond_b, table)
                self._key, self._table = key, table
            if key == self._key:
                return Conditioning(self._table, speaker.ids.to(dev))
            speaker = speaker.embedded()          # a capture met a table that is not built yet: per-call form
        g2 = speaker.to(device=dev, dtype=torch.float32).reshape(batch, -1).contiguous()
        out = torch.empty(batch, self.cond_w.shape[0], device=dev, dtype=torch.float32)
        native.speaker_linear(g2, self.cond_w, self.cond_b, out)
        return Conditioning(out)


def require_device(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise native.NativeError(
            f"{what}: the HIP path needs tensors on the MI355X (got {t.device}); there is no CPU fallback")
