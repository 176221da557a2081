"""Residual-coupling flow (mean-only affine coupling + channel flip), both directions.

Drop-in for /root/reference/module/model_component/flow.py: ``Flip`` (:18-21),
``ResidualCouplingLayer`` (:23-71) and ``Flow`` (:75-118) with the same constructor
arguments, ``forward`` signatures and ``state_dict`` keys (``flows.{0,2,4,6}.pre/wn/post``;
the odd entries are parameter-free ``Flip`` modules).

The channel flip costs nothing here.  The latent stays in one physical buffer and a
coupling layer that sits behind an odd number of flips reads/writes it through weights
whose channel axes were reversed when they were packed: its ``pre`` reads physical
channels 96..191 with reversed input columns, its ``post`` writes physical channels 0..95
with reversed output rows.  Flip, split and cat (flow.py:20,52,70) therefore never touch
memory, and the coupling arithmetic (flow.py:58-67) is the epilogue of the ``post`` conv.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ... import native
from ...engine import PackedConv, PlainConvParams, PlanCache, SpeakerTables, require_device
from .wn import WN


class Flip(nn.Module):
    def forward(self, x, *args, **kwargs):
        return torch.flip(x, [1])


class CouplingPlan:
    def __init__(self, layer: "ResidualCouplingLayer", device, flipped: bool):
        self.flipped = flipped
        self.half = layer.half_channels
        pre_w = layer.pre.effective_weight().to(device)
        post_w = layer.post.effective_weight().to(device)
        post_b = layer.post.bias.detach().to(device)
        if flipped:
            pre_w = torch.flip(pre_w, [1])
            post_w = torch.flip(post_w, [0])
            post_b = torch.flip(post_b, [0])
        self.pre = PackedConv(pre_w, layer.pre.bias, device)
        self.post = PackedConv(post_w, post_b, device)
        self.wn = layer.wn.plan(device)

    def run(self, phys: torch.Tensor, mask2, cond, cond_offset, reverse: bool, scratch, src0=None, src1=None):
        """Updates the [B, 2*half, T] buffer ``phys``: the half this layer transforms is written
        there, the other half is left as it is.  ``src0`` / ``src1``: where the conditioning half
        and the transformed half are read from, when not from ``phys`` itself (the first two layers
        of a pass read the caller's tensor, so that it is never copied)."""
        h = self.half
        lo, hi = slice(0, h), slice(h, 2 * h)
        s0 = phys if src0 is None else src0
        s1 = phys if src1 is None else src1
        x0 = s0[:, hi] if self.flipped else s0[:, lo]
        x1_in = s1[:, lo] if self.flipped else s1[:, hi]
        x1 = phys[:, lo] if self.flipped else phys[:, hi]
        hid, wn_out = scratch
        native.conv1d(x0, self.pre.w, hid, c_in=h, c_out=self.pre.rows, taps=1, dilation=1, pad=0,
                      tile=self.pre.tile, epilogue=native.EPI_STORE, bias=self.pre.bias, mask=mask2)
        self.wn.run(hid, mask2, cond, cond_offset, out=wn_out)
        native.conv1d(wn_out, self.post.w, x1, c_in=wn_out.shape[1], c_out=h, taps=1, dilation=1, pad=0,
                      tile=self.post.tile,
                      epilogue=native.EPI_COUPLE_REV if reverse else native.EPI_COUPLE_FWD,
                      bias=self.post.bias, res=x1_in, mask=mask2)


class ResidualCouplingLayer(nn.Module):
    def __init__(self, in_channels=192, phoneme_embedding_dim=192, speaker_id_embedding_dim=0,
                 kernel_size=5, dilation_rate=1, n_resblocks=4):
        assert in_channels % 2 == 0, "in_channels should be divisible by 2"
        super().__init__()
        self.in_channels = in_channels
        self.half_channels = in_channels // 2
        self.phoneme_embedding_dim = phoneme_embedding_dim
        self.kernel_size = kernel_size
        self.dilation_rate = dilation_rate
        self.n_resblocks = n_resblocks
        self.pre = PlainConvParams(self.half_channels, phoneme_embedding_dim, 1)
        self.wn = WN(phoneme_embedding_dim, kernel_size, dilation_rate, n_resblocks,
                     speaker_id_embedding_dim=speaker_id_embedding_dim)
        self.post = PlainConvParams(phoneme_embedding_dim, self.half_channels, 1)
        with torch.no_grad():                      # the reference zero-initialises post (flow.py:47-48)
            self.post.weight.zero_()
            self.post.bias.zero_()
        self._plans = PlanCache(self, lambda dev: (CouplingPlan(self, dev, False), CouplingPlan(self, dev, True)))

    def plan(self, device, flipped: bool) -> CouplingPlan:
        return self._plans.get(device)[1 if flipped else 0]

    @torch.no_grad()
    def forward(self, x, x_mask, speaker_id_embedded, reverse):
        require_device(x, "ResidualCouplingLayer.forward")
        plan = self.plan(x.device, False)
        B, C, T = x.shape
        phys = x.to(torch.float32).clone(memory_format=torch.contiguous_format)
        mask2 = x_mask.to(torch.float32).expand(B, 1, T).reshape(B, T).contiguous()
        cond = plan.wn.conditioning(speaker_id_embedded, B)
        hid = torch.empty(B, self.phoneme_embedding_dim, T, device=x.device, dtype=torch.float32)
        plan.run(phys, mask2, cond, 0, bool(reverse), (hid, torch.empty_like(hid)))
        return phys


class Flow(nn.Module):
    def __init__(self, speaker_id_embedding_dim=256, in_z_channels=192, phoneme_embedding_dim=192,
                 n_flows=4, kernel_size=5, dilation_rate=1, n_resblocks=4):
        super().__init__()
        self.speaker_id_embedding_dim = speaker_id_embedding_dim
        self.in_z_channels = in_z_channels
        self.phoneme_embedding_dim = phoneme_embedding_dim
        self.n_flows = n_flows
        self.kernel_size = kernel_size
        self.dilation_rate = dilation_rate
        self.n_resblocks = n_resblocks
        self.flows = nn.ModuleList()
        for _ in range(n_flows):
            self.flows.append(ResidualCouplingLayer(in_channels=in_z_channels,
                                                    phoneme_embedding_dim=phoneme_embedding_dim,
                                                    speaker_id_embedding_dim=speaker_id_embedding_dim,
                                                    kernel_size=kernel_size, dilation_rate=dilation_rate,
                                                    n_resblocks=n_resblocks))
            self.flows.append(Flip())
        self._plans = PlanCache(self, self._build)

    def _build(self, device):
        layers = [self.flows[2 * f] for f in range(self.n_flows)]
        # one GEMV for the conditioning of every coupling layer's WN stack
        cond_w = torch.cat([l.wn.plan(device).cond_w for l in layers], 0).contiguous()
        cond_b = torch.cat([l.wn.plan(device).cond_b for l in layers], 0).contiguous()
        return SpeakerTables(cond_w, cond_b)

    @torch.no_grad()
    def forward(self, z, z_mask, speaker_id_embedded, reverse=False):
        require_device(z, "Flow.forward")
        dev = z.device
        speakers = self._plans.get(dev)
        B, C, T = z.shape
        src = z if (z.dtype == torch.float32 and z.stride(2) == 1) else z.to(torch.float32).contiguous()
        mask2 = z_mask.to(torch.float32).expand(B, 1, T).reshape(B, T).contiguous()
        cond = speakers.conditioning(speaker_id_embedded, B)       # every coupling layer's WN stack at once
        per_layer = speakers.cond_w.shape[0] // self.n_flows
        hid = torch.empty(B, self.phoneme_embedding_dim, T, device=dev, dtype=torch.float32)
        scratch = (hid, torch.empty_like(hid))
        # forward: layer f runs behind f flips; reverse: behind the n - f flips that precede
        # it in the reversed module list.  After the whole pass n flips have been applied.
        # The input is never copied: the first layer of the pass reads both halves from it and
        # writes the half it transforms into a fresh buffer; the second layer's conditioning half
        # is that one, and the half it transforms is the input's other half, still untouched.
        n = self.n_flows
        order = list(range(n - 1, -1, -1) if reverse else range(n))
        if n < 2:
            phys = src.clone(memory_format=torch.contiguous_format)
            src = None
        else:
            phys = torch.empty(B, C, T, device=dev, dtype=torch.float32)
        for step, f in enumerate(order):
            flips = (n - f) if reverse else f
            layer = self.flows[2 * f].plan(dev, flipped=(flips % 2 == 1))
            layer.run(phys, mask2, cond, f * per_layer, bool(reverse), scratch,
                      src0=src if step == 0 else None, src1=src if step < 2 else None)
        if self.n_flows % 2 == 1:
            phys = torch.flip(phys, [1])
        return phys
