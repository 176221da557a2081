"""WN: the WaveNet-style gated stack used by the coupling layers and the posterior encoder.

Drop-in for /root/reference/module/model_component/wn.py (``WN`` :26-86,
``gated_activation_unit`` :15-22): same constructor arguments, same ``forward(x, x_mask,
speaker_id_embedded)`` and the same ``state_dict`` keys (``in_resblocks.{i}``,
``res_skip_layers.{i}``, ``condition_layer``, each ``bias / weight_g / weight_v``).

Per layer the reference runs a k-tap conv (H -> 2H), adds the layer's slice of the speaker
conditioning, gates tanh*sigmoid, runs a 1x1 conv (H -> 2H, last layer H) and splits the
result into a masked residual update and a skip accumulation.  For hidden = 192 that can be
one launch per layer (csrc/wn_layer.hpp, csrc/wn_layer_bf16.hpp: the gated activations stay in LDS
between the two convolutions; fp32 and bf16 matrix inputs).  Otherwise it is two launches of the convolution kernel: conv + conditioning
bias + gate in one epilogue, and 1x1 conv + residual/skip/mask in the other.  The conditioning 1x1 conv on the [B, G, 1]
speaker vector is one wavefront-reduced GEMV for all layers, with the in-conv biases
folded in.
"""
from __future__ import annotations

import torch
import torch.nn as nn

import os

from ... import native, packing
from ...engine import Conditioning, PackedConv, PlanCache, SpeakerTables, WeightNormConvParams, precision, require_device
from ...packing import pair_row_order


class WNPlan:
    """Packed weights of one WN stack on one device."""

    fused_max_columns = 12288
    # bf16 inputs: one launch per layer at every size (reverse flow, one launch / two launches per layer:
    # 1 x 258 frames 0.45 / 0.72 ms, 4 x 862 0.49 / 0.74, 16 x 862 0.61 / 0.94, 64 x 862 1.67 / 2.41);
    # the bf16 convolution kernel has no split channel loop for the two-launch form to win with
    fused_max_columns_bf16 = int(os.environ.get("VITS_AMD_WN_FUSED_MAX_BF16", 1 << 30))
    fused_min_columns_bf16 = int(os.environ.get("VITS_AMD_WN_FUSED_MIN_BF16", 0))
    fused_min_columns = 4096     # below: two launches per layer, each with its channel loop split over workgroups

    def __init__(self, wn: "WN", device):
        H, n = wn.hidden_channels, wn.n_resblocks
        order = pair_row_order(H).to(device)
        self.hidden, self.n_layers = H, n
        self.in_convs, self.rs_convs = [], []
        # one launch per layer where the fused kernel exists (VITS_AMD_WN_FUSION=0 disables it)
        prec = precision()
        self.bf16 = prec == native.PREC_BF16
        wi = native.wn_info(H) if (prec in (native.PREC_F32, native.PREC_BF16)
                                   and os.environ.get("VITS_AMD_WN_FUSION", "1") != "0") else None
        k = wn.kernel_size
        self.fused = wi is not None and all((k - 1) * wn.dilation_rate ** i <= 56 for i in range(n))
        self.fused_w = []
        cond_bias = wn.condition_layer.bias.detach().to(device).clone()
        for i in range(n):
            m = wn.in_resblocks[i]
            dil = wn.dilation_rate ** i
            self.in_convs.append(PackedConv(m.effective_weight(), None, device, tile=native.TILE_PAIR,
                                            row_order=order, dilation=dil, rows=2 * H))
            cond_bias[2 * H * i:2 * H * (i + 1)] += m.bias.detach().to(device)
            r = wn.res_skip_layers[i]
            self.rs_convs.append(PackedConv(r.effective_weight(), r.bias, device))
            if self.fused and self.bf16:
                self.fused_w.append((
                    packing.pack_conv_weight_bf16(m.effective_weight().to(device), wi.ck_bf16, wi.rows_per_block, order),
                    packing.pack_conv_weight_bf16(r.effective_weight().to(device), wi.ck_bf16, 32)))
            elif self.fused:
                self.fused_w.append((
                    packing.pack_conv_weight(m.effective_weight().to(device), wi.ck, wi.rows_per_block, order),
                    packing.pack_conv_weight(r.effective_weight().to(device), wi.ck, 32)))
        # conditioning GEMV: rows = 2H * n, in-conv biases folded into its bias
        self.cond_w = wn.condition_layer.effective_weight().to(device)[:, :, 0].contiguous()
        self.cond_b = cond_bias.contiguous()
        self.speakers = SpeakerTables(self.cond_w, self.cond_b)

    def conditioning(self, speaker, batch: int) -> Conditioning:
        """Speakers (a SpeakerRef, or embedded [B, G, 1]) -> per-(item, layer) gate biases, n_layers * 2H each."""
        return self.speakers.conditioning(speaker, batch)

    def run(self, x: torch.Tensor, mask2: torch.Tensor, cond: Conditioning, cond_offset: int = 0,
            out: torch.Tensor | None = None) -> torch.Tensor:
        """x [B,H,T] is consumed (updated in place); mask2 [B,T]; ``cond`` holds this stack's gate
        biases starting at column ``cond_offset``.  Returns the masked skip sum."""
        B, H, T = x.shape
        skip = torch.empty_like(x)
        if out is None:
            out = torch.empty_like(x)
        cstride = cond.stride
        # Measured on the MI355X (reverse flow, batch x 862 frames, one-launch layer / two launches):
        # 1: 1.37 / 0.87 ms, 4: 1.39 / 1.14, 8: 1.42 / 1.78, 16: 2.49 / 2.37.  A few thousand frames
        # give the one-launch layer a few dozen workgroups, each a serial loop over 960 channel taps,
        # while the two-launch form splits that loop over workgroups; in between the one-launch
        # layer's fewer dependent launches win; beyond, its 6-wave workgroups fill the SIMDs less evenly.
        fused_max = self.fused_max_columns_bf16 if self.bf16 else self.fused_max_columns
        fused_min = self.fused_min_columns_bf16 if self.bf16 else self.fused_min_columns
        if self.fused and (fused_min < B * T <= fused_max or not native.SPLIT_SMALL_LAUNCHES):
            other = torch.empty_like(x)             # layers ping-pong: a layer's output may not land on its input
            for i in range(self.n_layers):
                ic, rc = self.in_convs[i], self.rs_convs[i]
                w_in, w_rs = self.fused_w[i]
                last = i == self.n_layers - 1
                dst = out if last else other
                if last and out.data_ptr() == x.data_ptr():
                    raise native.NativeError("WN: the output buffer aliases the last layer's input")
                native.wn_layer(x, w_in, cond.columns(cond_offset + 2 * H * i), cstride, w_rs, rc.bias, dst, skip, mask2,
                                taps=ic.taps, dilation=ic.dilation, first=(i == 0), last=last, cond_index=cond.index)
                x, other = dst, x
            return out
        acts = torch.empty_like(x)
        for i in range(self.n_layers):
            ic, rc = self.in_convs[i], self.rs_convs[i]
            bias_view = cond.columns(cond_offset + 2 * H * i)
            native.conv1d(x, ic.w, acts, c_in=H, c_out=2 * H, taps=ic.taps, dilation=ic.dilation,
                          pad=ic.pad, tile=ic.tile, epilogue=native.EPI_GATE, bias=bias_view,
                          bias_bstride=cstride, bias_index=cond.index, hidden=H)
            if i < self.n_layers - 1:
                native.conv1d(acts, rc.w, x, c_in=H, c_out=2 * H, taps=1, dilation=1, pad=0, tile=rc.tile,
                              epilogue=native.EPI_WN_MID, bias=rc.bias, res=x, out2=skip, mask=mask2,
                              first=(i == 0), hidden=H)
            else:
                native.conv1d(acts, rc.w, out, c_in=H, c_out=H, taps=1, dilation=1, pad=0, tile=rc.tile,
                              epilogue=native.EPI_WN_LAST, bias=rc.bias, out2=skip, mask=mask2,
                              first=(i == 0), hidden=H)
        return out


class WN(nn.Module):
    def __init__(self, hidden_channels, kernel_size, dilation_rate, n_resblocks, speaker_id_embedding_dim):
        super().__init__()
        assert kernel_size % 2 == 1
        if hidden_channels % 32 != 0:
            raise ValueError("hidden_channels must be a multiple of 32 for the MFMA row tiles")
        self.hidden_channels = hidden_channels
        self.kernel_size = kernel_size
        self.dilation_rate = dilation_rate
        self.n_resblocks = n_resblocks
        self.speaker_id_embedding_dim = speaker_id_embedding_dim
        self.in_resblocks = nn.ModuleList(
            WeightNormConvParams(hidden_channels, 2 * hidden_channels, kernel_size) for _ in range(n_resblocks))
        self.res_skip_layers = nn.ModuleList(
            WeightNormConvParams(hidden_channels, 2 * hidden_channels if i < n_resblocks - 1 else hidden_channels, 1)
            for i in range(n_resblocks))
        self.condition_layer = WeightNormConvParams(speaker_id_embedding_dim, 2 * hidden_channels * n_resblocks, 1)
        self._plans = PlanCache(self, lambda dev: WNPlan(self, dev))

    def plan(self, device) -> WNPlan:
        return self._plans.get(device)

    @torch.no_grad()
    def forward(self, x, x_mask, speaker_id_embedded):
        require_device(x, "WN.forward")
        plan = self.plan(x.device)
        B, H, T = x.shape
        mask2 = x_mask.to(torch.float32).expand(B, 1, T).reshape(B, T).contiguous()
        cond = plan.conditioning(speaker_id_embedded, B)
        work = x.to(torch.float32).clone(memory_format=torch.contiguous_format)
        return plan.run(work, mask2, cond)
