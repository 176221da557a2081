"""Posterior encoder: linear spectrogram -> latent z (voice-conversion front end).

Drop-in for /root/reference/module/model_component/posterior_encoder.py:18-60: same
constructor arguments, ``forward(spectrogram, spectrogram_lengths, speaker_id_embedded)``
returning ``(z, mean, log_std, mask)``, same ``state_dict`` keys (``preprocess``, ``wn``,
``projection``).

preprocess (1x1, 513 -> H) * mask and the projection (1x1, H -> 2*out) * mask run on the
MFMA convolution kernel; the projection's epilogue also draws the sample
``z = (m + eps * exp(logs)) * mask`` (posterior_encoder.py:56-59).  The reference draws
``eps`` with ``torch.randn_like``; ``forward`` does the same unless ``noise`` is given,
which is how tests pin it.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ... import native
from ...engine import PackedConv, PlainConvParams, PlanCache, require_device
from ...packing import pair_row_order
from .wn import WN


class PosteriorPlan:
    def __init__(self, pe: "PosteriorEncoder", device):
        self.pre = PackedConv(pe.preprocess.effective_weight(), pe.preprocess.bias, device)
        zc = pe.out_z_channels
        self.proj = PackedConv(pe.projection.effective_weight(), pe.projection.bias, device,
                               tile=native.TILE_PAIR, row_order=pair_row_order(zc).to(device), rows=2 * zc)
        self.wn = pe.wn.plan(device)


class PosteriorEncoder(nn.Module):
    def __init__(self, speaker_id_embedding_dim, in_spec_channels=513, out_z_channels=192,
                 phoneme_embedding_dim=192, kernel_size=5, dilation_rate=1, n_resblocks=16):
        super().__init__()
        if out_z_channels % 32:
            raise ValueError("out_z_channels must be a multiple of 32 for the MFMA row tiles")
        self.speaker_id_embedding_dim = speaker_id_embedding_dim
        self.in_spec_channels = in_spec_channels
        self.out_z_channels = out_z_channels
        self.phoneme_embedding_dim = phoneme_embedding_dim
        self.kernel_size = kernel_size
        self.dilation_rate = dilation_rate
        self.n_resblocks = n_resblocks
        self.preprocess = PlainConvParams(in_spec_channels, phoneme_embedding_dim, 1)
        self.wn = WN(phoneme_embedding_dim, kernel_size, dilation_rate, n_resblocks,
                     speaker_id_embedding_dim=speaker_id_embedding_dim)
        self.projection = PlainConvParams(phoneme_embedding_dim, out_z_channels * 2, 1)
        self._plans = PlanCache(self, lambda dev: PosteriorPlan(self, dev))

    @torch.no_grad()
    def forward(self, spectrogram, spectrogram_lengths, speaker_id_embedded, noise=None):
        require_device(spectrogram, "PosteriorEncoder.forward")
        dev = spectrogram.device
        plan = self._plans.get(dev)
        spec = spectrogram.to(torch.float32).contiguous()
        B, _, T = spec.shape
        steps = torch.arange(T, dtype=spectrogram_lengths.dtype, device=spectrogram_lengths.device)
        mask = (steps.unsqueeze(0) < spectrogram_lengths.unsqueeze(1)).unsqueeze(1).to(spec.dtype).to(dev)
        mask2 = mask.reshape(B, T).contiguous()
        H, zc = self.phoneme_embedding_dim, self.out_z_channels

        x = torch.empty(B, H, T, device=dev, dtype=torch.float32)
        native.conv1d(spec, plan.pre.w, x, c_in=self.in_spec_channels, c_out=H, taps=1, dilation=1, pad=0,
                      tile=plan.pre.tile, epilogue=native.EPI_STORE, bias=plan.pre.bias, mask=mask2)
        cond = plan.wn.conditioning(speaker_id_embedded, B)
        h = plan.wn.run(x, mask2, cond)

        if noise is None:
            noise = torch.randn(B, zc, T, device=dev, dtype=torch.float32)
        noise = noise.to(device=dev, dtype=torch.float32).contiguous()
        mean = torch.empty(B, zc, T, device=dev, dtype=torch.float32)
        logs = torch.empty_like(mean)
        z = torch.empty_like(mean)
        native.conv1d(h, plan.proj.w, mean, c_in=H, c_out=2 * zc, taps=1, dilation=1, pad=0,
                      tile=plan.proj.tile, epilogue=native.EPI_SAMPLE, bias=plan.proj.bias, out2=z, out3=logs,
                      aux=noise, mask=mask2, hidden=zc)
        return z, mean, logs, mask
