"""CPU restatement of the reference generator's inference hot path.

TEST INFRASTRUCTURE ONLY.  This file is the checker for the HIP path: it may be
imported by ``tests/``, by ``__graft_entry__.smoke()`` and by the
``cpu_baseline`` leg of ``bench.py`` and by nothing else.  The product
(``vits_amd``) never imports it and never falls back to it.

It restates, as stateless fp32 functions over a ``state_dict``, the arithmetic
of these reference functions (paths relative to /root/reference):

  weight_norm (w = g * v / ||v||, norm over every axis but 0)   PyTorch semantics
                                                         used at decoder.py:27-42,96 and wn.py:42,50,58
  resblock                module/model_component/decoder.py:46-61
  decoder                 module/model_component/decoder.py:110-131
  gate / wn               module/model_component/wn.py:15-22, 61-86
  coupling                module/model_component/flow.py:50-71
  flow                    module/model_component/flow.py:18-21, 107-118
  posterior_encoder       module/model_component/posterior_encoder.py:45-60
  tts_tail                module/vits_generator.py:184-186
  voice_conversion        module/vits_generator.py:189-197

Parity pinning: the reference has no tests or fixtures of its own, so this
restatement is pinned by golden vectors captured from the imported reference
modules (tests/golden/make_golden.py, which also asserts equality here) and
replayed by tests/test_oracle_golden.py anywhere.  The convolution arithmetic
itself lives in PyTorch ATen (CPU: oneDNN), the same third-party dependency
the reference dispatches to.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

# model constants (module/vits_generator.py:70-77, decoder.py:66-72, flow.py:76-83)
UP_STRIDES = (8, 8, 2, 2)
UP_KERNELS = (16, 16, 4, 4)
RB_KERNELS = (3, 7, 11)
RB_DILATIONS = (1, 3, 5)
HIDDEN = 192
LRELU = 0.1


def weight_norm(sd, name: str) -> torch.Tensor:
    """Effective weight of an old-style weight-normed module."""
    v = sd[name + ".weight_v"]
    g = sd[name + ".weight_g"]
    # the ATen primitive the reference's weight-norm hook evaluates on every forward
    return torch._weight_norm(v, g, 0)


def resblock(x, sd, prefix: str, k: int):
    for j, d in enumerate(RB_DILATIONS):
        t = F.leaky_relu(x, LRELU)
        t = F.conv1d(t, weight_norm(sd, f"{prefix}.convs1.{j}"), sd[f"{prefix}.convs1.{j}.bias"],
                     padding=(k - 1) * d // 2, dilation=d)
        t = F.leaky_relu(t, LRELU)
        t = F.conv1d(t, weight_norm(sd, f"{prefix}.convs2.{j}"), sd[f"{prefix}.convs2.{j}.bias"],
                     padding=(k - 1) // 2)
        x = t + x
    return x


def decoder(z, g, sd, prefix: str = "decoder", taps: dict | None = None):
    """z [B,192,T], g [B,256,1] -> wav [B,1,256*T].  ``taps`` (optional dict)
    receives the per-stage intermediates for finer-grained comparisons."""
    x = F.conv1d(z, sd[f"{prefix}.conv1d_pre.weight"], sd[f"{prefix}.conv1d_pre.bias"], padding=3)
    x = x + F.conv1d(g, sd[f"{prefix}.cond.weight"], sd[f"{prefix}.cond.bias"])
    for i, (s, k) in enumerate(zip(UP_STRIDES, UP_KERNELS)):
        x = F.leaky_relu(x, LRELU)
        x = F.conv_transpose1d(x, weight_norm(sd, f"{prefix}.ups.{i}"), sd[f"{prefix}.ups.{i}.bias"],
                               stride=s, padding=(k - s) // 2)
        if taps is not None:
            taps[f"up{i}"] = x
        acc = None
        for j, rk in enumerate(RB_KERNELS):
            r = resblock(x, sd, f"{prefix}.resblocks.{i * len(RB_KERNELS) + j}", rk)
            acc = r if acc is None else acc + r
        x = acc / len(RB_KERNELS)
        if taps is not None:
            taps[f"mrf{i}"] = x
    x = F.leaky_relu(x)                      # default slope 0.01 (decoder.py:126)
    x = F.conv1d(x, sd[f"{prefix}.conv1d_post.weight"], None, padding=3)
    return torch.tanh(x)


def wn(x, mask, g, sd, prefix: str, n_layers: int, hidden: int = HIDDEN):
    """x [B,H,T], mask [B,1,T], g [B,256,1] -> [B,H,T]."""
    cond = F.conv1d(g, weight_norm(sd, f"{prefix}.condition_layer"),
                    sd[f"{prefix}.condition_layer.bias"])
    out = torch.zeros_like(x)
    for i in range(n_layers):
        w_in = weight_norm(sd, f"{prefix}.in_resblocks.{i}")
        k = w_in.shape[-1]
        a = F.conv1d(x, w_in, sd[f"{prefix}.in_resblocks.{i}.bias"], padding=(k - 1) // 2)
        a = a + cond[:, 2 * hidden * i:2 * hidden * (i + 1)]
        acts = torch.tanh(a[:, :hidden]) * torch.sigmoid(a[:, hidden:])
        rs = F.conv1d(acts, weight_norm(sd, f"{prefix}.res_skip_layers.{i}"),
                      sd[f"{prefix}.res_skip_layers.{i}.bias"])
        if i < n_layers - 1:
            x = (x + rs[:, :hidden]) * mask
            out = out + rs[:, hidden:]
        else:
            out = out + rs
    return out * mask


def coupling(x, mask, g, sd, prefix: str, reverse: bool, n_layers: int = 4):
    half = x.shape[1] // 2
    x0, x1 = x[:, :half], x[:, half:]
    h = F.conv1d(x0, sd[f"{prefix}.pre.weight"], sd[f"{prefix}.pre.bias"]) * mask
    h = wn(h, mask, g, sd, f"{prefix}.wn", n_layers)
    m = F.conv1d(h, sd[f"{prefix}.post.weight"], sd[f"{prefix}.post.bias"]) * mask
    x1 = (x1 - m) * mask if reverse else m + x1 * mask
    return torch.cat([x0, x1], 1)


def flow(z, mask, g, sd, prefix: str = "flow", reverse: bool = False, n_flows: int = 4):
    """The module list is [coupling, flip] * n_flows; reverse walks it backwards."""
    x = z
    if not reverse:
        for f in range(n_flows):
            x = coupling(x, mask, g, sd, f"{prefix}.flows.{2 * f}", False)
            x = torch.flip(x, [1])
    else:
        for f in reversed(range(n_flows)):
            x = torch.flip(x, [1])
            x = coupling(x, mask, g, sd, f"{prefix}.flows.{2 * f}", True)
    return x


def posterior_encoder(spec, lengths, g, sd, eps, prefix: str = "posterior_encoder",
                      n_layers: int = 16, zc: int = 192):
    """``eps`` replaces the reference's ``torch.randn_like`` draw (posterior_encoder.py:59)."""
    T = spec.shape[2]
    mask = (torch.arange(T, dtype=lengths.dtype).unsqueeze(0) < lengths.unsqueeze(1))
    mask = mask.unsqueeze(1).to(spec.dtype)
    x = F.conv1d(spec, sd[f"{prefix}.preprocess.weight"], sd[f"{prefix}.preprocess.bias"]) * mask
    x = wn(x, mask, g, sd, f"{prefix}.wn", n_layers)
    stats = F.conv1d(x, sd[f"{prefix}.projection.weight"], sd[f"{prefix}.projection.bias"]) * mask
    m, logs = stats[:, :zc], stats[:, zc:]
    z = (m + eps * torch.exp(logs)) * mask
    return z, m, logs, mask


def speaker_embed(ids, sd):
    return sd["speaker_embedding.weight"][ids].unsqueeze(-1)


def tts_tail(m_p, logs_p, mask, eps, g, sd, noise_scale: float = 0.667, max_len=None):
    """Everything ``text_to_speech`` does after the frame-rate prior is known."""
    z_p = m_p + eps * torch.exp(logs_p) * noise_scale
    z = flow(z_p, mask, g, sd, reverse=True)
    wav = decoder((z * mask)[:, :, :max_len], g, sd)
    return z, wav


def flow_reverse_decode(z_p, mask, g, sd):
    """The benchmarked path: reverse flow then decoder (vits_generator.py:185-186)."""
    z = flow(z_p, mask, g, sd, reverse=True)
    return decoder(z * mask, g, sd)


def voice_conversion(spec, lengths, src_ids, tgt_ids, sd, eps):
    g_src = speaker_embed(src_ids, sd)
    g_tgt = speaker_embed(tgt_ids, sd)
    z, _, _, mask = posterior_encoder(spec, lengths, g_src, sd, eps)
    z_p = flow(z, mask, g_src, sd, reverse=False)
    z_hat = flow(z_p, mask, g_tgt, sd, reverse=True)
    return decoder(z_hat * mask, g_tgt, sd)


def power_spectrogram(wav, filter_length: int = 1024, hop_length: int = 256, win_length: int = 1024):
    """Front end of voice conversion, /root/reference/vits_voice_converter.py:82-96: reflect
    padding by (filter_length - hop_length) / 2, un-centred frames every hop_length samples,
    periodic Hann window, one-sided DFT, squared magnitude.  Plain float64 numpy DFT per
    frame.  torchaudio (which the reference calls) is not installed, so this restates its
    documented ``spectrogram(power=2, normalized=False, center=False, pad=0)``: parity with
    torchaudio itself is unpinned."""
    import numpy as np
    x = wav.detach().cpu().to(torch.float64).numpy()
    lead = x.shape[:-1]
    x = x.reshape(-1, x.shape[-1])
    pad = (filter_length - hop_length) // 2
    x = np.pad(x, ((0, 0), (pad, pad)), mode="reflect")
    n = np.arange(win_length)
    window = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)          # torch.hann_window: periodic
    frames = 1 + (x.shape[1] - filter_length) // hop_length
    out = np.empty((x.shape[0], filter_length // 2 + 1, frames))
    for f in range(frames):
        seg = x[:, f * hop_length:f * hop_length + filter_length] * window
        out[:, :, f] = np.abs(np.fft.rfft(seg, n=filter_length, axis=1)) ** 2
    return torch.from_numpy(out.reshape(*lead, out.shape[1], frames)).to(torch.float32)
