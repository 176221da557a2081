"""Deterministic synthetic weights and inputs for the generator hot path.

No trained ``netG_cpu.pth`` exists for the reference (its README names only
placeholder paths), so every test, the smoke run and the benchmark use
random-init weights.  The numbers must be identical in the container where
the golden vectors are captured from the reference and on the GPU box where
they are replayed, so they come from a counter-based generator written here
(splitmix64 -> uniform -> Box-Muller in float64 -> float32) that is keyed by
``(seed, state_dict key)`` and does not depend on torch's or numpy's RNG
streams.

The state_dict layout reproduced here is the one the reference modules
create (old-style ``nn.utils.weight_norm``: ``bias, weight_g, weight_v``):
  decoder            /root/reference/module/model_component/decoder.py:24-44,65-108
  flow               /root/reference/module/model_component/flow.py:24-48,76-105
  WN                 /root/reference/module/model_component/wn.py:27-59
  posterior encoder  /root/reference/module/model_component/posterior_encoder.py:19-43
  speaker embedding  /root/reference/module/vits_generator.py:88-91
"""
from __future__ import annotations

import hashlib
from collections import OrderedDict

import numpy as np
import torch

_MASK64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """One splitmix64 output step applied elementwise to uint64 counters."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK64
        z = z ^ (z >> np.uint64(31))
    return z


def _key_base(seed: int, key: str) -> np.uint64:
    h = hashlib.blake2b(f"{seed}:{key}".encode(), digest_size=8).digest()
    return np.uint64(int.from_bytes(h, "little"))


def uniform(seed: int, key: str, shape) -> np.ndarray:
    """float64 uniforms in the open interval (0, 1)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        ctr = (_key_base(seed, key) + np.arange(n, dtype=np.uint64)) & _MASK64
    bits = _splitmix64(ctr) >> np.uint64(11)            # 53 random bits
    u = (bits.astype(np.float64) + 0.5) / float(1 << 53)
    return u.reshape(shape)


def normal(seed: int, key: str, shape) -> np.ndarray:
    """float64 standard normals (Box-Muller on two independent uniform streams)."""
    u1 = uniform(seed, key + "#u1", shape)
    u2 = uniform(seed, key + "#u2", shape)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def normal_f32(seed: int, key: str, shape, scale: float = 1.0) -> torch.Tensor:
    return torch.from_numpy((normal(seed, key, shape) * scale).astype(np.float32))


# --------------------------------------------------------------------------------------
# state_dict construction
# --------------------------------------------------------------------------------------

def _plain_conv(sd, seed, name, cout, cin, k, bias=True, w_scale=None, b_scale=0.05):
    fan_in = cin * k
    ws = (1.0 / np.sqrt(fan_in)) if w_scale is None else w_scale
    sd[name + ".weight"] = normal_f32(seed, name + ".weight", (cout, cin, k), ws)
    if bias:
        sd[name + ".bias"] = normal_f32(seed, name + ".bias", (cout,), b_scale)


def _wn_conv(sd, seed, name, d0, d1, k, fan_in, gain, b_len, b_scale=0.05):
    """Weight-normed conv: keys in the order the reference registers them
    (bias, weight_g, weight_v).  ``d0`` is the normalised axis: out-channels for
    Conv1d, in-channels for ConvTranspose1d."""
    v = normal(seed, name + ".weight_v", (d0, d1, k)) / np.sqrt(fan_in)
    norm = np.sqrt((v * v).sum(axis=(1, 2), keepdims=True))
    g = norm * (0.5 + uniform(seed, name + ".weight_g", (d0, 1, 1))) * gain
    sd[name + ".bias"] = normal_f32(seed, name + ".bias", (b_len,), b_scale)
    sd[name + ".weight_g"] = torch.from_numpy(g.astype(np.float32))
    sd[name + ".weight_v"] = torch.from_numpy(v.astype(np.float32))


def _wn_stack(sd, seed, prefix, hidden, n_layers, gdim, k=5):
    _wn_conv(sd, seed, prefix + ".condition_layer", 2 * hidden * n_layers, gdim, 1,
             fan_in=gdim, gain=1.0, b_len=2 * hidden * n_layers)
    for i in range(n_layers):
        _wn_conv(sd, seed, f"{prefix}.in_resblocks.{i}", 2 * hidden, hidden, k,
                 fan_in=hidden * k, gain=1.0, b_len=2 * hidden)
    for i in range(n_layers):
        cout = 2 * hidden if i < n_layers - 1 else hidden
        _wn_conv(sd, seed, f"{prefix}.res_skip_layers.{i}", cout, hidden, 1,
                 fan_in=hidden, gain=0.7, b_len=cout)


def _ordered_wn(sd, prefix, n_layers):
    """Key order of a reference WN: in_resblocks.*, res_skip_layers.*, condition_layer."""
    keys = []
    for i in range(n_layers):
        keys += [f"{prefix}.in_resblocks.{i}.{s}" for s in ("bias", "weight_g", "weight_v")]
    for i in range(n_layers):
        keys += [f"{prefix}.res_skip_layers.{i}.{s}" for s in ("bias", "weight_g", "weight_v")]
    keys += [f"{prefix}.condition_layer.{s}" for s in ("bias", "weight_g", "weight_v")]
    return keys


def decoder_state_dict(seed: int = 999, prefix: str = "decoder", gdim: int = 256, zc: int = 192,
                       c0: int = 512, strides=(8, 8, 2, 2), kernels=(16, 16, 4, 4),
                       rb_kernels=(3, 7, 11)) -> "OrderedDict[str, torch.Tensor]":
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    _plain_conv(sd, seed, f"{prefix}.conv1d_pre", c0, zc, 7)
    _plain_conv(sd, seed, f"{prefix}.cond", c0, gdim, 1)
    for i, (s, k) in enumerate(zip(strides, kernels)):
        cin, cout = c0 >> i, c0 >> (i + 1)
        # each output sample sees k/s taps of every input channel
        _wn_conv(sd, seed, f"{prefix}.ups.{i}", cin, cout, k, fan_in=cout * k,
                 gain=np.sqrt(cout * k / (cin * (k // s))), b_len=cout)
    r = 0
    for i in range(len(strides)):
        ch = c0 >> (i + 1)
        for k in rb_kernels:
            for grp in ("convs1", "convs2"):
                for j in range(3):
                    _wn_conv(sd, seed, f"{prefix}.resblocks.{r}.{grp}.{j}", ch, ch, k,
                             fan_in=ch * k, gain=0.6, b_len=ch)
            r += 1
    _plain_conv(sd, seed, f"{prefix}.conv1d_post", 1, c0 >> len(strides), 7, bias=False,
                w_scale=0.35 / np.sqrt((c0 >> len(strides)) * 7))
    return sd


def flow_state_dict(seed: int = 999, prefix: str = "flow", gdim: int = 256, zc: int = 192,
                    hidden: int = 192, n_flows: int = 4, n_layers: int = 4,
                    live_post: bool = True) -> "OrderedDict[str, torch.Tensor]":
    """``live_post`` gives ``post`` non-zero weights: the reference zero-initialises it
    (flow.py:47-48), which would make the WN stacks dead code under test."""
    tmp: dict = {}
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    half = zc // 2
    for f in range(n_flows):
        p = f"{prefix}.flows.{2 * f}"
        _plain_conv(tmp, seed, p + ".pre", hidden, half, 1)
        _wn_stack(tmp, seed, p + ".wn", hidden, n_layers, gdim)
        if live_post:
            _plain_conv(tmp, seed, p + ".post", half, hidden, 1, w_scale=0.05, b_scale=0.05)
        else:
            tmp[p + ".post.weight"] = torch.zeros(half, hidden, 1)
            tmp[p + ".post.bias"] = torch.zeros(half)
        for k in ([p + ".pre.weight", p + ".pre.bias"] + _ordered_wn(tmp, p + ".wn", n_layers)
                  + [p + ".post.weight", p + ".post.bias"]):
            out[k] = tmp[k]
    return out


def posterior_encoder_state_dict(seed: int = 999, prefix: str = "posterior_encoder",
                                 gdim: int = 256, spec: int = 513, zc: int = 192,
                                 hidden: int = 192, n_layers: int = 16):
    tmp: dict = {}
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    _plain_conv(tmp, seed, prefix + ".preprocess", hidden, spec, 1)
    _wn_stack(tmp, seed, prefix + ".wn", hidden, n_layers, gdim)
    _plain_conv(tmp, seed, prefix + ".projection", 2 * zc, hidden, 1, w_scale=0.3 / np.sqrt(hidden))
    for k in ([prefix + ".preprocess.weight", prefix + ".preprocess.bias"]
              + _ordered_wn(tmp, prefix + ".wn", n_layers)
              + [prefix + ".projection.weight", prefix + ".projection.bias"]):
        out[k] = tmp[k]
    return out


def speaker_embedding_state_dict(seed: int = 999, n_speakers: int = 100, gdim: int = 256):
    return OrderedDict([("speaker_embedding.weight",
                         normal_f32(seed, "speaker_embedding.weight", (n_speakers, gdim)))])


def hot_path_state_dict(seed: int = 999, n_speakers: int = 100, live_post: bool = True,
                        with_posterior: bool = True):
    """Every key the hot path reads, named as in ``VitsGenerator.state_dict()``."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    sd.update(speaker_embedding_state_dict(seed, n_speakers))
    if with_posterior:
        sd.update(posterior_encoder_state_dict(seed))
    sd.update(decoder_state_dict(seed))
    sd.update(flow_state_dict(seed, live_post=live_post))
    return sd


def front_half_state_dict(template, seed: int = 999):
    """Deterministic values for every key of ``template`` (a state_dict of the text encoder /
    duration predictor, or of a whole generator) that the hot-path generators above do not
    cover.  Scales are chosen by parameter role so that attention, splines and durations
    are all exercised (the reference zero-initialises the spline projections)."""
    hot = ("decoder.", "flow.", "posterior_encoder.", "speaker_embedding.")
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, v in template.items():
        if k.startswith(hot):
            continue
        shape = tuple(v.shape)
        leaf = k.split(".")[-1]
        parent = k.split(".")[-2] if "." in k else ""
        is_norm = parent.isdigit() and "norm" in k.split(".")[-3]
        if is_norm and leaf == "weight":
            t = 1.0 + 0.1 * normal(seed, k, shape)
        elif is_norm and leaf == "bias":
            t = 0.05 * normal(seed, k, shape)
        elif leaf in ("m", "logs"):
            t = 0.1 * normal(seed, k, shape)
        elif leaf.startswith("emb_rel") or k.endswith("emb.weight"):
            t = normal(seed, k, shape) * shape[-1] ** -0.5
        elif leaf == "bias":
            t = 0.05 * normal(seed, k, shape)
        elif leaf == "weight" and len(shape) == 3:
            fan_in = shape[1] * shape[2]
            gain = 3.0 if (parent == "proj" and "flows" in k) else (0.3 if k.endswith("text_encoder.projection.weight") else 1.0)
            t = gain * normal(seed, k, shape) / np.sqrt(fan_in)
        else:
            raise KeyError(f"no generation rule for {k} {shape}")
        out[k] = torch.from_numpy(np.asarray(t, dtype=np.float32))
    return out


def full_state_dict(template, seed: int = 999):
    """Complete generator state_dict (858 keys for n_phoneme=40, n_speakers=100) in the
    template's key order."""
    hot = hot_path_state_dict(seed, n_speakers=template["speaker_embedding.weight"].shape[0])
    front = front_half_state_dict(template, seed)
    return OrderedDict((k, hot[k] if k in hot else front[k]) for k in template.keys())


# phoneme ids of "k o r e w a t e s U t o o N s e e d e s U" interleaved with blanks (id 0), the
# sentence of the reference's text-to-speech script (vits_text_to_speech.py:77-97); the symbol
# table is not needed for random-init weights, any fixed ids below n_phoneme=40 do.
TTS_SENTENCE_IDS = [0, 19, 0, 25, 0, 28, 0, 12, 0, 36, 0, 7, 0, 31, 0, 12, 0, 29, 0, 5, 0, 31, 0, 25, 0, 25, 0, 3,
                    0, 29, 0, 12, 0, 12, 0, 11, 0, 12, 0, 29, 0, 5, 0]


def strip_prefix(sd, prefix: str):
    """Sub-dict for one sub-module, keys relative to it (for ``load_state_dict``)."""
    p = prefix + "."
    return OrderedDict((k[len(p):], v) for k, v in sd.items() if k.startswith(p))


# --------------------------------------------------------------------------------------
# inputs
# --------------------------------------------------------------------------------------

def latent(seed: int, key: str, batch: int, frames: int, channels: int = 192) -> torch.Tensor:
    return normal_f32(seed, "input:" + key, (batch, channels, frames))


def speaker_ids(seed: int, key: str, batch: int, n_speakers: int = 100) -> torch.Tensor:
    u = uniform(seed, "spk:" + key, (batch,))
    return torch.from_numpy(np.minimum((u * n_speakers).astype(np.int64), n_speakers - 1))


def length_mask(lengths, frames: int) -> torch.Tensor:
    """[B,1,T] float 0/1 mask, as the reference builds it
    (/root/reference/module/vits_generator.py:39-43,177)."""
    lengths = torch.as_tensor(lengths, dtype=torch.int64)
    return (torch.arange(frames).unsqueeze(0) < lengths.unsqueeze(1)).unsqueeze(1).to(torch.float32)


def power_spectrogram(seed: int, key: str, batch: int, frames: int, bins: int = 513) -> torch.Tensor:
    """Non-negative chi-square-like values standing in for a power spectrogram."""
    n = normal(seed, "spec:" + key, (batch, bins, frames))
    return torch.from_numpy((n * n).astype(np.float32))
