"""Stochastic duration predictor, inference (reverse) direction.

Upstream of the hot path; kept so that ``VitsGenerator`` loads a complete ``netG_cpu.pth``
and ``text_to_speech`` runs end to end.  On the GPU the reverse pass is five launches of the
token kernels (csrc/token_kernels.hip: the conditioning stack, then one launch per coupling
flow with its spline); on CPU tensors, or with ``use_token_kernels = False``, it is the
layer-by-layer stock PyTorch form below.  Same constructor arguments and ``state_dict`` keys
as /root/reference/module/model_component/stochastic_duration_predictor.py
(``StochasticDurationPredictor`` :325-415, ``ConvFlow`` :230-274, ``DDSConv`` :285-323,
``ElementwiseAffine`` :213-228, ``Flip`` :276-283, ``Log`` :203-211, splines :20-201);
the training-only parameters (``post_*``) are held but unused.  ``forward`` implements
``reverse=True`` (:406-415) and raises for the training direction.

The rational-quadratic spline is evaluated densely for every element and merged with the
identity tails by ``torch.where``, instead of the reference's boolean-mask gather/scatter,
so the GPU path has no data-dependent shapes or host synchronisation.
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import native
from ...engine import PlanCache, SpeakerRef, SpeakerTables

MIN_BIN_WIDTH = 1e-3
MIN_BIN_HEIGHT = 1e-3
MIN_DERIVATIVE = 1e-3


def _knots(unnormalized, lo, hi, min_size):
    """Softmax-normalised bin sizes -> (edges [.., n+1], sizes [.., n]) spanning [lo, hi]."""
    n = unnormalized.shape[-1]
    sizes = min_size + (1 - min_size * n) * F.softmax(unnormalized, dim=-1)
    edges = F.pad(torch.cumsum(sizes, dim=-1), (1, 0))
    edges = (hi - lo) * edges + lo
    edges = torch.cat([torch.full_like(edges[..., :1], lo), edges[..., 1:-1], torch.full_like(edges[..., :1], hi)], -1)
    return edges, edges[..., 1:] - edges[..., :-1]


def inverse_rational_quadratic_spline(y, unnorm_widths, unnorm_heights, unnorm_derivs, tail_bound):
    """Inverse of the monotone rational-quadratic spline on [-tail_bound, tail_bound] with
    identity (linear) tails: returns x with spline(x) = y."""
    inside = (y >= -tail_bound) & (y <= tail_bound)
    yc = y.clamp(-tail_bound, tail_bound)
    edge_d = math.log(math.exp(1 - MIN_DERIVATIVE) - 1)       # boundary derivative = 1
    derivs = F.pad(unnorm_derivs, (1, 1), value=edge_d)
    derivs = MIN_DERIVATIVE + F.softplus(derivs)
    xk, w = _knots(unnorm_widths, -tail_bound, tail_bound, MIN_BIN_WIDTH)
    yk, h = _knots(unnorm_heights, -tail_bound, tail_bound, MIN_BIN_HEIGHT)
    n = w.shape[-1]
    search = yk.clone()
    search[..., -1] += 1e-6
    idx = ((yc.unsqueeze(-1) >= search).sum(-1) - 1).clamp(0, n - 1).unsqueeze(-1)
    pick = lambda t: t.gather(-1, idx)[..., 0]
    x0, bw, y0, bh = pick(xk), pick(w), pick(yk), pick(h)
    delta = pick(h / w)
    d0, d1 = pick(derivs), pick(derivs[..., 1:])
    dy = yc - y0
    s = d0 + d1 - 2 * delta
    a = dy * s + bh * (delta - d0)
    b = bh * d0 - dy * s
    c = -delta * dy
    root = (2 * c) / (-b - torch.sqrt(b * b - 4 * a * c))
    return torch.where(inside, root * bw + x0, y)


class Log(nn.Module):
    def forward(self, x, x_mask, reverse=False, **kwargs):
        if not reverse:
            raise NotImplementedError("training direction")
        return torch.exp(x) * x_mask


class ElementwiseAffine(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.m = nn.Parameter(torch.zeros(channels, 1))
        self.logs = nn.Parameter(torch.zeros(channels, 1))

    def forward(self, x, x_mask, reverse=False, **kwargs):
        if not reverse:
            raise NotImplementedError("training direction")
        return (x - self.m) * torch.exp(-self.logs) * x_mask


class Flip(nn.Module):
    def forward(self, x, *args, reverse=False, **kwargs):
        if not reverse:
            raise NotImplementedError("training direction")
        return torch.flip(x, [1])


class DDSConv(nn.Module):
    """Dilated depth-separable conv stack: per layer depthwise k-tap (dilation k**i), LayerNorm,
    GELU, 1x1, LayerNorm, GELU, residual."""

    def __init__(self, channels, kernel_size, n_layers, p_dropout=0.):
        super().__init__()
        self.n_layers = n_layers
        self.convs_sep = nn.ModuleList()
        self.convs_1x1 = nn.ModuleList()
        self.norms_1 = nn.ModuleList()
        self.norms_2 = nn.ModuleList()
        for i in range(n_layers):
            dil = kernel_size ** i
            self.convs_sep.append(nn.Conv1d(channels, channels, kernel_size, groups=channels, dilation=dil,
                                            padding=(kernel_size * dil - dil) // 2))
            self.convs_1x1.append(nn.Conv1d(channels, channels, 1))
            self.norms_1.append(nn.LayerNorm(channels))
            self.norms_2.append(nn.LayerNorm(channels))

    def forward(self, x, x_mask, g=None):
        if g is not None:
            x = x + g
        for sep, pw, n1, n2 in zip(self.convs_sep, self.convs_1x1, self.norms_1, self.norms_2):
            y = F.gelu(n1(sep(x * x_mask).transpose(1, -1)).transpose(1, -1))
            y = F.gelu(n2(pw(y).transpose(1, -1)).transpose(1, -1))
            x = x + y
        return x * x_mask


class ConvFlow(nn.Module):
    def __init__(self, in_channels, filter_channels, kernel_size, n_layers, num_bins=10, tail_bound=5.0):
        super().__init__()
        self.filter_channels, self.num_bins, self.tail_bound = filter_channels, num_bins, tail_bound
        self.half_channels = in_channels // 2
        self.pre = nn.Conv1d(self.half_channels, filter_channels, 1)
        self.convs = DDSConv(filter_channels, kernel_size, n_layers, p_dropout=0.)
        self.proj = nn.Conv1d(filter_channels, self.half_channels * (num_bins * 3 - 1), 1)
        with torch.no_grad():
            self.proj.weight.zero_()
            self.proj.bias.zero_()

    def forward(self, x, x_mask, g=None, reverse=False):
        if not reverse:
            raise NotImplementedError("training direction")
        x0, x1 = torch.split(x, [self.half_channels] * 2, 1)
        h = self.proj(self.convs(self.pre(x0), x_mask, g=g)) * x_mask
        b, c, t = x0.shape
        h = h.reshape(b, c, -1, t).permute(0, 1, 3, 2)                      # [b, c, t, 3*bins - 1]
        nb, scale = self.num_bins, math.sqrt(self.filter_channels)
        x1 = inverse_rational_quadratic_spline(x1, h[..., :nb] / scale, h[..., nb:2 * nb] / scale,
                                               h[..., 2 * nb:], self.tail_bound)
        return torch.cat([x0, x1], 1) * x_mask


def _f32(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def _pack_dds(dds: DDSConv, device) -> torch.Tensor:
    """A DDSConv's parameters in the order the token kernels read them (1x1 weights input-major)."""
    parts = []
    for sep, pw, n1, n2 in zip(dds.convs_sep, dds.convs_1x1, dds.norms_1, dds.norms_2):
        parts += [sep.weight.reshape(-1), sep.bias, n1.weight, n1.bias,
                  pw.weight[:, :, 0].t().reshape(-1), pw.bias, n2.weight, n2.bias]
    return _f32(torch.cat([p.detach().reshape(-1).float() for p in parts]), device)


class SdpPlan:
    """The reverse pass's weights on one device, laid out for the token kernels, and its launch list."""

    def __init__(self, sdp: "StochasticDurationPredictor", device):
        self.cond_w = _f32(sdp.cond.weight[:, :, 0], device)
        self.cond_b = _f32(sdp.cond.bias, device)
        self.speakers = SpeakerTables(self.cond_w, self.cond_b)
        self.pre_wt = _f32(sdp.pre.weight[:, :, 0].t(), device)
        self.pre_b = _f32(sdp.pre.bias, device)
        self.proj_wt = _f32(sdp.proj.weight[:, :, 0].t(), device)
        self.proj_b = _f32(sdp.proj.bias, device)
        self.layers = _pack_dds(sdp.convs, device)
        self.n_layers = sdp.convs.n_layers
        self.taps = sdp.convs.convs_sep[0].kernel_size[0]
        # The reversed module list with the Flips folded away: a Flip only renames the two
        # channels, so each coupling launch is told which stored channel conditions.
        order = list(reversed(sdp.flows))
        order = order[:-2] + [order[-1]]
        flipped = 0
        self.couplings = []
        self.affine = None
        for flow in order:
            if isinstance(flow, Flip):
                flipped ^= 1
            elif isinstance(flow, ConvFlow):
                if self.affine is not None:
                    raise NotImplementedError("a coupling flow after the elementwise affine")
                self.couplings.append(dict(
                    pre_w=_f32(flow.pre.weight.reshape(-1), device), pre_b=_f32(flow.pre.bias, device),
                    layers=_pack_dds(flow.convs, device), n_layers=flow.convs.n_layers,
                    taps=flow.convs.convs_sep[0].kernel_size[0],
                    proj_wt=_f32(flow.proj.weight[:, :, 0].t(), device), proj_b=_f32(flow.proj.bias, device),
                    pass_channel=flipped, num_bins=flow.num_bins, tail_bound=flow.tail_bound))
            elif isinstance(flow, ElementwiseAffine):
                idx = [flipped, 1 - flipped]                   # stored channel c holds logical channel c ^ flipped
                inv = [idx.index(0), idx.index(1)]
                self.affine = (_f32(flow.m.reshape(-1)[inv], device), _f32(flow.logs.reshape(-1)[inv], device))
            else:
                raise NotImplementedError(type(flow).__name__)
        self.out_channel = flipped                              # stored channel that holds logical channel 0


class StochasticDurationPredictor(nn.Module):
    # VITS_AMD_TOKEN_KERNELS=0 (or the attribute on an instance) keeps the GPU pass in stock PyTorch
    use_token_kernels = os.environ.get("VITS_AMD_TOKEN_KERNELS", "1") != "0"

    def __init__(self, speaker_id_embedding_dim, phoneme_embedding_dim=192, filter_channels=192,
                 kernel_size=3, p_dropout=0.5, n_flows=4):
        super().__init__()
        filter_channels = phoneme_embedding_dim
        self.n_flows = n_flows
        self.log_flow = Log()
        self.flows = nn.ModuleList([ElementwiseAffine(2)])
        for _ in range(n_flows):
            self.flows.append(ConvFlow(2, filter_channels, kernel_size, n_layers=3))
            self.flows.append(Flip())
        self.post_pre = nn.Conv1d(1, filter_channels, 1)
        self.post_proj = nn.Conv1d(filter_channels, filter_channels, 1)
        self.post_convs = DDSConv(filter_channels, kernel_size, n_layers=3, p_dropout=p_dropout)
        self.post_flows = nn.ModuleList([ElementwiseAffine(2)])
        for _ in range(4):
            self.post_flows.append(ConvFlow(2, filter_channels, kernel_size, n_layers=3))
            self.post_flows.append(Flip())
        self.pre = nn.Conv1d(phoneme_embedding_dim, filter_channels, 1)
        self.proj = nn.Conv1d(filter_channels, filter_channels, 1)
        self.convs = DDSConv(filter_channels, kernel_size, n_layers=3, p_dropout=p_dropout)
        self.cond = nn.Conv1d(speaker_id_embedding_dim, filter_channels, 1)
        self._plans = PlanCache(self, lambda dev: SdpPlan(self, dev))

    def _kernels_apply(self) -> bool:
        """The token kernels are built for the reference's sizes: 192 channels, 3-tap stacks of up
        to three layers, 10 spline bins; anything else takes the layer-by-layer form."""
        if not self.use_token_kernels or len(self.flows) <= 3:
            return False
        stacks = [self.convs] + [f.convs for f in self.flows if isinstance(f, ConvFlow)]
        return (self.pre.in_channels == native.DDS_CHANNELS and self.pre.out_channels == native.DDS_CHANNELS
                and all(d.n_layers <= 3 and d.convs_sep[0].kernel_size[0] == 3 for d in stacks)
                and all(f.num_bins == 10 and f.half_channels == 1 for f in self.flows if isinstance(f, ConvFlow)))

    def _reverse_on_token_kernels(self, text_encoded, text_mask, speaker_id_embedded, noise_scale, noise):
        plan = self._plans.get(text_encoded.device)
        B, H, T = text_encoded.shape
        dev = text_encoded.device
        x = text_encoded.to(torch.float32).contiguous()
        mask2 = text_mask.to(torch.float32).expand(B, 1, T).reshape(B, T).contiguous()
        add_vec = add_index = None
        if speaker_id_embedded is not None:
            cond = plan.speakers.conditioning(speaker_id_embedded, B)
            add_vec, add_index = cond.rows, cond.index
        g = torch.empty(B, H, T, device=dev, dtype=torch.float32)
        native.dds_condition(x, plan.pre_wt, plan.pre_b, add_vec, plan.layers, plan.n_layers, plan.taps,
                             plan.proj_wt, plan.proj_b, mask2, g, add_index=add_index)
        z = noise.to(device=dev, dtype=torch.float32).contiguous()
        bufs = [torch.empty_like(z), torch.empty_like(z)]
        for i, c in enumerate(plan.couplings):
            last = i == len(plan.couplings) - 1
            native.dds_coupling(z, c["pre_w"], c["pre_b"], g, c["layers"], c["n_layers"], c["taps"], c["proj_wt"],
                                c["proj_b"], mask2, bufs[i & 1], pass_channel=c["pass_channel"],
                                num_bins=c["num_bins"], tail_bound=c["tail_bound"],
                                in_scale=float(noise_scale) if i == 0 else 1.0,
                                affine=plan.affine if last else None)
            z = bufs[i & 1]
        return z[:, plan.out_channel:plan.out_channel + 1]

    @torch.no_grad()
    def forward(self, text_encoded, text_mask, duration_of_each_phoneme=None, speaker_id_embedded=None,
                reverse=False, noise_scale=1.0, noise=None):
        """``reverse=True`` only: returns log-durations [B, 1, T_text].  ``noise`` ([B, 2, T_text],
        standard normal) replaces the internal ``torch.randn`` draw."""
        if not reverse:
            raise NotImplementedError("the training direction of the duration predictor is out of scope")
        if text_encoded.is_cuda and self._kernels_apply():
            if noise is None:
                noise = torch.randn(text_encoded.size(0), 2, text_encoded.size(2)).to(
                    device=text_encoded.device, dtype=text_encoded.dtype)
            return self._reverse_on_token_kernels(text_encoded, text_mask, speaker_id_embedded, noise_scale, noise)
        x = self.pre(text_encoded)
        if isinstance(speaker_id_embedded, SpeakerRef):
            speaker_id_embedded = speaker_id_embedded.embedded()
        if speaker_id_embedded is not None:
            x = x + self.cond(speaker_id_embedded)
        x = self.proj(self.convs(x, text_mask)) * text_mask
        if noise is None:
            noise = torch.randn(x.size(0), 2, x.size(2)).to(device=x.device, dtype=x.dtype)
        z = noise * noise_scale
        # module list [affine, (flow, flip) x n]; walked backwards, skipping the first ConvFlow
        order = list(reversed(self.flows))
        order = order[:-2] + [order[-1]]
        for flow in order:
            z = flow(z, text_mask, g=x, reverse=True)
        return z[:, :1]
