"""Text encoder (phoneme ids -> per-token prior statistics), inference only.

Upstream of the hot path; kept so that ``VitsGenerator`` loads a complete ``netG_cpu.pth``
and ``text_to_speech`` runs end to end.  On the GPU a layer is seven launches of the library's
kernels (the projections and the 3-tap feed-forward convolutions on the fp32 MFMA convolution
kernel, attention and the channel LayerNorm on the token kernels of csrc/token_kernels.hip);
on CPU tensors, with ``use_token_kernels = False`` or for sequences beyond the attention
kernel's limit it is the stock PyTorch form below.  Same constructor arguments, ``forward`` signature,
return values and ``state_dict`` keys as /root/reference/module/model_component/
text_encoder.py (``TextEncoder`` :248-297, ``Encoder`` :211-245, ``MultiHeadAttention``
:21-174, ``FeedForwardNetwork`` :177-208).

The windowed relative-position attention is written with explicit index tables instead of
the reference's pad-and-reshape skewing: for query i and key j the learned key/value
embedding at offset j - i is used when |j - i| <= window and nothing otherwise.
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import native, packing
from ...engine import PlanCache


class MultiHeadAttention(nn.Module):
    def __init__(self, channels, out_channels, n_heads, p_dropout=0., window_size=None, **unused):
        super().__init__()
        assert channels % n_heads == 0
        self.channels, self.out_channels, self.n_heads = channels, out_channels, n_heads
        self.window_size = window_size
        self.k_channels = channels // n_heads
        self.conv_q = nn.Conv1d(channels, channels, 1)
        self.conv_k = nn.Conv1d(channels, channels, 1)
        self.conv_v = nn.Conv1d(channels, channels, 1)
        self.conv_o = nn.Conv1d(channels, out_channels, 1)
        if window_size is not None:
            std = self.k_channels ** -0.5
            self.emb_rel_k = nn.Parameter(torch.randn(1, 2 * window_size + 1, self.k_channels) * std)
            self.emb_rel_v = nn.Parameter(torch.randn(1, 2 * window_size + 1, self.k_channels) * std)

    def _offset_table(self, t, device):
        """idx[i, j] = (j - i) + window clipped into the table, ok[i, j] = |j - i| <= window."""
        pos = torch.arange(t, device=device)
        rel = pos.unsqueeze(0) - pos.unsqueeze(1)
        ok = rel.abs() <= self.window_size
        return (rel + self.window_size).clamp(0, 2 * self.window_size), ok

    def forward(self, x, c, attn_mask=None):
        b, _, t = x.shape
        h, dk = self.n_heads, self.k_channels
        q = self.conv_q(x).view(b, h, dk, t).transpose(2, 3) / math.sqrt(dk)
        k = self.conv_k(c).view(b, h, dk, t).transpose(2, 3)
        v = self.conv_v(c).view(b, h, dk, t).transpose(2, 3)
        scores = q @ k.transpose(-2, -1)                                   # [b, h, t, t]
        if self.window_size is not None:
            idx, ok = self._offset_table(t, x.device)
            rel_logits = q @ self.emb_rel_k.unsqueeze(0).transpose(-2, -1)  # [b, h, t, 2w+1]
            gathered = rel_logits.gather(-1, idx.expand(b, h, t, t))
            scores = scores + gathered * ok
        if attn_mask is not None:
            scores = scores.masked_fill(attn_mask == 0, -1e4)
        p = F.softmax(scores, dim=-1)
        out = p @ v
        if self.window_size is not None:
            # weights by relative offset: w_rel[i, r] = p[i, i + r - window] inside the sequence
            w_rel = torch.zeros(b, h, t, 2 * self.window_size + 1, dtype=p.dtype, device=p.device)
            w_rel.scatter_add_(-1, idx.expand(b, h, t, t), p * ok)
            out = out + w_rel @ self.emb_rel_v.unsqueeze(0)
        out = out.transpose(2, 3).contiguous().view(b, h * dk, t)
        return self.conv_o(out)


class FeedForwardNetwork(nn.Module):
    def __init__(self, in_channels, out_channels, filter_channels, kernel_size, p_dropout=0., **unused):
        super().__init__()
        self.kernel_size = kernel_size
        self.conv_1 = nn.Conv1d(in_channels, filter_channels, kernel_size)
        self.conv_2 = nn.Conv1d(filter_channels, out_channels, kernel_size)

    def _pad(self, x):
        if self.kernel_size == 1:
            return x
        return F.pad(x, ((self.kernel_size - 1) // 2, self.kernel_size // 2))

    def forward(self, x, x_mask):
        x = torch.relu(self.conv_1(self._pad(x * x_mask)))
        return self.conv_2(self._pad(x * x_mask)) * x_mask


class Encoder(nn.Module):
    def __init__(self, phoneme_embedding_dim, filter_channels, n_heads, n_layers, kernel_size=1,
                 p_dropout=0., window_size=4, **unused):
        super().__init__()
        self.n_layers = n_layers
        d = phoneme_embedding_dim
        self.attention_layers = nn.ModuleList(
            MultiHeadAttention(d, d, n_heads, p_dropout=p_dropout, window_size=window_size) for _ in range(n_layers))
        self.norm_layers_1 = nn.ModuleList(nn.LayerNorm(d) for _ in range(n_layers))
        self.ffn_layers = nn.ModuleList(
            FeedForwardNetwork(d, d, filter_channels, kernel_size, p_dropout=p_dropout) for _ in range(n_layers))
        self.norm_layers_2 = nn.ModuleList(nn.LayerNorm(d) for _ in range(n_layers))

    def forward(self, x, x_mask):
        attn_mask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
        x = x * x_mask
        for att, n1, ffn, n2 in zip(self.attention_layers, self.norm_layers_1, self.ffn_layers, self.norm_layers_2):
            x = n1((x + att(x, x, attn_mask)).transpose(1, -1)).transpose(1, -1)
            x = n2((x + ffn(x, x_mask)).transpose(1, -1)).transpose(1, -1)
        return x * x_mask


class _F32Conv:
    """A convolution of the encoder packed for the fp32 MFMA kernel, whatever precision the hot
    path runs in: the durations derived from this encoder are rounded up to whole frames."""

    def __init__(self, weight, bias, device):
        w = weight.detach().to(device=device, dtype=torch.float32)
        self.rows, self.c_in, self.taps = w.shape
        self.pad = (self.taps - 1) // 2
        from ...engine import choose_tile
        self.tile = choose_tile(self.rows)
        ti = native.tile_info(self.tile)
        self.w = packing.pack_conv_weight(w, ti.ck, ti.rows_per_block)
        self.bias = bias.detach().to(device=device, dtype=torch.float32).contiguous()

    def __call__(self, x, out, *, mask=None, res=None, relu_in=False):
        """out = conv(x) + bias (* mask), or conv(relu?(x)) + bias + res."""
        if res is None and not relu_in:
            native.conv1d(x, self.w, out, c_in=self.c_in, c_out=self.rows, taps=self.taps, dilation=1, pad=self.pad,
                          tile=self.tile, epilogue=native.EPI_STORE, bias=self.bias, mask=mask)
        else:
            # the input activation is a leaky ReLU: slope 0 is the ReLU, slope 1 the identity
            native.conv1d(x, self.w, out, c_in=self.c_in, c_out=self.rows, taps=self.taps, dilation=1, pad=self.pad,
                          tile=self.tile, epilogue=native.EPI_RESID, bias=self.bias, res=res, act_in=True,
                          in_slope=0.0 if relu_in else 1.0)
        return out


class TextEncoderPlan:
    """The encoder's weights on one device in the layouts the kernels read."""

    def __init__(self, te: "TextEncoder", device):
        f32 = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()
        self.layers = []
        for att, n1, ffn, n2 in zip(te.encoder.attention_layers, te.encoder.norm_layers_1, te.encoder.ffn_layers,
                                    te.encoder.norm_layers_2):
            self.layers.append(dict(
                qkv=_F32Conv(torch.cat([att.conv_q.weight, att.conv_k.weight, att.conv_v.weight]),
                             torch.cat([att.conv_q.bias, att.conv_k.bias, att.conv_v.bias]), device),
                out=_F32Conv(att.conv_o.weight, att.conv_o.bias, device),
                emb_k=f32(att.emb_rel_k[0]), emb_v=f32(att.emb_rel_v[0]),
                heads=att.n_heads, window=att.window_size,
                n1=(f32(n1.weight), f32(n1.bias)), n2=(f32(n2.weight), f32(n2.bias)),
                ffn1=_F32Conv(ffn.conv_1.weight, ffn.conv_1.bias, device),
                ffn2=_F32Conv(ffn.conv_2.weight, ffn.conv_2.bias, device)))
        self.projection = _F32Conv(te.projection.weight, te.projection.bias, device)


class TextEncoder(nn.Module):
    # VITS_AMD_TOKEN_KERNELS=0 (or the attribute on an instance) keeps the GPU pass in stock PyTorch
    use_token_kernels = os.environ.get("VITS_AMD_TOKEN_KERNELS", "1") != "0"

    def __init__(self, n_phoneme, phoneme_embedding_dim=192, out_channels=192, n_heads=2, n_layers=6,
                 kernel_size=3, filter_channels=768, p_dropout=0.1):
        super().__init__()
        self.n_phoneme = n_phoneme
        self.phoneme_embedding_dim = phoneme_embedding_dim
        self.out_channels = out_channels
        self.emb = nn.Embedding(n_phoneme, phoneme_embedding_dim)
        nn.init.normal_(self.emb.weight, 0.0, phoneme_embedding_dim ** -0.5)
        self.encoder = Encoder(phoneme_embedding_dim, filter_channels, n_heads, n_layers, kernel_size, p_dropout)
        self.projection = nn.Conv1d(phoneme_embedding_dim, out_channels * 2, 1)
        self._plans = PlanCache(self, lambda dev: TextEncoderPlan(self, dev))

    def _kernels_apply(self) -> bool:
        att = self.encoder.attention_layers[0]
        ffn = self.encoder.ffn_layers[0]
        return (self.use_token_kernels and self.phoneme_embedding_dim == native.DDS_CHANNELS
                and att.k_channels == 96 and att.window_size == 4 and ffn.kernel_size % 2 == 1
                and ffn.conv_1.out_channels % 64 == 0)

    def _forward_on_kernels(self, x, text_mask):
        """Encoder layers and the projection on the library's kernels.  Padded columns are kept at
        zero wherever a layer's input is read (they never reach a valid column: keys are masked
        in the attention and the convolutions see masked inputs), and the outputs are masked."""
        plan = self._plans.get(x.device)
        B, C, T = x.shape
        mask2 = text_mask.reshape(B, T).contiguous()
        x = (x * text_mask).contiguous()
        new = lambda rows: torch.empty(B, rows, T, device=x.device, dtype=torch.float32)
        for L in plan.layers:
            qkv = L["qkv"](x, new(3 * C))
            att = new(C)
            native.rel_attention(qkv, L["emb_k"], L["emb_v"], mask2, att, heads=L["heads"], window=L["window"])
            y = L["out"](att, new(C), res=x)
            x1 = new(C)
            native.channel_norm(y, L["n1"][0], L["n1"][1], mask2, x1)
            hid = L["ffn1"](x1, new(L["ffn1"].rows), mask=mask2)
            y2 = L["ffn2"](hid, new(C), res=x1, relu_in=True)
            x = new(C)
            native.channel_norm(y2, L["n2"][0], L["n2"][1], mask2, x)
        stats = plan.projection(x, new(plan.projection.rows), mask=mask2)
        return x, stats

    @torch.no_grad()
    def forward(self, text_padded, text_lengths):
        x = (self.emb(text_padded) * math.sqrt(self.phoneme_embedding_dim)).transpose(1, -1)
        steps = torch.arange(x.size(2), dtype=text_lengths.dtype, device=text_lengths.device)
        text_mask = (steps.unsqueeze(0) < text_lengths.unsqueeze(1)).unsqueeze(1).to(x.dtype)
        if x.is_cuda and x.size(2) <= native.ATT_MAX_LENGTH and self._kernels_apply():
            text_encoded, stats = self._forward_on_kernels(x.to(torch.float32), text_mask.to(torch.float32))
            m, logs = torch.split(stats, self.out_channels, dim=1)
            return text_encoded, m, logs, text_mask
        text_encoded = self.encoder(x * text_mask, text_mask)
        stats = self.projection(text_encoded) * text_mask
        m, logs = torch.split(stats, self.out_channels, dim=1)
        return text_encoded, m, logs, text_mask
