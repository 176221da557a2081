"""Host-side weight preparation: weight-norm folding and MFMA fragment packing.

The reference recomputes ``w = g * v / ||v||`` in a forward pre-hook on every call
(old-style ``nn.utils.weight_norm``: decoder.py:27-42,96 and wn.py:42,50,58 under
/root/reference/module/model_component).  Here it is folded once when the weights are
bound, together with a re-layout into the order the convolution kernel's wavefronts
consume (csrc/conv_mfma.hpp):

    packed[mt][chunk][tap][g][lane][e] = W[row(32*mt + (lane & 31)), chunk*CK + 2*(4*g + e) + (lane >> 5), tap]

i.e. for each 32-row tile ``mt`` a linear stream of 64-lane x float4 groups; one group
feeds four consecutive k-steps (channel pairs) of v_mfma_f32_32x32x2_f32.  Rows are
zero-padded to the tile family's rows-per-workgroup and channels to its chunk size; spare
zero groups at the very end absorb the kernel's one-tap-ahead prefetch.
"""
from __future__ import annotations

import torch


def fold_weight_norm(v: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """w = g * v / ||v|| with the norm over every axis but 0 (ConvTranspose1d included:
    there axis 0 is the input channel)."""
    return torch._weight_norm(v, g, 0)


def pack_conv_weight(w: torch.Tensor, ck: int, rows_per_block: int, row_order=None) -> torch.Tensor:
    """w [rows, c_in, taps] -> flat fp32 tensor in fragment order.

    ``row_order``: optional 1-D index tensor giving, for each packed row, the source row
    of ``w`` (used to interleave paired row tiles)."""
    w = w.detach().to(torch.float32)
    if row_order is not None:
        w = w[row_order]
    rows, cin, taps = w.shape
    rpad = -(-rows // rows_per_block) * rows_per_block
    cpad = -(-cin // ck) * ck
    wp = torch.zeros(rpad, cpad, taps, dtype=torch.float32, device=w.device)
    wp[:rows, :cin] = w
    mt, nch, g = rpad // 32, cpad // ck, ck // 8
    # [mt, r, chunk, g, e, h, tap] -> [mt, chunk, tap, g, h, r, e]
    wp = wp.view(mt, 32, nch, g, 4, 2, taps).permute(0, 2, 6, 3, 5, 1, 4).contiguous()
    flat = wp.view(-1)
    return torch.cat([flat, torch.zeros(4 * 256, dtype=torch.float32, device=w.device)])


def pack_conv_weight_bf16(w: torch.Tensor, ck: int, rows_per_block: int, row_order=None) -> torch.Tensor:
    """w [rows, c_in, taps] -> flat bf16 tensor in the fragment order of the bf16-input kernels
    (csrc/conv_bf16.hpp):

        packed[mt][chunk][tap][kstep][lane][i] = bf16(W[row(32*mt + (lane & 31)), chunk*CK + 16*kstep + 8*(lane >> 5) + i, tap])

    one 16-byte fragment per lane per v_mfma_f32_32x32x16_bf16; rounding is to nearest even."""
    w = w.detach().to(torch.float32)
    if row_order is not None:
        w = w[row_order]
    rows, cin, taps = w.shape
    rpad = -(-rows // rows_per_block) * rows_per_block
    cpad = -(-cin // ck) * ck
    wp = torch.zeros(rpad, cpad, taps, dtype=torch.float32, device=w.device)
    wp[:rows, :cin] = w
    mt, nch, ks = rpad // 32, cpad // ck, ck // 16
    # [mt, r, chunk, kstep, h, i, tap] -> [mt, chunk, tap, kstep, h, r, i]
    wp = wp.view(mt, 32, nch, ks, 2, 8, taps).permute(0, 2, 6, 3, 4, 1, 5).contiguous()
    flat = wp.view(-1).to(torch.bfloat16)
    return torch.cat([flat, torch.zeros(8 * 512, dtype=torch.bfloat16, device=w.device)])


def split3(w: torch.Tensor):
    """Three-way split of fp32 values into bf16-representable slices with hi + mid + lo == w:
    each slice is what is left, rounded to nearest even in bf16 (the last one is exact)."""
    w = w.to(torch.float32)
    hi = w.to(torch.bfloat16).to(torch.float32)
    r1 = w - hi
    mid = r1.to(torch.bfloat16).to(torch.float32)
    lo = (r1 - mid).to(torch.bfloat16).to(torch.float32)
    return hi, mid, lo


def pack_conv_weight_split(w: torch.Tensor, ck: int, rows_per_block: int, row_order=None) -> torch.Tensor:
    """w [rows, c_in, taps] -> flat bf16 tensor for the sliced-fp32 kernels (csrc/conv_split.hpp):

        packed[mt][chunk][tap][kstep][slice][lane][i],  slice 0/1/2 = hi/mid/lo of
        W[row(32*mt + (lane & 31)), chunk*CK + 16*kstep + 8*(lane >> 5) + i, tap]

    The returned tensor carries ``vk_precision`` so the launcher picks the matching kernel."""
    from . import native
    w = w.detach().to(torch.float32)
    if row_order is not None:
        w = w[row_order]
    rows, cin, taps = w.shape
    rpad = -(-rows // rows_per_block) * rows_per_block
    cpad = -(-cin // ck) * ck
    wp = torch.zeros(rpad, cpad, taps, dtype=torch.float32, device=w.device)
    wp[:rows, :cin] = w
    mt, nch, ks = rpad // 32, cpad // ck, ck // 16
    slices = torch.stack(split3(wp), 0)                                     # [3, rpad, cpad, taps]
    # [s, mt, r, chunk, kstep, h, i, tap] -> [mt, chunk, tap, kstep, s, h, r, i]
    sp = slices.view(3, mt, 32, nch, ks, 2, 8, taps).permute(1, 3, 7, 4, 0, 5, 2, 6).contiguous()
    flat = sp.view(-1).to(torch.bfloat16)                                   # exact: slices have <= 8 significant bits
    out = torch.cat([flat, torch.zeros(3 * 8 * 512, dtype=torch.bfloat16, device=w.device)])
    out.vk_precision = native.PREC_F32_SPLIT
    return out


def pair_row_order(hidden: int) -> torch.Tensor:
    """Row order that interleaves 32-row tiles of the two halves of a 2*hidden-row output:
    [a0, b0, a1, b1, ...] with a_q = rows 32q.., b_q = rows hidden + 32q.."""
    assert hidden % 32 == 0
    idx = []
    for q in range(hidden // 32):
        idx += list(range(32 * q, 32 * q + 32))
        idx += list(range(hidden + 32 * q, hidden + 32 * q + 32))
    return torch.tensor(idx, dtype=torch.int64)


def polyphase_row_order(c_out: int, stride: int):
    """Packed-row order of the polyphase rows (phase * c_out + channel) the kernels expect, or None
    for the natural order.  For stride 8 a 32-row tile holds 8 channels x the 4 phases of one half
    of the phases (row-in-tile = 4 * channel + phase-in-half; tile = 2 * channel group + half), so
    that a lane's four consecutive accumulator registers are four consecutive output samples."""
    if stride != 8:
        return None
    assert c_out % 8 == 0
    idx = []
    for mt in range(c_out // 4):
        cg, half = mt >> 1, mt & 1
        for i in range(32):
            idx.append((4 * half + (i & 3)) * c_out + 8 * cg + (i >> 2))
    return torch.tensor(idx, dtype=torch.int64)


def transposed_to_polyphase(wt: torch.Tensor, stride: int) -> torch.Tensor:
    """ConvTranspose1d weight [c_in, c_out, 2*stride] (padding stride/2) -> the weight of an
    equivalent 2-tap convolution with stride*c_out output rows, row = phase*c_out + channel.

    Output sample o = stride*t + p receives, per input channel,
        p <  stride/2:  w[.., p + stride/2 + stride] * x[t-1] + w[.., p + stride/2] * x[t]
        p >= stride/2:  w[.., p + stride/2] * x[t]            + w[.., p - stride/2] * x[t+1]
    (from o = i*stride - stride/2 + k of nn.ConvTranspose1d, decoder.py:96-97), so tap j of
    phase p reads x[t + j - 1] for the lower phases and x[t + j] for the upper ones."""
    cin, cout, k = wt.shape
    assert k == 2 * stride and stride % 2 == 0
    hs = stride // 2
    out = torch.empty(stride, cout, cin, 2, dtype=wt.dtype, device=wt.device)
    for p in range(stride):
        if p < hs:
            k0, k1 = p + hs + stride, p + hs
        else:
            k0, k1 = p + hs, p - hs
        out[p, :, :, 0] = wt[:, :, k0].t()
        out[p, :, :, 1] = wt[:, :, k1].t()
    return out.reshape(stride * cout, cin, 2)
