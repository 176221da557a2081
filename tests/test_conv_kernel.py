"""Kernel-level GPU tests through the C ABI: the fused convolution against plain PyTorch fp32
ops on the CPU, over odd shapes (lengths below one tile, not multiples of 4, channel counts
that are not multiples of the K-chunk), every tile family and every epilogue."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from vits_amd import native, packing
from vits_amd.engine import choose_tile

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 2e-5


@pytest.fixture(params=["wide", "narrow"], autouse=True)
def time_tile_width(request):
    """Every kernel test runs with the families' full time tiles and with their half-width
    variants (which the launcher otherwise picks by grid size)."""
    saved = native.NARROW_BELOW_BLOCKS
    native.NARROW_BELOW_BLOCKS = 0 if request.param == "wide" else 1 << 30
    yield request.param
    native.NARROW_BELOW_BLOCKS = saved


def _rand(*shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


def _close(got, ref, what):
    got = got.cpu()
    assert torch.isfinite(got).all(), what
    err = (got - ref).abs().max().item()
    assert err <= TOL * max(1.0, ref.abs().max().item()), f"{what}: {err:.3e}"


def _pack(w, tile, row_order=None):
    ti = native.tile_info(tile)
    return packing.pack_conv_weight(w, ti.ck, ti.rows_per_block, row_order).to(DEV)


@pytest.mark.parametrize("cin,cout,k,dil,B,L,act", [
    (32, 32, 3, 1, 1, 1, True),          # a single sample: everything is halo
    (32, 32, 11, 5, 2, 37, True),        # halo (50) longer than the signal
    (64, 64, 7, 3, 3, 517, True),        # odd length: scalar staging
    (64, 64, 7, 3, 2, 520, True),        # aligned length: 16-byte staging
    (128, 128, 3, 5, 2, 1000, True),
    (256, 256, 11, 1, 1, 260, True),     # two row blocks
    (192, 512, 7, 1, 2, 45, False),      # conv_pre shape, per-item bias
    (96, 192, 1, 1, 2, 333, False),      # c_in = 3 chunks of 32
    (513, 192, 1, 1, 1, 50, False),      # c_in not a multiple of the chunk
    (40, 72, 5, 2, 2, 130, True),        # neither channel count is a multiple of 32
])
def test_conv_store_and_residual(cin, cout, k, dil, B, L, act):
    x = _rand(B, cin, L, seed=1)
    w = _rand(cout, cin, k, seed=2) / (cin * k) ** 0.5
    bias = _rand(B, cout, seed=3)
    res = _rand(B, cout, L, seed=4)
    pad = (k - 1) * dil // 2
    xin = F.leaky_relu(x, 0.1) if act else x
    ref = F.conv1d(xin, w, None, padding=pad, dilation=dil) + bias.unsqueeze(-1)
    tile = choose_tile(cout)
    wp = _pack(w, tile)
    out = torch.full((B, cout, L), float("nan"), device=DEV)
    native.conv1d(x.to(DEV), wp, out, c_in=cin, c_out=cout, taps=k, dilation=dil, pad=pad, tile=tile,
                  epilogue=native.EPI_STORE, bias=bias.to(DEV), bias_bstride=cout, act_in=act, in_slope=0.1)
    _close(out, ref, "store")
    if act:
        out2 = torch.full((B, cout, L), float("nan"), device=DEV)
        native.conv1d(x.to(DEV), wp, out2, c_in=cin, c_out=cout, taps=k, dilation=dil, pad=pad, tile=tile,
                      epilogue=native.EPI_RESID, bias=bias.to(DEV), bias_bstride=cout, res=res.to(DEV),
                      act_in=True, in_slope=0.1)
        _close(out2, ref + res, "residual")
        accb = _rand(B, cout, L, seed=5)
        out3 = accb.to(DEV).clone()
        native.conv1d(x.to(DEV), wp, out3, c_in=cin, c_out=cout, taps=k, dilation=dil, pad=pad, tile=tile,
                      epilogue=native.EPI_RESID_AVG, bias=bias.to(DEV), bias_bstride=cout, res=res.to(DEV),
                      out2=out3, act_in=True, in_slope=0.1, navg=3.0)
        _close(out3, (accb + (ref + res)) / 3.0, "residual + accumulate + mean")


@pytest.mark.parametrize("cin,cout,s,B,L", [(512, 256, 8, 1, 9), (256, 128, 8, 2, 131), (128, 64, 2, 2, 700),
                                            (64, 32, 2, 1, 1030), (64, 32, 4, 2, 64)])
def test_polyphase_upsample(cin, cout, s, B, L):
    x = _rand(B, cin, L, seed=11)
    wt = _rand(cin, cout, 2 * s, seed=12) / (cin * 2) ** 0.5
    bias = _rand(cout, seed=13)
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1), wt, bias, stride=s, padding=s // 2)
    poly = packing.transposed_to_polyphase(wt, s)
    tile = choose_tile(poly.shape[0])
    out = torch.full((B, cout, L * s), float("nan"), device=DEV)
    native.conv1d(x.to(DEV), _pack(poly, tile, packing.polyphase_row_order(cout, s)), out, c_in=cin, c_out=cout, taps=2, dilation=1, pad=1, tile=tile,
                  epilogue=native.EPI_UPSAMPLE, bias=bias.to(DEV), act_in=True, in_slope=0.1, up_stride=s)
    _close(out, ref, "upsample")


@pytest.mark.parametrize("B,T,lens", [(2, 50, (50, 21)), (3, 129, (129, 1, 77)), (1, 300, (300,))])
def test_gate_wn_and_coupling_epilogues(B, T, lens):
    H = 192
    mask = (torch.arange(T).unsqueeze(0) < torch.tensor(lens).unsqueeze(1)).float()
    x = _rand(B, H, T, seed=21)
    w_in = _rand(2 * H, H, 5, seed=22) / (H * 5) ** 0.5
    cond = _rand(B, 2 * H, seed=23)
    a = F.conv1d(x, w_in, None, padding=2) + cond.unsqueeze(-1)
    acts_ref = torch.tanh(a[:, :H]) * torch.sigmoid(a[:, H:])
    acts = torch.full((B, H, T), float("nan"), device=DEV)
    native.conv1d(x.to(DEV), _pack(w_in, native.TILE_PAIR, packing.pair_row_order(H)), acts, c_in=H, c_out=2 * H,
                  taps=5, dilation=1, pad=2, tile=native.TILE_PAIR, epilogue=native.EPI_GATE, bias=cond.to(DEV),
                  bias_bstride=2 * H, hidden=H)
    _close(acts, acts_ref, "gate")

    w_rs = _rand(2 * H, H, 1, seed=24) / H ** 0.5
    b_rs = _rand(2 * H, seed=25)
    rs = F.conv1d(acts_ref, w_rs, b_rs)
    skip0 = _rand(B, H, T, seed=26)
    xd, skipd = x.to(DEV).clone(), skip0.to(DEV).clone()
    tile = choose_tile(2 * H)
    native.conv1d(acts_ref.to(DEV), _pack(w_rs, tile), xd, c_in=H, c_out=2 * H, taps=1, dilation=1, pad=0, tile=tile,
                  epilogue=native.EPI_WN_MID, bias=b_rs.to(DEV), res=xd, out2=skipd, mask=mask.to(DEV),
                  first=False, hidden=H)
    _close(xd, (x + rs[:, :H]) * mask.unsqueeze(1), "WN residual half")
    _close(skipd, skip0 + rs[:, H:], "WN skip half")

    w_last = _rand(H, H, 1, seed=27) / H ** 0.5
    outd = torch.full((B, H, T), float("nan"), device=DEV)
    tile = choose_tile(H)
    native.conv1d(acts_ref.to(DEV), _pack(w_last, tile), outd, c_in=H, c_out=H, taps=1, dilation=1, pad=0, tile=tile,
                  epilogue=native.EPI_WN_LAST, bias=b_rs[:H].to(DEV), out2=skip0.to(DEV), mask=mask.to(DEV),
                  first=False, hidden=H)
    _close(outd, (skip0 + F.conv1d(acts_ref, w_last, b_rs[:H])) * mask.unsqueeze(1), "WN last layer")

    # coupling update written in place into a channel slice of a wider tensor
    half = 96
    w_post = _rand(half, H, 1, seed=28) * 0.05
    b_post = _rand(half, seed=29) * 0.05
    lat = _rand(B, 2 * half, T, seed=30)
    m = F.conv1d(acts_ref, w_post, b_post) * mask.unsqueeze(1)
    for rev in (True, False):
        phys = lat.to(DEV).clone()
        x1 = phys[:, half:]
        tile = choose_tile(half)
        native.conv1d(acts_ref.to(DEV), _pack(w_post, tile), x1, c_in=H, c_out=half, taps=1, dilation=1, pad=0,
                      tile=tile, epilogue=native.EPI_COUPLE_REV if rev else native.EPI_COUPLE_FWD,
                      bias=b_post.to(DEV), res=x1, mask=mask.to(DEV))
        want = (lat[:, half:] - m) * mask.unsqueeze(1) if rev else m + lat[:, half:] * mask.unsqueeze(1)
        _close(phys[:, half:], want, f"coupling reverse={rev}")
        assert torch.equal(phys[:, :half].cpu(), lat[:, :half])          # the other half is untouched


def test_speaker_linear_and_post_conv():
    g = _rand(5, 256, seed=41)
    w = _rand(777, 256, seed=42) / 16
    b = _rand(777, seed=43)
    out = torch.empty(5, 777, device=DEV)
    native.speaker_linear(g.to(DEV), w.to(DEV), b.to(DEV), out)
    _close(out, g @ w.t() + b, "speaker_linear")
    for L in (1, 7, 1024, 3001):
        x = _rand(2, 32, L, seed=44)
        wp = _rand(1, 32, 7, seed=45) * 0.1
        wav = torch.empty(2, 1, L, device=DEV)
        native.post_conv_tanh(x.to(DEV), wp[0].contiguous().to(DEV), wav, 0.01)
        _close(wav, torch.tanh(F.conv1d(F.leaky_relu(x, 0.01), wp, None, padding=3)), f"post conv L={L}")


def test_unsupported_arguments_are_rejected_on_the_gpu_box():
    x = torch.zeros(1, 32, 16, device=DEV)
    wp = _pack(torch.zeros(32, 32, 3), native.TILE_M32)
    out = torch.empty_like(x)
    with pytest.raises(native.NativeError):      # dilation * (taps - 1) beyond the LDS halo
        native.conv1d(x, wp, out, c_in=32, c_out=32, taps=3, dilation=40, pad=40, tile=native.TILE_M32,
                      epilogue=native.EPI_STORE)
    with pytest.raises(native.NativeError):      # residual epilogue without a residual
        native.conv1d(x, wp, out, c_in=32, c_out=32, taps=3, dilation=1, pad=1, tile=native.TILE_M32,
                      epilogue=native.EPI_RESID, act_in=True, in_slope=0.1)
    with pytest.raises(native.NativeError):      # slope outside [0, 1]
        native.conv1d(x, wp, out, c_in=32, c_out=32, taps=3, dilation=1, pad=1, tile=native.TILE_M32,
                      epilogue=native.EPI_STORE, act_in=True, in_slope=1.5)


@pytest.mark.parametrize("ch,k,dil,B,L", [
    (32, 3, 1, 2, 1),            # a single sample
    (32, 11, 5, 2, 37),          # halo longer than the signal
    (32, 7, 3, 1, 500),          # exactly one workgroup's worth of outputs
    (32, 7, 3, 2, 501),          # one column into the second workgroup, odd length
    (32, 3, 5, 2, 2000),         # aligned length: 16-byte staging, several workgroups
    (64, 11, 1, 1, 180),         # one 64-channel workgroup
    (64, 11, 5, 2, 463),
    (64, 3, 3, 3, 1200),
])
def test_fused_resblock_pair(ch, k, dil, B, L):
    x = _rand(B, ch, L, seed=51)
    w1 = _rand(ch, ch, k, seed=52) / (ch * k) ** 0.5
    w2 = _rand(ch, ch, k, seed=53) / (ch * k) ** 0.5
    b1, b2 = _rand(ch, seed=54), _rand(ch, seed=55)
    h = F.conv1d(F.leaky_relu(x, 0.1), w1, b1, padding=(k - 1) * dil // 2, dilation=dil)
    ref = F.conv1d(F.leaky_relu(h, 0.1), w2, b2, padding=(k - 1) // 2) + x
    pi = native.pair_info(ch)
    assert pi is not None
    p1 = packing.pack_conv_weight(w1, pi.ck, pi.rows_per_block).to(DEV)
    p2 = packing.pack_conv_weight(w2, pi.ck, pi.rows_per_block).to(DEV)
    xd = x.to(DEV)
    out = torch.full((B, ch, L), float("nan"), device=DEV)
    native.resblock_pair(xd, p1, b1.to(DEV), p2, b2.to(DEV), out, taps=k, dilation=dil,
                         epilogue=native.EPI_RESID, slope=0.1)
    _close(out, ref, "pair")
    accb = _rand(B, ch, L, seed=56)
    acc = accb.to(DEV).clone()
    native.resblock_pair(xd, p1, b1.to(DEV), p2, b2.to(DEV), acc, taps=k, dilation=dil,
                         epilogue=native.EPI_RESID_ACC, slope=0.1, acc=acc)
    _close(acc, accb + ref, "pair + accumulate")
    acc = accb.to(DEV).clone()
    native.resblock_pair(xd, p1, b1.to(DEV), p2, b2.to(DEV), acc, taps=k, dilation=dil,
                         epilogue=native.EPI_RESID_AVG, slope=0.1, acc=acc, navg=3.0)
    _close(acc, (accb + ref) / 3.0, "pair + accumulate + mean")
    # a channel-strided view of a wider tensor as input (rows not 16-byte aligned when L is odd)
    wide = torch.zeros(B, ch + 3, L, device=DEV)
    wide[:, 3:] = xd
    out2 = torch.full((B, ch, L), float("nan"), device=DEV)
    native.resblock_pair(wide[:, 3:], p1, b1.to(DEV), p2, b2.to(DEV), out2, taps=k, dilation=dil,
                         epilogue=native.EPI_RESID, slope=0.1)
    _close(out2, ref, "pair on a strided view")


def test_fused_pair_rejects_bad_arguments():
    pi = native.pair_info(32)
    x = torch.zeros(1, 32, 64, device=DEV)
    wp = packing.pack_conv_weight(torch.zeros(32, 32, 3), pi.ck, pi.rows_per_block).to(DEV)
    b = torch.zeros(32, device=DEV)
    with pytest.raises(native.NativeError):      # output on top of the input
        native.resblock_pair(x, wp, b, wp, b, x, taps=3, dilation=1, epilogue=native.EPI_RESID, slope=0.1)
    with pytest.raises(native.NativeError):      # halo too long
        native.resblock_pair(x, wp, b, wp, b, torch.empty_like(x), taps=3, dilation=30,
                             epilogue=native.EPI_RESID, slope=0.1)
    assert native.pair_info(128) is None and native.pair_info(512, native.PREC_BF16) is None
    x128 = torch.zeros(1, 128, 64, device=DEV)
    with pytest.raises(native.NativeError):      # width without a fused kernel
        native.resblock_pair(x128, wp, b, wp, b, torch.empty_like(x128), taps=3, dilation=1,
                             epilogue=native.EPI_RESID, slope=0.1)


# ---- bf16 matrix-core inputs: the kernel must equal an fp32 convolution of the bf16-rounded
# ---- operands (products of bf16 values are exact in fp32; only the summation order differs)
def _r16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _pack16(w, tile, row_order=None):
    ti = native.tile_info(tile)
    return packing.pack_conv_weight_bf16(w, ti.ck_bf16, ti.rows_per_block, row_order).to(DEV)


@pytest.mark.parametrize("cin,cout,k,dil,B,L,act", [
    (32, 32, 3, 1, 1, 1, True),
    (32, 32, 11, 5, 2, 37, True),
    (64, 64, 7, 3, 3, 517, True),
    (128, 128, 3, 5, 2, 1000, True),
    (256, 256, 11, 1, 1, 260, True),
    (192, 512, 7, 1, 2, 45, False),
    (513, 192, 1, 1, 1, 50, False),      # c_in not a multiple of the chunk
    (40, 72, 5, 2, 2, 130, True),
])
def test_bf16_conv_store_and_residual(cin, cout, k, dil, B, L, act):
    x = _rand(B, cin, L, seed=61)
    w = _rand(cout, cin, k, seed=62) / (cin * k) ** 0.5
    bias = _rand(B, cout, seed=63)
    res = _rand(B, cout, L, seed=64)
    pad = (k - 1) * dil // 2
    xin = _r16(F.leaky_relu(x, 0.1) if act else x)
    ref = F.conv1d(xin, _r16(w), None, padding=pad, dilation=dil) + bias.unsqueeze(-1)
    tile = choose_tile(cout)
    wp = _pack16(w, tile)
    out = torch.full((B, cout, L), float("nan"), device=DEV)
    native.conv1d(x.to(DEV), wp, out, c_in=cin, c_out=cout, taps=k, dilation=dil, pad=pad, tile=tile,
                  epilogue=native.EPI_STORE, bias=bias.to(DEV), bias_bstride=cout, act_in=act, in_slope=0.1)
    _close(out, ref, "bf16 store")
    if act:
        accb = _rand(B, cout, L, seed=65)
        out3 = accb.to(DEV).clone()
        native.conv1d(x.to(DEV), wp, out3, c_in=cin, c_out=cout, taps=k, dilation=dil, pad=pad, tile=tile,
                      epilogue=native.EPI_RESID_AVG, bias=bias.to(DEV), bias_bstride=cout, res=res.to(DEV),
                      out2=out3, act_in=True, in_slope=0.1, navg=3.0)
        _close(out3, (accb + (ref + res)) / 3.0, "bf16 residual + accumulate + mean")


@pytest.mark.parametrize("cin,cout,s,B,L", [(512, 256, 8, 1, 9), (128, 64, 2, 2, 700), (64, 32, 2, 1, 1030)])
def test_bf16_polyphase_upsample(cin, cout, s, B, L):
    x = _rand(B, cin, L, seed=71)
    wt = _rand(cin, cout, 2 * s, seed=72) / (cin * 2) ** 0.5
    bias = _rand(cout, seed=73)
    ref = F.conv_transpose1d(_r16(F.leaky_relu(x, 0.1)), _r16(wt), bias, stride=s, padding=s // 2)
    poly = packing.transposed_to_polyphase(wt, s)
    tile = choose_tile(poly.shape[0])
    out = torch.full((B, cout, L * s), float("nan"), device=DEV)
    native.conv1d(x.to(DEV), _pack16(poly, tile, packing.polyphase_row_order(cout, s)), out, c_in=cin, c_out=cout, taps=2, dilation=1, pad=1, tile=tile,
                  epilogue=native.EPI_UPSAMPLE, bias=bias.to(DEV), act_in=True, in_slope=0.1, up_stride=s)
    _close(out, ref, "bf16 upsample")


def test_bf16_gate_epilogue():
    B, T, H = 2, 131, 192
    x = _rand(B, H, T, seed=81)
    w_in = _rand(2 * H, H, 5, seed=82) / (H * 5) ** 0.5
    cond = _rand(B, 2 * H, seed=83)
    a = F.conv1d(_r16(x), _r16(w_in), None, padding=2) + cond.unsqueeze(-1)
    ref = torch.tanh(a[:, :H]) * torch.sigmoid(a[:, H:])
    acts = torch.full((B, H, T), float("nan"), device=DEV)
    native.conv1d(x.to(DEV), _pack16(w_in, native.TILE_PAIR, packing.pair_row_order(H)), acts, c_in=H, c_out=2 * H,
                  taps=5, dilation=1, pad=2, tile=native.TILE_PAIR, epilogue=native.EPI_GATE, bias=cond.to(DEV),
                  bias_bstride=2 * H, hidden=H)
    _close(acts, ref, "bf16 gate")


@pytest.mark.parametrize("ch,k,dil,B,L", [
    (32, 3, 1, 2, 1),
    (32, 11, 5, 2, 37),
    (32, 7, 3, 2, 501),          # one column into the second workgroup
    (64, 11, 5, 2, 463),
    (64, 3, 3, 3, 1200),
    (128, 7, 1, 1, 244),         # exactly one 128-channel workgroup
    (128, 11, 5, 2, 777),
    (256, 3, 1, 1, 245),         # one column into the second 256-channel workgroup
    (256, 11, 5, 2, 600),
])
def test_bf16_fused_resblock_pair(ch, k, dil, B, L):
    x = _rand(B, ch, L, seed=91)
    w1 = _rand(ch, ch, k, seed=92) / (ch * k) ** 0.5
    w2 = _rand(ch, ch, k, seed=93) / (ch * k) ** 0.5
    b1, b2 = _rand(ch, seed=94), _rand(ch, seed=95)
    h = F.conv1d(_r16(F.leaky_relu(x, 0.1)), _r16(w1), b1, padding=(k - 1) * dil // 2, dilation=dil)
    ref = F.conv1d(_r16(F.leaky_relu(h, 0.1)), _r16(w2), b2, padding=(k - 1) // 2) + x
    pi = native.pair_info(ch, native.PREC_BF16)
    assert pi is not None
    p1 = packing.pack_conv_weight_bf16(w1, pi.ck_bf16, pi.rows_per_block).to(DEV)
    p2 = packing.pack_conv_weight_bf16(w2, pi.ck_bf16, pi.rows_per_block).to(DEV)
    xd = x.to(DEV)
    out = torch.full((B, ch, L), float("nan"), device=DEV)
    native.resblock_pair(xd, p1, b1.to(DEV), p2, b2.to(DEV), out, taps=k, dilation=dil,
                         epilogue=native.EPI_RESID, slope=0.1)
    # h is rounded to bf16 between the convolutions: a value on a rounding boundary may fall
    # either way depending on the summation order, so allow one bf16 step of one h element
    got = out.cpu()
    assert torch.isfinite(got).all()
    err = (got - ref).abs().max().item()
    assert err <= 2e-3 * max(1.0, ref.abs().max().item()), err
    assert (got - ref).abs().mean().item() <= 2e-5
    accb = _rand(B, ch, L, seed=96)
    acc = accb.to(DEV).clone()
    native.resblock_pair(xd, p1, b1.to(DEV), p2, b2.to(DEV), acc, taps=k, dilation=dil,
                         epilogue=native.EPI_RESID_AVG, slope=0.1, acc=acc, navg=3.0)
    assert ((acc.cpu() - (accb + ref) / 3.0).abs().max().item()) <= 1e-3
    # and bit-for-bit what two separate bf16 convolution launches produce
    tile = choose_tile(ch)
    tmp = torch.empty_like(out)
    two = torch.empty_like(out)
    native.conv1d(xd, _pack16(w1, tile), tmp, c_in=ch, c_out=ch, taps=k, dilation=dil, pad=(k - 1) * dil // 2,
                  tile=tile, epilogue=native.EPI_STORE, bias=b1.to(DEV), act_in=True, in_slope=0.1)
    native.conv1d(tmp, _pack16(w2, tile), two, c_in=ch, c_out=ch, taps=k, dilation=1, pad=(k - 1) // 2,
                  tile=tile, epilogue=native.EPI_RESID, bias=b2.to(DEV), res=xd, act_in=True, in_slope=0.1)
    assert (two - out).abs().max().item() <= 2e-3


# ---- fp32 operands as three bf16 slices: fp32-level accuracy, same tolerance as the fp32 kernel
def _pack_split(w, tile, row_order=None):
    ti = native.tile_info(tile)
    return packing.pack_conv_weight_split(w.to(DEV), ti.ck_split, ti.rows_per_block, row_order)


def test_split3_is_exact():
    w = _rand(4096, seed=101) * torch.logspace(-6, 3, 4096)
    hi, mid, lo = packing.split3(w)
    assert torch.equal(hi + mid + lo, w) and torch.equal((hi + mid) + lo, w)
    for s in (hi, mid, lo):
        assert torch.equal(s.to(torch.bfloat16).to(torch.float32), s)


@pytest.mark.parametrize("cin,cout,k,dil,B,L,act", [
    (32, 32, 3, 1, 1, 1, True),
    (32, 32, 11, 5, 2, 37, True),
    (64, 64, 7, 3, 3, 517, True),
    (128, 128, 3, 5, 2, 1000, True),
    (256, 256, 11, 1, 1, 260, True),
    (192, 512, 7, 1, 2, 45, False),
    (513, 192, 1, 1, 1, 50, False),
    (40, 72, 5, 2, 2, 130, True),
])
def test_split_conv_matches_fp32(cin, cout, k, dil, B, L, act, monkeypatch):
    # the yardstick below is the fp32 kernel's one long accumulation chain: keep its channel loop whole
    monkeypatch.setattr(native, "SPLIT_SMALL_LAUNCHES", False)
    x = _rand(B, cin, L, seed=111)
    w = _rand(cout, cin, k, seed=112) / (cin * k) ** 0.5
    bias = _rand(B, cout, seed=113)
    res = _rand(B, cout, L, seed=114)
    pad = (k - 1) * dil // 2
    xin = F.leaky_relu(x, 0.1) if act else x
    ref = (F.conv1d(xin.double(), w.double(), None, padding=pad, dilation=dil) + bias.unsqueeze(-1).double())
    tile = choose_tile(cout)
    wp = _pack_split(w, tile)
    out = torch.full((B, cout, L), float("nan"), device=DEV)
    native.conv1d(x.to(DEV), wp, out, c_in=cin, c_out=cout, taps=k, dilation=dil, pad=pad, tile=tile,
                  epilogue=native.EPI_STORE, bias=bias.to(DEV), bias_bstride=cout, act_in=act, in_slope=0.1)
    err = (out.cpu().double() - ref).abs().max().item()
    # yardsticks against the same fp64 result: PyTorch's fp32 convolution on the CPU and this
    # library's own fp32 matrix-core kernel (one long fp32 accumulation chain, like the split one)
    cpu_err = (F.conv1d(xin, w, None, padding=pad, dilation=dil) + bias.unsqueeze(-1)).double().sub(ref).abs().max().item()
    out32 = torch.full((B, cout, L), float("nan"), device=DEV)
    native.conv1d(x.to(DEV), _pack(w, tile), out32, c_in=cin, c_out=cout, taps=k, dilation=dil, pad=pad, tile=tile,
                  epilogue=native.EPI_STORE, bias=bias.to(DEV), bias_bstride=cout, act_in=act, in_slope=0.1)
    mfma_err = (out32.cpu().double() - ref).abs().max().item()
    print(f"max-abs vs fp64: split {err:.2e}, fp32 MFMA kernel {mfma_err:.2e}, PyTorch CPU fp32 {cpu_err:.2e}")
    assert err <= max(2.0 * mfma_err, 4.0 * cpu_err, 1e-6), (err, mfma_err, cpu_err)
    if act:
        out2 = torch.full((B, cout, L), float("nan"), device=DEV)
        native.conv1d(x.to(DEV), wp, out2, c_in=cin, c_out=cout, taps=k, dilation=dil, pad=pad, tile=tile,
                      epilogue=native.EPI_RESID, bias=bias.to(DEV), bias_bstride=cout, res=res.to(DEV),
                      act_in=True, in_slope=0.1)
        _close(out2, (ref + res.double()).float(), "split residual")


@pytest.mark.parametrize("ch,k,dil,B,L", [
    (32, 3, 1, 2, 1),            # a single sample
    (32, 11, 5, 2, 37),          # halo longer than the signal
    (32, 7, 3, 1, 244),          # exactly one workgroup's worth of outputs
    (32, 7, 3, 2, 245),          # one column into the second workgroup, odd length
    (32, 3, 5, 2, 2000),
    (64, 11, 1, 1, 116),         # one 64-channel workgroup
    (64, 11, 5, 2, 463),
    (64, 3, 3, 3, 1200),
])
def test_sliced_fp32_fused_resblock_pair(ch, k, dil, B, L, monkeypatch):
    """The fused pair in sliced-fp32 arithmetic: as close to fp64 as the fp32 kernels, and the
    same bits as the two sliced convolution launches it replaces (same operations in the same order)."""
    monkeypatch.setattr(native, "SPLIT_SMALL_LAUNCHES", False)      # one accumulation chain in the yardstick launches
    x = _rand(B, ch, L, seed=151)
    w1 = _rand(ch, ch, k, seed=152) / (ch * k) ** 0.5
    w2 = _rand(ch, ch, k, seed=153) / (ch * k) ** 0.5
    b1, b2 = _rand(ch, seed=154), _rand(ch, seed=155)
    h = F.conv1d(F.leaky_relu(x.double(), 0.1), w1.double(), b1.double(), padding=(k - 1) * dil // 2, dilation=dil)
    ref = (F.conv1d(F.leaky_relu(h, 0.1), w2.double(), b2.double(), padding=(k - 1) // 2) + x.double())
    pi = native.pair_info(ch, native.PREC_F32_SPLIT)
    assert pi is not None and pi.ck_split == 16
    p1 = packing.pack_conv_weight_split(w1.to(DEV), pi.ck_split, pi.rows_per_block)
    p2 = packing.pack_conv_weight_split(w2.to(DEV), pi.ck_split, pi.rows_per_block)
    xd, b1d, b2d = x.to(DEV), b1.to(DEV), b2.to(DEV)
    out = torch.full((B, ch, L), float("nan"), device=DEV)
    native.resblock_pair(xd, p1, b1d, p2, b2d, out, taps=k, dilation=dil, epilogue=native.EPI_RESID, slope=0.1)
    _close(out, ref.float(), "sliced pair")
    # the two launches: conv1 stored, conv2 with the residual epilogue
    tile = choose_tile(ch)
    q1, q2 = _pack_split(w1, tile), _pack_split(w2, tile)
    hbuf = torch.empty(B, ch, L, device=DEV)
    two = torch.full((B, ch, L), float("nan"), device=DEV)
    native.conv1d(xd, q1, hbuf, c_in=ch, c_out=ch, taps=k, dilation=dil, pad=(k - 1) * dil // 2, tile=tile,
                  epilogue=native.EPI_STORE, bias=b1d, act_in=True, in_slope=0.1)
    native.conv1d(hbuf, q2, two, c_in=ch, c_out=ch, taps=k, dilation=1, pad=(k - 1) // 2, tile=tile,
                  epilogue=native.EPI_RESID, bias=b2d, res=xd, act_in=True, in_slope=0.1)
    assert torch.equal(out, two), (out - two).abs().max().item()
    err = (out.cpu().double() - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err
    accb = _rand(B, ch, L, seed=156)
    acc = accb.to(DEV).clone()
    native.resblock_pair(xd, p1, b1d, p2, b2d, acc, taps=k, dilation=dil, epilogue=native.EPI_RESID_ACC, slope=0.1, acc=acc)
    _close(acc, accb + ref.float(), "sliced pair + accumulate")
    acc = accb.to(DEV).clone()
    native.resblock_pair(xd, p1, b1d, p2, b2d, acc, taps=k, dilation=dil, epilogue=native.EPI_RESID_AVG, slope=0.1,
                         acc=acc, navg=3.0)
    _close(acc, (accb + ref.float()) / 3.0, "sliced pair + accumulate + mean")
    wide = torch.zeros(B, ch + 3, L, device=DEV)
    wide[:, 3:] = xd
    out2 = torch.full((B, ch, L), float("nan"), device=DEV)
    native.resblock_pair(wide[:, 3:], p1, b1d, p2, b2d, out2, taps=k, dilation=dil, epilogue=native.EPI_RESID, slope=0.1)
    assert torch.equal(out2, out)


def test_sliced_fp32_fused_pair_random_shapes(monkeypatch):
    """Random (channels, taps, dilation, batch, length) within the kernel's limits, 1 .. 13 taps: the same
    bits as the two sliced convolution launches (weight ring wrap-around, window edges, lengths around the
    window size)."""
    monkeypatch.setattr(native, "SPLIT_SMALL_LAUNCHES", False)
    g = torch.Generator().manual_seed(4242)
    for case in range(24):
        ch = (32, 64)[int(torch.randint(0, 2, (1,), generator=g))]
        k = (1, 3, 5, 7, 9, 11, 13)[int(torch.randint(0, 7, (1,), generator=g))]
        dmax = 52 // (k - 1) if k > 1 else 8
        dil = int(torch.randint(1, min(dmax, 8) + 1, (1,), generator=g))
        B = int(torch.randint(1, 4, (1,), generator=g))
        tno = native.pair_info(ch, native.PREC_F32_SPLIT).time_tile
        L = int(torch.randint(1, 3 * tno + 40, (1,), generator=g)) if case % 3 else tno * (1 + case % 2) + (case % 5) - 2
        x = torch.randn(B, ch, L, generator=g)
        w1 = torch.randn(ch, ch, k, generator=g) / (ch * k) ** 0.5
        w2 = torch.randn(ch, ch, k, generator=g) / (ch * k) ** 0.5
        b1, b2 = torch.randn(ch, generator=g).to(DEV), torch.randn(ch, generator=g).to(DEV)
        pi = native.pair_info(ch, native.PREC_F32_SPLIT)
        p1 = packing.pack_conv_weight_split(w1.to(DEV), pi.ck_split, pi.rows_per_block)
        p2 = packing.pack_conv_weight_split(w2.to(DEV), pi.ck_split, pi.rows_per_block)
        xd = x.to(DEV)
        out = torch.full((B, ch, L), float("nan"), device=DEV)
        native.resblock_pair(xd, p1, b1, p2, b2, out, taps=k, dilation=dil, epilogue=native.EPI_RESID, slope=0.1)
        tile = choose_tile(ch)
        q1, q2 = _pack_split(w1, tile), _pack_split(w2, tile)
        hbuf = torch.empty(B, ch, L, device=DEV)
        two = torch.full((B, ch, L), float("nan"), device=DEV)
        native.conv1d(xd, q1, hbuf, c_in=ch, c_out=ch, taps=k, dilation=dil, pad=(k - 1) * dil // 2, tile=tile,
                      epilogue=native.EPI_STORE, bias=b1, act_in=True, in_slope=0.1)
        native.conv1d(hbuf, q2, two, c_in=ch, c_out=ch, taps=k, dilation=1, pad=(k - 1) // 2, tile=tile,
                      epilogue=native.EPI_RESID, bias=b2, res=xd, act_in=True, in_slope=0.1)
        assert torch.isfinite(out).all(), (ch, k, dil, B, L)
        assert torch.equal(out, two), (ch, k, dil, B, L, (out - two).abs().max().item())


@pytest.mark.parametrize("B,T,lens,dil", [(2, 50, (50, 21), 1), (3, 129, (129, 1, 77), 1), (1, 300, (300,), 2),
                                          (16, 860, tuple(range(860, 700, -10)), 1)])
def test_fused_wn_layer(B, T, lens, dil):
    H, k = 192, 5
    wi = native.wn_info(H)
    assert wi is not None and native.wn_info(128) is None
    mask = (torch.arange(T).unsqueeze(0) < torch.tensor(lens).unsqueeze(1)).float()
    x = _rand(B, H, T, seed=121)
    w_in = _rand(2 * H, H, k, seed=122) / (H * k) ** 0.5
    cond = _rand(B, 2 * H, seed=123)
    w_rs = _rand(2 * H, H, 1, seed=124) / H ** 0.5
    b_rs = _rand(2 * H, seed=125)
    skip0 = _rand(B, H, T, seed=126)
    a = F.conv1d(x, w_in, None, padding=(k - 1) * dil // 2, dilation=dil) + cond.unsqueeze(-1)
    acts = torch.tanh(a[:, :H]) * torch.sigmoid(a[:, H:])
    rs = F.conv1d(acts, w_rs, b_rs)
    order = packing.pair_row_order(H)
    p_in = packing.pack_conv_weight(w_in, wi.ck, wi.rows_per_block, order).to(DEV)
    p_rs = packing.pack_conv_weight(w_rs, wi.ck, 32).to(DEV)
    xd, md = x.to(DEV), mask.to(DEV)
    for first in (True, False):
        x_out = torch.full((B, H, T), float("nan"), device=DEV)
        skip = skip0.to(DEV).clone()
        native.wn_layer(xd, p_in, cond.to(DEV), 2 * H, p_rs, b_rs.to(DEV), x_out, skip, md,
                        taps=k, dilation=dil, first=first, last=False)
        _close(x_out, (x + rs[:, :H]) * mask.unsqueeze(1), "WN layer residual half")
        _close(skip, rs[:, H:] if first else skip0 + rs[:, H:], "WN layer skip half")
    # last layer: H output rows, out = (skip + rs) * mask
    w_last = _rand(H, H, 1, seed=127) / H ** 0.5
    p_last = packing.pack_conv_weight(w_last, wi.ck, 32).to(DEV)
    out = torch.full((B, H, T), float("nan"), device=DEV)
    skip = skip0.to(DEV).clone()
    native.wn_layer(xd, p_in, cond.to(DEV), 2 * H, p_last, b_rs[:H].to(DEV), out, skip, md,
                    taps=k, dilation=dil, first=False, last=True)
    _close(out, (skip0 + F.conv1d(acts, w_last, b_rs[:H])) * mask.unsqueeze(1), "WN last layer")
    with pytest.raises(native.NativeError):          # output on top of the input
        native.wn_layer(xd, p_in, cond.to(DEV), 2 * H, p_rs, b_rs.to(DEV), xd, skip, md,
                        taps=k, dilation=dil, first=True, last=False)


@pytest.mark.parametrize("B,T,lens,dil", [(2, 50, (50, 21), 1), (3, 129, (129, 1, 77), 1), (1, 300, (300,), 2),
                                          (16, 860, tuple(range(860, 700, -10)), 1),      # 64-column windows
                                          (40, 862, tuple(range(862, 822, -1)), 1)])      # 128-column windows
def test_bf16_fused_wn_layer(B, T, lens, dil):
    """The one-launch layer with bf16 matrix inputs against a reference with the roundings made
    explicit (x, weights and gated activations rounded to bf16, everything else fp32)."""
    H, k = 192, 5
    wi = native.wn_info(H)
    assert wi is not None and wi.ck_bf16 == 32
    mask = (torch.arange(T).unsqueeze(0) < torch.tensor(lens).unsqueeze(1)).float()
    x = _rand(B, H, T, seed=131)
    w_in = _rand(2 * H, H, k, seed=132) / (H * k) ** 0.5
    cond = _rand(B, 2 * H, seed=133)
    w_rs = _rand(2 * H, H, 1, seed=134) / H ** 0.5
    b_rs = _rand(2 * H, seed=135)
    skip0 = _rand(B, H, T, seed=136)
    a = F.conv1d(_r16(x), _r16(w_in), None, padding=(k - 1) * dil // 2, dilation=dil) + cond.unsqueeze(-1)
    acts = torch.tanh(a[:, :H]) * torch.sigmoid(a[:, H:])
    rs = F.conv1d(_r16(acts), _r16(w_rs), b_rs)
    order = packing.pair_row_order(H)
    p_in = packing.pack_conv_weight_bf16(w_in, wi.ck_bf16, wi.rows_per_block, order).to(DEV)
    p_rs = packing.pack_conv_weight_bf16(w_rs, wi.ck_bf16, 32).to(DEV)
    xd, md = x.to(DEV), mask.to(DEV)

    def close16(got, ref, what):        # a gated activation on a bf16 rounding boundary may fall either way
        got = got.cpu()
        assert torch.isfinite(got).all(), what
        err = (got - ref).abs()
        scale = max(1.0, ref.abs().max().item())
        assert err.max().item() <= 6e-3 * scale, f"{what}: {err.max().item():.3e}"
        assert err.mean().item() <= 2e-5 * scale, f"{what}: mean {err.mean().item():.3e}"

    for first in (True, False):
        x_out = torch.full((B, H, T), float("nan"), device=DEV)
        skip = skip0.to(DEV).clone()
        native.wn_layer(xd, p_in, cond.to(DEV), 2 * H, p_rs, b_rs.to(DEV), x_out, skip, md,
                        taps=k, dilation=dil, first=first, last=False)
        close16(x_out, (x + rs[:, :H]) * mask.unsqueeze(1), "WN layer residual half")
        close16(skip, rs[:, H:] if first else skip0 + rs[:, H:], "WN layer skip half")
    w_last = _rand(H, H, 1, seed=137) / H ** 0.5
    p_last = packing.pack_conv_weight_bf16(w_last, wi.ck_bf16, 32).to(DEV)
    out = torch.full((B, H, T), float("nan"), device=DEV)
    skip = skip0.to(DEV).clone()
    native.wn_layer(xd, p_in, cond.to(DEV), 2 * H, p_last, b_rs[:H].to(DEV), out, skip, md,
                    taps=k, dilation=dil, first=False, last=True)
    close16(out, (skip0 + F.conv1d(_r16(acts), _r16(w_last), b_rs[:H])) * mask.unsqueeze(1), "WN last layer")
    # the same layer as two launches of the bf16 convolution kernel (gate epilogue, then the 1x1
    # conv with the residual / skip epilogue): same roundings, same order of sums
    ti = native.tile_info(native.TILE_PAIR)
    g_in = packing.pack_conv_weight_bf16(w_in, ti.ck_bf16, ti.rows_per_block, order).to(DEV)
    acts_d = torch.empty(B, H, T, device=DEV)
    native.conv1d(xd, g_in, acts_d, c_in=H, c_out=2 * H, taps=k, dilation=dil, pad=(k - 1) * dil // 2,
                  tile=native.TILE_PAIR, epilogue=native.EPI_GATE, bias=cond.to(DEV), bias_bstride=2 * H, hidden=H)
    from vits_amd.engine import choose_tile
    t2 = choose_tile(2 * H)
    ti2 = native.tile_info(t2)
    g_rs = packing.pack_conv_weight_bf16(w_rs, ti2.ck_bf16, ti2.rows_per_block).to(DEV)
    x2 = xd.clone()
    skip2 = skip0.to(DEV).clone()
    native.conv1d(acts_d, g_rs, x2, c_in=H, c_out=2 * H, taps=1, dilation=1, pad=0, tile=t2,
                  epilogue=native.EPI_WN_MID, bias=b_rs.to(DEV), res=x2, out2=skip2, mask=md, first=False, hidden=H)
    x_out = torch.empty(B, H, T, device=DEV)
    skip = skip0.to(DEV).clone()
    native.wn_layer(xd, p_in, cond.to(DEV), 2 * H, p_rs, b_rs.to(DEV), x_out, skip, md,
                    taps=k, dilation=dil, first=False, last=False)
    assert torch.equal(x_out, x2) and torch.equal(skip, skip2)


def test_split_conv_over_a_wide_dynamic_range():
    """Slices keep fp32's exponent range: channels scaled from 1e-18 to 1e+18 (weights scaled
    inversely so the products stay O(1)) give the same relative accuracy as unit-scale data."""
    cin, cout, k, B, L = 64, 64, 3, 2, 300
    scale = torch.logspace(-18, 18, cin).view(1, cin, 1)
    x = _rand(B, cin, L, seed=131) * scale
    w = _rand(cout, cin, k, seed=132) / (cin * k) ** 0.5 / scale.view(1, cin, 1)
    ref = F.conv1d(x.double(), w.double(), None, padding=1)
    tile = choose_tile(cout)
    out = torch.full((B, cout, L), float("nan"), device=DEV)
    native.conv1d(x.to(DEV), _pack_split(w, tile), out, c_in=cin, c_out=cout, taps=k, dilation=1, pad=1, tile=tile,
                  epilogue=native.EPI_STORE)
    out32 = torch.full((B, cout, L), float("nan"), device=DEV)
    native.conv1d(x.to(DEV), _pack(w, tile), out32, c_in=cin, c_out=cout, taps=k, dilation=1, pad=1, tile=tile,
                  epilogue=native.EPI_STORE)
    err = (out.cpu().double() - ref).abs().max().item()
    err32 = (out32.cpu().double() - ref).abs().max().item()
    print(f"wide range: split {err:.2e}, fp32 MFMA {err32:.2e}, |ref|max {ref.abs().max().item():.2f}")
    assert err <= max(2.0 * err32, 1e-6)


# ---- a whole multi-receptive-field stage in one launch (bf16 matrix inputs) ----
def _mrf_reference(x, blocks, slope=0.1, rnd=None):
    """decoder.py:119-125 around :46-61; ``rnd`` is applied to whatever enters a convolution
    (the bf16 rounding of the matrix inputs made explicit, or nothing for plain fp32)."""
    rnd = rnd or _r16
    tot = torch.zeros_like(x)
    for k, pairs in blocks:
        xr = x
        for w1, b1, w2, b2, d in pairs:
            h = F.conv1d(rnd(F.leaky_relu(xr, slope)), rnd(w1), b1, padding=(k - 1) * d // 2, dilation=d)
            xr = F.conv1d(rnd(F.leaky_relu(h, slope)), rnd(w2), b2, padding=(k - 1) // 2) + xr
        tot = tot + xr
    return tot / len(blocks)


@pytest.mark.parametrize("ch,B,L,kernels,dils", [
    (32, 2, 1, (3, 7, 11), (1, 3, 5)),
    (32, 2, 37, (3, 7, 11), (1, 3, 5)),
    (32, 3, 384, (3, 7, 11), (1, 3, 5)),       # exactly one workgroup of outputs
    (32, 1, 385, (3, 7, 11), (1, 3, 5)),       # one column into the second workgroup
    (32, 2, 1500, (3, 7, 11), (1, 3, 5)),
    (64, 2, 385, (3, 7, 11), (1, 3, 5)),
    (64, 1, 1031, (3, 7, 11), (1, 3, 5)),
    (64, 2, 300, (5, 3), (2, 1)),              # two resblocks of two pairs
    (32, 1, 700, (11,), (1, 3, 5)),            # a single resblock: the mean is the resblock
    (32, 2, 900, (3, 7, 11), (5,)),            # one pair per resblock: no chain for a rounding flip to travel down
    (64, 2, 900, (11, 7, 3), (1,)),
    (128, 2, 500, (3,), (1, 3, 5)),            # 128 channels: one resblock per launch
    (128, 1, 233, (3,), (1, 3, 5)),            # ... 232 output columns per workgroup at reach 12
    (128, 1, 300, (7,), (1, 3, 5)),
    (256, 2, 209, (3,), (1, 3, 5)),            # 104 output columns per workgroup
])
def test_bf16_fused_mrf_stage(ch, B, L, kernels, dils):
    mi = native.mrf_info(ch, native.PREC_BF16)
    assert mi is not None
    x = _rand(B, ch, L, seed=201)
    ref_blocks, dev_blocks = [], []
    seed = 210
    for k in kernels:
        rp, dp = [], []
        for d in dils:
            w1 = _rand(ch, ch, k, seed=seed) / (ch * k) ** 0.5
            w2 = _rand(ch, ch, k, seed=seed + 1) / (ch * k) ** 0.5
            b1, b2 = _rand(ch, seed=seed + 2) * 0.3, _rand(ch, seed=seed + 3) * 0.3
            seed += 4
            rp.append((w1, b1, w2, b2, d))
            dp.append((packing.pack_conv_weight_bf16(w1, mi.ck_bf16, mi.rows_per_block).to(DEV), b1.to(DEV),
                       packing.pack_conv_weight_bf16(w2, mi.ck_bf16, mi.rows_per_block).to(DEV), b2.to(DEV), d))
        ref_blocks.append((k, rp))
        dev_blocks.append((k, dp))
    ref = _mrf_reference(x, ref_blocks)
    out = torch.full((B, ch, L), float("nan"), device=DEV)
    native.mrf_stage(x.to(DEV), out, dev_blocks, slope=0.1)
    got = out.cpu()
    assert torch.isfinite(got).all()
    # Every conv input is rounded to bf16.  A value on a rounding boundary falls either way with the
    # summation order (a few per 100k elements), and down a chain of pairs one such flip grows to the
    # bf16 noise floor over its receptive cone.  So: within one bf16 step of the rounded reference
    # everywhere, and as close to the unrounded fp32 result as the rounded reference itself is.
    scale = max(1.0, ref.abs().max().item())
    err = (got - ref).abs()
    assert err.max().item() <= 4e-3 * scale, err.max().item()
    if len(dils) == 1:
        assert err.mean().item() <= 2e-6 * scale, err.mean().item()
        assert (err > 1e-5 * scale).float().mean().item() <= 0.02
    exact = _mrf_reference(x, ref_blocks, rnd=lambda t: t)
    noise_ref = (ref - exact).pow(2).mean().sqrt().item()
    noise_got = (got - exact).pow(2).mean().sqrt().item()
    assert noise_got <= 1.1 * noise_ref + 1e-7, (noise_got, noise_ref)
    assert err.pow(2).mean().sqrt().item() <= 1.5 * noise_ref + 1e-6


# ---- the same stage / resblock chain on the exact fp32 matrix instruction ----
def _mrf_blocks(ch, kernels, dils, pack, seed=310):
    ref_blocks, dev_blocks = [], []
    for k in kernels:
        rp, dp = [], []
        for d in dils:
            w1 = _rand(ch, ch, k, seed=seed) / (ch * k) ** 0.5
            w2 = _rand(ch, ch, k, seed=seed + 1) / (ch * k) ** 0.5
            b1, b2 = _rand(ch, seed=seed + 2) * 0.3, _rand(ch, seed=seed + 3) * 0.3
            seed += 4
            rp.append((w1, b1, w2, b2, d))
            dp.append((pack(w1).to(DEV), b1.to(DEV), pack(w2).to(DEV), b2.to(DEV), d))
        ref_blocks.append((k, rp))
        dev_blocks.append((k, dp))
    return ref_blocks, dev_blocks


@pytest.mark.parametrize("ch,B,L,kernels,dils", [
    (32, 2, 1, (3,), (1, 3, 5)),               # a single sample
    (32, 2, 37, (3,), (1, 3, 5)),              # the reach (12) against a short signal
    (32, 1, 488, (3,), (1, 3, 5)),             # exactly one workgroup of outputs at reach 12
    (32, 3, 489, (3,), (1, 3, 5)),             # one column into the second workgroup
    (32, 2, 2001, (3,), (1, 3, 5)),            # odd length, several workgroups
    (64, 2, 489, (3,), (1, 3, 5)),
    (64, 1, 1500, (3,), (1, 3, 5)),
    (32, 2, 700, (3, 7, 11), (1, 3, 5)),       # the whole stage: three kernel sizes summed in registers
    (64, 1, 500, (3, 7, 11), (1, 3, 5)),
    (64, 2, 393, (11, 7), (1, 5)),             # two resblocks of two pairs, longest reach first
    (32, 1, 300, (5,), (2, 1)),
])
def test_fp32_fused_mrf_stage(ch, B, L, kernels, dils):
    """decoder.py:119-125 around :46-61 as chains of F.conv1d against one launch."""
    mi = native.mrf_info(ch, native.PREC_F32)
    assert mi is not None and mi.max_blocks == 4
    x = _rand(B, ch, L, seed=301)
    ref_blocks, dev_blocks = _mrf_blocks(ch, kernels, dils, lambda w: packing.pack_conv_weight(w, mi.ck, mi.rows_per_block))
    ref = _mrf_reference(x, ref_blocks, rnd=lambda t: t)
    xd = x.to(DEV)
    out = torch.full((B, ch, L), float("nan"), device=DEV)
    native.mrf_stage(xd, out, dev_blocks, slope=0.1)
    _close(out, ref, "stage in one launch")
    # one launch per resblock, the sum carried through `out`, the last launch dividing
    out2 = torch.full((B, ch, L), float("nan"), device=DEV)
    for r, blk in enumerate(dev_blocks):
        native.mrf_stage(xd, out2, [blk], slope=0.1, acc=out2 if r else None,
                         navg=len(dev_blocks) if r + 1 == len(dev_blocks) else 1)
    _close(out2, ref, "one launch per resblock")
    if len(kernels) > 1:
        assert torch.equal(out2, out)          # same products, same order of sums
    # a channel-strided view of a wider tensor as input
    wide = torch.zeros(B, ch + 3, L, device=DEV)
    wide[:, 3:] = xd
    out3 = torch.full((B, ch, L), float("nan"), device=DEV)
    native.mrf_stage(wide[:, 3:], out3, dev_blocks, slope=0.1)
    assert torch.equal(out3, out)


@pytest.mark.parametrize("L", [37, 2001])
def test_fp32_resblock_chain_is_bit_identical_to_its_pair_launches(L):
    """32 channels: the chained launch runs the pair kernel's fma chains in the same order."""
    ch, k = 32, 3
    pi = native.pair_info(ch)
    mi = native.mrf_info(ch, native.PREC_F32)
    assert (pi.ck, pi.rows_per_block) == (mi.ck, mi.rows_per_block)
    x = _rand(2, ch, L, seed=401).to(DEV)
    _, dev_blocks = _mrf_blocks(ch, (k,), (1, 3, 5), lambda w: packing.pack_conv_weight(w, mi.ck, mi.rows_per_block), seed=410)
    prev = _rand(2, ch, L, seed=402).to(DEV)
    chained = torch.empty_like(x)
    native.mrf_stage(x, chained, dev_blocks, slope=0.1, acc=prev, navg=3)
    cur, bufs = x, [torch.empty_like(x), torch.empty_like(x)]
    pairs = dev_blocks[0][1]
    for j, (w1, b1, w2, b2, d) in enumerate(pairs):
        last = j + 1 == len(pairs)
        dst = bufs[j & 1]
        native.resblock_pair(cur, w1, b1, w2, b2, dst, taps=k, dilation=d, slope=0.1,
                             epilogue=native.EPI_RESID_AVG if last else native.EPI_RESID,
                             acc=prev if last else None, navg=3.0)
        cur = dst
    assert torch.equal(chained, cur)


def test_fused_mrf_stage_rejects_bad_arguments():
    assert native.mrf_info(512, native.PREC_BF16) is None
    assert native.mrf_info(128, native.PREC_F32) is None and native.mrf_info(32, native.PREC_F32_SPLIT) is None
    assert native.mrf_info(32, native.PREC_BF16).max_blocks == 4 and native.mrf_info(128, native.PREC_BF16).max_blocks == 1
    mi = native.mrf_info(32, native.PREC_BF16)
    w = packing.pack_conv_weight_bf16(_rand(32, 32, 11, seed=1), mi.ck_bf16, mi.rows_per_block).to(DEV)
    b = torch.zeros(32, device=DEV)
    x = torch.zeros(1, 32, 64, device=DEV)
    with pytest.raises(native.NativeError):          # reach 5 * (1+1 + 3+1 + 7+1) = 70 > 64
        native.mrf_stage(x, torch.empty_like(x), [(11, [(w, b, w, b, d) for d in (1, 3, 7)])], slope=0.1)
    with pytest.raises(native.NativeError):          # out on top of x
        native.mrf_stage(x, x, [(11, [(w, b, w, b, 1)])], slope=0.1)
    w1 = packing.pack_conv_weight_bf16(_rand(32, 32, 1, seed=3), mi.ck_bf16, mi.rows_per_block).to(DEV)
    with pytest.raises(native.NativeError):          # a one-tap kernel (the weight stream needs at least three)
        native.mrf_stage(x, torch.empty_like(x), [(1, [(w1, b, w1, b, 1)])], slope=0.1)
    m8 = native.mrf_info(128, native.PREC_BF16)
    w8 = packing.pack_conv_weight_bf16(_rand(128, 128, 3, seed=2), m8.ck_bf16, m8.rows_per_block).to(DEV)
    b8 = torch.zeros(128, device=DEV)
    x8 = torch.zeros(1, 128, 64, device=DEV)
    with pytest.raises(native.NativeError):          # two resblocks in one launch at 128 channels
        native.mrf_stage(x8, torch.empty_like(x8), [(3, [(w8, b8, w8, b8, 1)])] * 2, slope=0.1)


# ---- split channel loop of launches too small to fill the GPU (fp32 kernel, scratch workspace) ----
@pytest.mark.parametrize("cin,cout,k,dil,B,L", [
    (256, 256, 11, 1, 1, 2064),      # decoder stage 1 of a 3 s utterance: 66 workgroups, 8 splits
    (256, 256, 3, 5, 1, 70),
    (192, 512, 7, 1, 1, 258),
    (512, 256, 7, 1, 2, 33),
    (64, 64, 7, 3, 1, 300),          # exactly 4 chunks: 2 splits
])
def test_split_channel_loop_matches_the_single_workgroup_form(cin, cout, k, dil, B, L, monkeypatch):
    x = _rand(B, cin, L, seed=301).to(DEV)
    w = _rand(cout, cin, k, seed=302) / (cin * k) ** 0.5
    bias = _rand(cout, seed=303).to(DEV)
    res = _rand(B, cout, L, seed=304).to(DEV)
    pad = (k - 1) * dil // 2
    tile = choose_tile(cout)
    wp = _pack(w, tile)

    def run(split):
        monkeypatch.setattr(native, "SPLIT_SMALL_LAUNCHES", split)
        out = torch.full((B, cout, L), float("nan"), device=DEV)
        native.conv1d(x, wp, out, c_in=cin, c_out=cout, taps=k, dilation=dil, pad=pad, tile=tile,
                      epilogue=native.EPI_RESID, bias=bias, res=res, act_in=True, in_slope=0.1)
        return out

    a = native.ConvArgs()
    one = run(False)
    two = run(True)
    three = run(True)                                  # the counters were left at zero
    ref = F.conv1d(F.leaky_relu(x.cpu(), 0.1), w, bias.cpu(), padding=pad, dilation=dil) + res.cpu()
    _close(two, ref, "split channel loop")
    assert torch.equal(two, three)                     # fixed summation order: run-to-run identical
    assert (one - two).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    assert not torch.equal(one, two) or cin * k < 64   # the split really ran (another summation order)
    ws = native.workspace(DEV)
    assert int(ws[:native.WS_COUNTER_BYTES // 4].view(torch.int32).abs().max()) == 0


@pytest.mark.parametrize("cin,cout,k,dil,B,L", [
    (256, 256, 11, 1, 1, 2064),      # 17 tiles x 2 row blocks = 34 jobs: 4 shares of the 16 chunks
    (256, 256, 3, 5, 1, 70),
    (192, 384, 5, 1, 1, 237),        # 12 chunks: the share count must divide them
    (128, 128, 7, 1, 2, 500),
    (64, 64, 7, 3, 1, 300),          # exactly 4 chunks: 2 shares
])
def test_split_channel_loop_of_the_sliced_fp32_kernel(cin, cout, k, dil, B, L, monkeypatch):
    """The persistent sliced-fp32 kernel with its channel chunks shared out over several jobs per
    output tile: fp32-level agreement with the fp64 result and with the unsplit launch, run-to-run
    identical, counters left at zero."""
    x = _rand(B, cin, L, seed=321).to(DEV)
    w = _rand(cout, cin, k, seed=322) / (cin * k) ** 0.5
    bias = _rand(cout, seed=323).to(DEV)
    res = _rand(B, cout, L, seed=324).to(DEV)
    pad = (k - 1) * dil // 2
    tile = choose_tile(cout)
    wp = _pack_split(w, tile)

    def run(split):
        monkeypatch.setattr(native, "SPLIT_SMALL_LAUNCHES", split)
        out = torch.full((B, cout, L), float("nan"), device=DEV)
        native.conv1d(x, wp, out, c_in=cin, c_out=cout, taps=k, dilation=dil, pad=pad, tile=tile,
                      epilogue=native.EPI_RESID, bias=bias, res=res, act_in=True, in_slope=0.1)
        return out

    one, two, three = run(False), run(True), run(True)
    ref = (F.conv1d(F.leaky_relu(x.cpu(), 0.1).double(), w.double(), bias.cpu().double(), padding=pad, dilation=dil)
           + res.cpu().double())
    lim = 2e-5 * max(1.0, ref.abs().max().item())
    assert torch.isfinite(two).all()
    assert (two.cpu().double() - ref).abs().max().item() <= lim
    assert (one.cpu().double() - ref).abs().max().item() <= lim
    assert torch.equal(two, three)
    assert not torch.equal(one, two)                   # the shares really ran (another summation order)
    ws = native.workspace(DEV)
    assert int(ws[:native.WS_COUNTER_BYTES // 4].view(torch.int32).abs().max()) == 0


def test_split_channel_loop_workspace_contract():
    B, cin, cout, k, L = 1, 256, 256, 7, 2064
    x = _rand(B, cin, L, seed=311).to(DEV)
    wp = _pack(_rand(cout, cin, k, seed=312) / (cin * k) ** 0.5, choose_tile(cout))
    out = torch.empty(B, cout, L, device=DEV)
    a = native.ConvArgs()
    a.x, a.x_bstride, a.x_cstride = native._view(x)
    a.out, a.out_bstride, a.out_cstride = native._view(out)
    a.w = wp.data_ptr()
    a.batch, a.c_in, a.c_out, a.length = B, cin, cout, L
    a.taps, a.dilation, a.pad, a.tile, a.epilogue, a.narrow = k, 1, 3, choose_tile(cout), native.EPI_STORE, 1
    L_ = native.lib()
    need = L_.vk_conv1d_workspace_bytes(C.byref(a))
    # 33 tiles x 2 row blocks = 66 workgroups -> 8 splits of a 128 x 64 tile
    assert need == native.WS_COUNTER_BYTES + 66 * 8 * 128 * 64 * 4
    sp = native.stream_ptr(x.device)
    assert L_.vk_conv1d(C.byref(a), sp) == 0           # no workspace: runs unsplit
    small = torch.zeros(need // 4 - 1, device=DEV)
    a.workspace, a.workspace_bytes = small.data_ptr(), small.numel() * 4
    assert L_.vk_conv1d(C.byref(a), sp) == -2
    assert b"scratch" in L_.vk_last_error()
    a.tile, a.c_out = native.TILE_M128, 256
    big = _rand(16, cin, 6896, seed=313).to(DEV)       # a launch that fills the GPU does not split
    a.x, a.x_bstride, a.x_cstride = native._view(big)
    o2 = torch.empty(16, cout, 6896, device=DEV)
    a.out, a.out_bstride, a.out_cstride = native._view(o2)
    a.batch, a.length, a.narrow = 16, 6896, 0
    assert L_.vk_conv1d_workspace_bytes(C.byref(a)) == 0
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("sliced", [False, True])
def test_split_channel_loop_is_run_to_run_identical(sliced):
    """A launch too small for the GPU splits its channel loop over workgroups that meet in scratch
    memory (arrival counter + partial tiles, no fence): many back-to-back launches of one
    convolution must all give the bits of the first, and those must match the unsplit launch to
    rounding.  A lost or stale partial, or a miscounted arrival, shows up as a differing repeat."""
    from vits_amd.engine import choose_tile
    g = torch.Generator().manual_seed(5)
    before = native.SPLIT_SMALL_LAUNCHES
    try:
        for cin, cout, k, dil, B, L in [(256, 256, 3, 1, 1, 2064), (256, 256, 11, 5, 1, 2064), (192, 384, 5, 1, 1, 258),
                                        (512, 256, 7, 1, 2, 33), (128, 128, 7, 3, 1, 1500), (64, 64, 11, 1, 1, 3000)]:
            x = torch.randn(B, cin, L, generator=g).to(DEV)
            w = torch.randn(cout, cin, k, generator=g) / (cin * k) ** 0.5
            tile = choose_tile(cout)
            ti = native.tile_info(tile)
            wp = (packing.pack_conv_weight_split(w.to(DEV), ti.ck_split, ti.rows_per_block) if sliced
                  else packing.pack_conv_weight(w, ti.ck, ti.rows_per_block).to(DEV))
            pad = (k - 1) * dil // 2

            def run(split, out):
                native.SPLIT_SMALL_LAUNCHES = split
                native.conv1d(x, wp, out, c_in=cin, c_out=cout, taps=k, dilation=dil, pad=pad, tile=tile,
                              epilogue=native.EPI_STORE)
            ref = torch.empty(B, cout, L, device=DEV)
            run(False, ref)
            first = torch.empty_like(ref)
            run(True, first)
            outs = [torch.empty_like(ref) for _ in range(100)]
            for o in outs:
                run(True, o)
            torch.cuda.synchronize()
            differing = sum(int(not torch.equal(o, first)) for o in outs)
            assert differing == 0, (cin, cout, k, L, differing)
            assert (first - ref).abs().max().item() <= 1e-4, (cin, cout, k, L)
    finally:
        native.SPLIT_SMALL_LAUNCHES = before
