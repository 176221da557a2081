"""``vk_conv1d_static``: the convolution entry point that allocates nothing.  A launch too small
for the GPU splits its channel loop over workgroups; this entry point keeps the partial sums and
arrival counters in a device array that is part of the library instead of a caller's workspace.
Its output must be bit-identical to ``vk_conv1d``'s with a workspace (same kernel, same split, same
summation order), on two calls in a row, the second on a smaller input."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from vits_amd import native, packing
from vits_amd.engine import choose_tile

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rand(*shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def _both_entries(cin, cout, k, dil, B, L, seed, sliced=False):
    x = _rand(B, cin, L, seed=seed).to(DEV)
    w = _rand(cout, cin, k, seed=seed + 1) / (cin * k) ** 0.5
    bias = _rand(cout, seed=seed + 2).to(DEV)
    res = _rand(B, cout, L, seed=seed + 3).to(DEV)
    pad = (k - 1) * dil // 2
    tile = choose_tile(cout)
    ti = native.tile_info(tile)
    wp = (packing.pack_conv_weight_split(w.to(DEV), ti.ck_split, ti.rows_per_block) if sliced
          else packing.pack_conv_weight(w, ti.ck, ti.rows_per_block).to(DEV))
    outs = []
    for static in (False, True):
        out = torch.full((B, cout, L), float("nan"), device=DEV)
        native.conv1d(x, wp, out, c_in=cin, c_out=cout, taps=k, dilation=dil, pad=pad, tile=tile,
                      epilogue=native.EPI_RESID, bias=bias, res=res, act_in=True, in_slope=0.1, static_scratch=static)
        outs.append(out)
    ref = F.conv1d(F.leaky_relu(x.cpu(), 0.1), w, bias.cpu(), padding=pad, dilation=dil) + res.cpu()
    return outs[0], outs[1], ref


def test_static_scratch_entry_matches_the_workspace_entry():
    # first call: 66 workgroups x 8 splits; second call, smaller input: 22 workgroups x 8 splits
    for (cin, cout, k, dil, B, L), seed in (((256, 256, 7, 1, 1, 2064), 401), ((256, 256, 3, 3, 1, 700), 411)):
        with_ws, static, ref = _both_entries(cin, cout, k, dil, B, L, seed)
        assert torch.isfinite(static).all()
        assert torch.equal(with_ws, static), (with_ws - static).abs().max().item()
        err = (static.cpu() - ref).abs().max().item()
        assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err
    # and once more on the first shape: whatever the smaller call left in the array does not matter
    with_ws, static, _ = _both_entries(256, 256, 7, 1, 1, 2064, 421)
    assert torch.equal(with_ws, static)
    # a call with every tensor already there takes no device memory, from the driver's view either
    x = _rand(1, 256, 2064, seed=451).to(DEV)
    ti = native.tile_info(choose_tile(256))
    wp = packing.pack_conv_weight(_rand(256, 256, 7, seed=452), ti.ck, ti.rows_per_block).to(DEV)
    out = torch.empty_like(x)
    torch.cuda.synchronize()
    free0, alloc0 = torch.cuda.mem_get_info()[0], torch.cuda.memory_allocated()
    native.conv1d(x, wp, out, c_in=256, c_out=256, taps=7, dilation=1, pad=3, tile=choose_tile(256),
                  epilogue=native.EPI_STORE, static_scratch=True)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0 and torch.cuda.memory_allocated() == alloc0


def test_static_scratch_entry_with_the_sliced_fp32_kernel():
    """The sliced-fp32 kernel shares out its channel chunks through the same scratch: the static
    entry and the workspace entry agree bit for bit there too, on two calls in a row."""
    for (cin, cout, k, dil, B, L), seed in (((256, 256, 7, 1, 1, 2064), 461), ((192, 384, 5, 1, 1, 237), 471)):
        with_ws, static, ref = _both_entries(cin, cout, k, dil, B, L, seed, sliced=True)
        assert torch.isfinite(static).all()
        assert torch.equal(with_ws, static), (with_ws - static).abs().max().item()
        assert (static.cpu() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


def test_static_scratch_entry_without_a_split_and_beyond_its_size():
    # a launch that fills the GPU does not split: same result, the array is not touched
    with_ws, static, ref = _both_entries(64, 64, 7, 3, 4, 9000, 431)
    assert torch.equal(with_ws, static)
    assert (static.cpu() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    # wide tiles, 100 workgroups x 8 splits x 64 KiB of partials: more than the array holds.  The
    # static entry narrows the split to what fits (5 shares) and runs; the workspace entry, given
    # a workspace of the array's size, says what it would need
    B, cin, cout, k, L = 1, 256, 256, 7, 6400
    x = _rand(B, cin, L, seed=441).to(DEV)
    w = _rand(cout, cin, k, seed=442) / (cin * k) ** 0.5
    ti = native.tile_info(native.TILE_M128)
    wp = packing.pack_conv_weight(w, ti.ck, ti.rows_per_block).to(DEV)
    out = torch.full((B, cout, L), float("nan"), device=DEV)
    a = native.ConvArgs()
    a.x, a.x_bstride, a.x_cstride = native._view(x)
    a.out, a.out_bstride, a.out_cstride = native._view(out)
    a.w = wp.data_ptr()
    a.batch, a.c_in, a.c_out, a.length = B, cin, cout, L
    a.taps, a.dilation, a.pad, a.tile, a.epilogue, a.narrow = k, 1, 3, native.TILE_M128, native.EPI_STORE, 0
    lib = native.lib()
    need = lib.vk_conv1d_workspace_bytes(C.byref(a))
    static_bytes = 65536 + (32 << 20)
    assert need == 65536 + 100 * 8 * 128 * 128 * 4 and need > static_bytes
    for _ in range(2):
        assert lib.vk_conv1d_static(C.byref(a), native.stream_ptr(x.device)) == 0, lib.vk_last_error().decode()
    torch.cuda.synchronize()
    ref = F.conv1d(x.cpu(), w, None, padding=3)
    assert (out.cpu() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    small = torch.zeros(static_bytes // 4, device=DEV)
    a.workspace, a.workspace_bytes = small.data_ptr(), static_bytes
    assert lib.vk_conv1d(C.byref(a), native.stream_ptr(x.device)) == -2
    msg = lib.vk_last_error().decode()
    assert "scratch" in msg and str(need) in msg
    torch.cuda.synchronize()
