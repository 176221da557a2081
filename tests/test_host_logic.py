"""CPU-only tests of the host side: weight folding/packing, the polyphase rewrite of the
transposed convolution, state_dict compatibility of the drop-in modules, tile choice,
and that the C-ABI library loads and exports every symbol the header declares."""
import ctypes
import os
import re

import pytest
import torch
import torch.nn.functional as F

from vits_amd import native, packing, synth
from vits_amd.engine import choose_tile
from vits_amd.module.vits_generator import HotPathGenerator, generate_path, sequence_mask

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _unpack(packed, rows, cin, taps, ck, rows_per_block):
    """Inverse of pack_conv_weight, written from the kernel's point of view: walk row tiles,
    chunks, taps, groups, lanes and the four k-steps of a group."""
    rpad = -(-rows // rows_per_block) * rows_per_block
    cpad = -(-cin // ck) * ck
    w = torch.zeros(rpad, cpad, taps)
    p = packed[:-1024].view(rpad // 32, cpad // ck, taps, ck // 8, 64, 4)
    for lane in range(64):
        r, h = lane & 31, lane >> 5
        for g in range(ck // 8):
            for e in range(4):
                ch = 2 * (4 * g + e) + h
                # p[mt, chunk, tap, g, lane, e] -> w[32*mt + r, chunk*ck + ch, tap]
                w.view(rpad // 32, 32, cpad // ck, ck, taps)[:, r, :, ch, :] = p[:, :, :, g, lane, e]
    return w[:rows, :cin]


@pytest.mark.parametrize("rows,cin,taps,tile", [(128, 128, 3, 0), (384, 192, 5, 0), (64, 64, 11, 1),
                                                 (32, 32, 7, 2), (96, 192, 1, 2), (192, 513, 1, 1)])
def test_pack_roundtrip(rows, cin, taps, tile):
    ti = native.tile_info(tile)
    w = torch.randn(rows, cin, taps)
    packed = packing.pack_conv_weight(w, ti.ck, ti.rows_per_block)
    assert packed.numel() % 256 == 0
    assert torch.equal(_unpack(packed, rows, cin, taps, ti.ck, ti.rows_per_block), w)
    assert torch.count_nonzero(packed[-1024:]) == 0          # spare groups for the prefetch


def test_pair_row_order_interleaves_halves():
    order = packing.pair_row_order(192)
    assert sorted(order.tolist()) == list(range(384))
    assert order[:32].tolist() == list(range(32))
    assert order[32:64].tolist() == list(range(192, 224))
    assert order[64:96].tolist() == list(range(32, 64))


def test_fold_weight_norm_matches_definition():
    v = torch.randn(8, 5, 3)
    g = torch.rand(8, 1, 1) + 0.5
    w = packing.fold_weight_norm(v, g)
    ref = g * v / v.flatten(1).norm(dim=1).view(-1, 1, 1)
    assert torch.allclose(w, ref, atol=1e-6)


@pytest.mark.parametrize("stride,cin,cout", [(8, 6, 4), (2, 5, 3), (4, 3, 7)])
def test_polyphase_equals_conv_transpose(stride, cin, cout):
    """The 2-tap polyphase convolution reproduces nn.ConvTranspose1d(k=2s, stride=s, padding=s/2)."""
    L = 13
    wt = torch.randn(cin, cout, 2 * stride)
    x = torch.randn(2, cin, L)
    ref = F.conv_transpose1d(x, wt, None, stride=stride, padding=stride // 2)
    poly = packing.transposed_to_polyphase(wt, stride)          # [stride*cout, cin, 2]
    xp = F.pad(x, (1, 1))                                        # x[t-1], x[t], x[t+1]
    out = torch.zeros(2, cout, L * stride)
    for p in range(stride):
        wp = poly[p * cout:(p + 1) * cout]
        shift = 0 if p < stride // 2 else 1                      # lower phases read (t-1, t), upper (t, t+1)
        y = F.conv1d(xp[:, :, shift:shift + L + 1], wp)          # [B, cout, L]
        out[:, :, p::stride] = y
    assert ref.shape == out.shape
    assert torch.allclose(out, ref, atol=1e-5)


def test_choose_tile():
    assert choose_tile(512) == native.TILE_M128
    assert choose_tile(384) == native.TILE_M128
    assert choose_tile(192) == native.TILE_M64
    assert choose_tile(64) == native.TILE_M64
    assert choose_tile(96) == native.TILE_M32
    assert choose_tile(32) == native.TILE_M32


def test_state_dict_layout_matches_reference_keys():
    """Key names, order and shapes of the hot-path sub-modules (858-key generator minus text
    encoder and duration predictor), as the reference's modules register them."""
    sd = synth.hot_path_state_dict(999)
    gen = HotPathGenerator(100)
    own = gen.state_dict()
    assert list(own.keys()) == list(sd.keys())
    for k, v in sd.items():
        assert own[k].shape == v.shape, k
    assert len([k for k in own if k.startswith("decoder.")]) == 233
    assert len([k for k in own if k.startswith("flow.")]) == 124
    assert len([k for k in own if k.startswith("posterior_encoder.")]) == 103
    assert own["decoder.ups.0.weight_g"].shape == (512, 1, 1)          # norm axis = input channels
    assert own["decoder.ups.0.weight_v"].shape == (512, 256, 16)
    assert own["flow.flows.6.wn.res_skip_layers.3.weight_v"].shape == (192, 192, 1)
    assert sum(v.numel() for k, v in own.items() if k.startswith("decoder.")) == 14_468_608
    assert sum(v.numel() for k, v in own.items() if k.startswith("flow.")) == 8_687_232
    assert sum(v.numel() for k, v in own.items() if k.startswith("posterior_encoder.")) == 8_823_168
    gen.load_state_dict(sd, strict=True)


def test_reference_zero_init_of_post():
    gen = HotPathGenerator(4, with_posterior=False)
    for f in (0, 2, 4, 6):
        assert torch.count_nonzero(gen.flow.flows[f].post.weight) == 0
        assert torch.count_nonzero(gen.flow.flows[f].post.bias) == 0


def test_generate_path_and_mask():
    dur = torch.tensor([[[2., 0., 3.]]])
    mask = torch.ones(1, 1, 5, 3)
    path = generate_path(dur, mask)[0, 0]
    assert path.tolist() == [[1, 0, 0], [1, 0, 0], [0, 0, 1], [0, 0, 1], [0, 0, 1]]
    assert sequence_mask(torch.tensor([2, 4])).tolist() == [[True, True, False, False], [True] * 4]


def test_library_exports_exactly_the_declared_symbols():
    """Declared in include/vits_amd.h <=> exported with C linkage: nothing missing, and no
    undeclared entry point (diagnostic hooks live in the separate -DVK_DIAG build)."""
    import subprocess
    header = open(os.path.join(ROOT, "include", "vits_amd.h")).read()
    declared = set(re.findall(r"\b(vk_[a-z0-9_]+)\s*\(", header))
    assert {"vk_conv1d", "vk_speaker_linear", "vk_post_conv_tanh", "vk_abi_version",
            "vk_last_error", "vk_tile_query", "vk_conv1d_lds_bytes", "vk_sizeof_args"} <= declared
    lib = ctypes.CDLL(native.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/vits_amd.h but not exported"
    nm = subprocess.run(["nm", "-D", "--defined-only", native.LIB_PATH], capture_output=True, text=True, check=True)
    exported = {ln.split()[-1] for ln in nm.stdout.splitlines() if ln.split() and ln.split()[-1].startswith("vk_")}
    assert exported == declared, (sorted(exported - declared), sorted(declared - exported))
    assert native.lib().vk_abi_version() == native.header_abi_version()


def test_binding_structs_match_the_library():
    L = native.lib()
    for which, mirror in enumerate((native.ConvArgs, native.PairArgs, native.WnArgs, native.TileInfo)):
        assert L.vk_sizeof_args(which) == ctypes.sizeof(mirror), mirror.__name__
    assert L.vk_sizeof_args(99) == 0


def test_product_library_reads_no_environment():
    """The switches for timing experiments exist only in the diagnostic build."""
    blob = open(native.LIB_PATH, "rb").read()
    for name in (b"VK_DEBUG_SKIP", b"VK_LDS_MIN", b"VK_NO_VEC", b"VK_XCD_MAP", b"vk_debug_set_stamps"):
        assert name not in blob, name


def _conv_args(**kw):
    """A plausible vk_conv_args for host-side validation (pointers are never dereferenced)."""
    a = native.ConvArgs()
    B, C, L = 2, 32, 1024
    a.x, a.x_bstride, a.x_cstride = 0x10000000, C * L, L
    a.w = 0x20000000
    a.out, a.out_bstride, a.out_cstride = 0x30000000, C * L, L
    a.batch, a.c_in, a.c_out, a.length = B, C, C, L
    a.taps, a.dilation, a.pad = 3, 1, 1
    a.tile, a.epilogue, a.act_in, a.in_slope = native.TILE_M32, native.EPI_STORE, 1, 0.1
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def test_argument_validation_without_gpu():
    """Bad arguments are rejected on the host before anything is launched."""
    a = native.ConvArgs()
    L = native.lib()
    assert L.vk_conv1d(ctypes.byref(a), None) == -1                 # null pointers
    assert b"null" in L.vk_last_error()
    assert L.vk_post_conv_tanh(None, None, None, 1, 1, 1, 7, 0.01, None) == -1
    assert L.vk_speaker_linear(None, None, None, None, 1, 1, 1, None) == -1
    assert L.vk_conv1d_lds_bytes(ctypes.byref(_conv_args())) > 0
    # an output on top of the input races with the neighbours' halo reads: rejected
    assert L.vk_conv1d_lds_bytes(ctypes.byref(_conv_args(out=0x10000000))) == -1
    assert b"overlaps" in L.vk_last_error()
    assert L.vk_conv1d_lds_bytes(ctypes.byref(_conv_args(out=0x10000000 + 4 * 1024 * 63))) == -1    # partial overlap
    two = _conv_args(epilogue=native.EPI_RESID_ACC, res=0x40000000, res_bstride=32 * 1024, res_cstride=1024,
                     out2=0x10000000, out2_bstride=32 * 1024, out2_cstride=1024)
    assert L.vk_conv1d_lds_bytes(ctypes.byref(two)) == -1 and b"out2 overlaps" in L.vk_last_error()
    # a 1x1 convolution with a single row block may run in place (each workgroup reads only what it owns)
    assert L.vk_conv1d_lds_bytes(ctypes.byref(_conv_args(out=0x10000000, taps=1, pad=0))) > 0
    # ... but not with several row blocks (another workgroup reads the same tile)
    assert L.vk_conv1d_lds_bytes(ctypes.byref(_conv_args(out=0x10000000, taps=1, pad=0, c_out=64))) == -1


def test_token_kernel_argument_validation_without_gpu():
    """The token-level entry points check their arguments on the host before any launch."""
    L = native.lib()
    P = 0x10000000                                                     # never dereferenced
    assert L.vk_channel_norm(None, None, None, None, None, 1, 192, 4, None) == -1
    assert L.vk_channel_norm(P, P, P, None, P, 1, 64, 4, None) == -2 and b"192" in L.vk_last_error()
    assert L.vk_rel_attention(P, P, P, P, P, 1, 2, 96, 4, 0, None) == -1
    assert L.vk_rel_attention(P, P, P, P, P, 1, 2, 64, 4, 10, None) == -2
    assert L.vk_rel_attention(P, P, P, P, P, 1, 2, 96, 4, 1025, None) == -2 and b"VK_ATT_MAX_LENGTH" in L.vk_last_error()
    assert L.vk_expand_prior(P, P, P, P, P, None, 1, 192, 10, 20, None) == -1
    assert L.vk_expand_prior(P, P, P, P, P, P, 1, 192, 4096, 20, None) == -2
    assert L.vk_dds_workspace_bytes(2, 100) == 2 * 2 * 192 * 100 * 4
    a = native.DdsArgs()
    assert L.vk_dds_condition(ctypes.byref(a), None) == -1 and b"null" in L.vk_last_error()
    for f in ("x", "pre_wt", "pre_b", "layers", "proj_wt", "proj_b", "mask"):
        setattr(a, f, P)
    a.out = 0x20000000
    a.batch, a.length, a.n_layers, a.taps = 1, 50, 3, 3
    a.x_bstride, a.x_cstride, a.out_bstride, a.out_cstride = 192 * 50, 50, 192 * 50, 50
    assert L.vk_dds_condition(ctypes.byref(a), None) == -4 and b"workspace" in L.vk_last_error()
    a.workspace, a.workspace_bytes = 0x30000000, 100                   # too small
    assert L.vk_dds_condition(ctypes.byref(a), None) == -4
    a.taps = 5
    assert L.vk_dds_condition(ctypes.byref(a), None) == -2
    a.taps = 3
    a.workspace_bytes = L.vk_dds_workspace_bytes(1, 50)
    a.out = P + 64                                                     # on top of the input
    assert L.vk_dds_condition(ctypes.byref(a), None) == -1 and b"overlaps" in L.vk_last_error()
    a.out = 0x20000000
    assert L.vk_dds_coupling(ctypes.byref(a), None) == -1 and b"add_map" in L.vk_last_error()


def test_vector_staging_falls_back_when_its_alignment_slack_does_not_fit():
    """halo 64 with an odd left padding: the 16-byte stager would need 68 extra columns, the
    4-byte stager fits in the 64-column slack of the LDS row."""
    L = native.lib()
    assert L.vk_conv1d_lds_bytes(ctypes.byref(_conv_args(taps=65, pad=32))) > 0
    assert L.vk_conv1d_lds_bytes(ctypes.byref(_conv_args(taps=65, pad=31))) > 0
    assert L.vk_conv1d_lds_bytes(ctypes.byref(_conv_args(taps=66, pad=32))) == -2       # halo 65 never fits


def test_cpu_tensors_fail_loudly():
    from vits_amd.module.model_component.decoder import Decoder
    dec = Decoder(speaker_id_embedding_dim=256)
    with pytest.raises(native.NativeError):
        dec(torch.zeros(1, 192, 4), torch.zeros(1, 256, 1))


def test_graft_entry_build_accepts_the_current_library():
    """The driver's build check: compiles nothing new when up to date, loads the library and
    agrees with the header's ABI version."""
    import __graft_entry__ as entry
    entry.build()
