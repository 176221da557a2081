"""GPU parity of the token-level kernels (duration predictor stacks and coupling flows, through the
C ABI) against the same modules' layer-by-layer PyTorch form on the CPU, which the CPU suite pins
to the reference's goldens (test_front_half.py)."""
import pytest
import torch

from vits_amd import native, synth
from vits_amd.module.vits_generator import VitsGenerator

pytestmark = pytest.mark.gpu
SEED = 999
DEV = "cuda:0"


@pytest.fixture(scope="module")
def gens():
    cpu = VitsGenerator(40, 100)
    sd = synth.full_state_dict(cpu.state_dict(), SEED)
    cpu.load_state_dict(sd, strict=True)
    gpu = VitsGenerator(40, 100)
    gpu.load_state_dict(sd, strict=True)
    return cpu.eval(), gpu.to(DEV).eval()


def _inputs(B, T, lens, seed):
    g = torch.Generator().manual_seed(seed)
    enc = torch.randn(B, 192, T, generator=g)
    mask = synth.length_mask(lens, T)
    spk = torch.randn(B, 256, 1, generator=g) * 0.3
    noise = torch.randn(B, 2, T, generator=g)
    return enc * mask, mask, spk, noise


@pytest.mark.parametrize("B,T,lens", [(1, 43, [43]), (3, 87, [87, 60, 1]), (2, 200, [200, 131]), (1, 1, [1]),
                                      (2, 64, [64, 33]), (1, 65, [65]), (2, 333, [333, 77])])
def test_duration_predictor_reverse(gens, B, T, lens):
    """One window (T <= 64), window edges (64 / 65) and several windows with recomputed reach;
    ragged masks; compared with the CPU layer-by-layer pass on the same noise."""
    cpu, gpu = gens
    enc, mask, spk, noise = _inputs(B, T, lens, 100 + T)
    want = cpu.stochastic_duration_predictor(enc, mask, speaker_id_embedded=spk, reverse=True, noise_scale=0.8,
                                             noise=noise)
    sdp = gpu.stochastic_duration_predictor
    assert sdp.use_token_kernels
    got = sdp(enc.to(DEV), mask.to(DEV), speaker_id_embedded=spk.to(DEV), reverse=True, noise_scale=0.8,
              noise=noise.to(DEV))
    assert got.shape == want.shape
    err = (got.cpu() - want).abs().max().item()
    assert err <= 2e-5 * max(1.0, want.abs().max().item()), err
    # the stock PyTorch form on the GPU agrees too (the switch the module keeps for comparison)
    sdp.use_token_kernels = False
    try:
        stock = sdp(enc.to(DEV), mask.to(DEV), speaker_id_embedded=spk.to(DEV), reverse=True, noise_scale=0.8,
                    noise=noise.to(DEV))
    finally:
        del sdp.use_token_kernels
    assert (stock.cpu() - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())


def test_duration_predictor_without_speaker(gens):
    cpu, gpu = gens
    enc, mask, _, noise = _inputs(2, 50, [50, 20], 7)
    want = cpu.stochastic_duration_predictor(enc, mask, reverse=True, noise_scale=1.0, noise=noise)
    got = gpu.stochastic_duration_predictor(enc.to(DEV), mask.to(DEV), reverse=True, noise_scale=1.0,
                                            noise=noise.to(DEV))
    assert (got.cpu() - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())


def test_spline_tails_are_identity(gens):
    """Noise far outside the spline's [-5, 5] support passes through the coupling flows unchanged
    apart from the final affine."""
    cpu, gpu = gens
    enc, mask, spk, noise = _inputs(1, 30, [30], 3)
    noise = noise * 40.0
    want = cpu.stochastic_duration_predictor(enc, mask, speaker_id_embedded=spk, reverse=True, noise_scale=1.0,
                                             noise=noise)
    got = gpu.stochastic_duration_predictor(enc.to(DEV), mask.to(DEV), speaker_id_embedded=spk.to(DEV),
                                            reverse=True, noise_scale=1.0, noise=noise.to(DEV))
    assert (got.cpu() - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())


def test_dds_rejects_bad_arguments(gens):
    _, gpu = gens
    plan = gpu.stochastic_duration_predictor._plans.get(torch.device(DEV))
    c = plan.couplings[0]
    z = torch.zeros(1, 2, 20, device=DEV)
    g = torch.zeros(1, 192, 20, device=DEV)
    mask = torch.ones(1, 20, device=DEV)
    kw = dict(pass_channel=0, num_bins=10, tail_bound=5.0)
    with pytest.raises(native.NativeError, match="overlaps"):
        native.dds_coupling(z, c["pre_w"], c["pre_b"], g, c["layers"], 3, 3, c["proj_wt"], c["proj_b"], mask, z, **kw)
    out = torch.empty_like(z)
    with pytest.raises(native.NativeError, match="num_bins"):
        native.dds_coupling(z, c["pre_w"], c["pre_b"], g, c["layers"], 3, 3, c["proj_wt"], c["proj_b"], mask, out,
                            pass_channel=0, num_bins=8, tail_bound=5.0)
    with pytest.raises(native.NativeError, match="taps"):
        native.dds_coupling(z, c["pre_w"], c["pre_b"], g, c["layers"], 3, 5, c["proj_wt"], c["proj_b"], mask, out, **kw)
    with pytest.raises(native.NativeError, match="pass_channel"):
        native.dds_coupling(z, c["pre_w"], c["pre_b"], g, c["layers"], 3, 3, c["proj_wt"], c["proj_b"], mask, out,
                            pass_channel=2, num_bins=10, tail_bound=5.0)


@pytest.mark.parametrize("B,T,lens", [(1, 43, [43]), (3, 87, [87, 60, 1]), (2, 200, [200, 131]), (1, 1, [1]),
                                      (2, 8, [8, 3]), (1, 9, [9]), (2, 33, [33, 32])])
def test_text_encoder(gens, B, T, lens):
    """Encoder layers on the library's kernels against the CPU layer-by-layer pass: hidden states
    and prior statistics at the valid positions, zeros at the padded ones."""
    cpu, gpu = gens
    g = torch.Generator().manual_seed(T)
    ids = torch.randint(0, 40, (B, T), generator=g)
    lt = torch.tensor(lens)
    want = cpu.text_encoder(ids, lt)
    te = gpu.text_encoder
    assert te.use_token_kernels and te._kernels_apply()
    got = te(ids.to(DEV), lt.to(DEV))
    for name, a, b in zip(("text_encoded", "m", "logs", "mask"), got, want):
        assert a.shape == b.shape, name
        err = (a.cpu() - b).abs().max().item()
        assert err <= 2e-5 * max(1.0, b.abs().max().item()), (name, err)


def test_rel_attention_against_the_module(gens):
    """The attention kernel alone against MultiHeadAttention.forward on the same projected inputs,
    with a ragged mask and more keys than one staged block."""
    cpu, gpu = gens
    att = cpu.text_encoder.encoder.attention_layers[2]
    B, T = 2, 77
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, 192, T, generator=g)
    mask = synth.length_mask([77, 40], T)
    attn_mask = mask.unsqueeze(2) * mask.unsqueeze(-1)
    want = att(x, x, attn_mask)
    q, k, v = att.conv_q(x), att.conv_k(x), att.conv_v(x)
    qkv = torch.cat([q, k, v], 1).contiguous().to(DEV)
    out = torch.empty(B, 192, T, device=DEV)
    native.rel_attention(qkv, att.emb_rel_k[0].contiguous().to(DEV), att.emb_rel_v[0].contiguous().to(DEV),
                         mask.reshape(B, T).contiguous().to(DEV), out, heads=2, window=4)
    got = att.conv_o(out.cpu())
    valid = mask.bool().expand_as(want)
    assert (got - want)[valid].abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())
    with pytest.raises(native.NativeError, match="VK_ATT_MAX_LENGTH"):
        native.rel_attention(torch.zeros(1, 576, 1025, device=DEV), att.emb_rel_k[0].contiguous().to(DEV),
                             att.emb_rel_v[0].contiguous().to(DEV), torch.ones(1, 1025, device=DEV),
                             torch.empty(1, 192, 1025, device=DEV), heads=2, window=4)


def test_channel_norm():
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 192, 21, generator=g) * 3 + 1.5
    gamma, beta = torch.randn(192, generator=g), torch.randn(192, generator=g)
    mask = synth.length_mask([21, 20, 2], 21).reshape(3, 21)
    want = torch.nn.functional.layer_norm(x.transpose(1, 2), (192,), gamma, beta).transpose(1, 2) * mask.unsqueeze(1)
    out = torch.empty(3, 192, 21, device=DEV)
    native.channel_norm(x.to(DEV), gamma.to(DEV), beta.to(DEV), mask.to(DEV), out)
    assert (out.cpu() - want).abs().max().item() <= 1e-5
    with pytest.raises(native.NativeError, match="192"):
        native.channel_norm(torch.zeros(1, 64, 4, device=DEV), gamma.to(DEV), beta.to(DEV), None,
                            torch.empty(1, 64, 4, device=DEV))


def test_expand_prior_matches_the_alignment_matrix(gens):
    """The gather against the reference's construction (0/1 path matrix times the statistics),
    bit for bit: ragged token counts, zero-length tokens, an item with no frames at all."""
    from vits_amd.module.vits_generator import generate_path, sequence_mask
    g = torch.Generator().manual_seed(21)
    B, Tx, C = 4, 300, 192
    tmask = synth.length_mask([300, 120, 7, 40], Tx)
    w = torch.randint(0, 6, (B, 1, Tx), generator=g).float() * tmask
    w[3] = 0                                             # nothing to say: one silent frame
    m = torch.randn(B, C, Tx, generator=g)
    logs = torch.randn(B, C, Tx, generator=g)
    y_len = torch.clamp_min(w.sum([1, 2]), 1).long()
    smask = sequence_mask(y_len, None).unsqueeze(1).float()
    path = generate_path(w, tmask.unsqueeze(2) * smask.unsqueeze(-1)).squeeze(1)
    want_m = torch.matmul(path, m.transpose(1, 2)).transpose(1, 2)
    want_l = torch.matmul(path, logs.transpose(1, 2)).transpose(1, 2)
    got_m, got_l, fmask = native.expand_prior(w.reshape(B, Tx).to(DEV), m.to(DEV), logs.to(DEV), int(y_len.max()))
    assert torch.equal(got_m.cpu(), want_m) and torch.equal(got_l.cpu(), want_l)
    assert torch.equal(fmask.cpu(), smask.reshape(B, -1))
    with pytest.raises(native.NativeError, match="VK_EXPAND_MAX_TOKENS"):
        native.expand_prior(torch.ones(1, 2049, device=DEV), torch.zeros(1, 8, 2049, device=DEV),
                            torch.zeros(1, 8, 2049, device=DEV), 2049)


def test_frame_prior_with_and_without_the_expand_kernel(gens):
    _, gpu = gens
    ids = torch.tensor([synth.TTS_SENTENCE_IDS], device=DEV)
    lens = torch.tensor([ids.shape[1]], device=DEV)
    spk = torch.tensor([9], device=DEV)
    nw = torch.randn(1, 2, ids.shape[1], generator=torch.Generator().manual_seed(1)).to(DEV)
    a = gpu.frame_prior(ids, lens, spk, length_scale=3, noise_w=nw)
    gpu.use_expand_kernel = False
    try:
        b = gpu.frame_prior(ids, lens, spk, length_scale=3, noise_w=nw)
    finally:
        del gpu.use_expand_kernel
    for x, y in zip(a[:3], b[:3]):                  # (the fourth entry names the speakers)
        assert x.shape == y.shape and torch.equal(x, y)


def test_ragged_batch_text_to_speech_front(gens):
    """Three sentences of different lengths in one batch: durations identical to the CPU
    layer-by-layer pass, frame-level prior within the fp32 gate, frame masks identical; then the
    whole text_to_speech against the CPU front half + oracle tail on the first item."""
    from oracle import vits_oracle as O
    cpu, gpu = gens
    g = torch.Generator().manual_seed(77)
    lens = [43, 20, 7]
    ids = torch.randint(0, 40, (3, 43), generator=g)
    ids[0] = torch.tensor(synth.TTS_SENTENCE_IDS)
    lt = torch.tensor(lens)
    spk = torch.tensor([9, 57, 3])
    nw = torch.randn(3, 2, 43, generator=g)
    want = cpu.frame_prior(ids, lt, spk, length_scale=2, noise_w=nw)
    got = gpu.frame_prior(ids.to(DEV), lt.to(DEV), spk.to(DEV), length_scale=2, noise_w=nw.to(DEV))
    assert got[0].shape == want[0].shape, "a duration rounded differently"
    assert torch.equal(got[2].cpu(), want[2])
    for name, a, b in zip(("m_p", "logs_p"), got[:2], want[:2]):
        assert (a.cpu() - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item()), name
    nz = torch.randn(want[0].shape, generator=g)
    wav = gpu.text_to_speech(ids.to(DEV), lt.to(DEV), spk.to(DEV), length_scale=2, noise_w=nw.to(DEV),
                             noise_z=nz.to(DEV))
    sd = cpu.state_dict()
    _, ref = O.tts_tail(want[0], want[1], want[2], nz, want[3], sd)
    assert wav.shape == ref.shape
    assert (wav.cpu() - ref).abs().max().item() <= 1e-4


def test_text_encoder_beyond_the_attention_limit_uses_the_pytorch_form(gens):
    """More tokens than the attention kernel holds in LDS: the layer-by-layer form runs instead."""
    cpu, gpu = gens
    T = native.ATT_MAX_LENGTH + 6
    ids = torch.randint(0, 40, (1, T), generator=torch.Generator().manual_seed(2))
    lt = torch.tensor([T])
    want = cpu.text_encoder(ids, lt)
    got = gpu.text_encoder(ids.to(DEV), lt.to(DEV))
    for a, b in zip(got, want):
        assert (a.cpu() - b).abs().max().item() <= 5e-5 * max(1.0, b.abs().max().item())
