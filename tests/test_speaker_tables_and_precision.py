"""Host-side contracts around the kernels that do not change any arithmetic:

* speaker conditioning read as rows of a per-speaker table (one table per module, built once per
  weights) gives bit for bit what the per-call embedding gather + GEMV gives, runs no GEMV in
  steady state, and follows the weights;
* `z * spec_mask` in front of the decoder (/root/reference/module/vits_generator.py:186,196) taken
  while the first convolution stages its input equals the explicit product;
* the flow never writes to, or copies, its input;
* two generators in one process keep their own matrix-core precision;
* the token-level graph of `text_to_speech` is retired by an in-place load of other weights;
* the small-batch decoder captured without a warm-up run (branch streams first meet their split
  channel loops inside the capture) replays to the eager result.
"""
import pytest
import torch

from vits_amd import engine, native, synth
from vits_amd.module.vits_generator import HotPathGenerator, VitsGenerator

SEED = 999
DEV = "cuda:0"
pytestmark = pytest.mark.gpu


def _gen(seed=SEED, with_posterior=False):
    torch.set_grad_enabled(False)
    gen = HotPathGenerator(100, with_posterior=with_posterior)
    gen.load_state_dict(synth.hot_path_state_dict(seed, with_posterior=with_posterior), strict=True)
    return gen.to(DEV).eval()


def _inputs(B=3, T=37):
    lens = [T, T - 9, T - 20][:B]
    mask = synth.length_mask(lens, T)
    z = synth.latent(SEED, "tables.z", B, T) * mask
    ids = torch.tensor([5, 71, 5][:B])
    return z.to(DEV), mask.to(DEV), ids.to(DEV)


def _launches(fn):
    with native.LaunchTimer() as t:
        out = fn()
    torch.cuda.synchronize()
    return out, t.summary()


@pytest.mark.parametrize("prec", ["fp32", "bf16", "fp32-split"])
def test_table_rows_equal_the_per_call_conditioning(prec):
    gen = _gen().set_precision(prec)
    z, mask, ids = _inputs()
    want = gen.flow_reverse_decode(z, mask, gen.embed_speaker(ids))
    got = gen.flow_reverse_decode(z, mask, gen.speaker_ref(ids))          # builds the tables
    assert torch.equal(got, want)
    got, fams = _launches(lambda: gen.flow_reverse_decode(z, mask, gen.speaker_ref(ids)))
    assert torch.equal(got, want)
    assert "speaker_linear" not in fams, sorted(fams)
    _, fams = _launches(lambda: gen.flow_reverse_decode(z, mask, gen.embed_speaker(ids)))
    assert fams["speaker_linear"]["launches"] == 2                       # flow (all layers at once) + decoder


def test_one_launch_wn_layer_reads_table_rows():
    """Batch x frames inside the range where a WN layer is one launch (fp32)."""
    gen = _gen().set_precision("fp32")
    B, T = 8, 640
    mask = synth.length_mask([T - 7 * i for i in range(B)], T).to(DEV)
    z = (synth.latent(SEED, "tables.wn", B, T).to(DEV)) * mask
    ids = torch.arange(B, device=DEV) * 11 % 100
    want = gen.flow(z, mask, gen.embed_speaker(ids), reverse=True)
    got, fams = _launches(lambda: gen.flow(z, mask, gen.speaker_ref(ids), reverse=True))
    assert any(k.startswith("wn_layer") for k in fams), sorted(fams)
    assert torch.equal(got, want)


def test_tables_follow_the_weights():
    """A new embedding table, and new weights of the conditioning layers, both show in the next call."""
    gen = _gen()
    z, mask, ids = _inputs()
    gen.flow_reverse_decode(z, mask, gen.speaker_ref(ids))
    first = gen.decoder._plans.get(torch.device(DEV)).speakers._table
    with torch.no_grad():
        gen.speaker_embedding.weight.mul_(0.5)                          # in place: same storage, new version
    want = gen.flow_reverse_decode(z, mask, gen.embed_speaker(ids))
    got = gen.flow_reverse_decode(z, mask, gen.speaker_ref(ids))
    assert torch.equal(got, want)
    second = gen.decoder._plans.get(torch.device(DEV)).speakers._table
    assert second is not first
    other = synth.hot_path_state_dict(123, with_posterior=False)
    gen.load_state_dict(other, strict=True)                             # in place too
    fresh = _gen(123)
    want = fresh.flow_reverse_decode(z, mask, fresh.embed_speaker(ids))
    got = gen.flow_reverse_decode(z, mask, gen.speaker_ref(ids))
    assert torch.equal(got, want)
    assert gen.decoder._plans.get(torch.device(DEV)).speakers._table is not second


def test_out_of_range_speaker_id_is_clamped_into_the_table():
    gen = _gen()
    z, mask, ids = _inputs(B=2)
    hi = gen.flow_reverse_decode(z, mask, gen.speaker_ref(torch.tensor([99, 0], device=DEV)))
    bad = gen.flow_reverse_decode(z, mask, gen.speaker_ref(torch.tensor([1000, -3], device=DEV)))
    assert torch.equal(hi, bad)


@pytest.mark.parametrize("prec", ["fp32", "bf16", "fp32-split"])
def test_decoder_input_mask_equals_the_explicit_product(prec):
    gen = _gen().set_precision(prec)
    z, mask, ids = _inputs()
    soft = (mask * torch.linspace(0.25, 1.0, z.shape[2], device=DEV)).contiguous()   # not 0/1 on purpose
    g = gen.embed_speaker(ids)
    want = gen.decoder(z * soft, g)
    got = gen.decoder(z, g, z_mask=soft)
    assert torch.equal(got, want)
    cut = z.shape[2] - 5                                               # (z * mask)[:, :, :max_len] as a view
    want = gen.decoder((z * soft)[:, :, :cut].contiguous(), g)
    got = gen.decoder(z[:, :, :cut], g, z_mask=soft)
    assert torch.equal(got, want)


def test_flow_leaves_its_input_alone_and_copies_nothing():
    gen = _gen()
    z, mask, ids = _inputs()
    before = z.clone()
    for reverse in (True, False):
        out = gen.flow(z, mask, gen.speaker_ref(ids), reverse=reverse)
        assert torch.equal(z, before) and out.data_ptr() != z.data_ptr()
    # forward then reverse is the identity on the valid frames (to rounding)
    back = gen.flow(gen.flow(z, mask, gen.speaker_ref(ids)), mask, gen.speaker_ref(ids), reverse=True)
    assert (back - z).abs().max() < 1e-4


def test_two_generators_in_two_precisions_side_by_side():
    z, mask, ids = _inputs()
    exact = _gen()
    want32 = exact.flow_reverse_decode(z, mask, exact.speaker_ref(ids))
    engine.set_precision("bf16")
    try:
        alone16 = _gen()
        want16 = alone16.flow_reverse_decode(z, mask, alone16.speaker_ref(ids))
    finally:
        engine.set_precision("fp32")
    a, b = _gen().set_precision("fp32"), _gen().set_precision("bf16")
    for _ in range(2):                                                 # interleaved calls, either order
        got16 = b.flow_reverse_decode(z, mask, b.speaker_ref(ids))
        got32 = a.flow_reverse_decode(z, mask, a.speaker_ref(ids))
        assert torch.equal(got32, want32)
        assert torch.equal(got16, want16)
    assert not torch.equal(want16, want32) and (want16 - want32).abs().max() < 1e-2
    # the process default does not move a model that has its own setting, and the other way round
    engine.set_precision("bf16")
    try:
        assert torch.equal(a.flow_reverse_decode(z, mask, a.speaker_ref(ids)), want32)
    finally:
        engine.set_precision("fp32")
    assert torch.equal(b.flow_reverse_decode(z, mask, b.speaker_ref(ids)), want16)
    b.set_precision(None)                                              # back to the process default
    assert torch.equal(b.flow_reverse_decode(z, mask, b.speaker_ref(ids)), want32)


def test_in_place_load_retires_the_token_level_graph():
    """text_to_speech, in-place load of other weights, text_to_speech: the second waveform is the
    one a fresh module with the other weights gives (the captured kernels read packed copies of the
    parameters, which an in-place load leaves in their storage)."""
    torch.set_grad_enabled(False)
    gen = VitsGenerator(40, 100)
    sd_a = synth.full_state_dict(gen.state_dict(), SEED)
    sd_b = synth.full_state_dict(gen.state_dict(), 4242)
    gen.load_state_dict(sd_a, strict=True)
    gen = gen.to(DEV).eval()
    gen.use_front_graph = True
    ids = torch.tensor([synth.TTS_SENTENCE_IDS], device=DEV)
    lens = torch.tensor([ids.shape[1]], device=DEV)
    spk = torch.tensor([9], device=DEV)
    g = torch.Generator().manual_seed(3)
    nw = torch.randn(1, 2, ids.shape[1], generator=g).to(DEV)

    def speak(model):
        m_p, logs_p, mask, ref = model.frame_prior(ids, lens, spk, length_scale=3, noise_w=nw)
        nz = torch.ones_like(m_p) * 0.25
        return model.synthesize_from_prior(m_p, logs_p, mask, ref, noise=nz)

    first = speak(gen)
    assert len(gen._front_graphs) == 1
    gen.load_state_dict(sd_b, strict=True)                             # in place: every data_ptr stays
    second = speak(gen)
    fresh = VitsGenerator(40, 100)
    fresh.load_state_dict(sd_b, strict=True)
    fresh = fresh.to(DEV).eval()
    fresh.use_front_graph = False
    want = speak(fresh)
    assert second.shape == want.shape and torch.equal(second, want)
    assert first.shape != second.shape or not torch.equal(first, second)


def test_small_decoder_captured_without_a_warm_up_run():
    """One short utterance: the resblocks of a stage run on side streams and the launches split
    their channel loops.  Captured with no eager run of this shape before the capture, every stream
    that can overlap with another must still have arrival counters of its own."""
    from vits_amd.graphs import GraphedCall
    gen = _gen()
    dev = torch.device(DEV)
    z = synth.latent(SEED, "capture.z", 1, 64).to(dev)
    g = gen.embed_speaker(torch.tensor([9], device=dev))
    other = synth.latent(SEED, "capture.z2", 1, 96).to(dev)
    gen.decoder(other, g)                                              # plans and kernel attributes only (another shape)
    torch.cuda.synchronize()
    call = GraphedCall(lambda zz, gg: gen.decoder(zz, gg), z, g, warmup=0)
    want = gen.decoder(z, g)
    for _ in range(5):
        got = call(z, g)
        assert torch.equal(got, want)
