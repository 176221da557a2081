"""bf16 matrix-core inputs (fp32 accumulation, fp32 tensors in memory) against the goldens
captured from the fp32 reference.  The stated tolerance of this mode is on the waveform:
max-abs error <= BF16_WAV_TOL for waveforms of amplitude ~0.5 (the fp32 path's is 1e-4), and
the relative RMS error <= BF16_WAV_RMS (measured on the MI355X: 2.8e-3 / 3.1e-3 / 4.2e-3 max-abs and
0.5-0.6 % relative RMS for the decoder, TTS tail and voice-conversion goldens).  Kernel exactness of the bf16 path (equality with an
fp32 convolution of bf16-rounded operands) is covered in test_conv_kernel.py."""
import os

import numpy as np
import pytest
import torch

from vits_amd import engine, synth
from vits_amd.module.model_component.decoder import Decoder
from vits_amd.module.vits_generator import HotPathGenerator

pytestmark = pytest.mark.gpu
SEED = 999
DEV = "cuda:0"
BF16_WAV_TOL = 1e-2
BF16_WAV_RMS = 1.5e-2


@pytest.fixture()
def bf16_mode():
    engine.set_precision("bf16")
    try:
        yield
    finally:
        engine.set_precision("fp32")


def _gold(golden_dir, name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, name)).items()}


def _report(got, ref, what):
    got = got.detach().cpu()
    assert got.shape == ref.shape and torch.isfinite(got).all(), what
    err = (got - ref).abs().max().item()
    rms = ((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    print(f"{what}: max-abs {err:.3e}, relative rms {rms:.3e}, |ref|max {ref.abs().max().item():.3f}")
    assert err <= BF16_WAV_TOL, f"{what}: max-abs {err:.3e}"
    assert rms <= BF16_WAV_RMS, f"{what}: relative rms {rms:.3e}"


def test_bf16_decoder_tts_and_vc(sd, golden_dir, bf16_mode):
    from oracle import vits_oracle as O
    g1 = _gold(golden_dir, "g1_decoder.npz")
    dec = Decoder(256)
    dec.load_state_dict(synth.strip_prefix(sd, "decoder"), strict=True)
    dec = dec.to(DEV).eval()
    z = synth.latent(SEED, "g1.z", 2, 24)
    spk = O.speaker_embed(torch.tensor([9, 57]), sd)
    _report(dec(z.to(DEV), spk.to(DEV)), g1["wav"], "bf16 decoder")

    gen = HotPathGenerator(100)
    gen.load_state_dict(sd, strict=True)
    gen = gen.to(DEV).eval()
    g8 = _gold(golden_dir, "g8_tts_tail.npz")
    T = 50
    mask = synth.length_mask([50, 37], T)
    m_p = synth.latent(SEED, "g8.m_p", 2, T) * 0.5
    logs_p = synth.latent(SEED, "g8.logs_p", 2, T) * 0.2 - 0.5
    eps = synth.latent(SEED, "g8.eps", 2, T)
    wav = gen.synthesize_from_prior(m_p.to(DEV), logs_p.to(DEV), mask.to(DEV), spk.to(DEV), noise=eps.to(DEV))
    _report(wav, g8["wav"], "bf16 tts tail")

    g7 = _gold(golden_dir, "g7_posterior.npz")
    g9 = _gold(golden_dir, "g9_vc.npz")
    spec = synth.power_spectrogram(SEED, "g7.spec", 2, 40)
    wav = gen.voice_conversion(spec.to(DEV), g7["lengths"].to(DEV), g9["src"].to(DEV), g9["tgt"].to(DEV),
                               noise=g7["eps"].to(DEV))
    _report(wav, g9["wav"], "bf16 voice conversion")


def test_split_mode_meets_the_fp32_contract(sd, golden_dir):
    """fp32 operands as three exact bf16 slices (six slice products, fp32 accumulation): held to
    the fp32 path's own bars, 1e-4 max-abs on the waveform and 2e-5 relative internally."""
    from oracle import vits_oracle as O
    engine.set_precision("fp32-split")
    try:
        g1 = _gold(golden_dir, "g1_decoder.npz")
        dec = Decoder(256)
        dec.load_state_dict(synth.strip_prefix(sd, "decoder"), strict=True)
        dec = dec.to(DEV).eval()
        z = synth.latent(SEED, "g1.z", 2, 24)
        spk = O.speaker_embed(torch.tensor([9, 57]), sd)
        gen = HotPathGenerator(100)
        gen.load_state_dict(sd, strict=True)
        gen = gen.to(DEV).eval()
        g7 = _gold(golden_dir, "g7_posterior.npz")
        g9 = _gold(golden_dir, "g9_vc.npz")
        spec = synth.power_spectrogram(SEED, "g7.spec", 2, 40)
        cases = [("decoder", dec(z.to(DEV), spk.to(DEV)), g1["wav"]),
                 ("voice conversion", gen.voice_conversion(spec.to(DEV), g7["lengths"].to(DEV), g9["src"].to(DEV),
                                                           g9["tgt"].to(DEV), noise=g7["eps"].to(DEV)), g9["wav"])]
        zl = synth.latent(SEED, "split.z", 2, 300)
        gl = O.speaker_embed(torch.tensor([3, 71]), sd)
        cases.append(("decoder T=300", dec(zl.to(DEV), gl.to(DEV)), O.decoder(zl, gl, sd)))
        for what, got, ref in cases:
            got = got.cpu()
            err = (got - ref).abs().max().item()
            print(f"fp32-split {what}: max-abs {err:.3e} (|ref|max {ref.abs().max().item():.3f})")
            assert torch.isfinite(got).all()
            assert err <= 1e-4 and err <= 2e-5 * max(1.0, ref.abs().max().item()), (what, err)
    finally:
        engine.set_precision("fp32")


def test_precision_switch_rebuilds_plans(sd, golden_dir):
    """The same module object serves both precisions; switching back restores the fp32 result."""
    g1 = _gold(golden_dir, "g1_decoder.npz")
    from oracle import vits_oracle as O
    dec = Decoder(256)
    dec.load_state_dict(synth.strip_prefix(sd, "decoder"), strict=True)
    dec = dec.to(DEV).eval()
    z = synth.latent(SEED, "g1.z", 2, 24).to(DEV)
    spk = O.speaker_embed(torch.tensor([9, 57]), sd).to(DEV)
    a = dec(z, spk).cpu()
    engine.set_precision("bf16")
    try:
        b = dec(z, spk).cpu()
    finally:
        engine.set_precision("fp32")
    c = dec(z, spk).cpu()
    assert torch.equal(a, c)
    assert not torch.equal(a, b)
    assert (a - g1["wav"]).abs().max().item() <= 1e-4


def test_bf16_voice_conversion_at_full_size(sd, bf16_mode):
    """BASELINE config 4 at its real size (64 utterances of 862 frames, bf16 matrix inputs):
    first, middle and last item against the CPU oracle run on each item alone (items of a batch
    do not interact anywhere on the path), plus ragged lengths in the same batch."""
    from oracle import vits_oracle as O
    B, T = 64, 862
    gen = HotPathGenerator(100)
    gen.load_state_dict(sd, strict=True)
    gen = gen.to(DEV).eval()
    spec = synth.power_spectrogram(SEED, "cfg4.spec", B, T)
    lens = torch.full((B,), T, dtype=torch.int64)
    lens[37], lens[63] = 700, 431                                  # two shorter items
    spec = spec * synth.length_mask(lens, T)
    src, tgt = synth.speaker_ids(SEED, "cfg4.src", B), synth.speaker_ids(SEED, "cfg4.tgt", B)
    eps = synth.latent(SEED, "cfg4.eps", B, T)
    wav = gen.voice_conversion(spec.to(DEV), lens.to(DEV), src.to(DEV), tgt.to(DEV), noise=eps.to(DEV)).cpu()
    assert wav.shape == (B, 1, T * 256) and torch.isfinite(wav).all()
    for i in (0, 37, 63):
        ref = O.voice_conversion(spec[i:i + 1], lens[i:i + 1], src[i:i + 1], tgt[i:i + 1], sd, eps[i:i + 1])
        _report(wav[i:i + 1], ref, f"bf16 voice conversion B=64, item {i}")      # the whole padded item
