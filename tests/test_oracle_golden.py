"""The CPU restatement (oracle/) replayed against golden vectors captured from the
reference modules by tests/golden/make_golden.py.  Runs anywhere (no GPU, no reference).
The capture asserted bit-equality in the container; here a small tolerance absorbs
differences in host CPU / oneDNN code paths between machines."""
import os

import numpy as np
import torch

from oracle import vits_oracle as O
from vits_amd import synth

SEED = 999
TOL = 5e-6


def _load(golden_dir, name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, name)).items()}


def _close(a, b, tol=TOL):
    err = (a - b).abs().max().item()
    assert err <= tol * max(1.0, b.abs().max().item()), err


def test_decoder_and_stage_taps(sd, golden_dir):
    g1 = _load(golden_dir, "g1_decoder.npz")
    z = synth.latent(SEED, "g1.z", 2, 24)
    g = O.speaker_embed(g1["speakers"], sd)
    taps = {}
    wav = O.decoder(z, g, sd, taps=taps)
    _close(wav, g1["wav"])
    for i in range(4):
        _close(taps[f"up{i}"][0, :, :256], g1[f"up{i}_head"])
        _close(taps[f"mrf{i}"][0, :, :256], g1[f"mrf{i}_head"])


def test_resblocks(sd, golden_dir):
    g3 = _load(golden_dir, "g3_resblocks.npz")
    for j, k in enumerate((3, 7, 11)):
        x = synth.latent(SEED, f"g3.x{k}", 1, 512, channels=32)
        _close(O.resblock(x, sd, f"decoder.resblocks.{9 + j}", k), g3[f"y{k}"])


def test_wn_ragged(sd, golden_dir):
    g4 = _load(golden_dir, "g4_wn.npz")
    mask = synth.length_mask([48, 31], 48)
    g = O.speaker_embed(torch.tensor([9, 57]), sd)
    for n, pfx in ((4, "flow.flows.0.wn"), (16, "posterior_encoder.wn")):
        x = synth.latent(SEED, f"g4.x{n}", 2, 48) * mask
        _close(O.wn(x, mask, g, sd, pfx, n), g4[f"y{n}"])


def test_coupling_and_flow(sd, golden_dir):
    mask = synth.length_mask([48, 31], 48)
    g = O.speaker_embed(torch.tensor([9, 57]), sd)
    g5 = _load(golden_dir, "g5_coupling.npz")
    x = synth.latent(SEED, "g5.x", 2, 48)
    _close(O.coupling(x, mask, g, sd, "flow.flows.2", False), g5["fwd"])
    _close(O.coupling(x, mask, g, sd, "flow.flows.2", True), g5["rev"])
    g6 = _load(golden_dir, "g6_flow.npz")
    x = synth.latent(SEED, "g6.x", 2, 48)
    _close(O.flow(x, mask, g, sd, reverse=False), g6["fwd"])
    _close(O.flow(x, mask, g, sd, reverse=True), g6["rev"])
    # invertibility on the valid region
    xm = x * mask
    rt = O.flow(O.flow(xm, mask, g, sd), mask, g, sd, reverse=True)
    _close(rt, xm, 1e-5)


def test_posterior_and_voice_conversion(sd, golden_dir):
    g7 = _load(golden_dir, "g7_posterior.npz")
    spec = synth.power_spectrogram(SEED, "g7.spec", 2, 40)
    g = O.speaker_embed(torch.tensor([9, 57]), sd)
    z, m, logs, mask = O.posterior_encoder(spec, g7["lengths"], g, sd, g7["eps"])
    _close(m, g7["m"])
    _close(logs, g7["logs"])
    _close(z, g7["z"], 2e-5)
    g9 = _load(golden_dir, "g9_vc.npz")
    wav = O.voice_conversion(spec, g7["lengths"], g9["src"], g9["tgt"], sd, g7["eps"])
    _close(wav, g9["wav"])


def test_tts_tail(sd, golden_dir):
    g8 = _load(golden_dir, "g8_tts_tail.npz")
    T = 50
    mask = synth.length_mask([50, 37], T)
    g = O.speaker_embed(torch.tensor([9, 57]), sd)
    m_p = synth.latent(SEED, "g8.m_p", 2, T) * 0.5
    logs_p = synth.latent(SEED, "g8.logs_p", 2, T) * 0.2 - 0.5
    eps = synth.latent(SEED, "g8.eps", 2, T)
    z, wav = O.tts_tail(m_p, logs_p, mask, eps, g, sd)
    _close(z, g8["z"])
    _close(wav, g8["wav"])


def test_full_size_decoder_checksums(sd, golden_dir):
    g10 = np.load(os.path.join(golden_dir, "g10_decoder_full_checksums.npz"))
    z = synth.latent(SEED, "cfg2.z", 1, 862)
    wav = O.decoder(z, O.speaker_embed(torch.tensor([9]), sd), sd)
    assert wav.shape == (1, 1, 862 * 256)
    assert abs(wav.double().sum().item() - float(g10["sum"])) < 1e-2
    assert abs((wav.double() ** 2).sum().item() - float(g10["sumsq"])) < 1e-2
    np.testing.assert_allclose(wav.flatten()[torch.from_numpy(g10["idx"])].numpy(), g10["samples"],
                               atol=TOL)
