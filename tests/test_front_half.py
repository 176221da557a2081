"""The layer-by-layer PyTorch form of the front half of text_to_speech (text encoder, stochastic
duration predictor in reverse, path expansion; what CPU tensors run, and what the GPU token kernels
are compared with in test_token_kernels.py) and the complete generator's state_dict layout,
against vectors captured from the reference (tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import torch

from oracle import vits_oracle as O
from vits_amd import synth
from vits_amd.module.vits_generator import VitsGenerator

TOL = 2e-5


def _gen():
    torch.set_grad_enabled(False)
    g = VitsGenerator(40, 100)
    g.load_state_dict(synth.full_state_dict(g.state_dict(), 999), strict=True)
    return g.eval()


def test_full_state_dict_layout_matches_reference(golden_dir):
    layout = json.load(open(os.path.join(golden_dir, "netg_state_dict_layout.json")))
    own = VitsGenerator(40, 100).state_dict()
    assert len(layout) == 858
    assert [(k, list(v.shape)) for k, v in own.items()] == [(k, s) for k, s in layout]
    assert sum(int(np.prod(s)) for _, s in layout) == 39_671_408


def test_text_encoder_duration_predictor_and_tts(golden_dir):
    g11 = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, "g11_tts.npz")).items()}
    gen = _gen()
    ids, spk = g11["ids"], g11["speaker"]
    lens = torch.tensor([ids.shape[1]])
    enc, m, logs, mask = gen.text_encoder(ids, lens)
    assert (enc - g11["text_encoded"]).abs().max() <= TOL * 4
    assert (m - g11["m"]).abs().max() <= TOL and (logs - g11["logs"]).abs().max() <= TOL
    logw = gen.stochastic_duration_predictor(enc, mask, speaker_id_embedded=gen.embed_speaker(spk), reverse=True,
                                             noise_scale=0.8, noise=g11["noise_w"])
    assert (logw - g11["logw"]).abs().max() <= 5e-5
    # frame-level prior from the own front half, tail through the CPU oracle == reference text_to_speech
    m_p, logs_p, spec_mask, g = gen.frame_prior(ids, lens, spk, length_scale=int(g11["length_scale"]),
                                                noise_w=g11["noise_w"])
    assert m_p.shape[2] * 256 == g11["wav"].shape[2]
    sd = synth.full_state_dict(gen.state_dict(), 999)
    _, wav = O.tts_tail(m_p, logs_p, spec_mask, g11["noise_z"], g, sd)
    assert (wav - g11["wav"]).abs().max() <= TOL


def test_spline_inverse_is_inverse_of_forward_spline():
    """Property check independent of the reference: pushing the inverse's output through the
    forward rational-quadratic map returns the input."""
    from vits_amd.module.model_component.stochastic_duration_predictor import (
        MIN_DERIVATIVE, _knots, inverse_rational_quadratic_spline)
    import math
    import torch.nn.functional as F
    torch.manual_seed(0)
    n, B = 10, 5.0
    y = torch.linspace(-7, 7, 400).view(1, 1, 400)
    uw, uh, ud = torch.randn(1, 1, 400, n), torch.randn(1, 1, 400, n), torch.randn(1, 1, 400, n - 1)
    x = inverse_rational_quadratic_spline(y, uw, uh, ud, B)
    xk, w = _knots(uw, -B, B, 1e-3)
    yk, h = _knots(uh, -B, B, 1e-3)
    d = 1e-3 + F.softplus(F.pad(ud, (1, 1), value=math.log(math.exp(1 - MIN_DERIVATIVE) - 1)))
    idx = ((x.clamp(-B, B).unsqueeze(-1) >= xk).sum(-1) - 1).clamp(0, n - 1).unsqueeze(-1)
    pick = lambda t: t.gather(-1, idx)[..., 0]
    th = (x.clamp(-B, B) - pick(xk)) / pick(w)
    delta = pick(h / w)
    num = pick(h) * (delta * th * th + pick(d) * th * (1 - th))
    den = delta + (pick(d) + pick(d[..., 1:]) - 2 * delta) * th * (1 - th)
    y_back = torch.where(x.abs() <= B, pick(yk) + num / den, x)
    assert (y_back - y).abs().max() < 1e-4
