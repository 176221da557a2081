#!/usr/bin/env python3
"""Times the benchmark configurations other than the headline one (needs a GPU):
  1  text_to_speech, one ~3 s sentence: CPU oracle path vs the drop-in generator on the GPU
  2  decoder only, B=1, T=862 and T=258 (latency), eager and replayed as one HIP graph
  4  voice conversion (posterior encoder -> flow fwd -> flow rev -> decoder), B=64, T=862, in every
     precision mode
``run()`` returns them as a dict (bench.py reports it as ``other_configs``); as a script it prints
one line per case.  (Config 3 is bench.py's timed step, config 5 its ``corpus`` object.)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HOP, SAMPLE_RATE = 256, 22050


def _timed(fn, n, sync=True):
    fn()
    if sync:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    if sync:
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


def run(dev, seed: int = 999, vc_batch: int = 64, cpu_reps: int = 2):
    from oracle import vits_oracle as O                                   # CPU baseline and checker only
    from vits_amd import engine, synth
    from vits_amd.graphs import GraphedCall
    from vits_amd.module.vits_generator import VitsGenerator

    torch.set_grad_enabled(False)
    before = engine.precision_name()
    gen = VitsGenerator(40, 100)
    sd = synth.full_state_dict(gen.state_dict(), seed)
    gen.load_state_dict(sd, strict=True)
    cpu_gen = VitsGenerator(40, 100)
    cpu_gen.load_state_dict(sd, strict=True)
    gen = gen.to(dev).eval()
    out = {}

    # ---- config 1: one sentence end to end
    ids = torch.tensor([synth.TTS_SENTENCE_IDS]); lens = torch.tensor([ids.shape[1]]); spk = torch.tensor([9])
    torch.manual_seed(1)
    nw = torch.randn(1, 2, ids.shape[1])
    m_p, logs_p, mask, g = cpu_gen.frame_prior(ids, lens, spk, length_scale=3, noise_w=nw)
    nz = torch.randn_like(m_p)
    audio = m_p.shape[2] * HOP / SAMPLE_RATE
    dt_cpu, (_, ref) = _timed(lambda: O.tts_tail(*cpu_gen.frame_prior(ids, lens, spk, length_scale=3, noise_w=nw)[:3],
                                                 nz, g, sd), cpu_reps, sync=False)
    a = [t.to(dev) for t in (ids, lens, spk, nw, nz)]
    c1 = {"tokens": int(ids.shape[1]), "audio_s": audio,
          "cpu_ms": 1e3 * dt_cpu, "cpu_value": audio / dt_cpu, "cpu_threads": torch.get_num_threads()}
    for prec in ("fp32", "fp32-split"):
        engine.set_precision(prec)
        dt, wav = _timed(lambda: gen.text_to_speech(a[0], a[1], a[2], length_scale=3, noise_w=a[3], noise_z=a[4]), 10)
        c1[prec] = {"ms": 1e3 * dt, "value": audio / dt,
                    "parity_max_abs": float((wav.cpu() - ref).abs().max()) if wav.shape == ref.shape else None}
    out["config1_text_to_speech"] = c1

    # ---- config 2: decoder only, one utterance
    gd = gen.speaker_ref(torch.tensor([9], device=dev))       # conditioning as a row of the per-speaker table
    c2 = {}
    for prec in ("fp32", "fp32-split"):
        engine.set_precision(prec)
        for T in (862, 258):
            z = synth.latent(seed, "cfg2.z", 1, T).to(dev)
            dt, _ = _timed(lambda: gen.decoder(z, gd), 20)
            graphed = GraphedCall(lambda zz: gen.decoder(zz, gd), z)
            dtg, _ = _timed(lambda: graphed(z), 30)
            c2[f"{prec}_T{T}"] = {"ms": 1e3 * dt, "graph_ms": 1e3 * dtg, "value": T * HOP / SAMPLE_RATE / min(dt, dtg)}
    out["config2_decoder_b1"] = c2

    # ---- config 4: voice conversion
    B, T = vc_batch, 862
    spec = synth.power_spectrogram(seed, "cfg4.spec", B, T).to(dev)
    lens4 = torch.full((B,), T, device=dev)
    src = synth.speaker_ids(seed, "cfg4.src", B).to(dev); tgt = synth.speaker_ids(seed, "cfg4.tgt", B).to(dev)
    eps = synth.latent(seed, "cfg4.eps", B, T).to(dev)
    with torch.no_grad():                                             # the oracle on the first item, alone
        ref4 = O.voice_conversion(spec[:1].cpu(), lens4[:1].cpu(), src[:1].cpu(), tgt[:1].cpu(), sd, eps[:1].cpu())
    c4 = {"batch": B, "frames": T, "gflop_per_utterance": 566.89}
    for prec in ("fp32", "fp32-split", "bf16"):
        engine.set_precision(prec)
        dt, wav = _timed(lambda: gen.voice_conversion(spec, lens4, src, tgt, noise=eps), 3)
        c4[prec] = {"ms": 1e3 * dt, "value": B * T * HOP / SAMPLE_RATE / dt, "tflops": B * 566.89e9 / dt / 1e12,
                    "parity_max_abs_item0": float((wav[:1].cpu() - ref4).abs().max()),
                    "tol": 1e-2 if prec == "bf16" else 1e-4}
    out["config4_voice_conversion"] = c4
    engine.set_precision(before)
    return out


if __name__ == "__main__":
    import json
    res = run(torch.device("cuda:0"))
    for k, v in res.items():
        print(k, json.dumps(v))
