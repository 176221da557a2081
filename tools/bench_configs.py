#!/usr/bin/env python3
"""Times the BASELINE.json configurations other than the headline one (needs a GPU):
  1  text_to_speech, one ~4 s sentence: CPU oracle path vs the drop-in generator on the GPU
  2  decoder only, B=1, T=862 and T=258 (latency)
  4  voice conversion (posterior encoder -> flow fwd -> flow rev -> decoder), B=64, T=862, fp32 and
     with bf16 matrix-core inputs
Prints one line per case.  (Config 3 is bench.py's default, config 5 is bench.py --workload corpus.)"""
import sys
import time

import torch

sys.path.insert(0, ".")
from oracle import vits_oracle as O                                   # CPU baseline only
from vits_amd import native, synth
from vits_amd.module.vits_generator import VitsGenerator

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
gen = VitsGenerator(40, 100)
sd = synth.full_state_dict(gen.state_dict(), 999)
gen.load_state_dict(sd, strict=True)
cpu_gen = VitsGenerator(40, 100)
cpu_gen.load_state_dict(sd, strict=True)
gen = gen.to(dev).eval()


def timed(fn, n, sync=True):
    fn()
    if sync:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    if sync:
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


# ---- config 1
ids = torch.tensor([synth.TTS_SENTENCE_IDS]); lens = torch.tensor([ids.shape[1]]); spk = torch.tensor([9])
torch.manual_seed(1)
nw = torch.randn(1, 2, ids.shape[1])
m_p, logs_p, mask, g = cpu_gen.frame_prior(ids, lens, spk, length_scale=3, noise_w=nw)
nz = torch.randn_like(m_p)
dt_cpu, _ = timed(lambda: O.tts_tail(*cpu_gen.frame_prior(ids, lens, spk, length_scale=3, noise_w=nw)[:3], nz,
                                     g, sd), 3, sync=False)
audio = m_p.shape[2] * 256 / 22050
print(f"config 1  text_to_speech 43 tokens -> {audio:.2f} s: CPU oracle path {dt_cpu*1e3:8.1f} ms = {audio/dt_cpu:7.1f} x RT "
      f"({torch.get_num_threads()} threads)")
a = [t.to(dev) for t in (ids, lens, spk, nw, nz)]
dt, wav = timed(lambda: gen.text_to_speech(a[0], a[1], a[2], length_scale=3, noise_w=a[3], noise_z=a[4]), 10)
print(f"config 1  same on MI355X (stock-PyTorch front half + HIP hot path)  {dt*1e3:8.2f} ms = {audio/dt:7.1f} x RT")

# ---- config 2
gd = gen.embed_speaker(torch.tensor([9], device=dev))
for T in (862, 258):
    z = synth.latent(999, "cfg2.z", 1, T).to(dev)
    dt, _ = timed(lambda: gen.decoder(z, gd), 20)
    print(f"config 2  decoder only B=1 T={T}: {dt*1e3:7.2f} ms = {T*256/22050/dt:7.0f} x RT")
    from vits_amd.graphs import GraphedCall
    graphed = GraphedCall(gen.decoder, z, gd)
    dt, _ = timed(lambda: graphed(z, gd), 50)
    print(f"config 2  same, replayed as one HIP graph:  {dt*1e3:7.2f} ms = {T*256/22050/dt:7.0f} x RT")

# ---- config 4
B, T = 64, 862
spec = synth.power_spectrogram(999, "cfg4.spec", B, T).to(dev)
lens4 = torch.full((B,), T, device=dev)
src = synth.speaker_ids(999, "cfg4.src", B).to(dev); tgt = synth.speaker_ids(999, "cfg4.tgt", B).to(dev)
eps = synth.latent(999, "cfg4.eps", B, T).to(dev)
from vits_amd import engine
wav32 = None
for prec in ("fp32", "bf16"):
    engine.set_precision(prec)
    dt, wav = timed(lambda: gen.voice_conversion(spec, lens4, src, tgt, noise=eps), 3)
    extra = ""
    if prec == "fp32":
        wav32 = wav
    else:
        extra = f"  max-abs vs fp32 {(wav - wav32).abs().max().item():.2e} (|wav|max {wav32.abs().max().item():.2f})"
    print(f"config 4  voice conversion B={B} T={T} {prec}: {dt*1e3:8.1f} ms = {B*T*256/22050/dt:7.0f} x RT  "
          f"(566.9 GFLOP/utt -> {B*566.89e9/dt/1e12:.1f} TFLOP/s){extra}")
engine.set_precision("fp32")
