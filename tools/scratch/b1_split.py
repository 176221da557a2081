#!/usr/bin/env python3
"""B=1 decoder / flow latency per precision mode (eager)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vits_amd import engine, native, synth
if os.environ.get('VK_USE_DIAG'):
    native.use_diagnostic_library()
from vits_amd.module.vits_generator import HotPathGenerator
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
gen = HotPathGenerator(100, with_posterior=False)
gen.load_state_dict(synth.hot_path_state_dict(999, with_posterior=False), strict=True)
gen = gen.to(dev).eval()
g = gen.embed_speaker(torch.tensor([9], device=dev))
def timed(fn, n=30):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
for prec in ("fp32", "fp32-split"):
    engine.set_precision(prec)
    for T in (258, 862):
        z = synth.latent(999, "b1", 1, T).to(dev)
        mask = torch.ones(1, 1, T, device=dev)
        ref = None
        d = timed(lambda: gen.decoder(z, g))
        f = timed(lambda: gen.flow(z, mask, speaker_id_embedded=g, reverse=True))
        print(f"{prec:10s} T={T}: decoder {d:.3f} ms, reverse flow {f:.3f} ms")
a = None
engine.set_precision("fp32"); z = synth.latent(999, "b1", 1, 258).to(dev); a = gen.decoder(z, g)
engine.set_precision("fp32-split"); b = gen.decoder(z, g)
print("split vs fp32 max abs", (a - b).abs().max().item())
