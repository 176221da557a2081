#!/usr/bin/env python3
"""Kernel trace input: the decoder alone for one utterance (B=1), to see where a single-utterance
latency goes.  Run under `rocprofv3 --kernel-trace --output-format csv -- python3 tools/b1_trace.py T [precision]`;
the traced pass sits between two `erfinv` marker launches (tools/step_trace.py cut ... extracts it)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vits_amd import engine, synth
from vits_amd.module.vits_generator import HotPathGenerator

T = int(sys.argv[1]) if len(sys.argv) > 1 else 258
engine.set_precision(sys.argv[2] if len(sys.argv) > 2 else "fp32")
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
gen = HotPathGenerator(100, with_posterior=False)
gen.load_state_dict(synth.hot_path_state_dict(999, with_posterior=False), strict=True)
gen = gen.to(dev).eval()
z = synth.latent(999, "b1", 1, T).to(dev)
ids = torch.tensor([9], device=dev)
g = gen.speaker_ref(ids)
marker = torch.full((1,), 0.5, device=dev)
for _ in range(5):
    gen.decoder(z, g)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    gen.decoder(z, g)
e1.record()
torch.cuda.synchronize()
print(f"decoder B=1 T={T}: {e0.elapsed_time(e1) / 20:.3f} ms per call")
torch.erfinv(marker)
gen.decoder(z, g)
torch.erfinv(marker)
torch.cuda.synchronize()
