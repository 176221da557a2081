import sys, time, torch
sys.path.insert(0, ".")
from vits_amd import engine, native, synth
from vits_amd.module.vits_generator import HotPathGenerator
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
sd = synth.hot_path_state_dict(999, with_posterior=False)
gen = HotPathGenerator(100, with_posterior=False)
gen.load_state_dict(sd, strict=True)
gen = gen.to(dev).eval()
def timed(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for B, T in ((1, 258), (1, 500), (1, 862), (2, 862), (4, 862), (8, 862), (16, 862)):
    z = synth.latent(999, "b1", B, T).to(dev)
    mask = synth.length_mask(torch.full((B,), T), T).to(dev)
    g = gen.embed_speaker(torch.arange(B, device=dev))
    print(f"B={B} T={T}: reverse flow {timed(lambda: gen.flow(z, mask, speaker_id_embedded=g, reverse=True)):.3f} ms")
