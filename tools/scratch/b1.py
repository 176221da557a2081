import sys, time, torch
sys.path.insert(0, ".")
from vits_amd import engine, native, synth
from vits_amd.module.vits_generator import HotPathGenerator
from vits_amd.graphs import GraphedCall
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
for prec in sys.argv[1:] or ["fp32"]:
    engine.set_precision(prec)
    for T in (258, 862):
        z = synth.latent(999, "b1", 1, T).to(dev)
        mask = synth.length_mask(torch.tensor([T]), T).to(dev)
        g = gen.embed_speaker(torch.tensor([9], device=dev))
        dec = timed(lambda: gen.decoder(z, g))
        flow = timed(lambda: gen.flow(z, mask, speaker_id_embedded=g, reverse=True))
        gd = GraphedCall(gen.decoder, z, g)
        decg = timed(lambda: gd(z, g))
        gf = GraphedCall(lambda a, b, c: gen.flow(a, b, speaker_id_embedded=c, reverse=True), z, mask, g)
        flowg = timed(lambda: gf(z, mask, g))
        print(f"{prec} T={T}: decoder {dec:.3f} ms (graph {decg:.3f}), reverse flow {flow:.3f} ms (graph {flowg:.3f})")
