import sys, torch
sys.path.insert(0, ".")
from vits_amd import engine, synth
from vits_amd.module.vits_generator import HotPathGenerator
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
sd = synth.hot_path_state_dict(999, with_posterior=False)
gen = HotPathGenerator(100, with_posterior=False)
gen.load_state_dict(sd, strict=True)
gen = gen.to(dev).eval()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 258
z = synth.latent(999, "b1", 1, T).to(dev)
g = gen.embed_speaker(torch.tensor([9], device=dev))
for _ in range(5):
    gen.decoder(z, g)
torch.cuda.synchronize()
