import sys, torch
sys.path.insert(0, ".")
from vits_amd import synth
from vits_amd.module.vits_generator import VitsGenerator
DEV = "cuda:0"
torch.set_grad_enabled(False)
gen = VitsGenerator(40, 100)
gen.load_state_dict(synth.full_state_dict(gen.state_dict(), 999), strict=True)
gen = gen.to(DEV).eval()
worst = {}
for n in (43, 17, 80):
    lens = torch.tensor([n], device=DEV)
    for seed in range(6):
        g = torch.Generator().manual_seed(seed)
        ids = torch.randint(0, 40, (1, n), generator=g).to(DEV)
        spk = torch.tensor([seed * 7 % 100], device=DEV)
        nw = torch.randn(1, 2, n, generator=g).to(DEV)
        ref = [t.clone() for t in gen._token_stage(ids, lens, spk, nw, 1.0, 0.8)]
        got = gen._token_stage_graphed(ids, lens, spk, nw, 1.0, 0.8)
        for name, a, b in zip(("m_p", "logs_p", "text_mask", "w_ceil", "g"), got, ref):
            d = (a - b).abs().max().item()
            worst[name] = max(worst.get(name, 0.0), d)
print(worst)
