#!/usr/bin/env python3
"""Where one text_to_speech call (config 1) spends its time: token-level graph, path expansion, flow + decoder."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vits_amd import synth
from vits_amd.graphs import GraphedCall
from vits_amd.module.vits_generator import VitsGenerator

def timed(fn, n=30):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, out

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
gen = VitsGenerator(40, 100)
gen.load_state_dict(synth.full_state_dict(gen.state_dict(), 999), strict=True)
gen = gen.to(dev).eval()
ids = torch.tensor([synth.TTS_SENTENCE_IDS]); lens = torch.tensor([ids.shape[1]]); spk = torch.tensor([9])
torch.manual_seed(1)
nw = torch.randn(1, 2, ids.shape[1])
a = [t.to(dev) for t in (ids, lens, spk, nw)]
m_p, logs_p, mask, g = gen.frame_prior(a[0], a[1], a[2], length_scale=3, noise_w=a[3])
nz = torch.randn_like(m_p)
print("tokens", ids.shape[1], "frames", m_p.shape[2])
print("text_to_speech          %.3f ms" % timed(lambda: gen.text_to_speech(a[0], a[1], a[2], length_scale=3, noise_w=a[3], noise_z=nz))[0])
print("frame_prior             %.3f ms" % timed(lambda: gen.frame_prior(a[0], a[1], a[2], length_scale=3, noise_w=a[3]))[0])
print("token stage (graph)     %.3f ms" % timed(lambda: gen._token_stage_graphed(a[0], a[1], a[2], a[3], 3, 0.8))[0])
print("token stage (eager)     %.3f ms" % timed(lambda: gen._token_stage(a[0], a[1], a[2], a[3], 3, 0.8))[0])
print("synthesize_from_prior   %.3f ms" % timed(lambda: gen.synthesize_from_prior(m_p, logs_p, mask, g, noise=nz))[0])
te = GraphedCall(gen.text_encoder, a[0], a[1])
print("text encoder (graph)    %.3f ms" % timed(lambda: te(a[0], a[1]))[0])
enc, _, _, tmask = gen.text_encoder(a[0], a[1])
gg = gen.embed_speaker(a[2])
sdp = GraphedCall(lambda e, m, s, n: gen.stochastic_duration_predictor(e, m, speaker_id_embedded=s, reverse=True, noise_scale=0.8, noise=n), enc, tmask, gg, a[3])
print("duration pred. (graph)  %.3f ms" % timed(lambda: sdp(enc, tmask, gg, a[3]))[0])
z_p = m_p + nz * torch.exp(logs_p) * 0.667
print("flow reverse            %.3f ms" % timed(lambda: gen.flow(z_p, mask, speaker_id_embedded=g, reverse=True))[0])
z = gen.flow(z_p, mask, speaker_id_embedded=g, reverse=True)
print("decoder                 %.3f ms" % timed(lambda: gen.decoder(z * mask, speaker_id_embedded=g))[0])
fd = GraphedCall(lambda zp, mk, s: gen.flow_reverse_decode(zp, mk, s), z_p, mask, g)
print("flow + decoder (graph)  %.3f ms" % timed(lambda: fd(z_p, mask, g))[0])
