import sys, time, torch
sys.path.insert(0, "/root/repo")
from vits_amd import synth
from vits_amd.module.vits_generator import VitsGenerator
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
gen = VitsGenerator(40, 100)
gen.load_state_dict(synth.full_state_dict(gen.state_dict(), 999), strict=True)
gen = gen.to(dev).eval()
ids = torch.tensor([synth.TTS_SENTENCE_IDS], device=dev); lens = torch.tensor([ids.shape[1]], device=dev); spk = torch.tensor([9], device=dev)
torch.manual_seed(1)
nw = torch.randn(1, 2, ids.shape[1], device=dev)
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out
t_enc, enc = timed(lambda: gen.text_encoder(ids, lens))
g = gen.embed_speaker(spk)
t_sdp, logw = timed(lambda: gen.stochastic_duration_predictor(enc[0], enc[3], speaker_id_embedded=g, reverse=True, noise_scale=0.8, noise=nw))
t_front, pr = timed(lambda: gen.frame_prior(ids, lens, spk, length_scale=3, noise_w=nw))
nz = torch.randn_like(pr[0])
t_tail, wav = timed(lambda: gen.synthesize_from_prior(pr[0], pr[1], pr[2], pr[3], noise=nz))
t_all, _ = timed(lambda: gen.text_to_speech(ids, lens, spk, length_scale=3, noise_w=nw, noise_z=nz))
print(f"text encoder {t_enc:.2f} ms, SDP reverse {t_sdp:.2f} ms, frame_prior total {t_front:.2f} ms, hot-path tail {t_tail:.2f} ms (T={pr[0].shape[2]}), text_to_speech {t_all:.2f} ms")
