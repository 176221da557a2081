"""Capture golden vectors from the reference implementation.

Runs ONLY in the development container, where the reference lives at
/root/reference (it cannot travel to the GPU box).  It imports the reference's
own model-component modules, loads the deterministic synthetic weights of
``vits_amd.synth`` into them with ``load_state_dict(strict=True)``, runs them on
deterministic inputs and stores inputs-that-cannot-be-regenerated plus expected
outputs as small ``.npz`` files next to this script.  While doing so it checks
the CPU restatement (oracle/vits_oracle.py) against the reference, module by
module, and aborts if any differs by more than 2e-6.

The reference modules import ``torchvision`` without using it
(e.g. module/model_component/decoder.py:10-11); it is not installed, so inert
stand-in modules are registered before the import.  No reference file is
copied or modified.

Usage:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

for _n in ("torchvision", "torchvision.models", "torchvision.transforms", "torchvision.utils"):
    sys.modules.setdefault(_n, types.ModuleType(_n))
# module/vits_generator.py:17 imports the compiled monotonic-alignment extension (training only,
# not built in the read-only reference tree): an inert stand-in satisfies the import.
_mas = types.ModuleType("module.model_component.monotonic_align")
_mas.maximum_path = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("training-only"))
sys.modules["module.model_component.monotonic_align"] = _mas
sys.path.insert(0, "/root/reference")

from module.model_component.decoder import Decoder, ResNetBlock          # noqa: E402
from module.model_component.flow import Flow, ResidualCouplingLayer      # noqa: E402
from module.model_component.wn import WN                                  # noqa: E402
from module.model_component.posterior_encoder import PosteriorEncoder    # noqa: E402

from module.vits_generator import VitsGenerator as RefGenerator           # noqa: E402

from vits_amd import synth                                                # noqa: E402
from vits_amd.module.vits_generator import VitsGenerator as OwnGenerator  # noqa: E402
from oracle import vits_oracle as O                                       # noqa: E402

SEED = 999
TOL = 2e-6


def check(name, ref, mine, tol=TOL):
    err = (ref - mine).abs().max().item()
    scale = ref.abs().max().item()
    print(f"  {name:34s} max|ref|={scale:9.4f}  max-abs-diff={err:.3e}")
    if not err <= tol * max(1.0, scale):
        raise SystemExit(f"oracle differs from the reference on {name}: {err}")
    return err


def main():
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    sd = synth.hot_path_state_dict(SEED)
    out = {}

    # ---- G1/G2: decoder with per-stage taps ------------------------------------------------
    dec = Decoder(speaker_id_embedding_dim=256)
    dec.load_state_dict(synth.strip_prefix(sd, "decoder"), strict=True)
    dec.eval()
    z = synth.latent(SEED, "g1.z", 2, 24)
    spk = torch.tensor([9, 57])
    g = O.speaker_embed(spk, sd)
    taps_ref = {}
    hooks = []
    for i in range(4):
        hooks.append(dec.ups[i].register_forward_hook(
            lambda m, a, o, i=i: taps_ref.__setitem__(f"up{i}", o.detach().clone())))
    wav_ref = dec(z, g)
    for h in hooks:
        h.remove()
    taps = {}
    wav = O.decoder(z, g, sd, taps=taps)
    check("decoder wav", wav_ref, wav)
    for i in range(4):
        check(f"decoder up{i}", taps_ref[f"up{i}"], taps[f"up{i}"])
    np.savez_compressed(os.path.join(HERE, "g1_decoder.npz"),
                        speakers=spk.numpy(), wav=wav_ref.numpy(),
                        **{f"up{i}_head": taps_ref[f"up{i}"][0, :, :256].numpy() for i in range(4)},
                        **{f"mrf{i}_head": taps[f"mrf{i}"][0, :, :256].numpy() for i in range(4)})
    print("  decoder stage stats (max|x|):",
          {k: round(v.abs().max().item(), 3) for k, v in taps.items()}, "wav", wav.abs().max().item())

    # ---- G3: standalone resblocks, three kernel sizes ---------------------------------------
    g3 = {}
    for j, k in enumerate((3, 7, 11)):
        rb = ResNetBlock(32, k, (1, 3, 5))
        pfx = f"decoder.resblocks.{9 + j}"
        rb.load_state_dict(synth.strip_prefix(sd, pfx), strict=True)
        x = synth.latent(SEED, f"g3.x{k}", 1, 512, channels=32)
        y_ref = rb(x)
        check(f"resblock k={k}", y_ref, O.resblock(x, sd, pfx, k))
        g3[f"y{k}"] = y_ref.numpy()
    np.savez_compressed(os.path.join(HERE, "g3_resblocks.npz"), **g3)

    # ---- G4: WN with 4 and 16 layers, ragged mask --------------------------------------------
    mask = synth.length_mask([48, 31], 48)
    g4 = {}
    for n, pfx in ((4, "flow.flows.0.wn"), (16, "posterior_encoder.wn")):
        m = WN(192, 5, 1, n, speaker_id_embedding_dim=256)
        m.load_state_dict(synth.strip_prefix(sd, pfx), strict=True)
        x = synth.latent(SEED, f"g4.x{n}", 2, 48) * mask
        y_ref = m(x, mask, g)
        check(f"WN n={n}", y_ref, O.wn(x, mask, g, sd, pfx, n))
        g4[f"y{n}"] = y_ref.numpy()
    np.savez_compressed(os.path.join(HERE, "g4_wn.npz"), **g4)

    # ---- G5: one coupling layer, both directions --------------------------------------------
    rcl = ResidualCouplingLayer(192, 192, 256, 5, 1, 4)
    rcl.load_state_dict(synth.strip_prefix(sd, "flow.flows.2"), strict=True)
    x = synth.latent(SEED, "g5.x", 2, 48)
    g5 = {}
    for rev in (False, True):
        y_ref = rcl(x, mask, g, rev)
        check(f"coupling reverse={rev}", y_ref, O.coupling(x, mask, g, sd, "flow.flows.2", rev))
        g5["rev" if rev else "fwd"] = y_ref.numpy()
    np.savez_compressed(os.path.join(HERE, "g5_coupling.npz"), **g5)

    # ---- G6: whole flow, both directions, and round trip ------------------------------------
    fl = Flow()
    fl.load_state_dict(synth.strip_prefix(sd, "flow"), strict=True)
    x = synth.latent(SEED, "g6.x", 2, 48)
    y_fwd = fl(x, mask, g)
    y_rev = fl(x, mask, g, reverse=True)
    check("flow forward", y_fwd, O.flow(x, mask, g, sd, reverse=False))
    check("flow reverse", y_rev, O.flow(x, mask, g, sd, reverse=True))
    rt = fl(fl(x * mask, mask, g), mask, g, reverse=True)
    print(f"  reference flow round trip error {((rt - x * mask).abs().max().item()):.3e}")
    np.savez_compressed(os.path.join(HERE, "g6_flow.npz"), fwd=y_fwd.numpy(), rev=y_rev.numpy())

    # ---- G7: posterior encoder with the noise draw pinned -----------------------------------
    pe = PosteriorEncoder(speaker_id_embedding_dim=256)
    pe.load_state_dict(synth.strip_prefix(sd, "posterior_encoder"), strict=True)
    spec = synth.power_spectrogram(SEED, "g7.spec", 2, 40)
    lens = torch.tensor([40, 27])
    torch.manual_seed(1234)
    eps = torch.randn(2, 192, 40)
    torch.manual_seed(1234)
    z_ref, m_ref, logs_ref, mask_ref = pe(spec, lens, g)
    z_o, m_o, logs_o, mask_o = O.posterior_encoder(spec, lens, g, sd, eps)
    check("posterior m", m_ref, m_o)
    check("posterior logs", logs_ref, logs_o)
    check("posterior z", z_ref, z_o)
    assert torch.equal(mask_ref, mask_o)
    np.savez_compressed(os.path.join(HERE, "g7_posterior.npz"), eps=eps.numpy(), lengths=lens.numpy(),
                        z=z_ref.numpy(), m=m_ref.numpy(), logs=logs_ref.numpy())

    # ---- G9: voice-conversion chain (vits_generator.py:191-196) from reference components ----
    src, tgt = torch.tensor([3, 71]), torch.tensor([42, 8])
    g_src, g_tgt = O.speaker_embed(src, sd), O.speaker_embed(tgt, sd)
    torch.manual_seed(1234)
    zq, _, _, mk = pe(spec, lens, g_src)
    z_p = fl(zq, mk, g_src)
    z_hat = fl(z_p, mk, g_tgt, reverse=True)
    wav_vc = dec(z_hat * mk, g_tgt)
    check("voice conversion wav", wav_vc, O.voice_conversion(spec, lens, src, tgt, sd, eps))
    np.savez_compressed(os.path.join(HERE, "g9_vc.npz"), src=src.numpy(), tgt=tgt.numpy(),
                        wav=wav_vc.numpy())

    # ---- G8': the tail of text_to_speech (vits_generator.py:184-186) from reference components ----
    T = 50
    mask8 = synth.length_mask([50, 37], T)
    m_p = synth.latent(SEED, "g8.m_p", 2, T) * 0.5
    logs_p = synth.latent(SEED, "g8.logs_p", 2, T) * 0.2 - 0.5
    eps8 = synth.latent(SEED, "g8.eps", 2, T)
    z_p8 = m_p + eps8 * torch.exp(logs_p) * 0.667
    z8 = fl(z_p8, mask8, g, reverse=True)
    wav8 = dec((z8 * mask8)[:, :, :None], g)
    z8_o, wav8_o = O.tts_tail(m_p, logs_p, mask8, eps8, g, sd)
    check("tts tail z", z8, z8_o)
    check("tts tail wav", wav8, wav8_o)
    np.savez_compressed(os.path.join(HERE, "g8_tts_tail.npz"), z=z8.numpy(), wav=wav8.numpy())

    # ---- G10: checksums of a full-size decoder run (T=862) -----------------------------------
    z10 = synth.latent(SEED, "cfg2.z", 1, 862)
    g10 = O.speaker_embed(torch.tensor([9]), sd)
    w10 = dec(z10, g10)
    check("decoder T=862", w10, O.decoder(z10, g10, sd))
    idx = np.linspace(0, w10.numel() - 1, 64).astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "g10_decoder_full_checksums.npz"),
                        sum=np.float64(w10.double().sum().item()),
                        sumsq=np.float64((w10.double() ** 2).sum().item()),
                        idx=idx, samples=w10.flatten()[idx].numpy(),
                        absmax=np.float64(w10.abs().max().item()))
    # ---- G11: complete generator: state_dict layout, front half, text_to_speech ---------------
    import json
    ref = RefGenerator(n_phoneme=40, n_speakers=100)
    own = OwnGenerator(40, 100)
    layout = [(k, list(v.shape)) for k, v in ref.state_dict().items()]
    assert layout == [(k, list(v.shape)) for k, v in own.state_dict().items()], "state_dict layout differs"
    with open(os.path.join(HERE, "netg_state_dict_layout.json"), "w") as f:
        json.dump(layout, f)
    full = synth.full_state_dict(own.state_dict(), SEED)
    ref.load_state_dict(full, strict=True)
    own.load_state_dict(full, strict=True)
    ref.eval(); own.eval()
    ids = torch.tensor([synth.TTS_SENTENCE_IDS])
    lens = torch.tensor([ids.shape[1]])
    spk1 = torch.tensor([9])
    enc_ref = ref.text_encoder(ids, lens)
    enc_own = own.text_encoder(ids, lens)
    for name, a, b in zip(("text_encoded", "m", "logs", "mask"), enc_ref, enc_own):
        check(f"text encoder {name}", a, b, 1e-5)
    g1 = ref.speaker_embedding(spk1).unsqueeze(-1)
    torch.manual_seed(4321)
    logw_ref = ref.stochastic_duration_predictor(enc_ref[0], enc_ref[3], speaker_id_embedded=g1, reverse=True,
                                                 noise_scale=0.8)
    torch.manual_seed(4321)
    noise_w = torch.randn(1, 2, ids.shape[1])
    logw_own = own.stochastic_duration_predictor(enc_own[0], enc_own[3], speaker_id_embedded=g1, reverse=True,
                                                 noise_scale=0.8, noise=noise_w)
    check("duration predictor logw", logw_ref, logw_own, 1e-5)
    LS = 3
    torch.manual_seed(4321)
    wav_ref = ref.text_to_speech(ids, lens, spk1, length_scale=LS)
    T11 = wav_ref.shape[2] // 256
    torch.manual_seed(4321)
    noise_w = torch.randn(1, 2, ids.shape[1])
    noise_z = torch.randn_like(torch.empty(1, T11, 192).transpose(1, 2))    # same strides as the expanded prior
    m_p, logs_p, mask11, g11 = own.frame_prior(ids, lens, spk1, length_scale=LS, noise_w=noise_w)
    assert m_p.shape[2] == T11, (m_p.shape, T11)
    z11, wav11 = O.tts_tail(m_p, logs_p, mask11, noise_z, g11, full)
    check("text_to_speech wav (own front half + oracle tail)", wav_ref, wav11, 1e-5)
    np.savez_compressed(os.path.join(HERE, "g11_tts.npz"), ids=ids.numpy(), speaker=spk1.numpy(),
                        length_scale=np.int64(LS), noise_w=noise_w.numpy(), noise_z=noise_z.contiguous().numpy(),
                        text_encoded=enc_ref[0].numpy(), m=enc_ref[1].numpy(), logs=enc_ref[2].numpy(),
                        logw=logw_ref.numpy(), wav=wav_ref.numpy())
    print(f"  text_to_speech: {ids.shape[1]} tokens -> {T11} frames = {T11 * 256 / 22050:.2f} s")
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
