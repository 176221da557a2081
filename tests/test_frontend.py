"""Voice-conversion front end (reflect pad + STFT power spectrogram) against the oracle's
float64 DFT.  Parity with torchaudio, which the reference calls and which is not installed,
is unpinned; both sides restate its documented behaviour."""
import pytest
import torch

from oracle import vits_oracle as O
from vits_amd import frontend, synth


def _wav(n, batch=1):
    t = torch.arange(n, dtype=torch.float64) / 22050.0
    tone = 0.4 * torch.sin(2 * torch.pi * 220.0 * t) + 0.2 * torch.sin(2 * torch.pi * 3111.0 * t + 0.3)
    noise = torch.from_numpy(synth.normal(999, "frontend.noise", (batch, n))) * 0.05
    return (tone.unsqueeze(0) + noise).to(torch.float32)


def _check(spec, ref):
    assert spec.shape == ref.shape
    err = (spec.cpu() - ref).abs().max().item()
    assert err <= 2e-5 * ref.abs().max().item(), err


@pytest.mark.parametrize("n,batch", [(256 * 40, 1), (256 * 13 + 77, 2), (1024, 1)])
def test_power_spectrogram_cpu(n, batch):
    wav = _wav(n, batch)
    spec = frontend.power_spectrogram(wav)
    assert spec.shape == (batch, 513, (n + 768 - 1024) // 256 + 1)
    if n % 256 == 0:
        assert spec.shape[-1] == n // 256          # one frame per hop, as the generator's hop of 256 expects
    _check(spec, O.power_spectrogram(wav))
    assert (spec >= 0).all()


def test_int16_conversion():
    w = torch.tensor([-2.0, -1.0, -0.5, 0.0, 0.25, 1.0, 3.0])
    assert frontend.to_int16(w).tolist() == [-32767, -32767, -16384, 0, 8192, 32767, 32767]


def test_wav_writer_round_trip(tmp_path):
    """16-bit PCM file as the reference's scripts leave it (22050 Hz, one channel): header fields
    and samples survive a round trip, within half a quantisation step."""
    import wave
    wav = _wav(22050)[:, :22050] * 1.5                      # some samples clip
    frontend.save_wav(tmp_path / "o.wav", wav, 22050)
    with wave.open(str(tmp_path / "o.wav"), "rb") as f:
        assert (f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()) == (1, 2, 22050, 22050)
    back, rate = frontend.load_wav(tmp_path / "o.wav")
    assert rate == 22050 and back.shape == wav.shape
    assert (back - wav.clamp(-1, 1)).abs().max().item() <= 0.5 / 32767 + 1e-7
    stereo = torch.stack([wav[0, :100], -wav[0, :100]])
    frontend.save_wav(tmp_path / "s.wav", stereo)
    back2, _ = frontend.load_wav(tmp_path / "s.wav")
    assert back2.shape == (2, 100) and torch.allclose(back2[0], -back2[1])
    with pytest.raises(ValueError):
        frontend.save_wav(tmp_path / "bad.wav", torch.zeros(1, 1, 8))


@pytest.mark.gpu
def test_power_spectrogram_gpu_feeds_voice_conversion():
    from vits_amd.module.vits_generator import HotPathGenerator
    wav = _wav(256 * 48, 2)
    spec = frontend.power_spectrogram(wav.to("cuda:0"))
    ref = O.power_spectrogram(wav)
    _check(spec, ref)
    sd = synth.hot_path_state_dict(999, with_posterior=True)
    gen = HotPathGenerator(100)
    gen.load_state_dict(sd, strict=True)
    gen = gen.to("cuda:0").eval()
    lens = torch.tensor([48, 48], device="cuda:0")
    src, tgt = torch.tensor([3, 4], device="cuda:0"), torch.tensor([9, 57], device="cuda:0")
    eps = synth.latent(999, "frontend.eps", 2, 48)
    out = gen.voice_conversion(spec, lens, src, tgt, noise=eps.to("cuda:0"))
    want = O.voice_conversion(ref, lens.cpu(), src.cpu(), tgt.cpu(), sd, eps)
    assert out.shape == want.shape == (2, 1, 256 * 48)
    assert (out.cpu() - want).abs().max().item() <= 1e-4
