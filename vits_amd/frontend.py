"""Voice-conversion front end: waveform -> power spectrogram [.., 513, frames], on whatever
device the waveform lives on (rocFFT through ``torch.stft`` on the MI355X).

Follows /root/reference/vits_voice_converter.py:82-96 and module/dataset_util.py:63-77:
reflect padding by (filter_length - hop_length) / 2 on both sides, then
``torchaudio.functional.spectrogram(pad=0, window=hann(win_length), n_fft=filter_length,
hop_length, win_length, power=2, normalized=False, center=False)``, i.e. a one-sided STFT of
un-centred frames and the squared magnitude.  torchaudio is not installed here, so this is a
restatement of its documented behaviour, checked against a float64 DFT
(oracle/vits_oracle.py::power_spectrogram); parity with torchaudio itself is unpinned.
"""
from __future__ import annotations

import torch


def power_spectrogram(wav: torch.Tensor, filter_length: int = 1024, hop_length: int = 256,
                      win_length: int = 1024) -> torch.Tensor:
    """wav [..., N] float -> [..., filter_length // 2 + 1, frames], frames = N // hop_length
    when N is a multiple of the hop (the padding makes up for the un-centred frames)."""
    pad = (filter_length - hop_length) // 2
    lead = wav.shape[:-1]
    x = wav.reshape(-1, 1, wav.shape[-1]).to(torch.float32)
    x = torch.nn.functional.pad(x, (pad, pad), mode="reflect").squeeze(1)
    window = torch.hann_window(win_length, dtype=torch.float32, device=wav.device)
    spec = torch.stft(x, n_fft=filter_length, hop_length=hop_length, win_length=win_length, window=window,
                      center=False, normalized=False, onesided=True, return_complex=True)
    power = spec.real.square() + spec.imag.square()
    return power.reshape(*lead, power.shape[-2], power.shape[-1])


def to_int16(wav: torch.Tensor) -> torch.Tensor:
    """Waveform in [-1, 1] -> 16-bit PCM samples (what a wav writer stores)."""
    return (wav.clamp(-1.0, 1.0) * 32767.0).round().to(torch.int16)


def save_wav(path, wav: torch.Tensor, sample_rate: int = 22050) -> None:
    """Write a [channels, N] (or [N]) float waveform as 16-bit PCM, the last step of
    /root/reference/vits_voice_converter.py:103-104 and vits_text_to_speech.py:107
    (``torchaudio.save(path, output_wav, sample_rate)``).  Standard-library ``wave`` writer;
    samples are interleaved by channel as the RIFF format wants them."""
    import wave
    if wav.dim() == 1:
        wav = wav.unsqueeze(0)
    if wav.dim() != 2:
        raise ValueError(f"save_wav expects [channels, samples], got {tuple(wav.shape)}")
    pcm = to_int16(wav.detach().to("cpu", torch.float32)).t().contiguous()      # [N, channels]
    with wave.open(str(path), "wb") as f:
        f.setnchannels(pcm.shape[1])
        f.setsampwidth(2)
        f.setframerate(int(sample_rate))
        f.writeframes(pcm.numpy().astype("<i2").tobytes())


def load_wav(path):
    """Read 16-bit PCM written by ``save_wav`` back as ([channels, N] float in [-1, 1], rate)."""
    import wave

    import numpy as np
    with wave.open(str(path), "rb") as f:
        if f.getsampwidth() != 2:
            raise ValueError("only 16-bit PCM is supported")
        ch, rate, n = f.getnchannels(), f.getframerate(), f.getnframes()
        pcm = np.frombuffer(f.readframes(n), dtype="<i2").reshape(n, ch)
    return torch.from_numpy(pcm.astype(np.float32) / 32767.0).t().contiguous(), rate
