"""Generator facade: speaker table + posterior encoder + flow + decoder, inference methods.

Mirrors the inference half of /root/reference/module/vits_generator.py: ``VitsGenerator``
(:67-125), ``text_to_speech`` (:168-187), ``voice_conversion`` (:189-197), ``sequence_mask``
(:39-43) and ``generate_path`` (:45-64).  The hot path proper - reverse flow then decoder
(:185-186), and for voice conversion posterior encoder -> forward flow -> reverse flow ->
decoder (:193-196) - runs on the HIP kernels.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from .. import engine, native
from ..engine import SpeakerRef
from .model_component.decoder import Decoder
from .model_component.flow import Flow
from .model_component.posterior_encoder import PosteriorEncoder
from .model_component.stochastic_duration_predictor import StochasticDurationPredictor
from .model_component.text_encoder import TextEncoder


def sequence_mask(length, max_length=None):
    if max_length is None:
        max_length = length.max()
    steps = torch.arange(max_length, dtype=length.dtype, device=length.device)
    return steps.unsqueeze(0) < length.unsqueeze(1)


def generate_path(duration, mask):
    """duration [B,1,Tx] (integer-valued), mask [B,1,Ty,Tx] -> hard monotonic alignment
    [B,1,Ty,Tx]: frame y belongs to token x iff cum[x-1] <= y < cum[x]."""
    b, _, t_y, t_x = mask.shape
    cum = torch.cumsum(duration, -1).view(b * t_x)
    upto = sequence_mask(cum, t_y).to(mask.dtype).view(b, t_x, t_y)
    prev = torch.cat([torch.zeros_like(upto[:, :1]), upto[:, :-1]], dim=1)
    return (upto - prev).unsqueeze(1).transpose(2, 3) * mask


class HotPathGenerator(nn.Module):
    """The sub-modules the hot path needs, with the reference's attribute names and
    ``state_dict`` prefixes (``speaker_embedding``, ``posterior_encoder``, ``decoder``, ``flow``)."""

    def __init__(self, n_speakers: int, with_posterior: bool = True):
        super().__init__()
        self.n_speakers = n_speakers
        self.z_channels = 192
        self.spec_channels = 513
        self.phoneme_embedding_dim = 192
        self.speaker_id_embedding_dim = 256
        self.speaker_embedding = nn.Embedding(n_speakers, self.speaker_id_embedding_dim)
        if with_posterior:
            self.posterior_encoder = PosteriorEncoder(
                speaker_id_embedding_dim=self.speaker_id_embedding_dim, in_spec_channels=self.spec_channels,
                out_z_channels=self.z_channels, phoneme_embedding_dim=self.phoneme_embedding_dim)
        self.decoder = Decoder(speaker_id_embedding_dim=self.speaker_id_embedding_dim,
                               in_z_channel=self.z_channels)
        self.flow = Flow(speaker_id_embedding_dim=self.speaker_id_embedding_dim,
                         in_z_channels=self.z_channels, phoneme_embedding_dim=self.phoneme_embedding_dim)

    @torch.no_grad()
    def embed_speaker(self, speaker_id):
        """The embedded speakers [B, G, 1] as the reference forms them (vits_generator.py:170,191-192)."""
        return self.speaker_embedding(speaker_id).unsqueeze(-1)

    def speaker_ref(self, speaker_id) -> SpeakerRef:
        """Speaker ids bound to this generator's embedding table.  Accepted by every module here in
        place of the embedded speakers: the speaker conditioning is then a row of a per-speaker
        table (built once per weights), not an embedding gather and a GEMV per call."""
        return SpeakerRef(speaker_id, self.speaker_embedding)

    def set_precision(self, name):
        """Matrix-core input precision of this generator ("fp32", "bf16", "fp32-split"; None: the
        process default of ``engine.set_precision``), whatever other models in the process use."""
        engine.set_module_precision(self, name)
        return self

    @torch.no_grad()
    def synthesize_from_prior(self, m_p, logs_p, spec_mask, speaker_id_embedded, noise=None,
                              noise_scale=.667, max_len=None):
        """Lines 184-186 of the reference's ``text_to_speech``: sample z_p from the frame-rate
        prior, invert the flow, decode.  ``noise`` pins the ``randn_like`` draw."""
        if noise is None:
            noise = torch.randn_like(m_p)
        z_p = m_p + noise * torch.exp(logs_p) * noise_scale
        z = self.flow(z_p, spec_mask, speaker_id_embedded=speaker_id_embedded, reverse=True)
        # (z * spec_mask)[:, :, :max_len]: the slice is a view, the product is taken by the decoder's first conv
        return self.decoder(z[:, :, :max_len], speaker_id_embedded=speaker_id_embedded, z_mask=spec_mask)

    @torch.no_grad()
    def flow_reverse_decode(self, z_p, spec_mask, speaker_id_embedded):
        """The hot path (vits_generator.py:185-186); ``speaker_id_embedded``: [B, G, 1] or a SpeakerRef."""
        z = self.flow(z_p, spec_mask, speaker_id_embedded=speaker_id_embedded, reverse=True)
        return self.decoder(z, speaker_id_embedded=speaker_id_embedded, z_mask=spec_mask)

    @torch.no_grad()
    def voice_conversion(self, spec_padded, spec_lengths, source_speaker_id, target_speaker_id, noise=None):
        assert self.n_speakers > 0
        emb_source = self.speaker_ref(source_speaker_id)
        emb_target = self.speaker_ref(target_speaker_id)
        z, _, _, spec_mask = self.posterior_encoder(spec_padded, spec_lengths,
                                                    speaker_id_embedded=emb_source, noise=noise)
        z_p = self.flow(z, spec_mask, speaker_id_embedded=emb_source)
        z_hat = self.flow(z_p, spec_mask, speaker_id_embedded=emb_target, reverse=True)
        return self.decoder(z_hat, speaker_id_embedded=emb_target, z_mask=spec_mask)


class VitsGenerator(HotPathGenerator):
    """Drop-in for the reference's ``VitsGenerator(n_phoneme, n_speakers)``
    (/root/reference/module/vits_generator.py:67-197), inference methods only: the complete
    ``netG_cpu.pth`` loads with ``strict=True``; ``text_to_speech`` runs the text encoder,
    duration predictor and path expansion (:169-182) on the token-level kernels and the hot path
    (:184-186) on the convolution kernels; ``voice_conversion`` is inherited.  The training
    ``forward`` (:127-166) is out of scope."""

    def __init__(self, n_phoneme, n_speakers):
        nn.Module.__init__(self)
        self.n_phoneme = n_phoneme
        self.n_speakers = n_speakers
        self.phoneme_embedding_dim = 192
        self.spec_channels = 513
        self.z_channels = 192
        self.segment_size = 32
        self.speaker_id_embedding_dim = 256
        # registration order follows the reference so that state_dict() lists keys identically
        self.text_encoder = TextEncoder(n_phoneme=n_phoneme, phoneme_embedding_dim=self.phoneme_embedding_dim,
                                        out_channels=self.z_channels, p_dropout=0.1)
        self.speaker_embedding = nn.Embedding(n_speakers, self.speaker_id_embedding_dim)
        self.posterior_encoder = PosteriorEncoder(
            speaker_id_embedding_dim=self.speaker_id_embedding_dim, in_spec_channels=self.spec_channels,
            out_z_channels=self.z_channels, phoneme_embedding_dim=self.phoneme_embedding_dim)
        self.decoder = Decoder(speaker_id_embedding_dim=self.speaker_id_embedding_dim, in_z_channel=self.z_channels)
        self.flow = Flow(speaker_id_embedding_dim=self.speaker_id_embedding_dim, in_z_channels=self.z_channels,
                         phoneme_embedding_dim=self.phoneme_embedding_dim)
        self.stochastic_duration_predictor = StochasticDurationPredictor(
            speaker_id_embedding_dim=self.speaker_id_embedding_dim, phoneme_embedding_dim=self.phoneme_embedding_dim,
            filter_channels=192, kernel_size=3, p_dropout=0.5, n_flows=4)

    def forward(self, *args, **kwargs):
        raise NotImplementedError("training forward (vits_generator.py:127-166) is out of scope of the inference build")

    # The token-level half (text encoder, duration predictor) is some 60 small launches, each bound
    # by latency.  It is captured once per (batch, tokens, scales) into a HIP graph and replayed
    # (bit-identical to the eager run: same kernels, same order); everything after the durations
    # has data-dependent shapes and stays eager.  VITS_AMD_FRONT_GRAPH=0, or
    # ``use_front_graph = False`` on an instance, keeps this half eager too.
    use_front_graph = os.environ.get("VITS_AMD_FRONT_GRAPH", "1") != "0"
    use_expand_kernel = os.environ.get("VITS_AMD_TOKEN_KERNELS", "1") != "0"   # 0: alignment matrix + matmul in PyTorch
    max_front_graphs = 16           # captured shapes kept per instance (least recently used goes first)

    def _token_stage(self, text_padded, text_lengths, speaker_id, noise_w, length_scale, noise_scale_w):
        """Lines 169-176 of ``text_to_speech``: token-level prior and sampled durations."""
        text_encoded, m_p, logs_p, text_mask = self.text_encoder(text_padded, text_lengths)
        g = self.speaker_ref(speaker_id) if text_padded.is_cuda else self.embed_speaker(speaker_id)
        logw = self.stochastic_duration_predictor(text_encoded, text_mask, speaker_id_embedded=g, reverse=True,
                                                  noise_scale=noise_scale_w, noise=noise_w)
        w_ceil = torch.ceil(torch.exp(logw) * text_mask * length_scale)
        return m_p, logs_p, text_mask, w_ceil

    def _front_weights_key(self):
        """What the captured kernels read: packed or transposed copies of the parameters, made by
        the plan caches for the storage and version the parameters had at capture.  A graph is only
        good while both stay (an in-place ``load_state_dict`` keeps the storage and bumps the
        version; ``.to()`` moves the storage), and while the precision the plans were built for does."""
        if getattr(self, "_front_params_seen", -1) != engine._registrations[0]:
            mods = (self.text_encoder, self.stochastic_duration_predictor, self.speaker_embedding)
            self._front_params = [p for m in mods for p in m.parameters()]     # the tree walk is the costly part
            self._front_params_seen = engine._registrations[0]
        return hash((engine.module_precision(self),) + tuple((p.data_ptr(), p._version) for p in self._front_params))

    def _token_stage_graphed(self, text_padded, text_lengths, speaker_id, noise_w, length_scale, noise_scale_w):
        from ..graphs import GraphedCall
        if not hasattr(self, "_front_graphs"):
            self._front_graphs = {}
        if noise_w is None:
            # The duration predictor's draw stays outside the graph: captured, it would be frozen
            # at its capture-time value (and a host-to-device copy is not capturable).  Same
            # draw as the eager path (CPU generator, then copied), fed to the graph as an input.
            noise_w = torch.randn(text_padded.size(0), 2, text_padded.size(1)).to(
                device=text_padded.device, dtype=torch.float32)
        if torch.cuda.is_current_stream_capturing():        # inside somebody else's capture: no graph of our own
            return self._token_stage(text_padded, text_lengths, speaker_id, noise_w, length_scale, noise_scale_w)
        key = (tuple(text_padded.shape), str(text_padded.device), float(length_scale), float(noise_scale_w),
               self._front_weights_key())
        call = self._front_graphs.pop(key, None)
        if call is None:
            try:
                call = GraphedCall(lambda t, l, s, n: self._token_stage(t, l, s, n, length_scale, noise_scale_w),
                                   text_padded, text_lengths, speaker_id, noise_w)
            except Exception as e:                           # capture refused (allocator, library): stay eager
                import warnings
                warnings.warn(f"token-level graph capture failed ({type(e).__name__}: {e}); running it eagerly")
                self.use_front_graph = False
                return self._token_stage(text_padded, text_lengths, speaker_id, noise_w, length_scale, noise_scale_w)
            while len(self._front_graphs) >= self.max_front_graphs:
                self._front_graphs.pop(next(iter(self._front_graphs)))
        self._front_graphs[key] = call                       # most recently used last
        return call(text_padded, text_lengths, speaker_id, noise_w)     # consumed at once by the caller

    @torch.no_grad()
    def frame_prior(self, text_padded, text_lengths, speaker_id, length_scale=1, noise_scale_w=0.8, noise_w=None):
        """Lines 169-182 of ``text_to_speech``: token-level prior, sampled durations, hard
        monotonic path, frame-level (m_p, logs_p).  Returns (m_p, logs_p, spec_mask, g)."""
        stage = self._token_stage_graphed if (self.use_front_graph and text_padded.is_cuda) else self._token_stage
        m_p, logs_p, text_mask, w_ceil = stage(text_padded, text_lengths, speaker_id, noise_w, length_scale,
                                               noise_scale_w)
        g = self.speaker_ref(speaker_id) if m_p.is_cuda else self.embed_speaker(speaker_id)
        if m_p.is_cuda and self.use_expand_kernel and m_p.shape[2] <= native.EXPAND_MAX_TOKENS:
            # the alignment as a gather: the only host round trip is the number of frames
            frames = int(torch.clamp_min(w_ceil.sum([1, 2]), 1).max().item())
            B, _, Tx = w_ceil.shape
            m_f, logs_f, fmask = native.expand_prior(w_ceil.reshape(B, Tx).float().contiguous(),
                                                     m_p.float().contiguous(), logs_p.float().contiguous(), frames)
            return m_f, logs_f, fmask.unsqueeze(1).to(text_mask.dtype), g
        y_lengths = torch.clamp_min(torch.sum(w_ceil, [1, 2]), 1).long()
        spec_mask = torch.unsqueeze(sequence_mask(y_lengths, None), 1).to(text_mask.dtype)
        node_mask = torch.unsqueeze(text_mask, 2) * torch.unsqueeze(spec_mask, -1)
        path = generate_path(w_ceil, node_mask).squeeze(1)
        m_p = torch.matmul(path, m_p.transpose(1, 2)).transpose(1, 2)
        logs_p = torch.matmul(path, logs_p.transpose(1, 2)).transpose(1, 2)
        return m_p, logs_p, spec_mask, g

    @torch.no_grad()
    def text_to_speech(self, text_padded, text_lengths, speaker_id, noise_scale=.667, length_scale=1,
                       noise_scale_w=0.8, max_len=None, noise_w=None, noise_z=None):
        """Same positional arguments and defaults as the reference; ``noise_w`` / ``noise_z``
        optionally pin the two ``randn`` draws (duration predictor, latent)."""
        m_p, logs_p, spec_mask, g = self.frame_prior(text_padded, text_lengths, speaker_id,
                                                     length_scale=length_scale, noise_scale_w=noise_scale_w,
                                                     noise_w=noise_w)
        return self.synthesize_from_prior(m_p, logs_p, spec_mask, g, noise=noise_z, noise_scale=noise_scale,
                                          max_len=max_len)
