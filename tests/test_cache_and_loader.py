"""Edges of the host side around the kernels: the packed-weight cache must follow the module's
parameters (a second ``load_state_dict``, a move to another device), a checkpoint written the way
the reference writes ``netG_cpu.pth`` must load with the safe loader, a flow with an odd number
of coupling layers, the front-half graph with its own noise draw, and the environment switches
that choose between equivalent launch plans."""
import os

import pytest
import torch

from oracle import vits_oracle as O
from vits_amd import synth
from vits_amd.module.model_component.flow import Flow
from vits_amd.module.vits_generator import HotPathGenerator, VitsGenerator

SEED = 999
DEV = "cuda:0"


def test_checkpoint_round_trip_with_the_safe_loader(tmp_path):
    """``torch.save(netG.state_dict())`` (vits_train.py:252-256) -> ``torch.load(weights_only=True)``
    -> strict load: all 858 keys, same order, same values."""
    gen = VitsGenerator(40, 100)
    sd = synth.full_state_dict(gen.state_dict(), SEED)
    gen.load_state_dict(sd, strict=True)
    path = tmp_path / "netG_cpu.pth"
    torch.save(gen.to("cpu").state_dict(), path)
    loaded = torch.load(path, map_location="cpu", weights_only=True)
    assert len(loaded) == 858 and list(loaded) == list(sd)
    fresh = VitsGenerator(40, 100)
    fresh.load_state_dict(loaded, strict=True)
    for k, v in fresh.state_dict().items():
        assert torch.equal(v, sd[k]), k
    bad = dict(loaded)
    bad.pop("decoder.conv1d_post.weight")
    with pytest.raises(RuntimeError):
        VitsGenerator(40, 100).load_state_dict(bad, strict=True)


def _inputs(B=2, T=21):
    mask = synth.length_mask([T, T - 6][:B], T)
    z = synth.latent(SEED, "cache.z", B, T) * mask
    ids = torch.tensor([5, 71][:B])
    return z, mask, ids


def _run(gen, z, mask, ids, dev=DEV):
    return gen.flow_reverse_decode(z.to(dev), mask.to(dev), gen.embed_speaker(ids.to(dev))).cpu()


@pytest.mark.gpu
def test_second_load_state_dict_rebuilds_the_packed_weights():
    """forward -> load other weights in place -> forward: the result is that of a fresh module
    with the other weights (and differs from the first result), then back again."""
    torch.set_grad_enabled(False)
    sd_a = synth.hot_path_state_dict(SEED, with_posterior=False)
    sd_b = synth.hot_path_state_dict(123, with_posterior=False)
    z, mask, ids = _inputs()
    gen = HotPathGenerator(100, with_posterior=False)
    gen.load_state_dict(sd_a, strict=True)
    gen = gen.to(DEV).eval()
    wav_a = _run(gen, z, mask, ids)
    gen.load_state_dict(sd_b, strict=True)
    wav_b = _run(gen, z, mask, ids)
    fresh = HotPathGenerator(100, with_posterior=False)
    fresh.load_state_dict(sd_b, strict=True)
    assert torch.equal(wav_b, _run(fresh.to(DEV).eval(), z, mask, ids))
    assert (wav_a - wav_b).abs().max().item() > 1e-3
    ref_b = O.flow_reverse_decode(z, mask, O.speaker_embed(ids, sd_b), sd_b)
    assert (wav_b - ref_b).abs().max().item() <= 1e-4
    gen.load_state_dict(sd_a, strict=True)
    assert torch.equal(_run(gen, z, mask, ids), wav_a)
    # a single parameter edited in place is seen too
    gen.decoder.conv1d_post.weight.mul_(0.5)
    assert not torch.equal(_run(gen, z, mask, ids), wav_a)


@pytest.mark.gpu
def test_module_moved_after_a_forward_still_works():
    """``.to("cpu")`` and back allocates new parameter storage: plans are rebuilt, results
    unchanged; on the CPU the module refuses to run (no fallback)."""
    from vits_amd import native
    torch.set_grad_enabled(False)
    sd = synth.hot_path_state_dict(SEED, with_posterior=False)
    z, mask, ids = _inputs()
    gen = HotPathGenerator(100, with_posterior=False)
    gen.load_state_dict(sd, strict=True)
    gen = gen.to(DEV).eval()
    wav = _run(gen, z, mask, ids)
    gen = gen.to("cpu")
    with pytest.raises(native.NativeError):
        gen.flow_reverse_decode(z, mask, gen.embed_speaker(ids))
    gen = gen.to(DEV)
    assert torch.equal(_run(gen, z, mask, ids), wav)


@pytest.mark.gpu
@pytest.mark.parametrize("n_flows", [1, 3])
def test_flow_with_an_odd_number_of_coupling_layers(n_flows):
    """After an odd number of flips the physical buffer is the flipped latent: one real flip at
    the end (the reference flips after every layer, flow.py:104-105)."""
    torch.set_grad_enabled(False)
    sd = synth.hot_path_state_dict(SEED, with_posterior=False)
    sub = {k: v for k, v in synth.strip_prefix(sd, "flow").items() if int(k.split(".")[1]) < 2 * n_flows}
    fl = Flow(n_flows=n_flows)
    fl.load_state_dict(sub, strict=True)
    fl = fl.to(DEV).eval()
    z, mask, ids = _inputs()
    g = O.speaker_embed(ids, sd)
    for rev in (False, True):
        got = fl(z.to(DEV), mask.to(DEV), g.to(DEV), reverse=rev).cpu()
        ref = O.flow(z, mask, g, sd, reverse=rev, n_flows=n_flows)
        assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item()), (n_flows, rev)


@pytest.mark.gpu
def test_front_graph_draws_fresh_duration_noise():
    """``text_to_speech`` with the token-level half replayed as a HIP graph and no pinned noise
    (the default call): capture succeeds, and two calls draw different durations because the
    draw happens outside the graph."""
    torch.set_grad_enabled(False)
    gen = VitsGenerator(40, 100)
    gen.load_state_dict(synth.full_state_dict(gen.state_dict(), SEED), strict=True)
    gen = gen.to(DEV).eval()
    gen.use_front_graph = True
    ids = torch.tensor([synth.TTS_SENTENCE_IDS], device=DEV)
    lens = torch.tensor([ids.shape[1]], device=DEV)
    spk = torch.tensor([9], device=DEV)
    torch.manual_seed(7)
    draws = []
    for _ in range(3):
        _, _, _, w_ceil = gen._token_stage_graphed(ids, lens, spk, None, 3, 0.8)
        draws.append(w_ceil.clone().cpu())
    assert len(gen._front_graphs) == 1
    assert not (torch.equal(draws[0], draws[1]) and torch.equal(draws[1], draws[2]))
    # the draw comes from the CPU generator exactly as in the eager path: same seed, same durations
    torch.manual_seed(11)
    _, _, _, w_graph = gen._token_stage_graphed(ids, lens, spk, None, 3, 0.8)
    w_graph = w_graph.clone().cpu()
    torch.manual_seed(11)
    _, _, _, w_eager = gen._token_stage(ids, lens, spk, None, 3, 0.8)
    assert torch.equal(w_graph, w_eager.cpu())
    wav = gen.text_to_speech(ids, lens, spk, length_scale=3)
    assert torch.isfinite(wav).all() and wav.shape[2] % 256 == 0


@pytest.mark.gpu
@pytest.mark.parametrize("var,value", [("VITS_AMD_PAIR_FUSION", "none"), ("VITS_AMD_PAIR_FUSION", "all"),
                                       ("VITS_AMD_WN_FUSION", "0"), ("VITS_AMD_BRANCH_STREAMS", "off"),
                                       ("VITS_AMD_BRANCH_STREAMS", "on")])
def test_launch_plan_switches_do_not_change_results(monkeypatch, var, value):
    """The switches pick between launch plans of the same arithmetic (fused pair kernel or two
    launches, fused WN layer or two launches, resblocks on one stream or three)."""
    torch.set_grad_enabled(False)
    sd = synth.hot_path_state_dict(SEED, with_posterior=False)
    z, mask, ids = _inputs(T=37)
    monkeypatch.setenv(var, value)
    gen = HotPathGenerator(100, with_posterior=False)
    gen.load_state_dict(sd, strict=True)
    wav = _run(gen.to(DEV).eval(), z, mask, ids)
    ref = O.flow_reverse_decode(z, mask, O.speaker_embed(ids, sd), sd)
    assert (wav - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
