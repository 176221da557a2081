"""GPU parity of the token-level kernels (duration predictor stacks and coupling flows, through the
C ABI) against the same modules' layer-by-layer PyTorch form on the CPU, which the CPU suite pins
to the reference's goldens (test_front_half.py)."""
import pytest
import torch

from vits_amd import native, synth
from vits_amd.module.vits_generator import VitsGenerator

pytestmark = pytest.mark.gpu
SEED = 999
DEV = "cuda:0"


@pytest.fixture(scope="module")
def gens():
    cpu = VitsGenerator(40, 100)
    sd = synth.full_state_dict(cpu.state_dict(), SEED)
    cpu.load_state_dict(sd, strict=True)
    gpu = VitsGenerator(40, 100)
    gpu.load_state_dict(sd, strict=True)
    return cpu.eval(), gpu.to(DEV).eval()


def _inputs(B, T, lens, seed):
    g = torch.Generator().manual_seed(seed)
    enc = torch.randn(B, 192, T, generator=g)
    mask = synth.length_mask(lens, T)
    spk = torch.randn(B, 256, 1, generator=g) * 0.3
    noise = torch.randn(B, 2, T, generator=g)
    return enc * mask, mask, spk, noise


@pytest.mark.parametrize("B,T,lens", [(1, 43, [43]), (3, 87, [87, 60, 1]), (2, 200, [200, 131]), (1, 1, [1]),
                                      (2, 64, [64, 33]), (1, 65, [65]), (2, 333, [333, 77])])
def test_duration_predictor_reverse(gens, B, T, lens):
    """One window (T <= 64), window edges (64 / 65) and several windows with recomputed reach;
    ragged masks; compared with the CPU layer-by-layer pass on the same noise."""
    cpu, gpu = gens
    enc, mask, spk, noise = _inputs(B, T, lens, 100 + T)
    want = cpu.stochastic_duration_predictor(enc, mask, speaker_id_embedded=spk, reverse=True, noise_scale=0.8,
                                             noise=noise)
    sdp = gpu.stochastic_duration_predictor
    assert sdp.use_token_kernels
    got = sdp(enc.to(DEV), mask.to(DEV), speaker_id_embedded=spk.to(DEV), reverse=True, noise_scale=0.8,
              noise=noise.to(DEV))
    assert got.shape == want.shape
    err = (got.cpu() - want).abs().max().item()
    assert err <= 2e-5 * max(1.0, want.abs().max().item()), err
    # the stock PyTorch form on the GPU agrees too (the switch the module keeps for comparison)
    sdp.use_token_kernels = False
    try:
        stock = sdp(enc.to(DEV), mask.to(DEV), speaker_id_embedded=spk.to(DEV), reverse=True, noise_scale=0.8,
                    noise=noise.to(DEV))
    finally:
        del sdp.use_token_kernels
    assert (stock.cpu() - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())


def test_duration_predictor_without_speaker(gens):
    cpu, gpu = gens
    enc, mask, _, noise = _inputs(2, 50, [50, 20], 7)
    want = cpu.stochastic_duration_predictor(enc, mask, reverse=True, noise_scale=1.0, noise=noise)
    got = gpu.stochastic_duration_predictor(enc.to(DEV), mask.to(DEV), reverse=True, noise_scale=1.0,
                                            noise=noise.to(DEV))
    assert (got.cpu() - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())


def test_spline_tails_are_identity(gens):
    """Noise far outside the spline's [-5, 5] support passes through the coupling flows unchanged
    apart from the final affine."""
    cpu, gpu = gens
    enc, mask, spk, noise = _inputs(1, 30, [30], 3)
    noise = noise * 40.0
    want = cpu.stochastic_duration_predictor(enc, mask, speaker_id_embedded=spk, reverse=True, noise_scale=1.0,
                                             noise=noise)
    got = gpu.stochastic_duration_predictor(enc.to(DEV), mask.to(DEV), speaker_id_embedded=spk.to(DEV),
                                            reverse=True, noise_scale=1.0, noise=noise.to(DEV))
    assert (got.cpu() - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())


def test_dds_rejects_bad_arguments(gens):
    _, gpu = gens
    plan = gpu.stochastic_duration_predictor._plans.get(torch.device(DEV))
    c = plan.couplings[0]
    z = torch.zeros(1, 2, 20, device=DEV)
    g = torch.zeros(1, 192, 20, device=DEV)
    mask = torch.ones(1, 20, device=DEV)
    kw = dict(pass_channel=0, num_bins=10, tail_bound=5.0)
    with pytest.raises(native.NativeError, match="overlaps"):
        native.dds_coupling(z, c["pre_w"], c["pre_b"], g, c["layers"], 3, 3, c["proj_wt"], c["proj_b"], mask, z, **kw)
    out = torch.empty_like(z)
    with pytest.raises(native.NativeError, match="num_bins"):
        native.dds_coupling(z, c["pre_w"], c["pre_b"], g, c["layers"], 3, 3, c["proj_wt"], c["proj_b"], mask, out,
                            pass_channel=0, num_bins=8, tail_bound=5.0)
    with pytest.raises(native.NativeError, match="taps"):
        native.dds_coupling(z, c["pre_w"], c["pre_b"], g, c["layers"], 3, 5, c["proj_wt"], c["proj_b"], mask, out, **kw)
    with pytest.raises(native.NativeError, match="pass_channel"):
        native.dds_coupling(z, c["pre_w"], c["pre_b"], g, c["layers"], 3, 3, c["proj_wt"], c["proj_b"], mask, out,
                            pass_channel=2, num_bins=10, tail_bound=5.0)
