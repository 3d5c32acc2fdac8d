"""Next row 8f-4 (GPU): the reference's --self_condition variant on the HIP path, through the C ABI:
x_in over cat(x_self_cond, x) (latent_model.py:112-116, 210-212) and the sampler's pred_xstart feedback
(gaussian_diffusion.py:530-547), against goldens produced by the reference and against the oracle."""
import numpy as np
import pytest
import torch

from codlad_amd import synth
from codlad_amd.diffusion_and_flow import create_diffusion
from codlad_amd.diffusion_and_flow.schedule import Tables, named_betas, space_timesteps
from codlad_amd.engine import Denoiser
from codlad_amd.models.latent_model import MPNN_models
from tests import cases

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_err(a, b):
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    return float((a - b).abs().max() / b.abs().max())


@pytest.fixture(scope="module")
def sd_sc():
    return synth.denoiser_state_dict(cases.WEIGHT_SEED, self_condition=True)


def job_for(den, prot, B):
    frames = torch.from_numpy(prot["xyz_full"])[:, 1:-1]
    z = torch.from_numpy(prot["z_full"])[1:-1]
    st = den.prepare_structures([f for f in frames], [z for _ in frames])
    return den.make_job(st, list(range(B)))


@pytest.mark.parametrize("precision", ["f16x3", "f16x4", "f32"])
@pytest.mark.parametrize("name", list(cases.SELF_COND_CASES))
def test_forward_and_loop_match_reference(sd_sc, name, precision):
    L, B, seed, T = cases.SELF_COND_CASES[name]
    gold = np.load(cases.npz_path(f"g9_selfcond_{name}"))
    den = Denoiser(sd_sc, DEV, precision=precision)
    assert den.self_condition
    prot, batch, x, t, mask = cases.denoiser_inputs(L, B, seed)
    job = job_for(den, prot, B)
    xf = x.reshape(-1, 3).to(DEV)
    xsc = synth.gaussian((B, L, 3), 6000 + seed).reshape(-1, 3).to(DEV)
    assert rel_err(den.forward(job, xf, int(t[0])).view(B, L, 6), gold["out_none"]) < 1e-5
    assert rel_err(den.forward(job, xf, int(t[0]), x_self_cond=xsc).view(B, L, 6), gold["out_sc"]) < 1e-5
    zero = den.forward(job, xf, int(t[0]), x_self_cond=torch.zeros_like(xf))
    assert torch.equal(zero, den.forward(job, xf, int(t[0])))       # None == zeros, bit for bit
    z, eps = cases.loop_noise(T, B, L, seed)
    tb = Tables(named_betas("linear", 1000), space_timesteps(1000, str(T)))
    x0 = den.sample(job, z.reshape(-1, 3).to(DEV), eps.reshape(T, -1, 3).to(DEV), tb)
    assert rel_err(x0.view(B, L, 3), gold["sample"]) < 1e-4
    # stepwise (forward + ddpm_update, feeding pred_xstart by hand) == fused loop, bit for bit
    xs, x_start = z.reshape(-1, 3).to(DEV), None
    for k, i in enumerate(range(T - 1, -1, -1)):
        out = den.forward(job, xs, tb.timestep_map[i], x_self_cond=x_start)
        xs, x_start = den.ddpm_update(xs, out, eps[k].reshape(-1, 3).to(DEV), tb, i, return_x_start=True)
    assert torch.equal(xs, x0)


def test_plain_model_refuses_self_cond():
    den = Denoiser(synth.denoiser_state_dict(cases.WEIGHT_SEED), DEV)
    assert not den.self_condition
    L, B, seed = cases.DENOISER_CASES["L46_B2"]
    prot, batch, x, t, mask = cases.denoiser_inputs(L, B, seed)
    job = job_for(den, prot, B)
    with pytest.raises(ValueError):
        den.forward(job, x.reshape(-1, 3).to(DEV), 500, x_self_cond=x.reshape(-1, 3).to(DEV))


def test_dropin_api_like_test_py(sd_sc):
    """MPNN_models['mpnn_diffusion'](..., self_condition=True) + create_diffusion(self_condition=True)
    as reference test.py:192-197, 297-303 build them; fused loop == generic stepping == golden."""
    name = "L46_B2_T10"
    L, B, seed, T = cases.SELF_COND_CASES[name]
    gold = np.load(cases.npz_path(f"g9_selfcond_{name}"))
    model = MPNN_models["mpnn_diffusion"](input_size=3, unconditional=True, diffusion="diffusion", self_condition=True)
    model.load_state_dict(sd_sc, strict=True)
    model = model.to(DEV).eval()
    prot, batch, x, t, mask = cases.denoiser_inputs(L, B, seed)
    batch = {k: v.to(DEV) for k, v in batch.items()}
    z, eps = cases.loop_noise(T, B, L, seed)
    diffusion = create_diffusion(str(T), noise_schedule="linear", predict_xstart=False, rescale_learned_sigmas=False,
                                 self_condition=True)
    kwargs = dict(y=None, mask=mask.to(DEV), batch=batch)
    fused = diffusion.p_sample_loop(model.forward, z.shape, z.to(DEV), clip_denoised=False, model_kwargs=kwargs,
                                    step_noise=eps.to(DEV))
    assert rel_err(fused, gold["sample"]) < 1e-4
    last = None
    for last in diffusion.p_sample_loop_progressive(lambda a, b, **kw: model(a, b, **kw), z.shape, z.to(DEV),
                                                    clip_denoised=False, model_kwargs=kwargs, step_noise=eps.to(DEV)):
        assert last["pred_xstart"] is not None
    assert torch.equal(last["sample"], fused)
    with pytest.raises(ValueError):     # sampler and model must agree on self-conditioning
        create_diffusion(str(T), self_condition=False).p_sample_loop(
            model.forward, z.shape, z.to(DEV), clip_denoised=False, model_kwargs=kwargs, step_noise=eps.to(DEV))
