"""Next row 8f-4 on the GPU: a flow-matching model (velocity head, fractional timesteps) and the ODE samplers that
replace torchdiffeq.odeint in the reference's run_sampling (test.py:214-250).  The model evaluation is held to goldens
of the reference model; the fixed-grid solvers to trajectories integrated over the reference model; dopri5 (solver
unpinned, one stated deviation) to the oracle's restatement and to a fine fixed-grid reference solution."""
import numpy as np
import pytest
import torch

from codlad_amd import synth
from codlad_amd.diffusion_and_flow import ode
from codlad_amd.engine import Denoiser
from codlad_amd.models.latent_model import MPNN_models
from oracle import denoiser as oden
from oracle import flow as oflow
from tests import cases

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_err(a, b):
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture(scope="module")
def fsd():
    return synth.denoiser_state_dict(cases.WEIGHT_SEED, flow=True)


def engine_fn(den, prot, B):
    frames = torch.from_numpy(prot["xyz_full"])[:, 1:-1]
    z = torch.from_numpy(prot["z_full"])[1:-1]
    job = den.make_job(den.prepare_structures([f for f in frames], [z for _ in frames]), list(range(B)))
    return lambda t, y: den.forward(job, y, float(t)), job


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
@pytest.mark.parametrize("name", list(cases.FLOW_CASES))
def test_flow_model_forward_and_fixed_grid_sampling(fsd, name, precision):
    L, B, seed, times, n_steps = cases.FLOW_CASES[name]
    gold = np.load(cases.npz_path(f"g12_flow_{name}"))
    prot, batch, x, _t, mask = cases.denoiser_inputs(L, B, seed)
    den = Denoiser(fsd, DEV, precision=precision)
    assert den.weights.out_dim == 3
    f, job = engine_fn(den, prot, B)
    xd = x.reshape(-1, 3).to(DEV)
    for k, t in enumerate(times):
        v = f(t, xd)
        assert v.shape == (B * L, 3) and rel_err(v.view(B, L, 3), gold[f"v_t{k}"]) < 1e-5
    ts = torch.linspace(0, 1, n_steps + 1)
    for method in ("euler", "rk4"):
        y = ode.odeint(f, xd, ts, method=method)[-1]
        assert rel_err(y.view(B, L, 3), gold[method]) < 2e-5, method
    # the DDPM loop of the default (learned-variance) sampler refuses a 3-output model; a fixed-variance sampler takes one
    # (tests/test_dropin_api.py::test_sampler_branches_like_the_reference)
    from codlad_amd.diffusion_and_flow.schedule import Tables, named_betas, space_timesteps
    with pytest.raises(ValueError, match="fixed-variance sampler"):
        den.sample(job, xd, torch.zeros(10, B * L, 3, device=DEV), Tables(named_betas("linear", 1000), space_timesteps(1000, "10")))


def test_dropin_model_and_run_sampling_call_shape(fsd):
    """MPNN_models['mpnn_diffusion'](diffusion='fm') + odeint(f, x, t_span, rtol, atol, method) exactly as the
    reference's run_sampling builds them (test.py:221-236), scalar 0-d t included."""
    model = MPNN_models["mpnn_diffusion"](input_size=3, unconditional=True, diffusion="fm", self_condition=False)
    model.load_state_dict(fsd, strict=True)
    model = model.to(DEV).eval()
    L, B, seed, times, n_steps = cases.FLOW_CASES["L46_B2"]
    gold = np.load(cases.npz_path("g12_flow_L46_B2"))
    prot, batch, x, _t, mask = cases.denoiser_inputs(L, B, seed)
    batch = {k: (v.to(DEV) if hasattr(v, "to") else v) for k, v in batch.items()}
    fwd = lambda t, x_in: model.forward(x_in, t, None, mask=mask.to(DEV), batch=batch)  # noqa: E731
    v = fwd(torch.tensor(0.37, device=DEV), x.to(DEV))
    assert v.shape == (B, L, 3) and rel_err(v, gold["v_t1"]) < 1e-5
    t_span = torch.linspace(0, 1, n_steps + 1).to(DEV)
    traj = ode.odeint(fwd, x.to(DEV), t_span, rtol=1e-5, atol=1e-5, method="euler")
    assert traj.shape == (n_steps + 1, B, L, 3) and rel_err(traj[-1], gold["euler"]) < 2e-5


def test_dopri5_against_oracle_and_fine_grid(fsd):
    """Adaptive Dormand-Prince (reference default --method dopri5, atol = rtol = 1e-5): same accepted steps and result
    as the oracle's restatement of the same controller, and within the tolerance of a fine RK4 solution."""
    L, B, seed, times, n_steps = cases.FLOW_CASES["L46_B2"]
    prot, batch, x, _t, mask = cases.denoiser_inputs(L, B, seed)
    den = Denoiser(fsd, DEV, precision="f16x3")
    f, _job = engine_fn(den, prot, B)
    calls = []

    def counted(t, y):
        calls.append(float(t))
        return f(t, y)

    xd = x.reshape(-1, 3).to(DEV)
    y = ode.odeint(counted, xd, torch.tensor([0.0, 1.0]), rtol=1e-5, atol=1e-5, method="dopri5")[-1]
    cg_z, cg_xyz, m = oden.batch_to_dense(batch)
    fo = oflow.velocity_fn(fsd, cg_xyz, cg_z, mask)
    yo, n_eval = oflow.odeint_dopri5(fo, x, [0.0, 1.0], 1e-5, 1e-5)
    assert len(calls) == n_eval                                        # same step sequence
    assert rel_err(y.view(B, L, 3), yo[-1]) < 5e-5
    fine = ode.odeint(f, xd, torch.linspace(0, 1, 65), method="rk4")[-1]
    assert rel_err(y, fine) < 1e-3
