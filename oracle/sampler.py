"""Ancestral DDPM sampling loop on CPU (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows reference diffusion_and_flow/gaussian_diffusion.py:496-547 (loop), :404-449 (p_sample),
:262-360 (p_mean_variance: EPSILON / START_X mean, LEARNED_RANGE / FIXED_SMALL / FIXED_LARGE variance,
clip_denoised), :362-367, :240-260 and respace.py:124-129 (timestep_map).  C = 3.
"""
import torch

from . import denoiser
from .schedule import make_schedule


def _coef(arr, i):
    # gaussian_diffusion.py:728-740: float64 table -> .float() at use
    return torch.from_numpy(arr)[i].float()


def ddpm_update(sched, i, x, model_out, noise, return_x_start=False, predict_xstart=False, var_type="learned_range",
                clip_denoised=False):
    """One reverse step given the denoiser output; returns x_{i-1}.  The branches of p_mean_variance
    (gaussian_diffusion.py:303-349): model_out [N,L,6] = mean | variance logits with the learned(-range) variance,
    [N,L,3] with var_type "fixed_small" / "fixed_large"; predict_xstart: the mean channels are x_0 (START_X) instead
    of the noise; clip_denoised: pred_xstart clamped into [-1, 1]."""
    C = x.shape[-1]
    if var_type in ("learned_range", "learned"):
        out, v = torch.split(model_out, C, dim=-1)
        min_log = _coef(sched["posterior_log_variance_clipped"], i)
        max_log = _coef(sched["log_betas"], i)
        frac = (v + 1) / 2
        log_var = frac * max_log + (1 - frac) * min_log
    else:
        out = model_out
        assert out.shape[-1] == C
        log_var = _coef(sched["fixed_large_log_variance" if var_type == "fixed_large" else "posterior_log_variance_clipped"], i)
    if predict_xstart:
        x0 = out
    else:
        x0 = _coef(sched["sqrt_recip_alphas_cumprod"], i) * x - _coef(sched["sqrt_recipm1_alphas_cumprod"], i) * out
    if clip_denoised:
        x0 = x0.clamp(-1, 1)
    mean = _coef(sched["posterior_mean_coef1"], i) * x0 + _coef(sched["posterior_mean_coef2"], i) * x
    nonzero = 0.0 if i == 0 else 1.0
    sample = mean + nonzero * torch.exp(0.5 * log_var) * noise
    return (sample, x0) if return_x_start else sample


def p_sample_loop(sd, num_steps, z, noise, cg_xyz, cg_z, mask, hoist_features=False,
                  return_traj=False, self_condition=False, predict_xstart=False, var_type="learned_range",
                  clip_denoised=False):
    """z [N,L,3] = x_T; noise [T,N,L,3] consumed in loop order (first entry at i = T-1).

    hoist_features=False recomputes the CA features every step like the reference does;
    True computes them once (same values: they depend on cg_xyz only)."""
    sched = make_schedule(str(num_steps))
    T = len(sched["betas"])
    feats = denoiser.ca_features(sd, cg_xyz, mask.int()) if hoist_features else None
    x = z
    traj = []
    x_start = None       # gaussian_diffusion.py:530-547: each step sees the previous pred_xstart
    for k, i in enumerate(range(T - 1, -1, -1)):
        t = torch.full((x.shape[0],), int(sched["timestep_map"][i]), dtype=torch.int64)
        out = denoiser.forward(sd, x, t, cg_xyz, cg_z, mask, features=feats,
                               x_self_cond=x_start if self_condition else None)
        x, x_start = ddpm_update(sched, i, x, out, noise[k], return_x_start=True, predict_xstart=predict_xstart,
                                 var_type=var_type, clip_denoised=clip_denoised)
        if return_traj:
            traj.append(x)
    return (x, traj) if return_traj else x
