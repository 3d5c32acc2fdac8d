"""Respaced linear-beta DDPM schedule in float64 (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows reference diffusion_and_flow/gaussian_diffusion.py:104-121 (named linear schedule),
:159-209 (derived tables), respace.py:12-62 (space_timesteps) and :73-87 (re-derived betas).
"""
import numpy as np


def space_timesteps(num_timesteps, section_counts):
    """respace.py:12-62: which of the original steps are kept."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[4:])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    per, extra = divmod(num_timesteps, len(section_counts))
    start, steps = 0, []
    for i, cnt in enumerate(section_counts):
        size = per + (1 if i < extra else 0)
        if size < cnt:
            raise ValueError(f"cannot divide section of {size} steps into {cnt}")
        stride = 1 if cnt <= 1 else (size - 1) / (cnt - 1)
        cur = 0.0
        for _ in range(cnt):
            steps.append(start + round(cur))
            cur += stride
        start += size
    return set(steps)


def make_schedule(timestep_respacing, diffusion_steps=1000):
    """Tables the sampler reads, indexed by respaced step i (all float64 numpy)."""
    if timestep_respacing in (None, ""):
        timestep_respacing = [diffusion_steps]
    scale = 1000 / diffusion_steps
    base_betas = np.linspace(scale * 0.0001, scale * 0.02, diffusion_steps, dtype=np.float64)
    base_acp = np.cumprod(1.0 - base_betas, axis=0)
    keep = space_timesteps(diffusion_steps, timestep_respacing)
    last, betas, tmap = 1.0, [], []
    for i, acp in enumerate(base_acp):
        if i in keep:
            betas.append(1 - acp / last)
            last = acp
            tmap.append(i)
    betas = np.array(betas, dtype=np.float64)
    alphas = 1.0 - betas
    acp = np.cumprod(alphas, axis=0)
    acp_prev = np.append(1.0, acp[:-1])
    post_var = betas * (1.0 - acp_prev) / (1.0 - acp)
    return {
        "timestep_map": np.array(tmap, dtype=np.int64),
        "betas": betas,
        "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / acp),
        "sqrt_recipm1_alphas_cumprod": np.sqrt(1.0 / acp - 1),
        "posterior_mean_coef1": betas * np.sqrt(acp_prev) / (1.0 - acp),
        "posterior_mean_coef2": (1.0 - acp_prev) * np.sqrt(alphas) / (1.0 - acp),
        "posterior_log_variance_clipped": np.log(np.append(post_var[1], post_var[1:])),
        "log_betas": np.log(betas),
        # ModelVarType.FIXED_LARGE (gaussian_diffusion.py:324-327)
        "fixed_large_log_variance": np.log(np.append(post_var[1], betas[1:])),
    }
