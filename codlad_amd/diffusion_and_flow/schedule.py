"""DDPM schedule tables for the sampler (host side, float64 numpy).

Same quantities as the reference's GaussianDiffusion / SpacedDiffusion constructors
(reference diffusion_and_flow/gaussian_diffusion.py:104-128,159-209; respace.py:12-62,73-87),
restricted to what ancestral sampling reads.
"""
import math

import numpy as np


def named_betas(schedule_name, n):
    if schedule_name == "linear":
        scale = 1000 / n
        return np.linspace(scale * 0.0001, scale * 0.02, n, dtype=np.float64)
    if schedule_name == "squaredcos_cap_v2":
        f = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2  # noqa: E731
        return np.array([min(1 - f((i + 1) / n) / f(i / n), 0.999) for i in range(n)])
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def space_timesteps(num_timesteps, section_counts):
    """Original-process steps kept by a respacing spec such as "100", "10,15,20" or "ddim50"."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    base, extra = divmod(num_timesteps, len(section_counts))
    first, kept = 0, []
    for k, count in enumerate(section_counts):
        size = base + (1 if k < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        kept += [first + round(c) for c in _strided(stride, count)]
        first += size
    return set(kept)


def _strided(stride, count):
    cur = 0.0
    for _ in range(count):
        yield cur
        cur += stride


class Tables:
    """Everything p_sample needs, indexed by respaced step."""

    def __init__(self, base_betas, use_timesteps):
        base_acp = np.cumprod(1.0 - np.asarray(base_betas, dtype=np.float64), axis=0)
        last, betas, self.timestep_map = 1.0, [], []
        for i, acp in enumerate(base_acp):
            if i in use_timesteps:
                betas.append(1 - acp / last)
                last = acp
                self.timestep_map.append(i)
        self.betas = betas = np.array(betas, dtype=np.float64)
        assert betas.ndim == 1 and (betas > 0).all() and (betas <= 1).all()
        self.num_timesteps = int(betas.shape[0])
        alphas = 1.0 - betas
        self.alphas_cumprod = acp = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = prev = np.append(1.0, acp[:-1])
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / acp)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / acp - 1)
        self.posterior_variance = pv = betas * (1.0 - prev) / (1.0 - acp)
        self.posterior_log_variance_clipped = (np.log(np.append(pv[1], pv[1:]))
                                               if len(pv) > 1 else np.array([]))
        self.posterior_mean_coef1 = betas * np.sqrt(prev) / (1.0 - acp)
        self.posterior_mean_coef2 = (1.0 - prev) * np.sqrt(alphas) / (1.0 - acp)

    def step_coefficients(self, predict_xstart=False, var_type="learned_range", clip_denoised=False):
        """[T, 8] fp32 rows for codlad_sample_loop / codlad_ddpm_update: the float64 table entries
        cast to fp32 exactly where the reference casts them (_extract_into_tensor: `.float()`).
        var_type "fixed_small" / "fixed_large" (gaussian_diffusion.py:321-334): column 4 holds the step's log
        variance itself; column 7 is the mode word of include/codlad_hip.h (codlad_ddpm_update)."""
        T = self.num_timesteps
        c = np.zeros((T, 8), dtype=np.float32)
        c[:, 0] = self.sqrt_recip_alphas_cumprod.astype(np.float32)
        c[:, 1] = self.sqrt_recipm1_alphas_cumprod.astype(np.float32)
        c[:, 2] = self.posterior_mean_coef1.astype(np.float32)
        c[:, 3] = self.posterior_mean_coef2.astype(np.float32)
        c[:, 4] = self.posterior_log_variance_clipped.astype(np.float32)
        c[:, 5] = np.log(self.betas).astype(np.float32)
        c[1:, 6] = 1.0  # no noise when t == 0
        if var_type == "fixed_large":
            c[:, 4] = np.log(np.append(self.posterior_variance[1], self.betas[1:])).astype(np.float32)
        elif var_type not in ("fixed_small", "learned_range", "learned"):
            raise ValueError(f"unknown variance type {var_type!r}")
        c[:, 7] = (1 if predict_xstart else 0) + (2 if var_type.startswith("fixed") else 0) + (4 if clip_denoised else 0)
        return c
