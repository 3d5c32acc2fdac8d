"""Drop-in for the reference's `diffusion_and_flow.create_diffusion` (sampling half).

Same call surface as the reference (`diffusion_and_flow/__init__.py:10-60`): the returned object has
`.p_sample_loop(model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs, device,
progress)`, `.p_sample_loop_progressive`, `.p_sample`, the schedule tables and `timestep_map`.
The arithmetic is on the GPU: when `model` is the forward of a codlad_amd `ProteinMPNN_diffusion_new`
the whole loop is one `codlad_sample_loop` call; any other CUDA callable is stepped with
`codlad_ddpm_update`.  Training losses are out of scope.
"""
import enum

import torch

from .schedule import Tables, named_betas, space_timesteps


class ModelMeanType(enum.Enum):
    PREVIOUS_X = enum.auto()
    START_X = enum.auto()
    EPSILON = enum.auto()


class ModelVarType(enum.Enum):
    LEARNED = enum.auto()
    FIXED_SMALL = enum.auto()
    FIXED_LARGE = enum.auto()
    LEARNED_RANGE = enum.auto()


class SpacedDiffusion(Tables):
    """Respaced ancestral sampler (reference respace.py:65-114 + gaussian_diffusion.py:404-547) with every branch of
    p_mean_variance that `create_diffusion` can select (gaussian_diffusion.py:303-349): the model predicts the noise
    (EPSILON, the default) or x_0 (START_X, test.py --predict_xstart); the variance is the learned range
    (LEARNED_RANGE; LEARNED takes the same formula in the reference) or fixed (FIXED_SMALL / FIXED_LARGE, with a
    model whose head has no variance channels); pred_xstart is optionally clipped into [-1, 1] (clip_denoised)."""

    def __init__(self, use_timesteps, betas, model_mean_type=ModelMeanType.EPSILON,
                 model_var_type=ModelVarType.LEARNED_RANGE, loss_type=None, self_condition=False):
        if model_mean_type is ModelMeanType.PREVIOUS_X:
            raise NotImplementedError("ModelMeanType.PREVIOUS_X: create_diffusion never selects it and the reference's "
                                      "p_mean_variance has no branch for it either (gaussian_diffusion.py:343-349)")
        super().__init__(betas, set(use_timesteps))
        self.use_timesteps = set(use_timesteps)
        self.original_num_steps = len(betas)
        self.model_mean_type, self.model_var_type = model_mean_type, model_var_type
        # reference gaussian_diffusion.py:172, 530-547: each step is conditioned on the previous pred_xstart
        self.loss_type, self.self_condition = loss_type, bool(self_condition)

    @property
    def fixed_variance(self):
        return self.model_var_type in (ModelVarType.FIXED_SMALL, ModelVarType.FIXED_LARGE)

    def coefficients(self, clip_denoised):
        """The [T, 8] step table of the kernels for this sampler's branches (schedule.Tables.step_coefficients)."""
        var = {ModelVarType.FIXED_SMALL: "fixed_small", ModelVarType.FIXED_LARGE: "fixed_large"}.get(self.model_var_type,
                                                                                                      "learned_range")
        return self.step_coefficients(predict_xstart=self.model_mean_type is ModelMeanType.START_X, var_type=var,
                                      clip_denoised=bool(clip_denoised))

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _hip_module(model):
        from ..models.latent_model import ProteinMPNN_diffusion_new
        owner = getattr(model, "__self__", model)
        return owner if isinstance(owner, ProteinMPNN_diffusion_new) else None

    @staticmethod
    def _check_args(clip_denoised, denoised_fn, cond_fn):
        if denoised_fn is not None or cond_fn is not None:
            raise NotImplementedError("denoised_fn / cond_fn (arbitrary Python callables inside the step) are not used by "
                                      "the reference's sampling call (test.py:533) and are not built")

    def _draw_noise(self, x, generator=None):
        """T draws of randn_like(x), in loop order, consuming the device RNG stream exactly as the
        reference's per-step `th.randn_like(x)` does (gaussian_diffusion.py:440)."""
        return torch.stack([torch.randn(x.shape, device=x.device, dtype=x.dtype, generator=generator)
                            for _ in range(self.num_timesteps)])

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                      model_kwargs=None, device=None, progress=False, step_noise=None):
        """Returns x_0 with `shape`.  `step_noise` [T, *shape] optionally supplies the per-step noise
        explicitly (loop order) instead of drawing it from the device RNG."""
        self._check_args(clip_denoised, denoised_fn, cond_fn)
        model_kwargs = model_kwargs or {}
        mod = self._hip_module(model)
        if mod is None:
            final = None
            for final in self.p_sample_loop_progressive(model, shape, noise=noise, clip_denoised=clip_denoised,
                                                        model_kwargs=model_kwargs, device=device,
                                                        step_noise=step_noise):
                pass
            return final["sample"]
        if device is None:
            device = next(mod.parameters()).device
        img = noise if noise is not None else torch.randn(*shape, device=device)
        if not img.is_cuda:
            raise RuntimeError("p_sample_loop (codlad_amd) runs on the MI355X only")
        eps = step_noise if step_noise is not None else self._draw_noise(img)
        batch = model_kwargs["batch"]
        n_rep = img.shape[0] // int(batch["num_CGs"].shape[0])
        job, lens = mod.job_for(batch, n_rep)
        mod._check_mask(model_kwargs.get("mask"), lens, n_rep)
        if self.self_condition != mod.self_condition:
            raise ValueError("create_diffusion(self_condition=...) and the model's self_condition differ: the fused "
                             "loop conditions exactly when the model was built for it (test.py:297-303)")
        if len(set(lens)) != 1:
            raise NotImplementedError("fused loop on a padded mixed-length batch; pass equal-length "
                                      "structures per call (what the reference's loaders produce)")
        T = self.num_timesteps
        x0 = mod.engine().sample(job, img.reshape(-1, img.shape[-1]), eps.reshape(T, -1, img.shape[-1]), self,
                                 coef=self.coefficients(clip_denoised))
        return x0.view(img.shape)

    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None,
                                  cond_fn=None, model_kwargs=None, device=None, progress=False,
                                  step_noise=None):
        """Generic stepping for any CUDA model callable: model(x, t, **kwargs) -> [N,L,2C]."""
        self._check_args(clip_denoised, denoised_fn, cond_fn)
        model_kwargs = model_kwargs or {}
        img = noise if noise is not None else torch.randn(*shape, device=device)
        x_start = None
        for k, i in enumerate(range(self.num_timesteps - 1, -1, -1)):
            t = torch.tensor([i] * shape[0], device=img.device)
            eps = step_noise[k] if step_noise is not None else torch.randn_like(img)
            out = self.p_sample(model, img, t, clip_denoised=clip_denoised, model_kwargs=model_kwargs, noise=eps,
                                x_self_cond=x_start if self.self_condition else None)
            yield out
            img = out["sample"]
            x_start = out["pred_xstart"]

    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                 x_self_cond=None, noise=None):
        self._check_args(clip_denoised, denoised_fn, cond_fn)
        if not x.is_cuda:
            raise RuntimeError("p_sample (codlad_amd) runs on the MI355X only")
        from .. import _lib
        i = int(t.reshape(-1)[0])
        map_t = torch.tensor(self.timestep_map, device=t.device, dtype=t.dtype)[t]   # respace.py:124-129
        kwargs = dict(model_kwargs or {})
        if x_self_cond is not None:
            kwargs["x_self_cond"] = x_self_cond
        model_out = model(x, map_t, **kwargs)
        if noise is None:
            noise = torch.randn_like(x)
        C = x.shape[-1]
        assert C == 3 and model_out.shape[-1] == (C if self.fixed_variance else 2 * C), \
            "latent_size 3 only; a fixed-variance sampler takes a model without variance channels " \
            "(gaussian_diffusion.py:321-334)"
        import ctypes
        import numpy as np
        coef = np.ascontiguousarray(self.coefficients(clip_denoised)[i])
        xs = x.contiguous().float()
        out = torch.empty_like(xs)
        x_start = torch.empty_like(xs)
        rc = _lib.lib().codlad_ddpm_update(_lib.ptr(xs), _lib.ptr(model_out.contiguous().float()),
                                           _lib.ptr(noise.contiguous().float()),
                                           coef.ctypes.data_as(ctypes.c_void_p), xs.numel() // 3,
                                           _lib.ptr(out), _lib.ptr(x_start), _lib.stream_ptr(x.device))
        _lib.check(rc, "codlad_ddpm_update")
        return {"sample": out, "pred_xstart": x_start}


def create_diffusion(timestep_respacing, noise_schedule="linear", use_kl=False, rescale_learned_sigmas=False,
                     sigma_small=False, predict_xstart=False, learn_sigma=True, diffusion_steps=1000,
                     self_condition=False):
    if timestep_respacing is None or timestep_respacing == "":
        timestep_respacing = [diffusion_steps]
    # reference diffusion_and_flow/__init__.py:28-43
    mean_type = ModelMeanType.START_X if predict_xstart else ModelMeanType.EPSILON
    var_type = ModelVarType.LEARNED_RANGE if learn_sigma else (ModelVarType.FIXED_SMALL if sigma_small
                                                               else ModelVarType.FIXED_LARGE)
    return SpacedDiffusion(use_timesteps=space_timesteps(diffusion_steps, timestep_respacing),
                           betas=named_betas(noise_schedule, diffusion_steps),
                           model_mean_type=mean_type, model_var_type=var_type,
                           loss_type=None, self_condition=self_condition)
