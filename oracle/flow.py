"""TEST INFRASTRUCTURE - CPU restatement of the ODE sampling path of the flow-matching models (reference
test.py:214-250): the model evaluated at fractional times (pinned by goldens g12 from the reference model itself)
under fixed-grid / adaptive Runge-Kutta solvers restated from their published definitions (torchdiffeq is absent:
the solver layer is PARITY UNPINNED, see codlad_amd/diffusion_and_flow/ode.py for what is reproduced and the one
stated deviation).  Plain double-free torch CPU fp32 arithmetic, one separately rounded op per term.
"""
import torch

from codlad_amd.diffusion_and_flow import ode as _tables   # Butcher tableau constants only (data)
from . import denoiser


def velocity_fn(sd, cg_xyz, cg_z, mask):
    """f(t, x) = model.forward(x, t, y=None, mask, batch) of a flow-matching model (3 output channels)."""
    def f(t, x):
        tt = torch.as_tensor(t, dtype=torch.float32).reshape(1).expand(x.shape[0])
        return denoiser.forward(sd, x, tt, cg_xyz, cg_z, mask)
    return f


def _combine(y, ks, coefs, h):
    h32 = torch.tensor(h, dtype=torch.float32)
    acc = ks[0] * (torch.tensor(coefs[0], dtype=torch.float32) * h32)
    for k, c in zip(ks[1:], coefs[1:]):
        acc = acc + k * (torch.tensor(c, dtype=torch.float32) * h32)
    return y + acc


def odeint_fixed(func, y0, ts, method):
    ts = [float(v) for v in ts]
    out, y = [y0], y0
    for t0, t1 in zip(ts, ts[1:]):
        dt = t1 - t0
        if method == "euler":
            y = _combine(y, [func(t0, y)], [1.0], dt)
        elif method == "midpoint":
            k1 = func(t0, y)
            y = _combine(y, [func(t0 + 0.5 * dt, _combine(y, [k1], [0.5], dt))], [1.0], dt)
        elif method == "rk4":
            k1 = func(t0, y)
            k2 = func(t0 + dt / 3, _combine(y, [k1], [1 / 3], dt))
            k3 = func(t0 + dt * 2 / 3, _combine(y, [k2, k1], [1.0, -1 / 3], dt))
            k4 = func(t1, _combine(y, [k1, k2, k3], [1.0, -1.0, 1.0], dt))
            y = _combine(y, [k1, k2, k3, k4], [0.125, 0.375, 0.375, 0.125], dt)
        else:
            raise KeyError(method)
        out.append(y)
    return torch.stack(out)


def odeint_dopri5(func, y0, ts, rtol, atol):
    """Same controller as codlad_amd.diffusion_and_flow.ode._dopri5, on CPU tensors."""
    rms = lambda x: float(x.double().pow(2).mean().sqrt())  # noqa: E731
    ts = [float(v) for v in ts]
    t, y = ts[0], y0
    f = func(t, y)
    scale = atol + y.abs() * rtol
    d0, d1 = rms(y / scale), rms(f / scale)
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    f1 = func(t + h0, _combine(y, [f], [1.0], h0))
    d2 = rms((f1 - f) / scale) / h0
    h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** 0.2
    h = min(100 * h0, h1)
    out, n_eval = [y0], 2
    for t_end in ts[1:]:
        while t < t_end:
            clipped = h >= t_end - t
            hh = (t_end - t) if clipped else h
            ks = [f]
            for a, beta in zip(_tables._DP_ALPHA, _tables._DP_BETA):
                ks.append(func(t + a * hh, _combine(y, ks, beta, hh)))
                n_eval += 1
            y1 = _combine(y, ks, _tables._DP_C_SOL, hh)
            err = _combine(torch.zeros_like(y), ks, _tables._DP_C_ERR, hh)
            ratio = rms(err / (atol + rtol * torch.maximum(y.abs(), y1.abs())))
            if ratio <= 1.0:
                t, y, f = (t_end if clipped else t + hh), y1, ks[6]
            factor = 10.0 if ratio == 0.0 else min(10.0, max(0.9 / ratio ** 0.2, 1.0 if ratio < 1.0 else 0.2))
            h = max(h, hh * factor) if (clipped and ratio <= 1.0) else hh * factor
        out.append(y)
    return torch.stack(out), n_eval
