"""ODE sampling for the flow-matching models: the part of torchdiffeq.odeint the reference's
`run_sampling` uses (test.py:214-250: `odeint(f, x, t_span, rtol, atol, method)` with f(t, x) = model.forward(x, t, ..)).

torchdiffeq is not available offline, so the solvers are restated from their published definitions - PARITY
UNPINNED for the solver layer (the model evaluation under it is pinned as everywhere else):
  euler, midpoint   the classical one- and two-stage fixed-grid methods on the grid `t`;
  rk4               torchdiffeq's fixed-grid RK4 is the 3/8 rule, reproduced here;
  dopri5            Dormand-Prince 5(4) with torchdiffeq's step controller (RMS error norm, safety 0.9, growth in
                    [0.2, 10], Hairer's initial step).  Deviation, stated: steps are clipped to end on the output
                    times instead of overshooting them and evaluating a dense-output polynomial; results agree with
                    any other solver of the same tolerance to that tolerance.
State updates run on the device through codlad_ode_combine (one launch per stage); only the scalar error norm of
the adaptive method comes back to the host.
"""
import ctypes as C

import torch

from .. import _lib

# Dormand-Prince 5(4)
_DP_ALPHA = (1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0)
_DP_BETA = ((1 / 5,),
            (3 / 40, 9 / 40),
            (44 / 45, -56 / 15, 32 / 9),
            (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
            (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656),
            (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84))
_DP_C_SOL = (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0.0)
_DP_C_ERR = (35 / 384 - 1951 / 21600, 0.0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720,
             -2187 / 6784 - -12231 / 42400, 11 / 84 - 649 / 6300, -1.0 / 60.0)


def combine(y, ks, coefs, h):
    """y + sum_j ks[j] * (coefs[j] * h) on the device (codlad_ode_combine)."""
    if not y.is_cuda:
        raise RuntimeError("odeint (codlad_amd) runs on the MI355X only")
    y = y.contiguous().float()
    ks = [k.contiguous().float() for k in ks]
    assert all(k.shape == y.shape for k in ks) and 1 <= len(ks) <= 7
    out = torch.empty_like(y)
    ptrs = (C.c_void_p * len(ks))(*[k.data_ptr() for k in ks])
    cf = (C.c_float * len(ks))(*[float(c) for c in coefs])
    rc = _lib.lib().codlad_ode_combine(_lib.ptr(y), ptrs, cf, len(ks), C.c_float(float(h)), y.numel(), _lib.ptr(out),
                                       _lib.stream_ptr(y.device))
    _lib.check(rc, "codlad_ode_combine")
    return out


def _tt(t, like):
    return torch.as_tensor(t, dtype=torch.float32, device=like.device)


def _fixed_step(func, method, t0, dt, t1, y0):
    if method == "euler":
        return combine(y0, [func(_tt(t0, y0), y0)], [1.0], dt)
    if method == "midpoint":
        k1 = func(_tt(t0, y0), y0)
        y_mid = combine(y0, [k1], [0.5], dt)
        return combine(y0, [func(_tt(t0 + 0.5 * dt, y0), y_mid)], [1.0], dt)
    if method == "rk4":                                     # the 3/8 rule (torchdiffeq rk4_alt_step_func)
        k1 = func(_tt(t0, y0), y0)
        k2 = func(_tt(t0 + dt / 3, y0), combine(y0, [k1], [1 / 3], dt))
        k3 = func(_tt(t0 + dt * 2 / 3, y0), combine(y0, [k2, k1], [1.0, -1 / 3], dt))
        k4 = func(_tt(t1, y0), combine(y0, [k1, k2, k3], [1.0, -1.0, 1.0], dt))
        return combine(y0, [k1, k2, k3, k4], [0.125, 0.375, 0.375, 0.125], dt)
    raise NotImplementedError(f"odeint method {method!r}: euler, midpoint, rk4 and dopri5 are built")


def _rms(x):
    return float(x.double().pow(2).mean().sqrt())


def _initial_step(func, t0, y0, f0, rtol, atol, order=4):
    """Hairer, Norsett, Wanner: Solving ODEs I, II.4 (what torchdiffeq's _select_initial_step follows)."""
    scale = atol + y0.abs() * rtol
    d0, d1 = _rms(y0 / scale), _rms(f0 / scale)
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    f1 = func(_tt(t0 + h0, y0), combine(y0, [f0], [1.0], h0))
    d2 = _rms((f1 - f0) / scale) / h0
    h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** (1.0 / (order + 1))
    return min(100 * h0, h1)


def _dopri5(func, y0, ts, rtol, atol, max_steps=100000):
    out = [y0]
    t, y = float(ts[0]), y0
    f = func(_tt(t, y), y)
    h = _initial_step(func, t, y, f, rtol, atol)
    n_steps = 0
    for t_end in [float(v) for v in ts[1:]]:
        while t < t_end:
            n_steps += 1
            if n_steps > max_steps:
                raise RuntimeError("dopri5: max_steps exceeded")
            clipped = h >= t_end - t
            hh = (t_end - t) if clipped else h
            ks = [f]
            for a, beta in zip(_DP_ALPHA, _DP_BETA):
                ks.append(func(_tt(t + a * hh, y), combine(y, ks, beta, hh)))
            y1 = combine(y, ks, _DP_C_SOL, hh)
            err = combine(torch.zeros_like(y), ks, _DP_C_ERR, hh)
            tol = atol + rtol * torch.maximum(y.abs(), y1.abs())
            ratio = _rms(err / tol)
            if ratio <= 1.0:                                # accept; FSAL: k7 = f(t + h, y1)
                # a clipped step lands on the output time exactly (t + (t_end - t) can be one ulp short, which would
                # cost a further step of ~1e-16 and restart the controller from there)
                t, y, f = (t_end if clipped else t + hh), y1, ks[6]
            # torchdiffeq _optimal_step_size: safety 0.9, ifactor 10, dfactor 0.2 (1 when the step is accepted)
            if ratio == 0.0:
                factor = 10.0
            else:
                factor = min(10.0, max(0.9 / ratio ** 0.2, 1.0 if ratio < 1.0 else 0.2))
            # the controller's own step survives a clip at an output time: an ACCEPTED clipped step says nothing
            # against h (only that hh <= h was fine too), so the next interval starts from h, not from the remainder
            h = max(h, hh * factor) if (clipped and ratio <= 1.0) else hh * factor
        out.append(y)
    return torch.stack(out)


def odeint(func, y0, t, rtol=1e-7, atol=1e-9, method=None, options=None):
    """-> [len(t), *y0.shape]: y at the times t (same call shape as torchdiffeq.odeint; func(t, y) -> dy/dt)."""
    method = method or "dopri5"
    ts = [float(v) for v in torch.as_tensor(t).reshape(-1).tolist()]
    assert len(ts) >= 2 and all(b > a for a, b in zip(ts, ts[1:])), "t must be increasing"
    y0 = y0.contiguous().float()
    if method == "dopri5":
        return _dopri5(func, y0, ts, rtol, atol)
    out, y = [y0], y0
    for t0, t1 in zip(ts, ts[1:]):
        y = _fixed_step(func, method, t0, t1 - t0, t1, y)
        out.append(y)
    return torch.stack(out)
