"""Drop-in for the evaluation helpers of the reference's test loop (reference test.py:97-166, called at
test.py:589-593), computed on the GPU by one fused pass (`codlad_eval_metrics`) instead of ~60 ATen ops
and, for the clash metric, a sort-based `unique` per call.  Same names, argument order and return values
as the reference functions; tensors must be on the GPU (no CPU path).

`evaluate` computes all eight numbers in one launch; the per-metric functions call it with the lists
they need.  The clash list (rows of cat(edge_list, nbr_list) occurring once) depends on the topology
only and is cached per (edge_list, nbr_list) pair.
"""
import ctypes as C

import torch

from . import _lib

NAMES = ("loss_bond", "loss_angle", "loss_torsion", "loss_xyz", "loss_graph", "loss_nbr", "loss_inter", "loss_pi_pi")
_CLASH_CACHE = {}


def _need_cuda(t, what):
    if t is not None and not t.is_cuda:
        raise RuntimeError(f"{what} must be a CUDA tensor: the metrics run on the MI355X only")


def clash_list(edge_list, nbr_list):
    """Rows of cat(edge_list, nbr_list) that occur exactly once (reference test.py:121-123)."""
    key = (edge_list.data_ptr(), edge_list._version, tuple(edge_list.shape),
           nbr_list.data_ptr(), nbr_list._version, tuple(nbr_list.shape))
    if key not in _CLASH_CACHE:
        if len(_CLASH_CACHE) > 16:
            _CLASH_CACHE.clear()
        both = torch.cat((edge_list, nbr_list)).to(torch.int64)
        big = int(both.max()) + 1 if both.numel() else 1
        codes, counts = (both[:, 0] * big + both[:, 1]).unique(return_counts=True)   # sorted like unique(dim=0)
        once = codes[counts == 1]
        _CLASH_CACHE[key] = (torch.stack((once // big, once % big), 1).contiguous(), edge_list, nbr_list)
    return _CLASH_CACHE[key][0]


def evaluate(xyz_recon=None, xyz=None, edge_list=None, clash=None, bb_NO_list=None, interaction_list=None,
             pi_pi_list=None, ic_recon=None, ic=None, mask=None):
    """-> float32 tensor [8] on the device, in the order of NAMES; absent inputs leave their entries 0
    (or NaN for the reconstruction losses, which divide by the mask count like the reference)."""
    dev = next(t for t in (xyz_recon, ic_recon) if t is not None).device
    lib = _lib.lib()
    keep = []

    def f32(t, what):
        if t is None:
            return None, 0
        _need_cuda(t, what)
        t = t.detach().to(torch.float32).contiguous()
        keep.append(t)
        return t, t.shape[0]

    def i64(t, what, width):
        if t is None or t.shape[0] == 0:
            return None, 0
        _need_cuda(t, what)
        t = t.detach().to(torch.int64).contiguous()
        assert t.dim() == 2 and t.shape[1] == width, what
        keep.append(t)
        return t, t.shape[0]

    m = _lib.MetricInputs()
    xr, n_atoms = f32(xyz_recon, "xyz_recon")
    xt, _ = f32(xyz if xyz is not None else xyz_recon, "xyz")
    m.xyz_recon, m.xyz, m.n_atoms = _lib.ptr(xr), _lib.ptr(xt), n_atoms
    for field, count, t, width in (("edge_list", "n_edges", edge_list if xyz is not None else None, 2),
                                   ("clash_list", "n_clash", clash, 2), ("bb_NO_list", "n_bb", bb_NO_list, 2),
                                   ("interaction_list", "n_inter", interaction_list, 2),
                                   ("pi_pi_list", "n_pipi", pi_pi_list, 4)):
        tt, n = i64(t, field, width)
        setattr(m, field, _lib.ptr(tt))
        setattr(m, count, n)
    if ic_recon is not None:
        a, _ = f32(ic.reshape(-1, 3), "ic")
        b, n_ic = f32(ic_recon.reshape(-1, 3), "ic_recon")
        k, _ = f32(mask.reshape(-1), "mask")
        assert a.shape == b.shape and k.shape[0] == n_ic
        m.ic, m.ic_recon, m.ic_mask, m.n_ic = _lib.ptr(a), _lib.ptr(b), _lib.ptr(k), n_ic
    out = torch.zeros(8, dtype=torch.float32, device=dev)
    scratch = torch.empty(lib.codlad_metrics_scratch_bytes(), dtype=torch.uint8, device=dev)
    rc = lib.codlad_eval_metrics(C.byref(m), _lib.ptr(out), _lib.ptr(scratch), _lib.stream_ptr(dev))
    _lib.check(rc, "codlad_eval_metrics")
    return out


def recon_result(ic_recon, ic, mask_):
    o = evaluate(ic_recon=ic_recon, ic=ic, mask=mask_)
    return o[0], o[1], o[2]


def xyz_result(xyz_recon, xyz):
    return evaluate(xyz_recon=xyz_recon, xyz=xyz)[3]


def ged_result(xyz_recon, xyz, edge_list):
    return evaluate(xyz_recon=xyz_recon, xyz=xyz, edge_list=edge_list)[4]


def clash_result(edge_list, nbr_list, xyz_recon, bb_NO_list):
    return evaluate(xyz_recon=xyz_recon, clash=clash_list(edge_list, nbr_list), bb_NO_list=bb_NO_list)[5]


def inter_result(interaction_list, pi_pi_list, xyz_recon):
    o = evaluate(xyz_recon=xyz_recon, interaction_list=interaction_list, pi_pi_list=pi_pi_list)
    return o[6], o[7]


def all_results(ic_recon, ic, mask_, xyz_recon, xyz, edge_list, nbr_list, bb_NO_list, interaction_list, pi_pi_list):
    """The five calls of reference test.py:589-593 as one launch -> dict of 0-d device tensors."""
    o = evaluate(xyz_recon=xyz_recon, xyz=xyz, edge_list=edge_list, clash=clash_list(edge_list, nbr_list),
                 bb_NO_list=bb_NO_list, interaction_list=interaction_list, pi_pi_list=pi_pi_list,
                 ic_recon=ic_recon, ic=ic, mask=mask_)
    return dict(zip(NAMES, o))
