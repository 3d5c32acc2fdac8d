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


# Covalent cut-off radii by atomic number 1..107 (the reference's COVCUTOFFTABLE, utils/protein_module.py:128-234: a
# table of constants, kept as data)
COV_CUTOFF = (0.23, 0.93, 0.68, 0.35, 0.83, 0.68, 0.68, 0.68, 0.64, 1.12, 0.97, 1.1, 1.35, 1.2, 0.75, 1.02, 0.99, 1.57,
              1.33, 0.99, 1.44, 1.47, 1.33, 1.35, 1.35, 1.34, 1.33, 1.5, 1.52, 1.45, 1.22, 1.17, 1.21, 1.22, 1.21, 1.91,
              1.47, 1.12, 1.78, 1.56, 1.48, 1.47, 1.35, 1.4, 1.45, 1.5, 1.59, 1.69, 1.63, 1.46, 1.46, 1.47, 1.4, 1.98,
              1.67, 1.34, 1.87, 1.83, 1.82, 1.81, 1.8, 1.8, 1.99, 1.79, 1.76, 1.75, 1.74, 1.73, 1.72, 1.94, 1.72, 1.57,
              1.43, 1.37, 1.35, 1.37, 1.32, 1.5, 1.5, 1.7, 1.55, 1.54, 1.54, 1.68, 1.7, 2.4, 2.0, 1.9, 1.88, 1.79, 1.61,
              1.58, 1.55, 1.53, 1.51, 1.5, 1.5, 1.5, 1.5, 1.5, 1.5, 1.5, 1.5, 1.57, 1.49, 1.43, 1.41)


def bond_graph_counts(xyz, xyz_recon, num_atoms, atomic_nums, scale=1.3):
    """int32 [n_struct, 6] on the device: per structure {bonds in xyz, bonds in xyz_recon, differing pairs} over all
    atoms and over heavy atoms (codlad_bond_graph_counts)."""
    _need_cuda(xyz, "xyz")
    _need_cuda(xyz_recon, "xyz_recon")
    dev = xyz.device
    z = torch.as_tensor(atomic_nums).to(torch.int64).cpu()
    if int(z.min()) < 1 or int(z.max()) > len(COV_CUTOFF):
        raise ValueError("atomic number outside the covalent cut-off table (1..107)")
    radius = torch.tensor(COV_CUTOFF, dtype=torch.float32)[z - 1].to(dev)
    heavy = (z != 1).to(torch.int32).to(dev)
    na = [int(n) for n in torch.as_tensor(num_atoms).tolist()]
    assert sum(na) == xyz.shape[0] == xyz_recon.shape[0] == z.numel()
    ptr = torch.zeros(len(na) + 1, dtype=torch.int32)
    ptr[1:] = torch.cumsum(torch.tensor(na, dtype=torch.int64), 0).to(torch.int32)
    ptr = ptr.to(dev)
    counts = torch.empty(len(na), 6, dtype=torch.int32, device=dev)
    a = xyz.detach().to(torch.float32).contiguous()
    b = xyz_recon.detach().to(torch.float32).contiguous()
    rc = _lib.lib().codlad_bond_graph_counts(_lib.ptr(a), _lib.ptr(b), _lib.ptr(radius), _lib.ptr(heavy), _lib.ptr(ptr),
                                             len(na), max(na), C.c_float(scale), _lib.ptr(counts), _lib.stream_ptr(dev))
    _lib.check(rc, "codlad_bond_graph_counts")
    return counts


def valid_ratio_and_cut_off_result(xyz, xyz_recon, num_atoms, atomic_nums):
    """Drop-in for reference test.py:168-188: per structure, whether the bond graph of the reconstruction equals the
    reference's (heavy atoms / all atoms) and the relative difference of their bond counts, returned as the
    reference returns them: four lists with one entry per structure (the graph ratios as one-element lists, as
    count_valid_graphs does).  The ASE Atoms objects of the reference are plain containers here."""
    cnt = bond_graph_counts(xyz, xyz_recon, num_atoms, atomic_nums).cpu().to(torch.int64)
    heavy_valid, all_valid, heavy_ged, all_ged = [], [], [], []
    for ref_a, gen_a, diff_a, ref_h, gen_h, diff_h in cnt.tolist():
        heavy_valid.append(1.0 if diff_h == 0 else 0.0)
        all_valid.append(1.0 if diff_a == 0 else 0.0)
        # (ref_graph - gen_graph).sum().abs() / ref_graph.sum() on the full 0/1 matrices (each pair counted twice)
        heavy_ged.append([(torch.tensor(2 * (ref_h - gen_h)).abs() / torch.tensor(2 * ref_h)).item()])
        all_ged.append([(torch.tensor(2 * (ref_a - gen_a)).abs() / torch.tensor(2 * ref_a)).item()])
    return heavy_valid, all_valid, heavy_ged, all_ged


def superposed_rmsd(a, b):
    """Minimal RMSD of two conformations [n_atoms, 3] after optimal rigid superposition (what mdtraj's md.rmsd
    returns for single-frame trajectories, reference test.py:52-53, 74-75): Kabsch, RMSD^2 = (|a|^2 + |b|^2 -
    2 (s1 + s2 + sign(det) s3)) / n with s the singular values of the 3x3 covariance of the centred coordinates.
    mdtraj is not available offline: PARITY UNPINNED (published algorithm)."""
    a = a.to(torch.float64) - a.to(torch.float64).mean(0)
    b = b.to(torch.float64) - b.to(torch.float64).mean(0)
    cov = (a.t() @ b).cpu()
    u, sv, vt = torch.linalg.svd(cov)
    d = torch.sign(torch.linalg.det(u @ vt))
    e0 = float((a * a).sum() + (b * b).sum())
    msd = max(e0 - 2.0 * float(sv[0] + sv[1] + d * sv[2]), 0.0) / a.shape[0]
    return msd ** 0.5


def compute_div(gen_structures, ref_structure):
    """Diversity score of reference test.py:37-95: 1 - (mean RMSD of every generated frame to the mean generated
    structure) / (mean RMSD of every generated frame to the reference frame).  gen_structures: list (ensemble members)
    of [n_frames, n_atoms, 3]; ref_structure [n_frames, n_atoms, 3]."""
    gen = [torch.as_tensor(g) for g in gen_structures]
    ref = torch.as_tensor(ref_structure)
    mean_gen = torch.stack(gen).mean(0)
    to_ref = [superposed_rmsd(g[p], ref[p]) for g in gen for p in range(g.shape[0])]
    to_mean = [superposed_rmsd(g[p], mean_gen[p]) for g in gen for p in range(g.shape[0])]
    return 1.0 - (sum(to_mean) / len(to_mean)) / (sum(to_ref) / len(to_ref))
