"""Drop-in for the torch half of the reference's utils/utils_ic.py: `ic_to_xyz(CG_nxyz, ic_recon, info)`
(reference utils/utils_ic.py:242-268) on the GPU via codlad_ic_to_xyz, plus the residue template tables."""
import ctypes as C

import torch

from .. import _lib
from ..engine import info_tables
from .ic_tables import core_atoms, atom_order_list  # noqa: F401  (same names as the reference)

EPS = 1e-8


def ic_to_xyz(CG_nxyz, ic_recon, info):
    """CG_nxyz [B, L+2, 4] (type, xyz; flanking residues included), ic_recon [B, L, 13, 3],
    info = (permute, atom_idx, atom_orders) -> xyz [B, n_atoms, 3]."""
    if not ic_recon.is_cuda:
        raise RuntimeError("ic_to_xyz (codlad_amd) runs on the MI355X only")
    dev = ic_recon.device
    B, L = ic_recon.shape[0], ic_recon.shape[1]
    assert CG_nxyz.shape[0] == B and CG_nxyz.shape[1] == L + 2
    ca = CG_nxyz[:, :, 1:].to(dev).contiguous().float()
    ic = ic_recon.contiguous().float()
    orders, slot_to_out, n_atoms = info_tables(info, L, dev)
    out = torch.empty(B, n_atoms, 3, dtype=torch.float32, device=dev)
    rc = _lib.lib().codlad_ic_to_xyz(_lib.ptr(ca), _lib.ptr(ic), _lib.ptr(orders), _lib.ptr(slot_to_out),
                                     B, L, n_atoms, _lib.ptr(out), _lib.stream_ptr(dev))
    _lib.check(rc, "codlad_ic_to_xyz")
    return out
