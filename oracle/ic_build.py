"""Coordinates -> internal coordinates on CPU in float64 (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows reference utils/utils_ic.py:88-167 (unit_vector, angle_between, dihedral, get_sidechain_ic) and :170-196
(get_backbone_ic), and the reduction to [0, 2 pi) of utils/protein_module.py:773.

PIN: get_backbone_ic calls mdtraj.compute_distances / compute_angles / compute_dihedrals (mdtraj is absent here, so those
are restated from their published definitions: angle at the middle atom by arccos, dihedral by the atan2 form with the
IUPAC sign - the same quantities `dihedral` of utils_ic.py computes for the side chains).  What pins the restatement to
the reference is the inverse pair: reference-generated coordinates (tests/golden/g6_xyz_*, made by the reference's
ic_to_xyz from g5's internal coordinates) must survive xyz -> ic (here) -> xyz (oracle/vae_decode.ic_to_xyz, itself held
to the same goldens) - tests/test_dataset_builder.py.
"""
import numpy as np

TWO_PI = 2.0 * np.pi


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def angle_between(v1, v2):
    """utils_ic.py:95-106, row-wise."""
    return np.arccos(np.clip((_unit(v1) * _unit(v2)).sum(-1), -1.0, 1.0))


def dihedral(p0, p1, p2, p3):
    """utils_ic.py:109-138 (Praxeolitic formula), row-wise."""
    b0 = -1.0 * (p1 - p0)
    b1 = _unit(p2 - p1)
    b2 = p3 - p2
    v = b0 - b1 * (b0 * b1).sum(-1, keepdims=True)
    w = b2 - b1 * (b2 * b1).sum(-1, keepdims=True)
    x = (v * w).sum(-1)
    y = (np.cross(b1, v) * w).sum(-1)
    return np.arctan2(y, x)


def xyz_to_ic(xyz, quads):
    """xyz [F, n_atoms, 3], quads [Q, 4] (A1, A2, A3, A4; any index < 0: the slot is absent) -> ic [F, Q, 3] float64:
    (|A1 - A2|, angle(A1 - A2, A3 - A2), dihedral(A1, A2, A3, A4)), the last two modulo 2 pi."""
    xyz = np.asarray(xyz, dtype=np.float64)
    quads = np.asarray(quads)
    ok = (quads >= 0).all(-1)
    q = np.where(ok[:, None], quads, 0)
    a1, a2, a3, a4 = (xyz[:, q[:, k]] for k in range(4))
    with np.errstate(invalid="ignore", divide="ignore"):
        dist = np.sqrt(((a1 - a2) ** 2).sum(-1))
        ang = angle_between(a1 - a2, a3 - a2)
        tor = dihedral(a1, a2, a3, a4)
    tor = ((tor + np.pi) % TWO_PI) - np.pi                     # utils_ic.py:161
    ic = np.stack([dist, ang % TWO_PI, tor % TWO_PI], axis=-1)
    ic[:, ~ok] = 0.0
    return ic
