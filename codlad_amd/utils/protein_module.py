"""Host-side topology tables and trajectory output for the sampling path.

`traj_to_info` of the reference (utils/protein_module.py:434-494) turns an mdtraj topology into the index tables
`ic_to_xyz` consumes: `info = (permute, atom_idx, atom_orders)`.  mdtraj is only the PROVIDER of two lists there - the
residue names and, per residue, the atom names in file order - so the builder here takes those lists
(`info_from_residues`), and `read_pdb_topology` extracts them from a PDB file without mdtraj.  The reference writes
generated trajectories through mdtraj as .xtc and .pdb (test.py:787-803); `write_pdb` writes the multi-model PDB
directly (coordinates in Angstrom as they leave `ic_to_xyz`); the compressed .xtc format is written and read by
utils/xtc.py.

The data-set builder of the same reference file (`build_ic_peptide_dataset`, `build_split_dataset`, `CGDataset`,
`get_neighbor_list`, protein_module.py:567-951) lives in utils/dataset_builder.py and is re-exported here under the
reference's names.
"""
import numpy as np
import torch

from .ic_tables import atom_order_list, core_atoms
from .dataset_builder import (CGDataset, RES2IDX, THREE_LETTER_TO_ONE, Topology, build_ic_peptide_dataset,  # noqa: F401
                              build_split_dataset, read_pdb)
from .dataset_builder import neighbor_list as get_neighbor_list  # noqa: F401


def info_from_residues(res_names, atom_names):
    """res_names: residue names of the whole chain INCLUDING the two flanking residues; atom_names: per residue, the
    heavy-atom names in file order.  -> ((permute, atom_idx, atom_orders), n_cg) exactly as the reference's
    traj_to_info builds them for the interior residues: output atom p of the frame takes slot
    atom_idx[permute[p]] of the 14-slot-per-residue placement order (utils/utils_ic.py:267)."""
    n_cg = len(res_names)
    names, atoms = list(res_names[1:-1]), [list(a) for a in atom_names[1:-1]]
    permute, atom_idx, p_off, a_off = [], [], 0, 0
    for nm, present in zip(names, atoms):
        core = core_atoms[nm]
        if set(core) != set(present):
            raise ValueError(f"residue {nm}: atoms {sorted(present)} do not match the template {sorted(core)}")
        permute.append([core.index(a) + p_off for a in present])
        atom_idx.append(np.arange(a_off, a_off + len(present)))
        p_off += len(present)
        a_off += 14
    orders = np.zeros((10, len(names), 3), dtype=np.int64)
    orders[:, :, 1] = 1
    orders[:, :, 2] = 2                                            # unused side-chain slots: the reference's (0, 1, 2)
    for r, nm in enumerate(names):
        for i, trip in enumerate(atom_order_list[nm]):
            orders[i, r] = trip
    info = (torch.from_numpy(np.concatenate(permute).astype(np.int64)),
            torch.from_numpy(np.concatenate(atom_idx).astype(np.int64)), torch.from_numpy(orders))
    return info, n_cg


def read_pdb_topology(path):
    """Residue names and per-residue heavy-atom names (file order) of the first model of a PDB file, plus the
    coordinates [n_atoms, 3] of those atoms.  Hydrogens are skipped (the CG model is heavy-atom)."""
    res_names, atom_names, xyz, key = [], [], [], None
    with open(path) as f:
        for line in f:
            rec = line[:6]
            if rec.startswith("ENDMDL"):
                break
            if rec not in ("ATOM  ", "HETATM"):
                continue
            name, elem = line[12:16].strip(), line[76:78].strip()
            if elem == "H" or (not elem and name[:1] == "H"):
                continue
            k = (line[21], line[22:27])
            if k != key:
                key = k
                res_names.append(line[17:20].strip())
                atom_names.append([])
            atom_names[-1].append(name)
            xyz.append([float(line[30:38]), float(line[38:46]), float(line[46:54])])
    return res_names, atom_names, np.asarray(xyz, dtype=np.float32)


_ELEMENT = {"C": "C", "N": "N", "O": "O", "S": "S", "P": "P"}


def write_pdb(path, xyz, res_names, atom_names):
    """Multi-model PDB of the interior residues: xyz [n_frames, n_atoms, 3] (Angstrom, atom order of `info`),
    res_names / atom_names as given to info_from_residues (flanking residues included, not written)."""
    xyz = np.asarray(xyz, dtype=np.float64)
    if xyz.ndim == 2:
        xyz = xyz[None]
    names, atoms = list(res_names[1:-1]), [list(a) for a in atom_names[1:-1]]
    assert sum(len(a) for a in atoms) == xyz.shape[1], "atom count does not match the topology"
    with open(path, "w") as f:
        for m, frame in enumerate(xyz):
            f.write(f"MODEL     {m + 1:4d}\n")
            serial = 0
            for r, (nm, present) in enumerate(zip(names, atoms)):
                for a in present:
                    x, y, z = frame[serial]
                    serial += 1
                    label = a if len(a) == 4 else " " + a
                    f.write(f"ATOM  {serial % 100000:5d} {label:<4s} {nm:>3s} A{(r + 2) % 10000:4d}    "
                            f"{x:8.3f}{y:8.3f}{z:8.3f}  1.00  0.00          {_ELEMENT.get(a[0], a[0]):>2s}\n")
            f.write("ENDMDL\n")
        f.write("END\n")
