"""Coordinates -> data set: what the reference's `load_dataset` / `build_split_dataset` /
`build_ic_peptide_dataset` (utils/dataset_module.py:144-225, utils/protein_module.py:695-872, 923-951) produce from an
mdtraj trajectory, built here from a multi-model PDB file without mdtraj.

mdtraj is, on that path, a PROVIDER of five things: the per-atom table (name, element, residue name, residue number,
chain), the covalent bond graph of the standard residues, the frames' coordinates, and the distance / angle / dihedral
routines behind the internal coordinates.  Here: `read_pdb` + `Topology` give the table and the frames, `standard_bonds`
the bond graph (heavy-atom templates of the 20 residues + TPO / SEP, peptide bonds between consecutive residues of a
chain), and the internal coordinates are one launch of `codlad_xyz_to_ic` over all frames (the product path has no CPU
fallback: without the HIP library the builder raises).  Neighbour and interaction lists are torch index operations on
whatever device the frames are on, written to give the entries in the reference's order (torch.nonzero, row-major).

Units: Angstrom throughout, as the reference's frames are after `traj.xyz * 10` (protein_module.py:500).
"""
import ctypes as C

import numpy as np
import torch

from .ic_tables import atom_order_list, core_atoms

# utils/protein_module.py:46-93
THREE_LETTER_TO_ONE = {"ARG": "R", "HIS": "H", "HID": "H", "LYS": "K", "ASP": "D", "GLU": "E", "SER": "S", "THR": "T",
                       "ASN": "N", "GLN": "Q", "CYS": "C", "GLY": "G", "PRO": "P", "ALA": "A", "VAL": "V", "ILE": "I",
                       "LEU": "L", "MET": "M", "PHE": "F", "TYR": "Y", "TRP": "W", "TPO": "O", "SEP": "B"}
RES2IDX = {c: i for i, c in enumerate("NHAGRMSIELYDVWQKPFCTOB")}
ATOMIC_NUM = {"C": 6, "H": 1, "O": 8, "N": 7, "S": 16, "P": 15, "SE": 34}
BB_NAMES = ("CA", "C", "N", "O", "H")                      # bb_list, protein_module.py:119

# heavy-atom covalent bonds of a residue besides N-CA, CA-C, C-O (and CA-CB): what mdtraj's residue templates hold
_SIDE_BONDS = {
    "ALA": "", "GLY": "", "ARG": "CB-CG CG-CD CD-NE NE-CZ CZ-NH1 CZ-NH2", "ASP": "CB-CG CG-OD1 CG-OD2",
    "ASN": "CB-CG CG-OD1 CG-ND2", "CYS": "CB-SG", "GLU": "CB-CG CG-CD CD-OE1 CD-OE2", "GLN": "CB-CG CG-CD CD-OE1 CD-NE2",
    "HIS": "CB-CG CG-ND1 CG-CD2 ND1-CE1 CD2-NE2 CE1-NE2", "ILE": "CB-CG1 CB-CG2 CG1-CD1", "LEU": "CB-CG CG-CD1 CG-CD2",
    "LYS": "CB-CG CG-CD CD-CE CE-NZ", "MET": "CB-CG CG-SD SD-CE",
    "PHE": "CB-CG CG-CD1 CG-CD2 CD1-CE1 CD2-CE2 CE1-CZ CE2-CZ", "PRO": "CB-CG CG-CD CD-N", "SER": "CB-OG",
    "THR": "CB-OG1 CB-CG2",
    "TRP": "CB-CG CG-CD1 CG-CD2 CD1-NE1 NE1-CE2 CD2-CE2 CD2-CE3 CE2-CZ2 CE3-CZ3 CZ2-CH2 CZ3-CH2",
    "TYR": "CB-CG CG-CD1 CG-CD2 CD1-CE1 CD2-CE2 CE1-CZ CE2-CZ CZ-OH", "VAL": "CB-CG1 CB-CG2",
    "TPO": "CB-OG1 CB-CG2 OG1-P P-OE1 P-OE2 P-OE3", "SEP": "CB-OG OG-P P-OE1 P-OE2 P-OE3",
}


class Topology:
    """The per-atom table of a heavy-atom protein topology (the columns the reference reads off
    `top.to_dataframe()`): `name`, `element`, `resName`, `resSeq`, `chainID` per atom, and per residue its name and
    atom names in file order."""

    def __init__(self, res_names, atom_names, res_seqs=None, chain_ids=None, elements=None):
        self.res_names = [str(r) for r in res_names]
        self.atom_names = [list(a) for a in atom_names]
        n_res = len(self.res_names)
        self.res_seqs = list(range(1, n_res + 1)) if res_seqs is None else [int(s) for s in res_seqs]
        self.chain_ids = [0] * n_res if chain_ids is None else [int(c) for c in chain_ids]
        self.residue_of_atom = np.repeat(np.arange(n_res), [len(a) for a in self.atom_names])
        self.name = np.array([a for names in self.atom_names for a in names])
        self.resName = np.array(self.res_names)[self.residue_of_atom]
        self.resSeq = np.array(self.res_seqs)[self.residue_of_atom]
        self.chainID = np.array(self.chain_ids)[self.residue_of_atom]
        self.newSeq = self.resSeq + 5000 * self.chainID        # protein_module.py:707 (multi-chain residue key)
        if elements is None:
            elements = [a[0] for a in self.name]
        self.element = np.array([str(e).upper() for e in elements])
        self.first_atom = np.concatenate([[0], np.cumsum([len(a) for a in self.atom_names])])

    @property
    def n_atoms(self):
        return len(self.name)

    @property
    def n_residues(self):
        return len(self.res_names)

    def atom(self, residue, name):
        """Index of atom `name` of residue `residue` (-1: the residue does not list it)."""
        names = self.atom_names[residue]
        return int(self.first_atom[residue]) + names.index(name) if name in names else -1

    def select(self, name):
        return np.nonzero(self.name == name)[0]

    def atomic_nums(self):
        """get_atomNum (protein_module.py:419-428)."""
        return np.array([ATOMIC_NUM[e] for e in self.element])

    def subset_residues(self, first, last):
        """Topology of residues first .. last - 1."""
        sl = slice(first, last)
        a, b = int(self.first_atom[first]), int(self.first_atom[last])
        return Topology(self.res_names[sl], self.atom_names[sl], self.res_seqs[sl], self.chain_ids[sl],
                        list(self.element[a:b]))


def read_pdb(path):
    """-> (Topology, xyz float32 [n_models, n_atoms, 3], Angstrom).  Heavy atoms only (load_dataset's `mass > 1.1`
    selection, dataset_module.py:162-164); the topology is the first model's, every model must list the same atoms.
    Chains are numbered in order of appearance, as mdtraj's chainID column is."""
    res_names, atom_names, res_seqs, chain_ids, elements = [], [], [], [], []
    frames, cur, key, chains, first = [], [], None, {}, True
    with open(path) as f:
        for line in f:
            rec = line[:6]
            if rec.startswith("ENDMDL") or rec.startswith("END   ") or rec.strip() == "END":
                if cur:
                    frames.append(cur)
                    cur, first, key = [], False, None
                continue
            if rec not in ("ATOM  ", "HETATM"):
                continue
            name, elem = line[12:16].strip(), line[76:78].strip()
            if elem.upper() in ("H", "D") or (not elem and name.lstrip("0123456789")[:1] == "H"):
                continue
            if line[16] not in (" ", "A"):                  # alternate locations: the first one only
                continue
            if first:
                k = (line[21], line[22:27])
                if k != key:
                    key = k
                    res_names.append(line[17:20].strip())
                    atom_names.append([])
                    res_seqs.append(int(line[22:26]))
                    chain_ids.append(chains.setdefault(line[21], len(chains)))
                atom_names[-1].append(name)
                elements.append(elem if elem else name[0])
            cur.append((float(line[30:38]), float(line[38:46]), float(line[46:54])))
    if cur:
        frames.append(cur)
    n = sum(len(a) for a in atom_names)
    bad = [i for i, fr in enumerate(frames) if len(fr) != n]
    if bad:
        raise ValueError(f"{path}: model {bad[0] + 1} has {len(frames[bad[0]])} heavy atoms, the first has {n}")
    return Topology(res_names, atom_names, res_seqs, chain_ids, elements), np.asarray(frames, dtype=np.float32)


def alpha_mapping(top):
    """get_cg_and_xyz(cg_method='alpha') (protein_module.py:498-528): bead of every atom = index of its residue key in
    sorted order, walked the way the reference walks the table."""
    keys = sorted(set(top.newSeq.tolist()))
    out, j = [], 0
    for k in top.newSeq.tolist():
        if k != keys[j]:
            j += 1
        out.append(j)
    return torch.tensor(out, dtype=torch.int64)


def standard_bonds(top):
    """Covalent heavy-atom bonds (i < j, sorted) of the topology: residue templates + the peptide bond between
    consecutive residues of one chain - the edges of `top.to_bondgraph()` (protein_module.py:722-725)."""
    bonds = set()
    for r, nm in enumerate(top.res_names):
        pairs = [("N", "CA"), ("CA", "C"), ("C", "O")] + ([("CA", "CB")] if nm != "GLY" else [])
        pairs += [tuple(p.split("-")) for p in _SIDE_BONDS[nm].split()]
        for a, b in pairs:
            i, j = top.atom(r, a), top.atom(r, b)
            if i >= 0 and j >= 0:
                bonds.add((min(i, j), max(i, j)))
        if r + 1 < top.n_residues and top.chain_ids[r + 1] == top.chain_ids[r]:
            i, j = top.atom(r, "C"), top.atom(r + 1, "N")
            if i >= 0 and j >= 0:
                bonds.add((i, j))
    return torch.tensor(sorted(bonds), dtype=torch.int64).reshape(-1, 2)


def high_order_edges(edges, order, n_atoms):
    """get_high_order_edge (protein_module.py:536-564): pairs (i < j) within `order` bonds, row-major."""
    adj = torch.zeros(n_atoms, n_atoms)                      # float: the reachability products go through BLAS
    adj[edges[:, 0], edges[:, 1]] = 1
    adj[edges[:, 1], edges[:, 0]] = 1
    eye = torch.eye(n_atoms)
    mats = [eye, ((adj + eye) > 0).float()]
    for i in range(2, order + 1):
        mats.append(((mats[i - 1] @ mats[1]) > 0).float())
    om = torch.zeros_like(adj)
    for i in range(1, order + 1):
        om += (mats[i] - mats[i - 1]) * i
    return torch.triu(om).nonzero()


def ic_quads(top):
    """int32 [(n_res - 2) * 13, 4]: the atoms (A1, A2, A3, A4) of every internal coordinate of the interior residues, as
    atom indices of the full frame.  Backbone (get_backbone_ic, utils_ic.py:170-196): N = (N, CA, CA-, CA+),
    C = (C, CA, CA+, CA-), O = (O, C, CA, N); side chain j (get_sidechain_ic, utils_ic.py:141-167):
    (core[j + 4], core[order[2]], core[order[1]], core[order[0]]); absent slots -1."""
    ca = top.select("CA")
    if len(ca) != top.n_residues:
        raise ValueError("every residue needs exactly one CA")
    quads = -np.ones((top.n_residues - 2, 13, 4), dtype=np.int32)
    for r in range(1, top.n_residues - 1):
        nm = top.res_names[r]
        core = [top.atom(r, a) for a in core_atoms[nm]]
        if min(core) < 0:
            missing = [a for a, i in zip(core_atoms[nm], core) if i < 0]
            raise ValueError(f"residue {r} ({nm}) lacks {missing}")
        o, n, c, a = core[:4]
        q = quads[r - 1]
        q[0] = (n, a, ca[r - 1], ca[r + 1])
        q[1] = (c, a, ca[r + 1], ca[r - 1])
        q[2] = (o, c, a, n)
        for j, order in enumerate(atom_order_list[nm]):
            q[3 + j] = (core[j + 4], core[order[2]], core[order[1]], core[order[0]])
    return quads.reshape(-1, 4)


def xyz_to_ic(xyz, quads):
    """xyz [F, n_atoms, 3] float32 on the GPU, quads int32 [Q, 4] -> ic [F, Q, 3] (one launch of codlad_xyz_to_ic)."""
    from .. import _lib
    lib = _lib.lib()
    if xyz.device.type != "cuda":
        raise RuntimeError("xyz_to_ic runs on the GPU (codlad_xyz_to_ic); move the frames to a cuda device")
    xyz = xyz.contiguous().float()
    q = torch.as_tensor(quads, dtype=torch.int32, device=xyz.device).contiguous()
    if q.numel() and int(q.max()) >= xyz.shape[1]:
        raise ValueError("quad index beyond the frame's atoms")
    out = torch.empty(xyz.shape[0], q.shape[0], 3, device=xyz.device)
    _lib.check(lib.codlad_xyz_to_ic(C.c_void_p(xyz.data_ptr()), xyz.shape[0], xyz.shape[1], C.c_void_p(q.data_ptr()),
                                   q.shape[0], C.c_void_p(out.data_ptr()),
                                   C.c_void_p(torch.cuda.current_stream(xyz.device).cuda_stream)), "codlad_xyz_to_ic")
    return out


def neighbor_list(xyz, cutoff):
    """get_neighbor_list(undirected=True) (protein_module.py:567-584): pairs j > i with distance <= cutoff, row-major."""
    n = xyz.shape[0]
    dist = (xyz.expand(n, n, 3) - xyz.expand(n, n, 3).transpose(0, 1)).pow(2).sum(dim=2).sqrt()
    mask = dist <= cutoff
    mask.fill_diagonal_(False)
    nb = torch.nonzero(mask)
    return nb[nb[:, 1] > nb[:, 0]]


class _AtomFlags:
    """Per-atom booleans / keys of the (trimmed) table the interaction lists test (protein_module.py:806-868)."""

    def __init__(self, top, device):
        t = lambda a, dt=torch.bool: torch.as_tensor(np.asarray(a), dtype=dt, device=device)  # noqa: E731
        self.seq = t(top.newSeq, torch.int64)
        self.backbone = t(np.isin(top.name, BB_NAMES))
        self.nos = t(np.isin(top.element, ("N", "O", "S")))              # allow_list: every pair of N, O, S
        ring = np.isin(top.resName, ("PHE", "TYR", "TRP"))
        self.ring_cd1 = t(ring & (top.name == "CD1"))
        self.ring_cd2 = t(ring & (top.name == "CD2"))
        his = top.resName == "HIS"
        self.his_cd1, self.his_nd1 = t(his & (top.name == "CD1")), t(his & (top.name == "ND1"))
        self.is_n, self.is_o = t(top.name == "N"), t(top.name == "O")


def interaction_lists(xyz, fl):
    """One frame's (interaction_list, pi_pi_list, bb_NO_list) (protein_module.py:806-862)."""
    n = xyz.shape[0]
    dist = (xyz.expand(n, n, 3) - xyz.expand(n, n, 3).transpose(0, 1)).pow(2).sum(dim=2).sqrt()
    # hydrogen-bond / ion pairs: different, non-adjacent residues, not both backbone, both of N / O / S
    src, dst = torch.where((dist <= 3.3) & (dist > 0.93))
    s, d = fl.seq[src], fl.seq[dst]
    ok = (s != d) & (s != d + 1) & (d != s + 1) & (~fl.backbone[src] | ~fl.backbone[dst]) & fl.nos[src] & fl.nos[dst]
    inter = torch.stack([src[ok], dst[ok]], dim=-1)
    inter = inter[inter[:, 1] > inter[:, 0]]
    # ring pairs: (CD1, CD2) of the same aromatic residue, centres 2.0 .. 5.5 apart
    src, dst = torch.where((dist <= 8.0) & (dist > 1.5))
    ok = (fl.seq[src] == fl.seq[dst]) & ((fl.ring_cd1[src] & fl.ring_cd2[dst]) | (fl.his_cd1[src] & fl.his_nd1[dst]))
    e1, e2 = src[ok], dst[ok]
    centres = (xyz[e1] + xyz[e2]) / 2
    m = centres.shape[0]
    rd = (centres.expand(m, m, 3) - centres.expand(m, m, 3).transpose(0, 1)).pow(2).sum(dim=2).sqrt()
    a, b = torch.where((rd <= 5.5) & (rd >= 2.0))
    pipi = torch.stack([e1[a], e2[a], e1[b], e2[b]], dim=-1)
    pipi = pipi[pipi[:, 1] > pipi[:, 0]]
    pipi = pipi[pipi[:, 3] > pipi[:, 2]]
    pipi = pipi[pipi[:, 0] > pipi[:, 2]]
    # backbone N(i + 1) .. O(i)
    src, dst = torch.where((dist <= 4.0) & (dist > 1.5))
    ok = (fl.seq[src] == fl.seq[dst] + 1) & fl.is_n[src] & fl.is_o[dst]
    bb_no = torch.stack([src[ok], dst[ok]], dim=-1)
    return inter, pipi, bb_no


class CGDataset(torch.utils.data.Dataset):
    """protein_module.py:634-646: a dict of per-frame lists; item i = {key: list[i]}."""

    def __init__(self, props):
        self.props = props

    def __len__(self):
        return len(self.props["nxyz"])

    def __getitem__(self, idx):
        return {key: val[idx] for key, val in self.props.items()}

    def generate_neighbor_list(self, atom_cutoff, cg_cutoff, device="cpu"):
        """protein_module.py:651-691 (undirected, cut-off form)."""
        self.props["nbr_list"] = [neighbor_list(nxyz[:, 1:4].to(device), atom_cutoff).cpu() for nxyz in self.props["nxyz"]]
        self.props["CG_nbr_list"] = [neighbor_list(nxyz[:, 1:4].to(device), cg_cutoff).cpu()
                                     for nxyz in self.props["CG_nxyz"]]


def residue_masks(top):
    """(mask [(n_res - 2) * 13], mask_xyz) of protein_module.py:746-765: per interior residue ones for its heavy atoms
    besides CA, zeros for the residues that end a chain; mask_xyz = the atoms (trimmed numbering) of the chain ends
    that are not the first / last residue of the whole table."""
    keys = np.unique(top.newSeq)
    endpoints = set()
    for c in np.unique(top.chainID):
        ks = top.newSeq[top.chainID == c]
        endpoints.update((int(ks.min()), int(ks.max())))
    nfirst = int((top.newSeq == top.newSeq.min()).sum())
    mask = torch.zeros(len(keys) - 2, 13)
    for i in range(1, len(keys) - 1):
        if int(keys[i]) not in endpoints:
            mask[i - 1, :int((top.newSeq == keys[i]).sum()) - 1] = 1
    inner = endpoints - {int(top.newSeq.min()), int(top.newSeq.max())}
    idx = [int(i) for k in inner for i in np.nonzero(top.newSeq == k)[0]]       # the reference's set order is arbitrary too
    return mask.reshape(-1), torch.tensor(idx, dtype=torch.int64) - nfirst


def build_ic_peptide_dataset(mapping, traj, atom_cutoff, cg_cutoff, atomic_nums, top, order=1, cg_traj=None,
                             prot_idx=None, device="cuda"):
    """protein_module.py:695-872 with `top` a Topology and `traj` the frames [F, n_atoms, 3] (Angstrom).  The internal
    coordinates of all frames are one GPU launch; lists are built on `device` and stored on the host, like the
    reference's."""
    traj = torch.as_tensor(np.asarray(traj), dtype=torch.float32)
    n_frames = traj.shape[0]
    nfirst = int((top.newSeq == top.newSeq.min()).sum())
    nlast = int((top.newSeq == top.newSeq.max()).sum())
    inner = top.subset_residues(1, top.n_residues - 1)
    edges = high_order_edges(standard_bonds(inner), order, inner.n_atoms)
    ca = torch.as_tensor(top.select("CA"))
    zs = torch.as_tensor(np.asarray(atomic_nums), dtype=torch.float32)
    n_beads = len(torch.unique(mapping))
    cg_res = torch.tensor([RES2IDX[THREE_LETTER_TO_ONE[nm[:3]]] for nm in top.res_names[:n_beads]],
                          dtype=torch.float32).reshape(-1, 1)
    mask, mask_xyz = residue_masks(top)

    dev_xyz = traj.to(device)
    ic = xyz_to_ic(dev_xyz, ic_quads(top)).reshape(n_frames, -1, 13, 3)          # protein_module.py:770-773
    flags = _AtomFlags(inner, device)
    props = {k: [] for k in ("nxyz", "CG_nxyz", "OG_CG_nxyz", "num_atoms", "num_CGs", "CG_mapping", "bond_edge_list", "ic",
                             "mask", "mask_xyz_list", "prot_idx", "interaction_list", "pi_pi_list", "bb_NO_list")}
    for f in range(n_frames):
        xyz = traj[f]
        cg_xyz = torch.as_tensor(cg_traj[f], dtype=torch.float32) if cg_traj is not None else xyz[ca]
        og = torch.cat([cg_res, cg_xyz], dim=-1)
        props["nxyz"].append(torch.cat([zs[:, None], xyz], dim=-1)[nfirst:traj.shape[1] - nlast])
        props["OG_CG_nxyz"].append(og)
        props["CG_nxyz"].append(og[1:-1])
        props["num_atoms"].append(torch.tensor([traj.shape[1] - (nfirst + nlast)]))
        props["num_CGs"].append(torch.tensor([og.shape[0] - 2]))
        props["CG_mapping"].append(mapping[nfirst:traj.shape[1] - nlast] - 1)
        props["bond_edge_list"].append(edges)
        props["ic"].append(ic[f].cpu())
        props["mask"].append(mask)
        props["mask_xyz_list"].append(mask_xyz)
        props["prot_idx"].append(torch.tensor([float(prot_idx if prot_idx is not None else 0)]))
        inter, pipi, bb_no = interaction_lists(dev_xyz[f, nfirst:traj.shape[1] - nlast], flags)
        props["interaction_list"].append(inter.cpu())
        props["pi_pi_list"].append(pipi.cpu())
        props["bb_NO_list"].append(bb_no.cpu())
    dataset = CGDataset(props)
    dataset.generate_neighbor_list(atom_cutoff=atom_cutoff, cg_cutoff=cg_cutoff, device=device)
    return dataset


def build_split_dataset(top, frames, params, mapping=None, prot_idx=None, device="cuda"):
    """protein_module.py:923-951: alpha-carbon mapping, atomic numbers, data set."""
    mapping = alpha_mapping(top) if mapping is None else mapping
    dataset = build_ic_peptide_dataset(mapping, frames, params["atom_cutoff"], params["cg_cutoff"], top.atomic_nums(), top,
                                       order=params.get("edgeorder", 1), prot_idx=prot_idx, device=device)
    return dataset, mapping

