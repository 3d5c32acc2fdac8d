"""Coordinates -> data set (reference utils/protein_module.py:695-872, utils/utils_ic.py:141-196,
utils/dataset_module.py:144-225) without mdtraj.

Pin of the internal-coordinate definitions: reference-generated coordinates (g6_xyz_*: the reference's ic_to_xyz applied
to g5's internal coordinates) must come back from xyz -> ic -> xyz, and the bond lengths must be |g5's|.  The CPU tests
hold the oracle (oracle/ic_build.py) and the host tables to that; the GPU tests hold codlad_xyz_to_ic to the oracle and
run the builder and the CLI end to end from a PDB file."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import cases
from codlad_amd import synth
from codlad_amd.utils import dataset_builder as db
from codlad_amd.utils.protein_module import info_from_residues
from oracle import ic_build, vae_decode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def golden_frames(name):
    """(Topology incl. CA-only flanking residues, full frames [B, n_atoms + 2, 3], og [B, L + 2, 4], info, g5 ic)."""
    L, B, seed, vae_type = cases.DECODER_CASES[name]
    prot, batch, _latent, _dataname = cases.decoder_inputs(L, B, seed, vae_type)
    gold = np.load(cases.npz_path(f"g6_xyz_{name}"))["xyz"]
    names = [synth.IDX2THR[int(z)] for z in prot["z_full"]]
    atom_names = [["CA"]] + [synth.PDB_ATOM_ORDER[n] for n in names[1:-1]] + [["CA"]]
    og = batch["OG_CG_nxyz"].reshape(-1, L + 2, 4)
    full = np.concatenate([og[:, :1, 1:].numpy(), gold, og[:, -1:, 1:].numpy()], 1).astype(np.float32)
    g5 = np.load(cases.npz_path(f"g5_decode_{name}"))["ic_recon"].reshape(B, L, 13, 3)
    return db.Topology(names, atom_names), full, og, prot["info"], g5


def write_full_pdb(path, top, frames, chain_breaks=()):
    """Every residue of `top` (flanking ones too), one MODEL per frame; a new chain letter after each residue index in
    chain_breaks."""
    with open(path, "w") as f:
        for m, fr in enumerate(frames):
            f.write(f"MODEL     {m + 1:4d}\n")
            serial, chain = 0, 0
            for r, (nm, atoms) in enumerate(zip(top.res_names, top.atom_names)):
                if r in chain_breaks:
                    chain += 1
                for a in atoms:
                    x, y, z = fr[serial]
                    serial += 1
                    label = a if len(a) == 4 else " " + a
                    f.write(f"ATOM  {serial:5d} {label:<4s} {nm:>3s} {'ABCDEFG'[chain]}{r + 1:4d}    {x:8.3f}{y:8.3f}{z:8.3f}"
                            f"  1.00  0.00          {a[0]:>2s}\n")
            f.write("ENDMDL\n")
        f.write("END\n")


# ---------------------------------------------------------------------------------------------- CPU: oracle + host tables
@pytest.mark.parametrize("name", list(cases.DECODER_CASES))
def test_oracle_inverse_pair_on_reference_coordinates(name):
    top, full, og, info, g5 = golden_frames(name)
    B, L = og.shape[0], og.shape[1] - 2
    ic = ic_build.xyz_to_ic(full, db.ic_quads(top)).reshape(B, L, 13, 3)
    back = vae_decode.ic_to_xyz(og.double(), torch.from_numpy(ic), info)
    assert float((back - torch.from_numpy(full[:, 1:-1]).double()).abs().max()) < 1e-9
    present = ic[..., 0] > 0
    assert np.abs(ic[..., 0] - np.abs(g5[..., 0]))[present].max() < 5e-6          # bond lengths = the tables' |dist|
    assert (ic[..., 1:] >= 0).all() and (ic[..., 1:] < 2 * np.pi).all()
    absent = ~present
    assert (ic[absent] == 0).all()
    # which slots exist = the residue templates
    n_side = np.array([len(synth.PDB_ATOM_ORDER[nm]) - 1 for nm in top.res_names[1:-1]])
    assert (present.sum(-1) == n_side[None]).all()


def test_angle_and_dihedral_known_answers():
    """Textbook geometry: right angle, trans / cis / +90 degree dihedrals (IUPAC sign: looking down A2 -> A3, clockwise
    rotation of the far bond is positive)."""
    xyz = np.array([[[1, 0, 0], [0, 0, 0], [0, 1, 0], [-1, 1, 0],      # trans: 180
                     [1, 1, 0],                                          # cis with atoms 0,1,2 + this: 0
                     [0, 1, 1], [0, 1, -1]]], dtype=np.float64)
    quads = np.array([[0, 1, 2, 3], [0, 1, 2, 4], [0, 1, 2, 5], [0, 1, 2, 6], [0, 1, -1, 3]])
    ic = ic_build.xyz_to_ic(xyz, quads)[0]
    assert np.allclose(ic[:4, 0], 1.0) and np.allclose(ic[:4, 1], np.pi / 2)
    assert np.allclose(ic[0, 2], np.pi) and np.allclose(ic[1, 2], 0.0)
    assert np.allclose(sorted([ic[2, 2], ic[3, 2]]), [np.pi / 2, 3 * np.pi / 2])
    assert (ic[4] == 0).all()


def test_topology_tables_bonds_and_masks(tmp_path):
    top, full, _og, _info, _g5 = golden_frames("N6_L46_B3")
    path = str(tmp_path / "ens.pdb")
    write_full_pdb(path, top, full)
    top2, frames = db.read_pdb(path)
    assert top2.res_names == top.res_names and top2.atom_names == top.atom_names
    assert frames.shape == full.shape and np.abs(frames - full).max() <= 5.1e-4       # %8.3f
    assert top2.res_seqs == list(range(1, top.n_residues + 1)) and set(top2.chain_ids) == {0}
    assert list(top2.element) == [a[0] for a in top2.name]
    info_a, n_cg = info_from_residues(top2.res_names, top2.atom_names)
    assert n_cg == top.n_residues
    mapping = db.alpha_mapping(top2)
    assert mapping.tolist() == top2.residue_of_atom.tolist()
    inner = top2.subset_residues(1, top2.n_residues - 1)
    bonds = db.standard_bonds(inner)
    # a chain is a tree plus one extra bond per ring (PRO, PHE, TYR, HIS: 1; TRP: 2)
    rings = sum({"PRO": 1, "PHE": 1, "TYR": 1, "HIS": 1, "TRP": 2}.get(nm, 0) for nm in inner.res_names)
    assert bonds.shape[0] == inner.n_atoms - 1 + rings
    assert (bonds[:, 0] < bonds[:, 1]).all()
    # every template bond is a chemical bond in the reference-generated coordinates' own tables: |dist| of the decoder's
    # bond-length tables sits between 1.2 and 1.9 A, and the builder's bonds must be among the short pairs of the frame
    x = torch.from_numpy(full[0, 1:-1])
    d = (x[bonds[:, 0]] - x[bonds[:, 1]]).norm(dim=-1)
    assert bonds.shape[0] > 300 and torch.isfinite(d).all()
    e2 = db.high_order_edges(bonds, 2, inner.n_atoms)
    adj = torch.zeros(inner.n_atoms, inner.n_atoms)
    adj[bonds[:, 0], bonds[:, 1]] = 1
    adj = adj + adj.T
    two = ((adj @ adj) > 0) | (adj > 0)
    two.fill_diagonal_(False)
    assert torch.equal(e2, torch.triu(two).nonzero())
    assert torch.equal(db.high_order_edges(bonds, 1, inner.n_atoms), bonds)
    mask, mask_xyz = db.residue_masks(top2)
    assert mask.shape == ((top.n_residues - 2) * 13,) and mask_xyz.numel() == 0
    per_res = mask.reshape(-1, 13).sum(1)
    assert per_res.tolist() == [len(a) - 1 for a in inner.atom_names]


def test_multi_chain_masks(tmp_path):
    top, full, _og, _info, _g5 = golden_frames("N6_L46_B3")
    path = str(tmp_path / "two_chains.pdb")
    write_full_pdb(path, top, full[:1], chain_breaks=(20,))
    top2, _ = db.read_pdb(path)
    assert sorted(set(top2.chain_ids)) == [0, 1]
    assert top2.newSeq[top2.first_atom[20]] == 21 + 5000
    mask, mask_xyz = db.residue_masks(top2)
    m = mask.reshape(-1, 13)
    # the residues that end chain A (index 19) and start chain B (index 20) are masked out, their atoms listed
    assert m[18].sum() == 0 and m[19].sum() == 0 and m[17].sum() > 0 and m[20].sum() > 0
    want = np.concatenate([np.arange(top2.first_atom[19], top2.first_atom[21])]) - len(top2.atom_names[0])
    assert sorted(mask_xyz.tolist()) == want.tolist()
    inner = top2.subset_residues(1, top2.n_residues - 1)
    bonds = db.standard_bonds(inner)
    c19, n20 = inner.atom(18, "C"), inner.atom(19, "N")
    assert not ((bonds[:, 0] == c19) & (bonds[:, 1] == n20)).any()          # no peptide bond across the chain break


def _interaction_lists_by_name(xyz, top):
    """The reference's loops (protein_module.py:806-862) spelled with names, independently of the flag arrays."""
    n = xyz.shape[0]
    d = np.sqrt(((xyz[:, None] - xyz[None]) ** 2).sum(-1))
    inter, rings, bb = [], [], []
    for i in range(n):
        for j in range(n):
            si, sj = top.newSeq[i], top.newSeq[j]
            if 0.93 < d[i, j] <= 3.3 and si != sj and si != sj + 1 and sj != si + 1 \
                    and (top.name[i] not in db.BB_NAMES or top.name[j] not in db.BB_NAMES) \
                    and top.element[i] + top.element[j] in ('NO', 'ON', 'SN', 'NS', 'SO', 'OS', 'SS', 'NN', 'OO') and j > i:
                inter.append((i, j))
            if 1.5 < d[i, j] <= 8.0 and si == sj and top.resName[i] in ("PHE", "TYR", "TRP") and top.name[i] == "CD1" \
                    and top.name[j] == "CD2":
                rings.append((i, j))
            if 1.5 < d[i, j] <= 4.0 and si == sj + 1 and top.name[i] == "N" and top.name[j] == "O":
                bb.append((i, j))
    pipi = []
    cen = [(xyz[a] + xyz[b]) / 2 for a, b in rings]
    for p, (a, b) in enumerate(rings):
        for q, (c, e) in enumerate(rings):
            if 2.0 <= np.sqrt(((cen[p] - cen[q]) ** 2).sum()) <= 5.5 and b > a and e > c and a > c:
                pipi.append((a, b, c, e))
    return inter, pipi, bb


def test_interaction_lists_against_name_based_loops():
    top, full, _og, _info, _g5 = golden_frames("K3_L60_B2")
    inner = top.subset_residues(1, top.n_residues - 1)
    # compact the chain so that side chains of distant residues meet (the golden frames are extended)
    rng = np.random.default_rng(3)
    xyz = (full[0, 1:-1] * 0.35 + rng.normal(0, 0.3, (inner.n_atoms, 3))).astype(np.float32)
    fl = db._AtomFlags(inner, "cpu")
    inter, pipi, bb = db.interaction_lists(torch.from_numpy(xyz), fl)
    w_inter, w_pipi, w_bb = _interaction_lists_by_name(xyz.astype(np.float64), inner)
    assert len(w_inter) > 20 and len(w_bb) > 5
    assert sorted(map(tuple, inter.tolist())) == sorted(w_inter)
    assert sorted(map(tuple, pipi.tolist())) == sorted(w_pipi)
    assert sorted(map(tuple, bb.tolist())) == sorted(w_bb)
    nb = db.neighbor_list(torch.from_numpy(xyz), 5.0)
    d = np.sqrt(((xyz[:, None].astype(np.float64) - xyz[None]) ** 2).sum(-1))
    assert nb.shape[0] == int((np.triu(d <= 5.0, 1)).sum())


def test_builder_refuses_to_run_without_the_gpu_library():
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(Exception):
        db.xyz_to_ic(torch.zeros(1, 4, 3), np.zeros((1, 4), dtype=np.int32))


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", list(cases.DECODER_CASES))
def test_xyz_to_ic_kernel_against_oracle_and_back(name):
    from codlad_amd.utils.utils_ic import ic_to_xyz
    top, full, og, info, _g5 = golden_frames(name)
    B, L = og.shape[0], og.shape[1] - 2
    quads = db.ic_quads(top)
    want = ic_build.xyz_to_ic(full, quads)
    got = db.xyz_to_ic(torch.from_numpy(full).cuda(), quads).cpu().double().numpy()
    assert got.shape == want.shape
    assert np.abs(got[..., 0] - want[..., 0]).max() < 2e-6
    dang = np.abs(got[..., 1:] - want[..., 1:])
    dang = np.minimum(dang, 2 * np.pi - dang)                      # 0 and 2 pi are the same angle
    assert dang[..., 0].max() < 2e-6, dang[..., 0].max()
    # a dihedral is as well determined as its two bond angles are far from 0 / pi (these are random-weight decoders'
    # geometries: anything occurs): fp32 tolerance scaled by the smaller sine
    q = np.where((quads >= 0).all(-1)[:, None], quads, 0)
    x = full.astype(np.float64)
    a1, a2, a3, a4 = (x[:, q[:, k]] for k in range(4))
    with np.errstate(invalid="ignore", divide="ignore"):
        s1, s2 = np.sin(ic_build.angle_between(a1 - a2, a3 - a2)), np.sin(ic_build.angle_between(a2 - a3, a4 - a3))
        tol = 2e-6 + 2e-6 / np.minimum(s1, s2)
    ok = (quads >= 0).all(-1)[None].repeat(B, 0)
    assert (dang[..., 1][ok] < tol[ok]).all(), float((dang[..., 1][ok] / tol[ok]).max())
    assert (got[(quads < 0).any(-1)[None].repeat(B, 0)] == 0).all()
    back = ic_to_xyz(og.cuda(), torch.from_numpy(got).float().cuda().reshape(B, L, 13, 3), info).cpu()
    rmsd = float(((back - torch.from_numpy(full[:, 1:-1])) ** 2).sum(-1).mean().sqrt())
    assert rmsd < 1e-4, rmsd


@pytest.mark.gpu
def test_build_dataset_from_a_pdb_file(tmp_path):
    from codlad_amd.utils.dataset_module import CG_collate, load_dataset
    top, full, og, info, _g5 = golden_frames("N6_L46_B3")
    write_full_pdb(str(tmp_path / "ens.pdb"), top, full)
    loader, info_dict, n_atoms, n_cgs, atomic_nums, inner = load_dataset(str(tmp_path / "ens"),
                                                                          {"atom_cutoff": 9.0, "cg_cutoff": 21.0, "edgeorder": 2})
    assert n_cgs == 48 and n_atoms == full.shape[1] - 2 == len(atomic_nums) == inner.n_atoms
    for a, b in zip(info_dict[0], info):
        assert torch.equal(a, b)
    ds = loader.dataset
    assert len(ds) == 3
    _, frames = db.read_pdb(str(tmp_path / "ens.pdb"))              # the coordinates as the file holds them (%8.3f)
    want_ic = ic_build.xyz_to_ic(frames, db.ic_quads(top)).reshape(3, 46, 13, 3)
    for f in range(3):
        item = ds[f]
        assert item["nxyz"].shape == (n_atoms, 4) and torch.equal(item["nxyz"][:, 0], torch.tensor(atomic_nums).float())
        assert np.abs(item["nxyz"][:, 1:].numpy() - frames[f, 1:-1]).max() == 0
        assert item["OG_CG_nxyz"].shape == (48, 4) and item["CG_nxyz"].shape == (46, 4)
        assert np.abs(item["OG_CG_nxyz"][:, 1:].numpy() - og[f, :, 1:].numpy()).max() < 6e-4
        assert torch.equal(item["OG_CG_nxyz"][:, 0], og[f, :, 0])
        assert int(item["num_atoms"]) == n_atoms and int(item["num_CGs"]) == 46
        assert item["CG_mapping"].tolist() == inner.residue_of_atom.tolist()
        d = np.abs(item["ic"].double().numpy() - want_ic[f])
        d[..., 1:] = np.minimum(d[..., 1:], 2 * np.pi - d[..., 1:])
        assert d.max() < 5e-5
        assert item["mask"].shape == (46 * 13,) and item["mask_xyz_list"].numel() == 0
        x = item["nxyz"][:, 1:]
        assert torch.equal(item["nbr_list"], synth_pairs(x, 9.0)) and torch.equal(item["CG_nbr_list"], synth_pairs(item["CG_nxyz"][:, 1:], 21.0))
        assert item["bond_edge_list"].shape[1] == 2 and item["bb_NO_list"].shape[1] == 2 and item["pi_pi_list"].shape[1] == 4
    batch = CG_collate([ds[i] for i in range(3)])
    assert batch["nxyz"].shape[0] == 3 * n_atoms and batch["ic"].shape == (3 * 46, 13, 3)
    assert int(batch["nbr_list"].max()) < 3 * n_atoms and int(batch["CG_nbr_list"].max()) < 3 * 46


def synth_pairs(x, cutoff):
    return synth.cg_nbr_list(x, cutoff)


@pytest.mark.gpu
def test_cli_recon_from_a_pdb_file(tmp_path):
    """test.py --pdb_files: atoms of a PDB ensemble -> data set -> e3nn encoder -> VQ -> decoder -> atoms, with the
    evaluation block (every key it needs comes from the builder)."""
    top, full, _og, _info, _g5 = golden_frames("N6_L46_B3")
    write_full_pdb(str(tmp_path / "ens.pdb"), top, full)
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "test.py"), "--synthetic_weights", "--experiment", "recon",
                          "--vae_type", "N6", "--num_ensemble", "2", "--pdb_files", str(tmp_path / "ens.pdb"), "--save_pdb"],
                         cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "test_all_recon" in out.stdout and "test_all_valid_ratio" in out.stdout
    files = [os.path.join(dp, f) for dp, _d, fs in os.walk(str(tmp_path / "logs")) for f in fs]
    npy = [f for f in files if f.endswith(".npy")]
    assert npy, files
    xyz = np.load(npy[0])
    assert xyz.shape[-2:] == (full.shape[1] - 2, 3) and np.isfinite(xyz).all()
    assert any(f.endswith(".pdb") for f in files) and any(f.endswith(".xtc") for f in files)
