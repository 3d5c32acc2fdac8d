"""Deterministic synthetic inputs for the sampling hot path.

No PED/PDB/Atlas structures and no diffusion / VQ-VAE checkpoints ship with the
reference (SURVEY.md header), so parity tests, goldens and bench.py all run on
inputs produced here.  Everything is drawn from numpy's PCG64 generator, whose
stream is stable across platforms and numpy versions, so the container that
generates the golden vectors and the GPU box regenerate identical inputs from
the seeds alone.

Nothing here is product code: it only fabricates tensors with the shapes, key
names and value ranges the reference uses
(checkpoint keys: SURVEY.md §8b; batch dict: reference utils/dataset_module.py:259-295,
utils/protein_module.py:782-793).
"""
from collections import OrderedDict

import numpy as np
import torch

from .utils.ic_tables import core_atoms, atom_order_list

H = 128

# residue index -> three letter code (reference utils/protein_module.py:95-116)
IDX2THR = ['ASN', 'HIS', 'ALA', 'GLY', 'ARG', 'MET', 'SER', 'ILE', 'GLU', 'LEU', 'TYR',
           'ASP', 'VAL', 'TRP', 'GLN', 'LYS', 'PRO', 'PHE', 'CYS', 'THR', 'TPO', 'SEP']

# heavy-atom order a PDB file lists per residue (what mdtraj's topology would give)
PDB_ATOM_ORDER = {
    'ALA': ['N', 'CA', 'C', 'O', 'CB'],
    'ARG': ['N', 'CA', 'C', 'O', 'CB', 'CG', 'CD', 'NE', 'CZ', 'NH1', 'NH2'],
    'ASP': ['N', 'CA', 'C', 'O', 'CB', 'CG', 'OD1', 'OD2'],
    'ASN': ['N', 'CA', 'C', 'O', 'CB', 'CG', 'OD1', 'ND2'],
    'CYS': ['N', 'CA', 'C', 'O', 'CB', 'SG'],
    'GLU': ['N', 'CA', 'C', 'O', 'CB', 'CG', 'CD', 'OE1', 'OE2'],
    'GLN': ['N', 'CA', 'C', 'O', 'CB', 'CG', 'CD', 'OE1', 'NE2'],
    'GLY': ['N', 'CA', 'C', 'O'],
    'HIS': ['N', 'CA', 'C', 'O', 'CB', 'CG', 'ND1', 'CD2', 'CE1', 'NE2'],
    'ILE': ['N', 'CA', 'C', 'O', 'CB', 'CG1', 'CG2', 'CD1'],
    'LEU': ['N', 'CA', 'C', 'O', 'CB', 'CG', 'CD1', 'CD2'],
    'LYS': ['N', 'CA', 'C', 'O', 'CB', 'CG', 'CD', 'CE', 'NZ'],
    'MET': ['N', 'CA', 'C', 'O', 'CB', 'CG', 'SD', 'CE'],
    'PHE': ['N', 'CA', 'C', 'O', 'CB', 'CG', 'CD1', 'CD2', 'CE1', 'CE2', 'CZ'],
    'PRO': ['N', 'CA', 'C', 'O', 'CB', 'CG', 'CD'],
    'SER': ['N', 'CA', 'C', 'O', 'CB', 'OG'],
    'THR': ['N', 'CA', 'C', 'O', 'CB', 'OG1', 'CG2'],
    'TRP': ['N', 'CA', 'C', 'O', 'CB', 'CG', 'CD1', 'CD2', 'NE1', 'CE2', 'CE3', 'CZ2', 'CZ3', 'CH2'],
    'TYR': ['N', 'CA', 'C', 'O', 'CB', 'CG', 'CD1', 'CD2', 'CE1', 'CE2', 'CZ', 'OH'],
    'VAL': ['N', 'CA', 'C', 'O', 'CB', 'CG1', 'CG2'],
    'TPO': ['N', 'CA', 'C', 'O', 'CB', 'OG1', 'CG2', 'P', 'OE1', 'OE2', 'OE3'],
    'SEP': ['N', 'CA', 'C', 'O', 'CB', 'OG', 'P', 'OE1', 'OE2', 'OE3'],
}


def _rng(seed):
    return np.random.Generator(np.random.PCG64(int(seed)))


def _t(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


# ----------------------------------------------------------------------------
# weights
# ----------------------------------------------------------------------------
def _linear(rng, sd, name, out_f, in_f, bias=True, gain=1.0, bias_std=0.1):
    sd[f"{name}.weight"] = _t(rng.standard_normal((out_f, in_f)) * (gain / np.sqrt(in_f)))
    if bias:
        sd[f"{name}.bias"] = _t(rng.standard_normal((out_f,)) * bias_std)


def denoiser_state_dict(seed=1234, input_size=3, self_condition=False, flow=False):
    """Non-degenerate weights with the key layout of the reference's
    `ProteinMPNN_diffusion_new` (108 tensors; reference models/latent_model.py:119-148).

    The reference constructor zero-initialises every adaLN head
    (models/latent_model.py:155-165), which makes all gates 0 and the network
    output constant; trained checkpoints are absent, so the adaLN heads get
    ordinary random weights here.
    """
    rng = _rng(seed)
    sd = OrderedDict()
    _linear(rng, sd, "t_embedder.mlp.0", H, 256)
    _linear(rng, sd, "t_embedder.mlp.2", H, H)
    # self_condition: x_in sees cat(x_self_cond, x) (reference latent_model.py:112-116)
    _linear(rng, sd, "x_in", H, 2 * input_size if self_condition else input_size, gain=1.0)
    _linear(rng, sd, "features.embeddings.linear", 16, 66)
    _linear(rng, sd, "features.edge_embedding", H, 167, bias=False, gain=2.0)
    sd["features.norm_edges.weight"] = _t(1.0 + 0.1 * rng.standard_normal(H))
    sd["features.norm_edges.bias"] = _t(0.1 * rng.standard_normal(H))
    _linear(rng, sd, "W_e", H, H)
    sd["W_s.weight"] = _t(rng.standard_normal((30, H)))
    for l in range(3):
        p = f"encoder_layers.{l}"
        _linear(rng, sd, f"{p}.W1", H, 3 * H, gain=1.5)
        _linear(rng, sd, f"{p}.W2", H, H, gain=1.5)
        _linear(rng, sd, f"{p}.W3", H, H, gain=1.5)
        _linear(rng, sd, f"{p}.W11", H, 3 * H, gain=1.5)
        _linear(rng, sd, f"{p}.W12", H, H, gain=1.5)
        _linear(rng, sd, f"{p}.W13", H, H, gain=1.5)
        _linear(rng, sd, f"{p}.dense.W_in", 4 * H, H, gain=1.5)
        _linear(rng, sd, f"{p}.dense.W_out", H, 4 * H, gain=1.5)
        _linear(rng, sd, f"{p}.adaLN_modulation.1", 9 * H, H, gain=1.0, bias_std=0.5)
    for l in range(3):
        p = f"decoder_layers.{l}"
        _linear(rng, sd, f"{p}.W1", H, 4 * H, gain=1.5)
        _linear(rng, sd, f"{p}.W2", H, H, gain=1.5)
        _linear(rng, sd, f"{p}.W3", H, H, gain=1.5)
        _linear(rng, sd, f"{p}.dense.W_in", 4 * H, H, gain=1.5)
        _linear(rng, sd, f"{p}.dense.W_out", H, 4 * H, gain=1.5)
        _linear(rng, sd, f"{p}.adaLN_modulation.1", 6 * H, H, gain=1.0, bias_std=0.5)
    # flow-matching models (--model fm / otcfm / ...) predict the velocity only (reference latent_model.py:142-143)
    _linear(rng, sd, "W_out.linear", input_size if flow else 2 * input_size, H, gain=1.0)
    _linear(rng, sd, "W_out.adaLN_modulation.1", 2 * H, H, gain=1.0, bias_std=0.5)
    return sd


def _reference_module_plan(input_size=3):
    """(name, kind, fan_in, fan_out, bias) of every parameterised submodule of the reference's
    ProteinMPNN_diffusion_new in the order its constructor creates them (models/latent_model.py:119-148,
    protein_mpnn_utils.py:208-235, 274-295, 321-344, 347-366; tools/gen_golden.py asserts the order)."""
    plan = [("t_embedder.mlp.0", "linear", 256, H, True), ("t_embedder.mlp.2", "linear", H, H, True),
            ("x_in", "linear", input_size, H, True),
            ("features.embeddings.linear", "linear", 66, 16, True),
            ("features.edge_embedding", "linear", 167, H, False), ("features.norm_edges", "layernorm", H, H, True),
            ("W_e", "linear", H, H, True), ("W_s", "embedding", 30, H, False)]
    for l in range(3):
        p = f"encoder_layers.{l}"
        plan += [(f"{p}.W1", "linear", 3 * H, H, True), (f"{p}.W2", "linear", H, H, True), (f"{p}.W3", "linear", H, H, True),
                 (f"{p}.W11", "linear", 3 * H, H, True), (f"{p}.W12", "linear", H, H, True), (f"{p}.W13", "linear", H, H, True),
                 (f"{p}.dense.W_in", "linear", H, 4 * H, True), (f"{p}.dense.W_out", "linear", 4 * H, H, True),
                 (f"{p}.adaLN_modulation.1", "linear", H, 9 * H, True)]
    for l in range(3):
        p = f"decoder_layers.{l}"
        plan += [(f"{p}.W1", "linear", 4 * H, H, True), (f"{p}.W2", "linear", H, H, True), (f"{p}.W3", "linear", H, H, True),
                 (f"{p}.dense.W_in", "linear", H, 4 * H, True), (f"{p}.dense.W_out", "linear", 4 * H, H, True),
                 (f"{p}.adaLN_modulation.1", "linear", H, 6 * H, True)]
    plan += [("W_out.linear", "linear", H, 2 * input_size, True), ("W_out.adaLN_modulation.1", "linear", H, 2 * H, True)]
    return plan


def reference_init_state_dict(torch_seed=7, adaln_seed=99):
    """The weights the REFERENCE CONSTRUCTOR itself produces under torch.manual_seed(torch_seed): PyTorch's
    default Linear / Embedding init in construction order, then xavier_uniform_ on every parameter with more than
    one dimension (models/latent_model.py:152-154) - replayed here on the same torch generator stream so that the
    GPU box regenerates them without the reference (tools/gen_golden.py asserts bit-equality with the real
    constructor).  The constructor zero-initialises the adaLN heads (:155-165), which would switch every gate off;
    they get seeded non-zero values (adaln_seed) instead, as a trained model has."""
    plan = _reference_module_plan()
    sd = OrderedDict()
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(torch_seed)
        mods = []
        for name, kind, fi, fo, bias in plan:
            m = {"linear": lambda: torch.nn.Linear(fi, fo, bias=bias), "embedding": lambda: torch.nn.Embedding(fi, fo),
                 "layernorm": lambda: torch.nn.LayerNorm(fo)}[kind]()
            mods.append((name, m))
        for name, m in mods:
            for pn, p in m.named_parameters():
                if p.dim() > 1:
                    torch.nn.init.xavier_uniform_(p)
        for name, m in mods:
            for pn, p in m.named_parameters():
                sd[f"{name}.{pn}"] = p.detach().clone()
    rng = _rng(adaln_seed)
    for k in list(sd):
        if "adaLN_modulation" in k:
            shape = tuple(sd[k].shape)
            sd[k] = _t(rng.standard_normal(shape) * (1.0 / np.sqrt(H) if len(shape) == 2 else 0.5))
    return sd


from .utils.dataset_module import _BUILTIN_STATS as NORM_STATS  # noqa: E402  (reference datasets/miu_and_sigma)


def norm_stats(dataname="PED", vae_type="N6"):
    m, s = NORM_STATS[(dataname, vae_type)]
    return torch.tensor(m, dtype=torch.float32), torch.tensor(s, dtype=torch.float32)


def decoder_state_dict(seed=4321, angle=False, prefix="equivaraintconv."):
    """Weights for IC_Decoder (N6) / IC_Decoder_angle (K3, K4) with the reference's
    key layout (reference models/vae_model.py:318-373, 414-465)."""
    rng = _rng(seed)
    sd = OrderedDict()
    F = 40
    sd["res_embed.weight"] = _t(rng.standard_normal((25, 4)))
    for i in range(4):
        _linear(rng, sd, f"message_blocks.{i}.inv_dense.0", F, F)
        _linear(rng, sd, f"message_blocks.{i}.inv_dense.1", F, F)
        _linear(rng, sd, f"message_blocks.{i}.dist_embed.block.1", F, 15, gain=0.3)
    for i in range(4):
        _linear(rng, sd, f"dense_blocks.{i}.1", F, F)
        _linear(rng, sd, f"dense_blocks.{i}.3", F, F, gain=0.5)
    sd["backbone_dist.weight"] = _t(rng.uniform(1.2, 1.6, (25, 3)))
    sd["sidechain_dist.weight"] = _t(rng.uniform(1.2, 1.6, (25, 10)))
    _linear(rng, sd, "backbone_angle.1", 3, F)
    _linear(rng, sd, "backbone_angle.3", 3, 3, bias_std=1.0)
    if angle:
        _linear(rng, sd, "sidechain_angle.1", 10, F)
        _linear(rng, sd, "sidechain_angle.3", 10, 10, bias_std=1.0)
    else:
        sd["sidechain_angle.weight"] = _t(rng.uniform(1.5, 2.3, (25, 10)))
    _linear(rng, sd, "backbone_torsion.1", 3, F + 3)
    _linear(rng, sd, "backbone_torsion.3", 3, 3, bias_std=1.0)
    Ft = F + 10 if angle else F
    for i in range(4):
        _linear(rng, sd, f"sidechain_torsion_blocks.{i}.1", Ft, Ft)
        _linear(rng, sd, f"sidechain_torsion_blocks.{i}.3", Ft, Ft, gain=0.5)
    _linear(rng, sd, "final_torsion.1", 10, Ft)
    _linear(rng, sd, "final_torsion.3", 10, 10, bias_std=1.0)
    return OrderedDict((prefix + k, v) for k, v in sd.items())


def vqvae_state_dict(vae_type="N6", dataname="PED", seed=4321, codebook_size=4096,
                     quantizer_layout="lucidrains", c2_like_map_out=False):
    """Decoder-side VQ-VAE weights: equivaraintconv.*, map_in/map_out (3<->36) and the
    4096x3 codebook (reference utils/model_module.py:39-75, models/vae_model.py:704-706).
    The encoder half (e3nn) is out of scope and gets no tensors."""
    angle = vae_type in ("K3", "K4")
    sd = decoder_state_dict(seed, angle=angle)
    rng = _rng(seed + 17)
    _linear(rng, sd, "map_in", 3, 36)
    _linear(rng, sd, "map_out", 36, 3, gain=0.3)
    if c2_like_map_out:
        # decoder inputs in the range the shipped C2 decoder weights were trained on
        # (reference datasets/miu_and_sigma/PED_C2_y_{mean,std}.pt: |mean| <= 0.1, std ~ 3e-3)
        sd["map_out.weight"] = sd["map_out.weight"] * 2e-3
        sd["map_out.bias"] = sd["map_out.bias"] * 0.3
    mean, std = norm_stats(dataname, vae_type)
    code = rng.standard_normal((codebook_size, 3)).astype(np.float32) * std.numpy() + mean.numpy()
    if quantizer_layout == "lucidrains":
        # vector_quantize_pytorch==1.21.7 EuclideanCodebook buffers (layout unverified offline,
        # SURVEY.md §8c)
        sd["quantize._codebook.initted"] = torch.tensor([True])
        sd["quantize._codebook.cluster_size"] = torch.ones(1, codebook_size)
        sd["quantize._codebook.embed_avg"] = _t(code[None])
        sd["quantize._codebook.embed"] = _t(code[None])
    else:
        sd["quantize.embeddings"] = _t(code)  # in-repo VectorQuantizerEMA (utils/vq_module.py:52)
    return sd


# ----------------------------------------------------------------------------
# geometry
# ----------------------------------------------------------------------------
def ca_trace(n_res, seed, step=3.8, compact=0.02):
    """Seeded CA random walk with a fixed 3.8 A step, mild persistence and a weak pull
    to the centroid so that ~40-60 residues fall inside the 21 A CG cutoff."""
    rng = _rng(seed)
    xyz = np.zeros((n_res, 3), dtype=np.float64)
    d = rng.standard_normal(3)
    d /= np.linalg.norm(d)
    for i in range(1, n_res):
        for _ in range(64):
            r = rng.standard_normal(3)
            pull = (xyz[:i].mean(0) - xyz[i - 1]) * compact
            nd = 0.35 * d + r + pull
            nd /= np.linalg.norm(nd)
            cand = xyz[i - 1] + step * nd
            if i < 3 or np.min(np.linalg.norm(xyz[:i - 1] - cand, axis=1)) > 4.2:
                break
        d = nd
        xyz[i] = cand
    return xyz.astype(np.float32)


def perturb_frames(xyz, n_frames, seed, sigma=0.15):
    """n_frames conformers of one protein: small per-frame CA noise, same topology."""
    rng = _rng(seed)
    out = np.repeat(xyz[None], n_frames, 0).astype(np.float32)
    out[1:] += rng.standard_normal(out[1:].shape).astype(np.float32) * sigma
    return out


def sequence(n_res, seed, force=("GLY", "PRO", "TRP", "HIS", "ARG"), phospho=False):
    rng = _rng(seed)
    hi = 22 if phospho else 20
    z = rng.integers(0, hi, size=n_res)
    for k, name in enumerate(force):
        if 1 + k < n_res - 1:
            z[1 + k] = IDX2THR.index(name)
    return z.astype(np.int64)


def make_info(res_idx_full):
    """(permute, atom_idx, atom_orders) for the interior residues, as built by the reference's traj_to_info
    (utils/protein_module.py:434-494): the product builder fed with the atom order a PDB file lists."""
    from .utils.protein_module import info_from_residues
    names = [IDX2THR[int(z)] for z in res_idx_full]
    return info_from_residues(names, [PDB_ATOM_ORDER[nm] for nm in names])[0]


def cg_nbr_list(xyz, cutoff=21.0):
    """Undirected (j > i) CG pairs within the cutoff, in torch.nonzero order
    (reference utils/protein_module.py:567-584)."""
    x = torch.as_tensor(xyz, dtype=torch.float32)
    n = x.shape[0]
    dist = (x.expand(n, n, 3) - x.expand(n, n, 3).transpose(0, 1)).pow(2).sum(dim=2).sqrt()
    m = dist <= cutoff
    m[torch.arange(n), torch.arange(n)] = False
    nb = torch.nonzero(m)
    return nb[nb[:, 1] > nb[:, 0]]


def make_protein(n_cg, seed, n_frames=1, phospho=False):
    """One synthetic protein: n_cg interior residues (+2 flanking), n_frames conformers."""
    full = ca_trace(n_cg + 2, 1000 + seed)
    z_full = sequence(n_cg + 2, 2000 + seed, phospho=phospho)
    frames = perturb_frames(full, n_frames, 3000 + seed)
    return {"xyz_full": frames, "z_full": z_full, "info": make_info(z_full), "n_cg": n_cg}


def make_batch(protein, frame_ids=None):
    """Batch dict with the keys the hot path reads (CG_collate schema)."""
    frames = protein["xyz_full"]
    if frame_ids is None:
        frame_ids = range(frames.shape[0])
    L = protein["n_cg"]
    z_full = torch.from_numpy(protein["z_full"]).float()
    cg, og, nbr, num = [], [], [], []
    for b, f in enumerate(frame_ids):
        xyz = torch.from_numpy(frames[f])
        og.append(torch.cat([z_full[:, None], xyz], 1))
        cg.append(torch.cat([z_full[1:-1, None], xyz[1:-1]], 1))
        nbr.append(cg_nbr_list(xyz[1:-1]) + b * L)
        num.append(L)
    B = len(num)
    return {
        "CG_nxyz": torch.cat(cg, 0),
        "OG_CG_nxyz": torch.cat(og, 0),
        "CG_nbr_list": torch.cat(nbr, 0),
        "num_CGs": torch.tensor(num, dtype=torch.int64),
        "prot_idx": torch.zeros(B),
    }


# ----------------------------------------------------------------------------
# e3nn encoder / prior
# ----------------------------------------------------------------------------
TP_WEIGHT_NUMEL = (192, 288, 384)     # FullyConnectedTensorProduct weight counts of the three conv depths


def _conv_stack(rng, sd, name, n_layers=3):
    for l in range(n_layers):
        _linear(rng, sd, f"{name}.{l}.fc.0", 36, 36)
        _linear(rng, sd, f"{name}.{l}.fc.3", TP_WEIGHT_NUMEL[l], 36)


def _edge_embedding(rng, sd, name, n_in):
    _linear(rng, sd, f"{name}.0", 12, n_in)
    _linear(rng, sd, f"{name}.3", 12, 12)


def prior_state_dict(seed=777):
    """e3nnPrior tensors with the reference's key layout (models/vae_model.py:204-243; the shipped C2 checkpoint's
    `prior_net.*` without e3nn's own buffers)."""
    rng = _rng(seed)
    sd = OrderedDict()
    sd["cg_node_embedding.weight"] = _t(rng.standard_normal((30, 12)))
    _edge_embedding(rng, sd, "cg_edge_embedding", 14)
    _conv_stack(rng, sd, "cg_conv_layers")
    for head in ("mu", "sigma"):
        _linear(rng, sd, f"{head}.0", 36, 48)
        _linear(rng, sd, f"{head}.2", 36, 36)
    return sd


def encoder_state_dict(seed=778):
    """e3nnEncoder tensors with the reference's key layout (models/vae_model.py:21-107)."""
    rng = _rng(seed)
    sd = OrderedDict()
    sd["atom_node_embedding.weight"] = _t(rng.standard_normal((30, 12)))
    sd["cg_node_embedding.weight"] = _t(rng.standard_normal((30, 12)))
    _edge_embedding(rng, sd, "atom_edge_embedding", 14)
    _edge_embedding(rng, sd, "cg_edge_embedding", 14)
    _edge_embedding(rng, sd, "cross_edge_embedding", 8)
    for name in ("atom_conv_layers", "cg_conv_layers", "cg_to_atom_conv_layers", "atom_to_cg_conv_layers"):
        _conv_stack(rng, sd, name)
    _linear(rng, sd, "dense.0", 36, 84)
    _linear(rng, sd, "dense.2", 36, 36)
    return sd


_ELEMENT_Z = {"C": 6, "N": 7, "O": 8, "S": 16, "P": 15}


def make_atoms(protein, frame_ids=None, seed=0, atom_cutoff=9.0):
    """All-atom side of a batch, as the reference's CG_collate carries it for the encoder (utils/dataset_module.py:259-295):
    `nxyz` [n_atoms, 4] (atomic number, xyz), `CG_mapping` [n_atoms] (bead of every atom, batch offset added),
    `nbr_list` [E, 2] (atom pairs j > i within atom_cutoff, batch offset added), `num_atoms` [B].  Coordinates are
    synthetic: CA on the bead, the residue's other heavy atoms scattered 1-3 A around it (seeded)."""
    frames = protein["xyz_full"]
    if frame_ids is None:
        frame_ids = range(frames.shape[0])
    L = protein["n_cg"]
    names = [IDX2THR[int(z)] for z in protein["z_full"][1:-1]]
    atoms = [(r, a) for r, nm in enumerate(names) for a in PDB_ATOM_ORDER[nm]]
    zs = torch.tensor([_ELEMENT_Z[a[0]] for _r, a in atoms], dtype=torch.float32)
    res = torch.tensor([r for r, _a in atoms], dtype=torch.int64)
    n = len(atoms)
    off = _t(_rng(5000 + seed).standard_normal((n, 3))) * 1.2
    off[[i for i, (_r, a) in enumerate(atoms) if a == "CA"]] = 0.0
    nxyz, mapping, nbr, num = [], [], [], []
    for b, f in enumerate(frame_ids):
        ca = torch.from_numpy(frames[f])[1:-1].float()
        xyz = ca[res] + off
        nxyz.append(torch.cat([zs[:, None], xyz], 1))
        mapping.append(res + b * L)
        d = (xyz[:, None] - xyz[None]).pow(2).sum(-1).sqrt()
        m = torch.triu(d <= atom_cutoff, diagonal=1)
        nbr.append(torch.nonzero(m) + b * n)
        num.append(n)
    return {"nxyz": torch.cat(nxyz, 0), "CG_mapping": torch.cat(mapping, 0), "nbr_list": torch.cat(nbr, 0),
            "num_atoms": torch.tensor(num, dtype=torch.int64)}


def gaussian(shape, seed):
    return _t(_rng(seed).standard_normal(shape))


# ----------------------------------------------------------------------------
# BASELINE.json configurations as synthetic workloads
# ----------------------------------------------------------------------------
PED_LENGTHS = (46, 87, 92, 129)


def atlas_test_lengths():
    """The `seqlen` column of the reference's Atlas test list (datasets/protein/Atlas/new_atlas_test.csv:
    70 proteins, 39..505 residues, median 155), kept as a data fixture (tools/gen_golden.py g0)."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                        "atlas_test_seqlen.json")
    with open(path) as f:
        return [int(x) for x in json.load(f)["seqlen"]]


def baseline_config(name):
    """name -> dict(lengths, n_frames, n_ensemble, vae_type, dataname, decode_only, scaling, what).
    cfg2..cfg5 are BASELINE.json configs[1..4] with the sizes SURVEY.md §8(d) fixes; PDB lengths do not ship
    with the reference, so cfg3 takes the first 64 Atlas test lengths clipped to 50..400 (as §8d says)."""
    if name in ("cfg2", "cfg5"):
        return dict(lengths=list(PED_LENGTHS), n_frames=10, n_ensemble=10, vae_type="N6", dataname="PED",
                    decode_only=name == "cfg5", scaling="weak",
                    what=("cfg2: PED-shaped test set, 4 proteins L=46/87/92/129 x 10 frames x num_ensemble 10 = 400 "
                          "structures per GPU, 100-step DDPM (mpnn_diffusion) + VQ(4096x3) + IC_Decoder N6 + ic_to_xyz"
                          if name == "cfg2" else
                          "cfg5: --experiment recon on the cfg2 geometry (400 structures per GPU): VQ(4096x3) + IC_Decoder "
                          "N6 + ic_to_xyz only, latents drawn N(mean, std) in place of the e3nn encoder's"))
    if name == "cfg3":
        lengths = [max(50, min(400, L)) for L in atlas_test_lengths()[:64]]
        return dict(lengths=lengths, n_frames=1, n_ensemble=1, vae_type="K3", dataname="PDB", decode_only=False,
                    scaling="strong",
                    what="cfg3: 64 proteins (Atlas test lengths clipped to 50..400), 1 frame each, 100-step DDPM + "
                         "VQ + IC_Decoder_angle K3 + ic_to_xyz, units sharded over the GPUs (LPT)")
    if name in ("cfg4", "cfg4share"):
        return dict(lengths=atlas_test_lengths(), n_frames=4, n_ensemble=32, vae_type="K4", dataname="Atlas",
                    decode_only=False, scaling="strong", share_of=8 if name == "cfg4share" else None,
                    what=("cfg4: Atlas test set, 70 proteins (39..505 residues) x 4 frames x num_ensemble 32 = 8960 "
                          "structures, 100-step DDPM + VQ + IC_Decoder_angle K4 + ic_to_xyz, units sharded over the "
                          "GPUs (LPT), coordinates all-gathered" +
                          ("; THIS RUN: the 1/8 share (1120 structures) LPT deals to rank 0 of 8" if name == "cfg4share"
                           else "")))
    raise KeyError(name)
