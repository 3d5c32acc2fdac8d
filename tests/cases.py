"""Seeded case definitions shared by tools/gen_golden.py (runs the reference in the
build container) and the parity tests (regenerate the same inputs anywhere).

Only outputs (and explicit noise where the reference would draw it) are stored in
tests/golden/*.npz; inputs come back from these seeds through codlad_amd.synth.
"""
import numpy as np
import torch

from codlad_amd import synth

WEIGHT_SEED = 1234
VAE_SEED = 4321

# name -> (n_cg, n_frames, protein seed)
DENOISER_CASES = {
    "L20_B2": (20, 2, 11),     # K = L < 64, small enough to keep every intermediate
    "L46_B2": (46, 2, 12),     # PED00151-sized, K = 46
    "L87_B2": (87, 2, 13),     # PED00055-sized, K = 64
    "L129_B3": (129, 3, 14),   # PED00218-sized
}

# (name, [n_cg per sample]) : padded mixed-length batch, oracle-only
PADDED_CASE = ("padded_30_46_70", [30, 46, 70], 21)

# name -> (n_cg, n_frames, protein seed, T)
LOOP_CASES = {
    "L46_B2_T10": (46, 2, 12, 10),
    "L87_B2_T10": (87, 2, 13, 10),
    "L87_B1_T100": (87, 1, 13, 100),
}

DECODER_CASES = {
    # name -> (n_cg, n_frames, seed, vae_type)
    "N6_L46_B3": (46, 3, 31, "N6"),
    "N6_L87_B2": (87, 2, 32, "N6"),
    "K3_L60_B2": (60, 2, 33, "K3"),
    "K4_L129_B1": (129, 1, 34, "K4"),
}

E2E_CASES = {
    # name -> (n_cg, n_frames, seed, T, vae_type, dataname)
    "PED_N6_L46_B2_T10": (46, 2, 41, 10, "N6", "PED"),
    "PED_N6_L87_B2_T100": (87, 2, 42, 100, "N6", "PED"),
}


def denoiser_inputs(n_cg, n_frames, seed, t_value=None, phospho=False):
    prot = synth.make_protein(n_cg, seed, n_frames=n_frames, phospho=phospho)
    batch = synth.make_batch(prot)
    x = synth.gaussian((n_frames, n_cg, 3), 5000 + seed)
    if t_value is None:
        t_value = 500 + 7 * seed
    t = torch.full((n_frames,), int(t_value), dtype=torch.int64)
    batch["randn"] = synth.gaussian((n_frames, n_cg), 6000 + seed)
    mask = torch.ones(n_frames, n_cg, dtype=torch.bool)
    return prot, batch, x, t, mask


def padded_inputs(lengths, seed):
    cg, nbr, num = [], [], []
    off = 0
    for k, L in enumerate(lengths):
        prot = synth.make_protein(L, seed + k, n_frames=1)
        b = synth.make_batch(prot)
        cg.append(b["CG_nxyz"])
        nbr.append(b["CG_nbr_list"] + off)
        num.append(L)
        off += L
    Lmax = max(lengths)
    N = len(lengths)
    batch = {"CG_nxyz": torch.cat(cg), "CG_nbr_list": torch.cat(nbr),
             "num_CGs": torch.tensor(num, dtype=torch.int64),
             "randn": synth.gaussian((N, Lmax), 6000 + seed)}
    x = synth.gaussian((N, Lmax, 3), 5000 + seed)
    t = torch.full((N,), 321, dtype=torch.int64)
    mask = torch.arange(Lmax)[None, :] < batch["num_CGs"][:, None]
    return batch, x, t, mask


def loop_noise(T, n, n_cg, seed):
    """x_T and the per-step noise the sampler adds, step order i = T-1 .. 0."""
    z = synth.gaussian((n, n_cg, 3), 7000 + seed)
    eps = synth.gaussian((T, n, n_cg, 3), 8000 + seed)
    return z, eps


def decoder_inputs(n_cg, n_frames, seed, vae_type):
    prot = synth.make_protein(n_cg, seed, n_frames=n_frames, phospho=(vae_type != "N6"))
    batch = synth.make_batch(prot)
    dataname = {"N6": "PED", "K3": "PDB", "K4": "Atlas"}[vae_type]
    mean, std = synth.norm_stats(dataname, vae_type)
    latent = synth.gaussian((n_frames, n_cg, 3), 9000 + seed) * std + mean
    return prot, batch, latent, dataname


def npz_path(name):
    import os
    return os.path.join(os.path.dirname(__file__), "golden", name + ".npz")
