"""Drop-in for the two pieces of the reference's utils/dataset_module.py the hot path touches:
`get_norm_feature` (latent (de-)normalisation, dataset_module.py:230-256) and `CG_collate`'s batch
schema (dataset_module.py:259-295), plus `load_dataset` (dataset_module.py:144-225) for multi-model PDB files, built on
utils/dataset_builder.py and utils/xtc.py instead of mdtraj (the .xtc reader is a restatement of the published format with
no third-party file to check it against: parity unpinned)."""
import os

import numpy as np
import torch

# datasets/miu_and_sigma/{PED_N6,PDB_K3,Atlas_K4}_x_{mean,std}.pt of the reference (3 floats each),
# used when the files are not present under ./datasets/miu_and_sigma
_BUILTIN_STATS = {
    ("PED", "N6"): ([1.068959355354309, -0.8994553089141846, 0.5618639588356018],
                    [5.1957831382751465, 4.400951385498047, 5.270322799682617]),
    ("PDB", "K3"): ([-1.5160585641860962, 0.6747006773948669, -0.5968422293663025],
                    [8.262883186340332, 5.664480686187744, 6.969945907592773]),
    ("Atlas", "K4"): ([-0.29618993401527405, 1.7351123094558716, -0.05292452499270439],
                      [5.226162910461426, 7.113760948181152, 6.114980697631836]),
}


def load_norm_stats(feature_type, dataname="PED", norm_single=False, root="./datasets/miu_and_sigma"):
    if dataname == "IDRome_test_7":
        dataname = {"N6": "PED", "K3": "PDB", "K4": "Atlas"}.get(feature_type, dataname)
    tail = "_single" if norm_single else ""
    fm = os.path.join(root, f"{dataname}_{feature_type}_x_mean{tail}.pt")
    fs = os.path.join(root, f"{dataname}_{feature_type}_x_std{tail}.pt")
    if os.path.exists(fm) and os.path.exists(fs):
        return torch.load(fm), torch.load(fs)
    if (dataname, feature_type) in _BUILTIN_STATS and not norm_single:
        m, s = _BUILTIN_STATS[(dataname, feature_type)]
        return torch.tensor(m), torch.tensor(s)
    raise FileNotFoundError(f"no latent statistics for {dataname}/{feature_type}: {fm}")


def get_norm_feature(feature, feature_type, norm_channel=True, norm_single=False, norm_in=True, dataname="PED"):
    mean, std = load_norm_stats(feature_type, dataname, norm_single)
    mean, std = mean.to(feature.device), std.to(feature.device)
    if norm_in:
        return (feature - mean) / std
    return feature * std + mean


def CG_collate(dicts):
    """Batch the per-frame dicts: offsets added to index lists, tensors concatenated."""
    cum_atoms = np.cumsum([0] + [d['num_atoms'].item() for d in dicts])[:-1] if 'num_atoms' in dicts[0] else None
    cum_cgs = np.cumsum([0] + [d['num_CGs'].item() for d in dicts])[:-1]
    dicts = [dict(d) for d in dicts]
    if cum_atoms is not None:
        for n, d in zip(cum_atoms, dicts):
            for k in ('nbr_list', 'bond_edge_list', 'mask_xyz_list', 'bb_NO_list', 'interaction_list', 'pi_pi_list'):
                if k in d:
                    d[k] = d[k] + int(n)
    for n, d in zip(cum_cgs, dicts):
        for k in ('CG_mapping', 'CG_nbr_list'):
            if k in d:
                d[k] = d[k] + int(n)
    batch = {}
    for key, val in dicts[0].items():
        if hasattr(val, 'shape') and len(val.shape) > 0:
            batch[key] = torch.cat([d[key] for d in dicts], dim=0)
        elif isinstance(val, str):
            batch[key] = [d[key] for d in dicts]
        else:
            batch[key] = torch.stack([d[key] for d in dicts], dim=0)
    return batch


def load_dataset(data_path, params, single=True, device="cuda"):
    """dataset_module.py:144-225: `{data_path}.pdb` (a multi-model ensemble; `single=False`: the Atlas directory layout with
    .xtc replicas) -> (DataLoader of CG_collate
    batches, info_dict, n_atoms, n_cgs, atomic_nums, topology of the interior residues).  `params`: the VAE's
    modelparams (atom_cutoff, cg_cutoff, edgeorder).  The reference's file-specific trimming of thirteen PED entries
    (dataset_module.py:167-179: their first and last residue dropped before anything else) is kept."""
    from torch.utils.data import DataLoader

    from . import dataset_builder as db
    from .protein_module import info_from_residues
    if single:
        top, frames = db.read_pdb(f"{data_path}.pdb")
    else:
        # the Atlas layout (dataset_module.py:150-160): <dir>/<name>/<name>.pdb + three replica trajectories, the
        # reference structure first, every 10 000th frame kept
        from .xtc import read_xtc
        name = os.path.basename(data_path)
        top, ref = db.read_pdb(os.path.join(data_path, f"{name}.pdb"))
        parts = [ref[:1]]
        for r in (1, 2, 3):
            xyz = read_xtc(os.path.join(data_path, f"{name}_prod_R{r}_fit.xtc"))[0]
            if xyz.shape[1] != top.n_atoms:
                raise ValueError(f"{name}_prod_R{r}_fit.xtc has {xyz.shape[1]} atoms, the topology's heavy atoms are {top.n_atoms}: "
                                 "trajectories with hydrogens need the all-atom topology's heavy-atom selection")
            parts.append(xyz)
        frames = np.concatenate(parts, 0)[::10000]
    if os.path.basename(data_path) in _PED_TRIM_ENDS:
        a, b = int(top.first_atom[1]), int(top.first_atom[top.n_residues - 1])
        top, frames = top.subset_residues(1, top.n_residues - 1), frames[:, a:b]
    info, n_cgs = info_from_residues(top.res_names, top.atom_names)
    testset, _mapping = db.build_split_dataset(top, frames, params, prot_idx=0, device=device)
    nfirst, nlast = len(top.atom_names[0]), len(top.atom_names[-1])
    atomic_nums = top.atomic_nums()[nfirst:top.n_atoms - nlast]
    loader = DataLoader(testset, batch_size=min(len(testset), 96 if single else 24), collate_fn=CG_collate, shuffle=False,
                        drop_last=False)
    return loader, {0: info}, top.n_atoms - nfirst - nlast, n_cgs, atomic_nums, top.subset_residues(1, top.n_residues - 1)


_PED_TRIM_ENDS = ("PED00151e000", "PED00151e001", "PED00151e002", "PED00011e001", "PED00143e001", "PED00145e000",
                  "PED00145e001", "PED00148e001", "PED00148e002", "PED00150e000", "PED00150e001", "PED00150e002",
                  "PED00145e002")
