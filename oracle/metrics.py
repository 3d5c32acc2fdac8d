"""TEST INFRASTRUCTURE - CPU restatement of the evaluation helpers that follow the sampling path in the
reference's loop (reference test.py:97-166, called at test.py:589-593).  Pinned by
tests/golden/g8_metrics_*.npz, which tools/gen_golden.py produced by calling the reference's own
functions on the synthetic lists of tests/cases.py.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline may import this package; the product path (codlad_amd/metrics.py) never does.
"""
import torch

EPS = 1e-7  # test.py:27


def _pair_dist(xyz, pairs):
    return ((xyz[pairs[:, 0]] - xyz[pairs[:, 1]]).pow(2).sum(-1) + EPS).sqrt()


def recon_result(ic_recon, ic, mask):
    """test.py:153-166: masked bond MSE and chord-length angle / torsion errors per valid slot."""
    n = mask.sum()
    bond = ((ic_recon[:, :, 0] - ic[:, :, 0]).reshape(-1) * mask).pow(2).sum() / n
    chord = lambda k: ((2 * (1 - torch.cos(ic[:, :, k] - ic_recon[:, :, k])) + EPS).sqrt().reshape(-1) * mask).sum() / n  # noqa: E731
    return bond, chord(1), chord(2)


def xyz_result(xyz_recon, xyz):
    """test.py:148-151."""
    return (xyz_recon - xyz).pow(2).sum(-1).mean()


def ged_result(xyz_recon, xyz, edge_list):
    """test.py:141-146: squared error of the bonded distances."""
    return (_pair_dist(xyz_recon, edge_list) - _pair_dist(xyz, edge_list)).pow(2).mean()


def clash_list(edge_list, nbr_list):
    """Rows of cat(edge_list, nbr_list) that occur exactly once (test.py:121-123)."""
    uniques, counts = torch.cat((edge_list, nbr_list)).unique(dim=0, return_counts=True)
    return uniques[counts == 1]


def clash_result(edge_list, nbr_list, xyz_recon, bb_NO_list):
    """test.py:118-139: share of non-bonded neighbour pairs and of backbone N-O pairs closer than 1.2 A."""
    def share(pairs):
        d = _pair_dist(xyz_recon, pairs)
        return (d < 1.2).sum().float() / d.numel() if d.numel() > 0 else torch.tensor(0.0)
    return share(clash_list(edge_list, nbr_list)) + share(bb_NO_list)


def inter_result(interaction_list, pi_pi_list, xyz_recon):
    """test.py:97-116: hinge losses on interaction distances (4 A) and pi-pi ring-centre distances (6 A),
    the first weighted by list sizes."""
    n_inter, n_pipi = interaction_list.shape[0], pi_pi_list.shape[0]
    total = n_inter + n_pipi
    loss_inter, loss_pipi = torch.tensor(0.0), torch.tensor(0.0)
    if n_inter > 0:
        loss_inter = torch.clamp(_pair_dist(xyz_recon, interaction_list) - 4.0, min=0.0).mean() * (n_inter / total)
    if n_pipi > 0:
        c0 = (xyz_recon[pi_pi_list[:, 0]] + xyz_recon[pi_pi_list[:, 1]]) / 2
        c1 = (xyz_recon[pi_pi_list[:, 2]] + xyz_recon[pi_pi_list[:, 3]]) / 2
        loss_pipi = torch.clamp(((c0 - c1).pow(2).sum(-1) + EPS).sqrt() - 6.0, min=0.0).mean()
        loss_inter = loss_inter + loss_pipi * (n_pipi / total)
    return loss_inter, loss_pipi


def bond_graph(xyz, radius, scale=1.3):
    """utils/protein_module.py:251-288 (compute_bond_cutoff, compute_distance_mat, get_bond_graphs): 0/1 matrix
    of atom pairs closer than (r_i + r_j) * scale, diagonal cleared."""
    dist = (xyz[:, None, :] - xyz[None, :, :]).pow(2).sum(-1).sqrt()
    cutoff = (radius[None, :] + radius[:, None]) * scale
    bonds = dist < cutoff
    bonds[torch.arange(len(xyz)), torch.arange(len(xyz))] = False
    return bonds.to(torch.long)


def valid_ratio_and_cut_off_result(xyz, xyz_recon, num_atoms, atomic_nums, cov_cutoff):
    """test.py:168-188 -> eval_sample_qualities / count_valid_graphs (utils/protein_module.py:290-364) for one
    reconstruction per reference structure; ase.Atoms is only a container of (numbers, positions) there.
    cov_cutoff: radius by atomic number - 1 (COVCUTOFFTABLE, protein_module.py:128-234)."""
    hv, av, hg, ag = [], [], [], []
    table = torch.tensor(cov_cutoff, dtype=torch.float32)
    for x, y, z in zip(torch.split(xyz.float().cpu(), list(num_atoms)), torch.split(xyz_recon.float().cpu(), list(num_atoms)),
                       torch.split(torch.as_tensor(atomic_nums).long(), list(num_atoms))):
        for heavy_only, valid, ged in ((True, hv, hg), (False, av, ag)):
            keep = z != 1 if heavy_only else torch.ones_like(z, dtype=torch.bool)
            r = table[z[keep] - 1]
            ref, gen = bond_graph(x[keep], r), bond_graph(y[keep], r)
            valid.append(1.0 if int((gen != ref).sum()) == 0 else 0.0)
            ged.append([((ref - gen).sum().abs() / ref.sum()).item()])
    return hv, av, hg, ag
