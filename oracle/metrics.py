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
