"""Next row 8f-2 (GPU): codlad_amd.metrics against the goldens the reference's own helpers produced and
against the CPU oracle, through the C ABI."""
import numpy as np
import pytest
import torch

from codlad_amd import metrics as gm
from oracle import metrics as om
from tests import cases

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def on_gpu(d):
    return {k: v.to(DEV) for k, v in d.items()}


@pytest.mark.parametrize("name", list(cases.METRIC_CASES))
def test_metrics_match_reference_goldens(name):
    gold = np.load(cases.npz_path(f"g8_metrics_{name}"))
    d = on_gpu(cases.metric_inputs(name))
    bond, angle, torsion = gm.recon_result(d["ic_recon"], d["ic"], d["mask"])
    inter, pipi = gm.inter_result(d["interaction_list"], d["pi_pi_list"], d["xyz_recon"])
    got = dict(loss_bond=bond, loss_angle=angle, loss_torsion=torsion,
               loss_xyz=gm.xyz_result(d["xyz_recon"], d["xyz"]),
               loss_graph=gm.ged_result(d["xyz_recon"], d["xyz"], d["edge_list"]),
               loss_nbr=gm.clash_result(d["edge_list"], d["nbr_list"], d["xyz_recon"], d["bb_NO_list"]),
               loss_inter=inter, loss_pi_pi=pipi)
    for k, v in got.items():
        # sums are accumulated in double on the device, in fp32 by ATen: a few 1e-7 apart
        assert float(v) == pytest.approx(float(gold[k]), rel=3e-6, abs=1e-9), k
    fused = gm.all_results(d["ic_recon"], d["ic"], d["mask"], d["xyz_recon"], d["xyz"], d["edge_list"], d["nbr_list"],
                           d["bb_NO_list"], d["interaction_list"], d["pi_pi_list"])
    for k, v in got.items():
        assert float(fused[k]) == float(v), k      # one launch == five launches, bit for bit


def test_clash_list_is_the_reference_set_difference():
    d = cases.metric_inputs("small")
    want = om.clash_list(d["edge_list"], d["nbr_list"])
    got = gm.clash_list(d["edge_list"].to(DEV), d["nbr_list"].to(DEV))
    assert torch.equal(got.cpu(), want)
    # duplicated rows inside one list cancel, (i, j) and (j, i) are different rows
    e = torch.tensor([[1, 2], [3, 4], [3, 4], [5, 6]])
    n = torch.tensor([[2, 1], [5, 6], [7, 8]])
    assert torch.equal(gm.clash_list(e.to(DEV), n.to(DEV)).cpu(), om.clash_list(e, n))


def test_clash_counts_are_exact_and_replay_is_bit_identical():
    d = on_gpu(cases.metric_inputs("big"))
    a = gm.clash_result(d["edge_list"], d["nbr_list"], d["xyz_recon"], d["bb_NO_list"])
    b = gm.clash_result(d["edge_list"], d["nbr_list"], d["xyz_recon"], d["bb_NO_list"])
    assert float(a) == float(b)
    cpu = cases.metric_inputs("big")
    cl = om.clash_list(cpu["edge_list"], cpu["nbr_list"])
    xr = cpu["xyz_recon"]
    dist = lambda p: ((xr[p[:, 0]] - xr[p[:, 1]]).pow(2).sum(-1) + 1e-7).sqrt()  # noqa: E731
    want = (dist(cl) < 1.2).sum().item() / len(cl) + (dist(cpu["bb_NO_list"]) < 1.2).sum().item() / len(cpu["bb_NO_list"])
    assert float(a) == pytest.approx(want, abs=2e-7)       # same violation COUNT: ratios differ by rounding only


def test_cpu_tensors_are_refused():
    d = cases.metric_inputs("small")
    with pytest.raises(RuntimeError):
        gm.xyz_result(d["xyz_recon"].to(DEV), d["xyz"])


@pytest.mark.parametrize("name", list(cases.VALIDITY_CASES))
def test_bond_graph_validity_matches_reference(name):
    """codlad_bond_graph_counts / valid_ratio_and_cut_off_result against the goldens the reference's own function
    produced (test.py:168-188): validity flags and bond-count ratios exactly, and the raw counts against the oracle's
    bond matrices."""
    gold = np.load(cases.npz_path(f"g11_validity_{name}"))
    d = cases.validity_inputs(name)
    hv, av, hg, ag = gm.valid_ratio_and_cut_off_result(d["xyz"].to(DEV), d["xyz_recon"].to(DEV), d["num_atoms"],
                                                       d["atomic_nums"])
    assert hv == gold["heavy_valid"].tolist() and av == gold["all_valid"].tolist()
    assert np.array_equal(np.array(hg, dtype=np.float64), gold["heavy_ged"])
    assert np.array_equal(np.array(ag, dtype=np.float64), gold["all_ged"])
    cnt = gm.bond_graph_counts(d["xyz"].to(DEV), d["xyz_recon"].to(DEV), d["num_atoms"], d["atomic_nums"]).cpu()
    table = torch.tensor(gm.COV_CUTOFF)
    n = int(d["num_atoms"][0])
    for s in range(len(d["num_atoms"])):
        z = d["atomic_nums"][s * n:(s + 1) * n]
        ref = om.bond_graph(d["xyz"][s * n:(s + 1) * n], table[z - 1])
        gen = om.bond_graph(d["xyz_recon"][s * n:(s + 1) * n], table[z - 1])
        hvy = (z != 1)[:, None] & (z != 1)[None, :]
        want = [int(ref.sum()) // 2, int(gen.sum()) // 2, int((ref != gen).sum()) // 2,
                int((ref * hvy).sum()) // 2, int((gen * hvy).sum()) // 2, int(((ref != gen) & hvy).sum()) // 2]
        assert cnt[s].tolist() == want


def test_cli_evaluation_block_reports_the_reference_summary():
    """test.py's Evaluation (the reference loop's evaluation block, test.py:566-668, on the device): one ensemble
    member of a fabricated batch -> the summary keys the reference prints, values equal to the helper calls."""
    import importlib.util
    import os
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("codlad_cli_eval", os.path.join(root, "test.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    d = on_gpu(cases.metric_inputs("small"))
    n_atoms = d["xyz"].shape[0] // 2                                   # two frames of n_atoms atoms
    z = torch.full((2 * n_atoms,), 6.0, device=DEV)
    batch = {"nxyz": torch.cat([z[:, None], d["xyz"]], 1), "num_atoms": torch.tensor([n_atoms, n_atoms]),
             "bond_edge_list": d["edge_list"], "nbr_list": d["nbr_list"], "bb_NO_list": d["bb_NO_list"],
             "interaction_list": d["interaction_list"], "pi_pi_list": d["pi_pi_list"], "ic": d["ic"], "mask": d["mask"],
             "mask_xyz_list": torch.tensor([3, 77], device=DEV)}
    ev = cli.Evaluation()
    for member in range(2):
        ev.add(batch, d["ic_recon"], (d["xyz_recon"] + 0.01 * member).reshape(2, n_atoms, 3), n_atoms)
    args = types.SimpleNamespace(data_type="PED", num_ensemble=2, experiment="latent")
    stats = ev.report("fabricated", args)
    for k in ("test_all_recon", "test_xyz", "test_graph", "test_nbr", "test_inter", "test_pi_pi", "test_all_valid_ratio",
              "test_all_ged", "diversity"):
        assert k in stats and np.isfinite(stats[k]), k
    xyz, xr = d["xyz"].clone(), d["xyz_recon"].clone()
    xyz[batch["mask_xyz_list"]] *= 0
    xr[batch["mask_xyz_list"]] *= 0
    want0 = float(gm.xyz_result(xr, xyz))
    assert ev.rows[0]["loss_xyz"] == want0
    assert 0.0 <= stats["test_all_valid_ratio"] <= 1.0


def test_superposed_rmsd_is_invariant_to_rigid_motion():
    """Kabsch RMSD (stands in for mdtraj's md.rmsd in the diversity score; unpinned): zero for a rotated + shifted
    copy, equal to the plain RMSD when the optimal superposition is the identity."""
    g = torch.Generator().manual_seed(3)
    a = torch.randn(200, 3, generator=g, dtype=torch.float64) * 5
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))
    if torch.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    b = a @ q.t() + torch.tensor([1.0, -2.0, 3.0], dtype=torch.float64)
    assert gm.superposed_rmsd(a, b) < 1e-6
    noise = torch.randn(200, 3, generator=g, dtype=torch.float64) * 0.01
    r = gm.superposed_rmsd(a, a + noise)
    assert r <= float(noise.pow(2).sum(-1).mean().sqrt()) + 1e-12
    gen = [a[None] + 0.1 * torch.randn(1, 200, 3, generator=g, dtype=torch.float64) for _ in range(4)]
    div = gm.compute_div(gen, a[None] + 0.3)
    assert 0.0 < div < 1.0
