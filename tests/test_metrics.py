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
