"""The N > 1 flow on real hardware (runs FIRST among the GPU tests, so that the rank processes are started before
this process has initialised the GPU): two fresh processes on cuda:0 over gloo - real DenoiserWeights /
DecoderWeights broadcast from rank 0 onto a rank that started from different weights, `rebind()`, LPT sharding of a
12-unit job, sampling + decoding per rank, `gather_coordinates` on CUDA tensors - and the gathered result must equal
the single-process result bit for bit (what the 8-GPU runs of bench.py / test.py rely on; reference loop replaced:
test.py:413-481).  Also rehearses bench.py's own multi-rank code path (--config cfg3, 2 ranks)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    env = dict(os.environ)
    env.update(CODLAD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    return env


@pytest.mark.timeout(600)
def test_two_rank_job_equals_single_process(tmp_path):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multirank_worker.py"), str(r), "2",
                               str(port), str(tmp_path)], env=_env(), cwd=ROOT,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=500)[0] for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"
    got = np.load(os.path.join(tmp_path, "gathered.npz"))
    assert got["shard_sizes"].tolist() == [6, 6]
    # single process, rank 0's weights, all 12 units in one job
    from tests import multirank_worker as mw
    from tests import pipeline
    cfg = pipeline.Config("mr", mw.LENGTHS, mw.N_FRAMES, mw.N_ENSEMBLE, "N6", "PED", T=mw.T)
    whole = cfg.run_units(list(range(len(cfg.units))))
    assert len(cfg.units) == 12
    for u in range(12):
        assert np.array_equal(got[f"xyz_{u}"], whole[u][2].cpu().numpy()), f"unit {u} differs between 2 ranks and 1"


@pytest.mark.timeout(900)
def test_bench_sharded_config_two_ranks():
    """bench.py --config cfg3 --gpus 2 as the driver launches it (torch.distributed.run), both ranks on cuda:0 over
    gloo: one JSON line, strong scaling, all 64 structures accounted for."""
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
           "--warmup", "1", "--config", "cfg3"]
    res = subprocess.run(cmd, env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=850)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["config"]["structures_per_step"] == 64
    assert r["value"] > 0 and 0 < r["roofline"]["frac"] <= 1.0
