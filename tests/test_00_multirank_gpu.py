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


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode", ["r1_envelope", "r0_envelope", "r1_empty"])
def test_two_rank_job_equals_single_process(tmp_path, mode):
    """Rank 1 starts from weights with OTHER block exponents (or from none, with a decoder blob of the wrong layout):
    after the broadcast the exponents in every rank's kernel struct are rank 0's and the gathered coordinates equal
    the single-process result bit for bit."""
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multirank_worker.py"), str(r), "2",
                               str(port), str(tmp_path), mode], env=_env(), cwd=ROOT,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=800)[0] for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"
    assert "ranks=2 backend=gloo" in outs[0] and "equal on every rank" in outs[0]
    got = np.load(os.path.join(tmp_path, "gathered.npz"))
    assert got["shard_sizes"].tolist() == [6, 6]
    # single process, rank 0's weights, all 12 units in one job
    from tests import multirank_worker as mw
    cfg = mw.job_config(mode, 0)
    assert np.array_equal(got["exponents"], np.array(mw.exponent_table(cfg.den.weights)))
    assert bool(np.abs(got["exponents"]).sum()) == (mode == "r0_envelope")
    whole = cfg.run_units(list(range(len(cfg.units))))
    assert len(cfg.units) == 12
    for u in range(12):
        assert np.array_equal(got[f"xyz_{u}"], whole[u][2].cpu().numpy()), f"unit {u} differs between 2 ranks and 1"


def _bench_two_ranks(extra):
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
           "--warmup", "1"] + extra
    res = subprocess.run(cmd, env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=850)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_bench_sharded_config_two_ranks():
    """bench.py --config cfg3 --gpus 2 as the driver launches it (torch.distributed.run), both ranks on cuda:0 over
    gloo: one JSON line, strong scaling, all 64 structures accounted for."""
    r = _bench_two_ranks(["--config", "cfg3"])
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["config"]["structures_per_step"] == 64
    assert r["value"] > 0 and 0 < r["roofline"]["frac"] <= 1.0


@pytest.mark.timeout(900)
def test_bench_default_config_two_ranks():
    """The driver's N > 1 command on the default configuration (cfg2, weak scaling: one 400-structure replica per rank,
    weights broadcast, coordinates all-gathered inside the timed region)."""
    r = _bench_two_ranks([])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["config"]["structures_per_step"] == 800
    assert r["metric"].startswith("sampled all-atom structures/sec") and r["value"] > 0
    assert "cpu_baseline" not in r and "f32_mfma" not in r          # rank-0, N = 1 only


def _cli(extra, tmp_path, ranks=1, timeout=500):
    base = [os.path.join(ROOT, "test.py"), "--synthetic", "--synthetic_weights", "--synthetic_frames", "2", "--num_ensemble", "2",
            "--data_type", "PED", "--vae_type", "N6", "--exp", "clitest"] + extra
    if ranks > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port())] + base
    else:
        cmd = [sys.executable] + base
    return subprocess.run(cmd, env=_env(), cwd=str(tmp_path), capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(900)
def test_cli_single_rank_two_ranks_and_flow_sampling(tmp_path):
    """The drop-in CLI (reference test.py flags) end to end on the GPU: DDPM sampling on one rank (with the PDB
    writer) and dealt over two ranks (gloo rehearsal on one GPU) must report the same number of structures; the
    flow-matching branch (--model fm, fixed-grid Euler) runs the ODE sampler."""
    (tmp_path / "a").mkdir()
    one = _cli(["--num_sampling_steps", "10", "--save_pdb"], tmp_path / "a")
    assert one.returncode == 0, one.stdout[-1500:] + one.stderr[-3000:]
    assert "done: 16 structures on 1 GPU(s)" in one.stdout                # 4 proteins x 2 frames x 2 members
    out_dir = os.path.join(tmp_path, "a", "logs", "generated_samples_0_best", "clitest_PED")
    files = sorted(os.listdir(out_dir))
    assert [f for f in files if f.endswith(".npy")] == [f"synthetic_L{L}_xyz_recon.npy" for L in (129, 46, 87, 92)]
    pdb = open(os.path.join(out_dir, "generated_traj_synthetic_L46.pdb")).read()
    assert pdb.count("MODEL ") == 4
    xyz = np.load(os.path.join(out_dir, "synthetic_L46_xyz_recon.npy"))
    assert xyz.shape[:2] == (2, 2) and np.isfinite(xyz).all()
    (tmp_path / "b").mkdir()
    two = _cli(["--num_sampling_steps", "10"], tmp_path / "b", ranks=2)
    assert two.returncode == 0, two.stdout[-1500:] + two.stderr[-3000:]
    assert "done: 16 structures on 2 GPU(s)" in two.stdout
    assert "ranks=2 backend=gloo" in two.stdout and "equal on every rank" in two.stdout
    # only rank 0 loaded weights, the noise of a batch is a function of the batch alone: the files of the 2-rank run
    # are the 1-rank run's bit for bit
    out_two = os.path.join(tmp_path, "b", "logs", "generated_samples_0_best", "clitest_PED")
    assert sorted(f for f in os.listdir(out_two) if f.endswith(".npy")) == [f for f in files if f.endswith(".npy")]
    for f in (f for f in files if f.endswith(".npy")):
        assert np.array_equal(np.load(os.path.join(out_dir, f)), np.load(os.path.join(out_two, f))), \
            f"{f}: the 2-rank run's coordinates differ from the 1-rank run's"
    (tmp_path / "c").mkdir()
    fm = _cli(["--model", "fm", "--method", "euler", "--steps", "6"], tmp_path / "c")
    assert fm.returncode == 0, fm.stdout[-1500:] + fm.stderr[-3000:]
    assert "done: 16 structures on 1 GPU(s)" in fm.stdout


@pytest.mark.timeout(900)
def test_cli_recon_and_genzprot(tmp_path):
    """`--experiment recon` (the VQ-VAE's e3nn encoder on the batch's atoms -> VQ -> IC decoder -> xyz; BASELINE config 5
    from atoms) and `--experiment genzprot` (the C2 prior's sample -> C2 IC decoder) through the drop-in CLI; recon is
    deterministic, so its two ensemble members are equal and a 2-rank run writes the 1-rank run's files."""
    (tmp_path / "a").mkdir()
    one = _cli(["--experiment", "recon"], tmp_path / "a")
    assert one.returncode == 0, one.stdout[-1500:] + one.stderr[-3000:]
    assert "done: 16 structures on 1 GPU(s)" in one.stdout
    out_dir = os.path.join(tmp_path, "a", "logs", "generated_samples_0_best", "clitest_PED")
    xyz = np.load(os.path.join(out_dir, "synthetic_L87_xyz_recon.npy"))
    assert xyz.shape[:2] == (2, 2) and np.isfinite(xyz).all() and np.array_equal(xyz[0], xyz[1])
    (tmp_path / "b").mkdir()
    two = _cli(["--experiment", "recon"], tmp_path / "b", ranks=2)
    assert two.returncode == 0, two.stdout[-1500:] + two.stderr[-3000:]
    out_two = os.path.join(tmp_path, "b", "logs", "generated_samples_0_best", "clitest_PED")
    for f in sorted(os.listdir(out_dir)):
        assert np.array_equal(np.load(os.path.join(out_dir, f)), np.load(os.path.join(out_two, f))), f
    (tmp_path / "c").mkdir()
    gz = _cli(["--experiment", "genzprot"], tmp_path / "c")
    assert gz.returncode == 0, gz.stdout[-1500:] + gz.stderr[-3000:]
    assert "done: 16 structures on 1 GPU(s)" in gz.stdout
    xyz = np.load(os.path.join(tmp_path, "c", "logs", "generated_samples_0_best", "clitest_PED", "synthetic_L46_xyz_recon.npy"))
    assert np.isfinite(xyz).all() and not np.array_equal(xyz[0], xyz[1])          # members = different prior samples


@pytest.mark.timeout(300)
def test_rccl_backend_runs_the_collectives_on_device_tensors():
    """The `nccl` (RCCL) branch of codlad_amd.parallel on real device tensors, as far as one GPU allows: world size 1
    (tests/nccl_single_rank_worker.py).  N > 1 over RCCL needs N GPUs and is the driver's to run."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_single_rank_worker.py"), str(_free_port())],
                         env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
    assert "nccl single-rank collectives ok: backend nccl" in res.stdout
    assert "ranks=1 backend=nccl" in res.stdout
