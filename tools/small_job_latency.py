"""Latency of small jobs (the regime of one GPU's shard of cfg 3, and of cfg 1): the 100-step DDPM loop on
1 protein of 87 residues, 1 of 300, an 8-protein shard of cfg 3 (~1.3 k nodes) and all 64 proteins of cfg 3.

    python tools/small_job_latency.py [--only N] [--reps R]
"""
import argparse
import sys
import time

import torch

sys.path.insert(0, '.')
from codlad_amd import parallel, synth                                                   # noqa: E402
from codlad_amd.diffusion_and_flow.schedule import Tables, named_betas, space_timesteps   # noqa: E402
from codlad_amd.engine import Denoiser                                                    # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--only", type=int, default=-1)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--sweep", action="store_true", help="k copies of one 87-residue protein, k = 1 .. 24, instead of the four cases")
ap.add_argument("--ks", default="", help="--sweep: comma-separated copy counts")
ap.add_argument("--streams", type=int, default=None, help="Denoiser.sample(streams=...): the job as this many part-jobs on as many HIP streams")
args = ap.parse_args()

torch.set_grad_enabled(False)
den = Denoiser(synth.denoiser_state_dict(1234), "cuda:0")
tb = Tables(named_betas("linear", 1000), space_timesteps(1000, "100"))
cfg3 = synth.baseline_config("cfg3")["lengths"]
shard = parallel.shard_units([parallel.unit_cost(L) for L in cfg3], 8)[0]
CASES = [("1 protein, L=87", [87]), ("1 protein, L=300", [300]),
         ("cfg3 shard of one GPU (1/8, LPT): %d proteins" % len(shard), [cfg3[u] for u in shard]),
         ("cfg3, all 64 proteins", cfg3)]
if args.sweep:
    CASES = [("%d x L=87" % k, [87] * k) for k in ([int(v) for v in args.ks.split(",")] if args.ks else (1, 2, 3, 4, 6, 8, 10, 12, 16, 20, 24))]
for k, (label, lens) in enumerate(CASES):
    if args.only >= 0 and k != args.only:
        continue
    prots = [synth.make_protein(L, 50 + i, n_frames=1) for i, L in enumerate(lens)]
    xs = [torch.from_numpy(p["xyz_full"])[0, 1:-1] for p in prots]
    zs = [torch.from_numpy(p["z_full"])[1:-1] for p in prots]
    st = den.prepare_structures(xs, zs)
    job = den.make_job(st, list(range(len(lens))))
    xT = torch.randn(job.n_nodes, 3, device="cuda")
    eps = torch.randn(100, job.n_nodes, 3, device="cuda")
    den.sample(job, xT, eps, tb, streams=args.streams)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        den.sample(job, xT, eps, tb, streams=args.streams)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    print(f"{label:48s} {job.n_nodes:6d} nodes: {dt * 1e3:8.2f} ms per 100-step loop = {dt / 100 * 1e6:7.1f} us/step, "
          f"{len(lens) / dt:8.1f} structures/s", flush=True)
