"""Whole-job rate of bench.py's cfg2 / cfg3 workloads with the job dealt into k part-jobs on k HIP streams (Denoiser.sample(streams=k)),
k = 1 .. 4, alternating over rounds on ONE box.

    python tools/stream_count_probe.py [--config cfg2] [--rounds 3]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="cfg2")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--ks", default="1,2,3,4")
args = ap.parse_args()
torch.set_grad_enabled(False)
wl = bench.Workload(torch.device("cuda", 0), args.config)
ks = [int(k) for k in args.ks.split(",")]
for k in ks:
    wl.run(streams=k)
torch.cuda.synchronize()
for rnd in range(args.rounds):
    for k in ks:
        t0 = time.perf_counter()
        wl.run(streams=k)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"round {rnd}  streams {k}: {wl.n_structures / dt:8.1f} structures/s", flush=True)
