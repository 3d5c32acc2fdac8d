"""A/B of run-time kernel variants (codlad_set_option) on ONE box, in ONE process: per setting the whole-job rate of
bench.py's cfg2 workload, the in-job launch times of the edge kernels and the back-to-back launch times.

    python tools/ab_variants.py DEC_EDGE_VARIANT=0 DEC_EDGE_VARIANT=1 [--rounds 2] [--config cfg5]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from codlad_amd import _lib  # noqa: E402

OPTS = {"NODEQ_MAX_TILES": 0, "EDGE_TILE_MAX_NODES": 1, "DEC_EDGE_VARIANT": 3, "TP_CONV_VARIANT": 4, "EDGE_UPD_VARIANT": 5, "EDGE_CUS": 6}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("settings", nargs="+", help="NAME=VALUE[,NAME=VALUE...] per arm")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--streams", type=int, default=None, help="1: the whole job on one stream; default: as the product runs it")
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    dev = torch.device("cuda", 0)
    wl = bench.Workload(dev, args.config)
    wl.run(streams=args.streams)
    torch.cuda.synchronize()
    for rnd in range(args.rounds):
        for arm in args.settings:
            pairs = [kv.split("=") for kv in arm.split(",")]
            for k, v in pairs:
                _lib.set_option(OPTS[k], int(v))
            wl.run(streams=args.streams)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            wl.run(streams=args.streams)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            line = f"{arm:40s} {wl.n_structures / dt:8.1f} structures/s"
            if not wl.decode_only:
                insitu = bench.probe_edge_kernels(wl)
                b2b = wl.time_dominant_kernel(10)
                line += "   in job: " + "  ".join(f"{k} {v['ms']:.4f}" for k, v in insitu.items())
                line += f"   back to back: msg {b2b['message'] * 1e3:.4f} upd {b2b['edge_update'] * 1e3:.4f} ms"
            print(line, flush=True)


if __name__ == "__main__":
    main()
