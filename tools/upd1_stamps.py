"""Where a tile of upd1_kernel_h spends its cycles (diagnostic build -DU1_STAMP: python tools/edge_variants.py build
edge_upd1_kernel.hip stamp=-DU1_STAMP; then CODLAD_HIP_LIB=variants/libcodlad_stamp.so python tools/upd1_stamps.py)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from codlad_amd import _lib
torch.set_grad_enabled(False)
_lib.set_option(5, 1)
wl = bench.Workload(torch.device("cuda", 0), "cfg2")
job = wl.job
g = torch.Generator(device=job.hE.device).manual_seed(5)
halves = job.hE.view(torch.float16).view(job.n_nodes, 2, 8, 4, 32, 8)
halves[:, :, :, 0:2] = torch.randn(halves[:, :, :, 0:2].shape, generator=g, device=job.hE.device, dtype=torch.float16)
halves[:, :, :, 2:4] = torch.randn(halves[:, :, :, 2:4].shape, generator=g, device=job.hE.device, dtype=torch.float16) * 2.0 ** -12
job.PQ.normal_(generator=g)
for _ in range(3):
    r = wl.time_dominant_kernel(8)
print(f"back to back: upd {r['edge_update'] * 1e3:.4f} ms")
S = wl.job.S.view(-1)[: 256 * 4 * 16].view(1024, 16).cpu()
names = ["advance", "P+Q (wait for rows)", "layer 1 + residual", "walk taken, rows requested", "layer 2", "layer 3", "Q requested",
         "LayerNorm, split, stores"]
tiles = S[:, 9].clamp_min(1)
tot = S[:, 8]
print(f"waves {len(S)}  tiles per wave {tiles.mean():.1f}  cycles per wave {tot.mean():.0f} (min {tot.min():.0f} max {tot.max():.0f})  per tile {(tot / tiles).mean():.0f}")
for k, nm in enumerate(names):
    per = S[:, k] / tiles
    print(f"  {nm:30s} {per.mean():8.0f} cycles per tile  ({100 * S[:, k].sum() / tot.sum():5.1f} %)")
