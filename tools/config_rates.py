"""Whole-path rates (noise -> xyz, 100 steps) of BASELINE.json's other configurations on one GPU, as the
full-size tests build them (tests/pipeline.py, tests/test_full_size.py).  python tools/config_rates.py"""
import sys, time
import torch
sys.path.insert(0, '.')
from codlad_amd import parallel
from tests import pipeline
from tests.test_full_size import PED_LENGTHS, atlas_like_lengths


def rate(cfg, units, label, **kw):
    cfg.run_units(units, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cfg.run_units(units, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nodes = sum(cfg.lengths[cfg.units[u][0]] for u in units)
    print(f"{label}: {len(units)} structures, {nodes} nodes, {dt:.2f} s -> {len(units) / dt:.1f} structures/s", flush=True)


cfg = pipeline.Config("cfg2", PED_LENGTHS, 10, 10, "N6", "PED")
rate(cfg, list(range(400)), "cfg 2 (PED N6, incl. per-unit noise generation and per-unit ic_to_xyz calls)")
rate(cfg, list(range(400)), "cfg 5 (decoder only, N6)", decode_only=True)
lengths = [max(50, min(400, L)) for L in atlas_like_lengths(64, seed=11)]
cfg = pipeline.Config("cfg3", lengths, 1, 1, "K3", "PDB")
rate(cfg, list(range(64)), "cfg 3 (64 proteins of 50-400 residues, K3), all on one GPU")
lengths = atlas_like_lengths(70)
cfg = pipeline.Config("cfg4", lengths, 4, 32, "K4", "Atlas")
costs = [parallel.unit_cost(lengths[p]) for p, _f, _m in cfg.units]
rate(cfg, parallel.shard_units(costs, 8)[0], "cfg 4 (Atlas K4): one GPU's share of 8 960 structures")
