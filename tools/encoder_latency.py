"""Time of the e3nn encoder / prior kernels on cfg-2-shaped input: the 40 frames (4 proteins x 10) whose 400 ensemble
members cfg 5 decodes, synthetic atoms (codlad_amd.synth.make_atoms).   python tools/encoder_latency.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codlad_amd import synth  # noqa: E402
from codlad_amd.encoder import Encoder, Prior  # noqa: E402

torch.set_grad_enabled(False)
dev = "cuda:0"
enc, pri = Encoder(synth.encoder_state_dict(778), dev), Prior(synth.prior_state_dict(777), dev)
batches = []
for i, L in enumerate(synth.PED_LENGTHS):
    prot = synth.make_protein(L, 1000 + i, n_frames=10)
    b = synth.make_batch(prot)
    b.update(synth.make_atoms(prot, seed=i))
    batches.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()})


def run_enc():
    return [enc.forward(b["nxyz"][:, 0], b["nxyz"][:, 1:], b["CG_nxyz"][:, 0].long(), b["CG_nxyz"][:, 1:], b["CG_mapping"],
                        b["nbr_list"], b["CG_nbr_list"]) for b in batches]


def run_pri():
    return [pri.forward(b["CG_nxyz"][:, 0].long(), b["CG_nxyz"][:, 1:], b["CG_nbr_list"]) for b in batches]


for name, fn in (("encoder", run_enc), ("prior", run_pri)):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    n_atoms = sum(int(b["nxyz"].shape[0]) for b in batches)
    n_edges = sum(2 * int(b["nbr_list"].shape[0]) for b in batches)
    print(f"{name}: {dt * 1e3:.2f} ms for 40 frames ({n_atoms} atoms, {n_edges} directed atom edges, "
          f"{sum(int(b['CG_nxyz'].shape[0]) for b in batches)} beads) = {40 / dt:.0f} frames/s", flush=True)
