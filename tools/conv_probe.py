"""One graph, one layer: the depth-2 atom-graph convolution of the L = 129 protein's 10 frames (10 700 atoms, ~0.96 M
directed edges), launched 6 times - the target of tools/pmc_conv.sh.   python tools/conv_probe.py [depth]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codlad_amd import synth  # noqa: E402
from codlad_amd.encoder import Encoder, directed_csr  # noqa: E402

torch.set_grad_enabled(False)
dev = "cuda:0"
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 2
enc = Encoder(synth.encoder_state_dict(778), dev)
prot = synth.make_protein(129, 1003, n_frames=10)
atoms = synth.make_atoms(prot, seed=3)
xa, ta = atoms["nxyz"][:, 1:].to(dev).contiguous(), atoms["nxyz"][:, 0].to(dev).contiguous()
na = xa.shape[0]
csr = directed_csr(atoms["nbr_list"].to(dev), na)
g = torch.Generator(device=dev).manual_seed(1)
h = torch.randn(na, 12 * (depth + 1), generator=g, device=dev)
out = torch.empty(na, 12 * (depth + 2), device=dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for i in range(6):
    if i == 1:
        ev[0].record()
    enc.conv(f"atom_conv_layers.{depth}", depth, csr, xa, xa, ta, ta, 1.0, 14.0, "atom_edge_embedding", 14, h, h, True, out,
             False, 64)
ev[1].record()
torch.cuda.synchronize()
print(f"depth {depth}: {na} atoms, {int(csr[0][-1])} directed edges, {ev[0].elapsed_time(ev[1]) / 5 * 1e3:.1f} us per launch")
