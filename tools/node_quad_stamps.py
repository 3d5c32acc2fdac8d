"""Phase stamps of node_kernel_q (diagnostic build: python tools/edge_variants.py build node_quad_kernels.hip nqst=-DNQ_STAMP, then
CODLAD_HIP_LIB=variants/libcodlad_nqst.so python tools/node_quad_stamps.py on the GPU box).  The stamped build leaves the cycle
counts of workgroup 0's first wave in the first row of the LAST projecting node update's first output (decoder layer 1: PQ plane
0, two projections); results are garbage."""
import sys

import torch

sys.path.insert(0, '.')
from codlad_amd import synth                     # noqa: E402
from codlad_amd.engine import Denoiser           # noqa: E402

torch.set_grad_enabled(False)
den = Denoiser(synth.denoiser_state_dict(1234), "cuda:0")
names = ["arguments, rows + first quarters, S summed and published -> barrier", "W3, residual / 30, fp32 exchange -> barrier",
         "LayerNorm 1 + publish -> barrier", "W_in 0, W_in 1 + activations", "W_in 2, W_in 3 + activations -> barrier",
         "W_out x 4, residual, exchange -> barrier", "LayerNorm 2, store h_V, publish -> barrier", "projections (2) + stores issued",
         "stores drained", "  phase 1: arguments read (one batch)", "  phase 1: rows and three quarters requested",
         "  phase 1: rows arrive, S summed, published, barrier"]
for L in (87, 300):
    p = synth.make_protein(L, 50, n_frames=1)
    st = den.prepare_structures([torch.from_numpy(p["xyz_full"])[0, 1:-1]], [torch.from_numpy(p["z_full"])[1:-1]])
    job = den.make_job(st, [0])
    x = torch.randn(job.n_nodes, 3, device="cuda")
    acc = torch.zeros(12, dtype=torch.float64)
    n = 20
    for _ in range(n):
        den.forward(job, x, 500)
        torch.cuda.synchronize()
        acc += job.PQ[0, 0, :12].double().cpu()
    acc /= n
    tot = acc[:9].sum()
    print(f"L = {L}: {tot:.0f} s_memtime ticks per launch")
    for nm, v in zip(names, acc.tolist()):
        print(f"   {nm:72s} {v:9.0f}  ({100 * v / tot:4.1f} %)")
