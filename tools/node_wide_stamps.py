"""Phase stamps of node_kernel_w (diagnostic build -DNW_STAMP: python tools/edge_variants.py build node_wide_kernels.hip nwst=-DNW_STAMP,
then CODLAD_HIP_LIB=variants/libcodlad_nwst.so python tools/node_wide_stamps.py on the GPU box).  The stamped build leaves the
cycle counts of workgroup 0's first wave in h_V[0][0..8] of the LAST node update of a forward (results are garbage)."""
import sys

import torch

sys.path.insert(0, '.')
from codlad_amd import synth                     # noqa: E402
from codlad_amd.engine import Denoiser           # noqa: E402

torch.set_grad_enabled(False)
den = Denoiser(synth.denoiser_state_dict(1234), "cuda:0")
names = ["loads + publish S -> barrier", "W3 + xch -> barrier", "LN1 + publish -> barrier", "W_in x2 + publish -> barrier",
         "W_out x4 + xch -> barrier", "LN2", "store h_V, publish -> barrier", "projections + stores issued", "stores drained",
         "  phase 1: to the staging code's end", "  phase 1: S published, barrier",
         "    start -> all loads issued", "    node_info arrived", "    S planes arrived", "    h_V quarter arrived", "    W3, W_in 0 quarters arrived"]
for L in (87, 300):
    p = synth.make_protein(L, 50, n_frames=1)
    st = den.prepare_structures([torch.from_numpy(p["xyz_full"])[0, 1:-1]], [torch.from_numpy(p["z_full"])[1:-1]])
    job = den.make_job(st, [0])
    x = torch.randn(job.n_nodes, 3, device="cuda")
    acc = torch.zeros(16, dtype=torch.float64)
    n = 20
    for _ in range(n):
        den.forward(job, x, 500)
        torch.cuda.synchronize()
        acc += job.hV[0, :16].double().cpu()
    acc /= n
    tot = acc[:9].sum()
    print(f"L = {L}: {tot:.0f} s_memtime ticks per launch")
    for nm, v in zip(names, acc.tolist()):
        print(f"   {nm:44s} {v:9.0f}  ({100 * v / tot:4.1f} %)")
