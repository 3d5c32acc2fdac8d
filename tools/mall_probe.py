"""Does a job whose edge state fits the 256 MB Infinity Cache run faster per node, and from what size on do two half-jobs on two
streams pay?  100-step loops over n structures of
L = 87 (K = 64: 32 KB of edge state per node), one job at a time: ns per node and step.
    [CODLAD_HIP_LIB=variants/libcodlad_nont.so] python tools/mall_probe.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from codlad_amd import synth
from codlad_amd.engine import Denoiser
from codlad_amd.diffusion_and_flow.schedule import Tables, named_betas, space_timesteps
torch.set_grad_enabled(False)
DEV = torch.device("cuda", 0)
den = Denoiser(synth.denoiser_state_dict(1234), DEV, precision="f16x3")
T = 100
tables = Tables(named_betas("linear", 1000), space_timesteps(1000, str(T)))
prot = synth.make_protein(87, 1000, n_frames=10)
xyz = [torch.from_numpy(prot["xyz_full"])[f, 1:-1] for f in range(10)]
zz = [torch.from_numpy(prot["z_full"])[1:-1] for _ in range(10)]
st = den.prepare_structures(xyz, zz)
for n_struct in (400, 200, 100, 60, 40, 20):
    members = [i % 10 for i in range(n_struct)]
    job = den.make_job(st, members)
    n = job.n_nodes
    x = synth.gaussian((n, 3), 3).to(DEV)
    eps = synth.gaussian((T, n, 3), 4).to(DEV)
    line = f"{n_struct:4d} structures  {n:6d} nodes  edge state {n * 32768 / 2**20:7.1f} MiB"
    for streams in (1, 2):
        den.sample(job, x, eps, tables, streams=streams)
        torch.cuda.synchronize()
        reps = max(1, 400 // n_struct)
        t0 = time.perf_counter()
        for _ in range(reps):
            den.sample(job, x, eps, tables, streams=streams)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        line += f"   {streams} stream(s): {dt * 1e3:8.2f} ms  {dt / T / n * 1e9:7.2f} ns per node-step  {n_struct / dt:7.1f} structures/s"
    print(line, flush=True)
