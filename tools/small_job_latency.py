"""Latency of small jobs (launch-bound regime): 100-step loop for a few structures of L=87."""
import sys, time, torch
sys.path.insert(0, '.')
from codlad_amd import synth
from codlad_amd.engine import Denoiser
from codlad_amd.diffusion_and_flow.schedule import Tables, named_betas, space_timesteps
sd = synth.denoiser_state_dict(1234)
den = Denoiser(sd, "cuda:0")
tb = Tables(named_betas("linear", 1000), space_timesteps(1000, "100"))
for B in (1, 4, 10, 40):
    prot = synth.make_protein(87, 5, n_frames=1)
    x = torch.from_numpy(prot["xyz_full"])[0, 1:-1]; z = torch.from_numpy(prot["z_full"])[1:-1]
    st = den.prepare_structures([x], [z]); job = den.make_job(st, [0] * B)
    xT = torch.randn(job.n_nodes, 3, device="cuda"); eps = torch.randn(100, job.n_nodes, 3, device="cuda")
    den.sample(job, xT, eps, tb); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        den.sample(job, xT, eps, tb)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"B={B:3d} ensemble members of one L=87 frame: {dt*1e3:8.2f} ms per 100-step loop = {dt/100*1e6:7.1f} us/step, {B/dt:8.1f} structures/s")
