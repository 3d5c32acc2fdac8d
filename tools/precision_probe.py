"""f16x3 / f16x4 vs f32-MFMA vs the CPU oracle on one forward and a 100-step loop (GPU)."""
import sys, torch
sys.path.insert(0, '.')
from codlad_amd import synth
from codlad_amd.engine import Denoiser
from codlad_amd.diffusion_and_flow.schedule import Tables, named_betas, space_timesteps
from oracle import denoiser as oden
from tests import cases
sd = synth.denoiser_state_dict(1234)
sd64 = {k: v.double() for k, v in sd.items()}
L, B, seed = cases.DENOISER_CASES["L87_B2"]
prot, batch, x, t, mask = cases.denoiser_inputs(L, B, seed)
cg_z, cg_xyz, m = oden.batch_to_dense(batch)
ref32 = oden.forward(sd, x, t, cg_xyz, cg_z, mask)
def rel(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max())
outs = {}
for prec in ("f32", "f16x4", "f16x3"):
    d = Denoiser(sd, "cuda:0", precision=prec)
    frames = torch.from_numpy(prot["xyz_full"])[:, 1:-1]; z = torch.from_numpy(prot["z_full"])[1:-1]
    st = d.prepare_structures([f for f in frames], [z] * B); job = d.make_job(st, list(range(B)))
    o = d.forward(job, x.reshape(-1, 3).cuda(), int(t[0])).cpu().view(B, L, 6)
    outs[prec] = o
    T = 100
    tb = Tables(named_betas("linear", 1000), space_timesteps(1000, str(T)))
    zz, eps = cases.loop_noise(T, B, L, seed)
    outs[prec + "_loop"] = d.sample(job, zz.reshape(-1, 3).cuda(), eps.reshape(T, -1, 3).cuda(), tb).cpu()
    print(prec, "forward vs CPU fp32 oracle:", rel(o, ref32))
for prec in ("f16x4", "f16x3"):
    print(f"forward {prec} vs f32-MFMA:", rel(outs[prec], outs["f32"]))
    print(f"100-step loop {prec} vs f32-MFMA:", rel(outs[prec + "_loop"], outs["f32_loop"]))
