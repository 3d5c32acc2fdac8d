import sys; sys.path.insert(0, '.')
import torch
from codlad_amd import synth, engine
from codlad_amd.engine import Denoiser
from tests import cases
torch.set_grad_enabled(False)
L, B, seed = cases.ENVELOPE_GEOMETRY
prot, batch, x, t, mask = cases.denoiser_inputs(L, B, seed)
for scale in (1e3, 1e5, 1e6):
    sd = synth.denoiser_state_dict(cases.WEIGHT_SEED)
    sd["features.norm_edges.weight"] = sd["features.norm_edges.weight"] * scale
    den = Denoiser(sd, "cuda:0", precision="f16x3")
    frames = torch.from_numpy(prot["xyz_full"])[:, 1:-1]
    z = torch.from_numpy(prot["z_full"])[1:-1]
    st = den.prepare_structures([f for f in frames], [z for _ in frames])
    job = den.make_job(st, list(range(B)))
    K = min(64, L)
    hE0 = engine.edge_rows(st.h_E0)[:, :K]
    E1 = engine.edge_rows(st.E1)[:, :, :K]
    print(scale, "hE0 absmax", float(hE0.abs().max()), "E1 finite", bool(torch.isfinite(E1).all()), "E1 absmax", float(torch.nan_to_num(E1, nan=0.0, posinf=0.0, neginf=0.0).abs().max()),
          "nan frac", float(torch.isnan(E1).float().mean()), "inf frac", float(torch.isinf(E1).float().mean()))
    out = den.forward(job, x.reshape(-1, 3).to("cuda:0"), 500, check=False)
    print("   out finite", bool(torch.isfinite(out).all()), "status", int(job.status.item()), "S nan", float(torch.isnan(job.S).float().mean()),
          "hV nan", float(torch.isnan(job.hV).float().mean()), "hE nan", float(torch.isnan(engine.edge_rows(job.hE.view(-1, 32, 64, 4))[:, :K]).float().mean()))
