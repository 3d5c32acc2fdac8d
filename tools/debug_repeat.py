import sys, torch
sys.path.insert(0, '.')
from codlad_amd import synth
from codlad_amd.engine import Denoiser
from tests import cases
sd = synth.denoiser_state_dict(1234)
for prec in ("f32", "f16x4"):
    den = Denoiser(sd, "cuda:0", precision=prec)
    for name in ("L46_B2", "L129_B3"):
        L, B, seed = cases.DENOISER_CASES[name]
        prot, batch, x, t, mask = cases.denoiser_inputs(L, B, seed)
        frames = torch.from_numpy(prot["xyz_full"])[:, 1:-1]
        z = torch.from_numpy(prot["z_full"])[1:-1]
        st = den.prepare_structures([f for f in frames], [z] * B)
        job = den.make_job(st, list(range(B)))
        xx = x.reshape(-1, 3).cuda()
        ref = den.forward(job, xx, 500).clone()
        bad = 0
        for i in range(50):
            o = den.forward(job, xx, 500)
            if not torch.equal(o, ref):
                bad += 1
                if bad < 3:
                    d = (o - ref)
                    print("   mismatch", i, "nan:", int(torch.isnan(o).sum()), "maxdiff", float(d.abs().nan_to_num().max()),
                          "rows differing", torch.nonzero((d != 0).any(1) | torch.isnan(d).any(1)).flatten()[:10].tolist())
        print(prec, name, "mismatches", bad, "ref nan", int(torch.isnan(ref).sum()))
