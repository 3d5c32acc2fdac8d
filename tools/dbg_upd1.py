"""Debug aid: one edge-update launch (layer 1, in place) with the two kernel variants on the same inputs; where do the
stored tiles differ?"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from codlad_amd import _lib, engine, synth
from codlad_amd.engine import Denoiser
from tests import cases
torch.set_grad_enabled(False)
DEV = "cuda:0"
sd = synth.denoiser_state_dict(cases.WEIGHT_SEED)
_lib.set_option(_lib.OPT_EDGE_TILE_MAX_NODES, 0)
lib = _lib.lib()
for lens in ([87], [40, 87, 33]):
    prots = [synth.make_protein(L, 400 + i, n_frames=1) for i, L in enumerate(lens)]
    xyz = [torch.from_numpy(p["xyz_full"])[0, 1:-1] for p in prots]
    zz = [torch.from_numpy(p["z_full"])[1:-1] for p in prots]
    n = sum(lens)
    x = synth.gaussian((n, 3), 27).to(DEV)
    _lib.set_option(_lib.OPT_EDGE_UPD_VARIANT, 0)
    d = Denoiser(sd, DEV, precision="f16x3")
    st = d.prepare_structures(xyz, zz, hoist_layer0=False)
    job = d.make_job(st, list(range(len(lens))))
    d.forward(job, x, 600)                     # fills PQ, hE with real values
    mods = d.step_mods(torch.tensor([600]))
    stream = torch.cuda.current_stream()
    hE0 = job.hE.clone()
    outs = []
    for variant in (0, 2):
        _lib.set_option(_lib.OPT_EDGE_UPD_VARIANT, variant)
        job.hE.copy_(hE0)
        rc = lib.codlad_bench_edge_launch(C.byref(d.weights.struct), _lib.ptr(job.node_info), job.n_nodes, _lib.ptr(st.E_idx),
                                          _lib.ptr(st.h_E0), _lib.ptr(mods), C.byref(job.ws), 1, 1, C.c_void_p(stream.cuda_stream))
        _lib.check(rc, "launch")
        torch.cuda.synchronize()
        outs.append(engine.edge_rows(job.hE.view(job.n_nodes, 2, 32, 32, 4), split=True).clone())       # [n, 64, 128]
    a, b = outs
    if os.environ.get("DBG_ECHO"):
        a = engine.edge_rows(hE0.view(job.n_nodes, 2, 32, 32, 4), split=True)
    K = job.node_info[:, 2].long()
    valid = (torch.arange(64, device=DEV)[None, :] < K[:, None])
    fin = torch.isfinite(b) | ~valid[..., None]
    diff = ((a != b) & valid[..., None])
    print(f"lens {lens}: variant 1 finite on valid columns: {bool(fin.all())}; differing elements {int(diff.sum())} of {int(valid.sum()) * 128}")
    if int(diff.sum()):
        nodes = diff.any(-1).any(-1).nonzero().flatten()
        print("  differing nodes:", nodes[:20].tolist(), "count", len(nodes))
        nn = int(nodes[0])
        cols = diff[nn].any(-1).nonzero().flatten().tolist()
        print(f"  node {nn} (K={int(K[nn])}) differing columns: {cols}")
        feats = diff[nn, cols[0]].nonzero().flatten().tolist()
        print(f"  node {nn} col {cols[0]} differing features ({len(feats)}): {feats[:40]}")
        print("   ref", a[nn, cols[0], feats[:6]].tolist(), "\n   got", b[nn, cols[0], feats[:6]].tolist())
