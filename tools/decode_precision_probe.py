"""How far are the two message-kernel variants of the IC decoder (CODLAD_OPT_DEC_EDGE_VARIANT 0 / 1) and the fp32 CPU
oracle from an fp64 evaluation of the same decoder?  K3 decoder (synthetic weights) on cfg-3-sized proteins.
    python tools/decode_precision_probe.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from codlad_amd import _lib, synth  # noqa: E402
from codlad_amd.engine import Decoder  # noqa: E402
from oracle import vae_decode as odec  # noqa: E402

torch.set_grad_enabled(False)
dev = "cuda:0"
for vae_type, dataname, L, seed in (("K3", "PDB", 155, 1055), ("K3", "PDB", 400, 1007), ("K4", "Atlas", 505, 1003), ("N6", "PED", 129, 1002)):
    vsd = synth.vqvae_state_dict(vae_type, dataname, 4321)
    mean, std = synth.norm_stats(dataname, vae_type)
    prot = synth.make_protein(L, seed, n_frames=1, phospho=False)
    batch = synth.make_batch(prot)
    lat = synth.gaussian((1, L, 3), seed + 5) * std + mean
    angle = vae_type != "N6"
    idx, ic32 = odec.latent_decode(vsd, lat, batch, angle=angle)
    vsd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in vsd.items()}
    b64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()}
    _i, ic64 = odec.latent_decode(vsd64, lat.double(), b64, angle=angle)
    og = batch["OG_CG_nxyz"].reshape(-1, L + 2, 4)
    x64 = odec.ic_to_xyz(og.double(), ic64.reshape(-1, L, 13, 3), prot["info"])
    x32 = odec.ic_to_xyz(og, ic32.reshape(-1, L, 13, 3), prot["info"])
    dec = Decoder(vsd, dev)
    rows = [("oracle fp32", ic32, x32[0])]
    for v in (0, 1):
        _lib.set_option(_lib.OPT_DEC_EDGE_VARIANT, v)
        gi, zq, _ = dec.vq(lat.to(dev), normalised=False)
        assert torch.equal(gi.cpu(), idx.reshape(-1))
        ic = dec.ic_decode(zq.reshape(-1, 3), batch["CG_nxyz"][:, 0].long(), batch["CG_nxyz"][:, 1:], batch["CG_nbr_list"])
        xyz = dec.ic_to_xyz(og[:, :, 1:].to(dev), ic.view(1, L, 13, 3), prot["info"])
        rows.append((f"HIP variant {v}", ic.cpu(), xyz[0].cpu()))
    _lib.set_option(_lib.OPT_DEC_EDGE_VARIANT, 0)
    print(f"{vae_type} L={L}")
    for name, ic, xyz in rows:
        e_ic = float((ic.double() - ic64).abs().max() / ic64.abs().max())
        rm64 = float(((xyz.double() - x64[0]) ** 2).sum(-1).mean().sqrt())
        rm32 = float(((xyz - x32[0]) ** 2).sum(-1).mean().sqrt())
        print(f"   {name:14s} ic rel err vs fp64 {e_ic:.2e}   xyz RMSD vs fp64 {rm64:.2e} A   vs oracle fp32 {rm32:.2e} A")
