"""Drop-in for the reference's utils/model_module.py:get_vae_model.

Same signature, same checkpoint locations and file names (reference utils/model_module.py:20-123):
`./results/Vae_vqvae_ns36_vq3_vq4096` (N6), `./results/Vae_vqvaeangle_PDB_ns36_vq3_vq4096` (K3),
`./results/Vae_vqvaeangle_Atlas_ns36_vq3_vq4096` (K4), `./results/Vae_m1_12-23-23_12345` (C2, the GenZProt conditional
VAE); `model.pt` | `best_model.pt` (modelnum 999) | `model_{n}.pt` holding a bare state_dict, + `modelparams.json`.
The models are built with their e3nn encoder (and, for C2, the CG prior and posterior heads); e3nn's own buffers in a
checkpoint (`*.tp.weight`, `*.tp.output_mask`, `*.tp._compiled_main_left_right._w3j_*`) carry no parameters and are
skipped, as are legacy `dist_filter` keys; everything else must match.
"""
import json
import os

import torch

import torch.nn as nn

from ..models.vae_model import VAE, GenZProt, IC_Decoder, IC_Decoder_angle, e3nnEncoder, e3nnPrior
from .vq_module import build_quantize

_DIRS = {"N6": "Vae_vqvae_ns36_vq3_vq4096", "K3": "Vae_vqvaeangle_PDB_ns36_vq3_vq4096",
         "K4": "Vae_vqvaeangle_Atlas_ns36_vq3_vq4096", "C2": "Vae_m1_12-23-23_12345"}
_SKIPPED_PREFIXES = ("encoder.", "prior_net.", "atom_munet.", "atom_sigmanet.")     # of a decoder-only model


def load_params(model_dir):
    with open(os.path.join(model_dir, 'modelparams.json'), 'rt') as f:
        return json.load(f)


def build_vae(modeltype, device="cpu", with_encoder=False):
    """with_encoder=False: the decoder side alone (what `--experiment latent` needs; such a model loads a full
    checkpoint by skipping its encoder tensors).  True: as the reference builds it (utils/model_module.py:20-78)."""
    if modeltype not in _DIRS:
        raise NotImplementedError(f"vae type {modeltype!r}: N6 / K3 / K4 (VQ-VAE) and C2 (GenZProt) are built")
    embed_dim, cg_cutoff, atom_cutoff, enc_nconv, dec_nconv, n_rbf, activation = 36, 21.0, 9.0, 3, 4, 15, "swish"
    codebook_temp, codebook_ema_decay, vqdim, codebook_size = 0.25, 0.99, 3, 4096
    encoder = None
    if with_encoder or modeltype == "C2":
        encoder = e3nnEncoder(device=device, n_atom_basis=embed_dim, use_second_order_repr=False,
                              num_conv_layers=enc_nconv, cross_max_distance=cg_cutoff + 5,
                              atom_max_radius=atom_cutoff + 5, cg_max_radius=cg_cutoff + 5)
    dec_cls = IC_Decoder if modeltype in ("N6", "C2") else IC_Decoder_angle
    dec = dec_cls(n_atom_basis=embed_dim, n_rbf=n_rbf, cutoff=cg_cutoff, num_conv=dec_nconv, activation=activation)
    if modeltype == "C2":
        prior_net = e3nnPrior(device=device, n_atom_basis=embed_dim, use_second_order_repr=False,
                              num_conv_layers=enc_nconv, cg_max_radius=cg_cutoff + 5)
        munet = nn.Sequential(nn.Linear(embed_dim, embed_dim), nn.ReLU(), nn.Linear(embed_dim, embed_dim))
        sigmanet = nn.Sequential(nn.Linear(embed_dim, embed_dim), nn.ReLU(), nn.Linear(embed_dim, embed_dim))
        return GenZProt(encoder, dec, munet, sigmanet, 5, feature_dim=embed_dim, prior_net=prior_net, det=False,
                        equivariant=True).to(device)
    quantize = build_quantize("vqvae", codebook_size, vqdim, codebook_temp, codebook_ema_decay)
    return VAE(5, embed_dim, encoder, quantize=quantize, equivaraintconv=dec, prior_net=None, atom_munet=None,
               atom_sigmanet=None, vqdim=vqdim).to(device)


def load_decoder_state(model, state_dict):
    """strict for the decoder side, tolerant of the encoder side and of legacy `dist_filter` keys
    (reference model_module.py:91-108)."""
    own = model.state_dict()
    skip = tuple(p for p in _SKIPPED_PREFIXES if not any(k.startswith(p) for k in own))   # sides this model lacks
    kept = {k: v for k, v in state_dict.items()
            if not (skip and k.startswith(skip)) and ".dist_filter." not in k and ".tp." not in k}
    missing = [k for k in own if k not in kept]
    unexpected = [k for k in kept if k not in own]
    if missing or unexpected:
        raise RuntimeError(f"VQ-VAE checkpoint does not match the decoder layout: missing {missing[:5]}, "
                           f"unexpected {unexpected[:5]}")
    model.load_state_dict(kept, strict=True)
    return model


def get_vae_model(modeltype, modelpath=None, device="cpu", modelnum=-1, with_encoder=True):
    model = build_vae(modeltype, device, with_encoder=with_encoder)
    vqvae_path = modelpath if modelpath is not None else os.path.join("./results/", _DIRS[modeltype])
    if modelnum == -1 or modeltype == "C2":
        name = 'model.pt'
    elif modelnum == 999:
        name = 'best_model.pt'
    else:
        name = f'model_{modelnum}.pt'
    ckpt = torch.load(os.path.join(vqvae_path, name), map_location=torch.device('cpu'))
    load_decoder_state(model, ckpt)
    params = load_params(vqvae_path)
    return model, params
