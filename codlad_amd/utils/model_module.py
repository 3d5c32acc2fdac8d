"""Drop-in for the reference's utils/model_module.py:get_vae_model (decoder side).

Same signature, same checkpoint locations and file names (reference utils/model_module.py:20-123):
`./results/Vae_vqvae_ns36_vq3_vq4096` (N6), `./results/Vae_vqvaeangle_PDB_ns36_vq3_vq4096` (K3),
`./results/Vae_vqvaeangle_Atlas_ns36_vq3_vq4096` (K4); `model.pt` | `best_model.pt` (modelnum 999) |
`model_{n}.pt` holding a bare state_dict, + `modelparams.json`.  Encoder-side tensors of the
checkpoint (`encoder.*`, e3nn) are not instantiated and are skipped; every decoder-side tensor must
be present.  The C2 conditional prior is out of scope.
"""
import json
import os

import torch

from ..models.vae_model import VAE, IC_Decoder, IC_Decoder_angle
from .vq_module import build_quantize

_DIRS = {"N6": "Vae_vqvae_ns36_vq3_vq4096", "K3": "Vae_vqvaeangle_PDB_ns36_vq3_vq4096",
         "K4": "Vae_vqvaeangle_Atlas_ns36_vq3_vq4096"}
_SKIPPED_PREFIXES = ("encoder.", "prior_net.", "atom_munet.", "atom_sigmanet.")


def load_params(model_dir):
    with open(os.path.join(model_dir, 'modelparams.json'), 'rt') as f:
        return json.load(f)


def build_vae(modeltype, device="cpu"):
    if modeltype not in _DIRS:
        raise NotImplementedError(f"vae type {modeltype!r}: only the VQ-VAE decoders N6 / K3 / K4 are built "
                                  "(C2 = GenZProt prior needs the out-of-scope e3nn encoder)")
    embed_dim, cg_cutoff, dec_nconv, n_rbf, activation = 36, 21.0, 4, 15, "swish"
    codebook_temp, codebook_ema_decay, vqdim, codebook_size = 0.25, 0.99, 3, 4096
    dec_cls = IC_Decoder if modeltype == "N6" else IC_Decoder_angle
    dec = dec_cls(n_atom_basis=embed_dim, n_rbf=n_rbf, cutoff=cg_cutoff, num_conv=dec_nconv, activation=activation)
    quantize = build_quantize("vqvae", codebook_size, vqdim, codebook_temp, codebook_ema_decay)
    return VAE(5, embed_dim, None, quantize=quantize, equivaraintconv=dec, prior_net=None, atom_munet=None,
               atom_sigmanet=None, vqdim=vqdim).to(device)


def load_decoder_state(model, state_dict):
    """strict for the decoder side, tolerant of the encoder side and of legacy `dist_filter` keys
    (reference model_module.py:91-108)."""
    own = model.state_dict()
    kept = {k: v for k, v in state_dict.items()
            if not k.startswith(_SKIPPED_PREFIXES) and ".dist_filter." not in k}
    missing = [k for k in own if k not in kept]
    unexpected = [k for k in kept if k not in own]
    if missing or unexpected:
        raise RuntimeError(f"VQ-VAE checkpoint does not match the decoder layout: missing {missing[:5]}, "
                           f"unexpected {unexpected[:5]}")
    model.load_state_dict(kept, strict=True)
    return model


def get_vae_model(modeltype, modelpath=None, device="cpu", modelnum=-1):
    model = build_vae(modeltype, device)
    vqvae_path = modelpath if modelpath is not None else os.path.join("./results/", _DIRS[modeltype])
    if modelnum == -1:
        name = 'model.pt'
    elif modelnum == 999:
        name = 'best_model.pt'
    else:
        name = f'model_{modelnum}.pt'
    ckpt = torch.load(os.path.join(vqvae_path, name), map_location=torch.device('cpu'))
    load_decoder_state(model, ckpt)
    params = load_params(vqvae_path)
    return model, params
