"""Drop-in for the decoder half of the reference's models/vae_model.py.

`IC_Decoder`, `IC_Decoder_angle` and `VAE` keep the reference's constructor arguments and parameter
names (checkpoint keys `equivaraintconv.*`, `map_in.*`, `map_out.*`, `quantize.*`; reference
models/vae_model.py:318-373, 414-465, 686-706) and `VAE.latent_decode(latent, mask, batch)`
(vae_model.py:830-838).  The modules hold parameters only; the arithmetic runs in
libcodlad_hip.so (codlad_vq_lookup, codlad_ic_decode).  The e3nn encoder / prior are out of scope:
`get_latent*` raise.
"""
import torch
import torch.nn as nn

from ..engine import Decoder


def _act_linear_act_linear(n_in, n_mid, n_out):
    # reference: nn.Sequential(to_module(act), Linear, to_module(act), Linear) -> Linear keys "1" and "3"
    return nn.Sequential(nn.Identity(), nn.Linear(n_in, n_mid), nn.Identity(), nn.Linear(n_mid, n_out))


class _MessageParams(nn.Module):          # reference gcn_nn.InvariantMessage
    def __init__(self, feat, n_rbf):
        super().__init__()
        self.inv_dense = nn.Sequential(nn.Linear(feat, feat), nn.Linear(feat, feat))
        self.dist_embed = nn.Module()
        self.dist_embed.block = nn.Sequential(nn.Identity(), nn.Linear(n_rbf, feat))


class _ICDecoderBase(nn.Module):
    angle = False

    def __init__(self, n_atom_basis, n_rbf, cutoff, num_conv, activation, cross_flag=True):
        super().__init__()
        if (n_atom_basis, n_rbf, float(cutoff), num_conv, activation) != (36, 15, 21.0, 4, "swish"):
            raise NotImplementedError("the HIP IC decoder is built for n_atom_basis=36, n_rbf=15, "
                                      "cutoff=21, num_conv=4, swish (reference utils/model_module.py:22-26)")
        F = n_atom_basis + 4
        self.res_embed = nn.Embedding(25, 4)
        self.message_blocks = nn.ModuleList([_MessageParams(F, n_rbf) for _ in range(num_conv)])
        self.dense_blocks = nn.ModuleList([_act_linear_act_linear(F, F, F) for _ in range(num_conv)])
        self.backbone_dist = nn.Embedding(25, 3)
        self.sidechain_dist = nn.Embedding(25, 10)
        self.backbone_angle = _act_linear_act_linear(F, 3, 3)
        if self.angle:
            self.sidechain_angle = _act_linear_act_linear(F, 10, 10)
        else:
            self.sidechain_angle = nn.Embedding(25, 10)
        self.backbone_torsion = _act_linear_act_linear(F + 3, 3, 3)
        Ft = F + 10 if self.angle else F
        self.sidechain_torsion_blocks = nn.ModuleList([_act_linear_act_linear(Ft, Ft, Ft) for _ in range(num_conv)])
        self.final_torsion = _act_linear_act_linear(Ft, 10, 10)

    def forward(self, *a, **k):
        raise RuntimeError("IC_Decoder (codlad_amd) is driven through VAE.decoder / latent_decode")


class IC_Decoder(_ICDecoderBase):
    angle = False


class IC_Decoder_angle(_ICDecoderBase):
    angle = True


class VAE(nn.Module):
    """Decoder-side VQ-VAE.  `encoder`, `prior_net`, `atom_munet`, `atom_sigmanet` are accepted for
    signature compatibility and ignored."""

    def __init__(self, n_cgs, embed_dim, encoder, quantize=None, equivaraintconv=None, prior_net=None,
                 atom_munet=None, atom_sigmanet=None, vqdim=None):
        super().__init__()
        self.equivaraintconv = equivaraintconv
        self.quantize = quantize
        self.n_cgs, self.embed_dim, self.vqdim = n_cgs, embed_dim, vqdim
        if self.embed_dim != self.vqdim and self.quantize is not None:
            self.map_in = nn.Linear(embed_dim, vqdim)
            self.map_out = nn.Linear(vqdim, embed_dim)
        self._engine = None
        self._engine_key = None

    def engine(self):
        key = tuple((p.data_ptr(), p._version, str(p.device)) for p in list(self.parameters()) + list(self.buffers()))
        if self._engine is None or key != self._engine_key:
            dev = next(self.parameters()).device
            sd = {k: v for k, v in self.state_dict().items()}
            self._engine = Decoder(sd, dev)
            self._engine_key = key
        return self._engine

    def _no_encoder(self, *a, **k):
        raise NotImplementedError("the e3nn encoder / prior are outside the built hot path "
                                  "(SURVEY.md §8f item 1); latents come from the sampler")

    get_latent_cg = get_latent = get_latent_wovq = encode = forward = _no_encoder

    @staticmethod
    def _flatten(latent, num_CGs):
        lens = num_CGs.tolist()
        if len(set(lens)) == 1 and latent.shape[1] == lens[0]:
            return latent.reshape(-1, latent.shape[-1])
        return torch.cat([latent[b, :n] for b, n in enumerate(lens)], dim=0)   # gcn_nn.restore_shape

    def decoder(self, cg_z, cg_xyz, CG_nbr_list, mapping, S_I, num_CGs, mask=None):
        """S_I = quantized latents [B,L,vqdim] -> (None, ic_recon [sum L,13,3])."""
        flat = self._flatten(S_I, num_CGs)
        ic = self.engine().ic_decode(flat, cg_z, cg_xyz, CG_nbr_list)
        return None, ic

    def latent_decode(self, latent, mask, batch):
        """latent [B,L,3] de-normalised samples -> (ic, ic_recon) like the reference; `ic` is the
        batch's ground-truth internal coordinates when present, else None."""
        if not latent.is_cuda:
            raise RuntimeError("VAE.latent_decode (codlad_amd) runs on the MI355X only")
        cg_xyz = batch['CG_nxyz'][:, 1:]
        cg_z = batch['CG_nxyz'][:, 0].long()
        num_CGs = batch['num_CGs']
        if self.quantize is not None:
            latent, indices, _ = self.quantize(latent, mask=mask)
        _, ic_recon = self.decoder(cg_z, cg_xyz, batch['CG_nbr_list'], batch.get('CG_mapping'), latent, num_CGs)
        return batch.get('ic'), ic_recon
