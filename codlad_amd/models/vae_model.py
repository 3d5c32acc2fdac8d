"""Drop-in for the reference's models/vae_model.py (inference side).

`IC_Decoder`, `IC_Decoder_angle`, `e3nnEncoder`, `e3nnPrior`, `VAE` and `GenZProt` keep the reference's constructor
arguments and parameter names (checkpoint keys `encoder.*`, `prior_net.*`, `equivaraintconv.*`, `map_in.*`,
`map_out.*`, `quantize.*`, `atom_munet.*`, `atom_sigmanet.*`; reference models/vae_model.py:21-311, 318-373, 414-465,
509-530, 686-706) and the methods the sampling script calls: `latent_decode(latent, mask, batch)` (:830-838),
`get_latent_wovq(batch)` (:798-828, `--experiment recon`) and `get_latent_cg(batch)` (:776-787 / :643-652).  The modules
hold parameters only; the arithmetic runs in libcodlad_hip.so (codlad_vq_lookup, codlad_ic_decode, codlad_tp_conv, ...).
The encoder / prior restate e3nn's tensor product (e3nn itself is absent: "parity unpinned", oracle/e3nn_lite.py).
"""
import torch
import torch.nn as nn

from ..encoder import Encoder as _EncoderEngine
from ..encoder import Prior as _PriorEngine
from ..engine import Decoder

TP_WEIGHT_NUMEL = (192, 288, 384)      # o3.FullyConnectedTensorProduct(...).weight_numel of the three conv depths


class _TPConvParams(nn.Module):            # reference gcn_nn.TensorProductConvLayer (parameters only)
    def __init__(self, depth, n_edge_features=36, hidden_features=36, dropout=0.0):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(n_edge_features, hidden_features), nn.ReLU(), nn.Dropout(dropout),
                                nn.Linear(hidden_features, TP_WEIGHT_NUMEL[depth]))


def _edge_embedding(n_in, ns, dropout):
    return nn.Sequential(nn.Linear(n_in, ns), nn.ReLU(), nn.Dropout(dropout), nn.Linear(ns, ns))


class _Smearing(nn.Module):                # reference gcn_nn.GaussianSmearing: the `offset` buffer of the checkpoints
    def __init__(self, start, stop, n):
        super().__init__()
        self.register_buffer("offset", torch.linspace(start, stop, n))


def _check_e3nn_args(sh_lmax, ns, nv, num_conv_layers, distance_embed_dim, use_second_order_repr, batch_norm,
                     in_edge_features):
    if (sh_lmax, ns, nv, num_conv_layers, distance_embed_dim, use_second_order_repr, batch_norm, in_edge_features) != \
            (2, 12, 4, 3, 8, False, False, 4):
        raise NotImplementedError("the HIP encoder / prior are built for the configuration utils/model_module.py uses: "
                                  "sh_lmax=2, ns=12, nv=4, 3 conv layers, 8 distance Gaussians, first-order irreps")


class e3nnEncoder(nn.Module):
    """Parameters of the reference's e3nnEncoder (vae_model.py:21-107); forward = codlad_amd.encoder.Encoder."""

    def __init__(self, device, n_atom_basis, n_cgs=None, in_edge_features=4, cross_max_distance=30, sh_lmax=2, ns=12,
                 nv=4, num_conv_layers=3, atom_max_radius=12, cg_max_radius=30, distance_embed_dim=8,
                 cross_distance_embed_dim=8, use_second_order_repr=False, batch_norm=False, dropout=0.0,
                 lm_embedding_type=None):
        super().__init__()
        _check_e3nn_args(sh_lmax, ns, nv, num_conv_layers, distance_embed_dim, use_second_order_repr, batch_norm,
                         in_edge_features)
        if n_atom_basis != 36 or cross_distance_embed_dim != 8:
            raise NotImplementedError("n_atom_basis 36 and 8 cross-distance Gaussians only")
        self.radii = (float(atom_max_radius), float(cg_max_radius), float(cross_max_distance))
        self.atom_node_embedding = nn.Embedding(30, ns, padding_idx=0)
        self.atom_edge_embedding = _edge_embedding(2 + in_edge_features + distance_embed_dim, ns, dropout)
        self.cg_node_embedding = nn.Embedding(30, ns, padding_idx=0)
        self.cg_edge_embedding = _edge_embedding(2 + in_edge_features + distance_embed_dim, ns, dropout)
        self.cross_edge_embedding = _edge_embedding(cross_distance_embed_dim, ns, dropout)
        self.atom_distance_expansion = _Smearing(0.0, atom_max_radius, distance_embed_dim)
        self.cg_distance_expansion = _Smearing(0.0, cg_max_radius, distance_embed_dim)
        self.cross_distance_expansion = _Smearing(0.0, cross_max_distance, cross_distance_embed_dim)
        for name in ("atom_conv_layers", "cg_conv_layers", "cg_to_atom_conv_layers", "atom_to_cg_conv_layers"):
            setattr(self, name, nn.ModuleList([_TPConvParams(i, dropout=dropout) for i in range(num_conv_layers)]))
        self.dense = nn.Sequential(nn.Linear(84, n_atom_basis), nn.Tanh(), nn.Linear(n_atom_basis, n_atom_basis))
        self._engine, self._engine_key = None, None

    def engine(self):
        key = tuple((p.data_ptr(), p._version, str(p.device)) for p in self.parameters())
        if self._engine is None or key != self._engine_key:
            self._engine = _EncoderEngine(self.state_dict(), next(self.parameters()).device, *self.radii)
            self._engine_key = key
        return self._engine

    def forward(self, z, xyz, cg_z, cg_xyz, mapping, nbr_list, cg_nbr_list, num_CGs=None, num_atoms=None):
        return self.engine().forward(z, xyz, cg_z, cg_xyz, mapping, nbr_list, cg_nbr_list), None


class e3nnPrior(nn.Module):
    """Parameters of the reference's e3nnPrior (vae_model.py:204-243); forward = codlad_amd.encoder.Prior."""

    def __init__(self, device, n_atom_basis, n_cgs=None, in_edge_features=4, sh_lmax=2, ns=12, nv=4, num_conv_layers=3,
                 cg_max_radius=30, distance_embed_dim=8, use_second_order_repr=False, batch_norm=False, dropout=0.0,
                 lm_embedding_type=None):
        super().__init__()
        _check_e3nn_args(sh_lmax, ns, nv, num_conv_layers, distance_embed_dim, use_second_order_repr, batch_norm,
                         in_edge_features)
        if n_atom_basis != 36:
            raise NotImplementedError("n_atom_basis 36 only")
        self.cg_max_radius = float(cg_max_radius)
        self.cg_node_embedding = nn.Embedding(30, ns, padding_idx=0)
        self.cg_edge_embedding = _edge_embedding(2 + in_edge_features + distance_embed_dim, ns, dropout)
        self.cg_distance_expansion = _Smearing(0.0, cg_max_radius, distance_embed_dim)
        self.cg_conv_layers = nn.ModuleList([_TPConvParams(i, dropout=dropout) for i in range(num_conv_layers)])
        self.mu = nn.Sequential(nn.Linear(48, n_atom_basis), nn.Tanh(), nn.Linear(n_atom_basis, n_atom_basis))
        self.sigma = nn.Sequential(nn.Linear(48, n_atom_basis), nn.Tanh(), nn.Linear(n_atom_basis, n_atom_basis))
        self._engine, self._engine_key = None, None

    def engine(self):
        key = tuple((p.data_ptr(), p._version, str(p.device)) for p in self.parameters())
        if self._engine is None or key != self._engine_key:
            self._engine = _PriorEngine(self.state_dict(), next(self.parameters()).device, self.cg_max_radius)
            self._engine_key = key
        return self._engine

    def forward(self, cg_z, cg_xyz, cg_nbr_list):
        return self.engine().forward(cg_z, cg_xyz, cg_nbr_list)


_LENS_CACHE = {}


def batch_lengths(num_CGs):
    """`num_CGs.tolist()`, read from the device ONCE per tensor: the reference's helpers need the lengths on the host
    (gcn_nn.py:35-43, 45-52) and ask the device for them at every call - two synchronisations per encoder pass, 2 of the 5 ms
    of a 40-frame batch.  The copy is kept per (storage, version) together with the tensor itself, so a batch that is
    encoded, sampled and decoded pays one read."""
    if not num_CGs.is_cuda:
        return num_CGs.tolist()
    key = (num_CGs.data_ptr(), num_CGs._version, int(num_CGs.numel()), str(num_CGs.device))
    hit = _LENS_CACHE.get(key)
    if hit is None:
        if len(_LENS_CACHE) >= 16:
            _LENS_CACHE.clear()
        # the entry keeps the tensor alive: its address cannot be handed to another batch's lengths while the key exists
        hit = _LENS_CACHE[key] = (num_CGs.tolist(), num_CGs)
    return hit[0]


def reshape_and_create_mask(h, num_CGs):
    """reference gcn_nn.py:35-43"""
    lens = batch_lengths(num_CGs)
    reshaped = torch.nn.utils.rnn.pad_sequence(torch.split(h, lens, dim=0), batch_first=True)
    mask = torch.arange(max(lens), device=h.device)[None, :] < num_CGs[:, None]
    return reshaped, mask


def reparametrize(mu, sigma, generator=None):
    """reference utils/train_module.py:22-25 (eps = randn_like(sigma)); `generator` makes the draw reproducible."""
    eps = torch.randn(sigma.shape, device=sigma.device, dtype=sigma.dtype, generator=generator)
    return eps * sigma + mu


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


def _batch_inputs(batch):
    """get_inputs (reference vae_model.py:708-728): the all-atom side is present when the batch carries `nxyz`."""
    if "nxyz" in batch:
        xyz, z, nbr_list = batch["nxyz"][:, 1:], batch["nxyz"][:, 0], batch["nbr_list"]
    else:
        xyz = z = nbr_list = None
    return (z, batch["CG_nxyz"][:, 0].long(), xyz, batch["CG_nxyz"][:, 1:], nbr_list, batch["CG_nbr_list"],
            batch.get("CG_mapping"), batch["num_CGs"])


def _linear_rows(engine, x, linear):
    """y = x W^T + b through codlad_mlp_rows (no torch arithmetic on the product path)."""
    import ctypes as C
    from .. import _lib
    y = torch.empty(x.shape[0], linear.weight.shape[0], dtype=torch.float32, device=x.device)
    rc = engine.lib.codlad_mlp_rows(_lib.ptr(x.contiguous()), x.shape[0], x.shape[1], None, None, 0,
                                    _lib.ptr(linear.weight.detach().contiguous()), _lib.ptr(linear.bias.detach().contiguous()),
                                    int(linear.weight.shape[0]), 0, 0, _lib.ptr(y), _lib.stream_ptr(x.device))
    _lib.check(rc, "codlad_mlp_rows")
    return y


class VAE(nn.Module):
    """The VQ-VAE as `utils.model_module.get_vae_model` builds it for N6 / K3 / K4: e3nn encoder (optional: a decoder-only
    model passes encoder=None), map_in / map_out, quantizer, IC decoder."""

    def __init__(self, n_cgs, embed_dim, encoder, quantize=None, equivaraintconv=None, prior_net=None,
                 atom_munet=None, atom_sigmanet=None, vqdim=None):
        super().__init__()
        self.encoder = encoder
        self.prior_net = prior_net
        self.atom_munet, self.atom_sigmanet = atom_munet, atom_sigmanet
        self.equivaraintconv = equivaraintconv
        self.quantize = quantize
        self.n_cgs, self.embed_dim, self.vqdim = n_cgs, embed_dim, vqdim
        if self.embed_dim != self.vqdim and self.quantize is not None:
            self.map_in = nn.Linear(embed_dim, vqdim)
            self.map_out = nn.Linear(vqdim, embed_dim)
        self._engine = None
        self._engine_key = None

    def engine(self):
        skip = ("encoder.", "prior_net.", "atom_munet.", "atom_sigmanet.")
        own = [(k, v) for k, v in list(self.named_parameters()) + list(self.named_buffers()) if not k.startswith(skip)]
        key = tuple((v.data_ptr(), v._version, str(v.device)) for _k, v in own)
        if self._engine is None or key != self._engine_key:
            dev = own[0][1].device
            self._engine = Decoder({k: v for k, v in self.state_dict().items() if not k.startswith(skip)}, dev)
            self._engine_key = key
        return self._engine

    def _encode_wovq(self, batch):
        """encoder -> map_in: the un-quantized latent per bead [sum L, vqdim] (reference vae_model.py:803-808)."""
        if self.encoder is None:
            raise NotImplementedError("this VAE was built without an encoder (decoder-only checkpoint)")
        z, cg_z, xyz, cg_xyz, nbr_list, cg_nbr, mapping, num_CGs = _batch_inputs(batch)
        if z is None or mapping is None:
            raise ValueError("the encoder needs the all-atom side of the batch: nxyz, nbr_list, CG_mapping")
        h, _ = self.encoder(z, xyz, cg_z, cg_xyz, mapping, nbr_list, cg_nbr)
        if self.embed_dim != self.vqdim and self.quantize is not None:
            eng = self.encoder.engine()
            h = _linear_rows(eng, h, self.map_in)
        return h, num_CGs

    def get_latent_wovq(self, batch):
        """`--experiment recon` (reference test.py:501): -> (latent [B, L, vqdim], None, None, mask, num_CGs, None, None)."""
        h, num_CGs = self._encode_wovq(batch)
        reshape_h, mask = reshape_and_create_mask(h, num_CGs)
        return reshape_h, None, None, mask, num_CGs, None, None

    def get_latent(self, batch):
        """encoder -> map_in -> quantizer (reference vae_model.py:789-796)."""
        h, num_CGs = self._encode_wovq(batch)
        reshape_h, mask = reshape_and_create_mask(h, num_CGs)
        reshape_h, indices, emb_loss = self.quantize(reshape_h, mask=mask)
        return reshape_h, indices, emb_loss, mask, num_CGs, None, None

    def get_latent_cg(self, batch, generator=None):
        """reference vae_model.py:776-787: the model's `encoder` called as a CG prior (cgvae variant)."""
        if not isinstance(self.encoder, e3nnPrior):
            raise NotImplementedError("get_latent_cg needs a VAE whose encoder is an e3nnPrior (the cgvae variant); "
                                      "the conditional prior of the sampling script is GenZProt.get_latent_cg")
        _z, cg_z, _xyz, cg_xyz, _nbr, cg_nbr, _mapping, num_CGs = _batch_inputs(batch)
        mu, sigma = self.encoder(cg_z, cg_xyz, cg_nbr)
        reshape_h, mask = reshape_and_create_mask(reparametrize(mu, sigma, generator), num_CGs)
        return reshape_h, None, None, mask, num_CGs, mu, sigma

    def forward(self, *a, **k):
        raise NotImplementedError("training-time forward (losses) is not on the sampling path")

    encode = forward

    @staticmethod
    def _flatten(latent, num_CGs):
        lens = batch_lengths(num_CGs)
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


class GenZProt(nn.Module):
    """The conditional VAE `get_vae_model("C2")` builds (reference models/vae_model.py:509-530): e3nn encoder, CG prior,
    posterior heads and an IC decoder that takes the 36-dimensional latent as it is (no quantizer, no map_out).  The
    sampling script uses `get_latent_cg` (the prior's reparametrised sample, test.py:495) and, for `--experiment
    genzprot`, `latent_decode`."""

    def __init__(self, encoder, equivaraintconv, atom_munet, atom_sigmanet, n_cgs, feature_dim, prior_net=None,
                 det=False, equivariant=True, offset=True):
        super().__init__()
        if not equivariant:
            raise NotImplementedError("the non-equivariant (euclidean head) variant is not used by get_vae_model")
        self.encoder, self.equivaraintconv = encoder, equivaraintconv
        self.atom_munet, self.atom_sigmanet = atom_munet, atom_sigmanet
        self.n_cgs, self.prior_net, self.det, self.offset, self.equivariant = n_cgs, prior_net, det, offset, equivariant
        self.quantize = None
        self._engine, self._engine_key = None, None

    def engine(self):
        own = [(k, v) for k, v in self.named_parameters() if k.startswith("equivaraintconv.")]
        key = tuple((v.data_ptr(), v._version, str(v.device)) for _k, v in own)
        if self._engine is None or key != self._engine_key:
            self._engine = Decoder({k: v for k, v in self.state_dict().items() if k.startswith("equivaraintconv.")},
                                   own[0][1].device)
            self._engine_key = key
        return self._engine

    def get_latent_cg(self, batch, generator=None):
        """-> (y [B, L, 36], None, None, mask, num_CGs, H_prior_mu, H_prior_sigma)  (vae_model.py:643-652)"""
        _z, cg_z, _xyz, cg_xyz, _nbr, cg_nbr, _mapping, num_CGs = _batch_inputs(batch)
        mu, sigma = self.prior_net(cg_z, cg_xyz, cg_nbr)
        reshape_h, mask = reshape_and_create_mask(reparametrize(mu, sigma, generator), num_CGs)
        return reshape_h, None, None, mask, num_CGs, mu, sigma

    def _posterior(self, batch, generator):
        z, cg_z, xyz, cg_xyz, nbr_list, cg_nbr, mapping, num_CGs = _batch_inputs(batch)
        if z is None or mapping is None:
            raise ValueError("the encoder needs the all-atom side of the batch: nxyz, nbr_list, CG_mapping")
        h, _ = self.encoder(z, xyz, cg_z, cg_xyz, mapping, nbr_list, cg_nbr)
        eng = self.encoder.engine()
        mu = _mlp_rows(eng, h, self.atom_munet, "relu")
        sigma = _mlp_rows(eng, h, self.atom_sigmanet, "relu", mode=1) + (1e-12 - 1e-9)      # 1e-12 + exp(logvar / 2)
        return reshape_and_create_mask(reparametrize(mu, sigma, generator), num_CGs) + (num_CGs, mu, sigma)

    def get_latent(self, batch, generator=None):
        """vae_model.py:617-628"""
        reshape_h, mask, num_CGs, mu, sigma = self._posterior(batch, generator)
        return reshape_h, None, None, mask, num_CGs, mu, sigma

    def get_latent_wovq(self, batch, generator=None):
        """vae_model.py:630-641"""
        reshape_h, mask, num_CGs, _mu, _sigma = self._posterior(batch, generator)
        return reshape_h, None, None, mask, num_CGs, None, None

    def latent_decode(self, latent, mask, batch):
        """latent [B, L, 36] -> (ic, ic_recon [sum L, 13, 3])  (vae_model.py:668-677)"""
        if not latent.is_cuda:
            raise RuntimeError("GenZProt.latent_decode (codlad_amd) runs on the MI355X only")
        flat = VAE._flatten(latent, batch["num_CGs"])
        ic = self.engine().ic_decode(flat, batch["CG_nxyz"][:, 0].long(), batch["CG_nxyz"][:, 1:], batch["CG_nbr_list"])
        return batch.get("ic"), ic

    def forward(self, *a, **k):
        raise NotImplementedError("training-time forward (losses) is not on the sampling path")


def _mlp_rows(engine, x, seq, act, mode=0):
    """seq = Sequential(Linear, act, Linear) applied per row through codlad_mlp_rows."""
    from .. import _lib
    l1, l2 = seq[0], seq[2]
    y = torch.empty(x.shape[0], l2.weight.shape[0], dtype=torch.float32, device=x.device)
    rc = engine.lib.codlad_mlp_rows(_lib.ptr(x.contiguous()), x.shape[0], x.shape[1], _lib.ptr(l1.weight.detach().contiguous()),
                                    _lib.ptr(l1.bias.detach().contiguous()), int(l1.weight.shape[0]),
                                    _lib.ptr(l2.weight.detach().contiguous()), _lib.ptr(l2.bias.detach().contiguous()),
                                    int(l2.weight.shape[0]), {"tanh": 0, "relu": 1}[act], mode, _lib.ptr(y),
                                    _lib.stream_ptr(x.device))
    _lib.check(rc, "codlad_mlp_rows")
    return y
