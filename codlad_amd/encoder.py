"""Host side of the e3nn encoder / CG prior (reference models/vae_model.py:21-311): graph tables,
weights and the calls into libcodlad_hip.so.  All arithmetic is in the library (codlad_tp_conv, codlad_mlp_rows,
codlad_bead_mean, codlad_embed_rows); torch is used for device memory and the index bookkeeping of the graphs (sorting
edges by receiver), which the reference does on the host as well (make_directed, scatter).

e3nn is not part of the reference tree and not installed: the tensor product is a restatement of e3nn 0.5.1's published
definition ("parity unpinned", oracle/e3nn_lite.py), pinned only as far as reference-held data allows - the Wigner
symbols against the buffers e3nn left in the shipped C2 checkpoint, the whole prior statistically against the
reference's dataset statistics of that checkpoint's latent (tests/test_e3nn_encoder.py).
"""
import ctypes as C
from collections import OrderedDict

import torch

from . import _lib
from .weights import Blob

NS = 12


def _width(depth):
    return 12 * (depth + 1)


def receiver_csr(pairs, n, mode=0):
    """Pair list int64 [E, 2] on the GPU -> (ptr int32 [n + 1], snd int32 [<= 2 E]) of the receivers, one
    `codlad_receiver_csr` call, no host synchronisation.  mode 0: the reference's make_directed (gcn_nn.py:54-64: edge
    (a, b) = receiver a, sender b, reversed edges added unless the list already runs both ways); mode 1: directed as
    given.  Senders ascending inside a receiver."""
    pairs = pairs.to(torch.int64).contiguous()
    E = int(pairs.shape[0])
    dev = pairs.device
    ptr = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    snd = torch.zeros(max(2 * E, 1), dtype=torch.int32, device=dev)
    if E == 0:
        return ptr, snd
    work = torch.empty(2 * n + 2 + 4 * E, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().codlad_receiver_csr(_lib.ptr(pairs), E, n, mode, _lib.ptr(ptr), _lib.ptr(snd), _lib.ptr(work),
                                             _lib.stream_ptr(dev)), "codlad_receiver_csr")
    return ptr, snd


def directed_csr(pairs, n):
    return receiver_csr(pairs, n, 0)


def csr_by_receiver(recv, snd, n_recv):
    """Edges (recv[e] <- snd[e]) -> receivers' CSR."""
    return receiver_csr(torch.stack([recv.to(torch.int64), snd.to(torch.int64)], dim=1), n_recv, 1)


class ConvWeights:
    """The tensors of an e3nnEncoder / e3nnPrior state dict (prefix stripped) in one device blob."""

    def __init__(self, state_dict, device):
        t = OrderedDict()
        for k, v in state_dict.items():
            if ".tp." in k or k.endswith(".offset") or not v.is_floating_point():
                continue          # e3nn's own buffers (output mask, Wigner constants), GaussianSmearing offsets
            t[k] = v.detach().float().cpu().contiguous()
        self.blob = Blob(t, device)

    def p(self, name):
        ptr = self.blob.ptr(name)
        if ptr is None:
            raise KeyError(f"encoder weights: {name} missing from the state dict")
        return ptr

    def has(self, name):
        return name in self.blob.offsets


class _Stack:
    """Shared by encoder and prior: one TensorProductConvLayer launch."""

    def __init__(self, state_dict, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("codlad_amd runs on an MI355X only; no CPU path exists")
        self.lib = _lib.lib()
        assert C.sizeof(_lib.TpConvArgs) == self.lib.codlad_tp_conv_args_size()
        self.w = ConvWeights(state_dict, self.device)
        self._images = {}          # (layer, depth, embedding, emb_in) -> the layer's packed weight image (codlad_tp_conv_pack)
        self.pack_weights = True   # False: every workgroup of every launch packs the image itself (same bits, slower)

    def conv(self, layer, depth, csr, xyz_recv, xyz_snd, typ_recv, typ_snd, r_sign, smear_stop, emb, emb_in, h_recv,
             h_snd, recv_first, out, accumulate, group):
        ptr, snd = csr
        a = _lib.TpConvArgs()
        a.ptr, a.snd, a.n_recv = _lib.ptr(ptr), _lib.ptr(snd), int(ptr.numel() - 1)
        a.xyz_recv, a.xyz_snd = _lib.ptr(xyz_recv), _lib.ptr(xyz_snd)
        a.typ_recv, a.typ_snd = _lib.ptr(typ_recv), _lib.ptr(typ_snd)
        a.r_sign, a.smear_stop = float(r_sign), float(smear_stop)
        a.emb0_w, a.emb0_b = self.w.p(emb + ".0.weight"), self.w.p(emb + ".0.bias")
        a.emb3_w, a.emb3_b = self.w.p(emb + ".3.weight"), self.w.p(emb + ".3.bias")
        a.emb_in = emb_in
        a.h_recv, a.d_recv = _lib.ptr(h_recv), int(h_recv.shape[1])
        a.h_snd, a.d_snd = _lib.ptr(h_snd), int(h_snd.shape[1])
        assert a.d_snd == _width(depth) and out.shape == (a.n_recv, _width(depth + 1)) and out.is_contiguous()
        a.attr_recv_first = int(recv_first)
        a.fc0_w, a.fc0_b = self.w.p(layer + ".fc.0.weight"), self.w.p(layer + ".fc.0.bias")
        a.fc3_w, a.fc3_b = self.w.p(layer + ".fc.3.weight"), self.w.p(layer + ".fc.3.bias")
        a.depth, a.out, a.accumulate, a.group = depth, _lib.ptr(out), int(accumulate), group
        key = (layer, depth, emb, emb_in)
        if self.pack_weights and key not in self._images:                       # once per layer: the weights as the matrix-pipe kernel holds them in LDS
            img = torch.empty(self.lib.codlad_tp_conv_image_bytes(depth), dtype=torch.uint8, device=self.device)
            _lib.check(self.lib.codlad_tp_conv_pack(C.byref(a), _lib.ptr(img), _lib.stream_ptr(self.device)), "codlad_tp_conv_pack")
            self._images[key] = img
        a.packed = _lib.ptr(self._images[key]) if self.pack_weights else None
        _lib.check(self.lib.codlad_tp_conv(C.byref(a), _lib.stream_ptr(self.device)), "codlad_tp_conv")

    def embed(self, name, idx):
        tab = self.w.blob.view(name)
        out = torch.empty(idx.numel(), tab.shape[1], dtype=torch.float32, device=self.device)
        _lib.check(self.lib.codlad_embed_rows(_lib.ptr(tab), _lib.ptr(idx.to(torch.int32).contiguous()), idx.numel(),
                                              tab.shape[1], _lib.ptr(out), _lib.stream_ptr(self.device)),
                   "codlad_embed_rows")
        return out

    def mlp(self, x, first, second, act, mode=0):
        """second(act(first(x))) per row; first None: a single Linear."""
        n, in_dim = x.shape
        w2, b2 = self.w.blob.view(second + ".weight"), self.w.blob.view(second + ".bias")
        hidden = 0 if first is None else int(self.w.blob.view(first + ".bias").numel())
        y = torch.empty(n, w2.shape[0], dtype=torch.float32, device=self.device)
        rc = self.lib.codlad_mlp_rows(_lib.ptr(x.contiguous()), n, in_dim,
                                      None if first is None else self.w.p(first + ".weight"),
                                      None if first is None else self.w.p(first + ".bias"), hidden,
                                      _lib.ptr(w2), _lib.ptr(b2), int(w2.shape[0]), {"tanh": 0, "relu": 1}[act], mode,
                                      _lib.ptr(y), _lib.stream_ptr(self.device))
        _lib.check(rc, "codlad_mlp_rows")
        return y

    def _f(self, t):
        return t.to(self.device, torch.float32).contiguous()


class Prior(_Stack):
    """e3nnPrior.forward (reference vae_model.py:245-266): CG beads only -> (H_mu, H_sigma) [n_cg, 36]."""

    def __init__(self, state_dict, device, cg_max_radius=26.0):
        super().__init__(state_dict, device)
        self.cg_max_radius = cg_max_radius

    def forward(self, cg_z, cg_xyz, cg_nbr_list):
        n = int(cg_z.numel())
        xyz, typ = self._f(cg_xyz), self._f(cg_z)
        csr = directed_csr(cg_nbr_list.to(self.device), n)
        h = self.embed("cg_node_embedding.weight", cg_z.to(self.device).long())
        for l in range(3):
            out = torch.empty(n, _width(l + 1), dtype=torch.float32, device=self.device)
            self.conv(f"cg_conv_layers.{l}", l, csr, xyz, xyz, typ, typ, 1.0, self.cg_max_radius, "cg_edge_embedding", 14,
                      h, h, True, out, False, 64)
            h = out
        mu = self.mlp(h, "mu.0", "mu.2", "tanh")
        sigma = self.mlp(h, "sigma.0", "sigma.2", "tanh", mode=1)
        return mu, sigma


class Encoder(_Stack):
    """e3nnEncoder.forward (reference vae_model.py:109-162): atoms + beads + the atom <-> bead cross graph -> [n_cg, 36]."""

    def __init__(self, state_dict, device, atom_max_radius=14.0, cg_max_radius=26.0, cross_max_distance=26.0):
        super().__init__(state_dict, device)
        self.radii = (atom_max_radius, cg_max_radius, cross_max_distance)

    def forward(self, z, xyz, cg_z, cg_xyz, mapping, nbr_list, cg_nbr_list):
        dev = self.device
        na, nc = int(z.numel()), int(cg_z.numel())
        xa, xc, ta, tc = self._f(xyz), self._f(cg_xyz), self._f(z), self._f(cg_z)
        mapping = mapping.to(dev).long()
        csr_a = directed_csr(nbr_list.to(dev), na)
        csr_c = directed_csr(cg_nbr_list.to(dev), nc)
        # cross graph (vae_model.py:196-201): every atom <-> its bead
        csr_c2a = (torch.arange(na + 1, dtype=torch.int32, device=dev), mapping.to(torch.int32).contiguous())
        csr_a2c = csr_by_receiver(mapping, torch.arange(na, device=dev), nc)
        ha = self.embed("atom_node_embedding.weight", z.to(dev).long())
        hc = self.embed("cg_node_embedding.weight", cg_z.to(dev).long())
        ra, rc, rx = self.radii
        for l in range(3):
            ha_new = torch.empty(na, _width(l + 1), dtype=torch.float32, device=dev)
            self.conv(f"atom_conv_layers.{l}", l, csr_a, xa, xa, ta, ta, 1.0, ra, "atom_edge_embedding", 14, ha, ha, True,
                      ha_new, False, 64)
            # bead -> atom: r = x_atom - x_bead = receiver - sender
            self.conv(f"cg_to_atom_conv_layers.{l}", l, csr_c2a, xa, xc, None, None, -1.0, rx, "cross_edge_embedding", 8,
                      ha, hc, True, ha_new, True, 1)
            if l != 2:
                hc_new = torch.empty(nc, _width(l + 1), dtype=torch.float32, device=dev)
                self.conv(f"cg_conv_layers.{l}", l, csr_c, xc, xc, tc, tc, 1.0, rc, "cg_edge_embedding", 14, hc, hc, True,
                          hc_new, False, 64)
                # atom -> bead: the reference hands both cross directions the SAME edge attributes [e | atom | bead]
                self.conv(f"atom_to_cg_conv_layers.{l}", l, csr_a2c, xc, xa, None, None, 1.0, rx, "cross_edge_embedding", 8,
                          hc, ha, False, hc_new, True, 16)
                hc = hc_new
            ha = ha_new
        node = torch.empty(nc, 84, dtype=torch.float32, device=dev)
        _lib.check(self.lib.codlad_bead_mean(_lib.ptr(ha), _lib.ptr(hc), _lib.ptr(csr_a2c[0]), _lib.ptr(csr_a2c[1]), nc,
                                             _lib.ptr(node), _lib.stream_ptr(dev)), "codlad_bead_mean")
        return self.mlp(node, "dense.0", "dense.2", "tanh")
