"""Drop-in for the eval-mode quantizers of the reference's utils/vq_module.py.

`build_quantize(quantize_type, codebook_size, embed_dim, codebook_temp, codebook_ema_decay)` keeps the
reference signature (utils/vq_module.py:98-163).  Built types: 'vqvae' / 'vq_3' (the third-party
`vector_quantize_pytorch.VectorQuantize` the N6/K3/K4 models use; its buffers are named as in
release 1.21.7 so `quantize._codebook.*` checkpoint keys load - layout unverified offline, SURVEY.md
§8c) and 'vqema' (the in-repo VectorQuantizerEMA, utils/vq_module.py:40-94).  Only the lookup exists:
nearest code by `|z|^2 + |e|^2 - 2 z.e`, first index on ties, on the GPU (codlad_vq_lookup).  EMA
codebook updates and the FSQ / LFQ / residual variants are training-side or unused and raise.
"""
import torch
import torch.nn as nn

from .. import _lib


def _lookup(z, codebook):
    if not z.is_cuda:
        raise RuntimeError("VQ lookup (codlad_amd) runs on the MI355X only")
    if z.shape[-1] != 3 or codebook.shape[-1] != 3:
        raise NotImplementedError("the HIP VQ lookup is built for 3-dimensional codes (vq dim 3)")
    zs = z.contiguous().float()
    n = zs.numel() // 3
    dev = z.device
    idx = torch.empty(n, dtype=torch.int64, device=dev)
    zq = torch.empty_like(zs)
    zero, one = torch.zeros(3, device=dev), torch.ones(3, device=dev)
    cb = codebook.contiguous().float()
    rc = _lib.lib().codlad_vq_lookup(_lib.ptr(zs), n, _lib.ptr(zero), _lib.ptr(one), _lib.ptr(cb),
                                     cb.shape[0], _lib.ptr(idx), _lib.ptr(zq), None, _lib.stream_ptr(dev))
    _lib.check(rc, "codlad_vq_lookup")
    return zq, idx


class _EuclideanCodebook(nn.Module):
    def __init__(self, codebook_size, dim):
        super().__init__()
        self.register_buffer("initted", torch.tensor([True]))
        self.register_buffer("cluster_size", torch.ones(1, codebook_size))
        self.register_buffer("embed_avg", torch.zeros(1, codebook_size, dim))
        self.register_buffer("embed", torch.zeros(1, codebook_size, dim))


class VectorQuantize(nn.Module):
    """Eval-mode lookup with the buffer layout of vector_quantize_pytorch.VectorQuantize."""

    def __init__(self, dim, codebook_size, decay=0.99, commitment_weight=0.25):
        super().__init__()
        self.dim, self.codebook_size = dim, codebook_size
        self._codebook = _EuclideanCodebook(codebook_size, dim)

    @property
    def codebook(self):
        return self._codebook.embed[0]

    def forward(self, x, mask=None):
        if self.training:
            raise NotImplementedError("codebook EMA updates are training-side; call .eval()")
        zq, idx = _lookup(x, self.codebook)
        zq = zq.view(x.shape)
        idx = idx.view(x.shape[:-1])
        if mask is not None:       # padded positions pass through un-quantized
            zq = torch.where(mask.unsqueeze(-1).bool(), zq, x)
        return zq, idx, torch.zeros(1, device=x.device)


class VectorQuantizerEMA(nn.Module):
    """Eval-mode lookup of the reference's in-repo quantizer (buffer `embeddings`)."""

    def __init__(self, n_e, e_dim, beta, decay, epsilon=1e-5, freeze_codebook=False, sane_index_shape=False):
        super().__init__()
        self.n_e, self.e_dim, self.freeze_codebook = n_e, e_dim, freeze_codebook
        self.register_buffer("embeddings", torch.zeros(n_e, e_dim))
        self.ema_dw = nn.Module()
        self.ema_dw.register_buffer("hidden", torch.zeros(n_e, e_dim))
        self.ema_cluster_size = nn.Module()
        self.ema_cluster_size.register_buffer("hidden", torch.zeros(n_e))

    def forward(self, z, mask=None):
        if self.training and not self.freeze_codebook:
            raise NotImplementedError("codebook EMA updates are training-side; call .eval()")
        zq, idx = _lookup(z, self.embeddings)
        return zq.view(z.shape), idx, None


def build_quantize(quantize_type, codebook_size, embed_dim, codebook_temp, codebook_ema_decay):
    if quantize_type == 'vqema':
        return VectorQuantizerEMA(codebook_size, embed_dim, codebook_temp, codebook_ema_decay)
    if quantize_type == 'vqvae':
        return VectorQuantize(dim=embed_dim, codebook_size=codebook_size, decay=codebook_ema_decay,
                              commitment_weight=codebook_temp)
    if quantize_type == 'vq_3':
        return VectorQuantize(dim=3, codebook_size=codebook_size, decay=codebook_ema_decay,
                              commitment_weight=codebook_temp)
    raise NotImplementedError(f"quantize_type {quantize_type!r}: FSQ / LFQ / residual quantizers are not "
                              "used by the N6/K3/K4 models and are not built")
