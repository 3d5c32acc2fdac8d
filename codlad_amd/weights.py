"""Checkpoint tensors -> one device blob + the pointer structs of include/codlad_hip.h.

Inputs are state_dicts with the reference's key layout:
  denoiser: models/latent_model.py:119-148 (108 tensors), optional `module.` prefix;
  VQ-VAE  : utils/model_module.py:39-75 -> equivaraintconv.*, map_out.*, quantize.*.

The blob layout depends on tensor shapes only, so every rank of a multi-GPU job derives the
same offsets and rank 0 can broadcast the blob as one buffer over RCCL.
"""
import ctypes as C
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib

H = 128
_ALIGN = 64  # floats (256 bytes)


def _pack_index():
    """Row/col of the source 128x128 block for every element of the MFMA-operand order
    (must equal codlad_pack_block_host)."""
    b, r, lane, bo = np.meshgrid(np.arange(4), np.arange(16), np.arange(64), np.arange(4), indexing="ij")
    rows = 32 * bo + (lane & 31)
    cols = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    return torch.from_numpy(rows.reshape(-1)), torch.from_numpy(cols.reshape(-1))


_ROWS, _COLS = _pack_index()


def pack_block(w, scale=1.0):
    """[128,128] weight block (out, in) -> 16384 floats in MFMA A-operand order."""
    assert tuple(w.shape) == (H, H)
    out = w.detach().float().cpu()[_ROWS, _COLS]
    return out * scale if scale != 1.0 else out.clone()


def _pack_index_h():
    """(row, col) of the source block for every element of the split-fp16 (f16x3 / f16x4) order
    [k-step][out block][split][lane][8 halves] (split handled by the caller)."""
    ks, bo, lane, j = np.meshgrid(np.arange(8), np.arange(4), np.arange(64), np.arange(8), indexing="ij")
    b, s_, h = ks >> 1, ks & 1, lane >> 5
    rows = 32 * bo + (lane & 31)
    cols = 32 * b + 16 * s_ + 8 * (j >> 2) + 4 * h + (j & 3)
    return torch.from_numpy(rows), torch.from_numpy(cols)          # [8,4,64,8]


_ROWS_H, _COLS_H = _pack_index_h()


FP16_MAX = 65504.0
MAX_CHAIN_EXP = 16          # CODLAD_MAX_CHAIN_EXP of csrc/common.h: |accumulated exponent| at a GELU input
MAX_RANGE_LOSS = 6          # a matrix whose largest weight forces its exponent this far below the wanted one is refused


def pack_block_h(w, scale=1.0):
    """[128,128] block -> hi/lo fp16 split in MFMA f16 operand order, returned as 16384 float32
    words (bit container for 32768 halves = 64 KB).  hi + lo reproduces w to max(2^-22 |w|, 2^-25): below
    |w| ~ 2^-3 the `lo` half is a subnormal fp16 and the error is an absolute 2^-25 (which is why the blocks
    of a model are packed with block exponents, `block_exponent`).  A weight beyond the fp16 range has no
    such split (hi would be inf): ValueError."""
    assert tuple(w.shape) == (H, H)
    g = (w.detach().float().cpu() * scale)[_ROWS_H, _COLS_H]           # [ks, bo, lane, 8] fp32
    if not bool(torch.isfinite(g).all()) or float(g.abs().max()) > FP16_MAX:
        raise ValueError(f"weight block with max |w| = {float(g.abs().max()):.4g} is outside the fp16 range "
                         f"({FP16_MAX:.0f}) the split-fp16 contraction modes need; use precision='f32'")
    hi = g.to(torch.float16)                                           # round to nearest even
    lo = (g - hi.float()).to(torch.float16)
    packed = torch.stack([hi, lo], dim=2).contiguous()                 # [ks, bo, split, lane, 8]
    return packed.view(-1).view(torch.float32).clone()


def block_exponent(*mats):
    """Power-of-two exponent e for one weight matrix of the reference (all its 128x128 slices share it); the
    slices are stored as 2^e W.  Matrices of ordinary scale (rms in [2^-4, 1), rms of the smallest 99 % of the
    weights) keep e = 0 - the plain split, whose error on such weights is at fp32 rounding level; a smaller matrix is
    lifted to rms 2^e W in [2^-4, 2^-3), where the fp16 `lo` halves of the weights that matter are no longer lost to
    the subnormal range (representation error back to ~2^-22 relative to the matrix' scale); a larger one is brought
    down to [1/2, 1), away from the fp16 maximum.  Lifting costs activation headroom (the accumulators and GELU
    outputs of the layer carry 2^e), so it is not done where it is not needed.  The largest element must end up
    <= 2^15; a matrix where that would push e more than MAX_RANGE_LOSS below the wanted value has no accurate split:
    ValueError."""
    a = torch.cat([m.detach().double().abs().reshape(-1) for m in mats])
    if not bool(torch.isfinite(a).all()):
        raise ValueError("non-finite weight: outside the fp16 range of the split-fp16 contraction modes")
    mx = float(a.max())
    if mx == 0.0:
        return 0
    # the scale of the bulk: rms of all but the largest 1 % (a few outliers must not decide the exponent of the rest)
    bulk = a.sort().values[:max(1, int(0.99 * a.numel()))]
    rms = max(float(bulk.pow(2).mean().sqrt()), mx * 2.0 ** -40)
    lg = math.floor(math.log2(rms))
    want = (-4 - lg) if lg < -4 else ((-1 - lg) if lg >= 0 else 0)
    limit = 15 - math.ceil(math.log2(mx))                              # 2^e max|w| <= 2^15
    if limit < want - MAX_RANGE_LOSS:
        raise ValueError(f"weight matrix with max |w| / rms = {mx / rms:.3g} has no accurate fp16 hi/lo "
                         "split (outside the fp16 range of the split-fp16 contraction modes); use precision='f32'")
    return int(max(-MAX_CHAIN_EXP, min(MAX_CHAIN_EXP, min(want, limit))))


BIAS_HEADROOM_LOG2 = 12     # 2^E max|bias| <= 2^12 at a GELU input: a factor 16 below the fp16 maximum for the W x part


def _chain(*es, biases=()):
    """Clamp exponents along an MLP so that the accumulated sums that reach a GELU (after the first and the second
    layer) stay within +-MAX_CHAIN_EXP; a third layer's sum only scales a residual and a LayerNorm epsilon
    (2^(2E) must be a normal fp32): +-40.
    biases[i] (optional): everything ADDED to layer i's pre-activation besides W x (the bias, the decoder's W1s W_s
    table).  Those are stored pre-multiplied by the accumulated 2^E and ride on the GELU output that is then split
    into fp16 halves, so E is also capped by their magnitude: a layer of tiny weights and ordinary biases (b ~ 1
    with E = 16 would be 65536) must not be lifted out of the fp16 range by its own exponent."""
    out, acc = [], 0
    for i, e in enumerate(es):
        lim = MAX_CHAIN_EXP if i < 2 else 40
        hi = lim - acc
        if i < 2 and i < len(biases) and biases[i] is not None:
            mx = float(torch.as_tensor(biases[i]).detach().abs().max())
            if mx > 0.0 and math.isfinite(mx):
                hi = min(hi, BIAS_HEADROOM_LOG2 - math.ceil(math.log2(mx)) - acc)
        e = max(-lim - acc, min(hi, e))
        out.append(e)
        acc += e
    return out


def strip_module_prefix(sd):
    """Checkpoints saved from a DDP-wrapped model carry `module.` (reference test.py:279-286)."""
    if all(k.startswith("module.") for k in sd):
        return OrderedDict((k[7:], v) for k, v in sd.items())
    return sd


def denoiser_tensors(sd):
    """Ordered name -> host fp32 tensor, as laid out in the blob."""
    sd = strip_module_prefix(sd)
    g = lambda k: sd[k].detach().float().cpu().contiguous()  # noqa: E731
    t = OrderedDict()
    # latent_model.py:62-64 and protein_mpnn_utils.py:464-465, evaluated the way the reference does
    t["freqs"] = torch.exp(-math.log(10000) * torch.arange(0, 128, dtype=torch.float32) / 128)
    t["rbf_mu"] = torch.linspace(2.0, 22.0, 16)
    t["t_w0"], t["t_b0"] = g("t_embedder.mlp.0.weight"), g("t_embedder.mlp.0.bias")
    t["t_w2"], t["t_b2"] = g("t_embedder.mlp.2.weight"), g("t_embedder.mlp.2.bias")
    heads = [f"encoder_layers.{l}" for l in range(3)] + [f"decoder_layers.{l}" for l in range(3)] + ["W_out"]
    for i, hname in enumerate(heads):
        t[f"ada_w{i}"] = g(f"{hname}.adaLN_modulation.1.weight")
        t[f"ada_b{i}"] = g(f"{hname}.adaLN_modulation.1.bias")
    t["x_in_w"], t["x_in_b"] = g("x_in.weight"), g("x_in.bias")
    # [128,3], or [128,6] for a self_condition model: columns = cat(x_self_cond, x) (latent_model.py:112-116, 210-212)
    assert t["x_in_w"].shape in ((H, 3), (H, 6)), "only latent_size 3 (N6/K3/K4) is built"
    t["pos_w"], t["pos_b"] = g("features.embeddings.linear.weight"), g("features.embeddings.linear.bias")
    t["edge_wT"] = g("features.edge_embedding.weight").t().contiguous()
    t["norm_w"], t["norm_b"] = g("features.norm_edges.weight"), g("features.norm_edges.bias")
    t["We_wT"], t["We_b"] = g("W_e.weight").t().contiguous(), g("W_e.bias")
    t["out_w"], t["out_b"] = g("W_out.linear.weight"), g("W_out.linear.bias")
    # 6 rows: eps | variance logits (diffusion); 3 rows: velocity (flow matching, latent_model.py:142-143)
    assert t["out_w"].shape in ((6, H), (3, H)), "W_out.linear must have 2 * 3 (diffusion) or 3 (flow) rows"
    for l in range(3):
        p = f"encoder_layers.{l}"
        W1, W11 = g(f"{p}.W1.weight"), g(f"{p}.W11.weight")
        e = f"enc{l}."
        t[e + "W1e"] = pack_block(W1[:, 128:256]); t[e + "W2"] = pack_block(g(f"{p}.W2.weight"))
        t[e + "W3"] = pack_block(g(f"{p}.W3.weight"))
        t[e + "W11e"] = pack_block(W11[:, 128:256]); t[e + "W12"] = pack_block(g(f"{p}.W12.weight"))
        t[e + "W13"] = pack_block(g(f"{p}.W13.weight"))
        t[e + "W1a"] = pack_block(W1[:, 0:128]); t[e + "W1c"] = pack_block(W1[:, 256:384])
        t[e + "W11a"] = pack_block(W11[:, 0:128]); t[e + "W11c"] = pack_block(W11[:, 256:384])
        Win, Wout = g(f"{p}.dense.W_in.weight"), g(f"{p}.dense.W_out.weight")
        for c in range(4):
            t[e + f"Win{c}"] = pack_block(Win[128 * c:128 * c + 128, :])
            t[e + f"Wout{c}"] = pack_block(Wout[:, 128 * c:128 * c + 128])
        for b, k in (("b1", "W1"), ("b2", "W2"), ("b3", "W3"), ("b11", "W11"), ("b12", "W12"), ("b13", "W13")):
            t[e + b] = g(f"{p}.{k}.bias")
        t[e + "b_in"], t[e + "b_out"] = g(f"{p}.dense.W_in.bias"), g(f"{p}.dense.W_out.bias")
    Ws = g("W_s.weight")
    for l in range(3):
        p = f"decoder_layers.{l}"
        W1 = g(f"{p}.W1.weight")
        d = f"dec{l}."
        # h_ESV = [h_E|h_S_j|h_V_j] + [h_E|h_S_j|h_Venc_j] (latent_model.py:260-261): h_E and h_S doubled
        t[d + "W1e"] = pack_block(W1[:, 128:256], 2.0)
        t[d + "W2"] = pack_block(g(f"{p}.W2.weight")); t[d + "W3"] = pack_block(g(f"{p}.W3.weight"))
        t[d + "W1a"] = pack_block(W1[:, 0:128]); t[d + "W1v"] = pack_block(W1[:, 384:512])
        t[d + "TS"] = F.linear(2.0 * Ws, W1[:, 256:384]).contiguous()  # [30,128]
        Win, Wout = g(f"{p}.dense.W_in.weight"), g(f"{p}.dense.W_out.weight")
        for c in range(4):
            t[d + f"Win{c}"] = pack_block(Win[128 * c:128 * c + 128, :])
            t[d + f"Wout{c}"] = pack_block(Wout[:, 128 * c:128 * c + 128])
        for b, k in (("b1", "W1"), ("b2", "W2"), ("b3", "W3")):
            t[d + b] = g(f"{p}.{k}.bias")
        t[d + "b_in"], t[d + "b_out"] = g(f"{p}.dense.W_in.bias"), g(f"{p}.dense.W_out.bias")
    return t


def denoiser_tensors_h(sd, strict=True, use_exponents=True):
    """Split-fp16 (hi/lo) copies of every 128x128 block with their block exponents applied, the matching
    pre-scaled biases, same names with an `h.` prefix (layout: include/codlad_hip.h, codlad_enc_layer_h).
    Returns (tensors, exponents {layer: {name: e}}, n_unsplittable).
    strict=False (fp32-MFMA mode, where these copies are not read): a layer whose weights have no fp16 split is
    stored as zeros and counted instead of raising.  use_exponents=False packs every block with e = 0 (the
    plain split, for A/B measurements)."""
    sd = strip_module_prefix(sd)
    g = lambda k: sd[k].detach().float().cpu().contiguous()  # noqa: E731
    t, exps = OrderedDict(), OrderedDict()
    bad = [0]

    def expo(*mats):
        return block_exponent(*mats) if use_exponents else 0

    def pk(w, e, scale=1.0):
        return pack_block_h(w, scale * 2.0 ** e)

    def layer(prefix, build):
        try:
            build()
        except ValueError:
            if strict:
                raise
            bad[0] += 1
            for k in [k for k in t if k.startswith(prefix)]:
                del t[k]
            build(zero=True)

    for l in range(3):
        p, e_ = f"encoder_layers.{l}", f"h.enc{l}."

        def build(zero=False, p=p, e_=e_, l=l):
            W1, W11 = g(f"{p}.W1.weight"), g(f"{p}.W11.weight")
            W2, W3, W12, W13 = (g(f"{p}.{n}.weight") for n in ("W2", "W3", "W12", "W13"))
            Win, Wout = g(f"{p}.dense.W_in.weight"), g(f"{p}.dense.W_out.weight")
            if zero:
                Z = torch.zeros(H, H)
                W1, W11, Win, Wout = Z.repeat(1, 3), Z.repeat(1, 3), Z.repeat(4, 1), Z.repeat(1, 4)
                W2 = W3 = W12 = W13 = Z
            bias = (lambda k: None) if zero else (lambda k: g(f"{p}.{k}.bias"))
            e1, e2 = _chain(expo(W1), expo(W2), biases=(bias("W1"), bias("W2")))
            e11, e12, e13 = _chain(expo(W11), expo(W12), expo(W13), biases=(bias("W11"), bias("W12")))
            e_in, e_out = _chain(expo(Win), expo(Wout), biases=(bias("dense.W_in"),))
            e3 = expo(W3)                      # node kernel: taken out by a constant multiply, never reaches a GELU
            exps[f"enc{l}"] = dict(e1=e1, e2=e2, e3=e3, e11=e11, e12=e12, e13=e13, e_in=e_in, e_out=e_out)
            t[e_ + "W1e"] = pk(W1[:, 128:256], e1); t[e_ + "W2"] = pk(W2, e2); t[e_ + "W3"] = pk(W3, e3)
            t[e_ + "W11e"] = pk(W11[:, 128:256], e11); t[e_ + "W12"] = pk(W12, e12); t[e_ + "W13"] = pk(W13, e13)
            t[e_ + "W1a"] = pk(W1[:, 0:128], e1); t[e_ + "W1c"] = pk(W1[:, 256:384], e1)
            t[e_ + "W11a"] = pk(W11[:, 0:128], e11); t[e_ + "W11c"] = pk(W11[:, 256:384], e11)
            for c in range(4):
                t[e_ + f"Win{c}"] = pk(Win[128 * c:128 * c + 128, :], e_in)
                t[e_ + f"Wout{c}"] = pk(Wout[:, 128 * c:128 * c + 128], e_out)
            for b, k, e in (("b1", "W1", e1), ("b2", "W2", e1 + e2), ("b3", "W3", e3), ("b11", "W11", e11),
                            ("b12", "W12", e11 + e12), ("b13", "W13", e11 + e12 + e13),
                            ("b_in", "dense.W_in", e_in), ("b_out", "dense.W_out", e_in + e_out)):
                t[e_ + b] = (torch.zeros_like(g(f"{p}.{k}.bias")) if zero else g(f"{p}.{k}.bias")) * 2.0 ** e

        layer(e_, build)
    Ws = g("W_s.weight")
    for l in range(3):
        p, d_ = f"decoder_layers.{l}", f"h.dec{l}."

        def build(zero=False, p=p, d_=d_, l=l):
            W1 = g(f"{p}.W1.weight")
            W2, W3 = g(f"{p}.W2.weight"), g(f"{p}.W3.weight")
            Win, Wout = g(f"{p}.dense.W_in.weight"), g(f"{p}.dense.W_out.weight")
            if zero:
                Z = torch.zeros(H, H)
                W1, Win, Wout, W2, W3 = Z.repeat(1, 4), Z.repeat(4, 1), Z.repeat(1, 4), Z, Z
            # h_ESV = [h_E|h_S_j|h_V_j] + [h_E|h_S_j|h_Venc_j] (latent_model.py:260-261): h_E and h_S doubled
            bias = (lambda k: None) if zero else (lambda k: g(f"{p}.{k}.bias"))
            b1_all = None if zero else torch.cat([g(f"{p}.W1.bias"), F.linear(2.0 * Ws, W1[:, 256:384]).reshape(-1)])
            e1, e2 = _chain(expo(W1[:, 0:128], 2.0 * W1[:, 128:256], W1[:, 384:512]), expo(W2),
                            biases=(b1_all, bias("W2")))
            e_in, e_out = _chain(expo(Win), expo(Wout), biases=(bias("dense.W_in"),))
            e3 = expo(W3)
            exps[f"dec{l}"] = dict(e1=e1, e2=e2, e3=e3, e_in=e_in, e_out=e_out)
            t[d_ + "W1e"] = pk(W1[:, 128:256], e1, 2.0)
            t[d_ + "W2"] = pk(W2, e2); t[d_ + "W3"] = pk(W3, e3)
            t[d_ + "W1a"] = pk(W1[:, 0:128], e1); t[d_ + "W1v"] = pk(W1[:, 384:512], e1)
            t[d_ + "TS"] = (F.linear(2.0 * Ws, W1[:, 256:384]) * 2.0 ** e1).contiguous()
            for c in range(4):
                t[d_ + f"Win{c}"] = pk(Win[128 * c:128 * c + 128, :], e_in)
                t[d_ + f"Wout{c}"] = pk(Wout[:, 128 * c:128 * c + 128], e_out)
            for b, k, e in (("b1", "W1", e1), ("b2", "W2", e1 + e2), ("b3", "W3", e3),
                            ("b_in", "dense.W_in", e_in), ("b_out", "dense.W_out", e_in + e_out)):
                t[d_ + b] = (torch.zeros_like(g(f"{p}.{k}.bias")) if zero else g(f"{p}.{k}.bias")) * 2.0 ** e

        layer(d_, build)
    return t, exps, bad[0]


def decoder_tensors(sd):
    """IC decoder + map_out + codebook (reference models/vae_model.py:318-503, 704-706)."""
    g = lambda k: sd[k].detach().float().cpu().contiguous()  # noqa: E731
    p = "equivaraintconv."
    angle = (p + "sidechain_angle.1.weight") in sd
    t = OrderedDict()
    if "map_out.weight" in sd:        # VQ-VAE (N6 / K3 / K4): 3 -> 36; the C2 model decodes its 36-wide latent as it is
        t["map_out_w"], t["map_out_b"] = g("map_out.weight"), g("map_out.bias")
    t["res_embed"] = g(p + "res_embed.weight")
    for i in range(4):
        m = f"{p}message_blocks.{i}."
        t[f"inv0_w{i}"], t[f"inv0_b{i}"] = g(m + "inv_dense.0.weight"), g(m + "inv_dense.0.bias")
        t[f"inv1_w{i}"], t[f"inv1_b{i}"] = g(m + "inv_dense.1.weight"), g(m + "inv_dense.1.bias")
        t[f"dist_w{i}"], t[f"dist_b{i}"] = g(m + "dist_embed.block.1.weight"), g(m + "dist_embed.block.1.bias")
        d = f"{p}dense_blocks.{i}."
        t[f"dense1_w{i}"], t[f"dense1_b{i}"] = g(d + "1.weight"), g(d + "1.bias")
        t[f"dense3_w{i}"], t[f"dense3_b{i}"] = g(d + "3.weight"), g(d + "3.bias")
        s = f"{p}sidechain_torsion_blocks.{i}."
        t[f"tor1_w{i}"], t[f"tor1_b{i}"] = g(s + "1.weight"), g(s + "1.bias")
        t[f"tor3_w{i}"], t[f"tor3_b{i}"] = g(s + "3.weight"), g(s + "3.bias")
    t["bb_dist"], t["sc_dist"] = g(p + "backbone_dist.weight"), g(p + "sidechain_dist.weight")
    for nm, key in (("bb_ang", "backbone_angle"), ("bb_tor", "backbone_torsion"), ("fin", "final_torsion")):
        t[nm + "1_w"], t[nm + "1_b"] = g(f"{p}{key}.1.weight"), g(f"{p}{key}.1.bias")
        t[nm + "3_w"], t[nm + "3_b"] = g(f"{p}{key}.3.weight"), g(f"{p}{key}.3.bias")
    if angle:
        t["sc_ang1_w"], t["sc_ang1_b"] = g(p + "sidechain_angle.1.weight"), g(p + "sidechain_angle.1.bias")
        t["sc_ang3_w"], t["sc_ang3_b"] = g(p + "sidechain_angle.3.weight"), g(p + "sidechain_angle.3.bias")
    else:
        t["sc_angle_emb"] = g(p + "sidechain_angle.weight")
    if "quantize._codebook.embed" in sd:          # vector_quantize_pytorch EuclideanCodebook
        t["codebook"] = g("quantize._codebook.embed")[0].contiguous()
    elif "quantize.embeddings" in sd:             # in-repo VectorQuantizerEMA (utils/vq_module.py:52)
        t["codebook"] = g("quantize.embeddings")
    return t, angle


class Blob:
    """name -> fp32 tensor, stored back to back (256-byte aligned) in one device buffer."""

    def __init__(self, tensors, device):
        self.offsets = OrderedDict()
        off = 0
        for k, v in tensors.items():
            self.offsets[k] = (off, tuple(v.shape))
            off += (v.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        host = torch.zeros(off, dtype=torch.float32)
        for k, v in tensors.items():
            o, _ = self.offsets[k]
            host[o:o + v.numel()] = v.reshape(-1)
        self.data = host.to(device)

    def ptr(self, name):
        if name not in self.offsets:
            return None
        return C.c_void_p(self.data.data_ptr() + 4 * self.offsets[name][0])

    def view(self, name):
        o, shape = self.offsets[name]
        n = int(np.prod(shape))
        return self.data[o:o + n].view(shape)


PRECISIONS = {"f32": 0, "f16x4": 1, "f16x3": 2}      # include/codlad_hip.h, codlad_denoiser_weights.precision
DEFAULT_PRECISION = "f16x3"

# Blob header.  The first META_FLOATS floats of every blob hold, as exactly representable integers, everything the
# kernels need that is NOT a pointer: contraction mode, model flags and the block exponents of the split-fp16
# copies (they scale biases, GELU constants, the residual and the LayerNorm epsilon: a rank that received another
# rank's blocks but kept its own exponents would be silently wrong by powers of two).  `parallel.broadcast_weights`
# sends the header first, a receiving rank re-derives the layout from it (it may have started from any weights, or
# from none: `DenoiserWeights.empty`), then the blob travels as one buffer and `rebind()` reads the header back.
META_FLOATS = 64
META_MAGIC = 0xC0D1AD            # < 2^24: exact in fp32
META_LAYOUT_VERSION = 3
KIND_DENOISER, KIND_DECODER = 1, 2
ENC_EXP_NAMES = ("e1", "e2", "e3", "e11", "e12", "e13", "e_in", "e_out")
DEC_EXP_NAMES = ("e1", "e2", "e3", "e_in", "e_out")


def _check_header(hdr, kind):
    h = [int(round(float(v))) for v in hdr.reshape(-1)[:META_FLOATS].tolist()]
    if h[0] != META_MAGIC or h[1] != kind or h[2] != META_LAYOUT_VERSION:
        raise RuntimeError(f"weight blob header mismatch: magic {h[0]:#x} kind {h[1]} layout {h[2]} "
                           f"(expected {META_MAGIC:#x} / {kind} / {META_LAYOUT_VERSION})")
    return h


def denoiser_shapes(self_condition=False, out_dim=6):
    """name -> shape of the 108 checkpoint tensors of `mpnn_diffusion` (reference models/latent_model.py:119-148)."""
    sh = OrderedDict()

    def lin(name, o, i, bias=True):
        sh[name + ".weight"] = (o, i)
        if bias:
            sh[name + ".bias"] = (o,)

    lin("t_embedder.mlp.0", H, 256); lin("t_embedder.mlp.2", H, H)
    lin("x_in", H, 6 if self_condition else 3)
    lin("features.embeddings.linear", 16, 66)
    lin("features.edge_embedding", H, 167, bias=False)
    sh["features.norm_edges.weight"], sh["features.norm_edges.bias"] = (H,), (H,)
    lin("W_e", H, H)
    sh["W_s.weight"] = (30, H)
    for l in range(3):
        p = f"encoder_layers.{l}."
        for n, i in (("W1", 3 * H), ("W2", H), ("W3", H), ("W11", 3 * H), ("W12", H), ("W13", H)):
            lin(p + n, H, i)
        lin(p + "dense.W_in", 4 * H, H); lin(p + "dense.W_out", H, 4 * H)
        lin(p + "adaLN_modulation.1", 9 * H, H)
    for l in range(3):
        p = f"decoder_layers.{l}."
        for n, i in (("W1", 4 * H), ("W2", H), ("W3", H)):
            lin(p + n, H, i)
        lin(p + "dense.W_in", 4 * H, H); lin(p + "dense.W_out", H, 4 * H)
        lin(p + "adaLN_modulation.1", 6 * H, H)
    lin("W_out.linear", out_dim, H)
    lin("W_out.adaLN_modulation.1", 2 * H, H)
    return sh


def decoder_shapes(angle=False, n_codes=4096):
    """name -> shape of the decoder-side VQ-VAE tensors `decoder_tensors` reads (reference models/vae_model.py:318-373,
    414-465, 704-706; codebook in the vector_quantize_pytorch layout).  n_codes = 0: no codebook."""
    sh = OrderedDict()
    F = 40
    Ft = F + 10 if angle else F
    p = "equivaraintconv."

    def lin(name, o, i):
        sh[name + ".weight"], sh[name + ".bias"] = (o, i), (o,)

    lin("map_out", 36, 3)
    sh[p + "res_embed.weight"] = (25, 4)
    for i in range(4):
        lin(f"{p}message_blocks.{i}.inv_dense.0", F, F); lin(f"{p}message_blocks.{i}.inv_dense.1", F, F)
        lin(f"{p}message_blocks.{i}.dist_embed.block.1", F, 15)
        lin(f"{p}dense_blocks.{i}.1", F, F); lin(f"{p}dense_blocks.{i}.3", F, F)
        lin(f"{p}sidechain_torsion_blocks.{i}.1", Ft, Ft); lin(f"{p}sidechain_torsion_blocks.{i}.3", Ft, Ft)
    sh[p + "backbone_dist.weight"], sh[p + "sidechain_dist.weight"] = (25, 3), (25, 10)
    lin(p + "backbone_angle.1", 3, F); lin(p + "backbone_angle.3", 3, 3)
    lin(p + "backbone_torsion.1", 3, F + 3); lin(p + "backbone_torsion.3", 3, 3)
    lin(p + "final_torsion.1", 10, Ft); lin(p + "final_torsion.3", 10, 10)
    if angle:
        lin(p + "sidechain_angle.1", 10, F); lin(p + "sidechain_angle.3", 10, 10)
    else:
        sh[p + "sidechain_angle.weight"] = (25, 10)
    if n_codes:
        sh["quantize._codebook.embed"] = (1, n_codes, 3)
    return sh


def _zeros(shapes):
    return OrderedDict((k, torch.zeros(*v)) for k, v in shapes.items())


class DenoiserWeights:
    def __init__(self, state_dict, device, precision=DEFAULT_PRECISION, block_exponents=True):
        if precision not in PRECISIONS:
            raise ValueError(f"precision {precision!r}: one of {sorted(PRECISIONS)}")
        tensors = OrderedDict(meta=torch.zeros(META_FLOATS))
        tensors.update(denoiser_tensors(state_dict))
        split, self.exponents, n_bad = denoiser_tensors_h(state_dict, strict=precision != "f32",
                                                          use_exponents=block_exponents)
        self.splittable = n_bad == 0      # False: some layer has no fp16 split, fp32-MFMA mode only
        tensors.update(split)
        self.blob = Blob(tensors, device)
        self.precision = precision
        self.self_condition = tensors["x_in_w"].shape[1] == 6
        self.out_dim = int(tensors["out_w"].shape[0])
        self.generation = 0               # bumped whenever the blob's content or the host-side fields change
        self.sync_meta()
        self.struct = self._fill()

    @classmethod
    def empty(cls, device, self_condition=False, out_dim=6, precision=DEFAULT_PRECISION):
        """The layout of a model with these flags, every weight zero: what a rank that loads no checkpoint
        starts from before `parallel.broadcast_weights` fills it with rank 0's."""
        return cls(_zeros(denoiser_shapes(self_condition, out_dim)), device, precision)

    # -- header ------------------------------------------------------------------------------------
    def header(self):
        """[META_FLOATS] host fp32 tensor: magic, kind, layout version, precision, self_condition, out_dim,
        splittable, blob size, then the 3 x 8 encoder and 3 x 5 decoder block exponents."""
        h = [META_MAGIC, KIND_DENOISER, META_LAYOUT_VERSION, PRECISIONS[self.precision], int(self.self_condition),
             self.out_dim, int(self.splittable), self.blob.data.numel()]
        for l in range(3):
            h += [self.exponents[f"enc{l}"][n] for n in ENC_EXP_NAMES]
        for l in range(3):
            h += [self.exponents[f"dec{l}"][n] for n in DEC_EXP_NAMES]
        assert len(h) <= META_FLOATS and h[7] < 2 ** 24
        return torch.tensor(h + [0] * (META_FLOATS - len(h)), dtype=torch.float32)

    def sync_meta(self):
        """Host fields -> the header inside the blob (what travels with a broadcast)."""
        self.blob.view("meta").copy_(self.header())

    def adopt_header(self, hdr):
        """Take over another rank's header BEFORE its blob arrives: re-derive the layout when the model flags
        differ (x_in / W_out have other shapes then) and set every host-side field from it."""
        h = _check_header(hdr, KIND_DENOISER)
        sc, od = bool(h[4]), h[5]
        if (sc, od) != (self.self_condition, self.out_dim):
            fresh = DenoiserWeights.empty(self.blob.data.device, sc, od, self.precision)
            self.blob, self.self_condition, self.out_dim = fresh.blob, sc, od
        if h[7] != self.blob.data.numel():
            raise RuntimeError(f"weight blob of {h[7]} floats announced, local layout has {self.blob.data.numel()}")
        self._take_fields(h)

    def _take_fields(self, h):
        self.precision = {v: k for k, v in PRECISIONS.items()}[h[3]]
        self.splittable = bool(h[6])
        o = 8
        for l in range(3):
            self.exponents[f"enc{l}"] = dict(zip(ENC_EXP_NAMES, h[o:o + 8]))
            o += 8
        for l in range(3):
            self.exponents[f"dec{l}"] = dict(zip(DEC_EXP_NAMES, h[o:o + 5]))
            o += 5

    def _fill(self):
        w = _lib.DenoiserWeights()
        p = self.blob.ptr
        for n in ("freqs", "rbf_mu", "t_w0", "t_b0", "t_w2", "t_b2", "x_in_w", "x_in_b", "pos_w", "pos_b",
                  "edge_wT", "norm_w", "norm_b", "We_wT", "We_b", "out_w", "out_b"):
            setattr(w, n, p(n))
        for i in range(7):
            w.ada_w[i], w.ada_b[i] = p(f"ada_w{i}"), p(f"ada_b{i}")
        for l in range(3):
            e, d = w.enc[l], w.dec[l]
            for n in ("W1e", "W2", "W3", "W11e", "W12", "W13", "W1a", "W1c", "W11a", "W11c",
                      "b1", "b2", "b3", "b11", "b12", "b13", "b_in", "b_out"):
                setattr(e, n, p(f"enc{l}.{n}"))
            for n in ("W1e", "W2", "W3", "W1a", "W1v", "TS", "b1", "b2", "b3", "b_in", "b_out"):
                setattr(d, n, p(f"dec{l}.{n}"))
            for c in range(4):
                e.Win[c], e.Wout[c] = p(f"enc{l}.Win{c}"), p(f"enc{l}.Wout{c}")
                d.Win[c], d.Wout[c] = p(f"dec{l}.Win{c}"), p(f"dec{l}.Wout{c}")
            eh, dh = w.enc_h[l], w.dec_h[l]
            for n in ("W1e", "W2", "W3", "W11e", "W12", "W13", "W1a", "W1c", "W11a", "W11c",
                      "b1", "b2", "b3", "b11", "b12", "b13", "b_in", "b_out"):
                setattr(eh, n, p(f"h.enc{l}.{n}"))
            for n in ("W1e", "W2", "W3", "W1a", "W1v", "TS", "b1", "b2", "b3", "b_in", "b_out"):
                setattr(dh, n, p(f"h.dec{l}.{n}"))
            for n, e in self.exponents[f"enc{l}"].items():
                setattr(eh, n, e)
            for n, e in self.exponents[f"dec{l}"].items():
                setattr(dh, n, e)
            for c in range(4):
                eh.Win[c], eh.Wout[c] = p(f"h.enc{l}.Win{c}"), p(f"h.enc{l}.Wout{c}")
                dh.Win[c], dh.Wout[c] = p(f"h.dec{l}.Win{c}"), p(f"h.dec{l}.Wout{c}")
        w.precision = PRECISIONS[self.precision]
        w.self_condition = int(self.self_condition)
        w.out_dim = self.out_dim
        return w

    def set_precision(self, precision):
        if precision != "f32" and not self.splittable:
            raise ValueError("these weights hold a block outside the fp16 range: only precision 'f32' can run them")
        self.precision = precision
        self.struct.precision = PRECISIONS[precision]
        self.sync_meta()
        self.generation += 1

    def rebind(self):
        """After the blob's content or storage changed (a broadcast): read the header back from the blob - the
        block exponents, the contraction mode and the model flags are whatever the blob says, never what this rank
        derived from its own weights - and re-derive the pointers."""
        h = _check_header(self.blob.view("meta").cpu(), KIND_DENOISER)
        if (bool(h[4]), h[5]) != (self.self_condition, self.out_dim) or h[7] != self.blob.data.numel():
            raise RuntimeError("blob header does not match this layout: adopt_header() must precede the blob")
        self._take_fields(h)
        self.generation += 1
        self.struct = self._fill()

    def checksum(self):
        """64-bit sum of the blob's 32-bit words (ranks compare it after a broadcast)."""
        return int(self.blob.data.view(torch.int32).to(torch.int64).sum())


class DecoderWeights:
    def __init__(self, state_dict, device, mean3=None, std3=None):
        tensors = OrderedDict(meta=torch.zeros(META_FLOATS))
        # de-normalisation statistics of the latent (reference utils/dataset_module.py:230-256): part of the blob so
        # that they travel with the codebook ("packed weight blob + codebook + norm stats from rank 0")
        tensors["norm_mean"] = (torch.zeros(3) if mean3 is None else mean3).detach().float().cpu().reshape(3).clone()
        tensors["norm_std"] = (torch.ones(3) if std3 is None else std3).detach().float().cpu().reshape(3).clone()
        dec, self.angle = decoder_tensors(state_dict)
        tensors.update(dec)
        self.n_codes = int(tensors["codebook"].shape[0]) if "codebook" in tensors else 0
        self.blob = Blob(tensors, device)
        self.generation = 0
        self.sync_meta()
        self.struct = self._fill()

    @classmethod
    def empty(cls, device, angle=False, n_codes=4096):
        return cls(_zeros(decoder_shapes(angle, n_codes)), device)

    def header(self):
        h = [META_MAGIC, KIND_DECODER, META_LAYOUT_VERSION, int(self.angle), self.n_codes, self.blob.data.numel()]
        return torch.tensor(h + [0] * (META_FLOATS - len(h)), dtype=torch.float32)

    def sync_meta(self):
        self.blob.view("meta").copy_(self.header())

    def adopt_header(self, hdr):
        h = _check_header(hdr, KIND_DECODER)
        angle, n_codes = bool(h[3]), h[4]
        if (angle, n_codes) != (self.angle, self.n_codes):
            fresh = DecoderWeights.empty(self.blob.data.device, angle, n_codes)
            self.blob, self.angle, self.n_codes = fresh.blob, angle, n_codes
        if h[5] != self.blob.data.numel():
            raise RuntimeError(f"decoder blob of {h[5]} floats announced, local layout has {self.blob.data.numel()}")

    def _fill(self):
        w = _lib.DecoderWeights()
        p = self.blob.ptr
        w.angle = int(self.angle)
        for n in ("map_out_w", "map_out_b", "res_embed", "bb_dist", "sc_dist", "bb_ang1_w", "bb_ang1_b",
                  "bb_ang3_w", "bb_ang3_b", "sc_angle_emb", "sc_ang1_w", "sc_ang1_b", "sc_ang3_w",
                  "sc_ang3_b", "bb_tor1_w", "bb_tor1_b", "bb_tor3_w", "bb_tor3_b", "fin1_w", "fin1_b",
                  "fin3_w", "fin3_b"):
            setattr(w, n, p(n))
        for i in range(4):
            for n in ("inv0_w", "inv0_b", "inv1_w", "inv1_b", "dist_w", "dist_b", "dense1_w", "dense1_b",
                      "dense3_w", "dense3_b", "tor1_w", "tor1_b", "tor3_w", "tor3_b"):
                getattr(w, n)[i] = p(f"{n}{i}")
        return w

    def rebind(self):
        h = _check_header(self.blob.view("meta").cpu(), KIND_DECODER)
        if (bool(h[3]), h[4]) != (self.angle, self.n_codes) or h[5] != self.blob.data.numel():
            raise RuntimeError("blob header does not match this layout: adopt_header() must precede the blob")
        self.generation += 1
        self.struct = self._fill()

    def checksum(self):
        return int(self.blob.data.view(torch.int32).to(torch.int64).sum())

    @property
    def codebook(self):
        return self.blob.view("codebook")

    @property
    def mean(self):
        return self.blob.view("norm_mean")

    @property
    def std(self):
        return self.blob.view("norm_std")
