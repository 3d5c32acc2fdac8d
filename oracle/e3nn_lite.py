"""TEST INFRASTRUCTURE (CPU oracle) - the e3nn PRIMITIVES are PARITY UNPINNED; the encoder / prior built on them are pinned
by the reference's own forward passes (goldens g15: tools/gen_golden.py binds e3nn.o3 to adapters over the primitives below
and runs the reference's e3nnPrior / e3nnEncoder / TensorProductConvLayer; tests/test_oracle_vs_golden.py holds
prior_forward / encoder_forward / tp_conv_layer to them).

Restatement of the pieces of e3nn 0.5.1 (requirements.txt:6 of the reference; the package is NOT installed here and
is not vendored by the reference) that the reference's encoder / prior use, from e3nn's published definitions:

  * real spherical harmonics up to l = 2, `o3.spherical_harmonics(irreps, x, normalize=True,
    normalization='component')` (reference models/vae_model.py:178, 192, 199, 280);
  * the real Wigner 3j symbols `o3.wigner_3j(l1, l2, l3)` for l <= 2, built the way e3nn builds them: SU(2)
    Clebsch-Gordan coefficients (Condon-Shortley phases) conjugated into e3nn's real basis
    (`change_basis_real_to_complex`, with its factor (-i)^l), real part, Frobenius norm 1;
  * `o3.FullyConnectedTensorProduct(in, sh, out, shared_weights=False)` with e3nn's defaults
    irrep_normalization='component', path_normalization='element': instructions in the order
    (i_in1, i_in2, i_out), mode 'uvw', per-instruction coefficient sqrt((2 l_out + 1) / sum over the instructions that
    feed the same output slot of mul_in1 * mul_in2), weights laid end to end as [mul_in1, mul_in2, mul_out] blocks.

Because nothing in the reference exercises e3nn and the package is absent, none of this can be checked against e3nn
itself: a checkpoint trained with e3nn is only reproduced if every convention above (basis order, signs,
normalisation, weight order) was recalled correctly.  What IS checked (tests/test_e3nn_encoder.py): the symbols are
rotation-equivariant intertwiners consistent with the harmonics (so the layer is equivariant), orthonormal, and the
known closed forms (delta / sqrt(2l+1), epsilon / sqrt 6).
"""
import math
from fractions import Fraction
from functools import lru_cache

import torch


# ----------------------------------------------------------------------------------------------------------------
# irreps
# ----------------------------------------------------------------------------------------------------------------
def parse_irreps(s):
    """'12x0e + 4x1o' -> [(12, 0, +1), (4, 1, -1)]  (mul, l, parity)."""
    out = []
    for term in s.replace(" ", "").split("+"):
        mul, ir = term.split("x") if "x" in term else ("1", term)
        out.append((int(mul), int(ir[:-1]), 1 if ir[-1] == "e" else -1))
    return out


def irreps_dim(irreps):
    return sum(mul * (2 * l + 1) for mul, l, _p in irreps)


def sh_irreps(lmax):
    """o3.Irreps.spherical_harmonics(lmax): 1x0e + 1x1o + 1x2e ... (parity (-1)^l)."""
    return [(1, l, (-1) ** l) for l in range(lmax + 1)]


def irrep_seq(ns=12, nv=4):
    """The encoder's feature irreps per depth (reference vae_model.py:74-80, use_second_order_repr=False)."""
    return [f"{ns}x0e", f"{ns}x0e + {nv}x1o", f"{ns}x0e + {nv}x1o + {nv}x1e", f"{ns}x0e + {nv}x1o + {nv}x1e + {ns}x0o"]


# ----------------------------------------------------------------------------------------------------------------
# spherical harmonics (e3nn o3/_spherical_harmonics.py: polynomials with |Y_l(x)| = 1 on the unit sphere, y the polar
# axis, l = 1 in the order x, y, z; 'component' normalisation multiplies block l by sqrt(2l + 1))
# ----------------------------------------------------------------------------------------------------------------
def spherical_harmonics(lmax, vec, normalize=True):
    if normalize:
        vec = torch.nn.functional.normalize(vec, dim=-1)     # x / max(|x|, 1e-12), as e3nn does
    x, y, z = vec[..., 0], vec[..., 1], vec[..., 2]
    out = [torch.ones_like(x)]
    if lmax >= 1:
        s3 = math.sqrt(3.0)
        out += [s3 * x, s3 * y, s3 * z]
    if lmax >= 2:
        s5, s3 = math.sqrt(5.0), math.sqrt(3.0)
        x2z2 = x * x + z * z
        out += [s5 * (s3 * x * z), s5 * (s3 * x * y), s5 * (y * y - 0.5 * x2z2), s5 * (s3 * y * z),
                s5 * (s3 / 2.0 * (z * z - x * x))]
    assert lmax <= 2
    return torch.stack(out, dim=-1)


# ----------------------------------------------------------------------------------------------------------------
# Wigner 3j in the real basis
# ----------------------------------------------------------------------------------------------------------------
def _f(n):
    return math.factorial(round(n))


def _su2_cg(j1, m1, j2, m2, j3, m3):
    """<j1 m1 j2 m2 | j3 m3>, Condon-Shortley (Racah's formula), exact rational arithmetic under the root."""
    if m3 != m1 + m2:
        return 0.0
    pre = Fraction((2 * j3 + 1) * _f(j3 + j1 - j2) * _f(j3 - j1 + j2) * _f(j1 + j2 - j3), _f(j1 + j2 + j3 + 1))
    pre *= _f(j3 + m3) * _f(j3 - m3) * _f(j1 - m1) * _f(j1 + m1) * _f(j2 - m2) * _f(j2 + m2)
    s = Fraction(0)
    for k in range(0, j1 + j2 - j3 + 1):
        den = [k, j1 + j2 - j3 - k, j1 - m1 - k, j2 + m2 - k, j3 - j2 + m1 + k, j3 - j1 - m2 + k]
        if min(den) < 0:
            continue
        d = 1
        for v in den:
            d *= _f(v)
        s += Fraction((-1) ** k, d)
    return float(s) * math.sqrt(float(pre))


def change_basis_real_to_complex(l):
    """e3nn o3/_wigner.py: rows = complex m = -l..l, columns = real index; the factor (-i)^l makes the CG real."""
    q = torch.zeros(2 * l + 1, 2 * l + 1, dtype=torch.complex128)
    for m in range(-l, 0):
        q[l + m, l + abs(m)] = 1 / math.sqrt(2)
        q[l + m, l - abs(m)] = -1j / math.sqrt(2)
    q[l, l] = 1
    for m in range(1, l + 1):
        q[l + m, l + abs(m)] = (-1) ** m / math.sqrt(2)
        q[l + m, l - abs(m)] = 1j * (-1) ** m / math.sqrt(2)
    return (-1j) ** l * q


@lru_cache(maxsize=None)
def wigner_3j(l1, l2, l3):
    """[2l1+1, 2l2+1, 2l3+1] float64, Frobenius norm 1."""
    assert abs(l2 - l3) <= l1 <= l2 + l3
    C = torch.zeros(2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1, dtype=torch.complex128)
    for m1 in range(-l1, l1 + 1):
        for m2 in range(-l2, l2 + 1):
            m3 = m1 + m2
            if abs(m3) <= l3:
                C[l1 + m1, l2 + m2, l3 + m3] = _su2_cg(l1, m1, l2, m2, l3, m3)
    Q1, Q2, Q3 = (change_basis_real_to_complex(l) for l in (l1, l2, l3))
    C = torch.einsum("ij,kl,mn,ikn->jlm", Q1, Q2, torch.conj(Q3.T), C)
    assert float(C.imag.abs().max()) < 1e-9
    C = C.real
    return C / C.norm()


# ----------------------------------------------------------------------------------------------------------------
# FullyConnectedTensorProduct, shared_weights=False
# ----------------------------------------------------------------------------------------------------------------
class TensorProduct:
    """Instructions, weight layout and path coefficients of o3.FullyConnectedTensorProduct(in1, in2, out)."""

    def __init__(self, irreps_in1, irreps_in2, irreps_out):
        self.in1, self.in2, self.out = (parse_irreps(i) if isinstance(i, str) else list(i)
                                        for i in (irreps_in1, irreps_in2, irreps_out))
        self.instr = [(i1, i2, io) for i1, (_m1, l1, p1) in enumerate(self.in1)
                      for i2, (_m2, l2, p2) in enumerate(self.in2)
                      for io, (_mo, lo, po) in enumerate(self.out)
                      if abs(l1 - l2) <= lo <= l1 + l2 and po == p1 * p2]
        fan = {}
        for i1, i2, io in self.instr:
            fan[io] = fan.get(io, 0) + self.in1[i1][0] * self.in2[i2][0]
        self.coeff = [math.sqrt((2 * self.out[io][1] + 1) / fan[io]) for _i1, _i2, io in self.instr]
        self.weight_numel = sum(self.in1[i1][0] * self.in2[i2][0] * self.out[io][0] for i1, i2, io in self.instr)

    @staticmethod
    def _offsets(irreps):
        o, offs = 0, []
        for mul, l, _p in irreps:
            offs.append(o)
            o += mul * (2 * l + 1)
        return offs

    def __call__(self, x1, x2, weight):
        """x1 [E, dim(in1)], x2 [E, dim(in2)], weight [E, weight_numel] -> [E, dim(out)]."""
        E = x1.shape[0]
        o1, o2, oo = self._offsets(self.in1), self._offsets(self.in2), self._offsets(self.out)
        out = x1.new_zeros(E, irreps_dim(self.out))
        w_off = 0
        for (i1, i2, io), c in zip(self.instr, self.coeff):
            (m1, l1, _), (m2, l2, _), (mo, lo, _) = self.in1[i1], self.in2[i2], self.out[io]
            a = x1[:, o1[i1]:o1[i1] + m1 * (2 * l1 + 1)].reshape(E, m1, 2 * l1 + 1)
            b = x2[:, o2[i2]:o2[i2] + m2 * (2 * l2 + 1)].reshape(E, m2, 2 * l2 + 1)
            w = weight[:, w_off:w_off + m1 * m2 * mo].reshape(E, m1, m2, mo)
            w_off += m1 * m2 * mo
            w3j = wigner_3j(l1, l2, lo).to(x1.dtype)
            r = torch.einsum("euvw,eui,evj,ijk->ewk", w, a, b, w3j)
            out[:, oo[io]:oo[io] + mo * (2 * lo + 1)] += c * r.reshape(E, mo * (2 * lo + 1))
        assert w_off == self.weight_numel
        return out


def linear(sd, prefix, x):
    return torch.nn.functional.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"])


def scatter_mean(src, index, n):
    """torch_scatter.scatter(..., reduce='mean'): rows without an incoming edge stay 0."""
    out = src.new_zeros(n, src.shape[1]).index_add_(0, index, src)
    cnt = torch.zeros(n, dtype=src.dtype).index_add_(0, index, torch.ones(index.shape[0], dtype=src.dtype))
    return out / cnt.clamp_min(1.0)[:, None]


def tp_conv_layer(sd, prefix, tp, node_attr, edge_index, edge_attr, edge_sh, out_nodes=None):
    """TensorProductConvLayer.forward with residual=False, batch_norm=False, reduce='mean' (reference models/gcn_nn.py:
    181-219): tp(node_attr[edge_dst], edge_sh, fc(edge_attr)) averaged over the edges of edge_src."""
    src, dst = edge_index
    h = torch.relu(linear(sd, prefix + ".fc.0", edge_attr))
    w = linear(sd, prefix + ".fc.3", h)
    msg = tp(node_attr[dst], edge_sh, w)
    return scatter_mean(msg, src, node_attr.shape[0] if out_nodes is None else out_nodes)


def gaussian_smearing(dist, start, stop, n):
    """reference gcn_nn.GaussianSmearing (:163-173)."""
    offset = torch.linspace(start, stop, n)
    coeff = -0.5 / (offset[1] - offset[0]).item() ** 2
    return torch.exp(coeff * (dist.view(-1, 1) - offset.view(1, -1)) ** 2)


def make_directed(nbr_list):
    """reference gcn_nn.make_directed (:54-64)."""
    gtr_ij = bool((nbr_list[:, 0] > nbr_list[:, 1]).any())
    gtr_ji = bool((nbr_list[:, 1] > nbr_list[:, 0]).any())
    if gtr_ij and gtr_ji:
        return nbr_list
    return torch.cat([nbr_list, nbr_list.flip(1)], dim=0)


def layer_tps(num_conv_layers=3, ns=12, nv=4, sh_lmax=2):
    seq = irrep_seq(ns, nv)
    return [TensorProduct(seq[min(i, 3)], sh_irreps(sh_lmax), seq[min(i + 1, 3)]) for i in range(num_conv_layers)]


def _graph(z, xyz, nbr_list, max_radius, n_embed, sh_lmax, in_edge_features=4):
    """build_*_conv_graph (reference vae_model.py:164-194 / 268-283)."""
    nb = make_directed(nbr_list)
    edge_attr = torch.cat([z[nb[:, 0]].unsqueeze(-1).float(), z[nb[:, 1]].unsqueeze(-1).float(),
                           torch.zeros(nb.shape[0], in_edge_features)], -1)
    r = xyz[nb[:, 1]] - xyz[nb[:, 0]]
    edge_attr = torch.cat([edge_attr, gaussian_smearing(r.norm(dim=-1), 0.0, max_radius, n_embed)], -1)
    return (nb[:, 0], nb[:, 1]), edge_attr, spherical_harmonics(sh_lmax, r)


def _embed_edges(sd, prefix, x):
    return linear(sd, prefix + ".3", torch.relu(linear(sd, prefix + ".0", x)))


def prior_forward(sd, cg_z, cg_xyz, cg_nbr_list, prefix="", ns=12, nv=4, cg_max_radius=26.0, sh_lmax=2):
    """e3nnPrior.forward (reference vae_model.py:245-266) -> (H_mu, H_sigma) [n_cg, 36]."""
    tps = layer_tps(3, ns, nv, sh_lmax)
    (src, dst), edge_attr, sh = _graph(cg_z, cg_xyz, cg_nbr_list, cg_max_radius, 8, sh_lmax)
    h = sd[prefix + "cg_node_embedding.weight"][cg_z.long()]
    e = _embed_edges(sd, prefix + "cg_edge_embedding", edge_attr)
    for l, tp in enumerate(tps):
        ea = torch.cat([e, h[src, :ns], h[dst, :ns]], -1)
        upd = tp_conv_layer(sd, f"{prefix}cg_conv_layers.{l}", tp, h, (src, dst), ea, sh)
        h = torch.nn.functional.pad(h, (0, upd.shape[-1] - h.shape[-1])) + upd
    mu = linear(sd, prefix + "mu.2", torch.tanh(linear(sd, prefix + "mu.0", h)))
    logvar = linear(sd, prefix + "sigma.2", torch.tanh(linear(sd, prefix + "sigma.0", h)))
    return mu, 1e-9 + torch.exp(logvar / 2)


def encoder_forward(sd, z, xyz, cg_z, cg_xyz, mapping, nbr_list, cg_nbr_list, prefix="", ns=12, nv=4,
                    atom_max_radius=14.0, cg_max_radius=26.0, cross_max_distance=26.0, sh_lmax=2):
    """e3nnEncoder.forward (reference vae_model.py:109-162) -> [n_cg, 36]."""
    tps = layer_tps(3, ns, nv, sh_lmax)
    (a_src, a_dst), a_edge, a_sh = _graph(z, xyz, nbr_list, atom_max_radius, 8, sh_lmax)
    (c_src, c_dst), c_edge, c_sh = _graph(cg_z, cg_xyz, cg_nbr_list, cg_max_radius, 8, sh_lmax)
    ha = sd[prefix + "atom_node_embedding.weight"][z.long()]
    hc = sd[prefix + "cg_node_embedding.weight"][cg_z.long()]
    ea = _embed_edges(sd, prefix + "atom_edge_embedding", a_edge)
    ec = _embed_edges(sd, prefix + "cg_edge_embedding", c_edge)
    # cross graph: every atom <-> its CG bead (vae_model.py:196-201)
    x_atom, x_cg = torch.arange(mapping.shape[0]), mapping
    r = xyz - cg_xyz[mapping]
    ex = _embed_edges(sd, prefix + "cross_edge_embedding", gaussian_smearing(r.norm(dim=-1), 0.0, cross_max_distance, 8))
    x_sh = spherical_harmonics(sh_lmax, r)
    n_layers = len(tps)
    for l, tp in enumerate(tps):
        a_attr = torch.cat([ea, ha[a_src, :ns], ha[a_dst, :ns]], -1)
        a_intra = tp_conv_layer(sd, f"{prefix}atom_conv_layers.{l}", tp, ha, (a_src, a_dst), a_attr, a_sh)
        x_attr = torch.cat([ex, ha[x_atom, :ns], hc[x_cg, :ns]], -1)
        a_inter = tp_conv_layer(sd, f"{prefix}cg_to_atom_conv_layers.{l}", tp, hc, (x_atom, x_cg), x_attr, x_sh,
                                out_nodes=ha.shape[0])
        if l != n_layers - 1:
            c_attr = torch.cat([ec, hc[c_src, :ns], hc[c_dst, :ns]], -1)
            c_intra = tp_conv_layer(sd, f"{prefix}cg_conv_layers.{l}", tp, hc, (c_src, c_dst), c_attr, c_sh)
            c_inter = tp_conv_layer(sd, f"{prefix}atom_to_cg_conv_layers.{l}", tp, ha, (x_cg, x_atom), x_attr, x_sh,
                                    out_nodes=hc.shape[0])
        ha = torch.nn.functional.pad(ha, (0, a_intra.shape[-1] - ha.shape[-1])) + a_intra + a_inter
        if l != n_layers - 1:
            hc = torch.nn.functional.pad(hc, (0, c_intra.shape[-1] - hc.shape[-1])) + c_intra + c_inter
    node = torch.cat([ha, hc[mapping]], -1)                                         # 48 + 36 = 84
    node = scatter_mean(node, mapping, hc.shape[0])
    return linear(sd, prefix + "dense.2", torch.tanh(linear(sd, prefix + "dense.0", node)))
