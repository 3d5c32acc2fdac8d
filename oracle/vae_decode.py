"""De-normalise -> VQ lookup -> IC decoder -> internal coordinates to Cartesian, on CPU
(TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows reference utils/dataset_module.py:230-256 (get_norm_feature, norm_in=False),
utils/vq_module.py:56-71 (nearest code), models/vae_model.py:830-838, :759-764 (latent_decode,
decoder), :375-412 / :467-503 (IC_Decoder_angle / IC_Decoder), models/gcn_nn.py:54-70, :222-271,
:316-338, :372-380 (make_directed, preprocess_r, radial basis, envelope, message) and
utils/utils_ic.py:197-268 (rotation_matrix, add_atom_to_xyz, ic_to_xyz).
"""
import math

import torch
import torch.nn.functional as F


def denormalise(x, mean, std):
    """dataset_module.py:253: x * std + mean."""
    return x * std + mean


def vq_lookup(z, codebook):
    """vq_module.py:61-71: argmin of |z|^2 + |e|^2 - 2 z.e, first index on ties."""
    zf = z.reshape(-1, codebook.shape[1])
    d = (torch.sum(zf ** 2, dim=1, keepdim=True) + torch.sum(codebook ** 2, dim=1)
         - 2.0 * torch.einsum("bd,nd->bn", zf, codebook))
    idx = torch.argmin(d, dim=1)
    return F.embedding(idx, codebook).view(z.shape), idx


def codebook_of(sd):
    if "quantize.embeddings" in sd:
        return sd["quantize.embeddings"]
    return sd["quantize._codebook.embed"][0]


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def _swish(x):
    return x * torch.sigmoid(x)


def _seq(sd, p, x):
    """nn.Sequential(swish, Linear(.1), swish, Linear(.3))."""
    return _lin(sd, f"{p}.3", _swish(_lin(sd, f"{p}.1", _swish(x))))


def ic_decode(sd, z_q_flat, cg_z, cg_xyz, nbr_undirected, angle=False, p="equivaraintconv", latent_is_state=False):
    """z_q_flat [M,3] (restore_shape'd), cg_z [M], cg_xyz [M,3], nbr [E,2] j>i -> ic [M,13,3].
    latent_is_state: the C2 model (GenZProt.decoder, vae_model.py:556-561) hands its 36-wide latent to the IC decoder
    as it is - no map_out."""
    S = z_q_flat if latent_is_state else _lin(sd, "map_out", z_q_flat)  # vae_model.py:762
    nb = nbr_undirected
    gtr_ij = bool((nb[:, 0] > nb[:, 1]).any())
    gtr_ji = bool((nb[:, 1] > nb[:, 0]).any())
    if not (gtr_ij and gtr_ji):                                         # gcn_nn.py:54-64
        nb = torch.cat([nb, nb.flip(1)], dim=0)
    r_ij = cg_xyz[nb[:, 1]] - cg_xyz[nb[:, 0]]
    dist = ((r_ij ** 2 + 1e-8).sum(-1)) ** 0.5                           # gcn_nn.py:66-70
    bb_dist = F.embedding(cg_z, sd[f"{p}.backbone_dist.weight"]).unsqueeze(-1)
    sc_dist = F.embedding(cg_z, sd[f"{p}.sidechain_dist.weight"]).unsqueeze(-1)
    S = torch.cat([S, F.embedding(cg_z, sd[f"{p}.res_embed.weight"])], dim=-1)
    n = torch.arange(1, 16).float()
    coef = n * math.pi / 21.0
    sd_ = dist.unsqueeze(-1)
    num = torch.where(sd_ == 0, coef, torch.sin(coef * sd_))
    den = torch.where(sd_ == 0, torch.tensor(1.0), sd_)
    rbf = torch.where(sd_ >= 21.0, torch.tensor(0.0), num / den)        # gcn_nn.py:231-255
    env = 0.5 * (torch.cos(math.pi * dist / 21.0) + 1)
    env = torch.where(dist >= 21.0, torch.zeros_like(env), env)         # gcn_nn.py:265-271
    for i in range(4):
        mp = f"{p}.message_blocks.{i}"
        phi = _lin(sd, f"{mp}.inv_dense.1", _swish(_lin(sd, f"{mp}.inv_dense.0", S)))[nb[:, 1]]
        w_s = _lin(sd, f"{mp}.dist_embed.block.1", rbf) * env.reshape(-1, 1)
        v = torch.zeros_like(S).index_add_(0, nb[:, 0], phi * w_s)     # scatter_add
        S = S + _seq(sd, f"{p}.dense_blocks.{i}", v)
    bb_angle = _seq(sd, f"{p}.backbone_angle", S)
    bb_tors = _seq(sd, f"{p}.backbone_torsion", torch.cat([S, bb_angle], dim=-1))
    if angle:                                                           # vae_model.py:403-407
        sc_angle = _seq(sd, f"{p}.sidechain_angle", S)
        T = torch.cat([S, sc_angle], dim=-1)
    else:                                                               # vae_model.py:475,496-498
        sc_angle = F.embedding(cg_z, sd[f"{p}.sidechain_angle.weight"])
        T = S
    for i in range(4):
        T = T + _seq(sd, f"{p}.sidechain_torsion_blocks.{i}", T)
    sc_tors = _seq(sd, f"{p}.final_torsion", T)
    ic_bb = torch.cat([bb_dist, bb_angle.unsqueeze(-1), bb_tors.unsqueeze(-1)], dim=-1)
    ic_sc = torch.cat([sc_dist, sc_angle.unsqueeze(-1), sc_tors.unsqueeze(-1)], dim=-1)
    return torch.cat([ic_bb, ic_sc], dim=-2)


def latent_decode(sd, latent, batch, angle=False):
    """vae_model.py:830-838: latent [B,L,3] (already de-normalised) -> (idx, ic [B*L,13,3])."""
    z_q, idx = vq_lookup(latent, codebook_of(sd))
    num = batch["num_CGs"].tolist()
    flat = torch.cat([z_q[b, :n] for b, n in enumerate(num)], dim=0)    # gcn_nn.py:45-52
    ic = ic_decode(sd, flat, batch["CG_nxyz"][:, 0].long(), batch["CG_nxyz"][:, 1:],
                   batch["CG_nbr_list"], angle=angle)
    return idx, ic


# ----------------------------------------------------------------------------------------------
def _rotation(axis, angle):
    """utils_ic.py:197-210 (Euler-Rodrigues)."""
    axis = axis / torch.sqrt((axis * axis).sum(-1)).unsqueeze(-1)
    a = torch.cos(angle / 2).squeeze(-1)
    res = -axis * torch.sin(angle / 2)
    b, c, d = res[..., 0], res[..., 1], res[..., 2]
    rx = torch.stack((a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)), -1)
    ry = torch.stack((2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)), -1)
    rz = torch.stack((2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c), -1)
    return torch.stack((rx, ry, rz), -2)


def _place(ic, atom1, atom2, atom3):
    """utils_ic.py:213-239."""
    dist, ang, dih = ic[..., 0:1], ic[..., 1:2], ic[..., 2:3]
    a = atom2 - atom1
    b = atom2 - atom3
    a = torch.where(a == 0.0, a + 1e-8, a)
    b = torch.where(b == 0.0, b + 1e-8, b)
    d = torch.absolute(dist) * a / torch.sqrt((a * a).sum(-1)).unsqueeze(-1)
    normal = torch.cross(a, b, dim=-1)
    d = torch.matmul(_rotation(normal, ang), d.unsqueeze(-1))
    d = torch.matmul(_rotation(a, dih), d).squeeze(-1)
    return atom1 + d


def ic_to_xyz(og_cg_nxyz, ic, info):
    """utils_ic.py:242-268.  og_cg_nxyz [B,L+2,4], ic [B,L,13,3] -> [B,n_atoms,3].
    (Restated without the reference's .squeeze(), which only changes shapes at B=1 or L=1.)"""
    permute, atom_idx, orders = info
    ca = og_cg_nxyz[:, :, 1:]
    mid, prv, nxt = ca[:, 1:-1], ca[:, :-2], ca[:, 2:]
    N = _place(ic[:, :, 0], mid, prv, nxt)
    C = _place(ic[:, :, 1], mid, nxt, prv)
    O = _place(ic[:, :, 2], C, mid, N)
    atoms = torch.stack((O, N, C, mid), dim=2)
    B = ca.shape[0]
    for i in range(10):
        def pick(col):
            ix = orders[i, :, col].reshape(1, -1, 1, 1).repeat(B, 1, 1, 3)
            return torch.gather(atoms, 2, ix)[:, :, 0]
        new = _place(ic[:, :, 3 + i], pick(2), pick(1), pick(0))
        atoms = torch.cat([atoms, new.unsqueeze(2)], dim=2)
    return atoms.reshape(B, -1, 3)[:, atom_idx, :][:, permute, :]
