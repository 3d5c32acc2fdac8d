"""mpnn_diffusion denoiser forward on CPU (TEST INFRASTRUCTURE, see oracle/__init__.py).

Functional restatement over a reference-layout state_dict (key names: SURVEY.md §8b).
Follows reference models/latent_model.py:37-75 (timestep embedding), :175-268 (forward) and
models/protein_mpnn_utils.py:97-116 (gathers), :208-330 (enc/dec layers, FFN),
:333-344 (positional encodings), :347-523 (CA features).
"""
import math

import torch
import torch.nn.functional as F


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def gather_nodes(nodes, E_idx):
    """protein_mpnn_utils.py:103-111: [N,L,C] at [N,L,K] -> [N,L,K,C]."""
    N, L, K = E_idx.shape
    flat = E_idx.reshape(N, L * K, 1).expand(-1, -1, nodes.shape[-1])
    return torch.gather(nodes, 1, flat).reshape(N, L, K, -1)


def gather_edges(edges, E_idx):
    """protein_mpnn_utils.py:97-101: [N,L,L,C] at [N,L,K] -> [N,L,K,C]."""
    return torch.gather(edges, 2, E_idx.unsqueeze(-1).expand(-1, -1, -1, edges.shape[-1]))


def timestep_embedding(t, dim=256, max_period=10000):
    """latent_model.py:51-70."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def t_embed(sd, t):
    """latent_model.py:72-75: Linear -> SiLU -> Linear."""
    h = _lin(sd, "t_embedder.mlp.0", timestep_embedding(t))
    return _lin(sd, "t_embedder.mlp.2", F.silu(h))


# ----------------------------------------------------------------------------------------------
# CA features
# ----------------------------------------------------------------------------------------------
def knn(X, mask, top_k=64, eps=1e-6):
    """protein_mpnn_utils.py:447-459."""
    mask_2D = mask[:, None, :] * mask[:, :, None]
    dX = X[:, None, :, :] - X[:, :, None, :]
    D = mask_2D * torch.sqrt(torch.sum(dX ** 2, 3) + eps)
    D_max, _ = torch.max(D, -1, keepdim=True)
    D_adjust = D + (1.0 - mask_2D) * D_max
    return torch.topk(D_adjust, min(top_k, X.shape[1]), dim=-1, largest=False)


def _rbf(D):
    """protein_mpnn_utils.py:461-470: 16 Gaussians, centres linspace(2, 22), width 1.25."""
    mu = torch.linspace(2.0, 22.0, 16).view(1, 1, 1, -1)
    return torch.exp(-((D.unsqueeze(-1) - mu) / ((22.0 - 2.0) / 16)) ** 2)


def _pair_rbf(A, B, E_idx):
    """protein_mpnn_utils.py:472-476."""
    D = torch.sqrt(torch.sum((A[:, :, None, :] - B[:, None, :, :]) ** 2, -1) + 1e-6)
    return _rbf(gather_edges(D[..., None], E_idx)[..., 0])


def _quaternions(R):
    """protein_mpnn_utils.py:369-395."""
    diag = torch.diagonal(R, dim1=-2, dim2=-1)
    Rxx, Ryy, Rzz = diag.unbind(-1)
    mag = 0.5 * torch.sqrt(torch.abs(1 + torch.stack([Rxx - Ryy - Rzz, -Rxx + Ryy - Rzz,
                                                      -Rxx - Ryy + Rzz], -1)))
    sg = torch.sign(torch.stack([R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0],
                                 R[..., 1, 0] - R[..., 0, 1]], -1))
    w = torch.sqrt(F.relu(1 + diag.sum(-1, keepdim=True))) / 2.0
    return F.normalize(torch.cat((sg * mag, w), -1), dim=-1)


def orientation_features(X, E_idx):
    """protein_mpnn_utils.py:397-443 (only O_features is consumed by the forward)."""
    dX = X[:, 1:, :] - X[:, :-1, :]
    n = torch.norm(dX, dim=-1)
    dX = dX * ((3.6 < n) & (n < 4.0))[:, :, None]
    U = F.normalize(dX, dim=-1)
    u_2, u_1 = U[:, :-2, :], U[:, 1:-1, :]
    n_2 = F.normalize(torch.linalg.cross(u_2, u_1), dim=-1)
    o_1 = F.normalize(u_2 - u_1, dim=-1)
    O = torch.stack((o_1, n_2, torch.linalg.cross(o_1, n_2)), dim=2)
    O = F.pad(O.reshape(O.shape[0], O.shape[1], 9), (0, 0, 1, 2), "constant", 0)
    O_nb = gather_nodes(O, E_idx)
    X_nb = gather_nodes(X, E_idx)
    O = O.view(O.shape[0], O.shape[1], 3, 3)
    O_nb = O_nb.view(*O_nb.shape[:3], 3, 3)
    dXn = X_nb - X.unsqueeze(-2)
    dU = F.normalize(torch.matmul(O.unsqueeze(2), dXn.unsqueeze(-1)).squeeze(-1), dim=-1)
    R = torch.matmul(O.unsqueeze(2).transpose(-1, -2), O_nb)
    return torch.cat((dU, _quaternions(R)), dim=-1)


def ca_features(sd, Ca, mask, top_k=64):
    """protein_mpnn_utils.py:478-523 -> E [N,L,K,128] (before W_e), E_idx [N,L,K]."""
    maskf = mask.float()
    D_nb, E_idx = knn(Ca, maskf, top_k)
    Ca_0 = torch.zeros_like(Ca)
    Ca_2 = torch.zeros_like(Ca)
    Ca_0[:, 1:, :] = Ca[:, :-1, :]
    Ca_2[:, :-1, :] = Ca[:, 1:, :]
    O_feat = orientation_features(Ca, E_idx)
    rbf = [_rbf(D_nb)]
    for A, B in ((Ca_0, Ca_0), (Ca_2, Ca_2), (Ca_0, Ca), (Ca_0, Ca_2), (Ca, Ca_0), (Ca, Ca_2),
                 (Ca_2, Ca_0), (Ca_2, Ca)):
        rbf.append(_pair_rbf(A, B, E_idx))
    rbf = torch.cat(rbf, dim=-1)
    L = Ca.shape[1]
    ridx = torch.arange(L)
    offset = (ridx[:, None] - ridx[None, :])[None, :, :, None].expand(Ca.shape[0], -1, -1, -1)
    offset = gather_edges(offset, E_idx)[..., 0]
    d = torch.clip(offset + 32, 0, 64)  # same chain everywhere (latent_model.py:201)
    E_pos = _lin(sd, "features.embeddings.linear", F.one_hot(d, 66).float())
    E = torch.cat((E_pos, rbf, O_feat), -1)
    E = F.linear(E, sd["features.edge_embedding.weight"])
    E = F.layer_norm(E, (E.shape[-1],), sd["features.norm_edges.weight"],
                     sd["features.norm_edges.bias"], 1e-5)
    return E, E_idx


# ----------------------------------------------------------------------------------------------
# layers
# ----------------------------------------------------------------------------------------------
def _ln(x):
    return F.layer_norm(x, (x.shape[-1],), None, None, 1e-6)


def _mod(x, shift, scale):
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


def _mlp3(sd, p, names, x):
    a, b, c = names
    return _lin(sd, f"{p}.{c}", F.gelu(_lin(sd, f"{p}.{b}", F.gelu(_lin(sd, f"{p}.{a}", x)))))


def _ffn(sd, p, x):
    return _lin(sd, f"{p}.dense.W_out", F.gelu(_lin(sd, f"{p}.dense.W_in", x)))


def enc_layer(sd, p, h_V, h_E, E_idx, mask_V, mask_attend, c, scale=30.0):
    """protein_mpnn_utils.py:236-271."""
    sh1, sc1, g1, sh2, sc2, g2, sh3, sc3, g3 = _lin(sd, f"{p}.adaLN_modulation.1",
                                                    F.silu(c)).chunk(9, dim=1)
    K = E_idx.shape[-1]
    h_EV = torch.cat([h_V.unsqueeze(-2).expand(-1, -1, K, -1), h_E, gather_nodes(h_V, E_idx)], -1)
    msg = _mlp3(sd, p, ("W1", "W2", "W3"), h_EV)
    msg = mask_attend.unsqueeze(-1) * msg
    h_V = _ln(h_V + torch.sum(msg, -2) / scale)
    h_V = g1.unsqueeze(1) * _mod(h_V, sh1, sc1)
    h_V = _ln(h_V + _ffn(sd, p, h_V))
    h_V = g2.unsqueeze(1) * _mod(h_V, sh2, sc2)
    h_V = mask_V.unsqueeze(-1) * h_V
    h_EV = torch.cat([h_V.unsqueeze(-2).expand(-1, -1, K, -1), h_E, gather_nodes(h_V, E_idx)], -1)
    msg = _mlp3(sd, p, ("W11", "W12", "W13"), h_EV)
    h_E = _ln(h_E + msg)
    h_E = g3[:, None, None, :] * (h_E * (1 + sc3[:, None, None, :]) + sh3[:, None, None, :])
    return h_V, h_E


def dec_layer(sd, p, h_V, h_ESV, mask_V, c, scale=30.0):
    """protein_mpnn_utils.py:296-318 with mask_attend=None (latent_model.py:262): the sum runs
    over all K neighbours, padded ones included."""
    sh1, sc1, g1, sh2, sc2, g2 = _lin(sd, f"{p}.adaLN_modulation.1", F.silu(c)).chunk(6, dim=1)
    K = h_ESV.shape[-2]
    h_EV = torch.cat([h_V.unsqueeze(-2).expand(-1, -1, K, -1), h_ESV], -1)
    msg = _mlp3(sd, p, ("W1", "W2", "W3"), h_EV)
    h_V = _ln(h_V + torch.sum(msg, -2) / scale)
    h_V = g1.unsqueeze(1) * _mod(h_V, sh1, sc1)
    h_V = _ln(h_V + _ffn(sd, p, h_V))
    h_V = g2.unsqueeze(1) * _mod(h_V, sh2, sc2)
    return mask_V.unsqueeze(-1) * h_V


def final_layer(sd, h_V, c):
    """latent_model.py:31-35."""
    shift, scale = _lin(sd, "W_out.adaLN_modulation.1", F.silu(c)).chunk(2, dim=1)
    return _lin(sd, "W_out.linear", _mod(_ln(h_V), shift, scale))


def forward(sd, x, t, cg_xyz, cg_z, mask, features=None, taps=None, x_self_cond=None):
    """latent_model.py:175-268 for `mpnn_diffusion` (decoder_mask=False, use_seq_in_encoder=True).

    x [N,L,3]; t [N] (already mapped through timestep_map); cg_xyz [N,L,3]; cg_z [N,L] int64;
    mask [N,L] bool.  `features` = (E, E_idx) lets a caller hoist the step-invariant CA
    features; the reference recomputes them every call.  Returns [N,L,6].
    """
    c = t_embed(sd, t)
    maski = mask.int()
    E, E_idx = ca_features(sd, cg_xyz, maski) if features is None else features
    if sd["x_in.weight"].shape[1] == 2 * x.shape[-1]:     # self_condition model, latent_model.py:210-212
        x = torch.cat((torch.zeros_like(x) if x_self_cond is None else x_self_cond, x), dim=-1)
    h_V = _lin(sd, "x_in", x)
    h_E = _lin(sd, "W_e", E)
    if taps is not None:
        taps["E_idx"], taps["h_E0"] = E_idx, h_E
    mask_attend = gather_nodes(maski.unsqueeze(-1), E_idx).squeeze(-1)
    mask_attend = maski.unsqueeze(-1) * mask_attend
    for l in range(3):
        h_V, h_E = enc_layer(sd, f"encoder_layers.{l}", h_V, h_E, E_idx, maski, mask_attend, c)
        if taps is not None:
            taps[f"enc{l}_hV"], taps[f"enc{l}_hE"] = h_V, h_E
    h_S = F.embedding(cg_z, sd["W_s.weight"])
    h_ES = torch.cat([h_E, gather_nodes(h_S, E_idx)], -1)
    h_EXV_enc = torch.cat([h_ES, gather_nodes(h_V, E_idx)], -1)
    for l in range(3):
        h_ESV = torch.cat([h_ES, gather_nodes(h_V, E_idx)], -1) + h_EXV_enc
        h_V = dec_layer(sd, f"decoder_layers.{l}", h_V, h_ESV, maski, c)
        if taps is not None:
            taps[f"dec{l}_hV"] = h_V
    return final_layer(sd, h_V, c)


def batch_to_dense(batch):
    """Split the flat CG_nxyz of a batch dict into padded [N,L,...] tensors + mask
    (reference models/gcn_nn.py:35-43 via latent_model.py:168-188)."""
    num = batch["num_CGs"].tolist()
    z = torch.nn.utils.rnn.pad_sequence(torch.split(batch["CG_nxyz"][:, 0].long(), num),
                                        batch_first=True)
    xyz = torch.nn.utils.rnn.pad_sequence(torch.split(batch["CG_nxyz"][:, 1:], num),
                                          batch_first=True)
    mask = torch.arange(max(num))[None, :] < batch["num_CGs"][:, None]
    return z, xyz, mask
