"""ctypes binding of libcodlad_hip.so (include/codlad_hip.h).

There is no CPU fallback: if the library is missing or a call fails, this raises.
`import torch` happens first so that the library's libamdhip64.so.7 dependency resolves to the
HIP runtime PyTorch already loaded (same SONAME) - streams and device pointers are then shared.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede the dlopen below)

_HERE = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION = 13   # CODLAD_ABI_VERSION of include/codlad_hip.h this binding was written against
# CODLAD_HIP_LIB: an alternative build of the same ABI (A/B measurements, tools/ablate_edge.py)
LIB_PATH = os.environ.get("CODLAD_HIP_LIB") or os.path.join(_HERE, "libcodlad_hip.so")

P = C.c_void_p


class EncLayer(C.Structure):
    _fields_ = [(n, P) for n in ("W1e", "W2", "W3", "W11e", "W12", "W13", "W1a", "W1c", "W11a", "W11c")] + \
               [("Win", P * 4), ("Wout", P * 4)] + \
               [(n, P) for n in ("b1", "b2", "b3", "b11", "b12", "b13", "b_in", "b_out")]


class DecLayer(C.Structure):
    _fields_ = [(n, P) for n in ("W1e", "W2", "W3", "W1a", "W1v", "TS")] + \
               [("Win", P * 4), ("Wout", P * 4)] + \
               [(n, P) for n in ("b1", "b2", "b3", "b_in", "b_out")]


class EncLayerH(C.Structure):
    _fields_ = [(n, P) for n in ("W1e", "W2", "W3", "W11e", "W12", "W13", "W1a", "W1c", "W11a", "W11c")] + \
               [("Win", P * 4), ("Wout", P * 4)] + \
               [(n, P) for n in ("b1", "b2", "b3", "b11", "b12", "b13", "b_in", "b_out")] + \
               [(n, C.c_int) for n in ("e1", "e2", "e3", "e11", "e12", "e13", "e_in", "e_out")]


class DecLayerH(C.Structure):
    _fields_ = [(n, P) for n in ("W1e", "W2", "W3", "W1a", "W1v")] + [("Win", P * 4), ("Wout", P * 4)] + \
               [(n, P) for n in ("TS", "b1", "b2", "b3", "b_in", "b_out")] + \
               [(n, C.c_int) for n in ("e1", "e2", "e3", "e_in", "e_out")]


class DenoiserWeights(C.Structure):
    _fields_ = [(n, P) for n in ("freqs", "rbf_mu", "t_w0", "t_b0", "t_w2", "t_b2")] + \
               [("ada_w", P * 7), ("ada_b", P * 7)] + \
               [(n, P) for n in ("x_in_w", "x_in_b", "pos_w", "pos_b", "edge_wT", "norm_w", "norm_b",
                                 "We_wT", "We_b", "out_w", "out_b")] + \
               [("enc", EncLayer * 3), ("dec", DecLayer * 3), ("precision", C.c_int),
                ("enc_h", EncLayerH * 3), ("dec_h", DecLayerH * 3), ("self_condition", C.c_int), ("out_dim", C.c_int)]


class Workspace(C.Structure):
    _fields_ = [(n, P) for n in ("hV", "hVenc", "S", "PQ", "hE", "status", "tile_list")] + [("n_tiles", C.c_int32)]


class DecoderWeights(C.Structure):
    _fields_ = [("angle", C.c_int), ("map_out_w", P), ("map_out_b", P), ("res_embed", P)] + \
               [(n, P * 4) for n in ("inv0_w", "inv0_b", "inv1_w", "inv1_b", "dist_w", "dist_b",
                                     "dense1_w", "dense1_b", "dense3_w", "dense3_b")] + \
               [(n, P) for n in ("bb_dist", "sc_dist", "bb_ang1_w", "bb_ang1_b", "bb_ang3_w", "bb_ang3_b",
                                 "sc_angle_emb", "sc_ang1_w", "sc_ang1_b", "sc_ang3_w", "sc_ang3_b",
                                 "bb_tor1_w", "bb_tor1_b", "bb_tor3_w", "bb_tor3_b")] + \
               [(n, P * 4) for n in ("tor1_w", "tor1_b", "tor3_w", "tor3_b")] + \
               [(n, P) for n in ("fin1_w", "fin1_b", "fin3_w", "fin3_b")]


class TpConvArgs(C.Structure):
    _fields_ = [("ptr", P), ("snd", P), ("n_recv", C.c_int32), ("xyz_recv", P), ("xyz_snd", P), ("typ_recv", P),
                ("typ_snd", P), ("r_sign", C.c_float), ("smear_stop", C.c_float), ("emb0_w", P), ("emb0_b", P),
                ("emb3_w", P), ("emb3_b", P), ("emb_in", C.c_int32), ("h_recv", P), ("d_recv", C.c_int32), ("h_snd", P),
                ("d_snd", C.c_int32), ("attr_recv_first", C.c_int32), ("fc0_w", P), ("fc0_b", P), ("fc3_w", P),
                ("fc3_b", P), ("depth", C.c_int32), ("out", P), ("accumulate", C.c_int32), ("group", C.c_int32),
                ("packed", P)]


class XyzGroup(C.Structure):
    _fields_ = [("ca_full", P), ("ic", P), ("orders", P), ("slot_to_out", P), ("xyz_out", P),
                ("B", C.c_int32), ("L", C.c_int32), ("n_atoms", C.c_int32), ("first_row", C.c_int32)]


class MetricInputs(C.Structure):
    _fields_ = [("xyz_recon", P), ("xyz", P), ("n_atoms", C.c_int64),
                ("edge_list", P), ("n_edges", C.c_int64), ("clash_list", P), ("n_clash", C.c_int64),
                ("bb_NO_list", P), ("n_bb", C.c_int64), ("interaction_list", P), ("n_inter", C.c_int64),
                ("pi_pi_list", P), ("n_pipi", C.c_int64),
                ("ic", P), ("ic_recon", P), ("ic_mask", P), ("n_ic", C.c_int64)]


_SIGS = {
    "codlad_abi_version": (C.c_int, []),
    "codlad_probe_edge_launches": (C.c_int, [C.c_int]),
    "codlad_probe_read": (C.c_int, [C.c_int, C.POINTER(C.c_double)]),
    "codlad_metrics_scratch_bytes": (C.c_int, []),
    "codlad_eval_metrics": (C.c_int, [C.POINTER(MetricInputs), P, P, P]),
    "codlad_bond_graph_counts": (C.c_int, [P, P, P, P, P, C.c_int, C.c_int, C.c_float, P, P]),
    "codlad_last_error": (C.c_char_p, []),
    "codlad_struct_sizes": (None, [C.POINTER(C.c_int)]),
    "codlad_pack_block_host": (None, [P, C.c_int, C.c_float, P]),
    "codlad_features_prepass": (C.c_int, [C.POINTER(DenoiserWeights), P, P, C.c_int, C.c_int, P, P, P]),
    "codlad_step_mods": (C.c_int, [C.POINTER(DenoiserWeights), P, C.c_int, P, P]),
    "codlad_step_mods_f": (C.c_int, [C.POINTER(DenoiserWeights), P, C.c_int, P, P]),
    "codlad_ode_combine": (C.c_int, [P, P, P, C.c_int, C.c_float, C.c_size_t, P, P]),
    "codlad_layer0_edge_terms": (C.c_int, [C.POINTER(DenoiserWeights), P, C.c_int, P, P, P]),
    "codlad_denoiser_forward": (C.c_int, [C.POINTER(DenoiserWeights), P, C.c_int, P, P, P, C.c_int, P, P, P, P,
                                          C.POINTER(Workspace), P]),
    "codlad_status_check": (C.c_int, [P, P]),
    "codlad_set_option": (C.c_int, [C.c_int, C.c_int]),
    "codlad_ddpm_update": (C.c_int, [P, P, P, P, C.c_int, P, P, P]),
    "codlad_sample_loop": (C.c_int, [C.POINTER(DenoiserWeights), P, C.c_int, P, P, P, C.c_int, P, P, P, P, P,
                                     C.c_int, C.POINTER(Workspace), P]),
    "codlad_vq_lookup": (C.c_int, [P, C.c_int, P, P, P, C.c_int, P, P, P, P]),
    "codlad_ic_decode": (C.c_int, [C.POINTER(DecoderWeights), P, P, P, P, P, C.c_int, P, P, P]),
    "codlad_cg_graph": (C.c_int, [P, P, C.c_int, C.c_float, P, P, P, P]),
    "codlad_ic_to_xyz": (C.c_int, [P, P, P, P, C.c_int, C.c_int, C.c_int, P, P]),
    "codlad_ic_to_xyz_groups": (C.c_int, [P, C.c_int, C.c_int, P]),
    "codlad_xyz_to_ic": (C.c_int, [P, C.c_int, C.c_int, P, C.c_int, P, P]),
    "codlad_tp_conv_image_bytes": (C.c_int, [C.c_int]),
    "codlad_tp_conv_pack": (C.c_int, [P, P, P]),
    "codlad_receiver_csr": (C.c_int, [P, C.c_int, C.c_int, C.c_int, P, P, P, P]),
    "codlad_bench_edge_launch": (C.c_int, [C.POINTER(DenoiserWeights), P, C.c_int, P, P, P,
                                           C.POINTER(Workspace), C.c_int, C.c_int, P]),
    "codlad_tp_conv": (C.c_int, [C.POINTER(TpConvArgs), P]),
    "codlad_tp_conv_args_size": (C.c_int, []),
    "codlad_mlp_rows": (C.c_int, [P, C.c_int, C.c_int, P, P, C.c_int, P, P, C.c_int, C.c_int, C.c_int, P, P]),
    "codlad_bead_mean": (C.c_int, [P, P, P, P, C.c_int, P, P]),
    "codlad_embed_rows": (C.c_int, [P, P, C.c_int, C.c_int, P, P]),
    "codlad_selftest_gemm128": (C.c_int, [P, P, P, C.c_int, C.c_int, P, P]),
    "codlad_selftest_gemm128_h": (C.c_int, [P, P, P, C.c_int, C.c_int, C.c_int, P, P]),
}

_lib = None


def lib():
    """The loaded library; raises if it has not been built (python -m codlad_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is the only implementation of this path "
                "(build it with `python -m codlad_amd.build`)")
        handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if handle.codlad_abi_version() != ABI_VERSION:
            raise RuntimeError("libcodlad_hip.so ABI version mismatch")
        _lib = handle
    return _lib


OPT_NODEQ_MAX_TILES, OPT_EDGE_TILE_MAX_NODES, OPT_DEC_EDGE_VARIANT, OPT_TP_CONV_VARIANT, OPT_EDGE_UPD_VARIANT, OPT_EDGE_CUS, OPT_EDGE_WIDE_MAX_TILES, OPT_NODE_QUAD_MAX_TILES = 0, 1, 3, 4, 5, 6, 7, 2   # CODLAD_OPT_* of include/codlad_hip.h


def set_option(option, value):
    """Tuning switch of the library (speed only; include/codlad_hip.h codlad_set_option)."""
    check(lib().codlad_set_option(option, value), "codlad_set_option")


def exported_symbols():
    return list(_SIGS)


def check(rc, what):
    if rc != 0:
        msg = lib().codlad_last_error().decode() or "unknown error"
        raise RuntimeError(f"{what} failed (rc={rc}): {msg}")


class _TensorPtr(C.c_void_p):
    """c_void_p that keeps its tensor alive for as long as the argument object lives, i.e. until
    the foreign call has been enqueued.  (A bare integer would let a temporary such as
    `x.contiguous()` be freed - and its block reused by the next temporary - before the launch.)"""


def ptr(t):
    """Device (or host) pointer of a contiguous tensor, None -> NULL."""
    if t is None:
        return None
    assert t.is_contiguous(), "non-contiguous tensor handed to the C ABI"
    p = _TensorPtr(t.data_ptr())
    p._keepalive = t
    return p


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
