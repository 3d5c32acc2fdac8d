"""Host-side driver of the HIP sampling path: ragged job tables, workspaces and the calls into
libcodlad_hip.so.  PyTorch is used for device memory and streams only; every computation on the
path is a kernel of the library (there is no CPU or eager fallback).

A *structure* is one CA trace (one frame of one protein); a *sample* is one latent trajectory on a
structure.  Ensemble members of a frame are several samples on the same structure: they share the
k-NN graph and the initial edge embedding h_E0, which is computed once per structure instead of
once per denoiser call as the reference does (reference models/latent_model.py:208).
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .weights import DEFAULT_PRECISION, DenoiserWeights, DecoderWeights

H = 128
KNN = 64
MODS = 6016


def _require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what} must live on the GPU: this path has no CPU implementation")


def edge_rows(blocks, split=False):
    """Edge state as the kernels keep it in HBM ([..., 2 halves, 32 chunks, 32 edges, 4], include/codlad_hip.h)
    -> [..., 64 edges, 128 features].  split: the blocks are h_E0 / h_E of a split-fp16 contraction mode, whose 16-byte
    slots hold fp16 halves (slot 8 b + 4 s + h: the `hi` halves of features 32 b + 16 s + 4 h + {0..3, 8..11}, slot
    8 b + 4 s + 2 + h their `lo` halves); the values returned are hi + lo."""
    lead = blocks.shape[:-4]
    if not split:
        return blocks.permute(*range(len(lead)), -4, -2, -3, -1).reshape(*lead, 64, -1)
    hv = blocks.contiguous().view(torch.float16).view(*lead, 2, 4, 2, 2, 2, 32, 8)      # half, b, s, hi|lo, h, edge, 8
    val = hv[..., 0, :, :, :].float() + hv[..., 1, :, :, :].float()                       # [.., half, b, s, h, edge, 8]
    n = len(lead)
    val = val.permute(*range(n), n, n + 4, n + 1, n + 2, n + 3, n + 5)                    # [.., half, edge, b, s, h, 8]
    out = torch.empty(*lead, 2, 32, 4, 2, 16, dtype=torch.float32, device=blocks.device)  # feature = 32 b + 16 s + (0..15)
    for h in range(2):
        out[..., 4 * h:4 * h + 4] = val[..., h, 0:4]
        out[..., 8 + 4 * h:8 + 4 * h + 4] = val[..., h, 4:8]
    return out.reshape(*lead, 64, 128)


class Structures:
    """Flat structure-node arrays + the step-invariant graph/features of every structure."""

    def __init__(self, xyz_list, z_list, device):
        self.lens = [int(x.shape[0]) for x in xyz_list]
        assert all(L >= 1 for L in self.lens)
        self.offsets = np.concatenate([[0], np.cumsum(self.lens)]).astype(np.int64)
        self.n_snodes = int(self.offsets[-1])
        self.xyz = torch.cat([x.reshape(-1, 3).float() for x in xyz_list]).contiguous().to(device)
        self.z = torch.cat([z.reshape(-1).to(torch.int32) for z in z_list]).contiguous().to(device)
        info = np.empty((self.n_snodes, 2), dtype=np.int32)
        for f, L in enumerate(self.lens):
            info[self.offsets[f]:self.offsets[f + 1], 0] = self.offsets[f]
            info[self.offsets[f]:self.offsets[f + 1], 1] = L
        self.snode_info = torch.from_numpy(info).to(device)
        self.E_idx = None
        self.h_E0 = None
        self.E1 = None        # [2] x edge blocks: hoisted layer-0 edge terms (optional)
        self.features_tag = None   # contraction mode + block exponents E1 was computed with (Denoiser.features_tag)


class Job:
    """Samples on structures: node tables + per-step workspace."""

    def __init__(self, structures, sample_struct, device, edge_state=None):
        """edge_state: a [n_nodes, 64, 128] view to use as this job's edge state instead of a buffer of its own (the
        parts of a split job live in their parent's)."""
        st = structures
        self.structures = st
        self.device = device
        self._parts = {}
        self.sample_struct = [int(s) for s in sample_struct]
        lens = [st.lens[f] for f in self.sample_struct]
        self.sample_lens = lens
        self.sample_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        self.n_nodes = int(self.sample_off[-1])
        info = np.empty((self.n_nodes, 4), dtype=np.int32)
        z_host = st.z.cpu().numpy()
        for s, f in enumerate(self.sample_struct):
            a, b = self.sample_off[s], self.sample_off[s + 1]
            L = lens[s]
            info[a:b, 0] = np.arange(st.offsets[f], st.offsets[f] + L)
            info[a:b, 1] = a
            info[a:b, 2] = min(KNN, L)
            info[a:b, 3] = z_host[st.offsets[f]:st.offsets[f] + L]
        assert info[:, 3].min() >= 0 and info[:, 3].max() < 30, "residue type outside W_s vocabulary"
        self.node_info = torch.from_numpy(info).to(device)
        n = self.n_nodes
        f32 = dict(dtype=torch.float32, device=device)
        self.hV = torch.empty(n, H, **f32)
        self.hVenc = torch.empty(n, H, **f32)
        self.S = torch.empty(4, n, H, **f32)      # planes 1-3: per-half, per-lane-half partial sums of small jobs
        self.PQ = torch.empty(4, n, H, **f32)
        self.hE = torch.empty(n, KNN, H, **f32) if edge_state is None else edge_state
        assert tuple(self.hE.shape) == (n, KNN, H) and self.hE.is_contiguous()
        self.status = torch.zeros(1, dtype=torch.int32, device=device)   # sticky flags (CODLAD_STATUS_*)
        # non-empty 32-edge tiles {node, half}: small jobs deal the edge kernels' work out per tile
        halves = np.where(info[:, 2] > 32, 2, 1)
        tiles = np.empty((int(halves.sum()), 2), dtype=np.int32)
        tiles[:, 0] = np.repeat(np.arange(n, dtype=np.int32), halves)
        first = np.cumsum(halves) - halves
        tiles[:, 1] = np.arange(tiles.shape[0], dtype=np.int32) - np.repeat(first, halves).astype(np.int32)
        self.tile_list = torch.from_numpy(tiles).to(device)
        ws = _lib.Workspace()
        ws.hV, ws.hVenc, ws.S, ws.PQ, ws.hE, ws.status, ws.tile_list = (
            _lib.ptr(t) for t in (self.hV, self.hVenc, self.S, self.PQ, self.hE, self.status, self.tile_list))
        ws.n_tiles = tiles.shape[0]
        self.ws = ws

    def workspace_bytes(self):
        return 4 * (self.hV.numel() * 2 + self.S.numel() + self.PQ.numel() + self.hE.numel())

    def parts(self, k=2):
        """This job's samples dealt alternately into k independent jobs (the same mix of lengths in each) -> [(job, node
        indices of its samples in this job)].  The parts keep their edge state in slices of this job's buffer (32 KB per
        node: nothing else of a workspace is large), so a job and its parts are never in flight together."""
        if k not in self._parts:
            subs, start = [], 0
            for p in range(k):
                members = list(range(p, len(self.sample_struct), k))
                n = int(sum(self.sample_lens[m] for m in members))
                sub = Job(self.structures, [self.sample_struct[m] for m in members], self.device,
                          edge_state=self.hE[start:start + n])
                idx = np.concatenate([np.arange(self.sample_off[m], self.sample_off[m + 1]) for m in members])
                subs.append((sub, torch.from_numpy(idx).to(self.device)))
                start += n
            self._parts[k] = subs
        return self._parts[k]


class Denoiser:
    """mpnn_diffusion on the GPU."""

    def __init__(self, state_dict, device, precision=DEFAULT_PRECISION, block_exponents=True, self_condition=False,
                 out_dim=6):
        """state_dict None: the layout of a model with the given flags and NO weights - what a rank that loads no
        checkpoint starts from; `parallel.broadcast_weights(den.weights)` then fills it with rank 0's."""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("codlad_amd runs on an MI355X only; no CPU path exists")
        self.lib = _lib.lib()
        if state_dict is None:
            self.weights = DenoiserWeights.empty(self.device, self_condition, out_dim, precision)
        else:
            self.weights = DenoiserWeights(state_dict, self.device, precision, block_exponents=block_exponents)
        self._mods_cache = {}
        self._mods_generation = self.weights.generation

    # -- step-invariant part -------------------------------------------------------------------
    def new_structures(self, xyz_list, z_list, hoist_layer0=True):
        """Tables of a set of structures uploaded and the buffers of their step-invariant part
        allocated; `compute_features` fills them (split so that a caller can time the kernels apart
        from the uploads, bench.py)."""
        st = Structures(xyz_list, z_list, self.device)
        st.E_idx = torch.empty(st.n_snodes, KNN, dtype=torch.int32, device=self.device)
        st.h_E0 = torch.empty(st.n_snodes, 2, H // 4, KNN // 2, 4, dtype=torch.float32, device=self.device)
        if hoist_layer0:
            st.E1 = torch.empty(2, st.n_snodes, 2, H // 4, KNN // 2, 4, dtype=torch.float32, device=self.device)
        return st

    def compute_features(self, st):
        """k-NN graph + h_E0 per structure (codlad_features_prepass), and - when the structures were
        made with hoist_layer0 - the two layer-0 contractions of h_E0, which are the same in every
        step and for every ensemble member (costs 2x the h_E0 memory).  Only enqueues kernels."""
        rc = self.lib.codlad_features_prepass(C.byref(self.weights.struct), _lib.ptr(st.xyz),
                                              _lib.ptr(st.snode_info), st.n_snodes, max(st.lens),
                                              _lib.ptr(st.E_idx), _lib.ptr(st.h_E0),
                                              _lib.stream_ptr(self.device))
        _lib.check(rc, "codlad_features_prepass")
        if st.E1 is not None:
            rc = self.lib.codlad_layer0_edge_terms(C.byref(self.weights.struct), _lib.ptr(st.snode_info),
                                                   st.n_snodes, _lib.ptr(st.h_E0), _lib.ptr(st.E1),
                                                   _lib.stream_ptr(self.device))
            _lib.check(rc, "codlad_layer0_edge_terms")
        st.features_tag = self.features_tag()
        return st

    @property
    def split_edge_state(self):
        """True when h_E0 / h_E are kept as fp16 hi / lo halves (the split-fp16 modes), see edge_rows."""
        return self.weights.precision != "f32"

    def features_tag(self):
        """What the hoisted layer-0 terms E1 depend on besides the structure: in the split-fp16 modes they carry
        encoder layer 0's block exponents (E1[0] = 2^e1 W1e h_E0, E1[1] = 2^e11 W11e h_E0), in the fp32 mode none."""
        # generation: h_E0 and E1 are functions of the weights too, so a broadcast (rebind) makes them stale
        if self.weights.precision == "f32":
            return ("f32", self.weights.generation)
        ex = self.weights.exponents["enc0"]
        return ("split", self.weights.precision, ex["e1"], ex["e11"], self.weights.generation)

    def _fresh_features(self, st):
        if st.features_tag != self.features_tag():     # e.g. set_precision() after the structures were prepared
            self.compute_features(st)

    def prepare_structures(self, xyz_list, z_list, hoist_layer0=True):
        return self.compute_features(self.new_structures(xyz_list, z_list, hoist_layer0))

    def make_job(self, structures, sample_struct):
        return Job(structures, sample_struct, self.device)

    def step_mods(self, t_values, refresh=False):
        """[len(t_values), 6016] adaLN modulation vectors; cached per timestep list (refresh: run the
        kernel again even if cached).  Integer timesteps (diffusion) or fractional ones (flow matching: any
        non-integer value switches the whole list to the float entry point)."""
        fractional = any(float(t) != int(t) for t in t_values)
        key = tuple(float(t) for t in t_values) if fractional else tuple(int(t) for t in t_values)
        if self._mods_generation != self.weights.generation:      # the weights changed under the cache (broadcast)
            self._mods_cache.clear()
            self._mods_generation = self.weights.generation
        if refresh or key not in self._mods_cache:
            mods = torch.empty(len(key), MODS, dtype=torch.float32, device=self.device)
            if fractional:
                tv = torch.tensor(key, dtype=torch.float32, device=self.device)
                rc = self.lib.codlad_step_mods_f(C.byref(self.weights.struct), _lib.ptr(tv), len(key),
                                                 _lib.ptr(mods), _lib.stream_ptr(self.device))
            else:
                tv = torch.tensor(key, dtype=torch.int64, device=self.device)
                rc = self.lib.codlad_step_mods(C.byref(self.weights.struct), _lib.ptr(tv), len(key),
                                               _lib.ptr(mods), _lib.stream_ptr(self.device))
            _lib.check(rc, "codlad_step_mods")
            if len(self._mods_cache) > 64:
                self._mods_cache.clear()
            self._mods_cache[key] = mods
        return self._mods_cache[key]

    # -- per call ------------------------------------------------------------------------------
    @property
    def self_condition(self):
        return self.weights.self_condition

    def check_status(self, job):
        """Wait for the job's stream and raise if a forward produced inf / NaN (in the split-fp16 modes:
        also if an operand left the fp16 range, include/codlad_hip.h CODLAD_STATUS_NONFINITE)."""
        _lib.check(self.lib.codlad_status_check(_lib.ptr(job.status), _lib.stream_ptr(self.device)),
                   "codlad_status_check")

    def forward(self, job, x, t_value, x_self_cond=None, check=True):
        """One denoiser call: x [n_nodes,3] -> [n_nodes,6] (eps | variance logits).  x_self_cond
        [n_nodes,3]: previous pred_xstart, for a self-conditioned model only (None = zeros).
        check: synchronise and raise on a non-finite output."""
        _require_cuda(x, "x")
        x = x.contiguous().float()
        assert x.shape == (job.n_nodes, 3)
        if x_self_cond is not None:
            if not self.self_condition:
                raise ValueError("x_self_cond given to a model built without self_condition")
            _require_cuda(x_self_cond, "x_self_cond")
            x_self_cond = x_self_cond.contiguous().float()
            assert x_self_cond.shape == x.shape
        mods = self.step_mods([t_value])
        out = torch.empty(job.n_nodes, self.weights.out_dim, dtype=torch.float32, device=self.device)
        st = job.structures
        self._fresh_features(st)
        rc = self.lib.codlad_denoiser_forward(C.byref(self.weights.struct), _lib.ptr(job.node_info),
                                              job.n_nodes, _lib.ptr(st.E_idx), _lib.ptr(st.h_E0),
                                              _lib.ptr(st.E1), st.n_snodes, _lib.ptr(x), _lib.ptr(x_self_cond),
                                              _lib.ptr(mods), _lib.ptr(out), C.byref(job.ws),
                                              _lib.stream_ptr(self.device))
        _lib.check(rc, "codlad_denoiser_forward")
        if check:
            self.check_status(job)
        return out

    # A job of this many nodes or more runs as two half-jobs on two HIP streams (round 4; measured on BASELINE
    # configuration 2, 35 400 nodes: +2.5 %): the node kernel of such a job occupies ~140 of the 256 CUs and every kernel
    # has a tail, which the other half's edge kernels fill.  Every unit's result is independent of what shares its job
    # (tests hold that to the bit), so the split changes nothing but the schedule.  Three and more parts lose.
    # Where it starts to pay (tools/mall_probe.py, structures of 87 residues, one box): 34 800 nodes +2.0 %, 17 400 +2.1 %,
    # 8 700 +9.6 % (each half then takes the small-job node kernel), 5 220 -1.9 %, 3 480 +5.5 %, 1 740 -8 %.
    SPLIT_MIN_NODES = int(os.environ.get("CODLAD_SAMPLE_SPLIT_MIN_NODES", 8192))

    def sample(self, job, x_T, noise, tables, check=True, coef=None, streams=None):
        """Full ancestral loop.  x_T [n_nodes,3]; noise [T,n_nodes,3] in loop order (first entry
        is used at step T-1); tables = diffusion_and_flow.schedule.Tables.  Returns x_0.
        check: after the loop, synchronise and raise if any step's output was not finite.
        coef: the [T, 8] step table when it is not the default sampler's (SpacedDiffusion.coefficients).
        streams: 1 = the whole job on the current stream; 2 = two half-jobs on two streams; None = 2 from
        SPLIT_MIN_NODES nodes up (and at least two samples)."""
        _require_cuda(x_T, "x_T")
        _require_cuda(noise, "noise")
        T = tables.num_timesteps
        assert noise.shape == (T, job.n_nodes, 3) and x_T.shape == (job.n_nodes, 3)
        if streams is None:
            streams = 2 if job.n_nodes >= self.SPLIT_MIN_NODES and len(job.sample_struct) >= 2 else 1
        if streams > 1:
            parts = job.parts(streams)
            outs = self.sample_many([p for p, _i in parts], [x_T[i] for _p, i in parts], [noise[:, i] for _p, i in parts],
                                    tables, check=check, coef=coef)
            x0 = torch.empty(job.n_nodes, 3, dtype=torch.float32, device=self.device)
            for (_p, i), o in zip(parts, outs):
                x0[i] = o
            return x0
        coef = tables.step_coefficients() if coef is None else coef
        fixed_var = bool(int(coef[0, 7]) & 2)
        if self.weights.out_dim != (3 if fixed_var else 6):
            raise ValueError("the DDPM loop needs a model with 6 outputs (mean | variance logits), or 3 with a fixed-variance "
                             "sampler (create_diffusion(learn_sigma=False)); a flow-matching model is sampled with "
                             "codlad_amd.diffusion_and_flow.ode.odeint")
        x = x_T.detach().clone().contiguous().float()
        noise = noise.contiguous().float()
        mods = self.step_mods(tables.timestep_map)
        coef = torch.from_numpy(coef).to(self.device)
        st = job.structures
        self._fresh_features(st)
        x_start = torch.empty_like(x) if self.self_condition else None   # pred_xstart, step to step
        rc = self.lib.codlad_sample_loop(C.byref(self.weights.struct), _lib.ptr(job.node_info),
                                         job.n_nodes, _lib.ptr(st.E_idx), _lib.ptr(st.h_E0),
                                         _lib.ptr(st.E1), st.n_snodes, _lib.ptr(x), _lib.ptr(x_start),
                                         _lib.ptr(noise), _lib.ptr(mods), _lib.ptr(coef), T, C.byref(job.ws),
                                         _lib.stream_ptr(self.device))
        _lib.check(rc, "codlad_sample_loop")
        if check:
            self.check_status(job)
        return x

    def sample_many(self, jobs, x_Ts, noises, tables, check=True, coef=None):
        """Several independent jobs at once, each on its own HIP stream: the node kernel of a 35 000-node job occupies 139 of
        the 256 CUs and every kernel has a tail - with a second job in flight another job's edge kernels run there (two
        half-jobs of BASELINE configuration 2: 1.03 x; more than two parts lose).  Every job carries its
        own workspace and the library keeps no state between jobs, so the results are those of `sample` job by job."""
        if not hasattr(self, "_streams"):
            self._streams = []
        while len(self._streams) < len(jobs):
            self._streams.append(torch.cuda.Stream(device=self.device))
        cur = torch.cuda.current_stream(self.device)
        for job in jobs:                                  # features and the step tables once, on the caller's stream
            self._fresh_features(job.structures)
        self.step_mods(tables.timestep_map)
        outs = []
        for job, x_T, noise, st in zip(jobs, x_Ts, noises, self._streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(self.sample(job, x_T, noise, tables, check=False, coef=coef, streams=1))
        for st in self._streams[:len(jobs)]:
            cur.wait_stream(st)
        if check:
            for job in jobs:
                self.check_status(job)
        return outs

    def ddpm_update(self, x, model_out, noise, tables, i, return_x_start=False):
        _require_cuda(x, "x")
        n = x.numel() // 3
        x = x.contiguous().float()
        out = torch.empty_like(x)
        x_start = torch.empty_like(x) if return_x_start else None
        coef = np.ascontiguousarray(tables.step_coefficients()[i])
        rc = self.lib.codlad_ddpm_update(_lib.ptr(x), _lib.ptr(model_out.contiguous().float()),
                                         _lib.ptr(noise.contiguous().float()),
                                         coef.ctypes.data_as(C.c_void_p), n, _lib.ptr(out), _lib.ptr(x_start),
                                         _lib.stream_ptr(self.device))
        _lib.check(rc, "codlad_ddpm_update")
        return (out, x_start) if return_x_start else out


class Decoder:
    """De-normalise + VQ lookup + IC decoder + ic_to_xyz."""

    def __init__(self, state_dict, device, mean3=None, std3=None, angle=False, n_codes=4096):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("codlad_amd runs on an MI355X only; no CPU path exists")
        self.lib = _lib.lib()
        # state_dict None: a rank that loads no checkpoint (`angle` / `n_codes` give the layout; a broadcast whose
        # header says otherwise re-derives it)
        if state_dict is None:
            self.weights = DecoderWeights.empty(self.device, angle, n_codes)
        else:
            self.weights = DecoderWeights(state_dict, self.device, mean3, std3)
        f = dict(dtype=torch.float32, device=self.device)
        self._unit = (torch.zeros(3, **f), torch.ones(3, **f))

    # the de-normalisation statistics live in the weight blob (they travel with a broadcast)
    @property
    def mean(self):
        return self.weights.mean

    @property
    def std(self):
        return self.weights.std

    def vq(self, x, normalised=True):
        """x [..., 3] -> (idx int64 [n], z_q [..., 3], latent [..., 3]); de-normalises first when
        `normalised` (reference test.py:548), else looks x up as is."""
        _require_cuda(x, "latent")
        xs = x.contiguous().float()
        n = xs.numel() // 3
        idx = torch.empty(n, dtype=torch.int64, device=self.device)
        zq = torch.empty_like(xs)
        lat = torch.empty_like(xs)
        mean, std = (self.mean, self.std) if normalised else self._unit
        cb = self.weights.codebook
        rc = self.lib.codlad_vq_lookup(_lib.ptr(xs), n, _lib.ptr(mean), _lib.ptr(std), _lib.ptr(cb),
                                       cb.shape[0], _lib.ptr(idx), _lib.ptr(zq), _lib.ptr(lat),
                                       _lib.stream_ptr(self.device))
        _lib.check(rc, "codlad_vq_lookup")
        return idx, zq, lat

    @staticmethod
    def csr_from_pairs(pairs, n_nodes):
        """Undirected CG pairs [E,2] -> CSR over the receiving node of the directed graph, in the
        order the reference's scatter_add visits them (models/gcn_nn.py:54-64, vae_model.py:485)."""
        gtr_ij = bool((pairs[:, 0] > pairs[:, 1]).any())
        gtr_ji = bool((pairs[:, 1] > pairs[:, 0]).any())
        directed = pairs if (gtr_ij and gtr_ji) else torch.cat([pairs, pairs.flip(1)], dim=0)
        recv = directed[:, 0]
        order = torch.sort(recv, stable=True).indices
        src = directed[order, 1].to(torch.int32).contiguous()
        counts = torch.bincount(recv, minlength=n_nodes)
        ptr = torch.zeros(n_nodes + 1, dtype=torch.int32, device=pairs.device)
        ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
        return ptr, src

    def build_csr(self, cg_xyz, sample_lens, cutoff=21.0, sample_range=None, max_edges=None):
        """Directed CG graph of every sample as CSR, on the device (what the reference's host
        preprocessing + make_directed + scatter order amount to).  cg_xyz [M,3] flat over samples,
        sample_lens: residues per sample."""
        xyz = cg_xyz.to(self.device).contiguous().float()
        M = xyz.shape[0]
        rng = self.sample_ranges(sample_lens) if sample_range is None else sample_range
        assert rng.shape == (M, 2)
        deg = torch.empty(M, dtype=torch.int32, device=self.device)
        st = _lib.stream_ptr(self.device)
        _lib.check(self.lib.codlad_cg_graph(_lib.ptr(xyz), _lib.ptr(rng), M, C.c_float(cutoff), _lib.ptr(deg),
                                            None, None, st), "codlad_cg_graph(count)")
        ptr = torch.zeros(M + 1, dtype=torch.int32, device=self.device)
        ptr[1:] = torch.cumsum(deg, 0).to(torch.int32)
        # max_edges (an upper bound on the directed edge count, e.g. sum L*(L-1)) avoids the host
        # round trip that sizes csr_src exactly; the tail past ptr[-1] is then simply unused
        n_src = max_edges if max_edges is not None else int(ptr[-1])
        src = torch.empty(max(n_src, 1), dtype=torch.int32, device=self.device)
        _lib.check(self.lib.codlad_cg_graph(_lib.ptr(xyz), _lib.ptr(rng), M, C.c_float(cutoff), None,
                                            _lib.ptr(ptr), _lib.ptr(src), st), "codlad_cg_graph(fill)")
        return ptr, src[:n_src]

    def sample_ranges(self, sample_lens):
        """[M,2] int32 device table {first node, L} of the sample each flat node belongs to."""
        M = int(sum(sample_lens))
        rng = np.empty((M, 2), dtype=np.int32)
        o = 0
        for L in sample_lens:
            rng[o:o + L, 0] = o
            rng[o:o + L, 1] = L
            o += L
        return torch.from_numpy(rng).to(self.device)

    def ic_decode(self, z_q, cg_z, cg_xyz, pairs=None, csr=None):
        """z_q [M,3], cg_z [M], cg_xyz [M,3] and either the undirected CG pairs [E,2] (flat node
        indices, as in batch['CG_nbr_list']) or a prebuilt csr = (ptr, src) -> ic [M,13,3]."""
        _require_cuda(z_q, "z_q")
        M = z_q.shape[0]
        cg_z = cg_z.to(self.device, torch.int32).contiguous()
        lo, hi = int(cg_z.min()), int(cg_z.max())
        if lo < 0 or hi >= 25:      # res_embed / backbone_dist / sidechain_* tables have 25 rows (vae_model.py:330-345)
            raise ValueError(f"cg_z (residue types {lo}..{hi}) outside the decoder's 25-row embedding tables")
        ptr, src = csr if csr is not None else self.csr_from_pairs(pairs.to(self.device), M)
        assert ptr.numel() == M + 1 and ptr.dtype == torch.int32 and src.dtype == torch.int32
        scratch = torch.empty(M, 200, dtype=torch.float32, device=self.device)
        ic = torch.empty(M, 13, 3, dtype=torch.float32, device=self.device)
        rc = self.lib.codlad_ic_decode(C.byref(self.weights.struct), _lib.ptr(z_q.contiguous().float()),
                                       _lib.ptr(cg_z),
                                       _lib.ptr(cg_xyz.to(self.device).contiguous().float()),
                                       _lib.ptr(ptr), _lib.ptr(src), M, _lib.ptr(scratch), _lib.ptr(ic),
                                       _lib.stream_ptr(self.device))
        _lib.check(rc, "codlad_ic_decode")
        return ic

    def ic_to_xyz(self, ca_full, ic, info):
        """ca_full [B,L+2,3], ic [B,L,13,3], info = (permute, atom_idx, atom_orders) -> [B,n_atoms,3]."""
        _require_cuda(ic, "ic")
        B, L = ic.shape[0], ic.shape[1]
        orders, slot_to_out, n_atoms = info_tables(info, L, self.device)
        out = torch.empty(B, n_atoms, 3, dtype=torch.float32, device=self.device)
        rc = self.lib.codlad_ic_to_xyz(_lib.ptr(ca_full.to(self.device).contiguous().float()),
                                       _lib.ptr(ic.contiguous().float()), _lib.ptr(orders),
                                       _lib.ptr(slot_to_out), B, L, n_atoms, _lib.ptr(out),
                                       _lib.stream_ptr(self.device))
        _lib.check(rc, "codlad_ic_to_xyz")
        return out


    def ic_to_xyz_groups(self, groups, reuse=False):
        """groups: [(ca_full [B,L+2,3], ic [B,L,13,3], info)] of several proteins -> [xyz [B,n_atoms,3]] in ONE launch
        (codlad_ic_to_xyz_groups).  reuse: keep the descriptor table AND the output tensors of the previous call when the
        inputs sit at the same addresses (a job that is run again and again: the results of the earlier call are
        overwritten); default: a fresh table and fresh outputs per call (one small host-to-device copy)."""
        key = tuple((ca.data_ptr(), ic.data_ptr(), id(info[0]), tuple(ic.shape)) for ca, ic, info in groups)
        cache = getattr(self, "_xyz_groups", None) if reuse else None
        if cache is None or cache[0] != key:
            desc = (_lib.XyzGroup * len(groups))()
            outs, keep, row = [], [], 0
            for g, (ca, ic, info) in enumerate(groups):
                _require_cuda(ic, "ic")
                B, L = ic.shape[0], ic.shape[1]
                orders, s2o, n_atoms = info_tables(info, L, self.device)
                ca_d, ic_d = ca.to(self.device).contiguous().float(), ic.contiguous().float()
                assert ic_d.data_ptr() == ic.data_ptr() and ca_d.shape == (B, L + 2, 3)
                out = torch.empty(B, n_atoms, 3, dtype=torch.float32, device=self.device)
                d = desc[g]
                d.ca_full, d.ic, d.orders, d.slot_to_out, d.xyz_out = (ca_d.data_ptr(), ic_d.data_ptr(), orders.data_ptr(),
                                                                       s2o.data_ptr(), out.data_ptr())
                d.B, d.L, d.n_atoms, d.first_row = B, L, n_atoms, row
                row += B * L
                outs.append(out)
                keep += [ca_d, ic_d, orders, s2o]
            table = torch.frombuffer(bytearray(bytes(desc)), dtype=torch.uint8).to(self.device)
            cache = (key, table, outs, keep, row)
            if reuse:
                self._xyz_groups = cache
        _key, table, outs, _keep, rows = cache
        rc = self.lib.codlad_ic_to_xyz_groups(_lib.ptr(table), len(groups), rows, _lib.stream_ptr(self.device))
        _lib.check(rc, "codlad_ic_to_xyz_groups")
        return outs


_INFO_CACHE = {}


def info_tables(info, L, device):
    """(permute, atom_idx, atom_orders) of the reference (utils/protein_module.py:434-494) ->
    int32 device tables for codlad_ic_to_xyz.  Output atom p takes slot atom_idx[permute[p]]
    (utils/utils_ic.py:267)."""
    permute, atom_idx, orders = info
    key = (id(permute), id(atom_idx), id(orders), L, str(device))
    if key not in _INFO_CACHE:
        assert orders.shape == (10, L, 3), "atom_orders does not match the batch's residue count"
        n_atoms = int(permute.numel())
        slot = atom_idx.cpu()[permute.cpu()]
        assert int(slot.max()) < 14 * L and torch.unique(slot).numel() == n_atoms
        s2o = torch.full((14 * L,), -1, dtype=torch.int32)
        s2o[slot] = torch.arange(n_atoms, dtype=torch.int32)
        o32 = orders.to(torch.int32).contiguous()
        assert int(o32.min()) >= 0
        for i in range(10):  # slot i+4 may only reference earlier slots
            assert int(o32[i].max()) < 4 + i, "atom_orders references an atom that is not placed yet"
        if len(_INFO_CACHE) > 256:
            _INFO_CACHE.clear()
        _INFO_CACHE[key] = (o32.to(device), s2o.to(device), n_atoms, info)
    o, s, n, _keepalive = _INFO_CACHE[key]
    return o, s, n
