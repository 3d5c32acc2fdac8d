#!/usr/bin/env python3
"""Benchmark of the CODLAD sampling hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg3|cfg4|cfg4share|cfg5]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Default workload (BASELINE.json configs[1], "cfg2"): synthetic PED-shaped test set - 4 proteins with
L = 46/87/92/129 residues, 10 frames each, num_ensemble = 10 -> 400 structures per GPU, 100-step
respaced DDPM with the mpnn_diffusion denoiser, then de-normalise + VQ (4096 codes) + IC_Decoder
(N6) + ic_to_xyz.  One "step" of this benchmark = that whole job: CA traces and noise resident in HBM
-> all-atom coordinates resident in HBM, INCLUDING the per-structure work (k-NN graph + edge features,
hoisted layer-0 edge terms, the adaLN vectors of all timesteps, the decoder's CG graph).  For cfg2 / cfg5
the ranks are independent replicas of the job on different seeds ("scaling": "weak"); cfg3 / cfg4 are ONE
job whose units (protein, frame, ensemble member) are dealt to the ranks longest-first ("strong").  Weights
are broadcast from rank 0 over RCCL before timing and every rank's coordinates are all-gathered inside the
timed region.

Prints ONE JSON line (rank 0).  `roofline` times the dominant kernel (the per-edge message MLP) with HIP
events on its own stream and prices it against the matrix pipe it runs on; `cpu_baseline` times the CPU
oracle (oracle/, a port of the reference's PyTorch-CPU path) on a bounded sample of the same workload;
`f32_mfma` is the same job with every contraction on v_mfma_f32_32x32x2_f32 (IEEE fp32 products).
"""
import argparse
import ctypes as C
import glob
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import codlad_amd  # noqa: E402,F401  (before the first HIP call: the package asks the runtime for device-side kernel arguments)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T_STEPS = 100
WEIGHT_SEED, VAE_SEED = 1234, 4321
# /opt/skills/guides/MI355X_MICROARCH.md: dense F16/BF16 MFMA ~2.5 PFLOP/s; fp32 matrix 157.3 TFLOP/s; HBM3E 8 TB/s
F16_MFMA_PEAK_TFLOPS = 2500.0
FP32_MFMA_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0
DTYPE = {"f16x3": "f32 as f16x3 split (each fp32 operand = fp16 hi + fp16 lo, 22-bit; 3 f16 MFMAs per product, fp32 "
                  "accumulate); everything outside the contractions fp32",
         "f16x4": "f32 as f16x4 split (each fp32 operand = fp16 hi + fp16 lo, 22-bit; 4 f16 MFMAs per product, fp32 "
                  "accumulate); everything outside the contractions fp32",
         "f32": "f32 (v_mfma_f32_32x32x2_f32)"}


def algorithmic_flop_per_structure(L):
    """per denoiser step per real sample 2*(786432*L*K + 787584*L + 819200),
    K = min(64, L); features once per structure 2*37760*L*K (computed inside the timed region)."""
    K = min(64, L)
    return T_STEPS * 2 * (786432 * L * K + 787584 * L + 819200) + 2 * 37760 * L * K


def decode_bytes_per_structure(L, n_atoms, n_pairs):
    """SURVEY.md §8(d), cfg 5: read latent 12 L + CA 16 (L+2) + pairs 16 E_undirected + tables 8 n_atoms +
    240 L; write ic 156 L + xyz 12 n_atoms."""
    return 12 * L + 16 * (L + 2) + 16 * n_pairs + 8 * n_atoms + 240 * L + 156 * L + 12 * n_atoms


class Workload:
    """One rank's part of a BASELINE.json configuration, inputs resident on its GPU."""

    def __init__(self, device, cfg_name, rank=0, world=1, precision="f16x3"):
        from codlad_amd import parallel, synth
        from codlad_amd.engine import Decoder, Denoiser
        from codlad_amd.diffusion_and_flow.schedule import Tables, named_betas, space_timesteps
        cfg = synth.baseline_config(cfg_name)
        self.cfg, self.cfg_name, self.device, self.precision = cfg, cfg_name, device, precision
        self.decode_only = cfg["decode_only"]
        strong = cfg["scaling"] == "strong"
        if rank == 0 or world == 1:
            self.den = Denoiser(synth.denoiser_state_dict(WEIGHT_SEED), device, precision=precision)
            mean, std = synth.norm_stats(cfg["dataname"], cfg["vae_type"])
            self.dec = Decoder(synth.vqvae_state_dict(cfg["vae_type"], cfg["dataname"], VAE_SEED), device, mean, std)
        else:
            # ranks > 0 hold NO weights (and, for the decoder, not even the right layout: the N6 one whatever the
            # configuration) until rank 0's arrive by broadcast - header first, then the blob (parallel.py)
            self.den = Denoiser(None, device, precision=precision)
            self.dec = Decoder(None, device)
        self.tables = Tables(named_betas("linear", 1000), space_timesteps(1000, str(T_STEPS)))
        lengths, F, E = cfg["lengths"], cfg["n_frames"], cfg["n_ensemble"]
        # weak scaling: every rank its own proteins (different seeds); strong: one job, same proteins everywhere
        seed0 = 1000 if strong else 100 * rank + 1000
        self.proteins = [synth.make_protein(L, seed0 + i, n_frames=F, phospho=cfg["vae_type"] != "N6")
                         for i, L in enumerate(lengths)]
        units = [(p, f, m) for p in range(len(lengths)) for f in range(F) for m in range(E)]
        self.n_units_job = len(units)
        self.job_flop = sum(algorithmic_flop_per_structure(lengths[p]) for p, _f, _m in units)
        if strong:
            costs = [parallel.unit_cost(lengths[p]) for p, _f, _m in units]
            parts = cfg.get("share_of") or world
            shard = parallel.shard_units(costs, parts)[rank if not cfg.get("share_of") else 0]
            units = [units[u] for u in shard]
            if cfg.get("share_of"):
                self.n_units_job = len(units)
                self.job_flop = sum(algorithmic_flop_per_structure(lengths[p]) for p, _f, _m in units)
        self.units = units                                   # sorted by (protein, frame, member)
        s_key = sorted({u[:2] for u in units})
        s_of = {k: i for i, k in enumerate(s_key)}
        xyz_list = [torch.from_numpy(self.proteins[p]["xyz_full"])[f, 1:-1] for p, f in s_key]
        z_list = [torch.from_numpy(self.proteins[p]["z_full"])[1:-1] for p, _f in s_key]
        self.structures = self.den.new_structures(xyz_list, z_list)
        self.job = self.den.make_job(self.structures, [s_of[u[:2]] for u in units])
        self.n_structures = len(units)
        # decoder-side tables (host preprocessing in the reference: batch dict, info)
        ni = self.job.node_info
        self.cg_z = ni[:, 3].contiguous()
        self.cg_xyz = self.structures.xyz[ni[:, 0].long()].contiguous()
        self.sample_range = self.dec.sample_ranges(self.job.sample_lens)
        self.max_dir_edges = int(sum(L * (L - 1) for L in self.job.sample_lens))
        self.n_edges = int(ni[:, 2].sum())
        self.groups = []                                     # (first sample, count, protein, ca_full [count, L+2, 3])
        k = 0
        while k < len(units):
            p = units[k][0]
            k2 = k
            while k2 < len(units) and units[k2][0] == p:
                k2 += 1
            frames = torch.from_numpy(self.proteins[p]["xyz_full"]).to(device)
            idx = torch.tensor([u[1] for u in units[k:k2]], device=device)
            self.groups.append((k, k2 - k, self.proteins[p], frames[idx].contiguous()))
            k = k2
        g = torch.Generator(device=device)
        g.manual_seed(42 + rank)
        n = self.job.n_nodes
        if self.decode_only:
            self.latent = torch.randn(n, 3, generator=g, device=device)      # normalised, stands in for the encoder
        else:
            self.x_T = torch.randn(n, 3, generator=g, device=device)
            self.noise = torch.randn(T_STEPS, n, 3, generator=g, device=device)
        if not strong:
            self.job_flop *= world

    def prepass(self):
        """Per-structure and per-job work the reference redoes in every step (CA features, adaLN vectors) or on
        the host (CG neighbour list): kernels only, tables and buffers were made in __init__."""
        if not self.decode_only:
            self.den.compute_features(self.structures)
            self.den.step_mods(self.tables.timestep_map, refresh=True)
        self.csr = self.dec.build_csr(self.cg_xyz, self.job.sample_lens, sample_range=self.sample_range,
                                      max_edges=self.max_dir_edges)

    def run(self, streams=None):
        """CA traces + noise (HBM) -> all-atom coordinates (HBM) for this rank's structures.  streams: None = as a user's call
        runs (codlad_amd.engine.Denoiser.sample: a large job goes as two half-jobs on two HIP streams), 1 = everything on the
        current stream (the probe pass, where a launch's duration has to be the kernel's own)."""
        self.prepass()
        x0 = self.latent if self.decode_only else self.den.sample(self.job, self.x_T, self.noise, self.tables, streams=streams)
        return self.decode(x0)

    def decode(self, x0, normalised=True):
        idx, zq, _lat = self.dec.vq(x0, normalised=normalised)
        ic = self.dec.ic_decode(zq, self.cg_z, self.cg_xyz, csr=self.csr)
        groups = []
        for start, count, prot, ca in self.groups:
            L = prot["n_cg"]
            a = int(self.job.sample_off[start])
            b = int(self.job.sample_off[start + count])
            groups.append((ca, ic[a:b].view(count, L, 13, 3), prot["info"]))
        # every protein's placements in one launch (one launch per protein is latency-bound: 20 us each)
        return self.dec.ic_to_xyz_groups(groups, reuse=True), idx

    def timed(self, fn, n=1):
        """Average wall time of fn() in seconds by HIP events on the stream the kernels are enqueued on."""
        stream = torch.cuda.current_stream(self.device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(n):
            fn()
        e1.record(stream)
        e1.synchronize()
        return e0.elapsed_time(e1) / n * 1e-3

    def time_dominant_kernel(self, n_launch=20):
        """Average duration of one message-kernel launch (encoder layer 1: both GEMM layers over
        every edge of the job, edge state read from HBM - what 5 of the 6 message launches of a step
        look like) and of one edge-update launch, HIP events on the stream the kernels run on."""
        from codlad_amd import _lib
        lib = _lib.lib()
        stream = torch.cuda.current_stream(self.device)
        mods = self.den.step_mods(self.tables.timestep_map)
        st = self.structures

        def launch(which):
            rc = lib.codlad_bench_edge_launch(C.byref(self.den.weights.struct), _lib.ptr(self.job.node_info),
                                              self.job.n_nodes, _lib.ptr(st.E_idx), _lib.ptr(st.h_E0),
                                              _lib.ptr(mods), C.byref(self.job.ws), which, 1,
                                              C.c_void_p(stream.cuda_stream))
            _lib.check(rc, "codlad_bench_edge_launch")

        res = {}
        for which, name in ((0, "message"), (1, "edge_update")):
            for _ in range(3):
                launch(which)
            res[name] = self.timed(lambda: launch(which), n_launch)
        return res


def recon_from_atoms(wl):
    """cfg 5 the way `test.py --experiment recon` runs it (reference test.py:501): the VQ-VAE's e3nn encoder on every
    frame's atoms, map_in, and the frame's latent decoded once per ensemble member.  Synthetic atoms
    (codlad_amd.synth.make_atoms) and encoder weights; -> seconds per pass over the job's structures."""
    from codlad_amd import synth
    from codlad_amd.utils.model_module import build_vae, load_decoder_state
    cfg = wl.cfg
    vae = build_vae(cfg["vae_type"], with_encoder=True)
    sd = synth.vqvae_state_dict(cfg["vae_type"], cfg["dataname"], VAE_SEED)
    sd.update({"encoder." + k: v for k, v in synth.encoder_state_dict(778).items()})
    sd.update({k: v for k, v in vae.state_dict().items() if k.endswith(".offset")})
    load_decoder_state(vae, sd)
    vae = vae.to(wl.device).eval()
    batches, first_row, row = [], {}, 0
    for p, prot in enumerate(wl.proteins):
        b = synth.make_batch(prot)
        b.update(synth.make_atoms(prot, seed=p))
        batches.append({k: (v.to(wl.device) if torch.is_tensor(v) else v) for k, v in b.items()})
        for f in range(cfg["n_frames"]):
            first_row[(p, f)] = row
            row += prot["n_cg"]
    src = torch.cat([first_row[(p, f)] + torch.arange(wl.proteins[p]["n_cg"]) for p, f, _m in wl.units]).to(wl.device)
    n_atoms = sum(int(b["nxyz"].shape[0]) for b in batches)

    def run():
        wl.prepass()
        lat = torch.cat([vae.get_latent_wovq(b)[0].reshape(-1, 3) for b in batches], 0)   # [frames x L, 3], one per frame
        return wl.decode(lat[src], normalised=False)

    def encode():
        return [vae.get_latent_wovq(b)[0] for b in batches]

    run()
    return {"seconds": wl.timed(run, 5), "encoder_seconds": wl.timed(encode, 5), "frames": row and len(first_row),
            "atoms": n_atoms, "directed_atom_edges": sum(2 * int(b["nbr_list"].shape[0]) for b in batches)}


PROBE_KINDS = ((0, "message"), (1, "edge_update"), (2, "message_hoisted"), (3, "edge_update_hoisted"))


def probe_edge_kernels(wl):
    """One more pass of the job with the library's event probe on (codlad_probe_edge_launches): every edge-kernel launch
    bracketed by HIP events on the stream it runs on -> per kind the average duration IN the job, between the node
    kernels and at the job's clock, which is what `rocprofv3 --kernel-trace --stats` averages too."""
    from codlad_amd import _lib
    lib = _lib.lib()
    lib.codlad_probe_edge_launches(1)
    try:
        wl.run(streams=1)           # single stream: with two half-jobs in flight a launch shares the chip with the other half's
        torch.cuda.synchronize(wl.device)
    finally:
        lib.codlad_probe_edge_launches(0)
    res = {}
    for kind, name in PROBE_KINDS:
        total = C.c_double(0.0)
        n = lib.codlad_probe_read(kind, C.byref(total))
        if n < 0:
            _lib.check(n, "codlad_probe_read")
        if n:
            res[name] = {"ms": total.value / n, "launches": n}
    return res


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg_name):
    """The CPU oracle (port of the reference's PyTorch-CPU path) on BASELINE.md section 4's sample - the L = 87 protein,
    10 frames, num_ensemble 1 - with a bounded number of the 100 DDPM steps (10, every step costs the same: extrapolated
    x10), run as the reference runs them (batch duplicated, test.py:505; CA features recomputed every step), plus the
    decoder tail in full.  cfg5: the decoder tail alone."""
    from codlad_amd import synth
    from oracle import denoiser as oden, sampler as osam, vae_decode as odec
    torch.set_grad_enabled(False)
    present = os.cpu_count() or 1
    avail = present
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    cores = min(avail, 16)   # a one-GPU box owns 16 host cores; more threads only oversubscribe
    torch.set_num_threads(cores)
    sd = synth.denoiser_state_dict(WEIGHT_SEED)
    vsd = synth.vqvae_state_dict("N6", "PED", VAE_SEED)
    mean, std = synth.norm_stats("PED", "N6")
    L, B, Tsub = 87, 10, 10
    prot = synth.make_protein(L, 1001, n_frames=B)
    batch = synth.make_batch(prot)
    cg_z, cg_xyz, mask = oden.batch_to_dense(batch)
    dup = lambda t: torch.cat([t, t])  # noqa: E731
    z = synth.gaussian((2 * B, L, 3), 1)
    eps = synth.gaussian((Tsub, 2 * B, L, 3), 2)
    common = {"unit": "structures/s", "cores": cores, "cores_present": present, "cores_available": avail,
              "kind": "port", "cpu_model": cpu_model()}

    def decode(x):
        t0 = time.perf_counter()
        idx, ic = odec.latent_decode(vsd, odec.denormalise(x, mean, std), batch)
        odec.ic_to_xyz(batch["OG_CG_nxyz"].reshape(-1, L + 2, 4), ic.reshape(-1, L, 13, 3), prot["info"])
        return time.perf_counter() - t0

    if cfg_name == "cfg5":
        decode(z[:B])
        reps = 20
        t_dec = sum(decode(z[:B]) for _ in range(reps)) / reps
        return dict(common, value=B / t_dec,
                    sample=f"oracle (PyTorch-CPU fp32, {cores} of {present} hardware threads): decoder tail only (de-normalise "
                           f"+ VQ + IC_Decoder + ic_to_xyz) on L=87, {B} frames, mean of {reps} runs: {t_dec * 1e3:.1f} ms")
    t0 = time.perf_counter()
    x = osam.p_sample_loop(sd, Tsub, z, eps, dup(cg_xyz), dup(cg_z), dup(mask))
    t_loop = time.perf_counter() - t0
    t_dec = decode(x[:B])
    per_struct = (t_loop * (T_STEPS / Tsub) + t_dec) / B
    # the same work without the reference's redundancy (no duplicated batch, features once): what a
    # tuned CPU run of this algorithm would do, reported beside the headline baseline
    t0 = time.perf_counter()
    osam.p_sample_loop(sd, Tsub, z[:B], eps[:, :B], cg_xyz, cg_z, mask, hoist_features=True)
    t_dedup = time.perf_counter() - t0
    dedup = 1.0 / ((t_dedup * (T_STEPS / Tsub) + t_dec) / B)
    return dict(common, value=1.0 / per_struct, value_deduplicated=dedup,
                sample=f"oracle (PyTorch-CPU fp32, {cores} of {present} hardware threads): BASELINE.md 4's sample - L=87, {B} "
                       f"frames, ens 1 - {Tsub} of {T_STEPS} DDPM steps as the reference runs them (2x duplicated batch, "
                       f"features recomputed per step) + decode in full, loop extrapolated x{T_STEPS // Tsub}; "
                       f"{t_loop:.2f}s loop + {t_dec:.3f}s decode")


def committed_traffic(cfg_name, precision):
    """HBM bytes per launch of the dominant kernel.  PMC counters need rocprofv3 around the process, so the
    figure comes from the newest committed profile of this same workload and kernel (profiles/README.md says
    how it was taken) and is tagged with its source; null for workloads that have no such profile."""
    if cfg_name != "cfg2" or precision != "f16x3":
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    finals = [f for f in files if "_mid_" not in f]
    if not finals:
        return None, None
    with open(finals[-1]) as f:
        d = json.load(f)
    return d["hbm_bytes_per_launch"], f"profiles/{os.path.basename(finals[-1])} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, {d['kernel']})"


def decode_roofline(wl):
    """Roofline object of the decoder tail (BASELINE configuration 5): HBM / latency-bound, the whole decode (CG graph, VQ,
    IC decoder, ic_to_xyz) as one unit."""
    t_dec = wl.timed(wl.run, 5)
    n_atoms = sum(c * int(p["info"][0].numel()) for _s, c, p, _ca in wl.groups)
    nbytes = sum(decode_bytes_per_structure(L, 0, 0) for L in wl.job.sample_lens) + 20 * n_atoms + \
        16 * int(wl.csr[0][-1]) // 2
    # SURVEY.md 8d prices this configuration against HBM; the tail is arithmetic- and latency-bound, so the same
    # time is also priced against the fp32 rate: algorithmic FLOP of the decoder (reference vae_model.py:467-503:
    # per directed CG edge and message block a 15 -> 40 filter, its envelope and a 40-wide multiply-add = 2 (15 40
    # + 40 + 40) FLOP, four blocks; per residue the 40..53-wide dense layers, ~30 k MAC) and of the VQ scan
    # (4096 codes x 8 FLOP per residue)
    n_dir = int(wl.csr[0][-1])
    alg_flop = 4 * 2 * (15 * 40 + 80) * n_dir + (2 * 30000 + 8 * 4096) * wl.job.n_nodes
    return {"bound": "hbm", "achieved": nbytes / t_dec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": nbytes / t_dec / 1e9 / HBM_PEAK_GBS, "traffic": None,
            "kernel": "cg_graph + vq + dec_init/edge/dense/heads + ic_to_xyz (the whole decode; its largest kernel, "
                      "dec_edge_kernel: per CG edge one sine / cosine, a 15 -> 40 filter on the f16 matrix pipe "
                      "(split fp16) and a gathered 40-wide multiply-add)",
            "launch_ms": t_dec * 1e3, "algorithmic_bytes_per_launch": nbytes,
            "fp32_compute": {"algorithmic_flop_per_launch": alg_flop, "achieved_tflops": alg_flop / t_dec / 1e12,
                             "peak_tflops": FP32_MFMA_PEAK_TFLOPS,
                             "frac": alg_flop / t_dec / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                             "note": "algorithmic decoder + VQ FLOP / the same launch time / fp32 vector peak: "
                                     "the tail is bound by latency and instruction issue, not by bytes"}}


def recon_section(wl):
    r = recon_from_atoms(wl)
    return {"value": wl.n_structures / r["seconds"], "unit": "structures/s", "ms_per_step": r["seconds"] * 1e3,
            "encoder_ms": r["encoder_seconds"] * 1e3, "frames": r["frames"], "atoms": r["atoms"],
            "directed_atom_edges": r["directed_atom_edges"],
            "note": "the same structures as test.py --experiment recon produces them: e3nn encoder on every frame's "
                    "(synthetic) atoms + map_in, the frame's latent decoded once per ensemble member; `value` above "
                    "is the decoder tail alone (BASELINE configuration 5 isolates the decoder / codebook kernels)"}


def cfg5_section(device, steps=200, warmup=10):
    """BASELINE configuration 5 (decoder only, 1 GPU) measured in the same process, for the default run's line: the same
    timing rule as `--config cfg5` (CUDA-synchronised wall time over `steps` passes after `warmup`)."""
    wl5 = Workload(device, "cfg5")
    for _ in range(warmup):
        wl5.run()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        wl5.run()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    return {"metric": "reconstructed all-atom structures/sec (cfg5: --experiment recon, decoder only)",
            "value": wl5.n_structures * steps / dt, "unit": "structures/s", "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "config": {"workload": wl5.cfg["what"], "structures_per_step": wl5.n_structures},
            "roofline": decode_roofline(wl5), "recon_from_atoms": recon_section(wl5)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 3 (cfg5, a 0.6 ms step: 200)")
    ap.add_argument("--warmup", type=int, default=None, help="default 1 (cfg5: 10)")
    ap.add_argument("--config", choices=["cfg2", "cfg3", "cfg4", "cfg4share", "cfg5"], default="cfg2",
                    help="BASELINE.json configuration (default cfg2, the one the metric is quoted on)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f32-leg", action="store_true", help="skip the extra fp32-MFMA timing of the same job")
    ap.add_argument("--streams", type=int, default=None,
                    help="1: the whole job on one HIP stream (kernel traces / counter passes, where a kernel's duration must be "
                         "its own); default: as the product runs it (a large job = two half-jobs on two streams)")
    ap.add_argument("--no-cfg5", action="store_true", help="cfg2 run: skip the appended decoder-only measurement (BASELINE configuration 5)")
    ap.add_argument("--precision", choices=["f16x3", "f16x4", "f32"], default="f16x3",
                    help="contraction mode: f16x3 (default) / f16x4 = fp32 operands split into fp16 hi+lo halves on "
                         "the f16 matrix pipe, 3 or 4 cross products per fp32 product, fp32 accumulation "
                         "(fp32-equivalent); f32 = v_mfma_f32_32x32x2_f32")
    args = ap.parse_args()
    # a step of the decoder-only configuration is 0.6 ms of GPU work behind ~0.7 ms of launch calls: three steps would time the
    # host's enqueue rate, a few hundred fill the queue
    if args.steps is None:
        args.steps = 200 if args.config == "cfg5" else 3
    if args.warmup is None:
        args.warmup = 10 if args.config == "cfg5" else 1

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.config == "cfg4share" and world != 1:
        raise SystemExit("cfg4share is one GPU's eighth of cfg4: run it with --gpus 1, or run --config cfg4 --gpus N")
    torch.set_grad_enabled(False)
    backend = os.environ.get("CODLAD_DIST_BACKEND", "nccl")   # "gloo": rehearsal of N > 1 on a one-GPU box
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)

    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    wl = Workload(device, args.config, rank, world, args.precision)
    if world > 1:
        # weights travel once, rank 0 -> all, as one buffer each (RCCL broadcast over xGMI)
        from codlad_amd.parallel import broadcast_weights
        broadcast_weights(wl.den.weights, wl.dec.weights)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        xyz, idx = wl.run(streams=args.streams)
        if world > 1:
            from codlad_amd.parallel import gather_coordinates
            gather_coordinates(xyz, world)
        return xyz

    def timed_steps(n_warm, n_steps):
        for _ in range(n_warm):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax)
        return dt

    dt = timed_steps(args.warmup, args.steps)
    strong = wl.cfg["scaling"] == "strong"
    structs_per_step = wl.n_units_job if strong else wl.n_structures * world

    extra = {}
    if rank == 0 and world == 1:
        extra["prepass_ms"] = wl.timed(wl.prepass, 3) * 1e3     # share of ms_per_step spent before the first DDPM step
        if wl.decode_only:
            extra["recon_from_atoms"] = recon_section(wl)
    if not wl.decode_only:
        kern = wl.time_dominant_kernel()          # back to back, alone on the chip
        insitu = probe_edge_kernels(wl)           # inside the job
        t_msg = insitu["message"]["ms"] / 1e3 if "message" in insitu else kern["message"]
        if world == 1 and args.precision != "f32" and not args.no_f32_leg:
            # the same job, same weights, contractions on the fp32 matrix instruction (IEEE fp32 products)
            wl.den.weights.set_precision("f32")
            F32_STEPS = 3
            dt32 = timed_steps(1, F32_STEPS) / F32_STEPS
            k32 = wl.time_dominant_kernel(5)
            wl.den.weights.set_precision(args.precision)
            extra["f32_mfma"] = {"value": structs_per_step / dt32, "unit": "structures/s", "ms_per_step": dt32 * 1e3,
                                 "launch_ms": k32["message"] * 1e3, "steps": F32_STEPS, "warmup": 1,
                                 "note": "same job and weights with every contraction on v_mfma_f32_32x32x2_f32; "
                                         "roofline of that mode: algorithmic FLOP / launch / 157.3 TFLOP/s = "
                                         f"{2.0 * (384 * 128 + 128 * 128) * wl.n_edges / k32['message'] / 1e12 / FP32_MFMA_PEAK_TFLOPS:.3f}"
                                         " (above 1 is possible: the W1 split removes half of the algorithmic MACs)"}
    if rank == 0:
        value = structs_per_step * args.steps / dt
        terms = {"f16x3": 3, "f16x4": 4, "f32": 0}[args.precision]
        if wl.decode_only:
            roofline = decode_roofline(wl)
        else:
            # dominant kernel: layers 1-2 of the encoder message MLP.  Algorithmic 2*(384*128 + 128*128) FLOP per
            # edge (reference protein_mpnn_utils.py:240-243; W3 runs in the node kernel); EXECUTED after the W1
            # split: two 128x128 contractions per edge, each product as `terms` f16 MFMA products.
            # The timed population (in_job_launch_ms["message"]) is the 5 non-hoisted message launches of a step: encoder
            # layers 1 and 2 (65 536 algorithmic MAC per edge each) and the three decoder layers (81 920:
            # protein_mpnn_utils.py:296-318, W1 is 512 -> 128 there), so a launch's algorithmic work is their mean.
            alg = 2.0 * (2 * 65536 + 3 * 81920) / 5 * wl.n_edges
            exe = 2.0 * 2 * 128 * 128 * wl.n_edges
            traffic, traffic_src = committed_traffic(args.config, args.precision)
            if terms:
                achieved, peak = terms * exe / t_msg / 1e12, F16_MFMA_PEAK_TFLOPS
                pipe = f"f16 matrix pipe (v_mfma_f32_32x32x16_f16), {terms} MFMA products per fp32 product"
            else:
                achieved, peak = exe / t_msg / 1e12, FP32_MFMA_PEAK_TFLOPS
                pipe = "fp32 matrix instruction (v_mfma_f32_32x32x2_f32)"
            roofline = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                        "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src,
                        # the same launch priced by SURVEY.md 8(d)'s ALGORITHMIC work (fp32 MACs of the reference's layers 1-2,
                        # enc / dec mix above) instead of the matrix-pipe work executed for it (3 f16 products per fp32 product,
                        # half the MACs after the W1 split): what fraction of the pipe's peak the USEFUL arithmetic amounts to
                        "frac_algorithmic": alg / t_msg / 1e12 / peak,
                        "pipe": pipe,
                        "kernel": ("msg_kernel_h " if terms else "edge_kernel<false> ") + "(encoder message MLP, layers 1-2)",
                        "launch_ms": t_msg * 1e3,
                        "launch_timing": "HIP events around every launch of this kernel inside one pass of the job "
                                         f"({insitu.get('message', {}).get('launches', 0)} launches, on the stream they "
                                         "run on): the same population rocprofv3 --kernel-trace --stats averages",
                        "launch_ms_back_to_back": kern["message"] * 1e3,
                        "in_job_launch_ms": {k: round(v["ms"], 4) for k, v in insitu.items()},
                        "flop_note": "achieved = matrix-pipe FLOP the launch EXECUTES / HIP-event time (what the pipe is "
                                     "busy with); the algebraic W1 split halves the algorithmic MACs of SURVEY.md 8d, so "
                                     "algorithmic figures are listed separately and never divided into this peak",
                        "executed_pipe_flop_per_launch": (terms or 1) * exe,
                        "algorithmic_flop_per_launch": alg,
                        "algorithmic_tflops": alg / t_msg / 1e12,
                        "algorithmic_over_fp32_mfma_peak": alg / t_msg / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                        "edge_update_launch_ms": insitu.get("edge_update", {}).get("ms", kern["edge_update"] * 1e3),
                        "edge_update_launch_ms_back_to_back": kern["edge_update"] * 1e3,
                        "whole_job_algorithmic_tflops": wl.job_flop * args.steps / dt / 1e12}
        result = {
            "metric": "sampled all-atom structures/sec (100-step DDPM, PED)" if args.config == "cfg2" else
                      f"sampled all-atom structures/sec ({args.config})",
            "value": value, "unit": "structures/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": wl.cfg["scaling"], "vs_baseline": None, "dtype": DTYPE[args.precision], "data": "synthetic",
            "precision": args.precision,
            "config": {"workload": wl.cfg["what"],
                       "structures_per_step": structs_per_step, "structures_rank0": wl.n_structures,
                       "nodes_rank0": wl.job.n_nodes, "edges_rank0": wl.n_edges, "ddpm_steps": 0 if wl.decode_only else T_STEPS,
                       "timed_region": "CA traces + noise in HBM -> xyz in HBM: features + layer-0 edge terms + adaLN "
                                       "vectors + CG graph + DDPM loop + VQ + IC decode + ic_to_xyz"
                                       + (" + all-gather of coordinates" if world > 1 else ""),
                       "parallelism": (f"units sharded x{world} (LPT), no data-path collective" if strong
                                       else f"replicas x{world}"),
                       "streams": ("two half-jobs on two HIP streams per GPU (codlad_amd.engine.Denoiser.sample, jobs of "
                                   ">= 8 192 nodes); the roofline's per-launch times come from a separate single-stream pass"
                                   if args.streams is None and not wl.decode_only and wl.job.n_nodes >= wl.den.SPLIT_MIN_NODES
                                   else "one HIP stream")},
            "roofline": roofline,
        }
        result.update(extra)
        if world == 1 and args.config == "cfg2" and not args.no_cfg5:
            result["cfg5"] = cfg5_section(device)       # BASELINE configuration 5 (the other 1-GPU configuration) in the same line
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args.config)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
