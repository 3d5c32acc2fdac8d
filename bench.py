#!/usr/bin/env python3
"""Benchmark of the CODLAD sampling hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): synthetic PED-shaped test set - 4 proteins with
L = 46/87/92/129 residues, 10 frames each, num_ensemble = 10 -> 400 structures per GPU, 100-step
respaced DDPM with the mpnn_diffusion denoiser, then de-normalise + VQ (4096 codes) + IC_Decoder
(N6) + ic_to_xyz.  One "step" of this benchmark = that whole job, noise tensor resident in HBM
to all-atom coordinates resident in HBM.  Ranks are independent replicas of the job on different
seeds (weak scaling); weights are broadcast from rank 0 over RCCL before timing and every rank's
coordinates are all-gathered inside the timed region.

Prints ONE JSON line (rank 0).  `roofline` times the dominant kernel (the per-edge message MLP,
edge_kernel<false>) with HIP events on its own stream; `cpu_baseline` times the CPU oracle
(oracle/, a port of the reference's PyTorch-CPU path) on a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PED_LENGTHS = (46, 87, 92, 129)
N_FRAMES = 10
N_ENSEMBLE = 10
T_STEPS = 100
WEIGHT_SEED, VAE_SEED = 1234, 4321
FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, matrix FP32


def algorithmic_flop_per_structure(L):
    """per denoiser step per real sample 2*(786432*L*K + 787584*L + 819200),
    K = min(64, L); features once per structure 2*37760*L*K."""
    K = min(64, L)
    return T_STEPS * 2 * (786432 * L * K + 787584 * L + 819200) + 2 * 37760 * L * K


class Workload:
    """cfg 2 resident on one GPU."""

    def __init__(self, device, rank, precision="f16x3"):
        from codlad_amd import synth
        from codlad_amd.engine import Decoder, Denoiser
        from codlad_amd.diffusion_and_flow.schedule import Tables, named_betas, space_timesteps
        self.device = device
        self.precision = precision
        self.den = Denoiser(synth.denoiser_state_dict(WEIGHT_SEED), device, precision=precision)
        mean, std = synth.norm_stats("PED", "N6")
        self.dec = Decoder(synth.vqvae_state_dict("N6", "PED", VAE_SEED), device, mean, std)
        self.tables = Tables(named_betas("linear", 1000), space_timesteps(1000, str(T_STEPS)))
        self.proteins = [synth.make_protein(L, 100 * rank + 1000 + i, n_frames=N_FRAMES)
                         for i, L in enumerate(PED_LENGTHS)]
        xyz_list, z_list, sample_struct, self.groups = [], [], [], []
        for prot in self.proteins:
            frames = torch.from_numpy(prot["xyz_full"])[:, 1:-1]
            z = torch.from_numpy(prot["z_full"])[1:-1]
            first = len(xyz_list)
            for f in range(N_FRAMES):
                xyz_list.append(frames[f])
                z_list.append(z)
            members = [first + f for f in range(N_FRAMES) for _ in range(N_ENSEMBLE)]
            self.groups.append((len(sample_struct), len(members), prot))
            sample_struct += members
        self.n_structures = len(sample_struct)
        self.structures = self.den.prepare_structures(xyz_list, z_list)
        self.job = self.den.make_job(self.structures, sample_struct)
        # decoder-side tables (host preprocessing in the reference: CG_nbr_list, info)
        ni = self.job.node_info
        self.cg_z = ni[:, 3].contiguous()
        self.cg_xyz = self.structures.xyz[ni[:, 0].long()].contiguous()
        self.csr = self.dec.build_csr(self.cg_xyz, self.job.sample_lens)   # CG graph within 21 A, on the device
        self.n_edges = int(ni[:, 2].sum())
        self.ca_full = []
        for start, count, prot in self.groups:
            frames = torch.from_numpy(prot["xyz_full"]).to(device)
            idx = torch.arange(N_FRAMES, device=device).repeat_interleave(N_ENSEMBLE)
            self.ca_full.append(frames[idx].contiguous())
        g = torch.Generator(device=device)
        g.manual_seed(42 + rank)
        self.x_T = torch.randn(self.job.n_nodes, 3, generator=g, device=device)
        self.noise = torch.randn(T_STEPS, self.job.n_nodes, 3, generator=g, device=device)
        self.flop = sum(N_FRAMES * N_ENSEMBLE * algorithmic_flop_per_structure(L) for L in PED_LENGTHS)

    def run(self):
        """noise (HBM) -> all-atom coordinates (HBM) for the 400 structures."""
        x0 = self.den.sample(self.job, self.x_T, self.noise, self.tables)
        idx, zq, _lat = self.dec.vq(x0)
        ic = self.dec.ic_decode(zq, self.cg_z, self.cg_xyz, csr=self.csr)
        out = []
        for (start, count, prot), ca in zip(self.groups, self.ca_full):
            L = prot["n_cg"]
            a = int(self.job.sample_off[start])
            b = int(self.job.sample_off[start + count])
            out.append(self.dec.ic_to_xyz(ca, ic[a:b].view(count, L, 13, 3), prot["info"]))
        return out, idx

    def time_dominant_kernel(self, n_launch=20):
        """Average duration of one message-kernel launch (encoder layer 1: both GEMM layers over
        every edge of the job, edge state read from HBM - what 5 of the 6 message launches of a step
        look like), HIP events on the stream the kernel runs on."""
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
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(n_launch):
                launch(which)
            e1.record(stream)
            e1.synchronize()
            res[name] = e0.elapsed_time(e1) / n_launch * 1e-3
        return res


def cpu_baseline():
    """The CPU oracle (port of the reference's PyTorch-CPU path) on a bounded sample: 4 frames of
    the L=87 protein, 20 of the 100 DDPM steps run as the reference runs them (batch duplicated,
    test.py:505; CA features recomputed every step), plus the decoder tail; extrapolated linearly
    to 100 steps (every step costs the same)."""
    from codlad_amd import synth
    from oracle import denoiser as oden, sampler as osam, vae_decode as odec
    torch.set_grad_enabled(False)
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    cores = min(cores, 16)   # a one-GPU box owns 16 host cores; more threads only oversubscribe
    torch.set_num_threads(cores)
    sd = synth.denoiser_state_dict(WEIGHT_SEED)
    vsd = synth.vqvae_state_dict("N6", "PED", VAE_SEED)
    mean, std = synth.norm_stats("PED", "N6")
    L, B, Tsub = 87, 4, 20
    prot = synth.make_protein(L, 1001, n_frames=B)
    batch = synth.make_batch(prot)
    cg_z, cg_xyz, mask = oden.batch_to_dense(batch)
    dup = lambda t: torch.cat([t, t])  # noqa: E731
    z = synth.gaussian((2 * B, L, 3), 1)
    eps = synth.gaussian((Tsub, 2 * B, L, 3), 2)
    t0 = time.perf_counter()
    x = osam.p_sample_loop(sd, Tsub, z, eps, dup(cg_xyz), dup(cg_z), dup(mask))
    t_loop = time.perf_counter() - t0
    t0 = time.perf_counter()
    idx, ic = odec.latent_decode(vsd, odec.denormalise(x[:B], mean, std), batch)
    odec.ic_to_xyz(batch["OG_CG_nxyz"].reshape(-1, L + 2, 4), ic.reshape(-1, L, 13, 3), prot["info"])
    t_dec = time.perf_counter() - t0
    per_struct = (t_loop * (T_STEPS / Tsub) + t_dec) / B
    # the same work without the reference's redundancy (no duplicated batch, features once): what a
    # tuned CPU run of this algorithm would do, reported beside the headline baseline
    t0 = time.perf_counter()
    osam.p_sample_loop(sd, Tsub, z[:B], eps[:, :B], cg_xyz, cg_z, mask, hoist_features=True)
    t_dedup = time.perf_counter() - t0
    dedup = 1.0 / ((t_dedup * (T_STEPS / Tsub) + t_dec) / B)
    return {"value": 1.0 / per_struct, "unit": "structures/s", "cores": cores, "kind": "port",
            "value_deduplicated": dedup,
            "sample": f"oracle (PyTorch-CPU fp32, {cores} threads): L=87, {B} frames, {Tsub} of {T_STEPS} DDPM steps "
                      f"as the reference runs them (2x duplicated batch, features recomputed per step) + decode, "
                      f"extrapolated x{T_STEPS // Tsub}; {t_loop:.2f}s loop + {t_dec:.3f}s decode"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=["f16x3", "f16x4", "f32"], default="f16x3",
                    help="contraction mode: f16x3 (default) / f16x4 = fp32 operands split into fp16 hi+lo halves on "
                         "the f16 matrix pipe, 3 or 4 cross products per fp32 product, fp32 accumulation "
                         "(fp32-equivalent); f32 = v_mfma_f32_32x32x2_f32")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
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

    wl = Workload(device, rank, args.precision)
    if world > 1:
        # weights travel once, rank 0 -> all, as one buffer each (RCCL broadcast over xGMI)
        from codlad_amd.parallel import broadcast_weights
        broadcast_weights(wl.den.weights, wl.dec.weights)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        xyz, idx = wl.run()
        if world > 1:
            from codlad_amd.parallel import gather_coordinates
            gather_coordinates(xyz, world)
        return xyz

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)

    kern = wl.time_dominant_kernel()
    # HBM bytes per launch of the dominant kernel: PMC counters need rocprofv3, so the figure comes from
    # the committed profile of this same workload (profiles/README.md says how it was taken)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_final_traffic.json")
    if args.precision != "f32" and os.path.exists(tpath):
        with open(tpath) as f:
            traffic = json.load(f)["hbm_bytes_per_launch"]
    if rank == 0:
        total_structs = wl.n_structures * world * args.steps
        value = total_structs / dt
        # dominant kernel: layers 1-2 of the encoder message MLP, algorithmic 2*(384*128 + 128*128)
        # FLOP per edge (reference protein_mpnn_utils.py:240-243; W3 runs in the node kernel)
        flop_launch = 2.0 * (384 * 128 + 128 * 128) * wl.n_edges
        terms = {"f16x3": 3, "f16x4": 4, "f32": 0}[args.precision]
        achieved = flop_launch / kern["message"] / 1e12
        result = {
            "metric": "sampled all-atom structures/sec (100-step DDPM, PED)",
            "value": value, "unit": "structures/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "precision": (f"{args.precision}: fp32 operands split into fp16 hi+lo (|eps| <= 2^-22), {terms} f16 MFMAs "
                          "per product, fp32 accumulate; all other arithmetic fp32" if terms else
                          "f32: v_mfma_f32_32x32x2_f32, all arithmetic fp32"),
            "config": {"workload": "cfg2: PED-shaped test set, 4 proteins L=46/87/92/129 x 10 frames x "
                                   "num_ensemble 10 = 400 structures per GPU, 100-step DDPM (mpnn_diffusion) + "
                                   "VQ(4096x3) + IC_Decoder N6 + ic_to_xyz",
                       "structures_per_gpu": wl.n_structures, "nodes_per_gpu": wl.job.n_nodes,
                       "edges_per_gpu": wl.n_edges, "ddpm_steps": T_STEPS, "parallelism": f"replicas x{world}"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "kernel": ("msg_kernel_h " if terms else "edge_kernel<false> ") +
                                   "(encoder message MLP, layers 1-2)",
                         "peak_note": "fp32 matrix peak of MI355X_MICROARCH.md; algorithmic FLOP of SURVEY.md 8d "
                                      "(the kernel executes half of them after the W1 split, and in the split-fp16 "
                                      "modes runs them on the f16 pipe), so frac may exceed 1",
                         "f16_pipe_frac": (terms * 2.0 * 2 * 128 * 128 * wl.n_edges / kern["message"] / 2.5e15
                                           if terms else None),
                         "launch_ms": kern["message"] * 1e3,
                         "algorithmic_flop_per_launch": flop_launch,
                         "executed_flop_per_launch": 2.0 * 2 * 128 * 128 * wl.n_edges,
                         "edge_update_launch_ms": kern["edge_update"] * 1e3,
                         "whole_job_algorithmic_tflops": wl.flop * world * args.steps / dt / 1e12},
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline()
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
