#!/usr/bin/env python3
"""Inference entry point, drop-in for the sampling part of the reference's test.py
(`--experiment latent` and `--experiment recon`), running on the HIP path.

Kept from the reference (test.py:894-961): flag names and defaults, checkpoint locations
(`./results/<exp>/protein_weights_{best,last,step_N}.pt` with `net_model` / `ema_model`, VQ-VAE
directories of utils/model_module.py), the call sequence build_model -> create_diffusion ->
p_sample_loop -> get_norm_feature -> latent_decode -> ic_to_xyz, and the output directory.
Different by design: all ensemble members of a batch of frames are sampled in ONE launch set (the
reference loops over them); the C2 prior call that only supplies `mask` (test.py:495) is replaced by
the length mask; mdtraj I/O is replaced by saving coordinates as .npy (and multi-model PDB + .xtc, --save_pdb).
Data: `--pdb_files ens.pdb ...` (multi-model PDB ensembles -> the reference's load_dataset without mdtraj),
`--data_process --data_files f.pkl ...` (pickled per-frame dicts, as the reference's --data_process branch reads
them) or `--synthetic` (no PED/PDB/Atlas files ship with the reference).
"""
import argparse
import os
import pickle
import time

import numpy as np
import torch

from codlad_amd import metrics, synth
from codlad_amd.diffusion_and_flow import create_diffusion
from codlad_amd.models.latent_model import MPNN_models
from codlad_amd.utils.dataset_module import CG_collate, get_norm_feature
from codlad_amd.utils.model_module import build_vae, get_vae_model, load_decoder_state
from codlad_amd.utils.utils_ic import ic_to_xyz


def build_model(args):
    if "mpnn" not in args.backbone:
        raise NotImplementedError(f"backbone {args.backbone!r}: only mpnn_diffusion is built")
    return MPNN_models[args.backbone](input_size=args.latent_size, unconditional=not args.cond,
                                      diffusion=args.model, self_condition=args.self_condition)


def load_denoiser(args, device, load=True):
    """load=False (ranks > 0 of a multi-GPU run): the module with its constructor's initialisation and NO file
    access - rank 0's weights arrive by broadcast (`parallel.broadcast_module_state`)."""
    model = build_model(args)
    if not load:
        return model.to(device).eval()
    if args.synthetic_weights:
        model.load_state_dict(synth.denoiser_state_dict(1234, self_condition=args.self_condition,
                                                        flow=args.model != "diffusion"), strict=True)
    else:
        tag = {"best": "best", "last": "last"}.get(args.model_step, f"step_{args.model_step}")
        ckpt = torch.load(f"./results/{args.exp}/protein_weights_{tag}.pt", map_location="cpu")
        sd = ckpt["net_model" if args.ckpt_type == "net" else "ema_model"]
        try:
            model.load_state_dict(sd, strict=True)
        except RuntimeError:
            model.load_state_dict({k[7:]: v for k, v in sd.items()}, strict=True)
    return model.to(device).eval()


def load_vae(args, device, load=True):
    """The VQ-VAE; with its e3nn encoder when the run needs it (`--experiment recon` encodes the batch's atoms)."""
    enc = args.experiment == "recon"
    if not load:
        return build_vae(args.vae_type, with_encoder=enc).to(device).eval()
    if args.synthetic_weights:
        vae = build_vae(args.vae_type, with_encoder=enc)
        sd = synth.vqvae_state_dict(args.vae_type, args.data_type, 4321)
        if enc:
            sd.update({"encoder." + k: v for k, v in synth.encoder_state_dict(778).items()})
            sd.update({k: v for k, v in vae.state_dict().items() if k.endswith(".offset")})
        load_decoder_state(vae, sd)
        return vae.to(device).eval()
    vae, _params = get_vae_model(args.vae_type, device=device, modelnum=args.modelnum, with_encoder=enc)
    return vae.to(device).eval()


def load_cvae(args, device, load=True):
    """The conditional VAE (`--cvae_type C2`, reference test.py:317-320): its CG prior conditions `--cond` models and IS
    the generator of `--experiment genzprot`."""
    if not load:
        return build_vae(args.cvae_type).to(device).eval()
    if args.synthetic_weights:
        cvae = build_vae(args.cvae_type)
        sd = {k: v for k, v in cvae.state_dict().items()}
        sd.update({"prior_net." + k: v for k, v in synth.prior_state_dict(777).items()})
        sd.update({"encoder." + k: v for k, v in synth.encoder_state_dict(779).items()})
        sd.update(synth.decoder_state_dict(4322, angle=False))
        load_decoder_state(cvae, sd)
        return cvae.to(device).eval()
    cvae, _params = get_vae_model(args.cvae_type, device=device)
    return cvae.to(device).eval()


_TOPOLOGY = {}                     # output name -> (residue names, atom names per residue) where known (--save_pdb)
MAX_FRAMES_PER_BATCH = 96          # reference utils/dataset_module.py:220-226: batch_size = min(n_frames, 96)


def chunk_plan(n_frames, max_frames=MAX_FRAMES_PER_BATCH):
    """[(first frame, end frame)] of the batches one data file is cut into."""
    bs = min(n_frames, max_frames)
    return [(b, min(b + bs, n_frames)) for b in range(0, n_frames, bs)]


def output_name(name, chunk, n_chunks):
    """File stem of one batch's coordinates.  A data file of more than 96 frames is cut into several batches
    (possibly dealt to different ranks): each gets its own, index-tagged file, so no batch overwrites another;
    a single-batch file keeps the plain name."""
    return name if n_chunks == 1 else f"{name}_b{chunk:05d}"


def iter_batches(args):
    """Yields (output name, batch dict, info)."""
    if args.synthetic:
        lengths = {"PED": (46, 87, 92, 129), "PDB": (60, 120, 200), "Atlas": (39, 155, 505)}[args.data_type]
        for i, L in enumerate(lengths):
            prot = synth.make_protein(L, 1000 + i, n_frames=args.synthetic_frames,
                                      phospho=args.vae_type != "N6")
            plan = chunk_plan(args.synthetic_frames)
            names = [synth.IDX2THR[int(z)] for z in prot["z_full"]]
            for c, (a, b) in enumerate(plan):
                out = output_name(f"synthetic_L{L}", c, len(plan))
                _TOPOLOGY[out] = (names, [synth.PDB_ATOM_ORDER[nm] for nm in names])
                batch = synth.make_batch(prot, range(a, b))
                if getattr(args, "experiment", "latent") == "recon":      # the encoder reads the all-atom side of the batch
                    batch.update(synth.make_atoms(prot, range(a, b), seed=1000 + i))
                yield out, batch, prot["info"]
        return
    if getattr(args, "pdb_files", None):
        # reference test.py:424-436 -> load_dataset(f"{dir}/{name}", params): a multi-model PDB ensemble, here without
        # mdtraj (utils/dataset_builder.py; internal coordinates by codlad_xyz_to_ic).  Cut-offs: the VAE's modelparams
        # (the shipped ones: atom 9.0, CG 21.0, bond order 2), overridable on the command line.
        from codlad_amd.utils.dataset_module import load_dataset
        params = {"atom_cutoff": args.atom_cutoff, "cg_cutoff": args.cg_cutoff, "edgeorder": args.edgeorder}
        for path in args.pdb_files:
            stem = path[:-4] if path.endswith(".pdb") else path
            loader, info_dict, _n_atoms, _n_cgs, _z, top = load_dataset(stem, params, device="cuda")
            n_batches = len(loader)
            for c, batch in enumerate(loader):
                out = output_name(os.path.basename(stem), c, n_batches)
                _TOPOLOGY[out] = (["GLY"] + top.res_names + ["GLY"], [["CA"]] + top.atom_names + [["CA"]])
                yield out, batch, info_dict[0]
        return
    if not args.data_process:
        raise SystemExit("xtc loading is not built: use --pdb_files (multi-model PDB), --data_process --data_files ... or --synthetic")
    for path in args.data_files:
        with open(path, "rb") as f:
            testset, info = pickle.load(f)
        plan = chunk_plan(len(testset))
        for c, (a, b) in enumerate(plan):
            yield output_name(os.path.basename(path), c, len(plan)), CG_collate([testset[i] for i in range(a, b)]), info


EVAL_KEYS = ("nxyz", "num_atoms", "bond_edge_list", "nbr_list", "bb_NO_list", "interaction_list", "pi_pi_list", "ic",
             "mask", "mask_xyz_list")


class Evaluation:
    """The evaluation block of the reference's loop (test.py:566-668, 707-785) on the device: per batch and ensemble
    member the reconstruction / coordinate / bond-graph / clash / interaction losses (codlad_eval_metrics, one launch)
    and the bond-graph validity (codlad_bond_graph_counts); per data file the summary the reference prints."""

    def __init__(self):
        self.rows, self.valid, self.ged, self.rmsd = [], [], [], []
        self.recon, self.true = [], None

    def add(self, batch, ic_recon, xyz_recon, n_atoms):
        xyz = batch["nxyz"][:, 1:].clone()
        xr = xyz_recon.reshape(-1, 3).clone()
        mask_xyz = batch["mask_xyz_list"]
        xyz[mask_xyz] *= 0                                   # test.py:585-586
        xr[mask_xyz] *= 0
        r = metrics.all_results(ic_recon, batch["ic"], batch["mask"], xr, xyz, batch["bond_edge_list"], batch["nbr_list"],
                                batch["bb_NO_list"], batch["interaction_list"], batch["pi_pi_list"])
        _hv, av, _hg, ag = metrics.valid_ratio_and_cut_off_result(xyz, xr, batch["num_atoms"], batch["nxyz"][:, 0].cpu())
        self.rows.append({k: float(v) for k, v in r.items()})
        self.valid += av
        self.ged += ag
        a, b = xr.reshape(-1, n_atoms, 3), xyz.reshape(-1, n_atoms, 3)
        self.rmsd.append((a - b).pow(2).sum(-1).mean(-1).sqrt().mean().item())      # unaligned all-atom RMSD, test.py:661
        self.recon.append(a.cpu())
        self.true = b.cpu()

    def report(self, name, args):
        mean = lambda k: float(np.mean([r[k] for r in self.rows]))  # noqa: E731
        stats = {"data_name": name, "data_type": args.data_type, "num_ensemble": args.num_ensemble,
                 "experiment": args.experiment, "test_all_recon": float(np.mean(self.rmsd)),
                 "test_xyz": mean("loss_xyz"), "test_graph": mean("loss_graph"), "test_nbr": mean("loss_nbr"),
                 "test_inter": mean("loss_inter"), "test_pi_pi": mean("loss_pi_pi"),
                 "test_all_valid_ratio": float(np.mean(self.valid)), "test_all_ged": float(np.mean(self.ged)),
                 "diversity": metrics.compute_div(self.recon, self.true) if len(self.recon) > 1 else 0}
        print("############## vvvvvvvvv result test_stats:")
        for k, v in stats.items():
            print(k, v)
        print("############## ^^^^^^^^^ result test_stats:")
        return stats


def unit_generator(args, batch_id, device):
    """The noise source of one batch (a data file's <= 96 frames x num_ensemble members): seeded by the run's seed
    and the batch's position in the data set alone, so a structure is sampled from the same numbers on 1 or 8 GPUs."""
    gen = torch.Generator(device=device)
    gen.manual_seed(args.seed + args.sample_index + batch_id)
    return gen


def run_sampling(model, args, x, mask=None, batch=None):
    """Flow-matching sampling, reference test.py:214-250: integrate dx/dt = model(x, t) from t = 0 (noise) to 1 over
    t_span = linspace(0, 1, --steps) with --method / --atol / --rtol; torchdiffeq.odeint is replaced by
    codlad_amd.diffusion_and_flow.ode.odeint (euler / midpoint / rk4 / dopri5)."""
    from codlad_amd.diffusion_and_flow.ode import odeint
    t_span = torch.linspace(0, 1, args.steps).to(x.device)
    fwd = lambda t, x_in: model.forward(x_in, t, None, mask=mask, batch=batch)  # noqa: E731
    return odeint(fwd, x, t_span, rtol=args.rtol, atol=args.atol, method=args.method)[-1]


def main(args):
    if not torch.cuda.is_available():
        raise SystemExit("test.py (codlad_amd) needs an MI355X: there is no CPU path")
    # one process per GPU under torch.distributed.run: batches (independent units) are dealt to the
    # ranks longest-first, every rank samples and decodes its own, rank 0 reports the totals
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("CODLAD_DIST_BACKEND", "nccl")   # "gloo": rehearsal of N > 1 on a one-GPU box
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = f"cuda:{dev_index}"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.set_grad_enabled(False)
    # the global streams are seeded like the reference's (test.py:256) but NOT used for the latents: every batch
    # draws its noise from its own generator (unit_generator), so the numbers a structure gets do not depend on the
    # world size or on which rank the batch was dealt to
    torch.manual_seed(args.seed + args.sample_index)
    np.random.seed(args.seed + args.sample_index)
    if args.cfg_scale > 1.0:
        raise NotImplementedError("cfg_scale > 1 calls model.forward_with_cfg, which the reference model "
                                  "does not define (dead path)")
    # only rank 0 reads checkpoints; the other ranks receive its weights (reference: one process, test.py:264-286)
    vae = load_vae(args, device, load=rank == 0)
    model, cvae = None, None
    if args.experiment == "genzprot":
        cvae = load_cvae(args, device, load=rank == 0)
    if args.experiment == "latent":
        model = load_denoiser(args, device, load=rank == 0)
        diffusion = None if args.model != "diffusion" else create_diffusion(str(args.num_sampling_steps), noise_schedule=args.noise_schedule,
                                     predict_xstart=args.predict_xstart,
                                     rescale_learned_sigmas=args.rescale_learned_sigmas,
                                     # reference test.py:297-303
                                     self_condition=hasattr(model, "self_condition") and args.self_condition)
    elif args.experiment not in ("recon", "genzprot"):
        raise NotImplementedError(f"experiment {args.experiment!r}: latent, recon and genzprot are built")
    if world > 1:
        from codlad_amd import parallel
        mods = [m for m in (model, vae, cvae) if m is not None]
        parallel.broadcast_module_state(*mods, src=0)
        # every rank packs the same parameters into its device blobs; prove it (and leave a record of the rank
        # count and backend the collective ran on)
        sums = [m.engine().weights.checksum() for m in mods]
        parallel.verify_checksums(sums, what="weights broadcast from rank 0")
    save_dir = f"./logs/generated_samples_{args.sample_index}_{args.model_step}/{args.exp}_{args.data_type}"
    os.makedirs(save_dir, exist_ok=True)
    total, t_all = 0, time.time()
    units = [(g,) + u for g, u in enumerate(iter_batches(args))]      # g: the batch's id, the same on every world size
    if world > 1:
        from codlad_amd.parallel import shard_units, unit_cost
        costs = [int(b["num_CGs"].shape[0]) * unit_cost(int(b["num_CGs"][0])) for _g, _n, b, _i in units]
        units = [units[u] for u in shard_units(costs, world)[rank]]
    for g, name, batch, info in units:
        gen = unit_generator(args, g, device)
        batch = {k: (v.to(device) if hasattr(v, "to") else v) for k, v in batch.items()}
        B = int(batch["num_CGs"].shape[0])
        L = int(batch["num_CGs"][0])
        E = args.num_ensemble
        st = time.time()
        mask = torch.ones(B * E, L, dtype=torch.bool, device=device)
        # E ensemble members of every frame = the batch repeated E times along the sample axis
        rep = {k: v for k, v in batch.items()}
        if args.experiment == "latent":
            z = torch.randn(B * E, L, args.latent_size, device=device, generator=gen)
            if args.model == "diffusion":
                samples = diffusion.p_sample_loop(model.forward, z.shape, z, clip_denoised=False,
                                                  model_kwargs=dict(y=None, mask=mask, batch=rep), device=device,
                                                  step_noise=diffusion._draw_noise(z, generator=gen))
            else:                                               # --model fm / icfm / otcfm ...: ODE sampling
                samples = run_sampling(model, args, z, mask=mask, batch=rep)
            samples = get_norm_feature(samples, args.vae_type, norm_channel=args.norm, norm_single=args.norm_single,
                                       norm_in=False, dataname=args.data_type)
        elif args.experiment == "recon":
            # reference test.py:501: the VQ-VAE's own encoder on the batch's atoms, un-quantized (latent_decode
            # quantizes); deterministic, so the E ensemble members of a frame are equal, as in the reference's loop
            feat = vae.get_latent_wovq(batch)[0]
            samples = feat.repeat(E, 1, 1)
        else:
            # --experiment genzprot (reference test.py:495-498, 559): a sample of the conditional prior per member
            samples = torch.cat([cvae.get_latent_cg(batch, generator=gen)[0] for _ in range(E)], 0)
        decoder_model = cvae if args.experiment == "genzprot" else vae
        nres = L + 2
        og = batch["OG_CG_nxyz"].reshape(-1, nres, 4)
        xyz_all = []
        evaluation = Evaluation() if all(k in batch for k in EVAL_KEYS) else None
        for e in range(E):
            _, ic_recon = decoder_model.latent_decode(samples[e * B:(e + 1) * B], mask[:B], batch)
            xyz_all.append(ic_to_xyz(og, ic_recon.reshape(-1, nres - 2, 13, 3), info))
            if evaluation is not None:                               # reference test.py:589-594, per ensemble member
                evaluation.add(batch, ic_recon, xyz_all[-1], xyz_all[-1].shape[1])
        xyz = torch.stack(xyz_all)                                   # [E, B, n_atoms, 3]
        torch.cuda.synchronize()
        dt = time.time() - st
        total += B * E
        if evaluation is not None:
            evaluation.report(name, args)
        np.save(os.path.join(save_dir, f"{name}_xyz_recon.npy"), xyz.cpu().numpy())
        if getattr(args, "save_pdb", False) and name in _TOPOLOGY:
            # reference test.py:787-796 writes the generated ensemble through mdtraj (.xtc + .pdb); here both directly, frames
            # of member 0 first (multi-model PDB in Angstrom, .xtc in nm as the format has it)
            from codlad_amd.utils.protein_module import write_pdb
            from codlad_amd.utils.xtc import write_xtc
            frames = xyz.reshape(-1, xyz.shape[2], 3).cpu().numpy()
            write_pdb(os.path.join(save_dir, f"generated_traj_{name}.pdb"), frames, *_TOPOLOGY[name])
            write_xtc(os.path.join(save_dir, f"generated_traj_{name}.xtc"), frames)
        print(f"{name}: {B} frames x {E} members, L={L}, {xyz.shape[2]} atoms: {dt:.2f}s "
              f"({B * E / dt:.1f} structures/s)", flush=True)
    if world > 1:
        import torch.distributed as dist
        tot = torch.tensor([total], dtype=torch.int64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tot)
        total = int(tot)
        dist.destroy_process_group()
    if rank == 0:
        print(f"done: {total} structures on {world} GPU(s) in {time.time() - t_all:.1f}s -> {save_dir}")


if __name__ == "__main__":
    p = argparse.ArgumentParser("parameters")
    # names and defaults as in the reference (test.py:894-961)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--exp", default="experiment_cifar_default")
    p.add_argument("--cond", action="store_true", default=False)
    p.add_argument("--backbone", type=str, default="mpnn_diffusion")
    p.add_argument("--latent_size", type=int, default=3)
    p.add_argument("--cfg_scale", type=float, default=1.0)
    p.add_argument("--model_step", type=str, default="best")
    p.add_argument("--vae_type", type=str, default="N6")
    p.add_argument("--cvae_type", type=str, default="C2")
    p.add_argument("--model", type=str, default="diffusion")
    p.add_argument("--num_sampling_steps", type=int, default=250)
    p.add_argument("--norm", action="store_true", default=True)
    p.add_argument("--norm_single", action="store_true", default=False)
    p.add_argument("--data_type", type=str, default="PED")
    p.add_argument("--data_process", action="store_true", default=False)
    p.add_argument("--num_ensemble", type=int, default=1)
    p.add_argument("--modelnum", type=int, default=-1)
    p.add_argument("--self_condition", action="store_true", default=False)
    p.add_argument("--predict_xstart", action="store_true", default=False)
    p.add_argument("--rescale_learned_sigmas", action="store_true", default=False)
    p.add_argument("--noise_schedule", type=str, default="linear", choices=["linear", "squaredcos_cap_v2"])
    p.add_argument("--experiment", type=str, default="latent", choices=["genzprot", "recon", "latent"])
    p.add_argument("--ckpt_type", type=str, default="net")
    p.add_argument("--sample_index", type=int, default=0)
    for ignored, kw in (("--compute_nfe", dict(action="store_true")), ("--iteration", dict(type=int, default=1000)),
                        ("--n_sample", dict(type=int, default=50000)), ("--dataset", dict(default="cifar10")),
                        ("--num_steps", dict(type=int, default=40)), ("--batch_size", dict(type=int, default=200)),
                        ("--feature_path", dict(type=str, default="./datasets/features_N6")),
                        ("--gcn_layernorm", dict(action="store_true", default=True)),
                        ("--forward_inf", dict(action="store_true", default=False))):
        p.add_argument(ignored, **kw)
    p.add_argument("--atol", type=float, default=1e-5)
    p.add_argument("--rtol", type=float, default=1e-5)
    p.add_argument("--method", type=str, default="dopri5")
    p.add_argument("--steps", type=int, default=2)
    # additions
    p.add_argument("--data_files", nargs="*", default=[], help="with --data_process: pickles of (list of frame dicts, info)")
    p.add_argument("--synthetic", action="store_true", help="synthetic PED/PDB/Atlas-shaped proteins")
    p.add_argument("--synthetic_frames", type=int, default=10)
    p.add_argument("--synthetic_weights", action="store_true", help="seeded random weights (no checkpoints ship)")
    p.add_argument("--save_pdb", action="store_true", help="also write the generated ensemble as a multi-model PDB and an .xtc trajectory")
    p.add_argument("--pdb_files", nargs="*", default=None,
                   help="multi-model PDB ensembles to build the test set from (the reference's load_dataset, without mdtraj)")
    p.add_argument("--atom_cutoff", type=float, default=9.0)
    p.add_argument("--cg_cutoff", type=float, default=21.0)
    p.add_argument("--edgeorder", type=int, default=2)
    main(p.parse_args())
