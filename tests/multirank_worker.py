"""One rank of the multi-process GPU test (tests/test_00_multirank_gpu.py): NOT a test module.

    python tests/multirank_worker.py <rank> <world> <port> <outdir> [mode]

Every rank is its own process on cuda:0 with the gloo backend (a one-GPU box; on an 8-GPU node the same code
runs one rank per GPU over RCCL).  Rank r > 0 starts from DIFFERENT weights and must end up with rank 0's - blocks,
BLOCK EXPONENTS, contraction mode and model flags - after `broadcast_weights`; the 12 units of a small job are dealt
by `shard_units`, sampled and decoded on their rank, and the coordinates all-gathered; rank 0 writes what it gathered.
Modes (what the non-source ranks start from / what rank 0 holds):
  r1_envelope  rank 0: the default weights (every block exponent 0); rank 1: `small_first_1e-2` (exponents 7 / -4 / 6 / -3)
  r0_envelope  the other way round: the job's weights are `small_first_1e-2`, rank 1 starts from the default ones
  r1_empty     rank 1 holds no weights at all and a decoder blob of the wrong layout (`*.empty`)
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LENGTHS, N_FRAMES, N_ENSEMBLE, T = [40, 70, 101], 2, 2, 10
ENVELOPE = "small_first_1e-2"
VAE_TYPE = {"r1_envelope": ("N6", "PED"), "r0_envelope": ("N6", "PED"), "r1_empty": ("K3", "PDB")}


def job_config(mode, rank):
    """The pipeline.Config rank `rank` builds BEFORE the broadcast (rank 0's is the job's)."""
    from tests import cases, pipeline
    vae_type, dataname = VAE_TYPE[mode]
    kw = dict(T=T)
    if rank == 0:
        if mode == "r0_envelope":
            kw["denoiser_sd"] = cases.envelope_state_dict(ENVELOPE)
    elif mode == "r1_envelope":
        kw.update(denoiser_sd=cases.envelope_state_dict(ENVELOPE), vae_seed=pipeline.VAE_SEED + 77 * rank)
    elif mode == "r0_envelope":
        kw.update(weight_seed=pipeline.WEIGHT_SEED + 77 * rank, vae_seed=pipeline.VAE_SEED + 77 * rank)
    else:
        kw["no_weights"] = True
    return pipeline.Config("mr", LENGTHS, N_FRAMES, N_ENSEMBLE, vae_type, dataname, **kw)


def exponent_table(w):
    return [[getattr(w.struct.enc_h[l], n) for n in ("e1", "e2", "e3", "e11", "e12", "e13", "e_in", "e_out")] +
            [getattr(w.struct.dec_h[l], n) for n in ("e1", "e2", "e3", "e_in", "e_out")] for l in range(3)]


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    mode = sys.argv[5] if len(sys.argv) > 5 else "r1_envelope"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_grad_enabled(False)
    from codlad_amd import parallel
    from tests import pipeline
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank 0: the weights of the job; other ranks: something else (or nothing), to be overwritten by the broadcast
        cfg = job_config(mode, rank)
        before = cfg.den.weights.blob.data.clone()
        exp_before = exponent_table(cfg.den.weights)
        parallel.broadcast_weights(cfg.den.weights, cfg.dec.weights)
        changed = not torch.equal(before, cfg.den.weights.blob.data)
        assert changed == (rank != 0), "broadcast must overwrite exactly the non-source ranks' blobs"
        # the struct the kernels read must now hold RANK 0's exponents on every rank
        exps = torch.tensor(exponent_table(cfg.den.weights))
        every = [torch.zeros_like(exps) for _ in range(world)]
        dist.all_gather(every, exps)
        assert all(torch.equal(e, every[0]) for e in every), "block exponents differ between ranks after the broadcast"
        if mode != "r1_empty" and rank != 0:
            assert exponent_table(cfg.den.weights) != exp_before, "the test must start rank 1 from OTHER exponents"
        want_nonzero = mode == "r0_envelope"
        assert (int(exps.abs().sum()) != 0) == want_nonzero
        assert cfg.dec.weights.angle == (VAE_TYPE[mode][0] != "N6")
        costs = [parallel.unit_cost(LENGTHS[p]) for p, _f, _m in cfg.units]
        shards = parallel.shard_units(costs, world)
        mine = cfg.run_units(shards[rank])
        xyz = [mine[u][2] for u in shards[rank]]
        gathered = parallel.gather_coordinates(xyz, world)
        if rank == 0:
            out = {}
            for r in range(world):
                flat, o = gathered[r].cpu().numpy(), 0
                for u in shards[r]:
                    p = cfg.units[u][0]
                    n = int(cfg.proteins[p]["info"][0].numel()) * 3
                    out[f"xyz_{u}"] = flat[o:o + n].reshape(-1, 3)
                    o += n
                assert o == flat.size
            out["shard_sizes"] = np.array([len(s) for s in shards])
            out["exponents"] = exps.numpy()
            np.savez(os.path.join(outdir, "gathered.npz"), **out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
