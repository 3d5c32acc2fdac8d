"""One rank of the multi-process GPU test (tests/test_00_multirank_gpu.py): NOT a test module.

    python tests/multirank_worker.py <rank> <world> <port> <outdir>

Every rank is its own process on cuda:0 with the gloo backend (a one-GPU box; on an 8-GPU node the same code
runs one rank per GPU over RCCL).  Rank r > 0 starts from DIFFERENT weights (other seeds) and must end up with
rank 0's after `broadcast_weights` + `rebind()`; the 12 units of a small job are dealt by `shard_units`, sampled
and decoded on their rank, and the coordinates all-gathered; rank 0 writes what it gathered.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LENGTHS, N_FRAMES, N_ENSEMBLE, T = [40, 70, 101], 2, 2, 10


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_grad_enabled(False)
    from codlad_amd import parallel
    from tests import pipeline
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank 0: the weights of the job; other ranks: something else, to be overwritten by the broadcast
        cfg = pipeline.Config("mr", LENGTHS, N_FRAMES, N_ENSEMBLE, "N6", "PED", T=T,
                              weight_seed=pipeline.WEIGHT_SEED + 77 * rank, vae_seed=pipeline.VAE_SEED + 77 * rank)
        before = cfg.den.weights.blob.data.clone()
        ptr_before = cfg.den.weights.struct.enc[1].W2
        parallel.broadcast_weights(cfg.den.weights, cfg.dec.weights)
        changed = not torch.equal(before, cfg.den.weights.blob.data)
        assert changed == (rank != 0), "broadcast must overwrite exactly the non-source ranks' blobs"
        assert cfg.den.weights.struct.enc[1].W2 == ptr_before        # same storage, pointers re-derived
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
            np.savez(os.path.join(outdir, "gathered.npz"), **out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
