"""NOT a test module: one process, backend "nccl" (= RCCL), world size 1, on cuda:0.  Runs the collectives of
codlad_amd.parallel on DEVICE tensors through RCCL itself (the multi-rank tests stage through the host over gloo, two
processes on one GPU being something RCCL refuses): header + blob broadcast with rebind, checksum all-gather, ragged
coordinate all-gather, the bench's MAX all-reduce.  python tests/nccl_single_rank_worker.py <port>"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1], HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from codlad_amd import parallel, synth, weights
        den = weights.DenoiserWeights(synth.denoiser_state_dict(1234), dev)
        dec = weights.DecoderWeights(synth.vqvae_state_dict("K3", "PDB", 4321), dev, *synth.norm_stats("PDB", "K3"))
        before = den.blob.data.clone()
        sums = parallel.broadcast_weights(den, dec)
        assert torch.equal(before, den.blob.data) and den.generation == 1 and len(sums) == 2
        xyz = [torch.randn(7, 3, device=dev), torch.randn(11, 3, device=dev)]
        got = parallel.gather_coordinates(xyz, 1)
        assert len(got) == 1 and got[0].is_cuda and torch.equal(got[0], torch.cat([x.reshape(-1) for x in xyz]))
        empty = parallel.gather_coordinates([], 1)
        assert empty[0].numel() == 0
        t = torch.tensor([1.25], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t) == 1.25
        m = torch.nn.Linear(4, 3).to(dev)
        w0 = m.weight.detach().clone()
        parallel.broadcast_module_state(m)
        assert torch.equal(m.weight, w0)
        dist.barrier()
        print("nccl single-rank collectives ok: backend", dist.get_backend())
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
