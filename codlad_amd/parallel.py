"""One process per GPU; independent structures shard embarrassingly.

No collective sits on the data path: every (protein, frame, ensemble member) unit is sampled and
decoded entirely on the rank that owns it.  RCCL (torch.distributed backend "nccl") is used twice:
  * once at start-up, rank 0 broadcasts the packed weight blobs (denoiser ~10 MB, decoder +
    codebook ~0.3 MB), one buffer each;
  * once per job, the coordinates of every rank's units are all-gathered (a few MB).
The same functions run on the gloo backend for the CPU tests (the tensors are then host tensors).
"""
import torch
import torch.distributed as dist


def shard_units(costs, world_size):
    """Longest-processing-time assignment of units to ranks.  costs[u] ~ L_u * min(64, L_u).
    Returns per-rank lists of unit indices (each sorted), every unit exactly once."""
    order = sorted(range(len(costs)), key=lambda u: (-costs[u], u))
    load = [0.0] * world_size
    shards = [[] for _ in range(world_size)]
    for u in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        shards[r].append(u)
        load[r] += costs[u]
    return [sorted(s) for s in shards]


def unit_cost(L):
    return L * min(64, L)


def _stage(t):
    """gloo (CPU tests / single-GPU rehearsal) moves CUDA tensors through host memory; RCCL takes
    them as they are."""
    return t.cpu() if (t.is_cuda and dist.get_backend() == "gloo") else t


def broadcast_weights(*weight_sets, src=0):
    """Overwrite every rank's blobs with rank `src`'s (layout is shape-derived, hence identical)."""
    for ws in weight_sets:
        buf = _stage(ws.blob.data)
        dist.broadcast(buf, src=src)
        if buf is not ws.blob.data:
            ws.blob.data.copy_(buf)
        ws.rebind()


def gather_coordinates(xyz_list, world_size):
    """All-gather a rank's coordinate tensors.  Ranks may hold different numbers of atoms, so each
    tensor list is flattened to one buffer, padded to the largest rank's size and trimmed after.
    Returns a list (per rank) of flat fp32 tensors."""
    if len(xyz_list):
        flat = torch.cat([x.reshape(-1) for x in xyz_list])
    else:       # a rank without units still takes part; RCCL wants its (empty) buffer on the device
        flat = torch.zeros(0, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    flat = _stage(flat)
    n = torch.tensor([flat.numel()], dtype=torch.int64, device=flat.device)
    sizes = [torch.zeros_like(n) for _ in range(world_size)]
    dist.all_gather(sizes, n)
    sizes = [int(s) for s in sizes]
    cap = max(sizes)
    buf = torch.zeros(cap, dtype=flat.dtype, device=flat.device)
    buf[:flat.numel()] = flat
    out = [torch.empty_like(buf) for _ in range(world_size)]
    dist.all_gather(out, buf)
    return [o[:s] for o, s in zip(out, sizes)]
