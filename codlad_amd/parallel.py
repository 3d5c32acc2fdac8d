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


def _comm_device():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def broadcast_weights(*weight_sets, src=0, verify=True, announce=True):
    """Make every rank's weight sets (weights.DenoiserWeights / DecoderWeights) equal to rank `src`'s.

    Per set: (1) the 64-float header - contraction mode, model flags, blob size and the block exponents of the
    split-fp16 copies - goes first; a receiving rank takes every host-side field from it and re-derives its blob
    layout if the model flags differ, so it may have started from ANY weights or from none (`*.empty`);
    (2) the blob (which carries the same header in its first 64 floats) follows as one buffer; (3) `rebind()` reads
    the header back from the blob and re-derives the pointers.  verify: the ranks all-gather a 64-bit checksum of
    each blob and raise if one differs; rank 0 prints one line with the rank count and the backend (the record that
    RCCL saw N ranks).  Returns the checksums."""
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = _comm_device()
    for ws in weight_sets:
        if rank == src and hasattr(ws, "sync_meta"):
            ws.sync_meta()
        if hasattr(ws, "header"):
            hdr = (ws.header() if rank == src else torch.zeros_like(ws.header())).to(dev)
            dist.broadcast(hdr, src=src)
            if rank != src:
                ws.adopt_header(hdr.cpu())
        buf = _stage(ws.blob.data)
        dist.broadcast(buf, src=src)
        if buf is not ws.blob.data:
            ws.blob.data.copy_(buf)
        ws.rebind()
    sums = [ws.checksum() for ws in weight_sets if hasattr(ws, "checksum")]
    if verify and sums:
        sums = verify_checksums(sums, what="weights broadcast from rank %d" % src, announce=announce)
    return sums


def verify_checksums(sums, what="weights", announce=True):
    """All-gather 64-bit checksums and raise unless every rank holds the same ones."""
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = torch.tensor(sums, dtype=torch.int64, device=_comm_device())
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    every = [e.cpu().tolist() for e in every]
    bad = [r for r in range(world) if every[r] != every[0]]
    if bad:
        raise RuntimeError(f"{what}: ranks {bad} hold different bytes than rank 0 ({every})")
    if announce and rank == 0:
        print(f"codlad_amd: {what}: ranks={world} backend={dist.get_backend()} "
              f"checksums={[hex(c & 0xFFFFFFFFFFFFFFFF) for c in every[0]]} equal on every rank", flush=True)
    return every[0]


def broadcast_module_state(*modules, src=0):
    """Every floating-point parameter and buffer of the nn.Modules as ONE flat buffer from rank `src` (what the
    drop-in CLI does instead of letting every rank read the checkpoint files: only rank 0 touches the disk)."""
    tensors = [t for m in modules for t in list(m.parameters()) + list(m.buffers()) if t.is_floating_point()]
    if not tensors:
        return
    flat = torch.cat([t.detach().reshape(-1).float() for t in tensors])
    buf = _stage(flat)
    dist.broadcast(buf, src=src)
    flat = buf.to(flat.device)
    o = 0
    with torch.no_grad():
        for t in tensors:
            n = t.numel()
            t.copy_(flat[o:o + n].view(t.shape))       # in place: bumps the tensor's version, engines repack lazily
            o += n


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
