"""Multi-process path on CPU: world_size 2, gloo backend, 127.0.0.1 rendezvous.  Covers what the
N > 1 bench/driver does around the (GPU-only) compute: unit sharding, weight-blob broadcast from
rank 0, and the final ragged all-gather of coordinates."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from codlad_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeWeights:
    """Stands in for weights.DenoiserWeights on CPU: one flat blob + rebind()."""

    def __init__(self, n, fill):
        self.blob = type("B", (), {})()
        self.blob.data = torch.full((n,), float(fill))
        self.rebinds = 0

    def rebind(self):
        self.rebinds += 1


def _unit_result(u, L):
    g = torch.Generator().manual_seed(1000 + u)        # per-unit seed: result independent of the sharding
    return torch.randn(7 * L, 3, generator=g)


def _worker(rank, world, port, lengths, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        den, dec = _FakeWeights(1000, rank + 1), _FakeWeights(10, 10 * (rank + 1))
        parallel.broadcast_weights(den, dec)
        assert float(den.blob.data.min()) == 1.0 == float(den.blob.data.max())      # rank 0's values everywhere
        assert float(dec.blob.data[0]) == 10.0 and den.rebinds == dec.rebinds == 1
        shards = parallel.shard_units([parallel.unit_cost(L) for L in lengths], world)
        mine = [_unit_result(u, lengths[u]) for u in shards[rank]]
        gathered = parallel.gather_coordinates(mine, world)
        assert len(gathered) == world
        # reassemble in unit order on every rank
        out = {}
        for r in range(world):
            flat, o = gathered[r], 0
            for u in shards[r]:
                n = 7 * lengths[u] * 3
                out[u] = flat[o:o + n].view(-1, 3)
                o += n
            assert o == flat.numel()
        ok = all(torch.equal(out[u], _unit_result(u, lengths[u])) for u in range(len(lengths)))
        q.put((rank, ok, [len(s) for s in shards]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("world", [2, 8])
def test_broadcast_shard_gather(world):
    """world 8: the rank count of the node the scaling bench runs on (more ranks than some shards have units: 9 units on 8
    ranks leaves seven ranks one unit each)."""
    lengths = [46, 87, 92, 129, 505, 39, 155, 60, 87]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, lengths, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(200)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert all(ok for _, ok, _ in res)
    assert all(r[2] == res[0][2] for r in res) and sum(res[0][2]) == len(lengths)


def _exps(w):
    return [[getattr(w.struct.enc_h[l], n) for n in ("e1", "e2", "e3", "e11", "e12", "e13", "e_in", "e_out")] +
            [getattr(w.struct.dec_h[l], n) for n in ("e1", "e2", "e3", "e_in", "e_out")] for l in range(3)]


def _real_weights_worker(rank, world, port, mode, q):
    """The real weights.DenoiserWeights / DecoderWeights (host blobs) through the real broadcast: what rank 1 holds
    BEFORE must not leak into the struct the kernels read AFTER."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from codlad_amd import synth, weights
    from tests import cases
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        env = cases.envelope_state_dict("small_first_1e-2")
        default = synth.denoiser_state_dict(cases.WEIGHT_SEED)
        if mode == "r1_envelope":
            den = weights.DenoiserWeights(default if rank == 0 else env, "cpu")
            dec = weights.DecoderWeights(synth.vqvae_state_dict("N6", "PED", 4321 + rank), "cpu",
                                         *synth.norm_stats("PED", "N6"))
        elif mode == "r0_envelope":
            den = weights.DenoiserWeights(env if rank == 0 else default, "cpu", precision="f16x4" if rank == 0 else "f16x3")
            dec = weights.DecoderWeights(synth.vqvae_state_dict("K4", "Atlas", 4321 + rank), "cpu",
                                         *synth.norm_stats("Atlas", "K4"))
        else:       # rank 1 holds nothing, and layouts of OTHER model flags (self-conditioned flow model, N6 decoder)
            if rank == 0:
                den = weights.DenoiserWeights(env, "cpu")
                dec = weights.DecoderWeights(synth.vqvae_state_dict("K3", "PDB", 4321), "cpu", *synth.norm_stats("PDB", "K3"))
            else:
                den = weights.DenoiserWeights.empty("cpu", self_condition=True, out_dim=3, precision="f32")
                dec = weights.DecoderWeights.empty("cpu")
        before = _exps(den)
        sums = parallel.broadcast_weights(den, dec, announce=False)
        ref = weights.DenoiserWeights(env if mode != "r1_envelope" else default, "cpu")
        ok = (_exps(den) == _exps(ref) and den.exponents == ref.exponents
              and den.precision == ("f16x4" if mode == "r0_envelope" else "f16x3")
              and den.self_condition is False and den.out_dim == 6
              and torch.equal(den.blob.data[weights.META_FLOATS:], ref.blob.data[weights.META_FLOATS:])
              and dec.angle == (mode != "r1_envelope") and den.generation >= 1
              and torch.equal(dec.mean, synth.norm_stats({"r1_envelope": "PED", "r0_envelope": "Atlas", "r1_empty": "PDB"}[mode],
                                                         {"r1_envelope": "N6", "r0_envelope": "K4", "r1_empty": "K3"}[mode])[0]))
        q.put((rank, ok, before != _exps(den), sums))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("mode", ["r1_envelope", "r0_envelope", "r1_empty"])
def test_broadcast_carries_block_exponents_and_flags(mode):
    """Round-2 defect: `rebind()` re-filled the struct from the receiving rank's OWN exponents.  Rank 1 now starts
    from weights with exponents 7 / -4 / 6 / -3 against rank 0's zeros (and the other way round, and from an empty
    blob of another layout): blocks, exponents, contraction mode, model flags and norm statistics must be rank 0's."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_real_weights_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(280)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert all(ok for _, ok, _, _ in res), res
    assert res[0][2] is False and res[1][2] is True          # rank 1's exponents changed, rank 0's did not
    assert res[0][3] == res[1][3] and len(res[0][3]) == 2    # equal checksums


def test_shard_units_balance_and_coverage():
    lengths = [39, 46, 87, 92, 129, 155, 155, 200, 505, 505, 60, 61, 300]
    costs = [parallel.unit_cost(L) for L in lengths]
    for world in (1, 2, 4, 8):
        shards = parallel.shard_units(costs, world)
        assert sorted(u for s in shards for u in s) == list(range(len(lengths)))
        loads = [sum(costs[u] for u in s) for s in shards]
        assert max(loads) - min(loads) <= max(costs)          # LPT bound
    assert parallel.shard_units([], 4) == [[], [], [], []]
    assert parallel.unit_cost(46) == 46 * 46 and parallel.unit_cost(129) == 129 * 64
