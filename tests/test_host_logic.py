"""CPU-only checks of the host side: the C-ABI library loads and exports every declared symbol,
the weight packer matches the library's own packer, schedule tables match the reference goldens,
and the table builders (CSR, info tables, job node tables) are right.  No GPU compute here."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from codlad_amd import _lib, synth
from codlad_amd.diffusion_and_flow.schedule import Tables, named_betas, space_timesteps
from codlad_amd.weights import denoiser_tensors, decoder_tensors, pack_block
from tests import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "codlad_hip.h")).read()
    declared = set(re.findall(r"\b(codlad_[a-z0-9_]+)\s*\(", header))
    declared -= {"codlad_pack_block_host"} - {"codlad_pack_block_host"}
    assert declared == set(_lib.exported_symbols())
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.codlad_abi_version() == _lib.ABI_VERSION
    header = open(os.path.join(os.path.dirname(__file__), "..", "include", "codlad_hip.h")).read()
    assert f"#define CODLAD_ABI_VERSION {_lib.ABI_VERSION}\n" in header


def test_struct_layouts_match_header():
    # the ctypes mirrors must have exactly the C structs' sizes and field offsets
    out = (C.c_int * 5)()
    _lib.lib().codlad_struct_sizes(out)
    assert C.sizeof(_lib.DenoiserWeights) == out[0]
    assert C.sizeof(_lib.DecoderWeights) == out[1]
    assert C.sizeof(_lib.Workspace) == out[2]
    assert _lib.DenoiserWeights.precision.offset == out[3]
    assert _lib.DenoiserWeights.enc_h.offset == out[4]
    assert C.sizeof(_lib.EncLayer) == 26 * 8 and C.sizeof(_lib.DecLayer) == 19 * 8
    assert C.sizeof(_lib.EncLayerH) == 26 * 8 + 8 * 4 and C.sizeof(_lib.DecLayerH) == 19 * 8 + 5 * 4 + 4


def test_argument_errors_are_reported_not_crashed():
    lib = _lib.lib()
    rc = lib.codlad_vq_lookup(None, 10, None, None, None, 4096, None, None, None, None)
    assert rc < 0 and b"null pointer" in lib.codlad_last_error()
    rc = lib.codlad_ic_to_xyz(None, None, None, None, 1, 1, 1, None, None)
    assert rc < 0


def test_pack_block_matches_library_packer():
    g = torch.Generator().manual_seed(3)
    W = torch.randn(128, 384, generator=g)
    blk = W[:, 128:256]
    mine = pack_block(blk, 2.0)
    src = np.ascontiguousarray(W.numpy())
    dst = np.empty(16384, dtype=np.float32)
    _lib.lib().codlad_pack_block_host(src.ctypes.data_as(C.c_void_p).value + 128 * 4, 384, C.c_float(2.0),
                                      dst.ctypes.data_as(C.c_void_p))
    assert np.array_equal(mine.numpy(), dst)
    # every source element appears exactly once
    assert np.array_equal(np.sort(dst), np.sort(2.0 * blk.numpy().reshape(-1)))


def test_split_f16_block_packing():
    """hi + lo fp16 halves reproduce the fp32 weight to <= 2^-22 relative (or the fp16 subnormal
    spacing), in the [k-step][out block][split][lane][8] order the kernels read."""
    from codlad_amd.weights import pack_block_h
    g = torch.Generator().manual_seed(4)
    W = torch.randn(128, 128, generator=g) * 0.13
    W[5, 77] = 3.0
    pk = pack_block_h(W).view(torch.float16).view(8, 4, 2, 64, 8)
    ks, bo, lane, j = 5, 3, 41, 6
    row = 32 * bo + (lane & 31)
    col = 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3)
    rec = pk[ks, bo, 0, lane, j].double() + pk[ks, bo, 1, lane, j].double()
    assert abs(float(rec) - float(W[row, col])) <= max(2.0 ** -22 * abs(float(W[row, col])), 2.0 ** -25)
    # whole block: every element exactly once, error bound everywhere
    rec_all = (pk[:, :, 0].double() + pk[:, :, 1].double()).reshape(-1)
    assert torch.allclose(rec_all.sort().values, W.double().reshape(-1).sort().values, rtol=2.0 ** -21, atol=2.0 ** -24)
    two = pack_block_h(W, 2.0).view(torch.float16).view(8, 4, 2, 64, 8)
    rec2 = (two[:, :, 0].double() + two[:, :, 1].double()).reshape(-1)
    assert torch.allclose(rec2, 2 * rec_all, rtol=2.0 ** -21, atol=2.0 ** -24)   # decoder blocks carry the factor 2


@pytest.mark.parametrize("T", ["10", "100", "250"])
def test_schedule_tables_match_reference(T):
    gold = np.load(cases.npz_path(f"g1_schedule_{T}"))
    tb = Tables(named_betas("linear", 1000), space_timesteps(1000, T))
    assert tb.timestep_map == gold["timestep_map"].tolist()
    for k in ("betas", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1",
              "posterior_mean_coef2", "posterior_log_variance_clipped"):
        np.testing.assert_array_equal(getattr(tb, k), gold[k], err_msg=k)
    c = tb.step_coefficients()
    np.testing.assert_array_equal(c[:, 5], gold["log_betas"].astype(np.float32))
    assert c[0, 6] == 0 and (c[1:, 6] == 1).all()


def test_space_timesteps_variants():
    assert space_timesteps(300, [10, 15, 20]) == space_timesteps(300, "10,15,20")
    assert len(space_timesteps(1000, "ddim50")) == 50
    with pytest.raises(ValueError):
        space_timesteps(10, "20")


def test_denoiser_blob_covers_every_checkpoint_tensor():
    sd = synth.denoiser_state_dict(cases.WEIGHT_SEED)
    t = denoiser_tensors(sd)
    # 96 packed 128x128 blocks + TS tables + plain tensors; total parameter mass is preserved
    n_blocks = sum(1 for v in t.values() if v.numel() == 16384 and v.dim() == 1)
    assert n_blocks == 3 * 18 + 3 * 13
    mod = {("module." + k): v for k, v in sd.items()}
    t2 = denoiser_tensors(mod)
    assert all(torch.equal(t[k], t2[k]) for k in t)


def test_decoder_tensors_both_variants_and_codebook_layouts():
    for vt, dn in (("N6", "PED"), ("K3", "PDB")):
        for layout in ("lucidrains", "inrepo"):
            vsd = synth.vqvae_state_dict(vt, dn, cases.VAE_SEED, quantizer_layout=layout)
            t, angle = decoder_tensors(vsd)
            assert angle == (vt != "N6")
            assert t["codebook"].shape == (4096, 3)
            assert t["tor1_w0"].shape == ((50, 50) if angle else (40, 40))


def test_csr_from_pairs_keeps_reference_scatter_order():
    from codlad_amd.engine import Decoder
    pairs = torch.tensor([[0, 1], [0, 3], [1, 2], [2, 3]])
    ptr, src = Decoder.csr_from_pairs(pairs, 5)
    assert ptr.tolist() == [0, 2, 4, 6, 8, 8]
    # receiver 1: first the forward pair (1<-2), then the flipped one (1<-0)
    assert src.tolist() == [1, 3, 2, 0, 3, 1, 0, 2]


def test_info_tables_invert_the_reference_gather():
    from codlad_amd.engine import info_tables
    prot = synth.make_protein(30, 5)
    permute, atom_idx, orders = prot["info"]
    o, s2o, n_atoms = info_tables(prot["info"], 30, "cpu")
    slots = torch.arange(30 * 14)
    ref = slots[atom_idx][permute]          # what utils_ic.py:267 selects
    back = torch.full((n_atoms,), -1, dtype=torch.long)
    valid = s2o >= 0
    back[s2o[valid].long()] = slots[valid]
    assert torch.equal(back, ref)
    bad = (permute, atom_idx, orders.clone())
    bad[2][0, 0, 0] = 9                      # CB built from an atom that does not exist yet
    with pytest.raises(AssertionError):
        info_tables(bad, 30, "cpu")


def test_product_code_never_imports_the_oracle():
    for dirpath, _dirs, files in os.walk(os.path.join(ROOT, "codlad_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f


def test_graft_entry_build_runs_on_cpu():
    """What the driver runs in the GPU-less container every round: compile (a no-op when up to date),
    dlopen, resolve every symbol, check the ABI version, import the oracle."""
    import __graft_entry__
    __graft_entry__.build()


def test_shard_units_properties():
    """LPT sharding (codlad_amd/parallel.py) for arbitrary unit mixes: every unit exactly once, shards
    sorted, deterministic, and the classic LPT bound: max load <= mean load + the largest unit."""
    from hypothesis import given, settings, strategies as st
    from codlad_amd import parallel

    @settings(max_examples=200, deadline=None)
    @given(st.lists(st.integers(min_value=1, max_value=505), min_size=0, max_size=300),
           st.integers(min_value=1, max_value=8))
    def check(lengths, world):
        costs = [parallel.unit_cost(L) for L in lengths]
        shards = parallel.shard_units(costs, world)
        assert len(shards) == world
        assert sorted(u for s in shards for u in s) == list(range(len(costs)))
        assert all(s == sorted(s) for s in shards)
        assert shards == parallel.shard_units(costs, world)
        if costs:
            loads = [sum(costs[u] for u in s) for s in shards]
            assert max(loads) <= sum(costs) / world + max(costs)

    check()


def test_edge_block_layout_helper():
    """engine.edge_rows undoes the edge-block layout documented in include/codlad_hip.h
    (feature f of edge e at 4096*(e/32) + 128*(f/4) + 4*(e%32) + f%4)."""
    from codlad_amd.engine import edge_rows
    n = 3
    rows = torch.arange(n * 64 * 128, dtype=torch.float32).view(n, 64, 128)
    blocks = torch.empty(n, 2, 32, 32, 4)
    for f in range(128):
        for half in range(2):
            blocks[:, half, f // 4, :, f % 4] = rows[:, 32 * half:32 * half + 32, f]
    assert torch.equal(edge_rows(blocks), rows)
    flat = blocks.view(n, -1)
    for e, f in ((13, 77), (45, 6)):
        assert float(flat[1, 4096 * (e // 32) + 128 * (f // 4) + 4 * (e % 32) + f % 4]) == float(rows[1, e, f])
    assert tuple(edge_rows(torch.zeros(2, 5, 2, 32, 32, 4)).shape) == (2, 5, 64, 128)
    # the stored form of the split-fp16 modes (csrc/common.h "pre-split edge state"): build it from values the way
    # tile_presplit does and read it back
    vals = torch.randn(3, 64, 128) * 4
    hi = vals.to(torch.float16)
    lo = (vals - hi.float()).to(torch.float16)
    stored = torch.zeros(3, 2, 32, 32, 8, dtype=torch.float16)
    for b in range(4):
        for s_ in range(2):
            for h in range(2):
                f0 = 32 * b + 16 * s_ + 4 * h
                feats = list(range(f0, f0 + 4)) + list(range(f0 + 8, f0 + 12))
                for half in range(2):
                    e = slice(32 * half, 32 * half + 32)
                    stored[:, half, 8 * b + 4 * s_ + h] = hi[:, e][:, :, feats]
                    stored[:, half, 8 * b + 4 * s_ + 2 + h] = lo[:, e][:, :, feats]
    back = edge_rows(stored.view(torch.float32).view(3, 2, 32, 32, 4), split=True)
    assert torch.equal(back, hi.float() + lo.float())
    assert float((back - vals).abs().max()) < 4 * 2.0 ** -22 * 16


def test_split_pack_refuses_weights_outside_fp16_range():
    """hi = f16(w) would be inf: the split-fp16 packer raises; a weight matrix gets a power-of-two block exponent
    that brings its rms to [1/4, 1/2) and its largest element inside the fp16 range - or is refused when one outlier
    would cost the rest of the matrix its precision; the non-strict builder (fp32-MFMA mode) counts instead."""
    from codlad_amd.weights import block_exponent, denoiser_tensors_h, pack_block_h
    W = torch.zeros(128, 128)
    W[2, 3] = 7.0e4
    with pytest.raises(ValueError, match="fp16 range"):
        pack_block_h(W)
    pack_block_h(W, 0.5)                                   # 3.5e4 fits
    g = torch.Generator().manual_seed(8)
    for rms in (3e-4, 0.02, 0.09, 0.7, 1.5, 40.0):
        M = torch.randn(128, 384, generator=g) * rms
        e = block_exponent(M)
        assert (e == 0) == (0.07 < rms < 0.9), (rms, e)           # ordinary scale: the plain split
        assert 2.0 ** -4.2 <= float(M.pow(2).mean().sqrt()) * 2.0 ** e < 1.05, (rms, e)
    M = torch.randn(128, 128, generator=g) * 1e-3
    e_plain = block_exponent(M)
    M[0, 0] = 1.0e4                                        # an outlier the block can absorb: 2^e max <= 2^15 limits e
    assert block_exponent(M) == 1 and e_plain == 7
    M[0, 0] = 1.0e9
    with pytest.raises(ValueError, match="fp16 range"):
        block_exponent(M)
    sd = synth.denoiser_state_dict(cases.WEIGHT_SEED)
    t, exps, bad = denoiser_tensors_h(sd)
    assert bad == 0 and set(exps) == {f"{k}{l}" for k in ("enc", "dec") for l in range(3)}
    assert all(v == 0 for d in exps.values() for v in d.values())      # the default synthetic weights are ordinary
    exps_small = denoiser_tensors_h(cases.envelope_state_dict("small_first_1e-2"))[1]
    assert exps_small["enc0"]["e1"] > 0 > exps_small["enc0"]["e2"]
    # biases are stored with the power of two their accumulator carries
    sd_s = cases.envelope_state_dict("small_first_1e-2")
    t_s, ex_s, _ = denoiser_tensors_h(sd_s)
    e = ex_s["enc1"]
    assert torch.equal(t_s["h.enc1.b12"], sd_s["encoder_layers.1.W12.bias"] * 2.0 ** (e["e11"] + e["e12"]))
    # ... and e = 0 everywhere is the plain split
    t0, exps0, _ = denoiser_tensors_h(sd, use_exponents=False)
    assert all(v == 0 for d in exps0.values() for v in d.values())
    assert torch.equal(t0["h.enc1.W2"], pack_block_h(sd["encoder_layers.1.W2.weight"]))
    sd["decoder_layers.2.dense.W_out.weight"] = sd["decoder_layers.2.dense.W_out.weight"].clone()
    sd["decoder_layers.2.dense.W_out.weight"][0, 300] = float("inf")
    with pytest.raises(ValueError, match="fp16 range"):
        denoiser_tensors_h(sd)
    assert denoiser_tensors_h(sd, strict=False)[2] == 1


def test_cli_batches_of_one_file_get_distinct_output_names():
    """A data file of more than 96 frames is cut into several batches (reference dataset_module.py:220-226);
    each must land in its own output file (round-1 bug: all chunks were saved under the file's name)."""
    import importlib.util
    import types
    spec = importlib.util.spec_from_file_location("codlad_cli", os.path.join(ROOT, "test.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    assert cli.chunk_plan(10) == [(0, 10)] and cli.chunk_plan(96) == [(0, 96)]
    assert cli.chunk_plan(200) == [(0, 96), (96, 192), (192, 200)]
    assert cli.output_name("a.pkl", 0, 1) == "a.pkl"
    args = types.SimpleNamespace(synthetic=True, data_type="PED", vae_type="N6", synthetic_frames=100)
    names, frames = [], 0
    for name, batch, info in cli.iter_batches(args):
        names.append(name)
        frames += int(batch["num_CGs"].shape[0])
    assert len(names) == len(set(names)) == 8 and frames == 400          # 4 proteins x (96 + 4) frames
    assert names[:2] == ["synthetic_L46_b00000", "synthetic_L46_b00001"]


def test_hot_edge_kernels_keep_their_register_budget():
    """The per-node edge kernels (88 % of a DDPM step) are written to fit the 256-register limit of two waves per SIMD
    with at most a handful of spilled registers outside the contraction loops.  hipcc's allocation for them is
    fragile - other kernels in the same translation unit were enough to push 20-30 registers into scratch inside the
    loops (-9 %) - so the build records what it got (codlad_amd/csrc/kernel_resources.json) and this holds it."""
    import json
    from codlad_amd import build
    build.build(force=False, verbose=False)
    if not os.path.exists(build.RESOURCES):
        build.build(force=True, verbose=False)
    with open(build.RESOURCES) as f:
        table = json.load(f)
    budget = {"_Z12msg_kernel_hILi8ELb0ELi3EEv8EdgeArgs": 32, "_Z12msg_kernel_hILi8ELb1ELi3EEv8EdgeArgs": 32,
              "_Z12upd_kernel_hILi8ELb0ELi3EEv8EdgeArgs": 32, "_Z12upd_kernel_hILi8ELb1ELi3EEv8EdgeArgs": 48}
    for name, scratch in budget.items():
        r = table[name]
        assert r["VGPRs"] <= 256 and r["Occupancy"] == 2, (name, r)
        assert r["ScratchSize"] <= scratch, (name, r)
    # the one-wave-per-SIMD edge update (round 4) has the whole 512-register file and must not touch scratch at all: a
    # scratch access counts in vmcnt and gets an s_waitcnt vmcnt(0), which waits for the tile prefetch in flight
    for name in ("_Z13upd1_kernel_hILb0ELi3EEv8EdgeArgs", "_Z13upd1_kernel_hILb1ELi3EEv8EdgeArgs",
                 "_Z13upd1_kernel_hILb0ELi4EEv8EdgeArgs", "_Z13upd1_kernel_hILb1ELi4EEv8EdgeArgs"):
        r = table[name]
        assert r["Occupancy"] == 1 and r["ScratchSize"] == 0, (name, r)
    # small jobs: the four-wave message kernels keep their two weight quarters in registers without scratch; the edge
    # update (three quarters in flight) may park a few registers of W11e's
    for name, r in table.items():
        if "msg_wide_kernel" in name:
            assert r["ScratchSize"] == 0 and r["Occupancy"] >= 2, (name, r)
        if "upd_wide_kernel" in name:
            assert r["ScratchSize"] <= 64 and r["Occupancy"] == 2, (name, r)
        # the four-wave node update owns the whole register file; its ring of weight quarters overflows into accumulator
        # registers, never into scratch
        if "node_kernel_q" in name:
            assert r["ScratchSize"] == 0 and r["Occupancy"] == 1, (name, r)


@pytest.mark.parametrize("name", list(cases.INFO_CASES))
def test_info_tables_match_reference_traj_to_info(name):
    """codlad_amd.utils.protein_module.info_from_residues against the tables the reference's own traj_to_info built
    (golden g13; mdtraj stood in for by the pandas table traj_to_info reads from it)."""
    from codlad_amd.utils.protein_module import info_from_residues
    n_cg, seed, phospho = cases.INFO_CASES[name]
    gold = np.load(cases.npz_path(f"g13_info_{name}"))
    z_full = synth.sequence(n_cg + 2, 2000 + seed, phospho=phospho)
    names = [synth.IDX2THR[int(z)] for z in z_full]
    (permute, atom_idx, orders), n = info_from_residues(names, [synth.PDB_ATOM_ORDER[nm] for nm in names])
    assert n == int(gold["n_cg"]) == n_cg + 2
    assert np.array_equal(permute.numpy(), gold["permute"]) and np.array_equal(atom_idx.numpy(), gold["atom_idx"])
    assert np.array_equal(orders.numpy(), gold["atom_orders"])
    with pytest.raises(ValueError, match="do not match the template"):
        bad = [list(synth.PDB_ATOM_ORDER[nm]) for nm in names]
        bad[3] = bad[3][:-1] if len(bad[3]) > 4 else bad[3] + ["XX"]
        info_from_residues(names, bad)


def test_pdb_writer_round_trips_through_the_reader(tmp_path):
    from codlad_amd.utils.protein_module import info_from_residues, read_pdb_topology, write_pdb
    z_full = synth.sequence(12, 5)
    names = [synth.IDX2THR[int(z)] for z in z_full]
    atoms = [synth.PDB_ATOM_ORDER[nm] for nm in names]
    n_atoms = sum(len(a) for a in atoms[1:-1])
    xyz = np.random.Generator(np.random.PCG64(1)).uniform(-50, 50, (3, n_atoms, 3)).astype(np.float32)
    path = os.path.join(tmp_path, "traj.pdb")
    write_pdb(path, xyz, names, atoms)
    r_names, r_atoms, r_xyz = read_pdb_topology(path)
    assert r_names == names[1:-1] and r_atoms == [list(a) for a in atoms[1:-1]]
    assert np.allclose(r_xyz, xyz[0], atol=5e-4)                       # 3 decimals in the PDB format
    text = open(path).read()
    assert text.count("MODEL ") == 3 and text.count("ENDMDL") == 3
    info, _ = info_from_residues(["GLY"] + r_names + ["GLY"], [["N", "CA", "C", "O"]] + r_atoms + [["N", "CA", "C", "O"]])
    assert info[0].numel() == n_atoms


def test_job_parts_deal_the_samples_and_share_the_edge_state():
    """engine.Job.parts (round 4: a large job runs as two half-jobs on two HIP streams): the samples are dealt alternately,
    every node of the job belongs to exactly one part, a part's tables are those of a job built from its samples alone, and
    the parts keep their edge state in disjoint slices of the parent's buffer (no second 32 KB per node)."""
    from codlad_amd.engine import Job, Structures
    lens = [7, 40, 33, 64, 12, 70, 5]
    prots = [synth.make_protein(L, 700 + i, n_frames=1) for i, L in enumerate(lens)]
    st = Structures([torch.from_numpy(p["xyz_full"])[0, 1:-1] for p in prots],
                    [torch.from_numpy(p["z_full"])[1:-1] for p in prots], "cpu")
    members = [0, 1, 2, 3, 4, 5, 6, 3, 1]
    job = Job(st, members, "cpu")
    for k in (2, 3):
        parts = job.parts(k)
        assert parts is job.parts(k) and len(parts) == k                       # built once
        seen = torch.zeros(job.n_nodes, dtype=torch.int32)
        start = 0
        for p, (sub, idx) in enumerate(parts):
            mine = members[p::k]
            assert sub.sample_struct == mine and sub.n_nodes == int(idx.numel()) == sum(lens[m] for m in mine)
            seen[idx] += 1
            alone = Job(st, mine, "cpu")
            assert torch.equal(sub.node_info, alone.node_info) and torch.equal(sub.tile_list, alone.tile_list)
            # the node indices are those of the part's samples in the parent, in order
            want = np.concatenate([np.arange(job.sample_off[m], job.sample_off[m + 1]) for m in range(p, len(members), k)])
            assert idx.tolist() == want.tolist()
            # source structure node of every part node = the parent's at the same place
            assert torch.equal(sub.node_info[:, 0], job.node_info[idx][:, 0])
            assert sub.hE.data_ptr() == job.hE[start:start + sub.n_nodes].data_ptr() and sub.hE.is_contiguous()
            start += sub.n_nodes
        assert start == job.n_nodes and bool((seen == 1).all())


def test_package_asks_for_device_side_kernel_arguments_unless_told_otherwise():
    """`import codlad_amd` sets HIP_FORCE_DEV_KERNARG=1 if the caller has not set it (the runtime reads it at its first
    call; 17 short launches per DDPM step of a small job each start with a read of the argument segment), and leaves a
    caller's value alone."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import os, codlad_amd; print(os.environ['HIP_FORCE_DEV_KERNARG'])"
    for given, want in ((None, "1"), ("0", "0")):
        env = {k: v for k, v in os.environ.items() if k != "HIP_FORCE_DEV_KERNARG"}
        if given is not None:
            env["HIP_FORCE_DEV_KERNARG"] = given
        out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, check=True)
        assert out.stdout.strip() == want
