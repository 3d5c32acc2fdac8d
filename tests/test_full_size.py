"""BASELINE.json's configurations at their full per-GPU sizes (GPU, through the C ABI).

The oracle cannot run these sizes in seconds, so they are checked through properties that hold at any
size: replaying a job reproduces it bit for bit; a unit's result does not depend on which other units
share its job (the fact the multi-GPU sharding of SURVEY.md §8e rests on), checked bit for bit between
the whole job and the shards `codlad_amd.parallel.shard_units` deals; code indices are in range and are
fixed points of the lookup; and a few units of each job are sampled against the CPU oracle.
"""
import numpy as np
import pytest
import torch

from codlad_amd import parallel, synth
from oracle import denoiser as oden
from oracle import sampler as osam
from tests import pipeline

pytestmark = pytest.mark.gpu

PED_LENGTHS = list(synth.PED_LENGTHS)


def check_decode_against_oracle(cfg, u, x0, idx, xyz, ic):
    """The decoder tail of unit u against the CPU oracle ON THE HIP PATH'S OWN LATENT (so a code flip at a Voronoi
    boundary cannot enter): code indices bit-exact; internal coordinates within 2e-6 of an fp64 evaluation of the decoder
    (the fp32 oracle itself sits at 1.4e-7 .. 1.8e-7, the kernels at 1.8e-7 .. 2.4e-7: tools/decode_precision_probe.py);
    Cartesian coordinates RMSD <= 1e-4 A against the fp32 oracle - or, where ic -> xyz is ill-conditioned (untrained
    K3 / K4 heads give near-collinear placement triplets at L = 155 .. 505: the reference's OWN fp32 result is then
    2e-5 .. 1e-2 A from the fp64 one), no farther from the fp64 coordinates than three times the fp32 oracle is."""
    from oracle import vae_decode as odec
    p, f, _m = cfg.units[u]
    prot = cfg.proteins[p]
    L = prot["n_cg"]
    batch = synth.make_batch(prot, frame_ids=[f])
    angle = cfg.vae_type != "N6"
    vsd = synth.vqvae_state_dict(cfg.vae_type, cfg.dataname, cfg.vae_seed)
    mean, std = synth.norm_stats(cfg.dataname, cfg.vae_type)
    lat = odec.denormalise(x0.cpu()[None], mean, std)
    ridx, ic32 = odec.latent_decode(vsd, lat, batch, angle=angle)
    og = batch["OG_CG_nxyz"].reshape(-1, L + 2, 4)
    x32 = odec.ic_to_xyz(og, ic32.reshape(-1, L, 13, 3), prot["info"])[0]
    assert torch.equal(idx.cpu(), ridx.reshape(-1)), (u, L)
    vsd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in vsd.items()}
    b64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()}
    _i, ic64 = odec.latent_decode(vsd64, lat.double(), b64, angle=angle)
    x64 = odec.ic_to_xyz(og.double(), ic64.reshape(-1, L, 13, 3), prot["info"])[0]
    ic_err = float((ic.cpu().double() - ic64).abs().max() / ic64.abs().max())
    assert ic_err < 2e-6, (u, L, ic_err)
    rmsd = lambda a, b: float(((a.double() - b.double()) ** 2).sum(-1).mean().sqrt())  # noqa: E731
    got = xyz.cpu()
    assert rmsd(got, x32) < 1e-4 or rmsd(got, x64) < 3.0 * rmsd(x32, x64) + 2e-5, (u, L, rmsd(got, x32), rmsd(got, x64),
                                                                                   rmsd(x32, x64))


def check_job(cfg, unit_ids, n_shards, shards_to_run, oracle_units=(), decode_oracle_units=()):
    whole = cfg.run_units(unit_ids)
    again = cfg.run_units(unit_ids)
    for u in unit_ids:                               # deterministic replay
        assert pipeline.same(whole[u], again[u]), u
    costs = [parallel.unit_cost(cfg.lengths[cfg.units[u][0]]) for u in unit_ids]
    shards = parallel.shard_units(costs, n_shards)
    assert sorted(i for s in shards for i in s) == list(range(len(unit_ids)))
    for r in shards_to_run:                          # sharding invariance, bit for bit
        ids = [unit_ids[i] for i in shards[r]]
        part = cfg.run_units(ids)
        for u in ids:
            assert pipeline.same(whole[u], part[u]), (r, u)
    cb = cfg.dec.weights.codebook
    for u in unit_ids[:: max(1, len(unit_ids) // 50)]:
        x0, idx, xyz, _ic = whole[u]
        assert bool(torch.isfinite(x0).all()) and bool(torch.isfinite(xyz).all())
        assert int(idx.min()) >= 0 and int(idx.max()) < cb.shape[0]
        idx2, zq2, _ = cfg.dec.vq(cb[idx], normalised=False)      # codes are fixed points of the lookup
        assert torch.equal(idx2, idx) and torch.equal(zq2, cb[idx])
    for u in oracle_units:                           # a unit against the CPU oracle, same noise
        p, f, _m = cfg.units[u]
        prot = cfg.proteins[p]
        batch = synth.make_batch(prot, frame_ids=[f])
        cg_z, cg_xyz, mask = oden.batch_to_dense(batch)
        x_T, eps = cfg.unit_noise(u)
        sd = synth.denoiser_state_dict(pipeline.WEIGHT_SEED)
        ref = osam.p_sample_loop(sd, cfg.T, x_T.cpu()[None], eps.cpu()[:, None], cg_xyz, cg_z, mask,
                                 hoist_features=True)
        got = whole[u][0].cpu()
        err = float((got - ref[0]).abs().max() / ref[0].abs().max())
        assert err < 1e-4, (u, err)
    for u in list(oracle_units) + list(decode_oracle_units):     # VQ + IC decoder + ic_to_xyz of a unit against the oracle
        check_decode_against_oracle(cfg, u, *whole[u])
    return whole


def test_cfg2_ped_n6_full_size():
    """cfg 2: 4 PED-sized proteins x 10 frames x num_ensemble 10 = 400 structures, 100 steps, N6."""
    cfg = pipeline.Config("cfg2", PED_LENGTHS, n_frames=10, n_ensemble=10, vae_type="N6", dataname="PED")
    units = list(range(len(cfg.units)))
    assert len(units) == 400
    whole = check_job(cfg, units, n_shards=2, shards_to_run=[0, 1], oracle_units=[0])
    # ensemble members of one frame differ (different noise) but share one structure
    assert not torch.equal(whole[0][0], whole[1][0])
    # the one-wave-per-SIMD edge update (round 4) and a single stream instead of two half-jobs: the same bits at full size
    from codlad_amd import _lib
    _lib.set_option(_lib.OPT_EDGE_UPD_VARIANT, 1)
    try:
        other = cfg.run_units(units)
    finally:
        _lib.set_option(_lib.OPT_EDGE_UPD_VARIANT, 0)
    for u in units:
        assert pipeline.same(whole[u], other[u]), u


def test_cfg5_recon_decoder_only_full_size():
    """cfg 5 (--experiment recon): cfg-2 geometry, VQ + IC decoder + ic_to_xyz only, on latents that stand in
    for the (out-of-scope) e3nn encoder's; replay, sharding invariance, and two units against the CPU oracle."""
    from oracle import vae_decode as odec
    cfg = pipeline.Config("cfg5", PED_LENGTHS, n_frames=10, n_ensemble=10, vae_type="N6", dataname="PED")
    units = list(range(400))
    whole = cfg.run_units(units, decode_only=True)
    again = cfg.run_units(units, decode_only=True)
    shards = parallel.shard_units([parallel.unit_cost(cfg.lengths[cfg.units[u][0]]) for u in units], 2)
    part = cfg.run_units([units[i] for i in shards[1]], decode_only=True)
    for u in units:
        assert pipeline.same(whole[u], again[u]), u
    for i in shards[1]:
        assert pipeline.same(whole[units[i]], part[units[i]]), i
    vsd = synth.vqvae_state_dict("N6", "PED", pipeline.VAE_SEED)
    mean, std = synth.norm_stats("PED", "N6")
    for u in (0, 399):
        p, f, _m = cfg.units[u]
        prot = cfg.proteins[p]
        L = prot["n_cg"]
        batch = synth.make_batch(prot, frame_ids=[f])
        lat = odec.denormalise(cfg.unit_latent(u).cpu()[None], mean, std)
        idx, ic = odec.latent_decode(vsd, lat, batch)
        xyz = odec.ic_to_xyz(batch["OG_CG_nxyz"].reshape(-1, L + 2, 4), ic.reshape(-1, L, 13, 3), prot["info"])
        assert torch.equal(whole[u][1].cpu(), idx.reshape(-1))                      # code indices bit-exact
        rmsd = float(((whole[u][2].cpu() - xyz[0]) ** 2).sum(-1).mean().sqrt())
        assert rmsd < 1e-4, (u, rmsd)


def test_cfg3_pdb_k3_full_size():
    """cfg 3: 64 proteins, one frame each, angle decoder (K3); lengths = the first 64 of the reference's Atlas test
    list clipped to 50..400 (PDB lengths do not ship); all 64 on one GPU and two of the eight
    per-GPU shards."""
    c = synth.baseline_config("cfg3")
    lengths = c["lengths"]
    assert len(lengths) == 64 and min(lengths) >= 50 and max(lengths) <= 400
    cfg = pipeline.Config("cfg3", lengths, n_frames=c["n_frames"], n_ensemble=c["n_ensemble"], vae_type=c["vae_type"],
                          dataname=c["dataname"])
    # sampler: the shortest protein against the oracle's 100-step loop; decoder tail (K3 angle decoder): the shortest,
    # the longest (L = 400) and a median one against the oracle on the HIP latents
    order = np.argsort(lengths)
    check_job(cfg, list(range(64)), n_shards=8, shards_to_run=[0, 5], oracle_units=[int(order[0])],
              decode_oracle_units=[int(order[-1]), int(order[32])])
    assert lengths[int(order[-1])] == 400


def test_cfg4_atlas_k4_one_gpu_share():
    """cfg 4: the reference's Atlas test list (datasets/protein/Atlas/new_atlas_test.csv `seqlen`: 70 proteins,
    39..505 residues, median 155; fixture tests/golden/atlas_test_seqlen.json) x 4 frames x num_ensemble 32 =
    8 960 structures over 8 GPUs: the 1 120 structures LPT deals to rank 0, then that share split again."""
    c = synth.baseline_config("cfg4")
    lengths = c["lengths"]
    assert len(lengths) == 70 and min(lengths) == 39 and max(lengths) == 505 and sorted(lengths)[35] == 155
    cfg = pipeline.Config("cfg4", lengths, n_frames=c["n_frames"], n_ensemble=c["n_ensemble"], vae_type=c["vae_type"],
                          dataname=c["dataname"])
    assert len(cfg.units) == 8960
    costs = [parallel.unit_cost(lengths[p]) for p, _f, _m in cfg.units]
    shards = parallel.shard_units(costs, 8)
    loads = [sum(costs[i] for i in s) for s in shards]
    assert max(loads) / min(loads) < 1.01            # LPT balance across the 8 ranks
    # decoder tail (K4 angle decoder) of this share's longest and shortest units against the oracle on the HIP latents,
    # and the 100-step sampler of its shortest
    by_len = sorted(shards[0], key=lambda u: (lengths[cfg.units[u][0]], u))
    assert lengths[cfg.units[by_len[-1]][0]] == 505
    check_job(cfg, shards[0], n_shards=2, shards_to_run=[1], oracle_units=[by_len[0]],
              decode_oracle_units=[by_len[-1], by_len[len(by_len) // 2]])
