"""Pins the CPU oracle to the reference: every oracle function against golden vectors that
tools/gen_golden.py produced by running the reference itself."""
import numpy as np
import pytest
import torch

from codlad_amd import synth
from oracle import denoiser, sampler, schedule, vae_decode
from tests import cases


def g(name):
    return np.load(cases.npz_path(name))


def rel_err(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture(scope="module")
def sd():
    return synth.denoiser_state_dict(cases.WEIGHT_SEED)


@pytest.mark.parametrize("T", ["10", "100", "250"])
def test_g1_schedule(T):
    gold = g(f"g1_schedule_{T}")
    mine = schedule.make_schedule(T)
    assert np.array_equal(mine["timestep_map"], gold["timestep_map"])
    for k in gold.files:
        if k != "timestep_map":
            np.testing.assert_array_equal(mine[k], gold[k], err_msg=k)  # float64, bit-exact


@pytest.mark.parametrize("name", list(cases.DENOISER_CASES))
def test_g2_forward(sd, name):
    L, B, seed = cases.DENOISER_CASES[name]
    gold = g(f"g2_forward_{name}")
    prot, batch, x, t, mask = cases.denoiser_inputs(L, B, seed)
    cg_z, cg_xyz, m = denoiser.batch_to_dense(batch)
    taps = {}
    out = denoiser.forward(sd, x, t, cg_xyz, cg_z, mask, taps=taps)
    assert torch.equal(taps["E_idx"], torch.from_numpy(gold["E_idx"]))      # int64: exact
    assert rel_err(out, gold["out"]) < 2e-6
    if "h_E0" in gold.files:
        nn_ = gold["h_E0"].shape[1]
        assert rel_err(taps["h_E0"][:, :nn_], gold["h_E0"]) < 2e-6
        for l in range(3):
            assert rel_err(taps[f"enc{l}_hV"], gold[f"enc{l}_hV"]) < 2e-6
            assert rel_err(taps[f"enc{l}_hE"][:, :nn_], gold[f"enc{l}_hE"]) < 2e-6
            assert rel_err(taps[f"dec{l}_hV"], gold[f"dec{l}_hV"]) < 2e-6
    # the reference's doubled "CFG" batch (test.py:505) does not change the first half
    assert float(gold["max_abs_diff_doubled"]) < 1e-5


def test_g2_forward_padded(sd):
    name, lengths, seed = cases.PADDED_CASE
    gold = g(f"g2_forward_{name}")
    batch, x, t, mask = cases.padded_inputs(lengths, seed)
    cg_z, cg_xyz, m = denoiser.batch_to_dense(batch)
    assert torch.equal(m, mask)
    taps = {}
    out = denoiser.forward(sd, x, t, cg_xyz, cg_z, mask, taps=taps)
    assert torch.equal(taps["E_idx"], torch.from_numpy(gold["E_idx"]))
    assert rel_err(out, gold["out"]) < 2e-6


@pytest.mark.parametrize("name", list(cases.LOOP_CASES))
def test_g3_loop(sd, name):
    L, B, seed, T = cases.LOOP_CASES[name]
    if T > 10:
        pytest.skip("100-step CPU loop: covered by test_g3_loop_100 (slow marker)")
    gold = g(f"g3_loop_{name}")
    prot, batch, _x, _t, mask = cases.denoiser_inputs(L, B, seed)
    cg_z, cg_xyz, _ = denoiser.batch_to_dense(batch)
    z, eps = cases.loop_noise(T, B, L, seed)
    x, traj = sampler.p_sample_loop(sd, T, z, eps, cg_xyz, cg_z, mask, return_traj=True)
    for k in range(T):
        assert rel_err(traj[k], gold["traj"][k]) < 5e-6, k
    assert rel_err(x, gold["sample"]) < 5e-6
    # hoisting the step-invariant CA features out of the loop changes nothing
    xh = sampler.p_sample_loop(sd, T, z, eps, cg_xyz, cg_z, mask, hoist_features=True)
    assert torch.equal(x, xh)


def test_g3_loop_100(sd):
    L, B, seed, T = cases.LOOP_CASES["L87_B1_T100"]
    gold = g("g3_loop_L87_B1_T100")
    prot, batch, _x, _t, mask = cases.denoiser_inputs(L, B, seed)
    cg_z, cg_xyz, _ = denoiser.batch_to_dense(batch)
    z, eps = cases.loop_noise(T, B, L, seed)
    x, traj = sampler.p_sample_loop(sd, T, z, eps, cg_xyz, cg_z, mask, hoist_features=True,
                                    return_traj=True)
    for j, k in enumerate(range(9, 100, 10)):
        assert rel_err(traj[k], gold["traj_every10"][j]) < 2e-5, k
    assert rel_err(x, gold["sample"]) < 2e-5


@pytest.mark.parametrize("vae_type,dataname", [("N6", "PED"), ("K3", "PDB"), ("K4", "Atlas")])
def test_g4_vq(vae_type, dataname):
    gold = g(f"g4_vq_{vae_type}")
    vsd = synth.vqvae_state_dict(vae_type, dataname, cases.VAE_SEED)
    mean, std = synth.norm_stats(dataname, vae_type)
    x = synth.gaussian((4, 77, 3), 123 + len(dataname))
    lat = vae_decode.denormalise(x, mean, std)
    assert torch.equal(lat, torch.from_numpy(gold["latent"]))
    zq, idx = vae_decode.vq_lookup(lat, vae_decode.codebook_of(vsd))
    assert torch.equal(idx, torch.from_numpy(gold["idx"]))                   # bit-exact indices
    assert torch.equal(zq, torch.from_numpy(gold["z_q"]))


def _vae_sd(vae_type, dataname, real_c2=False):
    vsd = synth.vqvae_state_dict(vae_type, dataname, cases.VAE_SEED, c2_like_map_out=real_c2)
    if real_c2:
        w = g("c2_decoder_weights")
        for k in w.files:
            vsd[k] = torch.from_numpy(w[k])
    return vsd


@pytest.mark.parametrize("name", list(cases.DECODER_CASES) + ["realC2_L87_B2"])
def test_g5_decoder(name):
    real = name.startswith("realC2")
    L, B, seed, vae_type = cases.DECODER_CASES["N6_L87_B2" if real else name]
    gold = g(f"g5_decode_{name}")
    prot, batch, latent, dataname = cases.decoder_inputs(L, B, seed, vae_type)
    vsd = _vae_sd(vae_type, dataname, real)
    idx, ic = vae_decode.latent_decode(vsd, latent, batch, angle=vae_type in ("K3", "K4"))
    if "idx" in gold.files:
        assert torch.equal(idx, torch.from_numpy(gold["idx"]))
    assert rel_err(ic, gold["ic_recon"]) < 5e-6


@pytest.mark.parametrize("name", list(cases.DECODER_CASES))
def test_g6_ic_to_xyz(name):
    L, B, seed, vae_type = cases.DECODER_CASES[name]
    prot, batch, latent, dataname = cases.decoder_inputs(L, B, seed, vae_type)
    ic = torch.from_numpy(g(f"g5_decode_{name}")["ic_recon"]).reshape(-1, L, 13, 3)
    og = batch["OG_CG_nxyz"].reshape(-1, L + 2, 4)
    xyz = vae_decode.ic_to_xyz(og, ic, prot["info"])
    gold = torch.from_numpy(g(f"g6_xyz_{name}")["xyz"])
    assert xyz.shape == gold.shape
    assert float((xyz - gold).abs().max()) < 1e-4      # Angstrom


def test_g7_end_to_end_T10(sd):
    name = "PED_N6_L46_B2_T10"
    L, B, seed, T, vae_type, dataname = cases.E2E_CASES[name]
    gold = g(f"g7_e2e_{name}")
    prot, batch, _x, _t, mask = cases.denoiser_inputs(L, B, seed)
    cg_z, cg_xyz, _ = denoiser.batch_to_dense(batch)
    z, eps = cases.loop_noise(T, B, L, seed)
    samples = sampler.p_sample_loop(sd, T, z, eps, cg_xyz, cg_z, mask)
    assert rel_err(samples, gold["samples"]) < 5e-6
    mean, std = synth.norm_stats(dataname, vae_type)
    vsd = synth.vqvae_state_dict(vae_type, dataname, cases.VAE_SEED)
    idx, ic = vae_decode.latent_decode(vsd, vae_decode.denormalise(samples, mean, std), batch)
    assert torch.equal(idx, torch.from_numpy(gold["idx"]))
    xyz = vae_decode.ic_to_xyz(batch["OG_CG_nxyz"].reshape(-1, L + 2, 4),
                               ic.reshape(-1, L, 13, 3), prot["info"])
    rmsd = float(((xyz - torch.from_numpy(gold["xyz"])) ** 2).sum(-1).mean().sqrt())
    assert rmsd < 1e-4


@pytest.mark.parametrize("name", list(cases.METRIC_CASES))
def test_g8_metrics(name):
    """Evaluation helpers after the path (reference test.py:97-166), incl. empty-list branches."""
    from oracle import metrics as om
    gold = np.load(cases.npz_path(f"g8_metrics_{name}"))
    d = cases.metric_inputs(name)
    bond, angle, torsion = om.recon_result(d["ic_recon"], d["ic"], d["mask"])
    inter, pipi = om.inter_result(d["interaction_list"], d["pi_pi_list"], d["xyz_recon"])
    got = dict(loss_bond=bond, loss_angle=angle, loss_torsion=torsion,
               loss_xyz=om.xyz_result(d["xyz_recon"], d["xyz"]),
               loss_graph=om.ged_result(d["xyz_recon"], d["xyz"], d["edge_list"]),
               loss_nbr=om.clash_result(d["edge_list"], d["nbr_list"], d["xyz_recon"], d["bb_NO_list"]),
               loss_inter=inter, loss_pi_pi=pipi)
    for k, v in got.items():
        assert float(v) == pytest.approx(float(gold[k]), rel=1e-6, abs=1e-9), k


@pytest.mark.parametrize("name", list(cases.SELF_COND_CASES))
def test_g9_self_conditioning(name):
    """--self_condition: x_in over cat(x_self_cond, x) and the pred_xstart feedback of the sampler
    (latent_model.py:210-212, gaussian_diffusion.py:530-547)."""
    L, B, seed, T = cases.SELF_COND_CASES[name]
    gold = g(f"g9_selfcond_{name}")
    sd_sc = synth.denoiser_state_dict(cases.WEIGHT_SEED, self_condition=True)
    prot, batch, x, t, mask = cases.denoiser_inputs(L, B, seed)
    cg_z, cg_xyz, m = denoiser.batch_to_dense(batch)
    xsc = synth.gaussian((B, L, 3), 6000 + seed)
    assert rel_err(denoiser.forward(sd_sc, x, t, cg_xyz, cg_z, m), gold["out_none"]) < 1e-5
    assert rel_err(denoiser.forward(sd_sc, x, t, cg_xyz, cg_z, m, x_self_cond=xsc), gold["out_sc"]) < 1e-5
    z, eps = cases.loop_noise(T, B, L, seed)
    xs, traj = sampler.p_sample_loop(sd_sc, T, z, eps, cg_xyz, cg_z, m, return_traj=True, self_condition=True)
    assert rel_err(torch.stack(traj), gold["traj"]) < 2e-5
    assert rel_err(xs, gold["sample"]) < 2e-5


@pytest.mark.parametrize("name", list(cases.ENVELOPE_CASES))
def test_g10_envelope_weight_sets(name):
    """The oracle on the weight sets that probe the split-fp16 envelope, incl. the reference constructor's own
    initialisation replayed by synth.reference_init_state_dict (gen_golden asserts it IS the constructor's)."""
    L, B, seed = cases.ENVELOPE_GEOMETRY
    gold = g(f"g10_envelope_{name}")
    wsd = cases.envelope_state_dict(name)
    prot, batch, x, t, mask = cases.denoiser_inputs(L, B, seed)
    cg_z, cg_xyz, m = denoiser.batch_to_dense(batch)
    out = denoiser.forward(wsd, x, t, cg_xyz, cg_z, mask)
    assert rel_err(out, gold["out"]) < 2e-6
    if name == "xavier":
        assert len(wsd) == 108 and sum(v.numel() for v in wsd.values()) == 2449974
        z, eps = cases.loop_noise(10, B, L, seed)
        xs = sampler.p_sample_loop(wsd, 10, z, eps, cg_xyz, cg_z, mask)
        assert rel_err(xs, gold["sample"]) < 1e-5


@pytest.mark.parametrize("name", list(cases.VALIDITY_CASES))
def test_g11_bond_graph_validity(name):
    """oracle/metrics.valid_ratio_and_cut_off_result against the reference's (ase.Atoms stood in for by a plain
    container when the golden was made: it carries no arithmetic on this path)."""
    from codlad_amd.metrics import COV_CUTOFF
    from oracle import metrics as om
    gold = g(f"g11_validity_{name}")
    d = cases.validity_inputs(name)
    hv, av, hg, ag = om.valid_ratio_and_cut_off_result(d["xyz"], d["xyz_recon"], d["num_atoms"].tolist(),
                                                       d["atomic_nums"], COV_CUTOFF)
    assert hv == gold["heavy_valid"].tolist() and av == gold["all_valid"].tolist()
    assert np.array_equal(np.array(hg, dtype=np.float64), gold["heavy_ged"])
    assert np.array_equal(np.array(ag, dtype=np.float64), gold["all_ged"])


@pytest.mark.parametrize("name", list(cases.FLOW_CASES))
def test_g12_flow_matching_model_and_fixed_grid_solvers(name):
    """Flow-matching model (velocity head, fractional t) against the reference model's outputs, and the oracle's
    fixed-grid Euler / RK4 (3/8) loops over it against the trajectories gen_golden integrated over the reference
    model (the solver layer itself is unpinned: torchdiffeq is absent)."""
    from oracle import flow as oflow
    L, B, seed, times, n_steps = cases.FLOW_CASES[name]
    gold = g(f"g12_flow_{name}")
    fsd = synth.denoiser_state_dict(cases.WEIGHT_SEED, flow=True)
    assert fsd["W_out.linear.weight"].shape == (3, 128)
    prot, batch, x, _t, mask = cases.denoiser_inputs(L, B, seed)
    cg_z, cg_xyz, m = denoiser.batch_to_dense(batch)
    f = oflow.velocity_fn(fsd, cg_xyz, cg_z, mask)
    for k, t in enumerate(times):
        v = f(t, x)
        assert v.shape == (B, L, 3) and rel_err(v, gold[f"v_t{k}"]) < 2e-6
    ts = torch.linspace(0, 1, n_steps + 1).tolist()
    for method in ("euler", "rk4"):
        y = oflow.odeint_fixed(f, x, ts, method)[-1]
        assert rel_err(y, gold[method]) < 1e-5, method


# ----------------------------------------------------------------------------------------------------------------
# Row 8f-1: g15 = the reference's OWN e3nnPrior.forward / e3nnEncoder.forward / TensorProductConvLayer.forward
# (models/vae_model.py:112-204, 275-311, models/gcn_nn.py:200-219), executed by tools/gen_golden.py with e3nn.o3 bound
# to thin adapters over oracle/e3nn_lite.py's primitives.  What this pins: the oracle's restatement of those reference
# lines - graph construction, the edge attributes the four conv stacks share, fc, scatter-mean, padding residuals, the
# bead mean and the dense heads - GIVEN the restated primitives.  e3nn's primitives themselves stay unpinned against
# e3nn (what reference-held data says about them: tests/test_e3nn_encoder.py).
# ----------------------------------------------------------------------------------------------------------------
def _c2_prior_sd():
    fx = np.load(cases.npz_path("c2_prior_e3nn"))
    return {k[len("prior_net."):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("prior_net.")}


@pytest.mark.parametrize("name", list(cases.E3NN_PRIOR_CASES))
def test_g15_prior_forward_is_the_references(name):
    from oracle import e3nn_lite as e3
    L, frames, wseed, weights = cases.E3NN_PRIOR_CASES[name]
    gold = g(f"g15_prior_{name}")
    sd_p = _c2_prior_sd() if weights == "trained_c2" else synth.prior_state_dict(wseed)
    batch = synth.make_batch(synth.make_protein(L, 40 + L, n_frames=frames))
    mu, sigma = e3.prior_forward(sd_p, batch["CG_nxyz"][:, 0].long(), batch["CG_nxyz"][:, 1:], batch["CG_nbr_list"])
    assert mu.shape == gold["mu"].shape == (L * frames, 36)
    assert rel_err(mu, gold["mu"]) < 2e-6 and rel_err(sigma, gold["sigma"]) < 2e-6


@pytest.mark.parametrize("name", list(cases.E3NN_ENCODER_CASES))
def test_g15_encoder_forward_is_the_references(name):
    from oracle import e3nn_lite as e3
    L, frames, wseed = cases.E3NN_ENCODER_CASES[name]
    gold = g(f"g15_encoder_{name}")
    prot = synth.make_protein(L, 50 + L, n_frames=frames)
    batch, atoms = synth.make_batch(prot), synth.make_atoms(prot, seed=L)
    out = e3.encoder_forward(synth.encoder_state_dict(wseed), atoms["nxyz"][:, 0], atoms["nxyz"][:, 1:],
                             batch["CG_nxyz"][:, 0].long(), batch["CG_nxyz"][:, 1:], atoms["CG_mapping"],
                             atoms["nbr_list"], batch["CG_nbr_list"])
    assert out.shape == gold["latent"].shape == (L * frames, 36)
    assert rel_err(out, gold["latent"]) < 2e-6


def test_g15_conv_layer_updates_of_every_stack():
    """The small encoder case carries what every TensorProductConvLayer of the reference returned (atom, bead, bead ->
    atom, atom -> bead stacks, three depths): the oracle's layer restatement, fed the same inputs, returns the same."""
    from oracle import e3nn_lite as e3
    L, frames, wseed = cases.E3NN_ENCODER_CASES["L12"]
    gold = g("g15_encoder_L12")
    sd_e = synth.encoder_state_dict(wseed)
    prot = synth.make_protein(L, 50 + L, n_frames=frames)
    batch, atoms = synth.make_batch(prot), synth.make_atoms(prot, seed=L)
    mids = {}
    orig = e3.tp_conv_layer

    def spy(sd, prefix, *a, **k):
        out = orig(sd, prefix, *a, **k)
        mids["upd_" + prefix.replace(".", "_")] = out
        return out

    e3.tp_conv_layer = spy
    try:
        e3.encoder_forward(sd_e, atoms["nxyz"][:, 0], atoms["nxyz"][:, 1:], batch["CG_nxyz"][:, 0].long(),
                           batch["CG_nxyz"][:, 1:], atoms["CG_mapping"], atoms["nbr_list"], batch["CG_nbr_list"])
    finally:
        e3.tp_conv_layer = orig
    keys = [k for k in gold.files if k.startswith("upd_")]
    assert len(keys) == 10 and set(keys) == set(mids)        # 3 + 3 atom-side, 2 + 2 bead-side layers run
    for k in keys:
        assert rel_err(mids[k], gold[k]) < 2e-6, k


def _branch_kwargs(kw):
    return dict(predict_xstart=bool(kw.get("predict_xstart")),
                var_type="learned_range" if kw.get("learn_sigma", True) else ("fixed_small" if kw.get("sigma_small") else "fixed_large"))


@pytest.mark.parametrize("name", list(cases.SAMPLER_BRANCH_CASES))
def test_g16_sampler_branches(sd, name):
    """The other branches of p_mean_variance (gaussian_diffusion.py:303-349): x_0-prediction, fixed small / large variance
    (on the network's 3-output variant), clip_denoised - the reference's own create_diffusion(...) loops."""
    L, B, seed, T, kw, clip, three = cases.SAMPLER_BRANCH_CASES[name]
    gold = g(f"g16_sampler_{name}")
    prot, batch, _x, _t, mask = cases.denoiser_inputs(L, B, seed)
    cg_z, cg_xyz, _ = denoiser.batch_to_dense(batch)
    z, eps = cases.loop_noise(T, B, L, seed)
    weights = synth.denoiser_state_dict(cases.WEIGHT_SEED, flow=True) if three else sd
    x, traj = sampler.p_sample_loop(weights, T, z, eps, cg_xyz, cg_z, mask, return_traj=True, hoist_features=True,
                                    clip_denoised=clip, **_branch_kwargs(kw))
    for k in range(T):
        assert rel_err(traj[k], gold["traj"][k]) < 5e-6, k
    assert rel_err(x, gold["sample"]) < 5e-6
    if clip:        # the clamp is live in these cases: without it the trajectory is another one
        x_noclip = sampler.p_sample_loop(weights, T, z, eps, cg_xyz, cg_z, mask, hoist_features=True, **_branch_kwargs(kw))
        assert rel_err(x_noclip, gold["sample"]) > 1e-3
