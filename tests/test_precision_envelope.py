"""The envelope of the split-fp16 contraction modes (f16x3 default, f16x4), pinned on the GPU against goldens
the reference produced for weight sets chosen to stress it (tests/cases.py ENVELOPE_CASES, tools/gen_golden.py g10):
the reference constructor's own initialisation, uniformly small weight matrices, small / large first layers that
still carry the signal, large edge-feature gains, large adaLN vectors.  Plus the overflow sentinel: an operand
beyond the fp16 range must come back as an error code, never as a number."""
import numpy as np
import pytest
import torch

from codlad_amd import synth
from codlad_amd.diffusion_and_flow.schedule import Tables, named_betas, space_timesteps
from codlad_amd.engine import Denoiser
from tests import cases

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_err(a, b):
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def job_of(den, prot, B):
    frames = torch.from_numpy(prot["xyz_full"])[:, 1:-1]
    z = torch.from_numpy(prot["z_full"])[1:-1]
    st = den.prepare_structures([f for f in frames], [z for _ in frames])
    return den.make_job(st, list(range(B)))


@pytest.mark.parametrize("precision", ["f16x3", "f16x4", "f32"])
@pytest.mark.parametrize("name", list(cases.ENVELOPE_CASES))
def test_envelope_weight_sets_against_reference(name, precision):
    """Same tolerance as every other denoiser test (1e-5 relative to the output's max) in all three modes."""
    L, B, seed = cases.ENVELOPE_GEOMETRY
    gold = np.load(cases.npz_path(f"g10_envelope_{name}"))
    prot, batch, x, t, mask = cases.denoiser_inputs(L, B, seed)
    den = Denoiser(cases.envelope_state_dict(name), DEV, precision=precision)
    job = job_of(den, prot, B)
    out = den.forward(job, x.reshape(-1, 3).to(DEV), int(t[0])).cpu().view(B, L, 6)
    assert rel_err(out, gold["out"]) < 1e-5, (name, precision, rel_err(out, gold["out"]))
    if name == "xavier":
        T = 10
        z, eps = cases.loop_noise(T, B, L, seed)
        tb = Tables(named_betas("linear", 1000), space_timesteps(1000, str(T)))
        x0 = den.sample(job, z.reshape(-1, 3).to(DEV), eps.reshape(T, -1, 3).to(DEV), tb)
        assert rel_err(x0.cpu().view(B, L, 3), gold["sample"]) < 2e-5


@pytest.mark.parametrize("precision", ["f16x3", "f16x4"])
def test_fp16_range_overflow_is_an_error_not_a_number(precision):
    """Edge features of magnitude ~1e6 (features.norm_edges.weight x 1e6) are far outside the fp16 range: the
    split modes must report it through the status word (RuntimeError from the host layer, CODLAD_E_NONFINITE
    from codlad_status_check) in forward AND in the fused loop; the fp32-MFMA mode computes the same input
    to a finite result; and the status word is cleared by the check, so the job is usable again."""
    L, B, seed = cases.ENVELOPE_GEOMETRY
    prot, batch, x, t, mask = cases.denoiser_inputs(L, B, seed)
    sd = synth.denoiser_state_dict(cases.WEIGHT_SEED)
    sd["features.norm_edges.weight"] = sd["features.norm_edges.weight"] * 1e6
    xd = x.reshape(-1, 3).to(DEV)
    den = Denoiser(sd, DEV, precision=precision)
    job = job_of(den, prot, B)
    with pytest.raises(RuntimeError, match="not finite"):
        den.forward(job, xd, 500)
    T = 3
    z, eps = cases.loop_noise(T, B, L, seed)
    tb = Tables(named_betas("linear", 1000), space_timesteps(1000, str(T)))
    with pytest.raises(RuntimeError, match="not finite"):
        den.sample(job, z.reshape(-1, 3).to(DEV), eps.reshape(T, -1, 3).to(DEV), tb)
    assert int(job.status.item()) == 0                                   # cleared by the failed check
    out = den.forward(job, xd, 500, check=False)                         # unchecked call: NaNs come back as they are
    assert not bool(torch.isfinite(out).all())
    den.weights.set_precision("f32")
    job.status.zero_()
    out32 = den.forward(job, xd, 500)
    assert bool(torch.isfinite(out32).all())


@pytest.mark.parametrize("precision", ["f16x3", "f16x4"])
@pytest.mark.parametrize("site", ["edge_state", "node_state", "decoder_node_state"])
def test_overflow_is_reported_wherever_it_first_appears(precision, site):
    """The sentinel must not depend on WHERE a value first leaves the fp16 range (the edge kernels are compiled with
    -fno-honor-nans, under which a compiler is free to assume NaN away: a change that swallowed one of these would pass
    the features-only test above).  edge_state: encoder layer 0's gate3 / scale3 / shift3 heads x 1e6 - h_E overflows
    when the edge update stores it as fp16 halves; node_state: x_in x 1e7 - h_V overflows when the first node kernel
    splits it for the P / Q projections; decoder_node_state: decoder layer 1's adaLN heads x 1e7 - h_V overflows in
    the middle of the decoder.  In each case the f32 mode runs the same weights to a finite result."""
    L, B, seed = cases.ENVELOPE_GEOMETRY
    prot, batch, x, t, mask = cases.denoiser_inputs(L, B, seed)
    sd = synth.denoiser_state_dict(cases.WEIGHT_SEED)
    if site == "edge_state":
        for k in ("weight", "bias"):
            w = sd[f"encoder_layers.0.adaLN_modulation.1.{k}"].clone()
            w[6 * 128:] *= 1e6
            sd[f"encoder_layers.0.adaLN_modulation.1.{k}"] = w
    elif site == "node_state":
        sd["x_in.weight"] = sd["x_in.weight"] * 1e7
    else:
        for k in ("weight", "bias"):
            sd[f"decoder_layers.1.adaLN_modulation.1.{k}"] = sd[f"decoder_layers.1.adaLN_modulation.1.{k}"] * 1e7
    xd = x.reshape(-1, 3).to(DEV)
    den = Denoiser(sd, DEV, precision=precision)
    job = job_of(den, prot, B)
    with pytest.raises(RuntimeError, match="not finite"):
        den.forward(job, xd, 500)
    den.weights.set_precision("f32")
    job.status.zero_()
    assert bool(torch.isfinite(den.forward(job, xd, 500)).all())


def test_weights_beyond_fp16_range_are_refused_at_pack_time():
    """A weight matrix with no accurate fp16 hi/lo split (an element that no block exponent brings into range without
    costing the rest its precision) is refused when packed, instead of storing inf or garbage."""
    sd = synth.denoiser_state_dict(cases.WEIGHT_SEED)
    sd["encoder_layers.1.W2.weight"] = sd["encoder_layers.1.W2.weight"].clone()
    sd["encoder_layers.1.W2.weight"][3, 5] = 1.0e9
    with pytest.raises(ValueError, match="fp16 range"):
        Denoiser(sd, DEV, precision="f16x3")
    den = Denoiser(sd, DEV, precision="f32")                           # the fp32-MFMA mode takes it
    assert den.weights.precision == "f32"


def test_block_exponents_are_what_keeps_small_layers_accurate():
    """With every block exponent 0 (the plain hi/lo split) a net whose first MLP layers are small but carry the signal
    misses the 1e-5 bar (the `lo` halves of its weights are subnormal fp16: absolute, not relative, precision); with
    the exponents it is as accurate as on ordinary weights.  On ordinary weights the two differ by rounding only."""
    L, B, seed = cases.ENVELOPE_GEOMETRY
    prot, batch, x, t, mask = cases.denoiser_inputs(L, B, seed)
    xd = x.reshape(-1, 3).to(DEV)
    errs = {}
    for name in ("small_first_1e-2", "xavier"):
        gold = np.load(cases.npz_path(f"g10_envelope_{name}"))["out"]
        for on in (False, True):
            den = Denoiser(cases.envelope_state_dict(name), DEV, precision="f16x3", block_exponents=on)
            out = den.forward(job_of(den, prot, B), xd, int(t[0])).cpu().view(B, L, 6)
            errs[name, on] = rel_err(out, gold)
    assert errs["small_first_1e-2", True] < 5e-6 < 1e-5 < errs["small_first_1e-2", False], errs
    assert errs["xavier", True] < 5e-6 and errs["xavier", False] < 5e-6, errs
