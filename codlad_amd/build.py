"""Builds libcodlad_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m codlad_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcodlad_hip.so")
SOURCES = ["api.hip", "denoiser_kernels.hip", "edge_msg_kernel.hip", "edge_upd_kernel.hip", "edge_upd1_kernel.hip", "edge_tile_kernels.hip",
           "edge_wide_kernels.hip",
           "node_wide_kernels.hip", "node_quad_kernels.hip", "ode_kernels.hip", "features_kernels.hip", "decode_kernels.hip",
           "ic_decoder_kernels.hip", "encoder_kernels.hip", "encoder_mfma_kernel.hip", "metrics_kernels.hip"]
# Geometry / VQ kernels must round like the reference's unfused CPU ops (bit-exact neighbour lists
# and code indices): no implicit FMA contraction there; intended FMAs are written as fmaf().
EXTRA_FLAGS = {"features_kernels.hip": ["-ffp-contract=off"], "ode_kernels.hip": ["-ffp-contract=off"], "decode_kernels.hip": ["-ffp-contract=off"],
               "ic_decoder_kernels.hip": ["-ffp-contract=off"],
               "metrics_kernels.hip": ["-ffp-contract=off"], "encoder_kernels.hip": ["-ffp-contract=off"],
               # the SLP vectoriser pairs multiply-adds of different result blocks into v_pk_fma_f32 and pays for it in register
               # moves (337 -> 80 v_mov at depth 0, 14 scratch accesses -> 0 at depth 2 without it)
               "encoder_mfma_kernel.hip": ["-ffp-contract=off", "-fno-slp-vectorize"],
               # SLP packing of the shuffle-reduction adds blocks their fusion into v_add_f32_dpp;
               # the packed math that pays (GELU) is written out explicitly in common.h
               # -fno-honor-nans: min/max on MFMA results otherwise get a canonicalising v_max x,x
               # each (3 instructions for min(|x|, c)); nothing on this path produces or tests NaN
               "denoiser_kernels.hip": ["-fno-slp-vectorize", "-fno-honor-nans"],
               "edge_tile_kernels.hip": ["-fno-slp-vectorize", "-fno-honor-nans"],
               "edge_wide_kernels.hip": ["-fno-slp-vectorize", "-fno-honor-nans"],
               "edge_msg_kernel.hip": ["-fno-slp-vectorize", "-fno-honor-nans"],
               "edge_upd_kernel.hip": ["-fno-slp-vectorize", "-fno-honor-nans"],
               # one wave per SIMD: with a 512-register budget hipcc selects the AGPR form of every MFMA (accumulators in
               # AGPRs, ~450 v_accvgpr_read/write per tile around the vector work on them); the VGPR form keeps the
               # accumulators where GELU / LayerNorm read them, and the register-resident weights are placed in AGPRs by hand
               "edge_upd1_kernel.hip": ["-fno-slp-vectorize", "-fno-honor-nans", "-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "node_wide_kernels.hip": ["-fno-slp-vectorize", "-fno-honor-nans"],
               "node_quad_kernels.hip": ["-fno-slp-vectorize", "-fno-honor-nans"]}
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "encoder_common.h"), os.path.join(CSRC, "edge_args.h"), os.path.join(CSRC, "node_args.h"),
           os.path.join(HERE, "..", "include", "codlad_hip.h")]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


RESOURCES = os.path.join(CSRC, "kernel_resources.json")


def record_resources(source, remarks):
    """Registers / scratch / occupancy hipcc reports per kernel (-Rpass-analysis=kernel-resource-usage), kept
    beside the objects: the edge kernels sit at the 256-register limit and a few spilled registers in their inner
    loops cost ~10 % (tests/test_host_logic.py holds them to their budget)."""
    import json
    import re
    table = {}
    if os.path.exists(RESOURCES):
        with open(RESOURCES) as f:
            table = json.load(f)
    # this source's old entries go, and those of sources that are no longer built (an experiment's leftovers would be
    # held to the budgets of kernels with similar names)
    table = {k: v for k, v in table.items() if v.get("source") != source and v.get("source") in SOURCES}
    cur = None
    for line in remarks.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = table.setdefault(m.group(1), {"source": source})
            continue
        m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" [")[0]] = int(m.group(2))
    with open(RESOURCES, "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not stale():
        return LIB
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(
                os.path.getmtime(d) for d in [src] + HEADERS):
            cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + EXTRA_FLAGS.get(s, []) + \
                  os.environ.get("CODLAD_CXXFLAGS", "").split() + \
                  ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            res = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
            if res.returncode:
                sys.stderr.write(res.stderr)
                raise subprocess.CalledProcessError(res.returncode, cmd)
            record_resources(s, res.stderr)
        objs.append(obj)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
