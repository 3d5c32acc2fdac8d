"""Builds libcodlad_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m codlad_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcodlad_hip.so")
SOURCES = ["api.hip", "denoiser_kernels.hip", "features_kernels.hip", "decode_kernels.hip", "metrics_kernels.hip"]
# Geometry / VQ kernels must round like the reference's unfused CPU ops (bit-exact neighbour lists
# and code indices): no implicit FMA contraction there; intended FMAs are written as fmaf().
EXTRA_FLAGS = {"features_kernels.hip": ["-ffp-contract=off"], "decode_kernels.hip": ["-ffp-contract=off"],
               "metrics_kernels.hip": ["-ffp-contract=off"],
               # SLP packing of the shuffle-reduction adds blocks their fusion into v_add_f32_dpp;
               # the packed math that pays (GELU) is written out explicitly in common.h
               # -fno-honor-nans: min/max on MFMA results otherwise get a canonicalising v_max x,x
               # each (3 instructions for min(|x|, c)); nothing on this path produces or tests NaN
               "denoiser_kernels.hip": ["-fno-slp-vectorize", "-fno-honor-nans"]}
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "codlad_hip.h")]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


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
                  ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
