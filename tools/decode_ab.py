"""Bit-for-bit A/B of the decoder tail (VQ + IC decoder + ic_to_xyz) between library builds on ONE box.

    python tools/decode_ab.py variants/libcodlad_other.so

Runs the cfg5 job (400 structures, N6) and a K3 job (angle decoder) through the default build and through the other
library (CODLAD_HIP_LIB, same ABI), each in its own process, and compares code indices, internal coordinates' effect
(the Cartesian output) and timing.  Used in round 2 to show that the lane-per-residue / wave-per-receiver decoder
kernels return exactly the bits of the round-1 one-wave-per-residue kernels they replace."""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    from codlad_amd import synth
    from tests import pipeline
    out = {}
    for name, vae in (("cfg5", "N6"), ("k3", "K3")):
        lengths = list(synth.PED_LENGTHS) if vae == "N6" else [57, 128, 301]
        cfg = pipeline.Config(name, lengths, n_frames=10 if vae == "N6" else 3, n_ensemble=10 if vae == "N6" else 4,
                              vae_type=vae, dataname="PED" if vae == "N6" else "PDB")
        units = list(range(len(cfg.units)))
        res = cfg.run_units(units, decode_only=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = cfg.run_units(units, decode_only=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        h_idx, h_xyz = hashlib.sha256(), hashlib.sha256()
        for u in units:
            h_idx.update(res[u][1].cpu().numpy().tobytes())
            h_xyz.update(res[u][2].cpu().numpy().tobytes())
        out[name] = {"units": len(units), "idx": h_idx.hexdigest(), "xyz": h_xyz.hexdigest(), "host_s": round(dt, 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    if os.environ.get("DECODE_AB_CHILD"):
        child()
        sys.exit(0)
    results = {}
    for lib in [None] + sys.argv[1:]:
        env = dict(os.environ, DECODE_AB_CHILD="1")
        if lib:
            env["CODLAD_HIP_LIB"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stdout[-2000:], r.stderr[-4000:])
            sys.exit(1)
        results[lib or "default"] = json.loads(r.stdout.strip().splitlines()[-1])
        print(lib or "default", json.dumps(results[lib or "default"]), flush=True)
    base = results["default"]
    ok = all(r[k]["idx"] == base[k]["idx"] and r[k]["xyz"] == base[k]["xyz"] for r in results.values() for k in base)
    print("bit-identical across builds" if ok else "BUILDS DIFFER")
    sys.exit(0 if ok else 1)
