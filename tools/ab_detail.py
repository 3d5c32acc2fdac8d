"""A/B of library builds on ONE box with the per-kind in-job launch times: python tools/ab_detail.py [lib.so ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [None] + sys.argv[1:]
for rep in range(2):
    for lib in libs:
        env = dict(os.environ)
        if lib:
            env["CODLAD_HIP_LIB"] = os.path.abspath(lib)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                              "--no-f32-leg", "--no-cfg5"], env=env, capture_output=True, text=True).stdout
        d = json.loads(out.strip().splitlines()[-1])
        r = d["roofline"]
        print(f"{os.path.basename(lib) if lib else 'default':24s} {d['value']:7.1f} structures/s  {d['ms_per_step']:.1f} ms  prepass {d.get('prepass_ms', 0):.2f} ms  "
              + "  ".join(f"{k} {v:.4f}" for k, v in r["in_job_launch_ms"].items()), flush=True)
