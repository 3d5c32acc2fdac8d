"""A/B of library builds on ONE box: whole-job rate and the two edge kernels' launch times for each library given
(default build first).  python tools/ab_bench.py variants/libcodlad_x.so ..."""
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
        print(f"{os.path.basename(lib) if lib else 'default':28s} {d['value']:7.1f} structures/s   msg {d['roofline']['launch_ms']:.4f} ms   "
              f"upd {d['roofline']['edge_update_launch_ms']:.4f} ms", flush=True)
