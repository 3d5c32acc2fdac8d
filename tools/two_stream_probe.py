"""Does splitting cfg 2's job into independent parts on separate HIP streams fill the CUs the node kernel leaves idle (139
workgroups on 256 CUs) and the kernels' tails?  Times the DDPM loop of the whole job on one stream against the same
structures as P jobs (units dealt round-robin by protein) on P streams.   python tools/two_stream_probe.py [P ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
wl = bench.Workload(dev, "cfg2")
wl.prepass()
parts_list = [int(a) for a in sys.argv[1:]] or [2]


def timed(fn, n=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def whole():
    return wl.den.sample(wl.job, wl.x_T, wl.noise, wl.tables, check=False, streams=1)


t_whole = timed(whole)
print(f"one job, one stream: {t_whole * 1e3:.1f} ms per 100-step loop ({wl.n_structures / t_whole:.1f} structures/s, loop only)", flush=True)
s_key = sorted({u[:2] for u in wl.units})
s_of = {k: i for i, k in enumerate(s_key)}
for P in parts_list:
    jobs = []
    for p in range(P):
        units = wl.units[p::P]                                  # every P-th unit: the same mix of lengths in every part
        job = wl.den.make_job(wl.structures, [s_of[u[:2]] for u in units])
        g = torch.Generator(device=dev).manual_seed(7 + p)
        x_T = torch.randn(job.n_nodes, 3, generator=g, device=dev)
        noise = torch.randn(bench.T_STEPS, job.n_nodes, 3, generator=g, device=dev)
        jobs.append((job, x_T, noise, torch.cuda.Stream(device=dev)))

    def split():
        cur = torch.cuda.current_stream(dev)
        for job, x_T, noise, st in jobs:
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                wl.den.sample(job, x_T, noise, wl.tables, check=False, streams=1)
        for _job, _x, _n, st in jobs:
            cur.wait_stream(st)

    t = timed(split)
    print(f"{P} jobs on {P} streams: {t * 1e3:.1f} ms ({wl.n_structures / t:.1f} structures/s, loop only; x{t_whole / t:.3f})", flush=True)
