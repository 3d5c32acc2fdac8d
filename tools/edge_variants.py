"""Timing-only builds of ONE edge-kernel translation unit with extra -D flags, and their back-to-back launch times.

    python tools/edge_variants.py build edge_upd1_kernel.hip u1a1=-DU1_ABLATE=1 u1a5=-DU1_ABLATE=5 ...   (here, CPU)
    python tools/edge_variants.py time [--opt EDGE_UPD_VARIANT=1] variants/libcodlad_u1a1.so ...        (GPU box)

`build` compiles the named source with the flags of codlad_amd/build.py plus the given ones and links it with the other
objects of the current build into variants/libcodlad_<name>.so; `time` loads each library in a process of its own
(CODLAD_HIP_LIB) and prints the average back-to-back launch time of the message and edge-update kernels on bench.py's cfg2
job (codlad_bench_edge_launch) - ablated builds compute wrong values on purpose, so nothing else is run with them.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "variants")


def build(argv):
    from codlad_amd import build as B
    src = argv[0]
    B.build()
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for spec in argv[1:]:
        name, _, flags = spec.partition("=")
        obj = os.path.join(OUT, f"{name}_{src.replace('.hip', '.o')}")
        cmd = [B.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + B.EXTRA_FLAGS.get(src, []) + \
              [f for f in flags.split(",") if f] + ["-c", os.path.join(B.CSRC, src), "-o", obj]
        procs.append((name, obj, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
    for name, obj, p in procs:
        err = p.communicate()[1]
        if p.returncode:
            sys.stderr.write(err)
            raise SystemExit(f"{name}: compile failed")
        objs = [obj if s == src else os.path.join(B.CSRC, s.replace(".hip", ".o")) for s in B.SOURCES]
        lib = os.path.join(OUT, f"libcodlad_{name}.so")
        subprocess.check_call([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
        print(lib, flush=True)


def time_one():
    import torch
    import bench
    from codlad_amd import _lib
    torch.set_grad_enabled(False)
    for kv in filter(None, os.environ.get("EDGE_VARIANTS_OPTS", "").split(",")):
        k, v = kv.split("=")
        _lib.set_option({"EDGE_UPD_VARIANT": 5, "EDGE_MSG_VARIANT": 6}[k], int(v))
    wl = bench.Workload(torch.device("cuda", 0), "cfg2")
    # realistic operands (a workspace fresh from torch.empty is mostly zeros, on which the chip clocks ~10 % higher and every
    # kernel looks faster): N(0,1) edge state in the stored hi / lo form, N(0,1) P / Q rows
    job = wl.job
    g = torch.Generator(device=job.hE.device).manual_seed(5)
    halves = job.hE.view(torch.float16).view(job.n_nodes, 2, 8, 4, 32, 8)          # [node][half][chunk / 4][chunk % 4][edge][8]
    halves[:, :, :, 0:2] = torch.randn(halves[:, :, :, 0:2].shape, generator=g, device=job.hE.device, dtype=torch.float16)
    halves[:, :, :, 2:4] = torch.randn(halves[:, :, :, 2:4].shape, generator=g, device=job.hE.device, dtype=torch.float16) * 2.0 ** -12
    job.PQ.normal_(generator=g)
    torch.cuda.synchronize()
    r = wl.time_dominant_kernel(12)
    print(f"msg {r['message'] * 1e3:.4f} ms   upd {r['edge_update'] * 1e3:.4f} ms", flush=True)


def time_all(argv):
    opts = ""
    if argv and argv[0] == "--opt":
        opts, argv = argv[1], argv[2:]
    for rep in range(2):
        for lib in [None] + argv:
            env = dict(os.environ, EDGE_VARIANTS_OPTS=opts)
            if lib:
                env["CODLAD_HIP_LIB"] = os.path.abspath(lib)
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "_one"], env=env, capture_output=True, text=True)
            line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr.strip().splitlines()[-1]
            print(f"{os.path.basename(lib) if lib else 'default':28s} {line}", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    elif sys.argv[1] == "time":
        time_all(sys.argv[2:])
    else:
        time_one()
