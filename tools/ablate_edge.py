"""Timing-only ablations of the two edge kernels (results of the ablated builds are WRONG on purpose).

    python tools/ablate_edge.py build     # here: patched copies of csrc/ -> variants/libcodlad_<name>.so
    python tools/ablate_edge.py run       # on the GPU box: time one launch of each kernel per variant

Each variant removes one ingredient of the kernels (a textual patch on a temporary copy of the
sources), so the difference to `base` is what that ingredient costs in place - which is how the
"where does the time go" table in DESIGN.md was obtained.
"""
import os
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "variants")
sys.path.insert(0, ROOT)

# Patches on the edge-UPDATE kernel (upd_kernel_h) and, for nogelu, on both kernels.
UPD = "edge_upd_kernel.hip"      # one hot kernel per translation unit (csrc/edge_args.h says why)
MSG = "edge_msg_kernel.hip"
XLOAD = "tile_load_edge<!HOISTED>(x, rows, colc, h);                          // layer-1 operand and residual"
QLOAD = "tile_load_row(acc, a.Q + (size_t)(base + j) * HD, h);"
STORE = "            if (valid) tile_store_edge<true>(x, out_rows, col, h);"
SPAN = "    const NodeSpan span = wave_node_span(a.n_nodes, NWAVES, wave);\n"
NOSTORE = "            if (valid && x.b[0][0] == 12345.f) tile_store_edge<true>(x, out_rows, col, h);"
# name -> (extra flags, [(file, old, new), ...])
VARIANTS = {
    "base": ([], []),
    "nogelu": ([], [("common.h", "    if (GELU_IN) {\n        f32x2 t[1] = {x};", "    if (false) {\n        f32x2 t[1] = {x};"),
                    (MSG, "        tile_gelu(t2, a.gelu_b);\n", "")]),
    # node kernel (timed through the whole job: tools/ablate_edge.py bench <variants>)
    "node_nofetch": ([], [("denoiser_kernels.hip", "        if (more) fetch(cur + 1);\n", ""),
                          ("denoiser_kernels.hip", "        if (more) commit(cur + 1);\n", "")]),
    "node_nobarrier": ([], [("denoiser_kernels.hip", "        if (more) commit(cur + 1);\n        __syncthreads();\n", "        if (more) commit(cur + 1);\n")]),
    "noglb": ([], [(UPD, "                tail1.run(acc, x, lane, a.gelu_a);                                    // layer 1, streamed k-steps\n", "")]),
    "noln": ([], [(UPD, "            tile_layernorm_affine(x, a.ln_eps, c_modA, c_modB, h);\n", "")]),
    "nostore": ([], [(UPD, STORE, NOSTORE)]),
    "noxload": ([], [(UPD, XLOAD, "tile_load_row(x, Pslot, h);")]),
    "noq": ([], [(UPD, QLOAD, "tile_load_row(acc, Pslot, h);")]),
    "nomem": ([], [(UPD, XLOAD, "tile_load_row(x, Pslot, h);"),
                   (UPD, QLOAD, "tile_load_row(acc, Pslot, h);"),
                   (UPD, STORE, NOSTORE)]),
    # GELU's Horner steps as scalar v_fma_f32 instead of v_pk_fma_f32 (packed fp32 does not co-issue with MFMAs)
    "scalar_gelu": ([], [("common.h", """    for (int i = 0; i < N; ++i) p[i] = t[i] * gk.c[0] + gk.c[1];
#pragma unroll
    for (int k = 2; k <= CODLAD_GELU_DEGREE; ++k)
#pragma unroll
        for (int i = 0; i < N; ++i) p[i] = p[i] * t[i] + gk.c[k];
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = p[i] * t[i] + -1.0f;     // s g(s) - 1""", """    for (int i = 0; i < N; ++i) { p[i].x = fmaf(t[i].x, gk.c[0], gk.c[1]); p[i].y = fmaf(t[i].y, gk.c[0], gk.c[1]); }
#pragma unroll
    for (int k = 2; k <= CODLAD_GELU_DEGREE; ++k)
#pragma unroll
        for (int i = 0; i < N; ++i) { p[i].x = fmaf(p[i].x, t[i].x, gk.c[k]); p[i].y = fmaf(p[i].y, t[i].y, gk.c[k]); }
#pragma unroll
    for (int i = 0; i < N; ++i) { p[i].x = fmaf(p[i].x, t[i].x, -1.0f); p[i].y = fmaf(p[i].y, t[i].y, -1.0f); }""")]),
    # one wave per SIMD (4 waves per workgroup) instead of two: what does the second wave add?
    "msg_4waves": ([], [(MSG, "    constexpr int NW = 8;\n    const size_t lds = 16 * edge_lds_u4<false, NW>();", "    constexpr int NW = 4;\n    const size_t lds = 16 * edge_lds_u4<false, NW>();"),
                        (MSG, "    static_assert(16 * edge_lds_u4<false, NW>() <= 160 * 1024", "    static_assert(16 * edge_lds_u4<false, 4>() <= 160 * 1024")]),
    # all four operand pairs of the next k-step converted in the first group of a k-step (four independent
    # GELU + split chains side by side) instead of one pair per group
    "ilp4": ([], [("common.h", "        if (ks + 1 < KS0 + NKS) split_pair<GELU_IN>(xn, in, ks + 1, bo, gk);\n        mfma_f16<TERMS, TRANSPOSED>(acc.b[bo], as_f16x8(ring[g % 3][0]), as_f16x8(ring[g % 3][1]), x);",
                   "        if (ks + 1 < KS0 + NKS && bo == 0) {\n#pragma unroll\n            for (int p = 0; p < 4; ++p) split_pair<GELU_IN>(xn, in, ks + 1, p, gk);\n        }\n        mfma_f16<TERMS, TRANSPOSED>(acc.b[bo], as_f16x8(ring[g % 3][0]), as_f16x8(ring[g % 3][1]), x);")]),
    "ilp2": ([], [("common.h", "        if (ks + 1 < KS0 + NKS) split_pair<GELU_IN>(xn, in, ks + 1, bo, gk);\n        mfma_f16<TERMS, TRANSPOSED>(acc.b[bo], as_f16x8(ring[g % 3][0]), as_f16x8(ring[g % 3][1]), x);",
                   "        if (ks + 1 < KS0 + NKS && (bo & 1) == 0) {\n            split_pair<GELU_IN>(xn, in, ks + 1, bo, gk);\n            split_pair<GELU_IN>(xn, in, ks + 1, bo + 1, gk);\n        }\n        mfma_f16<TERMS, TRANSPOSED>(acc.b[bo], as_f16x8(ring[g % 3][0]), as_f16x8(ring[g % 3][1]), x);")]),
    # message kernel
    "msg_noepi": ([], [(MSG, "        tile_gelu(t2, a.gelu_b);\n", "")]),
    # start the second wave of every SIMD (and, _wc, the workgroups) a fraction of a tile period late: identical waves
    # that start together ask for memory together for the whole launch (units of s_sleep 127 = 8 128 cycles ~ 4.3 us)
    "upd_stag_w": ([], [(UPD, SPAN, SPAN + "    for (int i = 0; i < (wave >= 4 ? 3 : 0); ++i) __builtin_amdgcn_s_sleep(127);\n")]),
    "upd_stag_wc": ([], [(UPD, SPAN, SPAN + "    for (int i = 0; i < (wave >= 4 ? 3 : 0) + (int)(blockIdx.x & 3); ++i) __builtin_amdgcn_s_sleep(127);\n")]),
    "upd_stag_fine": ([], [(UPD, SPAN, SPAN + "    for (int i = 0; i < (int)((wave * 37 + blockIdx.x * 11) % 48); ++i) __builtin_amdgcn_s_sleep(15);\n")]),
    "msg_stag_w": ([], [(MSG, SPAN, SPAN + "    for (int i = 0; i < (wave >= 4 ? 2 : 0); ++i) __builtin_amdgcn_s_sleep(127);\n")]),
    "msg_stag_fine": ([], [(MSG, SPAN, SPAN + "    for (int i = 0; i < (int)((wave * 37 + blockIdx.x * 11) % 32); ++i) __builtin_amdgcn_s_sleep(15);\n")]),
    # half / none of the weight-fragment reads from LDS (stale fragments reused): what do the 64 KB of LDS reads per
    # contraction and wave cost in time and in power?
    "half_lds": ([], [("common.h", "        if (g + 2 < NG) {\n            ring[(g + 2) % 3][0] = w[((G0 + g + 2) * 2 + 0) * 64];", "        if (g + 2 < NG && (g & 1) == 0) {\n            ring[(g + 2) % 3][0] = w[((G0 + g + 2) * 2 + 0) * 64];")]),
    "no_lds": ([], [("common.h", "        if (g + 2 < NG) {\n            ring[(g + 2) % 3][0] = w[((G0 + g + 2) * 2 + 0) * 64];", "        if (g + 2 < NG && g < 1) {\n            ring[(g + 2) % 3][0] = w[((G0 + g + 2) * 2 + 0) * 64];")]),
    # the edge update's memory side alone: loads (tile, Q rows, P row) and the store stay, the three contractions and the
    # LayerNorm go (one add keeps the Q rows live)
    "upd_memonly": ([], [(UPD, "                gemm_h_lds<TERMS, 0, UPD_W1_KS, false>(acc, x, w1, lane, a.gelu_a);   // layer 1, resident k-steps\n", ""),
                         (UPD, "                tail1.run(acc, x, lane, a.gelu_a);                                    // layer 1, streamed k-steps\n", ""),
                         (UPD, "            if (!HOISTED) tail1.start(a.W1h, lane);\n", ""),
                         (UPD, "            gemm128_h_lds<TERMS, true>(t2, acc, w2, lane, a.gelu_a);   // layer 2 on GELU(layer 1)\n", "            for (int b = 0; b < 4; ++b) x.b[b] += acc.b[b];\n"),
                         (UPD, "            gemm128_h_lds<TERMS, true>(x, t2, w3, lane, a.gelu_b);     // layer 3 on GELU(layer 2)\n", "            for (int b = 0; b < 4; ++b) x.b[b] += t2.b[b];\n"),
                         (UPD, "            tile_layernorm_affine(x, a.ln_eps, c_modA, c_modB, h);\n", "")]),
    "msg_prio": ([], [(MSG, "    const int h = lane >> 5, c = lane & 31;\n    const NodeSpan span", "    if (wave >= 4) __builtin_amdgcn_s_setprio(1);\n    const int h = lane >> 5, c = lane & 31;\n    const NodeSpan span")]),
}


def build_one(name):
    flags, patches = VARIANTS[name]
    tmp = tempfile.mkdtemp(prefix="ablate_" + name)
    src = os.path.join(tmp, "pkg", "csrc")      # the sources include "../../include/codlad_hip.h"
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    shutil.copytree(os.path.join(ROOT, "codlad_amd", "csrc"), src, ignore=shutil.ignore_patterns("*.o"))
    for fn, old, new in patches:
        p = os.path.join(src, fn)
        s = open(p).read()
        n = s.count(old)
        assert n >= 1, (name, fn, old[:40])
        open(p, "w").write(s.replace(old, new))
    from codlad_amd import build as b
    objs = []
    for s in b.SOURCES:
        obj = os.path.join(tmp, s.replace(".hip", ".o"))
        cmd = [b.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include")] + \
              b.EXTRA_FLAGS.get(s, []) + flags + ["-c", os.path.join(src, s), "-o", obj]
        subprocess.check_call(cmd)
        objs.append(obj)
    lib = os.path.join(OUT, f"libcodlad_{name}.so")
    subprocess.check_call([b.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    shutil.rmtree(tmp)
    return name


def run_one(name):
    import torch
    from codlad_amd import _lib
    _lib.LIB_PATH = os.path.join(OUT, f"libcodlad_{name}.so")
    import bench
    w = bench.Workload(torch.device("cuda:0"), "cfg2", precision=os.environ.get("CODLAD_PRECISION", "f16x3"))
    w.prepass()
    w.den.forward(w.job, w.x_T, 500, check=False)      # fills the workspace the hook reads
    t = w.time_dominant_kernel()
    print(f"{name:10s} message {t['message'] * 1e3:.4f} ms   edge_update {t['edge_update'] * 1e3:.4f} ms", flush=True)


MEMONLY = VARIANTS["upd_memonly"][1]
VARIANTS["memonly_nostore"] = ([], MEMONLY + [(UPD, STORE, NOSTORE)])
VARIANTS["memonly_noq"] = ([], MEMONLY + [(UPD, QLOAD, "tile_load_row(acc, Pslot, h);")])
VARIANTS["memonly_nox"] = ([], MEMONLY + [(UPD, XLOAD, "tile_load_row(x, Pslot, h);")])
VARIANTS["memonly_noq_nostore"] = ([], MEMONLY + [(UPD, QLOAD, "tile_load_row(acc, Pslot, h);"), (UPD, STORE, NOSTORE)])
# where does the Q gather's cost come from: every lane the same row (one line per instruction) / consecutive rows
VARIANTS["memonly_qsame"] = ([], MEMONLY + [(UPD, QLOAD, "tile_load_row(acc, a.Q + (size_t)base * HD, h);")])
VARIANTS["memonly_qseq"] = ([], MEMONLY + [(UPD, QLOAD, "tile_load_row(acc, a.Q + (size_t)(base + (c < K ? c : 0)) * HD, h);")])
# fewer k-steps of W11e resident in LDS (3 fit): what does each streamed k-step (8 fragment loads per tile and wave) cost?
for _ks in (2, 1, 0):
    VARIANTS[f"upd_ks{_ks}"] = ([], [("edge_args.h", "constexpr int UPD_W1_KS = 3;", f"constexpr int UPD_W1_KS = {_ks};")])
# what an edge state stored pre-split (fp16 hi / lo fragments instead of fp32) could save: the lo half of the split of a
# contraction's un-activated input removed (data stay finite; the whole split is 2.5 instructions per element, this is 1.5)
VARIANTS["nolo_plain"] = ([], [("common.h", "    const f16x2 hh = __builtin_convertvector(x, f16x2);\n    const f16x2 ll = split_lo_pair(hh, x);",
                                 "    const f16x2 hh = __builtin_convertvector(x, f16x2);\n    const f16x2 ll = GELU_IN ? split_lo_pair(hh, x) : f16x2{(_Float16)0, (_Float16)0};")])
# GELU polynomial one degree lower (9 instead of 10 vector instructions per element; max error 6e-7 .. 9e-7 instead of 2e-7 .. 5e-7)
VARIANTS["gelu_deg4"] = (["-DCODLAD_GELU_DEGREE=4"], [])


if __name__ == "__main__":
    mode = sys.argv[1]
    names = sys.argv[2:] or list(VARIANTS)
    if mode == "build":
        os.makedirs(OUT, exist_ok=True)
        with ThreadPoolExecutor(4) as ex:
            for n in ex.map(build_one, names):
                print("built", n, flush=True)
    elif mode == "run":
        for n in names:   # one process per variant: a process binds one library
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "run1", n])
    elif mode == "run1":
        run_one(names[0])
    elif mode == "bench":     # whole-job rate per variant (for kernels the launch hook does not cover)
        for n in names:
            env = dict(os.environ, CODLAD_HIP_LIB=os.path.join(OUT, f"libcodlad_{n}.so"))
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                                  "--no-cpu-baseline"], env=env, capture_output=True, text=True).stdout
            import json
            d = json.loads(out.strip().splitlines()[-1])
            print(f"{n:16s} {d['value']:.1f} structures/s   {d['ms_per_step']:.1f} ms per job", flush=True)
