"""Socket power and shader clock while the default bench job runs (evidence for the power-limit reading of DESIGN.md 6).

    python tools/power_trace.py > gpurun_out/power_trace.txt        (on the GPU box)

Starts `bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-f32-leg` as a child and samples `rocm-smi` (read-only queries)
about five times a second until it exits; then the same around an idle period for the baseline."""
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sample():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--showuse"], capture_output=True, text=True).stdout
    power = re.search(r"Socket Graphics Package Power \(W\):\s*([\d.]+)", out) or re.search(r"Power \(W\):\s*([\d.]+)", out)
    sclk = re.search(r"sclk clock level:\s*\d+:?\s*\(?(\d+)Mhz", out)
    temp = re.search(r"Temperature \(Sensor (?:junction|hotspot|edge)\) \(C\):\s*([\d.]+)", out)
    use = re.search(r"GPU use \(%\):\s*(\d+)", out)
    return (float(power.group(1)) if power else None, int(sclk.group(1)) if sclk else None,
            float(temp.group(1)) if temp else None, int(use.group(1)) if use else None, out)


def trace(label, proc_args, seconds=None):
    print(f"# {label}")
    proc = subprocess.Popen(proc_args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) if proc_args else None
    t0 = time.time()
    rows = []
    while (proc.poll() is None) if proc else (time.time() - t0 < seconds):
        p, c, t, u, _raw = sample()
        rows.append((time.time() - t0, p, c, t, u))
        time.sleep(0.1)
    for r in rows:
        print("t=%6.2f s  power %s W  sclk %s MHz  temp %s C  use %s %%" % r)
    busy = [r for r in rows if r[4] is not None and r[4] > 90 and r[1] is not None]
    if busy:
        print(f"# while busy (> 90 % use, {len(busy)} samples): power mean {sum(r[1] for r in busy) / len(busy):.0f} W, max "
              f"{max(r[1] for r in busy):.0f} W; sclk mean {sum(r[2] or 0 for r in busy) / len(busy):.0f} MHz, min "
              f"{min(r[2] or 0 for r in busy)} MHz")
    if proc:
        print("# child said:", (proc.stdout.read().strip().splitlines() or [""])[-1][:300])
    sys.stdout.flush()


if __name__ == "__main__":
    _p, _c, _t, _u, raw = sample()
    print("# one raw rocm-smi answer:\n" + "\n".join("#   " + l for l in raw.splitlines() if l.strip()))
    cap = subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True).stdout
    print("\n".join("#   " + l for l in cap.splitlines() if "Power" in l))
    trace("idle, 2 s", None, 2.0)
    trace("bench.py cfg2, 6 jobs", [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "1",
                                    "--no-cpu-baseline", "--no-f32-leg", "--no-cfg5"])
