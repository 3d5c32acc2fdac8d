"""profiles/<tag>_traffic.json (what bench.py quotes as roofline.traffic) from the counter summary of a profile round.

    python tools/traffic_json.py gpurun_out/r04_final/pmc_summary.txt profiles/r04_final_traffic.json

Reads FETCH_SIZE and WRITE_SIZE (KB per launch, rocprofv3 --pmc, separate passes) of msg_kernel_h<8, false, 3> and of
upd_kernel_h<8, false, 3>; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (the 128-byte requests of wide
coalesced reads are tallied at 64 bytes)."""
import json
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
vals = {}
cur = None
for line in open(src):
    m = re.match(r"\S.*?((?:msg|upd)_kernel_h<8, (?:false|true), 3>).*calls=(\d+)", line)
    if m:
        cur = m.group(1)
        vals.setdefault(cur, {})["calls"] = int(m.group(2))
        continue
    if not line.startswith(" "):
        cur = None
    m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.e+]+)", line)
    if m and cur:
        vals[cur][m.group(1)] = float(m.group(2))
msg, upd = vals["msg_kernel_h<8, false, 3>"], vals["upd_kernel_h<8, false, 3>"]
hbm = (2 * msg["FETCH_SIZE"] + msg["WRITE_SIZE"]) * 1024
out = {
    "kernel": "msg_kernel_h<8, false, 3>",
    "workload": f"bench.py cfg2 --streams 1, 1 GPU, f16x3, avg over the {msg['calls']} launches of the traced passes that read the per-sample "
                "edge state (rocprofv3 --pmc, the round's last build)",
    "FETCH_SIZE_KB": msg["FETCH_SIZE"], "WRITE_SIZE_KB": msg["WRITE_SIZE"],
    "hbm_bytes_per_launch": hbm,
    "algorithmic_bytes_per_launch": 1171968000,
    "note": "separate --pmc passes (tools/profile_round.sh: FETCH_SIZE, then WRITE_SIZE, each with --kernel-trace only); FETCH_SIZE doubled per "
            "MI355X_MICROARCH.md (gfx950 tallies the 128-B requests of wide coalesced reads at 64 B). Algorithmic: 2 182 800 edges x 512 B of "
            "edge state + P, Q, S rows of 35 400 nodes x 512 B.  upd_kernel_h<8, false, 3> of the same runs: FETCH "
            f"{upd['FETCH_SIZE']:.0f} KB x 2 + WRITE {upd['WRITE_SIZE']:.0f} KB = {(2 * upd['FETCH_SIZE'] + upd['WRITE_SIZE']) * 1024 / 1e9:.2f} GB per launch.",
}
with open(dst, "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
