# kernel stats of the small-job DDPM loop (cases 0 and 2 of tools/small_job_latency.py) with the wide edge kernels on
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in 0 2; do
  CODLAD_EDGE_WIDE_MAX_TILES=1048576 timeout -k 10 250 rocprofv3 --kernel-trace --stats -d gpurun_out/wide_c$c -o p --output-format csv -- python3 tools/small_job_latency.py --only $c --reps 2 > gpurun_out/wide_c$c.log 2>&1
  python - <<PY
import csv, glob
f = glob.glob("gpurun_out/wide_c$c/**/p_kernel_stats.csv", recursive=True)[0]
print("case $c")
for r in list(csv.DictReader(open(f)))[:9]:
    print(f"{r['Name'][:70]:70s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e3:7.1f} us {float(r['Percentage']):5.1f}%")
PY
done
