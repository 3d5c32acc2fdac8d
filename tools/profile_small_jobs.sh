# Round-4 small-job record: tools/small_job_latency.py as shipped, with the four-wave edge kernels off, with host-side kernel
# arguments; the size sweep; per-kernel stats (rocprofv3 --kernel-trace --stats) of one 87-residue protein and of the cfg-3 shard.
#   bash tools/profile_small_jobs.sh <tag>      ->  gpurun_out/<tag>_small_jobs.txt, gpurun_out/<tag>_small_job_kernel_stats.txt
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${TAG}_small_jobs.txt
{
echo "# tools/small_job_latency.py, build as shipped (four-wave edge kernels up to 22 x 256 tiles, four-wave node update up to 2 x 256 tiles, device-side kernel arguments)"
timeout -k 10 300 python tools/small_job_latency.py 2>/dev/null
echo "# CODLAD_NODE_QUAD_MAX_TILES=0 (node update on node_kernel_w, eight waves per tile, instead of node_kernel_q)"
CODLAD_NODE_QUAD_MAX_TILES=0 timeout -k 10 300 python tools/small_job_latency.py 2>/dev/null
echo "# CODLAD_EDGE_WIDE_MAX_TILES=0 (one-wave tile kernels / per-node kernels, as in round 3; node kernel and arguments as shipped)"
CODLAD_EDGE_WIDE_MAX_TILES=0 timeout -k 10 300 python tools/small_job_latency.py 2>/dev/null
echo "# HIP_FORCE_DEV_KERNARG=0 (kernel arguments in host memory, the runtime's default)"
HIP_FORCE_DEV_KERNARG=0 timeout -k 10 300 python tools/small_job_latency.py 2>/dev/null
echo "# size sweep, k x one 87-residue protein: as shipped"
timeout -k 10 300 python tools/small_job_latency.py --sweep --ks 1,2,4,8,12,16,20,24,32,40,48,64 2>/dev/null
echo "# size sweep: CODLAD_EDGE_WIDE_MAX_TILES=0"
CODLAD_EDGE_WIDE_MAX_TILES=0 timeout -k 10 300 python tools/small_job_latency.py --sweep --ks 1,2,4,8,12,16,20,24,32,40,48,64 2>/dev/null
} > $OUT
cat $OUT
KS=gpurun_out/${TAG}_small_job_kernel_stats.txt
: > $KS
for c in 0 2; do
  timeout -k 10 250 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_sj_c$c -o p --output-format csv -- python3 tools/small_job_latency.py --only $c --reps 2 > gpurun_out/${TAG}_sj_c$c.log 2>&1
  python - <<PY >> $KS
import csv, glob
f = glob.glob("gpurun_out/${TAG}_sj_c$c/**/p_kernel_stats.csv", recursive=True)[0]
print("# python tools/small_job_latency.py --only $c --reps 2 under rocprofv3 --kernel-trace --stats (3 x 100 DDPM steps)")
for r in list(csv.DictReader(open(f)))[:10]:
    print(f"{r['Name'][:72]:72s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e3:7.1f} us {float(r['Percentage']):5.1f}%")
PY
done
cat $KS
