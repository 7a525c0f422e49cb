# Profiling recipe (runs on the GPU box via gpurun). Outputs under gpurun_out/prof_<tag>/; copy summaries to profiles/.
# usage: bash tools/gpu_profile.sh <tag> [bench args]
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 50 --warmup 5 --no-cpu-baseline $*"
# 1) kernel trace + stats (no counters)
rocprofv3 --kernel-trace --stats -d $OUT/stats -o t -- $BENCH > $OUT/stats.log 2>&1
# 2) counters, one group per pass (TCC: FETCH_SIZE costs 3 of 4 slots, WRITE_SIZE 2)
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/pmc_$name -o c -- $BENCH > $OUT/pmc_$name.log 2>&1 || echo "pmc group failed: $grp" >> $OUT/errors.log
done
ls -R $OUT | head -80
# compact summaries
python3 - <<PY
import csv, glob, os, collections
out = "$OUT"
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats", f)
    print(open(f).read()[:3000])
for f in sorted(glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        k = (row.get("Kernel_Name", "")[:40], row.get("Counter_Name", ""))
        agg[k][0] += float(row.get("Counter_Value", 0) or 0); agg[k][1] += 1
    print("== pmc", os.path.relpath(f, out))
    for (kn, cn), (s, n) in sorted(agg.items()):
        print(f"{kn:42s} {cn:32s} mean/dispatch {s/n:16.1f}  dispatches {n}")
PY
