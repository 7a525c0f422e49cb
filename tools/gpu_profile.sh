# Profiling recipe (runs on the GPU box via gpurun). Outputs under gpurun_out/prof_<tag>/; copy summaries to profiles/.
# usage: bash tools/gpu_profile.sh <tag> [bench args]
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-settle --no-wide --no-secondary $*"
# 1) kernel trace + stats (no counters)
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o t -- $BENCH > $OUT/stats.log 2>&1
# 2) counters, one group per pass (TCC: FETCH_SIZE costs 3 of 4 slots, WRITE_SIZE 2), kernel-trace only
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $OUT/pmc_$i -o c -- $BENCH > $OUT/pmc_$i.log 2>&1 || echo "pmc group failed: $grp" >> $OUT/errors.log
done
python3 - <<PY > $OUT/summary.txt 2>&1
import glob, sqlite3
out = "$OUT"
for f in glob.glob(out + "/stats/*.db"):
    c = sqlite3.connect(f).cursor()
    print("== kernel stats (rocprofv3 --kernel-trace --stats): name, calls, total, average, pct of GPU time")
    for r in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        print("   %-48s calls %5d total %10.1f us avg %9.2f us  %5.1f%%" % (r[0][:48], r[1], r[2], r[3], r[4]))
    print("== per-kernel launch config")
    for r in c.execute("select name, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, count(*) from kernels group by name, grid_x"):
        print("  ", r)
for f in sorted(glob.glob(out + "/pmc_*/*.db")):
    c = sqlite3.connect(f).cursor()
    print("== pmc", f.split("/")[-2])
    for r in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"):
        if "rocclr" in r[0]: continue
        print("   %-34s %-30s mean/dispatch %18.1f  dispatches %d" % (r[0][:34], r[1], r[2], r[3]))
PY
cat $OUT/summary.txt
python3 $REPO/tools/pmc_to_json.py $OUT $OUT/pmc.json --command "$BENCH"
