# Column-sliced gather probe: timings of every probe variant next to the product FT kernel,
# then L2 / fabric / L1 counters per probe kernel (one counter group per pass, kernel-trace only; the sliced variants share
# kernel names, so each gets its own passes).
#   bash tools/gpu_sliced_probe.sh <tag> [pmc variants, default "0 12 14 17"]   -> gpurun_out/sliced_<tag>/{timing.json, pmc.txt}
set -u
TAG=${1:-r04}
PMCV=${2:-"0 12 14 17"}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/sliced_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -n "${SKIP_TIMING:-}" ] || timeout 900 python $REPO/tools/gpu_gather_ceiling.py --rounds 5 --iters 50 ${TIMING_VARIANTS:+--variants=$TIMING_VARIANTS} --out $OUT/timing.json > $OUT/timing.log 2>&1 || echo "timing failed" >> $OUT/errors.log
for v in $PMCV; do
  i=0
  for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
             "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $grp --kernel-trace -d $OUT/pmc_v${v}_$i -o c -- python $REPO/tools/gpu_gather_ceiling.py --rounds 1 --iters 6 --variants=$v > $OUT/pmc_v${v}_$i.log 2>&1 || echo "pmc group failed: v$v $grp" >> $OUT/errors.log
  done
done
python3 - <<PY > $OUT/pmc.txt 2>&1
import glob, sqlite3, json
names = {r["variant"]: r["name"] for r in json.load(open("$OUT/timing.json"))["variants"]} if glob.glob("$OUT/timing.json") else {}
for v in "$PMCV".split():
    print("== variant", v, names.get(int(v), ""))
    for f in sorted(glob.glob("$OUT/pmc_v%s_*/*.db" % v)):
        c = sqlite3.connect(f).cursor()
        for r in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%gather_kernel%' group by kernel_name, counter_name"):
            print("   %-66s %-30s mean/dispatch %16.1f  n %d" % (r[0][:66], r[1], r[2], r[3]))
PY
cat $OUT/pmc.txt
rm -rf $OUT/pmc_v*/
