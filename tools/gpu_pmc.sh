# ad-hoc PMC groups: bash tools/gpu_pmc.sh <tag> "<grp1>" "<grp2>" ...   (kernel-trace only, one pass per group)
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o c -- python $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/g$i.log 2>&1 || echo "FAILED: $grp"
done
python3 - <<PY
import glob, sqlite3
for f in sorted(glob.glob("$OUT/g*/*.db")):
    c = sqlite3.connect(f).cursor()
    for r in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%ft_kernel%' group by kernel_name, counter_name"):
        print("%-28s %-34s %18.1f" % (r[0][:28], r[1], r[2]))
PY
