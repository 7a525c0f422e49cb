# PMC passes of the column-sliced pipeline's kernels (stream-ordered calls): bash tools/gpu_pmc_ftx.sh <tag>
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_ftx_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SPX_OPTIONS=ftx=1
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o c -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-settle --no-wide --no-secondary --no-pipeline > $OUT/g$i.log 2>&1 || echo "FAILED: $grp"
done
python3 - <<PY > $REPO/gpurun_out/pmc_ftx_$TAG.txt
import glob, sqlite3
print("# rocprofv3 --pmc <group> --kernel-trace (one counter group per pass) of: SPX_OPTIONS=ftx=1 bench.py --steps 20 --warmup 5 --no-pipeline --no-settle --no-wide --no-secondary")
print("# mean per dispatch; FETCH_SIZE / WRITE_SIZE in KiB (FETCH_SIZE x 2 on gfx950); 65 536 positions per step, tame net")
for f in sorted(glob.glob("$OUT/g*/*.db")):
    c = sqlite3.connect(f).cursor()
    for r in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%ftx%' or kernel_name like '%mlp%' group by kernel_name, counter_name"):
        print("%-44s %-32s %18.1f   n %d" % (r[0].replace("spx::", "").replace("(spx::FtxParams)", "")[:44], r[1], r[2], r[3]))
PY
cat $REPO/gpurun_out/pmc_ftx_$TAG.txt
rm -rf $OUT
