# round 2, fourth GPU run: all GPU tests, pipelined incremental bench vs stream-ordered, full bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
for extra in "" "--no-pipeline"; do
  python bench.py --mode incremental --no-cpu-baseline $extra > $O/bench_inc$extra.json 2> $O/err.txt; python - <<PY
import json
j=json.loads(open("$O/bench_inc$extra.json").read().strip().splitlines()[-1])
print("incremental $extra: %.3e  %.1f us/ply  update+refresh %.1f us  sort+mlp %.1f us  exact %s" % (j["value"], j["ms_per_step"]*1e3, j["roofline"]["update_kernel_ms"]*1e3, j["roofline"]["sort_mlp_ms"]*1e3, j["config"]["bit_exact_vs_full_refresh"]))
PY
done
for b in 1024 4096 16384 262144; do python bench.py --mode incremental --no-cpu-baseline --batch $b --steps 100 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('games $b: %.3e %.1f us/ply' % (j['value'], j['ms_per_step']*1e3), j['config']['bit_exact_vs_full_refresh'])"; done
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2>> $O/err.txt; python - <<PY
import json
j=json.loads(open("$O/bench_driver_args.json").read().strip().splitlines()[-1])
print("full (driver args): %.4e  %.4f ms/step ft %.4f  exact %s  wide %.4e  cpu %.3e  settle %d" % (j["value"], j["ms_per_step"], j["config"]["kernel_ms"]["ft"], j["bit_exact_sample"], j["wide_psq_rows"]["value"], j["cpu_baseline"]["value"], j["config"]["settle_steps"]))
print("roofline frac", j["roofline"]["frac"], "achieved", j["roofline"]["achieved"])
PY
python bench.py --no-cpu-baseline > $O/bench_default.json 2>> $O/err.txt; python -c "
import json
j=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print('full (default 200 steps): %.4e %.4f ms/step' % (j['value'], j['ms_per_step']))"
tail -3 $O/err.txt
