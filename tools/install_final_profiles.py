"""Copy what tools/gpu_final.sh wrote under gpurun_out/final_<tag>/ into profiles/<tag>_* under the names the documents and
bench.py use, and print the headline numbers.   usage: python tools/install_final_profiles.py [tag]"""
import glob
import json
import os
import shutil
import sys

TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", f"final_{TAG}")
DST = os.path.join(ROOT, "profiles")


def last(name):
    return json.loads(open(os.path.join(SRC, name)).read().strip().splitlines()[-1])


secondary = {"secondary_gpu_measure_py": last("secondary.json")}
for key, name in [("selfplay_4096", "selfplay_4096.json"), ("selfplay_4096_long", "selfplay_4096_long.json"),
                  ("selfplay_16384", "selfplay_16384.json"),
                  ("selfplay_1024", "selfplay_1024.json"), ("config3_replay", "config3_replay.json"),
                  ("movegen_rate", "movegen_rate.json"), ("selfplay_search_64_nodes", "selfplay_search_64_nodes.json"),
                  ("selfplay_search_1000_nodes", "selfplay_search_1000_nodes.json")]:
    secondary[key] = last(name)
secondary["latency_txt"] = open(os.path.join(SRC, "latency.txt")).read()
json.dump(secondary, open(os.path.join(DST, f"{TAG}_secondary_measurements.json"), "w"), indent=1)
names = {
    "bench_n1.json": "bench_n1.json", "bench_n1_driver_args.json": "bench_n1_driver_args_steps20_warmup5.json",
    "bench_n1_strict_stream_order.json": "bench_n1_strict_stream_order.json",
    "bench_n1_device_positions.json": "bench_n1_device_positions.json", "bench_incremental_n1.json": "bench_incremental_n1.json",
    "bench_incremental_n1_strict_stream_order.json": "bench_incremental_n1_strict_stream_order.json",
    "bench_incremental_n1_round1_kernel.json": "bench_incremental_n1_round1_kernel.json",
    "bench_incremental_n1_262144.json": "bench_incremental_n1_262144_games.json", "bench_n1_batch4M.json": "bench_n1_batch4M.json",
    "pmc_full_refresh.json": "pmc_full_refresh.json", "pmc_incremental.json": "pmc_incremental.json",
    "rocprofv3_summary_full_refresh.txt": "rocprofv3_summary_full_refresh.txt",
    "rocprofv3_summary_incremental.txt": "rocprofv3_summary_incremental.txt",
    "rocprofv3_kernel_stats_default_cmd.txt": "rocprofv3_kernel_stats_default_cmd.txt",
    "rocprofv3_incremental_kernel_stats.txt": "rocprofv3_incremental_kernel_stats.txt",
    "rocprofv3_kernel_stats_headline_only.txt": "rocprofv3_kernel_stats_headline_only.txt",
    "reference_differential.json": "reference_differential.json", "reference_trace_differential.json": "reference_trace_differential.json",
    "pmc_selfplay_4096_seats.txt": "pmc_selfplay_4096_seats.txt",
    "replay_segment_ab.txt": "ab_replay_segment_length.txt",
    "raweval_walk.txt": "raweval_walk_evaluate_by_pending_plies.txt", "selfplay_gpu_busy.txt": "selfplay_gpu_busy_4096_seats.txt",
    "bench_n2_two_ranks_sharing_one_gpu.json": "bench_n2_two_ranks_sharing_one_gpu.json",
    "bench_n1_config5_hbm_filling.json": "bench_n1_config5_hbm_filling.json",
    "kstats_sliced_pipeline_stream_ordered.txt": "kstats_sliced_pipeline_stream_ordered_final_binary.txt",
    "pmc_sliced_pipeline.txt": "pmc_sliced_pipeline.txt",
    "kstats_one_kernel_path_stream_ordered.txt": "kstats_one_kernel_path_stream_ordered.txt",
    "timeline_pipelined_steps.txt": "timeline_pipelined_steps.txt",
    "sliced_pipeline_crossover.txt": "sliced_pipeline_crossover.txt",
    "reference_differential_one_kernel_path.json": "reference_differential_one_kernel_path.json",
    "reference_engine_on_gpu_evaluator.txt": "reference_engine_on_gpu_evaluator.txt",
}
for src, dst in names.items():
    target = os.path.join(DST, f"{TAG}_{dst}")
    if not os.path.exists(os.path.join(SRC, src)):
        print("missing:", src)
        continue
    if src.startswith("bench") and src.endswith(".json") and "config5" not in src:
        open(target, "w").write(open(os.path.join(SRC, src)).read().strip().splitlines()[-1] + "\n")
    else:
        shutil.copy(os.path.join(SRC, src), target)
for f in sorted(glob.glob(os.path.join(DST, f"{TAG}_bench*.json"))):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except ValueError:  # (a pretty-printed record, not a bench line)
        continue
    r = j.get("roofline", {})
    print(os.path.basename(f), "%.4e" % j["value"], "ms/step %.4f" % j["ms_per_step"], "frac %.3f" % r.get("frac", 0),
          "wide", (j.get("wide_psq_rows") or {}).get("value"), "cpu", (j.get("cpu_baseline") or {}).get("value"),
          "kernel_ms", j["config"].get("kernel_ms"), "update", r.get("update_kernel_ms"), "traffic/hbm", r.get("traffic_over_hbm_peak"))
print("selfplay search 64 / 1000 nodes", *(secondary[k]["value"] for k in ("selfplay_search_64_nodes", "selfplay_search_1000_nodes")))
print("selfplay", secondary["selfplay_4096"]["value"], secondary["selfplay_4096_long"]["value"], secondary["selfplay_16384"]["value"], secondary["selfplay_1024"]["value"],
      "replay ms", secondary["config3_replay"]["native_device_ms"])
print(secondary["latency_txt"])
print(json.dumps(secondary["secondary_gpu_measure_py"]))
