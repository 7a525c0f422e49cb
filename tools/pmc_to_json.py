#!/usr/bin/env python3
"""rocprofv3 output directory (tools/gpu_profile.sh / gpu_pmc_inc.sh: stats/ + pmc_*/ sqlite files) -> one JSON with the
per-kernel mean counters, the form bench.py reads from profiles/ (load_pmc):
    python tools/pmc_to_json.py gpurun_out/prof_r02 profiles/r02_pmc_full_refresh.json --batch 65536 --preset tame
"""
import argparse
import glob
import json
import sqlite3
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--preset", default="tame")
    ap.add_argument("--net", default=None)
    ap.add_argument("--command", default="")
    ap.add_argument("--valu-cycles", type=float, default=4.0,
                    help="SIMD cycles per wave64 VALU instruction (tools/probes/valu_rate_probe: ~4 for v_perm_b32 / "
                         "v_add3_u32 / v_pk_add_u16 / shifts / multiplies, ~2.5 for v_add_u32 / v_and_b32)")
    args = ap.parse_args()
    kernels = {}
    for f in glob.glob(args.src + "/stats/*.db"):
        c = sqlite3.connect(f).cursor()
        for name, calls, total, avg, pct in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
            if "rocclr" in name or "at::native" in name:
                continue
            kernels.setdefault(name, {"counters": {}}).update(calls=calls, avg_us=avg / 1e3 if avg > 1e4 else avg, pct=pct)
    for f in sorted(glob.glob(args.src + "/pmc_*/*.db")):
        c = sqlite3.connect(f).cursor()
        for name, counter, mean, n in c.execute(
                "select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"):
            if "rocclr" in name or "at::native" in name:
                continue
            k = kernels.setdefault(name, {"counters": {}})
            k["counters"][counter] = mean
            k["pmc_dispatches"] = n
    out = {"source": f"rocprofv3 --kernel-trace --stats and --pmc passes (one counter group per pass) of: {args.command}",
           "config": {"batch": args.batch, "preset": args.preset, "net": args.net},
           "valu_cycles_per_wave_instr": args.valu_cycles,
           "units": "counters: mean per dispatch; FETCH_SIZE / WRITE_SIZE in KiB (FETCH_SIZE x2 on gfx950, MI355X_MICROARCH.md); "
                    "avg_us: kernel-trace average duration",
           "kernels": kernels}
    json.dump(out, open(args.dst, "w"), indent=1)
    print(f"{args.dst}: {len(kernels)} kernels", file=sys.stderr)


if __name__ == "__main__":
    main()
