"""Turns the rocprofv3 databases tools/profile.sh leaves under gpurun_out/prof_<tag>/ into the
text summaries committed under profiles/ (and profiles/traffic.json, read by bench.py)."""
import json
import os
import sqlite3
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join("gpurun_out", "prof_" + tag)
os.makedirs("profiles", exist_ok=True)
lines = ["rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline   (MI355X, %s)" % tag, ""]
c = sqlite3.connect(os.path.join(src, "kt", "kt_results.db"))
lines.append("%-110s %6s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    lines.append("%-110s %6d %14.3f %12.3f %7.2f" % (name[:110], calls, total, avg, pct))
rows = [(e - b) / 1000.0 for b, e in c.execute("select start, end from kernels where name like '%trace_kernel<double, false, 0, false, true, 4>%' order by start")]
if rows:
    import statistics
    lines += ["", "trace_kernel<double, false, 0, false, true, 4> (the steady-state variant: launch order with split units, 4 waves per SIMD) per launch, in launch order (us);",
              "the first frames of a view -- cost measurement, plain order -- run trace_kernel<double, false, 0, false, false, 4>, and two timed frames the 2-wave variant, listed above:",
              " ".join("%.0f" % r for r in rows[:40]) + (" ... (%d launches)" % len(rows) if len(rows) > 40 else ""),
              "median %.1f us, mean %.1f us, min %.1f us, max %.1f us over %d launches" % (
                  statistics.median(rows), statistics.mean(rows), min(rows), max(rows), len(rows))]
with open(os.path.join(src, "bench_under_rocprof.json")) as f:
    lines += ["", "bench.py JSON line of the same run:", f.read().strip()]
open(os.path.join("profiles", tag + "_kernel_trace_stats.txt"), "w").write("\n".join(lines) + "\n")

out = ["rocprofv3 --pmc passes (one counter group per run) over python3 bench.py --steps 6 --warmup 4 --min-seconds 0   (%s)" % tag,
       "FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them; per MI355X_MICROARCH.md FETCH_SIZE counts 64 B per",
       "128-B request for wide coalesced reads (x2 correction); this kernel's reads are scattered 8..16-B loads, for which",
       "the guide calls the absolute value uncalibrated, so both raw and x2 are given.", ""]
traffic = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_mem", "pmc_cache"):
    p = os.path.join(src, sub, "pmc_results.db")
    if not os.path.exists(p):
        continue
    c = sqlite3.connect(p)
    for k, cn, n, avg in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                   "where kernel_name like '%trace_kernel%' group by kernel_name, counter_name"):
        short = k[k.index("trace_kernel"):k.index("(rtc")]
        out.append("%-28s %-22s n=%-3d avg=%.1f" % (short, cn, n, avg))
        traffic.setdefault(short, {})[cn] = avg
# the steady-state kernel of the dragons frame is the variant for orders with split units (..., true>); the first frames of a
# view (cost measurement, and the one before the host knows the order holds splits) run the plain variant
k64 = next((v for k, v in traffic.items() if k.startswith("trace_kernel<double, false") and k.rstrip().endswith("true, %d>" % 4)), None) or \
      next((v for k, v in traffic.items() if k.startswith("trace_kernel<double, false")), {})
if "FETCH_SIZE" in k64 and "WRITE_SIZE" in k64:
    js = {"source": "profiles/%s_pmc.txt" % tag,
          "fetch_bytes_raw_fp64": k64["FETCH_SIZE"] * 1024, "write_bytes_fp64": k64["WRITE_SIZE"] * 1024,
          "hbm_bytes_per_launch_fp64": (2 * k64["FETCH_SIZE"] + k64["WRITE_SIZE"]) * 1024}
    json.dump(js, open(os.path.join("profiles", "traffic.json"), "w"), indent=1)
    out += ["", "fp64 trace kernel per launch: fetch raw %.1f MB (x2 = %.1f MB), write %.1f MB" % (
        js["fetch_bytes_raw_fp64"] / 1e6, 2 * js["fetch_bytes_raw_fp64"] / 1e6, js["write_bytes_fp64"] / 1e6)]
open(os.path.join("profiles", tag + "_pmc.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(lines[:8]))
print("\n".join(out[-12:]))
