"""Render a rocprofv3 --kernel-trace --stats run (kernel_stats.csv + kernel_trace.csv) as a markdown summary."""
import csv, sys, subprocess, json, os
stats, trace, benchjson, title, cmd = sys.argv[1:6]
rows = list(csv.DictReader(open(stats)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
ig = [r for r in rows if "dj_igemm" in r["Name"]]
igc = sum(int(r["Calls"]) for r in ig); igt = sum(int(r["TotalDurationNs"]) for r in ig)
b = json.loads(open(benchjson).read().strip().splitlines()[-1])
print("# %s\n" % title)
print("Command: `%s`\n" % cmd)
print("bench line of this run: %.1f img/s, %.2f ms/step; dominant_kernel (HIP events, serialized step): %d launches/step, "
      "avg %.1f us, %.1f TFLOP/s.\n" % (b["value"], b["ms_per_step"], b["roofline"]["dominant_kernel"]["launches_per_step"],
                                       b["roofline"]["dominant_kernel"]["avg_launch_us"], b["roofline"]["dominant_kernel"]["achieved"]))
print("Implicit-GEMM family in this trace: %d calls, %.1f ms total, **average %.1f us per launch** "
      "(%.1f %% of all kernel time %.1f ms).\n" % (igc, igt / 1e6, igt / igc / 1e3, 100.0 * igt / tot, tot / 1e6))
print("## rocprofv3 --stats (kernel_stats.csv), top 30 by total time\n")
print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"]))[:30]:
    print("| `%s` | %s | %.2f | %.1f | %.2f |" % (r["Name"].split("(")[0].replace("void ", "")[:80], r["Calls"], int(r["TotalDurationNs"]) / 1e6,
                                              float(r["AverageNs"]) / 1e3, 100.0 * int(r["TotalDurationNs"]) / tot))
print("\n## Timeline of the last 5 steps (tools/timeline.py)\n\n```")
print(subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "timeline.py"), trace, "5"], capture_output=True, text=True).stdout.strip().split("\n   ")[0])
print("```")
