"""Per-launch view of the NON-GEMM kernels of one training step from a rocprofv3 kernel trace (one stream, so a
duration is the kernel alone): grouped by kernel name and grid size, sorted by time per step.
    python tools/nonconv_breakdown.py <kernel_trace.csv> [steps]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
            for r in rows)
marks = [e for e in ev if "dj_ssd_loss_bwd" in e[2]]
t1, t0 = marks[-1][1], marks[-1 - nsteps][1]
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
tot = sum(e[1] - e[0] for e in win) / 1e6 / nsteps
gemm = sum(e[1] - e[0] for e in win if "igemm" in e[2]) / 1e6 / nsteps
print("kernel time per step %.3f ms, implicit-GEMM %.3f ms, other %.3f ms (%d launches/step)"
      % (tot, gemm, tot - gemm, sum(1 for e in win if "igemm" not in e[2]) / nsteps))
byname = collections.defaultdict(lambda: [0, 0])
bygrid = collections.defaultdict(lambda: [0, 0])
for s, e, n, g in win:
    if "igemm" in n: continue
    n = n.split("(")[0][:70]
    byname[n][0] += 1; byname[n][1] += e - s
    bygrid[(n, g)][0] += 1; bygrid[(n, g)][1] += e - s
for n, (c, t) in sorted(byname.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%8.3f ms/step %6.1f calls/step  avg %6.1f us  %s" % (t / 1e6 / nsteps, c / nsteps, t / c / 1e3, n))
print("--- by grid (workgroups)")
for (n, g), (c, t) in sorted(bygrid.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%8.3f ms/step %6.1f calls/step  avg %6.1f us  wg %6d  %s" % (t / 1e6 / nsteps, c / nsteps, t / c / 1e3, g, n))
