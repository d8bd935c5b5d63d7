"""Timeline statistics of a rocprofv3 kernel trace: per-queue busy time, idle gaps, overlap."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows]
ev.sort()
# last nsteps steps: locate sgd kernel ends
sgd = [e for e in ev if "dj_ssd_loss_bwd" in e[3]]
t1 = sgd[-1][1]
t0 = sgd[-1 - nsteps][1]
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
wall = (t1 - t0) / 1e6
print("window %.3f ms for %d steps = %.3f ms/step, %d kernels/step" % (wall, nsteps, wall / nsteps, len(win) / nsteps))
byq = collections.defaultdict(list)
for e in win: byq[e[2]].append(e)
for q, l in byq.items():
    busy = sum(e[1] - e[0] for e in l) / 1e6
    print("queue %s: %d kernels/step, busy %.3f ms/step" % (q, len(l) / nsteps, busy / nsteps))
# union busy
pts = sorted((e[0], e[1]) for e in win)
u = 0; cs, ce = pts[0]
for s, e in pts[1:]:
    if s > ce: u += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
u += ce - cs
print("GPU busy (union) %.3f ms/step, idle %.3f ms/step" % (u / 1e6 / nsteps, (t1 - t0 - u) / 1e6 / nsteps))
ov = sum(e[1] - e[0] for e in win) - u
print("overlapped kernel time %.3f ms/step" % (ov / 1e6 / nsteps))
# gap histogram on union
gaps = []
cs, ce = pts[0]
for s, e in pts[1:]:
    if s > ce: gaps.append(s - ce); cs, ce = s, e
    else: ce = max(ce, e)
gaps.sort()
import statistics
print("gaps: n/step %.0f, median %.2f us, mean %.2f us, p95 %.2f us, max %.1f us" % (len(gaps) / nsteps, statistics.median(gaps) / 1e3, sum(gaps) / len(gaps) / 1e3, gaps[int(.95 * len(gaps))] / 1e3, gaps[-1] / 1e3))
agg = collections.defaultdict(lambda: [0, 0])
for e in win:
    n = e[3].split("(")[0][:90]
    agg[n][0] += 1; agg[n][1] += e[1] - e[0]
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%8.3f ms/step %6.1f calls/step  %s" % (t / 1e6 / nsteps, c / nsteps, n))
# ---- exclusive / shared time of the two queues inside the window
qs = sorted(byq)
if len(qs) == 2:
    evs = []
    for qi, q in enumerate(qs):
        for e in byq[q]:
            evs.append((e[0], 1, qi)); evs.append((e[1], -1, qi))
    evs.sort()
    act = [0, 0]; last = evs[0][0]; tot = {"none": 0, "only0": 0, "only1": 0, "both": 0}
    for t, d, qi in evs:
        key = "both" if act[0] and act[1] else ("only0" if act[0] else ("only1" if act[1] else "none"))
        tot[key] += t - last; last = t
        act[qi] += d
    print("per step: main only %.2f ms, side only %.2f ms, both %.2f ms, none %.2f ms" % tuple(tot[k] / 1e6 / nsteps for k in ("only0", "only1", "both", "none")))
    # last kernel end per queue relative to each step's marker
