"""Per-layer HBM traffic of the implicit-GEMM launches of one training step: joins the per-dispatch FETCH_SIZE / WRITE_SIZE
counters of two rocprofv3 --pmc passes over bench.py with the ordered call list tools/profile_layers.py writes
(PROFILE_LAYERS_JSON), and prints measured / algorithmic bytes per geometry.  Counter corrections as in tools/pmc_traffic.py.
    python tools/pmc_layers.py fetch/counter_collection.csv write/counter_collection.csv calls.json"""
import csv, sys, json, collections
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))


def last_step_igemm(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    sgd = [i for i, r in enumerate(rows) if "dj_sgd" in r["Kernel_Name"]]
    loss = [i for i, r in enumerate(rows) if "dj_ssd_loss_bwd" in r["Kernel_Name"]]
    hi = sgd[-1]
    lo = max([i for i in sgd if i < loss[-1]] or [-1]) + 1
    # the SGD update is two launches: skip back over the previous step's second one
    return [r for r in rows[lo:hi + 1] if "dj_igemm" in r["Kernel_Name"]]


fetch = last_step_igemm(sys.argv[1], "FETCH_SIZE")
write = last_step_igemm(sys.argv[2], "WRITE_SIZE")
calls = json.load(open(sys.argv[3]))
assert len(fetch) == len(write) == len(calls), (len(fetch), len(write), len(calls))
agg = collections.OrderedDict()
for c, f, w in zip(calls, fetch, write):
    k = (c["op"], tuple(c["key"]))
    a = agg.setdefault(k, [0, 0.0, 0.0, 0.0, 0.0, f["Kernel_Name"][:48]])
    a[0] += 1
    a[1] += float(f["Counter_Value"]) * 1024 * 2
    a[2] += float(w["Counter_Value"]) * 1024
    a[3] += c["bytes"]
    a[4] += c["ms"]
tot_m = sum(a[1] + a[2] for a in agg.values())
tot_a = sum(a[3] for a in agg.values())
print("family: measured %.2f GB, algorithmic %.2f GB, ratio %.2f" % (tot_m / 1e9, tot_a / 1e9, tot_m / tot_a))
print("%-18s %-42s %3s %9s %9s %9s %6s %8s %7s" % ("op", "B,H,W,Cin,OH,Cout,k,s,d", "n", "fetch MB", "write MB", "alg MB", "ratio", "ms", "TB/s"))
for (op, key), a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2] - kv[1][3])):
    print("%-18s %-42s %3d %9.1f %9.1f %9.1f %6.2f %8.3f %7.2f" % (op, str(key), a[0], a[1] / 1e6, a[2] / 1e6, a[3] / 1e6,
                                                              (a[1] + a[2]) / a[3], a[4], (a[1] + a[2]) / a[4] / 1e9))
