"""Second tuning pass: the in-tree table holds, per conv geometry, the tile variant that is fastest when the launch runs
ALONE; in the training step the weight-gradient GEMMs share the CUs with the data-gradient chain, and the fastest
variant beside other work need not be the same.  This tool takes the plan of one workload, walks its geometries from
the most expensive down, tries every variant (and, for weight gradients, neighbouring split-K factors) IN the step and
keeps a change only when the whole step gets faster, re-measured A/B/A against the current choice.

    python tools/tune_in_step.py deconv 32 gpurun_out/step_tune.json [max_geometries] [seconds] [protect_archi|-] [floatx]

`protect_archi`: geometries that also occur in that workload's plan are left alone (their entry was chosen there).

Output: {"<dir>,<geometry>": [cfg, splits, step_gain_ms, 1]}; the trailing 1 marks "splits measured beside the other
stream: use as is" for engine.Plan.autotune."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from jpeg_detection_resnet_ssd_amd import _lib, engine, workloads  # noqa: E402

archi = sys.argv[1] if len(sys.argv) > 1 else "deconv"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
out_path = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/step_tune.json"
max_geoms = int(sys.argv[4]) if len(sys.argv) > 4 else 40
budget_s = float(sys.argv[5]) if len(sys.argv) > 5 else 420.0
protect = sys.argv[6] if len(sys.argv) > 6 and sys.argv[6] != "-" else None
floatx = sys.argv[7] if len(sys.argv) > 7 else "float32"
from jpeg_detection_resnet_ssd_amd.keras import backend as K  # noqa: E402
K.set_floatx(floatx)

lib = _lib.load()
ncfg = lib.dj_conv2d_tune_configs()
if archi.startswith("cls:"):      # classifier workloads (BASELINE configs 1-2): cls:deconv, cls:late_concat_rfa_thinner, cls:rgb
    import numpy as np
    from jpeg_detection_resnet_ssd_amd.keras.optimizers import SGD
    from jpeg_detection_resnet_ssd_amd.vgg_jpeg_keras.networks.resnet_dct import ResNet50Custom, ResNet50RGB
    K.clear_session()
    model = ResNet50RGB(weights=None) if archi == "cls:rgb" else ResNet50Custom(weights=None, archi=archi[4:])
    model.compile(optimizer=SGD(lr=0.1, momentum=0.9, decay=1e-4, nesterov=True), loss="categorical_crossentropy")
    rng = np.random.default_rng(0)
    x = [rng.normal(0, 30, (B,) + tuple(int(d) for d in t.shape[1:])).astype(np.float32) for t in model.inputs]
    x = x if len(x) > 1 else x[0]
    y = np.eye(1000, dtype=np.float32)[rng.integers(0, 1000, B)]
else:
    model, sizes = workloads.build_ssd(archi)
    x, y = workloads.synthetic_batch(archi, sizes, B, fast=True)
model.optimizer.lr = 0.0     # hundreds of steps on one batch: keep the weights (and the mining workload) where they are
plan = model._plan(B, True, True)
model._upload(plan, x, y)
names = [n for n, _ in _lib.ConvDesc._fields_][:15]


def step_ms(n=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        model.run_train_step(plan)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


geoms = {}
for direction, desc, _ in plan.conv_calls:
    key = (direction,) + tuple(getattr(desc, n) for n in names)
    g = geoms.setdefault(key, [direction, desc, 0])
    g[2] += 1
cur = {}
for key, (direction, desc, count) in geoms.items():
    ms, cfg, sp = engine._TUNED[key]
    known = engine._tune_db().get(",".join(str(int(v)) for v in key), ())
    if (direction & 3) == 2 and plan.side_stream is not None and not (len(known) > 3 and known[3]):
        sp = max(1, (sp + 1) // 2)      # what Plan.autotune registered for a fastest-alone entry
    cur[key] = (cfg, sp, ms * count)
# input gradients that take BatchNormalization backward statistics never split (their launch ignores a registered factor)
unsplit = {(d,) + tuple(getattr(desc, n) for n in names) for d, desc, fn in plan.conv_calls if getattr(fn, "no_split", False)}
protected = set()
if protect:
    other, _ = workloads.build_ssd(protect)
    for direction, desc, _ in other._plan(B, True, True).conv_calls:
        protected.add((direction,) + tuple(getattr(desc, n) for n in names))
    del other
order = [k for k in sorted(geoms, key=lambda k: -cur[k][2]) if k not in protected][:max_geoms]


def apply(key, cfg, sp):
    _lib.check(lib.dj_conv2d_tune_set(geoms[key][0], geoms[key][1], int(cfg), int(sp)), "tune_set")


for _ in range(3):
    step_ms(10)
t_start = step_ms(30)
print("step before: %.3f ms" % t_start, flush=True)
changes, wall0 = {}, time.time()
for key in order:
    if time.time() - wall0 > budget_s:
        print("time budget reached", flush=True)
        break
    direction, desc, count = geoms[key]
    cfg0, sp0, _ = cur[key]
    cands = [(c, sp0) for c in range(ncfg) if c != cfg0]
    base_a = step_ms(10)
    best = (None, base_a)
    for c, sp in cands:
        apply(key, c, sp)
        t = step_ms(6)
        if t < best[1]:
            best = ((c, sp), t)
    # split-K factors around the current one at the best variant so far (weight gradients), or the small factors a
    # forward / data-gradient GEMM can use (a forward split-K launch is followed by a ReLU pass of its own: only the
    # step time sees that).  Forward launches that take BatchNormalization statistics (dir 4) cannot split.
    c = best[0][0] if best[0] is not None else cfg0
    if (direction & 3) == 2:
        sp_opts = {max(1, sp0 // 2), sp0 * 2}
    elif direction in (0, 1) and key not in unsplit:
        sp_opts = {1, 2, 4, 8}
    else:
        sp_opts = set()
    for sp in sorted(sp_opts - {sp0}):
        apply(key, c, sp)
        t = step_ms(6)
        if t < best[1]:
            best = ((c, sp), t)
    apply(key, cfg0, sp0)
    if best[0] is None:
        continue
    # confirm A/B/A/B with longer runs
    a1 = step_ms(15)
    apply(key, *best[0])
    b1 = step_ms(15)
    apply(key, cfg0, sp0)
    a2 = step_ms(15)
    apply(key, *best[0])
    b2 = step_ms(15)
    gain = (a1 + a2) / 2 - (b1 + b2) / 2
    if gain > 0.03 and b1 < a1 and b2 < a2:   # > 0.1 % of the step, both times
        changes[",".join(str(int(v)) for v in key)] = [int(best[0][0]), int(best[0][1]), round(gain, 4), 1]
        cur[key] = (best[0][0], best[0][1], cur[key][2])
        print("keep  dir %d %s x%d: cfg %d splits %d -> cfg %d splits %d, step -%.3f ms" %
              (direction, key[1:9], count, cfg0, sp0, best[0][0], best[0][1], gain), flush=True)
    else:
        apply(key, cfg0, sp0)
        print("drop  dir %d %s: candidate cfg %d splits %d (%.3f ms)" % (direction, key[1:9], best[0][0], best[0][1], gain),
              flush=True)
t_end = step_ms(30)
print("step after: %.3f ms (before %.3f)" % (t_end, t_start), flush=True)
os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
with open(out_path, "w") as f:
    json.dump({"workload": "%s B=%d" % (archi, B), "step_before_ms": t_start, "step_after_ms": t_end, "entries": changes}, f,
              indent=0, sort_keys=True)
