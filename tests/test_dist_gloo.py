"""CPU, world_size 2 over gloo: the data-parallel gradient exchange (bucketing over one flat buffer, summed
all-reduce, 1/world folded into the optimizer) and metric averaging."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_plan_buckets_partition_and_order():
    from jpeg_detection_resnet_ssd_amd.dist import plan_buckets
    so = {"a": (0, 100), "b": (100, 60), "c": (160, 40), "d": (200, 1000), "e": (1200, 8)}
    ready = {"e": 3, "d": 5, "c": 9, "b": 9, "a": 14}          # backward finishes the last layers first
    buckets = plan_buckets(ready, so, bucket_bytes=4 * 200)
    covered = sorted(r for _, rs in buckets for r in rs)
    merged = []
    for lo, hi in covered:
        if merged and merged[-1][1] == lo:
            merged[-1] = (merged[-1][0], hi)
        else:
            merged.append((lo, hi))
    assert merged == [(0, 1208)]                                # every gradient exactly once
    assert [idx for idx, _ in buckets] == sorted(idx for idx, _ in buckets)
    assert buckets[0] == (5, [(200, 1208)])                     # e+d fill the first bucket, contiguous -> one range
    assert buckets[1][0] == 14 and buckets[1][1] == [(0, 200)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from jpeg_detection_resnet_ssd_amd import dist as dj
    r, w, _ = dj.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(5000, generator=g)
    mine = flat.clone()
    ex = dj.GradientExchange(flat, world)
    so = {"k%d" % i: (i * 500, 500) for i in range(10)}
    ready = {"k%d" % i: 20 - 2 * i for i in range(10)}
    for _, ranges in dj.plan_buckets(ready, so, bucket_bytes=4 * 1200):
        ex.launch(ranges)
    scale = ex.finish()
    # weights replicated: broadcast then identical SGD step on every rank
    wts = torch.full((5000,), float(rank))
    dist.broadcast(wts, src=0)
    wts -= 0.1 * flat * scale

    class M:
        pass
    m = M()
    dp = dj.DataParallel.__new__(dj.DataParallel)
    dp.world, dp.rank = world, rank
    logs = dp.average_metrics({"loss": float(rank + 1), "name": "x"})
    out.put((rank, mine.numpy(), flat.numpy(), scale, wts.numpy(), logs["loss"]))
    dist.destroy_process_group()


def test_gradient_exchange_world2_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, g0, s0, sc0, w0, l0), (_, g1, s1, sc1, w1, l1) = res
    np.testing.assert_allclose(s0, g0 + g1, rtol=1e-6)          # summed over ranks ...
    np.testing.assert_array_equal(s0, s1)                        # ... bit-identical on every rank
    assert sc0 == sc1 == 0.5                                     # ... averaged by the optimizer's grad_scale
    np.testing.assert_array_equal(w0, w1)                        # replicas stay identical after the step
    assert l0 == l1 == 1.5


def _plan_worker(rank, world, port, out):
    """Structure-only lowering of the real SSD300 training plan on the host (side-stream predictor heads, input
    gradients fused with the BatchNormalization backward statistics, the deferred bias-gradient launch), then the real
    DataParallel.attach on it and the bucketed exchange over gloo on the flat gradient buffer."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), DJ_AUTOTUNE="table")
    torch.set_num_threads(2)
    from jpeg_detection_resnet_ssd_amd import dist as dj
    from jpeg_detection_resnet_ssd_amd import engine, workloads
    dj.init_from_env(backend="gloo")
    model, _ = workloads.build_ssd("deconv", weight_seed=42 + rank)
    model._ensure_params(device=torch.device("cpu"))
    plan = model._plan(2, True, True)
    n_bwd = len(plan.bwd)
    colsum_at = [i for i, f in enumerate(plan.bwd) if getattr(f, "_keep", None) is not None
                 and f.__qualname__.startswith("colsum_multi")]
    fused_dgrads = sum(1 for d, _, _ in plan.conv_calls if d == 9)
    side_heads = sum(1 for v in plan.fused_outputs.values() if getattr(v, "side_done", None) is not None)
    before = list(plan.bwd)
    dp = dj.DataParallel(model, bucket_mb=16)       # the plan is cached already: the constructor splices the exchange in
    assert plan._dp_attached is dp
    known = {id(f) for f in before}
    inserted = [(i, f) for i, f in enumerate(plan.bwd) if id(f) not in known]
    launched = []
    orig = dp.exchange.launch
    dp.exchange.launch = lambda ranges: (launched.append(list(ranges)), orig(ranges))[1]
    # gradient-ready bookkeeping: every trainable weight has an index, none beyond the end of the (original) list, and
    # the bias gradients that one deferred dj_colsum_multi launch produces are ready only AFTER that launch
    offs = model._store["offsets"]
    ready = {w.key: plan.grad_ready.get(w.key) for w in model.weight_specs if w.trainable}
    deferred_keys = [k for k, v in ready.items() if colsum_at and v is not None and v == colsum_at[-1] + 1]
    g = torch.Generator().manual_seed(7 + rank)
    mine = torch.randn(model.flat_gradients.numel(), generator=g)
    model.flat_gradients.copy_(mine)
    plan.side_enabled = False        # (host: a collective is issued where it stands in the list)
    for _, f in inserted:
        f()
    scale = dp.finish_gradients(plan)
    out.put(dict(rank=rank, n_bwd=n_bwd, colsum_at=colsum_at, fused_dgrads=fused_dgrads, side_heads=side_heads,
                 inserted_at=[i for i, _ in inserted], launched=launched, n_train=int(model._store["n_train"]),
                 missing_ready=[k for k, v in ready.items() if v is None], max_ready=max(v for v in ready.values() if v is not None),
                 n_deferred=len(deferred_keys), mine=mine.numpy(), summed=model.flat_gradients.numpy().copy(), scale=scale,
                 n_buckets=dp.n_buckets))
    dist.destroy_process_group()


def test_buckets_of_the_real_ssd_plan_world2_gloo():
    """the plan as the GPU runs it -- predictor heads lowered for the side stream, input gradients that
    take the BatchNormalization backward statistics (tuner direction 9), the bias gradients of the heads deferred into one
    dj_colsum_multi launch at the end of backward -- must still give buckets that tile the flat gradient buffer exactly
    once, placed after the launches that finish their gradients; two ranks then exchange through them."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_plan_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=600) for _ in procs], key=lambda d: d["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    a, b = res
    for r in res:
        assert r["fused_dgrads"] >= 10 and r["side_heads"] >= 6, (r["fused_dgrads"], r["side_heads"])
        assert len(r["colsum_at"]) == 1 and r["colsum_at"][0] == r["n_bwd"] - 1       # the deferred launch closes the list
        assert not r["missing_ready"] and r["max_ready"] == r["n_bwd"] and r["n_deferred"] >= 10
        assert r["n_buckets"] >= 4 and len(r["launched"]) == r["n_buckets"] == len(r["inserted_at"])
        assert r["inserted_at"] == sorted(r["inserted_at"]) and r["inserted_at"][-1] == r["n_bwd"] + r["n_buckets"] - 1
        pos = 0
        for lo, hi in sorted(rg for ranges in r["launched"] for rg in ranges):
            assert lo == pos and hi > lo, (lo, hi, pos)
            pos = hi
        assert pos == r["n_train"]
    assert a["launched"] == b["launched"] and a["inserted_at"] == b["inserted_at"]     # replicas issue the same collectives
    np.testing.assert_allclose(a["summed"], a["mine"] + b["mine"], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(a["summed"], b["summed"])
    assert a["scale"] == b["scale"] == 0.5
