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
