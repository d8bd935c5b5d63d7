"""GPU (one card, two processes, gloo transport): the data-parallel training step end to end --
bucketed all-reduces inserted into the backward launch list, 1/world in the SGD kernel, weight broadcast.
Checks: (a) bit-identical trainable weights on every rank after each step, (b) the 2-rank
result equals a single-process emulation that averages the two per-shard gradients."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ARCHI, BATCH, STEPS = "ssd_custom", 2, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import torch
    from jpeg_detection_resnet_ssd_amd import dist as dj
    from jpeg_detection_resnet_ssd_amd import workloads
    dj.init_from_env(backend="gloo")
    torch.cuda.set_device(0)
    model, sizes = workloads.build_ssd(ARCHI, weight_seed=42 + rank)   # different init: the broadcast must fix it
    model._ensure_params()
    dp = dj.DataParallel(model, bucket_mb=16)
    dp.broadcast_weights(0)
    w_init = model.flat_trainable.detach().cpu().clone() if rank == 0 else None
    x, y = workloads.synthetic_batch(ARCHI, sizes, BATCH, seed=1234 + rank)
    losses = []
    for _ in range(STEPS):
        losses.append(model.train_on_batch(x, y))
    torch.cuda.synchronize()
    out.put((rank, model.flat_trainable.detach().cpu().numpy(), losses, dp.n_buckets,
             None if w_init is None else w_init.numpy()))
    torch.distributed.destroy_process_group()


def test_two_rank_step_matches_gradient_averaging(cuda, monkeypatch):
    # identical launch configurations in the ranks and in the emulation (the plan-time autotuner may pick different
    # tiles / split-K factors per process, i.e. different fp32 summation orders, which is irrelevant to this test)
    monkeypatch.setenv("DJ_AUTOTUNE", "0")
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, w0, l0, nb, w_init), (_, w1, l1, _, _) = res
    assert nb >= 2
    np.testing.assert_array_equal(w0, w1)            # (a) replicas stay bit-identical

    # (b) single-process emulation: per-shard gradients at the same weights, averaged, one SGD kernel call
    from jpeg_detection_resnet_ssd_amd import workloads
    model, sizes = workloads.build_ssd(ARCHI, weight_seed=42)
    model._ensure_params()
    model.flat_trainable.copy_(torch.from_numpy(w_init).cuda())

    class FakeDist:
        rank = 0

        def finish_gradients(self):
            return 0.5

    plan = model._plan(BATCH, True, True)
    shards = [workloads.synthetic_batch(ARCHI, sizes, BATCH, seed=1234 + r) for r in range(2)]
    bn_state = model.flat_all[model._store["n_train"]:].clone()
    for _ in range(STEPS):
        acc = torch.zeros_like(model.flat_gradients)
        for x, y in shards:
            model.flat_all[model._store["n_train"]:].copy_(bn_state)   # BN moving stats do not feed the training math
            model._upload(plan, x, y)
            plan.run_forward()
            plan.run_backward()
            acc += model.flat_gradients
        model.flat_gradients.copy_(acc)
        model.dist = FakeDist()
        model._apply_optimizer()
        model.dist = None
    torch.cuda.synchronize()
    ref = model.flat_trainable.detach().cpu().numpy()
    err = np.abs(ref - w0).max() / np.abs(ref).max()
    # split-K gradient GEMMs accumulate with fp32 atomics in arrival order: two runs of the same step differ by a few 1e-6,
    # occasionally 5e-5 after two steps; a missing 1/world or a dropped bucket shows up at 1e-2 and above
    assert err <= 2e-4, err
