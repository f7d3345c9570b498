"""Data-parallel path (kws_amd/dp.py) on CPU with gloo, world_size 2 and 8.

The DP harness is backend-agnostic host logic (sharding, one flattened bucket, one
all-reduce, unflatten); on the GPU box the same code runs over RCCL.  The compute
injected here is the torch CPU port from oracle/ (tests may use the oracle as the
checker/stand-in; the product's compute path is HIP only).  Identity checked
(SURVEY.md section 8e): all-reduced gradients of the 2 shards == single-process gradients
of the whole batch."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _PortModule(torch.nn.Module):
    """FastGRNNCUDA-shaped callable backed by the CPU port (x:[T,B,F] -> hs:[T,B,H])."""

    def __init__(self, F, H, wRank=None, uRank=None):
        super().__init__()
        from oracle.fastgrnn_torch_port import FastGRNNCellPort
        self.cell = FastGRNNCellPort(F, H, wRank=wRank, uRank=uRank)

    def forward(self, x, h0=None):
        from oracle.fastgrnn_torch_port import unroll
        return unroll(self.cell, x, h0)


def _worker(rank, world, port, lowrank, ragged, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kws_amd.dp import GradBucket, data_parallel_step, shard_batch, shard_range
        torch.set_num_threads(1)
        T, B, F, H = 7, (9 if ragged else 8), 5, 12
        torch.manual_seed(0)                      # same replica everywhere
        r = 3 if lowrank else None
        model = _PortModule(F, H, r, r).double()
        g = torch.Generator().manual_seed(1)
        x = torch.randn(T, B, F, generator=g, dtype=torch.float64)
        G = torch.randn(T, B, H, generator=g, dtype=torch.float64)
        params = list(model.parameters())
        bucket = GradBucket(params, world, divisor=1)       # keep the SUM: gradient of the summed loss
        xs, Gs = shard_batch(x, rank, world), shard_batch(G, rank, world)
        lo, hi = shard_range(B, rank, world)
        assert xs.shape[1] == hi - lo
        hs_local = data_parallel_step(model, xs, Gs, bucket)
        dp_grads = [p.grad.clone() for p in params]
        # single-process reference on the whole batch
        for p in params:
            p.grad = None
        hs_full = model(x)
        hs_full.backward(G)
        ok = torch.allclose(hs_local, hs_full[:, lo:hi].detach(), atol=1e-12)
        for a, p in zip(dp_grads, params):
            ok = ok and torch.allclose(a, p.grad, rtol=1e-10, atol=1e-12)
        # mean convention (default divisor = world)
        b2 = GradBucket(params, world)
        for p in params:
            p.grad = torch.full_like(p, float(rank + 1))
        b2.all_reduce_()
        ok = ok and all(torch.allclose(p.grad, torch.full_like(p, (1 + 2) / 2.0)) for p in params)
        ok = ok and bucket.total == sum(p.numel() for p in params)
        q.put((rank, bool(ok), bucket.total))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("lowrank,ragged", [(False, False), (True, True)])
def test_dp_two_ranks_matches_single_process(lowrank, ragged):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, lowrank, ragged, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    F, H = 5, 12
    expect = (3 * (F + H) + 2 * 3 * H + 2 * H + 2) if lowrank else (H * F + H * H + 2 * H + 2)
    assert res[0][2] == expect


def _worker8(rank, world, port, q):
    """world size 8, ragged shards (B = 32768 + 5: the first five ranks hold one utterance more), the sum identity
    against a single-process run (rank 0 only), then both bucket paths under a real collective: gradients that alias
    one flat buffer (zero-copy) and the pack -> all-reduce -> unpack fallback."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kws_amd.dp import GradBucket, data_parallel_step, shard_batch, shard_range
        torch.set_num_threads(1)
        T, B, F, H = 4, 32768 + 5, 5, 12
        torch.manual_seed(0)
        model = _PortModule(F, H).double()
        g = torch.Generator().manual_seed(1)
        x = torch.randn(T, B, F, generator=g, dtype=torch.float64)
        G = torch.randn(T, B, H, generator=g, dtype=torch.float64)
        params = list(model.parameters())
        lo, hi = shard_range(B, rank, world)
        ok = (hi - lo) == (4097 if rank < 5 else 4096)
        bucket = GradBucket(params, world, divisor=1)
        data_parallel_step(model, shard_batch(x, rank, world), shard_batch(G, rank, world), bucket)
        dp_grads = [p.grad.clone() for p in params]
        if rank == 0:                                  # the whole batch in one process
            for p in params:
                p.grad = None
            model(x).backward(G)
            for a, p in zip(dp_grads, params):
                ok = ok and torch.allclose(a, p.grad, rtol=1e-9, atol=1e-10)
        # zero-copy path: the gradients are views of one flat buffer, in parameter order
        sizes = [p.numel() for p in params]
        flat = torch.full((sum(sizes),), float(rank + 1), dtype=torch.float64)
        for p, v in zip(params, flat.split(sizes)):
            p.grad = v.view_as(p)
        b2 = GradBucket(params, world)                 # mean over ranks
        out = b2.all_reduce_()
        ok = ok and out.data_ptr() == flat.data_ptr() and torch.allclose(flat, torch.full_like(flat, 4.5))
        # fallback: one gradient lives elsewhere -> pack, collective, unpack
        for p in params:
            p.grad = torch.full_like(p, float(rank + 1))
        ok = ok and b2.shared_flat_() is None
        out = b2.all_reduce_()
        ok = ok and out.data_ptr() == b2.flat.data_ptr()
        ok = ok and all(torch.allclose(p.grad, torch.full_like(p, 4.5)) for p in params)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_dp_eight_ranks_ragged_shards_and_both_bucket_paths():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r for r, _ in res) == list(range(8)) and all(ok for _, ok in res), res


def test_shard_range_covers_everything():
    from kws_amd.dp import shard_range
    for n in (0, 1, 7, 8, 4096, 32768):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(32768, 3, 8) == (12288, 16384)


def test_bucket_size_northstar():
    """82 952 B dense (H=128,F=32) / 53 256 B low-rank (H=256,r=16): SURVEY.md section 8e."""
    from kws_amd import FastGRNNCUDA
    from kws_amd.dp import GradBucket
    m = FastGRNNCUDA(32, 128, device="cpu")
    assert GradBucket(list(m.parameters()), world=8).total * 4 == 82952
    m = FastGRNNCUDA(32, 256, wRank=16, uRank=16, device="cpu")
    assert GradBucket(list(m.parameters()), world=8).total * 4 == 53256


def test_grad_bucket_detects_gradients_that_share_one_buffer():
    """Zero-copy path of GradBucket: gradients that are back-to-back views of one storage (how the operator
    shim allocates them) are all-reduced in place; anything else falls back to pack/unpack."""
    sys.path.insert(0, ROOT)
    from kws_amd.dp import GradBucket
    params = [torch.nn.Parameter(torch.zeros(3, 4)), torch.nn.Parameter(torch.zeros(1, 5)), torch.nn.Parameter(torch.zeros(1, 1))]
    flat = torch.arange(18, dtype=torch.float32)
    for p_, v in zip(params, flat.split([12, 5, 1])):
        p_.grad = v.view_as(p_)
    bucket = GradBucket(params, world=1, divisor=1)
    shared = bucket.shared_flat_()
    assert shared is not None and shared.data_ptr() == flat.data_ptr() and torch.equal(shared, flat)
    shared.mul_(2.0)                                            # what an in-place collective does
    assert torch.equal(params[1].grad, 2.0 * torch.arange(12, 17, dtype=torch.float32).view(1, 5))
    params[1].grad = params[1].grad.clone()                     # one gradient elsewhere: no aliasing possible
    assert bucket.shared_flat_() is None
    params[1].grad = None
    assert bucket.shared_flat_() is None


def _worker_bench_step(rank, world, port, q):
    """bench.py's OWN step (bench.make_step) and spin-up agreement (bench.agree_on_count) over gloo, the CPU port as the
    compute: after the step every rank holds, bit for bit, what one process gets by running the shards one after the
    other and averaging -- SUM of two addends does not depend on their order, the scale by 1/2 is exact."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from kws_amd.dp import GradBucket, shard_batch
        torch.set_num_threads(1)
        T, B, F, H = 9, 22, 5, 12
        torch.manual_seed(0)
        model = _PortModule(F, H)
        g = torch.Generator().manual_seed(1)
        x = torch.randn(T, B, F, generator=g)
        G = torch.randn(T, B, H, generator=g)
        params = list(model.parameters())
        # every rank sizes its own spin-up; all must run the same number of (collective-holding) steps
        n = bench.agree_on_count(3 + 2 * rank, world, torch.device("cpu"))
        bucket = GradBucket(params, world)                     # mean over ranks (gloo: SUM + one scale)
        step = bench.make_step(model, shard_batch(x, rank, world), shard_batch(G, rank, world), params, bucket)
        for _ in range(n):
            step()
        got = [p.grad.clone() for p in params]
        # one process, the shards one after the other
        per_shard = []
        for r in range(world):
            for p in params:
                p.grad = None
            model(shard_batch(x, r, world)).backward(shard_batch(G, r, world))
            per_shard.append([p.grad.clone() for p in params])
        ok = n == 3 + 2 * (world - 1)
        for k, a in enumerate(got):
            ref = (per_shard[0][k] + per_shard[1][k]) * (1.0 / world)
            ok = ok and torch.equal(a, ref)
        q.put((rank, bool(ok), n))
    finally:
        dist.destroy_process_group()


def test_bench_step_over_gloo_equals_the_single_process_average_bit_for_bit():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bench_step, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert {n for _, _, n in res} == {5}
