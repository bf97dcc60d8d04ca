"""World-size-2 gloo test of the data-parallel wrapper's protocol (runs on CPU): bucketed in-place all-reduce of the
flat gradient buffer, sum/world semantics, no_sync accumulation, overlap bookkeeping driven by grad-ready hooks.
(reference behaviour: fairseq/distributed/legacy_distributed_data_parallel.py:76-160)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Parameter(torch.zeros(300, 7))
        self.b = nn.Parameter(torch.zeros(1000))
        self.c = nn.Parameter(torch.zeros(64, 64))
        self.flat = None


def _worker(rank, world, port, q, reduce_dtype=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from s2t_amd import functional as Fn
        from s2t_amd.flat_params import FlatParameters
        from s2t_amd.legacy_distributed_data_parallel import LegacyDistributedDataParallel

        m = Toy()
        m.flat = FlatParameters(m, torch.float32)
        ddp = LegacyDistributedDataParallel(m, buffer_size=2048, reduce_dtype=reduce_dtype)  # several buckets, parameters straddle them
        assert len(ddp.buckets) >= 3 and ddp.world_size == world
        params = [m.c, m.b, m.a]  # "backward order"

        def fake_backward(scale):
            for i, p in enumerate(params):
                p.grad.add_(torch.full_like(p.grad, scale * (rank + 1) * (i + 1)))
                Fn._ready(p)
                if p is m.b:  # a tied parameter reports twice
                    p.grad.add_(torch.full_like(p.grad, scale))
                    Fn._ready(p)

        expect = lambda i, scale: scale * (i + 1) * sum(r + 1 for r in range(world)) / world  # noqa: E731
        for step in range(3):  # step 0 learns the ready counts, later steps launch buckets from the hooks
            m.flat.zero_grad()
            ddp.begin_backward()
            fake_backward(1.0)
            if step > 0:
                assert len(ddp._launched) >= 1, "no bucket was reduced before all_reduce_grads()"
            ddp.all_reduce_grads()
            assert torch.allclose(m.c.grad, torch.full_like(m.c.grad, expect(0, 1.0)))
            assert torch.allclose(m.b.grad, torch.full_like(m.b.grad, expect(1, 1.0) + 1.0))
            assert torch.allclose(m.a.grad, torch.full_like(m.a.grad, expect(2, 1.0)))
        # no_sync: gradients stay local and keep accumulating
        m.flat.zero_grad()
        with ddp.no_sync():
            ddp.begin_backward()
            fake_backward(1.0)
            ddp.all_reduce_grads()
        assert torch.allclose(m.c.grad, torch.full_like(m.c.grad, float(rank + 1)))
        ddp.begin_backward()
        fake_backward(1.0)
        ddp.all_reduce_grads()
        assert torch.allclose(m.c.grad, torch.full_like(m.c.grad, 2 * expect(0, 1.0)))
        # attribute access is proxied to the wrapped module (module_proxy_wrapper.py)
        assert ddp.a is m.a
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("reduce_dtype", [None, torch.bfloat16])
def test_ddp_world2_gloo(reduce_dtype):
    """(bf16: the buckets travel in half precision as under the reference's --fp16; the test values are exact in bf16)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, reduce_dtype)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


def _worker_bound(rank, world, port, q):
    """Random gradients of realistic spread reduced once with fp32 buckets and once with bf16 buckets."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from s2t_amd import functional as Fn
        from s2t_amd.flat_params import FlatParameters
        from s2t_amd.legacy_distributed_data_parallel import LegacyDistributedDataParallel

        out = {}
        for tag, rd in (("fp32", None), ("bf16", torch.bfloat16)):
            m = Toy()
            m.flat = FlatParameters(m, torch.float32)
            ddp = LegacyDistributedDataParallel(m, buffer_size=2048, reduce_dtype=rd)
            g = torch.Generator().manual_seed(100 + rank)
            m.flat.zero_grad()
            ddp.begin_backward()
            for p in (m.c, m.b, m.a):
                # log-normal magnitudes over four decades, random signs: what a gradient buffer looks like
                mag = torch.exp(torch.randn(p.shape, generator=g) * 2.0 - 4.0)
                p.grad.copy_(mag * torch.sign(torch.randn(p.shape, generator=g)))
                Fn._ready(p)
            ddp.all_reduce_grads()
            out[tag] = m.flat.grad.clone()
        q.put((rank, out["fp32"], out["bf16"]))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


def test_bf16_buckets_stay_within_a_rounding_of_fp32_buckets():
    """bench.py sends the gradient buckets in the training dtype (bf16), the reference's wire format under --fp16
    (legacy_distributed_data_parallel.py:44-48: the buffer has the parameters' half-precision dtype).  What that costs against
    fp32 buckets: each rank's contribution is rounded to bf16 once and the sum once more (8 mantissa bits: 2^-9 relative each
    way), so an element of the averaged gradient is within 3 * 2^-9 of the largest contribution's magnitude, and the buffer's
    norm moves by far less (the roundings are unbiased and independent)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bound, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, a, b in res:
        assert b is not None, "rank %d: %s" % (rank, a)
        # per element: bounded by the roundings of the contributions (their magnitudes sum to at most world * |mean| + slack;
        # use the fp32 mean's own scale plus the largest single contribution seen through the bf16 result)
        scale = torch.maximum(a.abs(), b.abs()).clamp_min(1e-30)
        big = (b - a).abs() / scale
        # cancellation (contributions of opposite sign) can make a tiny mean out of large parts: compare those against the parts
        assert float(((b - a).abs() <= 3 * 2.0 ** -9 * (scale + 0.2)).float().mean()) == 1.0
        assert float(big.median()) < 2.0 ** -8
        assert abs(float(b.norm() / a.norm()) - 1.0) < 1e-3
    assert torch.equal(res[0][2], res[1][2]), "both ranks must hold the same reduced gradient"
