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
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)
