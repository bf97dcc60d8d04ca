"""One rank of the two-rank data-parallel check on ONE GPU (started by tests/conftest.py before the test session touches
the GPU, read by tests/test_ddp_two_ranks_gpu.py).  Both ranks sit on cuda:0; the host group is gloo, so the gradient
buckets travel through torch.distributed (s2t_amd.comm's RCCL communicator needs one device per rank).

What a rank establishes, per model variant, on the REAL model (reference protocol:
fairseq/distributed/legacy_distributed_data_parallel.py:76-160, fairseq/trainer.py:714-741):

  1. g_r = the plain single-process gradient of batch r, for r = 0, 1 (no wrapper);
  2. wrapped in LegacyDistributedDataParallel, rank r runs batch r: after all_reduce_grads() the flat gradient buffer is
     (g_0 + g_1) / 2 — on the learning pass (everything reduced at the end) AND on the overlapped passes, where every bucket
     is launched from functional._ready hooks while backward is still running (a bucket launched before one of its
     parameters' last contribution would miss that contribution);
  3. the number of _ready reports per parameter is the same on every pass, on both ranks;
  4. three Trainer updates: eager data-parallel steps against the captured form (graph 1 = forward + backward, eager bucket
     all-reduce, graph 2 = clip + Adam) end at the same fp32 masters, and the two ranks' masters agree (the reduced gradients are equal bit for bit; each rank's
     gradient norm is an fp32 atomic sum of its own, so the clip coefficients may differ in the last bit).

Usage: ddp_two_ranks_worker.py RANK WORLD PORT OUTDIR
"""
import json
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

V = 200


def _batch(torch, seed, B, T, dev):
    g = torch.Generator().manual_seed(seed)
    lens = sorted([T] + [int(torch.randint(int(0.6 * T), T + 1, (1,), generator=g)) for _ in range(B - 1)], reverse=True)
    src = torch.randn(B, T, 80, generator=g)
    for b, l in enumerate(lens):
        src[b, l:] = 0
    ul = [int(torch.randint(5, 12, (1,), generator=g)) for _ in range(B)]
    U = max(ul) + 1
    target = torch.full((B, U), 1, dtype=torch.long)
    prev = torch.full((B, U), 1, dtype=torch.long)
    for b, u in enumerate(ul):
        toks = torch.randint(4, V, (u,), generator=g)
        target[b, :u] = toks
        target[b, u] = 2
        prev[b, 0] = 2
        prev[b, 1:u + 1] = toks
    return {"net_input": {"src_tokens": src.to(dev), "src_lengths": torch.tensor(lens).to(dev), "prev_output_tokens": prev.to(dev)},
            "target": target.to(dev), "ntokens": int(sum(ul) + B)}


def _variant(name, rank, world, torch, dist):
    from s2t_amd import criterions as C
    from s2t_amd import functional as Fn
    from s2t_amd import s2t_transformer as M
    from s2t_amd.legacy_distributed_data_parallel import LegacyDistributedDataParallel
    from s2t_amd.trainer import Trainer

    dev = torch.device("cuda", 0)
    conformer = name == "conformer_bf16"
    dtype = torch.bfloat16 if conformer else torch.float32
    res = {}

    def model():
        torch.manual_seed(17)
        a = M.recipe_args(conformer=conformer, vocab_size=V, encoder_layers=2, decoder_layers=1)  # d = 256, F = 2048, 4 heads
        return M.S2TTransformerModel.build_model(a, M.FakeTask(V)).prepare(dtype, dev)

    crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
    batches = [_batch(torch, 100 + r, 6, 400, dev) for r in range(world)]  # (same shapes on every rank, other contents)

    def fwd_bwd(m, sample, ddp=None, tally=None):
        m.train()
        m.flat.zero_grad()
        if ddp is not None:
            ddp.begin_backward()
            cb = Fn._HOOKS["grad_ready"]  # the wrapper's per-parameter callback (counting pass: _count, later: _on_ready)
            assert cb is not None

            def spy(p):
                tally[names[id(p)]] = tally.get(names[id(p)], 0) + 1
                cb(p)

            Fn._HOOKS["grad_ready"] = spy
        loss, _, _ = crit(m, sample)
        loss.backward()
        early = len(ddp._launched) if ddp is not None else 0
        if ddp is not None:
            ddp.all_reduce_grads()
        torch.cuda.synchronize()
        return float(loss.detach()), early

    # 1. single-process gradients of both batches
    m = model()
    singles = []
    for r in range(world):
        fwd_bwd(m, batches[r])
        singles.append(m.flat.grad.detach().clone())
    fwd_bwd(m, batches[0])
    res["repeat_noise"] = float((m.flat.grad - singles[0]).norm() / singles[0].norm())
    mean = sum(singles) / world
    res["ranks_differ"] = float((singles[0] - singles[1]).norm() / singles[0].norm())

    # 2. + 3. the wrapped model: learning pass, then two overlapped passes
    counts = []
    tally = {}
    names = {id(p): n for n, p in m.named_parameters()}
    ddp = LegacyDistributedDataParallel(m, buffer_size=2 ** 18)  # 1 MiB buckets: a dozen of them, parameters straddle them
    assert ddp.world_size == world and ddp.active
    res["buckets"] = len(ddp.buckets)
    errs, earlies = [], []
    for _ in range(3):
        tally.clear()
        _, early = fwd_bwd(ddp, batches[rank], ddp, tally)
        counts.append(dict(tally))
        errs.append(float((m.flat.grad - mean).norm() / mean.norm()))
        earlies.append(early)
    res["mean_err"] = errs
    res["launched_before_the_end"] = earlies
    res["ready_counts_stable"] = counts[0] == counts[1] == counts[2]
    res["ready_counts"] = counts[0]
    res["ready_params"] = len(counts[0])
    res["params"] = len(list(m.parameters()))
    # a parameter that never reports may not have a gradient either (its bucket would go out without it)
    res["silent_with_gradient"] = [n for n, p in m.named_parameters()
                                   if n not in counts[0] and float(singles[rank][m.flat.offsets[id(p)]:m.flat.offsets[id(p)] + p.numel()].abs().max()) > 0]
    gathered = [None] * world
    dist.all_gather_object(gathered, counts[0])
    res["ready_counts_equal_on_ranks"] = all(g_ == gathered[0] for g_ in gathered)
    # the reduced buffer is the same on both ranks
    mine = m.flat.grad.detach().cpu()
    both = [None] * world
    dist.all_gather_object(both, mine)
    res["reduced_equal_on_ranks"] = bool(torch.equal(both[0], both[1]))
    del ddp, m

    # 4. three updates, eager against captured
    ssg = sum(b["ntokens"] for b in batches)
    finals = {}
    for mode in ("eager", "graph"):
        m = model()
        ddp = LegacyDistributedDataParallel(m, buffer_size=2 ** 18)
        tr = Trainer(m, crit, ddp=ddp, lr=2e-3, warmup_updates=4)
        sample = batches[rank]
        losses = []
        if mode == "eager":
            for _ in range(6):
                losses.append(float(tr.train_step(sample, ssg)[0]))
        else:
            losses.append(float(tr.train_step(sample, ssg)[0]))  # learns the ready counts
            tr.capture(sample, ssg, warmup=2)
            assert tr._graph2 is not None  # torch.distributed collectives stay outside the capture
            losses += [None, None]
            for _ in range(3):
                losses.append(float(tr.replay()[0]))
        torch.cuda.synchronize()
        finals[mode] = (losses, m.flat.master.detach().float().cpu().clone())
        if mode == "graph":
            tr.release()
        del tr, ddp, m
    p0 = model().flat.master.detach().float().cpu()
    pe, pg = finals["eager"][1], finals["graph"][1]
    res["moved"] = float((pe - p0).abs().mean())
    res["traj_mean_diff"] = float((pe - pg).abs().mean())
    res["traj_max_diff_rel"] = float((pe - pg).abs().max() / pe.abs().max())
    res["losses_eager"] = finals["eager"][0]
    res["losses_graph"] = finals["graph"][0]
    both = [None] * world
    dist.all_gather_object(both, pg)
    res["masters_rank_diff_graph"] = float((both[0] - both[1]).abs().max() / both[0].abs().max())
    dist.all_gather_object(both, pe)
    res["masters_rank_diff_eager"] = float((both[0] - both[1]).abs().max() / both[0].abs().max())
    return res


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    result = {"rank": rank}
    try:
        import torch
        import torch.distributed as dist

        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        from s2t_amd import _lib
        from s2t_amd import functional as Fn

        _lib.lib()  # fails loudly without the HIP library
        old = (Fn._RB_MIN_ROWS, Fn._FFN_FUSED_MIN_ROWS)
        Fn._RB_MIN_ROWS = Fn._FFN_FUSED_MIN_ROWS = 0  # 600 encoder rows still take the bench's row-block / fused FFN kernels
        try:
            for name in ("transformer_fp32", "conformer_bf16"):
                result[name] = _variant(name, rank, world, torch, dist)
        finally:
            Fn._RB_MIN_ROWS, Fn._FFN_FUSED_MIN_ROWS = old
        result["ok"] = True
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001 — reported to the test
        result["ok"] = False
        result["error"] = traceback.format_exc()
    with open(os.path.join(out, "rank%d.json.tmp" % rank), "w") as f:
        json.dump(result, f)
    os.replace(os.path.join(out, "rank%d.json.tmp" % rank), os.path.join(out, "rank%d.json" % rank))


if __name__ == "__main__":
    main()
