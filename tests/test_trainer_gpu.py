"""Trainer on the GPU: eager steps, one captured hipGraph per update, and the data-parallel variant (the library's own
RCCL communicator, here with one rank: bucketed all-reduce on a side stream INSIDE the captured graph, weight gradients
flushed per gradient stage) must walk the same parameter trajectory."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from s2t_amd import comm as Comm  # noqa: E402
from s2t_amd import criterions as C  # noqa: E402
from s2t_amd import s2t_transformer as M  # noqa: E402
from s2t_amd.legacy_distributed_data_parallel import LegacyDistributedDataParallel  # noqa: E402
from s2t_amd.trainer import Trainer  # noqa: E402

DEV = "cuda"
V = 61


def _model(seed, enc_layers=2):
    torch.manual_seed(seed)
    args = M.recipe_args(conformer=True, encoder_embed_dim=128, encoder_ffn_embed_dim=256, encoder_layers=enc_layers, decoder_layers=1,
                         decoder_embed_dim=128, decoder_ffn_embed_dim=256, encoder_attention_heads=2,
                         decoder_attention_heads=2, subsampling_filter=96, vocab_size=V, dropout=0.1,
                         attention_dropout=0.1, activation_dropout=0.1)
    return M.S2TTransformerModel.build_model(args, M.FakeTask(V)).prepare(torch.bfloat16, DEV)


def _sample():
    g = torch.Generator().manual_seed(3)
    B, T, U = 4, 200, 9
    lens = torch.tensor([200, 190, 111, 150])
    src = torch.randn(B, T, 80, generator=g)
    for b in range(B):
        src[b, lens[b]:] = 0
    target = torch.randint(4, V, (B, U), generator=g)
    target[:, -1] = 2
    prev = torch.roll(target, 1, 1)
    prev[:, 0] = 2
    return {"net_input": {"src_tokens": src.to(DEV), "src_lengths": lens.to(DEV), "prev_output_tokens": prev.to(DEV)},
            "target": target.to(DEV), "ntokens": int((target != 1).sum())}


def _run(mode, steps=4, enc_layers=2):
    from s2t_amd import functional as Fn
    model = _model(5, enc_layers)
    crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
    ddp = None
    if mode in ("ddp_graph", "ddp_eager", "ddp_graph_bf16"):
        Comm.init(0, 1, torch.device("cuda", 0))
        ddp = LegacyDistributedDataParallel(model, single_rank_collectives=True, buffer_size=2 ** 18,
                                            reduce_dtype=torch.bfloat16 if mode.endswith("bf16") else None)
    tr = Trainer(model, crit, ddp=ddp)
    sample = _sample()
    Fn.DROPOUT.begin_step(torch.device(DEV))
    Fn.DROPOUT.set_seed(100)
    losses = []
    if mode in ("eager", "ddp_eager"):
        for _ in range(steps + 3):
            losses.append(float(tr.train_step(sample)[0]))
        if ddp is not None:
            assert len(ddp._launched) == len(ddp.buckets)
    else:
        if ddp is not None:
            losses.append(float(tr.train_step(sample)[0]))  # learns the ready counts (as bench.py does)
        else:
            losses.append(float(tr.train_step(sample)[0]))
        tr.capture(sample, warmup=2)
        if ddp is not None:
            assert tr._graph2 is None, "the data-parallel update must be ONE graph with the collective inside"
        losses += [None, None]
        for _ in range(steps):
            losses.append(float(tr.replay()[0]))
    torch.cuda.synchronize()
    return losses, model.flat.master.detach().float().cpu().clone()


def test_graph_and_ddp_graph_follow_the_eager_trajectory():
    le, pe = _run("eager")
    lg, pg = _run("graph")
    ld, pd = _run("ddp_graph")
    lx, px = _run("ddp_eager")
    lb, pb = _run("ddp_graph_bf16")  # buckets reduced in bf16 (the reference's --fp16 wire format): gradients rounded once
    Comm.destroy()
    # same seeds, same masks, same arithmetic: the captured variants replay the eager step kernel for kernel
    for a, b in zip(le[3:], lg[3:]):
        assert abs(a - b) <= 2e-3 * abs(a), (le, lg)
    for a, b in zip(le[3:], ld[3:]):
        assert abs(a - b) <= 2e-3 * abs(a), (le, ld)
    # (at warm-up learning rates of 1e-7..1e-6 the loss itself only shows dropout noise over a handful of steps)
    assert all(v == v and abs(v) < 1e9 for v in le)
    p0 = _model(5).flat.master.detach().float().cpu()
    assert float((pe - p0).abs().max()) > 0  # the optimizer moved the parameters
    assert (pe - pg).abs().max() <= 1e-3 * pe.abs().max()
    assert (pe - pd).abs().max() <= 1e-3 * pe.abs().max()
    assert (pe - px).abs().max() <= 1e-3 * pe.abs().max()
    assert (pe - pb).abs().max() <= 1e-3 * pe.abs().max()
    for a, b in zip(le[3:], lb[3:]):
        assert abs(a - b) <= 2e-3 * abs(a), (le, lb)


@pytest.mark.parametrize("stages", [1, 2, 3, 4])
def test_gradient_stage_counts_walk_the_same_trajectory(stages, monkeypatch):
    """S2T_GRAD_STAGES: however many grouped weight-gradient launches a data-parallel backward pass is cut into (six encoder
    layers: cuts in front of layers 2 and 4, plus the one behind the decoder at 4), the captured data-parallel step must end
    where the single-launch eager step ends."""
    from s2t_amd import functional as Fn
    le, pe = _run("eager", enc_layers=6)
    monkeypatch.setattr(Fn, "GRAD_STAGES", stages)
    ld, pd = _run("ddp_graph", enc_layers=6)
    Comm.destroy()
    for a, b in zip(le[3:], ld[3:]):
        assert abs(a - b) <= 2e-3 * abs(a), (le, ld)
    assert (pe - pd).abs().max() <= 1e-3 * pe.abs().max()


def _sample2():
    """Same shapes as _sample, other frames / lengths / labels and two padded label positions (another sample size)."""
    g = torch.Generator().manual_seed(11)
    B, T, U = 4, 200, 9
    lens = torch.tensor([200, 120, 177, 64])
    src = torch.randn(B, T, 80, generator=g)
    for b in range(B):
        src[b, lens[b]:] = 0
    target = torch.randint(4, V, (B, U), generator=g)
    target[:, -1] = 2
    target[1, -2:] = torch.tensor([2, 1])
    target[3, -3:] = torch.tensor([2, 1, 1])
    prev = torch.roll(target, 1, 1)
    prev[:, 0] = 2
    return {"net_input": {"src_tokens": src.to(DEV), "src_lengths": lens.to(DEV), "prev_output_tokens": prev.to(DEV)},
            "target": target.to(DEV), "ntokens": int((target != 1).sum())}


def test_a_captured_step_replays_on_new_batches():
    """The per-batch target bookkeeping and the sample-size normaliser live OUTSIDE the graph (criterions.batch_bookkeeping,
    the hyper row): replay(sample) must give what eager steps on the same sequence of batches give."""
    from s2t_amd import functional as Fn
    seq = [0, 1, 1, 0, 1]
    out = []
    for mode in ("eager", "graph"):
        model = _model(5)
        crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
        tr = Trainer(model, crit, lr=2e-3, warmup_updates=2)
        batches = [_sample(), _sample2()]
        assert batches[0]["ntokens"] != batches[1]["ntokens"]
        Fn.DROPOUT.begin_step(torch.device(DEV))
        Fn.DROPOUT.set_seed(100)
        losses = []
        if mode == "eager":
            for _ in range(3):
                tr.train_step(batches[0])
            for i in seq:
                losses.append(float(tr.train_step(batches[i])[0]))
        else:
            static = _sample()
            tr.train_step(static)
            tr.capture(static, warmup=2)
            for i in seq:
                losses.append(float(tr.replay(batches[i])[0]))
        torch.cuda.synchronize()
        out.append((losses, model.flat.master.detach().float().cpu().clone()))
    (le, pe), (lg, pg) = out
    assert abs(le[0] - le[1]) > 1e-3 * abs(le[0])  # the two batches really differ
    for a, b in zip(le, lg):
        assert abs(a - b) <= 2e-3 * abs(a), (le, lg)
    # at lr 2e-3 Adam moves every parameter by about lr per update whatever the gradient's size, so the atomics' rounding
    # noise on near-zero gradients shows as up to lr per update on single entries: compare the movement on average
    p0 = _model(5).flat.master.detach().float().cpu()
    moved = float((pe - p0).abs().mean())
    assert moved > 1e-3
    assert float((pe - pg).abs().mean()) <= 0.02 * moved, (float((pe - pg).abs().mean()), moved)


@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_trainer_follows_the_reference_trajectory(golden_dir, mode):
    """Row a22 against the reference itself: five updates of the fixture model in fp32 with s2t_amd.trainer.Trainer
    (flat-buffer scale / clip / Adam kernels, inverse-sqrt schedule) vs the trajectory the reference's FairseqAdam,
    clip_grad_norm_ and InverseSquareRootSchedule produced in Trainer.train_step's order (oracle/gen_golden.py:
    trainer_case; fairseq/trainer.py:714-759, optim/adam.py:146-226, utils.py:328-369)."""
    import os

    import numpy as np

    from test_model_parity_gpu import build  # the fixture -> HIP model loader of the parity tests (same directory)

    z = np.load(os.path.join(golden_dir, "trainer_conformer_small.npz"))
    model, cfg = build(z, torch.float32)
    hp = {k[4:]: z[k] for k in z.files if k.startswith("hp::")}
    crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(40), label_smoothing=0.1, ctc_weight=float(cfg["ctc_weight"]))
    tr = Trainer(model, crit, lr=float(hp["lr"]), betas=tuple(float(b) for b in hp["betas"]), eps=float(hp["eps"]),
                 weight_decay=float(hp["weight_decay"]), clip_norm=float(hp["clip_norm"]),
                 warmup_updates=int(hp["warmup_updates"]), warmup_init_lr=float(hp["warmup_init_lr"]))
    sample = {"net_input": {"src_tokens": torch.from_numpy(z["in::src_tokens"]).to(DEV),
                            "src_lengths": torch.from_numpy(z["in::src_lengths"]).to(DEV),
                            "prev_output_tokens": torch.from_numpy(z["in::prev_output_tokens"]).to(DEV)},
              "target": torch.from_numpy(z["in::target"]).to(DEV), "ntokens": int(z["in::ntokens"])}
    n = len(z["out::loss"])
    losses, gnorms, lrs = [], [], []
    if mode == "eager":
        for _ in range(n):
            loss, log = tr.train_step(sample)
            losses.append(float(loss))
            gnorms.append(float(log["gnorm"]))
            lrs.append(float(log["lr"]))
    else:
        tr.capture(sample, warmup=0)
        for _ in range(n):
            losses.append(float(tr.replay()[0]))
            gnorms.append(float(tr.hyper[3]))
            lrs.append(float(tr.hyper[0]))
    np.testing.assert_allclose(lrs, z["out::lr"], rtol=1e-6)
    np.testing.assert_allclose(losses, z["out::loss"], rtol=1e-3)
    np.testing.assert_allclose(gnorms, z["out::gnorm"], rtol=5e-3)
    sd = model.state_dict()
    for k in z.files:
        if k.startswith("after::") and z[k].dtype.kind == "f" and not k.endswith(("linear_k.bias", "k_proj.bias")):
            if k[7:] not in sd or "_float_tensor" in k:
                continue
            got = sd[k[7:]].detach().float().cpu().numpy()
            assert np.abs(got - z[k]).max() <= 5e-3 * max(np.abs(z[k]).max(), 1e-3), k


@pytest.mark.parametrize("style", ["torch_optim", "data_inplace"])
def test_an_external_optimizer_moves_the_bf16_shadow(style):
    """The fairseq seam in bf16: an optimizer that is NOT the bundled Trainer (torch.optim stepping ``p`` in place; fairseq's
    Adam stepping ``p.data``, optim/adam.py:200-224) changes the fp32 masters behind the flat buffer's back — no version
    counter of the flat buffer moves.  The next forward must nevertheless run on the NEW weights: the bf16 shadow (and the
    transposed copies the fused backward kernels read) follow the masters, and the loss moves."""
    model = _model(7)
    assert model.shadow_managed is False
    crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
    sample = _sample()
    from s2t_amd import functional as Fn
    model.train()  # (the convolution module's backward needs training-mode BatchNorm)
    opt = torch.optim.SGD(model.parameters(), lr=0.5)

    def loss_and_grad(m=None, cr=None):
        m, cr = m or model, cr or crit
        Fn.DROPOUT.begin_step(torch.device(DEV))
        Fn.DROPOUT.set_seed(11)  # the same dropout masks every time: the losses differ through the weights alone
        m.flat.zero_grad()
        loss, _, _ = cr(m, sample)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss)

    l0 = loss_and_grad()
    before = model.flat.master.clone()
    if style == "torch_optim":
        opt.step()
    else:
        with torch.no_grad():
            for p in model.parameters():
                p.data.add_(p.grad.data, alpha=-0.5)
    assert not torch.equal(before, model.flat.master), "the optimizer must have stepped the flat fp32 masters in place"
    l1 = loss_and_grad()
    assert torch.equal(model.flat.shadow, model.flat.master.bfloat16()), "stale bf16 shadow after an external optimizer step"
    assert abs(l1 - l0) > 1e-3 * abs(l0), (l0, l1)
    # and the same weights through a fresh model give the same loss: nothing else went stale
    fresh = _model(7)
    fresh.train()
    fresh.load_state_dict(model.state_dict())
    crit2 = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
    l2 = loss_and_grad(fresh, crit2)
    assert abs(l2 - l1) <= 2e-3 * abs(l1), (l1, l2)


def test_attached_optimizer_refreshes_the_shadow_in_eval_mode_too():
    """weights stepped while the model is in eval() (dropout-off / frozen-BatchNorm fine-tuning), or followed by a
    forward under no_grad without a mode switch, used to run on a stale bf16 shadow: the refresh heuristic only fires for
    training-mode forwards.  ``model.attach_optimizer(opt)`` ties the refresh to the optimizer's steps instead."""
    model = _model(9)
    model.eval()
    sample = _sample()
    ni = sample["net_input"]
    opt = torch.optim.SGD(model.parameters(), lr=0.5)
    handle = model.attach_optimizer(opt)
    with torch.no_grad():
        a = model.encoder(ni["src_tokens"], ni["src_lengths"])["encoder_out"][0].float().clone()
    with torch.no_grad():
        for p in model.parameters():
            p.grad = torch.full_like(p, 1e-2) if p.grad is None else p.grad.fill_(1e-2)
    opt.step()
    with torch.no_grad():
        b = model.encoder(ni["src_tokens"], ni["src_lengths"])["encoder_out"][0].float().clone()
    assert torch.equal(model.flat.shadow, model.flat.master.bfloat16()), "stale bf16 shadow after a step in eval mode"
    assert not torch.equal(a, b)
    handle.remove()
