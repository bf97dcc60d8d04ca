"""Dropout on the HIP path (FairseqDropout semantics, counter-based masks regenerated in backward).
No bit-parity with the CPU Philox/MT stream is required (SURVEY.md §8b "Seeding"); what is checked:
keep-rate / scaling / determinism of the mask, and that forward and backward use the SAME masks — a directional finite
difference of the (deterministic, fixed-seed) loss must match the analytic gradient."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from s2t_amd import criterions as C  # noqa: E402
from s2t_amd import functional as Fn  # noqa: E402
from s2t_amd import kernels as K  # noqa: E402
from s2t_amd import s2t_transformer as M  # noqa: E402

DEV = "cuda"


def test_mask_statistics_and_determinism():
    seed = torch.tensor([12345], dtype=torch.int64, device=DEV)
    x = torch.ones(4096, 256, device=DEV)
    out1, out2, out3 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    K.dropout(x, 256, out1, 256, 4096, 256, (0.1, seed, 7))
    K.dropout(x, 256, out2, 256, 4096, 256, (0.1, seed, 7))
    K.dropout(x, 256, out3, 256, 4096, 256, (0.1, seed, 8))
    assert torch.equal(out1, out2)
    assert not torch.equal(out1, out3)
    keep = (out1 != 0).float().mean().item()
    assert abs(keep - 0.9) < 3e-3
    vals = out1[out1 != 0]
    assert torch.allclose(vals, torch.full_like(vals, 1 / 0.9))
    # the GEMM epilogue uses the same (row, col) -> mask map as the standalone kernel
    A = torch.eye(256, device=DEV).repeat(16, 1)  # [4096, 256]
    Bm = torch.eye(256, device=DEV)
    y = torch.empty(4096, 256, device=DEV)
    K.gemm(A, Bm, y, M=4096, N=256, K=256, lda=256, ldb=256, ldc=256, drop=(0.1, seed, 7))
    ref = torch.empty_like(A)
    K.dropout(A, 256, ref, 256, 4096, 256, (0.1, seed, 7))
    assert torch.equal(y, ref)


@pytest.mark.parametrize("conformer", [False, True])
def test_forward_and_backward_share_masks(conformer):
    torch.manual_seed(3)
    V = 50
    args = M.recipe_args(conformer=conformer, encoder_embed_dim=32, encoder_ffn_embed_dim=64, encoder_layers=2,
                         decoder_layers=1, decoder_embed_dim=32, decoder_ffn_embed_dim=64, encoder_attention_heads=2,
                         decoder_attention_heads=2, subsampling_filter=32, vocab_size=V, dropout=0.25,
                         attention_dropout=0.25, activation_dropout=0.25)
    model = M.S2TTransformerModel.build_model(args, M.FakeTask(V)).prepare(torch.float32, DEV)
    model.train()
    crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
    g = torch.Generator().manual_seed(1)
    B, T = 3, 41
    lens = torch.tensor([41, 33, 25])
    src = torch.randn(B, T, 80, generator=g)
    for b in range(B):
        src[b, lens[b]:] = 0
    target = torch.tensor([[5, 6, 7, 2], [9, 10, 2, 1], [11, 2, 1, 1]])
    prev = torch.tensor([[2, 5, 6, 7], [2, 9, 10, 1], [2, 11, 1, 1]])
    sample = {"net_input": {"src_tokens": src.to(DEV), "src_lengths": lens.to(DEV), "prev_output_tokens": prev.to(DEV)},
              "target": target.to(DEV), "ntokens": 9}

    def loss_value():
        Fn.DROPOUT.begin_step(DEV)  # same seed, sites restart -> identical masks
        with torch.no_grad():
            loss, _, _ = crit(model, sample, sync_logging=False)
        return float(loss)

    Fn.DROPOUT.begin_step(DEV)
    Fn.DROPOUT.set_seed(99)
    model.flat.zero_grad()
    # BatchNorm running stats move during training forwards but do not enter the training-mode output
    loss, _, log = crit(model, sample)
    loss.backward()
    torch.cuda.synchronize()
    l0 = loss_value()
    assert abs(l0 - log["loss"]) < 1e-4 * abs(l0), "the same seed/sites must reproduce the same loss"
    grad = model.flat.grad.clone()
    direction = torch.randn(model.flat.numel, generator=torch.Generator().manual_seed(5)).to(DEV)
    direction = direction / direction.norm()
    eps = 2e-2
    w0 = model.flat.master.clone()
    model.flat.master.copy_(w0 + eps * direction)
    lp = loss_value()
    model.flat.master.copy_(w0 - eps * direction)
    lm = loss_value()
    model.flat.master.copy_(w0)
    fd = (lp - lm) / (2 * eps)
    an = float((grad * direction).sum())
    assert abs(fd - an) <= 3e-2 * max(abs(an), abs(fd), 1.0), (fd, an)
    # and dropout really is on: an eval-mode forward gives a different loss
    model.eval()
    with torch.no_grad():
        le, _, _ = crit(model, sample, sync_logging=False)
    assert abs(float(le) - l0) > 1e-5 * abs(l0)


def test_layernorm_backward_hands_over_the_dropped_branch_gradient():
    """The LayerNorm backward (bf16, d = 256) also writes dropout(dx) under the mask of the block in front; the kernel's
    second output is bit-identical to s2t_dropout on dx, the hand-over really happens in a model backward, and switching
    it off gives the same gradients."""
    from s2t_amd import functional as Fn
    g = torch.Generator().manual_seed(2)
    rows, cols = 777, 256
    x = (torch.randn(rows, cols, generator=g)).to(torch.bfloat16).to(DEV)
    dy = (torch.randn(rows, cols, generator=g)).to(torch.bfloat16).to(DEV)
    dres = (torch.randn(rows, cols, generator=g)).to(torch.bfloat16).to(DEV)
    w = (1 + 0.1 * torch.randn(cols, generator=g)).to(DEV)
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    y = torch.empty_like(x)
    K.layernorm_fwd(x, w, torch.zeros(cols, device=DEV), y, mean, rstd, rows, cols)
    seed = torch.tensor([99], dtype=torch.int64, device=DEV)
    drop = (0.25, seed, 5)
    dx, dxd = torch.empty_like(x), torch.empty_like(x)
    dg, db = torch.zeros(cols, device=DEV), torch.zeros(cols, device=DEV)
    K.layernorm_bwd(x, w, dy, mean, rstd, dx, dg, db, rows, cols, dres=dres, dx_drop=dxd, drop=drop)
    ref = torch.empty_like(x)
    K.dropout(dx, cols, ref, cols, rows, cols, drop)
    assert torch.equal(dxd, ref) and float((dxd == 0).float().mean()) > 0.2

    def grads(fuse):
        torch.manual_seed(4)
        V = 50
        args = M.recipe_args(conformer=True, encoder_layers=2, decoder_layers=1, vocab_size=V, dropout=0.2,
                             attention_dropout=0.2, activation_dropout=0.2)  # d = 256: the fast LayerNorm kernels
        model = M.S2TTransformerModel.build_model(args, M.FakeTask(V)).prepare(torch.bfloat16, DEV)
        model.train()
        crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
        gg = torch.Generator().manual_seed(1)
        B, T = 2, 83
        src = torch.randn(B, T, 80, generator=gg).to(DEV)
        lens = torch.tensor([83, 61]).to(DEV)
        tgt = torch.randint(4, V, (B, 7), generator=gg)
        tgt[:, -1] = 2
        prev = torch.roll(tgt, 1, 1)
        sample = {"net_input": {"src_tokens": src, "src_lengths": lens, "prev_output_tokens": prev.to(DEV)},
                  "target": tgt.to(DEV), "ntokens": 14}
        Fn._FUSE_LN_DROP = fuse
        Fn.DROP_STATS.update(handed_over=0, launched=0)
        Fn.DROPOUT.begin_step(torch.device(DEV))
        Fn.DROPOUT.set_seed(7)
        model.flat.zero_grad()
        crit(model, sample)[0].backward()
        torch.cuda.synchronize()
        return model.flat.grad.clone(), dict(Fn.DROP_STATS)

    try:
        g1, st1 = grads(True)
        g0, st0 = grads(False)
    finally:
        Fn._FUSE_LN_DROP = True
    assert st0["handed_over"] == 0 and st0["launched"] > 8
    assert st1["handed_over"] >= 8, st1  # 4 per Conformer layer + decoder blocks
    assert float((g1 - g0).norm() / g0.norm()) < 1e-5  # same arithmetic; parameter-level sums use float atomics
