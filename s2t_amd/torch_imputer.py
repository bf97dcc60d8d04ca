"""Imputer (force-emit CTC) loss and CTC best alignment on the HIP path.

Mirror of fairseq/torch_imputer/imputer.py:87-152 (``imputer_loss``) and :284-325 (``best_alignment``); the CUDA
kernels they wrap (imputer.cu, best_alignment.cu) are replaced by ``s2t_ctc_loss_fwd/_bwd`` (with ``force_emits`` /
``paths``) and ``s2t_ctc_backtrace``.  Callers in the reference: criterions/ctc.py:283-345,455-465.
Same argument meaning: ``log_prob`` (T, N, C) log-softmax output, ``targets`` (N, S), ``force_emits`` (N, T) with the
index of the pinned extended-label state or -1, lengths (N,).
"""
import torch

from . import kernels as K


def _prep(log_prob, targets, input_lengths, target_lengths):
    T, B, V = log_prob.shape
    lp = log_prob.transpose(0, 1).contiguous()  # batch-major rows for the kernels
    dev = lp.device
    S = max(int(targets.shape[1]), 1)
    tm = targets.to(device=dev, dtype=torch.int64).contiguous()
    if tm.shape[1] == 0:
        tm = torch.zeros(B, 1, dtype=torch.int64, device=dev)
    il = torch.as_tensor(input_lengths).to(device=dev, dtype=torch.int32)
    tl = torch.as_tensor(target_lengths).to(device=dev, dtype=torch.int32)
    zero_lse = torch.zeros(B * T, dtype=torch.float32, device=dev)  # inputs are already log-probabilities
    return lp, tm, il, tl, zero_lse, B, T, V, S, 2 * S + 1


class _ImputerLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_prob, targets, force_emits, input_lengths, target_lengths, blank, zero_infinity):
        lp, tm, il, tl, lse0, B, T, V, S, L = _prep(log_prob, targets, input_lengths, target_lengths)
        dev = lp.device
        fe = force_emits.to(device=dev, dtype=torch.int64).contiguous()
        alpha = torch.empty(B, T, L, dtype=torch.float32, device=dev)
        beta = torch.empty(B, T, L, dtype=torch.float32, device=dev)
        nll = torch.empty(B, dtype=torch.float32, device=dev)
        K.ctc_loss_fwd(lp, V, B, T, V, lse0, tm, tm.shape[1], tl, il, blank, alpha, beta, L, nll, force_emits=fe)
        ctx.save_for_backward(lp, lse0, alpha, beta, nll, tm, tl, il)
        ctx.dims = (B, T, V, L, blank, zero_infinity)
        out = nll.to(log_prob.dtype)
        return torch.where(torch.isinf(out), torch.zeros_like(out), out) if zero_infinity else out

    @staticmethod
    def backward(ctx, g):
        lp, lse0, alpha, beta, nll, tm, tl, il = ctx.saved_tensors
        B, T, V, L, blank, zero_infinity = ctx.dims
        grad = torch.empty_like(lp)
        # The reference's kernel returns (exp(lp) - exp(log_alpha_beta + nll - lp)) * grad_out on valid frames, 0 beyond
        # (imputer.cu:626-633) — ATen's CTC convention, i.e. the gradient w.r.t. the LOGITS under a log_softmax, not the
        # plain derivative w.r.t. log_prob (-occupancy).  With lse = 0 the logits-gradient form of s2t_ctc_loss_bwd is
        # exactly that expression.  (Through a log_softmax the two conventions give the same gradient.)
        K.ctc_loss_bwd(lp, V, B, T, V, lse0, tm, tm.shape[1], tl, il, blank, alpha, beta, L, nll, 1.0, grad, V,
                       wrt_logprobs=False)
        grad = grad.view(B, T, V) * g.view(B, 1, 1).to(grad.dtype)
        return grad.transpose(0, 1), None, None, None, None, None, None


def imputer_loss(log_prob, targets, force_emits, input_lengths, target_lengths, blank=0, reduction="mean",
                 zero_infinity=False):
    """torch_imputer/imputer.py:87-152."""
    loss = _ImputerLoss.apply(log_prob, targets, force_emits, input_lengths, target_lengths, blank, zero_infinity)
    tl = torch.as_tensor(target_lengths).to(loss.device)
    if reduction == "mean":
        return (loss / tl.to(loss.dtype).clamp(min=1)).mean()
    if reduction == "sum":
        return loss.sum()
    return loss


@torch.no_grad()
def best_alignment_states(log_prob, targets, input_lengths, target_lengths, blank=0, zero_infinity=False):
    """The same alignment as a DEVICE tensor: int32 [B, T], the most probable CTC state per frame, 0 beyond an utterance's
    length (what the reference's callers build from the lists: ``a + [0] * (T - len(a))``, criterions/ctc.py:317-320).  No
    host round trip, so a training step that draws its PAE curriculum from the alignment can be captured into a hipGraph."""
    lp, tm, il, tl, lse0, B, T, V, S, L = _prep(log_prob, targets, input_lengths, target_lengths)
    dev = lp.device
    alpha = torch.empty(B, T, L, dtype=torch.float32, device=dev)
    paths = torch.zeros(B, T, L, dtype=torch.int32, device=dev)
    nll = torch.empty(B, dtype=torch.float32, device=dev)
    K.ctc_loss_fwd(lp, V, B, T, V, lse0, tm, tm.shape[1], tl, il, blank, alpha, None, L, nll, paths=paths)
    states = torch.empty(B, T, dtype=torch.int32, device=dev)
    K.ctc_backtrace(alpha, paths, tl, il, B, T, L, states)
    return states.masked_fill_(torch.arange(T, device=dev)[None, :] >= il.view(B, 1), 0)


@torch.no_grad()
def best_alignment(log_prob, targets, input_lengths, target_lengths, blank=0, zero_infinity=False):
    """torch_imputer/imputer.py:284-325 — list (per utterance) of the most probable CTC state sequence."""
    st = best_alignment_states(log_prob, targets, input_lengths, target_lengths, blank, zero_infinity).cpu()
    ilc = torch.as_tensor(input_lengths).cpu()
    return [st[b, : int(ilc[b])].tolist() for b in range(st.shape[0])]
