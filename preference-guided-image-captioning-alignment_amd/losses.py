"""Loss objects with the reference's call signatures, evaluated by the HIP kernels.

* ``ContrastiveLoss``     reference model.py:957-1000 (the trainer's NT-Xent; inputs already normalised)
* ``PreferenceLoss``      reference model.py:1003-1085 (2-forward, reference-free, length-mean log-prob)
* ``DPOPreferenceLoss``   reference components.py:148-249 (4-forward DPO, metrics dict)
* ``compute_sequence_logprobs``  reference components.py:321-362

These are the API-compatible entry points for callers that already hold embeddings / materialised
logits.  The training hot path does not go through materialised logits: see ``steps.DPOStep``.
Like the reference's ``nn.Module`` losses they can be differentiated: inputs that carry ``requires_grad`` receive their
gradient from ``loss.backward()`` (``torch.autograd.Function`` wrappers around the HIP forward / backward kernels).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import hip
from .engine import F32, I32, I64, NTXentEngine, Workspace, make_seq_batch


def _dev(t: torch.Tensor) -> torch.device:
    if not t.is_cuda:
        raise RuntimeError("pgca_amd losses run on the MI355X: pass device tensors (no CPU fallback)")
    return t.device


class _NTXentFn(torch.autograd.Function):
    """NT-Xent value (and, when an input carries ``requires_grad``, its gradient - evaluated by the HIP kernels in the
    forward, because the engine's buffers are reused by the next call) for ``loss.backward()`` callers."""

    @staticmethod
    def forward(ctx, img, txt, owner):
        dev = _dev(img)
        if owner._eng is None or owner._eng.ws.device != dev or owner._eng.P != img.shape[1]:
            owner._eng = NTXentEngine(Workspace(dev), img.shape[1], owner.temperature, tag="loss.ntx")
        eng = owner._eng
        eng.tau = float(owner.temperature)
        loss, _, _ = eng.forward(img.detach().to(F32).contiguous(), txt.detach().to(F32).contiguous())
        ctx.grad = any(ctx.needs_input_grad[:2])
        if ctx.grad:
            dI, dT = eng.backward()
            ctx.save_for_backward(dI.clone(), dT.clone())
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g):
        dI, dT = ctx.saved_tensors
        return (dI * g if ctx.needs_input_grad[0] else None, dT * g if ctx.needs_input_grad[1] else None, None)


class ContrastiveLoss:
    def __init__(self, temperature: float = 0.07) -> None:
        self.temperature = temperature
        self._eng = None

    def __call__(self, image_embeddings: torch.Tensor, text_embeddings: torch.Tensor) -> torch.Tensor:
        return _NTXentFn.apply(image_embeddings, text_embeddings, self)

    forward = __call__


def _token_logprobs(logits: torch.Tensor, labels: torch.Tensor, mask: Optional[torch.Tensor]):
    dev = _dev(logits)
    B, S, V = logits.shape
    if mask is None:
        mask = torch.ones_like(labels)
    sb = make_seq_batch(labels, mask, dev, pack=False)
    lg = logits.detach().to(F32).contiguous()
    tok = torch.empty(sb.n_rows, dtype=F32, device=dev)
    hip.logits_logprob(lg, V, V, sb.row_map, sb.targets, sb.n_rows, tok)
    return tok, sb, lg


class _SeqLogProbFn(torch.autograd.Function):
    """Per-sequence log-prob of ``labels[:, 1:]`` under ``logits[:, :-1]`` (sum: components.py:340-362; length-mean:
    model.py:1069-1083) with the gradient w.r.t. the materialised logits (``pgca_logits_logprob_bwd``)."""

    @staticmethod
    def forward(ctx, logits, labels, mask, mode):
        tok, sb, lg = _token_logprobs(logits, labels, mask)
        out = torch.empty(sb.Bq, dtype=F32, device=tok.device)
        hip.seq_reduce(tok, sb.seq_of_row, sb.n_rows, sb.Bq, sb.counts, mode, out)
        if ctx.needs_input_grad[0]:
            ctx.sb, ctx.mode, ctx.dtype = sb, mode, logits.dtype
            ctx.save_for_backward(lg)
        return out

    @staticmethod
    def backward(ctx, g):
        (lg,), sb = ctx.saved_tensors, ctx.sb
        B, S, V = lg.shape
        d = torch.zeros_like(lg)
        if sb.n_rows:
            rs = torch.empty(sb.n_rows, dtype=F32, device=lg.device)
            hip.row_scale(g.to(F32).contiguous(), sb.seq_of_row, sb.counts, sb.n_rows, ctx.mode & 1, rs)
            hip.logits_logprob_bwd(lg, V, V, sb.row_map, sb.targets, rs, sb.n_rows, d)
        return d.to(ctx.dtype), None, None, None


def compute_sequence_logprobs(logits: torch.Tensor, labels: torch.Tensor,
                              attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    return _SeqLogProbFn.apply(logits, labels, attention_mask, 0)


class _DPOFn(torch.autograd.Function):
    """-mean log sigma(beta [(pi_w - pi_l) - (ref_w - ref_l)]) (+ label smoothing) with d/d(pi_w, pi_l) from
    ``pgca_dpo_loss``; the reference log-probs receive the opposite gradients."""

    @staticmethod
    def forward(ctx, pw, pl, rw, rl, beta, ls, metrics):
        dev = _dev(pw)
        f = lambda t: None if t is None else t.detach().to(F32).contiguous()  # noqa: E731
        B = pw.numel()
        loss = torch.empty(1, dtype=F32, device=dev)
        need = any(ctx.needs_input_grad[:4])
        gw = torch.empty(B, dtype=F32, device=dev) if need else None
        gl = torch.empty(B, dtype=F32, device=dev) if need else None
        hip.dpo_loss(f(pw), f(pl), f(rw), f(rl), B, float(beta), float(ls), loss, gw, gl, metrics)
        if need:
            ctx.save_for_backward(gw, gl)
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g):
        gw, gl = ctx.saved_tensors
        n = ctx.needs_input_grad
        return (gw * g if n[0] else None, gl * g if n[1] else None, -gw * g if n[2] else None,
                -gl * g if n[3] else None, None, None, None)


class PreferenceLoss:
    def __init__(self, beta: float = 0.1) -> None:
        self.beta = beta

    def _compute_log_probs(self, logits, labels, mask) -> torch.Tensor:
        return _SeqLogProbFn.apply(logits, labels, mask, 1)

    def __call__(self, preferred_logits, rejected_logits, preferred_labels, rejected_labels, preferred_mask,
                 rejected_mask) -> torch.Tensor:
        lw = self._compute_log_probs(preferred_logits, preferred_labels, preferred_mask)
        ll = self._compute_log_probs(rejected_logits, rejected_labels, rejected_mask)
        return _DPOFn.apply(lw, ll, None, None, self.beta, 0.0, None)

    forward = __call__


class DPOPreferenceLoss:
    def __init__(self, beta: float = 0.1, reference_free: bool = False, label_smoothing: float = 0.0):
        self.beta, self.reference_free, self.label_smoothing = beta, reference_free, label_smoothing

    def __call__(self, policy_chosen_logprobs, policy_rejected_logprobs, reference_chosen_logprobs=None,
                 reference_rejected_logprobs=None) -> Tuple[torch.Tensor, dict]:
        dev = _dev(policy_chosen_logprobs)
        use_ref = not (self.reference_free or reference_chosen_logprobs is None)
        met = torch.empty(4, dtype=F32, device=dev)
        loss = _DPOFn.apply(policy_chosen_logprobs, policy_rejected_logprobs,
                            reference_chosen_logprobs if use_ref else None,
                            reference_rejected_logprobs if use_ref else None, self.beta, self.label_smoothing, met)
        vals = torch.cat([loss.detach().reshape(1), met]).tolist()  # ONE device->host copy (the reference: five .item())
        metrics = {"dpo_loss": vals[0], "reward_margin": vals[1], "reward_accuracy": vals[2],
                   "policy_chosen_logprob": vals[3], "policy_rejected_logprob": vals[4]}
        return loss, metrics

    forward = __call__
