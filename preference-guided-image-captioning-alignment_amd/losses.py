"""Loss objects with the reference's call signatures, evaluated by the HIP kernels.

* ``ContrastiveLoss``     reference model.py:957-1000 (the trainer's NT-Xent; inputs already normalised)
* ``PreferenceLoss``      reference model.py:1003-1085 (2-forward, reference-free, length-mean log-prob)
* ``DPOPreferenceLoss``   reference components.py:148-249 (4-forward DPO, metrics dict)
* ``compute_sequence_logprobs``  reference components.py:321-362

These are the API-compatible entry points for callers that already hold embeddings / materialised
logits.  The training hot path does not go through materialised logits: see ``steps.DPOStep``.
Values are plain device tensors (no autograd graph).
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


class ContrastiveLoss:
    def __init__(self, temperature: float = 0.07) -> None:
        self.temperature = temperature
        self._eng = None

    def __call__(self, image_embeddings: torch.Tensor, text_embeddings: torch.Tensor) -> torch.Tensor:
        dev = _dev(image_embeddings)
        if self._eng is None or self._eng.ws.device != dev or self._eng.P != image_embeddings.shape[1]:
            self._eng = NTXentEngine(Workspace(dev), image_embeddings.shape[1], self.temperature, tag="loss.ntx")
        self._eng.tau = float(self.temperature)
        loss, _, _ = self._eng.forward(image_embeddings.to(F32).contiguous(), text_embeddings.to(F32).contiguous())
        return loss[0].clone()

    forward = __call__


def _token_logprobs(logits: torch.Tensor, labels: torch.Tensor, mask: Optional[torch.Tensor]):
    dev = _dev(logits)
    B, S, V = logits.shape
    if mask is None:
        mask = torch.ones_like(labels)
    sb = make_seq_batch(labels, mask, dev)
    lg = logits.to(F32).contiguous()
    tok = torch.empty(sb.n_rows, dtype=F32, device=dev)
    hip.logits_logprob(lg, V, V, sb.row_map, sb.targets, sb.n_rows, tok)
    return tok, sb


def compute_sequence_logprobs(logits: torch.Tensor, labels: torch.Tensor,
                              attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    tok, sb = _token_logprobs(logits, labels, attention_mask)
    out = torch.empty(sb.Bq, dtype=F32, device=tok.device)
    hip.seq_reduce(tok, sb.seq_of_row, sb.n_rows, sb.Bq, sb.counts, 0, out)
    return out


class PreferenceLoss:
    def __init__(self, beta: float = 0.1) -> None:
        self.beta = beta

    def _compute_log_probs(self, logits, labels, mask) -> torch.Tensor:
        tok, sb = _token_logprobs(logits, labels, mask)
        out = torch.empty(sb.Bq, dtype=F32, device=tok.device)
        hip.seq_reduce(tok, sb.seq_of_row, sb.n_rows, sb.Bq, sb.counts, 1, out)
        return out

    def __call__(self, preferred_logits, rejected_logits, preferred_labels, rejected_labels, preferred_mask,
                 rejected_mask) -> torch.Tensor:
        lw = self._compute_log_probs(preferred_logits, preferred_labels, preferred_mask)
        ll = self._compute_log_probs(rejected_logits, rejected_labels, rejected_mask)
        loss = torch.empty(1, dtype=F32, device=lw.device)
        hip.dpo_loss(lw, ll, None, None, lw.numel(), float(self.beta), 0.0, loss)
        return loss[0]

    forward = __call__


class DPOPreferenceLoss:
    def __init__(self, beta: float = 0.1, reference_free: bool = False, label_smoothing: float = 0.0):
        self.beta, self.reference_free, self.label_smoothing = beta, reference_free, label_smoothing

    def __call__(self, policy_chosen_logprobs, policy_rejected_logprobs, reference_chosen_logprobs=None,
                 reference_rejected_logprobs=None) -> Tuple[torch.Tensor, dict]:
        dev = _dev(policy_chosen_logprobs)
        use_ref = not (self.reference_free or reference_chosen_logprobs is None)
        B = policy_chosen_logprobs.numel()
        loss = torch.empty(1, dtype=F32, device=dev)
        met = torch.empty(4, dtype=F32, device=dev)
        f = lambda t: t.to(F32).contiguous()  # noqa: E731
        hip.dpo_loss(f(policy_chosen_logprobs), f(policy_rejected_logprobs),
                     f(reference_chosen_logprobs) if use_ref else None,
                     f(reference_rejected_logprobs) if use_ref else None, B, float(self.beta),
                     float(self.label_smoothing), loss, None, None, met)
        vals = torch.cat([loss, met]).tolist()  # ONE device->host copy (the reference does five .item() calls)
        metrics = {"dpo_loss": vals[0], "reward_margin": vals[1], "reward_accuracy": vals[2],
                   "policy_chosen_logprob": vals[3], "policy_rejected_logprob": vals[4]}
        return loss[0], metrics

    forward = __call__
