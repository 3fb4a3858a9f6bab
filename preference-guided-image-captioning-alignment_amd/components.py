"""Product-side twins of the reference's standalone loss components (reference models/components.py).

* ``TemperatureScaledSimilarity``  components.py:24-83  (normalises internally, clamps tau to [min_temp, max_temp])
* ``ContrastiveLoss``              components.py:86-145 (NT-Xent on top of it, reduction 'mean' | 'sum')
* ``DPOPreferenceLoss`` / ``compute_sequence_logprobs`` live in ``losses`` (components.py:148-249,321-362) and are
  re-exported here so ``from ...components import X`` reads as in the reference.

These differ from the trainer's ``model.ContrastiveLoss`` (``losses.ContrastiveLoss`` here) exactly as in the
reference: inputs need not be pre-normalised and tau is clamped.  Same HIP kernels underneath (l2norm -> hi/lo split
-> MFMA GEMM with the row-statistics epilogue); device tensors in, device tensors out, no CPU fallback.
"""
from __future__ import annotations

import torch

from . import hip
from .engine import BF16, F32, NTXentEngine, Workspace
from .losses import DPOPreferenceLoss, _dev, compute_sequence_logprobs  # noqa: F401  (re-export)


def _normalised(x: torch.Tensor) -> torch.Tensor:
    x = x.to(F32).contiguous()
    y = torch.empty_like(x)
    hip.l2norm_fwd(x, x.shape[0], x.shape[1], y)
    return y


class TemperatureScaledSimilarity:
    def __init__(self, temperature: float = 0.5, learnable: bool = False, min_temp: float = 0.1,
                 max_temp: float = 2.0):
        if learnable:
            raise NotImplementedError("a learnable temperature is not on the hot path (every shipped config fixes it)")
        self.temperature, self.min_temp, self.max_temp = float(temperature), float(min_temp), float(max_temp)

    @property
    def clamped(self) -> float:
        return min(max(self.temperature, self.min_temp), self.max_temp)

    def __call__(self, vision_embeds: torch.Tensor, text_embeds: torch.Tensor) -> torch.Tensor:
        """[Bv, P], [Bt, P] -> materialised similarity [Bv, Bt] f32 (API compatibility; the loss never builds it)."""
        dev = _dev(vision_embeds)
        v, t = _normalised(vision_embeds), _normalised(text_embeds)
        (Bv, P), Bt = v.shape, t.shape[0]
        a3 = torch.empty(Bv, 3 * P, dtype=BF16, device=dev)
        b3 = torch.empty(Bt, 3 * P, dtype=BF16, device=dev)
        hip.split_bf16(v, Bv, P, Bv, 0, a3)
        hip.split_bf16(t, Bt, P, Bt, 1, b3)
        ldo = (Bt + 3) // 4 * 4
        out = torch.empty(Bv, ldo, dtype=F32, device=dev)
        hip.gemm(a3, b3, Bv, Bt, 3 * P, hip.NT, alpha=1.0 / self.clamped, out_f32=out, ld_out_f32=ldo)
        return out[:, :Bt]

    forward = __call__


class _NormalisedNTXentFn(torch.autograd.Function):
    """components.py:129-145: normalise, similarity / clamped tau, symmetric cross-entropy - with the gradient w.r.t. the
    UN-normalised inputs (NT-Xent backward -> l2-normalise backward) for ``loss.backward()`` callers."""

    @staticmethod
    def forward(ctx, vis, txt, owner):
        dev = _dev(vis)
        v, t = vis.detach().to(F32).contiguous(), txt.detach().to(F32).contiguous()
        (B, P) = v.shape
        if owner._eng is None or owner._eng.ws.device != dev or owner._eng.P != P:
            owner._eng = NTXentEngine(Workspace(dev), P, owner.similarity.clamped, tag="comp.ntx")
        eng = owner._eng
        eng.tau = owner.similarity.clamped
        vn, vnorm = eng.normalize(v, "i")
        tn, tnorm = eng.normalize(t, "t")
        loss, _, _ = eng.forward(vn, tn)
        scale = float(B) if owner.reduction == "sum" else 1.0
        if any(ctx.needs_input_grad[:2]):
            dI, dT = eng.backward(loss_scale=scale)
            dv, dt = torch.empty_like(v), torch.empty_like(t)
            hip.l2norm_bwd(dI, vn, vnorm, B, P, dv)
            hip.l2norm_bwd(dT, tn, tnorm, t.shape[0], P, dt)
            ctx.save_for_backward(dv, dt)
        return loss[0] * scale

    @staticmethod
    def backward(ctx, g):
        dv, dt = ctx.saved_tensors
        return (dv * g if ctx.needs_input_grad[0] else None, dt * g if ctx.needs_input_grad[1] else None, None)


class ContrastiveLoss:
    def __init__(self, temperature: float = 0.5, reduction: str = "mean"):
        if reduction not in ("mean", "sum"):
            raise ValueError(f"unsupported reduction {reduction!r}")
        self.similarity = TemperatureScaledSimilarity(temperature=temperature)
        self.reduction = reduction
        self._eng = None

    def __call__(self, vision_embeds: torch.Tensor, text_embeds: torch.Tensor) -> torch.Tensor:
        return _NormalisedNTXentFn.apply(vision_embeds, text_embeds, self)

    forward = __call__
