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


class ContrastiveLoss:
    def __init__(self, temperature: float = 0.5, reduction: str = "mean"):
        if reduction not in ("mean", "sum"):
            raise ValueError(f"unsupported reduction {reduction!r}")
        self.similarity = TemperatureScaledSimilarity(temperature=temperature)
        self.reduction = reduction
        self._eng = None

    def __call__(self, vision_embeds: torch.Tensor, text_embeds: torch.Tensor) -> torch.Tensor:
        dev = _dev(vision_embeds)
        v, t = _normalised(vision_embeds), _normalised(text_embeds)
        if self._eng is None or self._eng.ws.device != dev or self._eng.P != v.shape[1]:
            self._eng = NTXentEngine(Workspace(dev), v.shape[1], self.similarity.clamped, tag="comp.ntx")
        self._eng.tau = self.similarity.clamped
        loss, _, _ = self._eng.forward(v, t)
        out = loss[0].clone()
        return out * v.shape[0] if self.reduction == "sum" else out

    forward = __call__
