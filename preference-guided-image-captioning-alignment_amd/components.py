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
    """``learnable=True`` (components.py:48-57): ``temperature`` is a 0-d ``torch.nn.Parameter`` (put it in an optimiser like
    any parameter); ``ContrastiveLoss`` then also returns its gradient.  Reading its value costs one host read per call -
    the trainer's hot path fixes tau (every shipped config) and never goes through here."""

    def __init__(self, temperature: float = 0.5, learnable: bool = False, min_temp: float = 0.1,
                 max_temp: float = 2.0):
        self.learnable = bool(learnable)
        self.temperature = torch.nn.Parameter(torch.tensor(float(temperature))) if learnable else float(temperature)
        self.min_temp, self.max_temp = float(min_temp), float(max_temp)

    def parameters(self):
        return [self.temperature] if self.learnable else []

    def to(self, device):
        if self.learnable:
            self.temperature.data = self.temperature.data.to(device)
        return self

    @property
    def clamped(self) -> float:
        return min(max(float(self.temperature), self.min_temp), self.max_temp)

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
        if self.learnable:     # cosines from the kernel, the division in torch so that autograd reaches the parameter
            hip.gemm(a3, b3, Bv, Bt, 3 * P, hip.NT, alpha=1.0, out_f32=out, ld_out_f32=ldo)
            return out[:, :Bt] / torch.clamp(self.temperature.to(dev), self.min_temp, self.max_temp)
        hip.gemm(a3, b3, Bv, Bt, 3 * P, hip.NT, alpha=1.0 / self.clamped, out_f32=out, ld_out_f32=ldo)
        return out[:, :Bt]

    forward = __call__


class _NormalisedNTXentFn(torch.autograd.Function):
    """components.py:129-145: normalise, similarity / clamped tau, symmetric cross-entropy - with the gradient w.r.t. the
    UN-normalised inputs (NT-Xent backward -> l2-normalise backward) for ``loss.backward()`` callers."""

    @staticmethod
    def forward(ctx, vis, txt, tau_param, owner):
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
        ctx.tau_device = tau_param.device if tau_param is not None else dev
        scale = float(B) if owner.reduction == "sum" else 1.0
        if any(ctx.needs_input_grad[:3]):
            dI, dT = eng.backward(loss_scale=scale)
            dv, dt = torch.empty_like(v), torch.empty_like(t)
            hip.l2norm_bwd(dI, vn, vnorm, B, P, dv)
            hip.l2norm_bwd(dT, tn, tnorm, t.shape[0], P, dt)
            # d loss / d tau: the loss sees tau only through S = cos / tau, and sum_ij (dL/dS_ij) S_ij = sum_i <dL/dI_i, I_i>
            # with I the normalised embeddings, so dL/dtau = -(1/tau) sum_i <dI_i, I_i>; clamp passes it inside its range
            tau = owner.similarity.clamped
            raw = float(owner.similarity.temperature)
            inside = owner.similarity.min_temp <= raw <= owner.similarity.max_temp
            dtau = -(dI * vn).sum() / tau if inside else torch.zeros((), device=dev)
            ctx.save_for_backward(dv, dt, dtau)
        return loss[0] * scale

    @staticmethod
    def backward(ctx, g):
        dv, dt, dtau = ctx.saved_tensors
        return (dv * g if ctx.needs_input_grad[0] else None, dt * g if ctx.needs_input_grad[1] else None,
                (dtau * g).to(ctx.tau_device) if ctx.needs_input_grad[2] else None, None)


class ContrastiveLoss:
    def __init__(self, temperature: float = 0.5, reduction: str = "mean"):
        if reduction not in ("mean", "sum"):
            raise ValueError(f"unsupported reduction {reduction!r}")
        self.similarity = TemperatureScaledSimilarity(temperature=temperature)
        self.reduction = reduction
        self._eng = None

    def __call__(self, vision_embeds: torch.Tensor, text_embeds: torch.Tensor) -> torch.Tensor:
        tau = self.similarity.temperature if self.similarity.learnable else None
        return _NormalisedNTXentFn.apply(vision_embeds, text_embeds, tau, self)

    forward = __call__


class NaNSafeGradientNorm:
    """Reference ``NaNSafeGradientNorm`` (components.py:252-318): global L2 gradient norm, a finite check, and - only
    when the norm is finite - ``clip_grad_norm_`` to ``max_norm``.  ``parameters``: anything with a ``.grad`` tensor on the
    device (``torch.nn.Parameter``s, or this build's flat ``params.Segment``s).  Same kernels as the trainer's fused
    optimiser (``pgca_sqnorm`` / ``pgca_clip_coef`` / ``pgca_scale_dev``); returns ``(total_norm, is_finite)`` like the
    reference, which costs the one host read the reference's ``.item()`` costs too."""

    def __init__(self, max_norm: float = 1.0, norm_type: float = 2.0, error_if_nonfinite: bool = False):
        if float(norm_type) != 2.0:
            raise NotImplementedError("only the L2 norm the reference uses (norm_type=2.0) is provided")
        self.max_norm, self.norm_type, self.error_if_nonfinite = float(max_norm), 2.0, bool(error_if_nonfinite)

    def __call__(self, parameters):
        grads = [p.grad for p in parameters if getattr(p, "grad", None) is not None]
        if not grads:
            return torch.tensor(0.0), True
        dev = grads[0].device
        flats = [g if (g.is_contiguous() and g.dtype == F32) else None for g in grads]
        if any(f is None for f in flats):
            raise ValueError("gradients must be contiguous float32 device tensors")
        nb = [hip.sqnorm_blocks(g.numel()) for g in grads]
        part = torch.zeros(sum(nb), dtype=F32, device=dev)
        off = 0
        for g, n in zip(grads, nb):
            hip.sqnorm(g.view(-1), g.numel(), part[off:off + n])
            off += n
        coef = torch.empty(2, dtype=F32, device=dev)
        hip.clip_coef(part, part.numel(), self.max_norm, coef)          # [clip factor, norm]
        total = coef[1].clone()
        finite = bool(torch.isfinite(total))
        if not finite:
            if self.error_if_nonfinite:
                raise RuntimeError("Non-finite gradient norm detected")
            return total, False
        for g in grads:
            if g.numel() % 4 == 0 and g.data_ptr() % 16 == 0:
                hip.scale_dev(g.view(-1), g.numel(), coef)
            else:                                   # odd-sized tensors: the same device-side factor, no read-back
                g.mul_(coef[0])
        return total, True

    forward = __call__
