"""NT-Xent on the HIP path (SURVEY 8a rows A5 / A5', BASELINE config C3) against the reference's own outputs
(tests/golden/nt_xent.npz) and against the oracle's autograd on the CONCATENATED batch: the
global-negative formulation is new capability whose oracle is reference model.py:984-1000 evaluated
single-process on all N rows.  Ranks are simulated serially on one GPU: rank r calls the engine with its
B local rows, the gathered [N, P] tables and ``offset = r * B`` - exactly what ``steps.ContrastiveStep`` does
after its all-gather."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import restatement as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = torch.from_numpy
LOSS_TOL = 5e-3      # loss |d| <= 5e-3 in bf16 mode (measured: < 1e-4 with the hi/lo split operands)
GRAD_COS = 0.9999


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float(a @ b / (a.norm() * b.norm() + 1e-300))


def _engine(P, tau):
    from pgca_amd.engine import NTXentEngine, Workspace
    return NTXentEngine(Workspace(DEV), P, tau)


def test_fixture_losses_and_gradients(golden):
    """Loss AND dL/dimg, dL/dtxt of the reference's ContrastiveLoss (model.py:984-1000) at B in {2, 8, 64},
    tau in {0.07 (the constructor default, model.py:958), 0.5 (configs/default.yaml:21)}."""
    g = golden("nt_xent")
    for b in (2, 8, 64):
        for tau in (0.07, 0.5):
            k = f"b{b}_t{tau}"
            img, txt = T(g[k + "_img"]).to(DEV).contiguous(), T(g[k + "_txt"]).to(DEV).contiguous()
            eng = _engine(img.shape[1], tau)
            loss, _, _ = eng.forward(img, txt)
            dI, dT = eng.backward()
            assert abs(float(loss) - float(g[k + "_loss"])) <= 2e-4, (k, float(loss), float(g[k + "_loss"]))
            for got, ref, name in ((dI, g[k + "_dimg"], "dimg"), (dT, g[k + "_dtxt"], "dtxt")):
                ref = T(ref)
                got = got.cpu()
                assert _cos(got, ref) >= GRAD_COS, (k, name, _cos(got, ref))
                assert float((got - ref).abs().max()) <= 2e-2 * float(ref.abs().max()) + 1e-7, (k, name)


def _oracle(img, txt, tau):
    img = img.clone().requires_grad_()
    txt = txt.clone().requires_grad_()
    loss = R.nt_xent(img, txt, tau)
    loss.backward()
    # log-sum-exps of the rows of S and of S^t (what the ranks exchange in the backward)
    with torch.no_grad():
        sim = img @ txt.t() / tau
        return float(loss), img.grad, txt.grad, torch.logsumexp(sim, 1), torch.logsumexp(sim, 0)


def _run_ranks(img, txt, tau, world):
    """Serial simulation of ``world`` data-parallel ranks.  Returns (sum of the local loss contributions, dI, dT,
    lse_r, lse_c) assembled over the ranks; gradients are d(global loss)/d(embeddings)."""
    N, P = img.shape
    B = N // world
    ia, ta = img.to(DEV).contiguous(), txt.to(DEV).contiguous()
    eng = _engine(P, tau)
    parts = []
    for r in range(world):                     # forward of every rank first: the backward needs ALL lse's
        loss, lr, lc = eng.forward(ia[r * B:(r + 1) * B], ta[r * B:(r + 1) * B], ia, ta, offset=r * B)
        parts.append((float(loss), lr.clone(), lc.clone()))
    lse_r = torch.cat([p[1] for p in parts])
    lse_c = torch.cat([p[2] for p in parts])
    dI, dT = [], []
    for r in range(world):
        eng.forward(ia[r * B:(r + 1) * B], ta[r * B:(r + 1) * B], ia, ta, offset=r * B)
        di, dt = eng.backward(lse_r, lse_c)
        dI.append(di.clone())
        dT.append(dt.clone())
    return sum(p[0] for p in parts), torch.cat(dI).cpu(), torch.cat(dT).cpu(), lse_r.cpu(), lse_c.cpu()


@pytest.mark.parametrize("world,B,P,tau", [(8, 5, 64, 0.07), (8, 5, 64, 0.5), (2, 96, 512, 0.07), (3, 7, 64, 0.2)])
def test_global_negatives_small(world, B, P, tau):
    """Ragged sizes (N = 40, 21: not multiples of any tile; B = 5, 7: padded K of the gradient GEMMs)."""
    g = torch.Generator().manual_seed(1000 + world * B)
    N = world * B
    img = F.normalize(torch.randn(N, P, generator=g), dim=-1)
    txt = F.normalize(0.6 * img + 0.8 * torch.randn(N, P, generator=g), dim=-1)   # positives are correlated
    loss, gi, gt, lr, lc = _oracle(img, txt, tau)
    hl, hi, ht, hlr, hlc = _run_ranks(img, txt, tau, world)
    assert abs(hl - loss) <= 2e-4, (hl, loss)
    np.testing.assert_allclose(hlr.numpy(), lr.numpy(), atol=5e-4)
    np.testing.assert_allclose(hlc.numpy(), lc.numpy(), atol=5e-4)
    assert _cos(hi, gi) >= GRAD_COS and _cos(ht, gt) >= GRAD_COS
    assert float((hi - gi).abs().max()) <= 2e-2 * float(gi.abs().max())
    assert float((ht - gt).abs().max()) <= 2e-2 * float(gt.abs().max())


@pytest.mark.parametrize("tau", [0.07, 0.5])
def test_global_negatives_config3_scale(tau):
    """BASELINE config C3: global batch 8192 = 8 ranks x 1024, P = 512.  Every rank's call (offset = r * 1024,
    targetsN = -1 columns, rectangular DLOGITS GEMMs) against R.nt_xent autograd on the concatenated batch."""
    world, B, P = 8, 1024, 512
    N = world * B
    g = torch.Generator().manual_seed(8192)
    img = F.normalize(torch.randn(N, P, generator=g), dim=-1)
    txt = F.normalize(0.5 * img + 0.87 * torch.randn(N, P, generator=g), dim=-1)
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    loss, gi, gt, lr, lc = _oracle(img, txt, tau)
    hl, hi, ht, hlr, hlc = _run_ranks(img, txt, tau, world)
    assert abs(hl - loss) <= 5e-4, (hl, loss)
    assert abs(hl - loss) <= LOSS_TOL
    np.testing.assert_allclose(hlr.numpy(), lr.numpy(), atol=1e-3)
    np.testing.assert_allclose(hlc.numpy(), lc.numpy(), atol=1e-3)
    for r in range(world):                                       # per-rank slices, so one bad offset cannot hide
        sl = slice(r * B, (r + 1) * B)
        assert _cos(hi[sl], gi[sl]) >= GRAD_COS, (r, _cos(hi[sl], gi[sl]))
        assert _cos(ht[sl], gt[sl]) >= GRAD_COS, (r, _cos(ht[sl], gt[sl]))
    assert abs(float(hi.norm()) / float(gi.norm()) - 1) <= 2e-3
    assert abs(float(ht.norm()) / float(gt.norm()) - 1) <= 2e-3


def test_components_contrastive_loss_product_side():
    """A5': components.ContrastiveLoss / TemperatureScaledSimilarity (components.py:61-145): internal normalise and
    tau clamp to [0.1, 2.0]."""
    from pgca_amd.components import ContrastiveLoss as CLoss, TemperatureScaledSimilarity
    g = torch.Generator().manual_seed(3)
    v, t = torch.randn(12, 64, generator=g) * 3, torch.randn(12, 64, generator=g) * 0.2
    for tau in (0.01, 0.07, 0.5, 5.0):
        want = float(R.nt_xent_components(v, t, tau))
        got = float(CLoss(temperature=tau)(v.to(DEV), t.to(DEV)))
        assert abs(got - want) <= 2e-4, (tau, got, want)
    sim = TemperatureScaledSimilarity(temperature=0.01)(v.to(DEV), t.to(DEV)).cpu()
    ref = F.normalize(v, dim=-1) @ F.normalize(t, dim=-1).t() / 0.1
    np.testing.assert_allclose(sim.numpy(), ref.numpy(), atol=2e-4)


def cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def test_learnable_temperature_gradient_and_clamp():
    """components.py:48-57,73-83: ``learnable=True`` makes tau an nn.Parameter; its gradient through
    ``loss = NT-Xent(cos / clamp(tau))`` against torch autograd on the host; outside the clamp range the gradient is 0."""
    from pgca_amd.components import ContrastiveLoss as CLoss, TemperatureScaledSimilarity
    g = torch.Generator().manual_seed(9)
    v, t = torch.randn(10, 64, generator=g), torch.randn(10, 64, generator=g)
    for tau0, reduction in ((0.5, "mean"), (0.3, "sum"), (0.05, "mean")):
        loss_fn = CLoss(temperature=tau0, reduction=reduction)
        loss_fn.similarity = TemperatureScaledSimilarity(temperature=tau0, learnable=True).to(DEV)
        vd, td = v.to(DEV).requires_grad_(), t.to(DEV).requires_grad_()
        loss = loss_fn(vd, td)
        loss.backward()
        tau = torch.tensor(tau0, requires_grad=True)
        vh, th = v.clone().requires_grad_(), t.clone().requires_grad_()
        sim = F.normalize(vh, dim=-1) @ F.normalize(th, dim=-1).t() / torch.clamp(tau, 0.1, 2.0)
        lab = torch.arange(10)
        want = (F.cross_entropy(sim, lab) + F.cross_entropy(sim.t(), lab)) / 2 * (10.0 if reduction == "sum" else 1.0)
        want.backward()
        assert abs(float(loss) - float(want)) <= 2e-4 * max(1.0, abs(float(want)))
        gt = float(loss_fn.similarity.temperature.grad)
        assert abs(gt - float(tau.grad)) <= 2e-3 * max(1e-3, abs(float(tau.grad))) + 1e-6, (tau0, gt, float(tau.grad))
        assert cos(vd.grad.cpu(), vh.grad) >= 0.9999
    sim = TemperatureScaledSimilarity(temperature=0.7, learnable=True).to(DEV)
    out = sim(v.to(DEV), t.to(DEV))
    out.sum().backward()
    ref_t = torch.tensor(0.7, requires_grad=True)
    (F.normalize(v, dim=-1) @ F.normalize(t, dim=-1).t() / ref_t).sum().backward()
    assert abs(float(sim.temperature.grad) - float(ref_t.grad)) <= 1e-3 * abs(float(ref_t.grad))


def test_nan_safe_gradient_norm_surface():
    """Reference NaNSafeGradientNorm (components.py:252-318): (total_norm, is_finite); clips only when finite."""
    from pgca_amd.components import NaNSafeGradientNorm
    g = torch.Generator().manual_seed(4)
    ps = [torch.nn.Parameter(torch.zeros(n, device=DEV)) for n in (1000, 37, 70001)]
    grads = [torch.randn(p.numel(), generator=g) * 3 for p in ps]
    for p, gr in zip(ps, grads):
        p.grad = gr.to(DEV).clone()
    total, ok = NaNSafeGradientNorm(max_norm=1.0)(ps)
    want = torch.sqrt(sum((gr.double() ** 2).sum() for gr in grads))
    assert ok and abs(float(total) - float(want)) <= 1e-4 * float(want)
    after = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in ps))
    assert abs(float(after) - 1.0) <= 1e-4
    ps[1].grad[3] = float("nan")
    before = ps[0].grad.clone()
    total, ok = NaNSafeGradientNorm(max_norm=1.0)(ps)
    assert not ok and not bool(torch.isfinite(total)) and torch.equal(ps[0].grad, before)      # untouched
    with pytest.raises(RuntimeError, match="Non-finite"):
        NaNSafeGradientNorm(max_norm=1.0, error_if_nonfinite=True)(ps)
    assert NaNSafeGradientNorm()([torch.nn.Parameter(torch.zeros(3))]) [1] is True                # nothing has a gradient
