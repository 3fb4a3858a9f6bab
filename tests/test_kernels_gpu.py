"""HIP kernels (through the C ABI) against plain fp32 PyTorch / the oracle restatement.

Tolerances: bf16 operands are exact in fp32, accumulation is fp32 in MFMA, so f32 outputs
are compared at 2e-4 relative to the output scale; bf16 outputs at one bf16 ulp (2^-8
relative).  Index work (targets, row maps) is bit-exact.
"""
import math

import numpy as np
import pytest
import torch

from oracle import restatement as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from pgca_amd import hip as H
    H.load()
    return H


def dev():
    return torch.device("cuda:0")


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev())


def close(a, b, rel, what=""):
    a, b = a.float(), b.float()
    scale = float(b.abs().max()) + 1e-12
    err = float((a - b).abs().max())
    assert err <= rel * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.3e} > {rel})"


def operands(layout, M, N, K, seed):
    a = rnd(M, K, seed=seed).bfloat16()
    b = rnd(K, N, seed=seed + 1).bfloat16()  # math: C = a @ b
    A = a.contiguous() if layout != 2 else a.t().contiguous()          # TN: stored [K, M]
    B = b.t().contiguous() if layout == 0 else b.contiguous()          # NT: stored [N, K]
    return a, b, A, B


SHAPES = [(128, 128, 64), (256, 384, 1024), (200, 136, 72), (77, 1000, 128), (1, 8, 8), (130, 264, 200),
          (512, 768, 256), (300, 520, 192), (1000, 1000, 64)]


@pytest.fixture(params=[128, 256, 262, 263])
def tile(request, hip):
    """Force the 128^2 register-staged kernel, the 256^2 LDS-DMA kernel with the 2-stage BK=64 loop (256), its
    phase-staggered 4-stage BK=32 loop (262), or the latter with the start stagger of the first workgroups on (263).
    The LDS-DMA kernels need K % 64 == 0."""
    hip.set_option("gemm_tile", 128 if request.param == 128 else 256)
    hip.set_option("gemm_schedule", {262: 6, 263: 6}.get(request.param, 0))
    hip.set_option("gemm_stagger", 2 if request.param == 263 else 0)
    yield 256 if request.param > 256 else request.param
    hip.set_option("gemm_tile", 0)
    hip.set_option("gemm_schedule", -1)
    hip.set_option("gemm_stagger", 0)


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("shape", SHAPES)
def test_gemm_layouts_and_edges(hip, tile, layout, shape):
    M, N, K = shape
    if tile == 256 and (K % 64 or M < 8):
        pytest.skip("256^2 kernel is only dispatched for K % 64 == 0")
    if layout == 2 and M % 8:   # TN: A is [K, M], row stride must be a multiple of 8 elements
        ldm = (M + 7) // 8 * 8
    else:
        ldm = M
    a, b, A, B = operands(layout, M, N, K, seed=layout * 100 + M)
    ldb = None
    if layout == 2 and ldm != M:
        A = torch.zeros(K, ldm, dtype=torch.bfloat16, device=dev())
        A[:, :M] = a.t()
    if layout != 0 and N % 8:
        ldn = (N + 7) // 8 * 8
        Bp = torch.zeros(K, ldn, dtype=torch.bfloat16, device=dev())
        Bp[:, :N] = b
        B, ldb = Bp, ldn
    if layout != 2 and K % 8:
        pytest.skip("K-contiguous operands need K % 8 == 0")
    bias = rnd(N, seed=5)
    out_f = torch.full((M, N), float("nan"), device=dev())
    out_b = torch.zeros(M, N, dtype=torch.bfloat16, device=dev())
    hip.gemm(A, B, M, N, K, layout, lda=(ldm if layout == 2 else None), ldb=ldb, bias=bias, out_f32=out_f,
             out_bf16=out_b)
    ref = a.float() @ b.float() + bias
    close(out_f, ref, 2e-4, "f32 out")
    close(out_b, ref, 1.0 / 128, "bf16 out")


def test_gemm_asymmetric_identity(hip, tile):
    """A = I with asymmetric B catches a swapped row/col map in the C write."""
    n = 128 if tile == 128 else 512
    eye = torch.eye(n, device=dev()).bfloat16()
    b = (torch.arange(n * n, device=dev()).float().view(n, n) % 251 - 125).bfloat16()
    for layout in (0, 1, 2):
        out = torch.zeros(n, n, device=dev())
        B = b.t().contiguous() if layout == 0 else b
        hip.gemm(eye, B, n, n, n, layout, out_f32=out)
        assert torch.equal(out, b.float()), f"layout {layout}"


@pytest.mark.parametrize("epi", ["gelu", "quick_gelu", "relu", "tanh"])
def test_gemm_activation_epilogues(hip, tile, epi):
    M, N, K = 192, 320, 256
    a, b, A, B = operands(1, M, N, K, seed=3)
    bias = rnd(N, seed=9)
    pre = a.float() @ b.float() + bias
    code = {"gelu": hip.EPI_GELU_NEW, "quick_gelu": hip.EPI_QUICK_GELU, "relu": hip.EPI_RELU, "tanh": hip.EPI_TANH}[epi]
    ref = {"gelu": R.gelu_new, "quick_gelu": R.quick_gelu, "relu": torch.relu, "tanh": torch.tanh}[epi](pre)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev())
    aux = torch.zeros(M, N, dtype=torch.bfloat16, device=dev())
    hip.gemm(A, B, M, N, K, 1, epilogue=code, bias=bias, out_bf16=out, aux_out=aux if "gelu" in epi else None)
    close(out, ref, 1.0 / 100, epi)
    if "gelu" in epi:
        close(aux, pre, 1.0 / 128, "pre-activation")


def test_gemm_derivative_epilogues_residual_accumulate(hip, tile):
    M, N, K = 160, 256, 128
    a, b, A, B = operands(0, M, N, K, seed=21)
    acc = a.float() @ b.float()
    pre = rnd(M, N, seed=4).bfloat16()
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev())
    hip.gemm(A, B, M, N, K, 0, epilogue=hip.EPI_DGELU_NEW, aux_in=pre, out_bf16=out)
    x = pre.float().requires_grad_()
    R.gelu_new(x).backward(acc)
    close(out, x.grad, 1.0 / 100, "dgelu")
    act = torch.relu(pre)
    hip.gemm(A, B, M, N, K, 0, epilogue=hip.EPI_DRELU, aux_in=act, out_bf16=out)
    close(out, acc * (act.float() > 0), 1.0 / 128, "drelu")
    th = torch.tanh(pre.float()).bfloat16()
    hip.gemm(A, B, M, N, K, 0, epilogue=hip.EPI_DTANH, aux_in=th, out_bf16=out)
    close(out, acc * (1 - th.float() ** 2), 1.0 / 128, "dtanh")
    # residual add (in place on the f32 stream) and gradient accumulation
    res = rnd(M, N, seed=6)
    stream = res.clone()
    hip.gemm(A, B, M, N, K, 0, residual=stream, out_f32=stream)
    close(stream, acc + res, 2e-4, "residual in place")
    hip.gemm(A, B, M, N, K, 0, out_f32=stream, accumulate=True, alpha=0.5)
    close(stream, 1.5 * acc + res, 2e-4, "accumulate")


def test_lm_head_rowstats_and_dlogits(hip, tile):
    """Fused LM head: log-prob gather without logits in HBM (model.py:1069-1079), and the
    recomputed dlogits = g * (softmax - onehot)."""
    M, V, K = 150, 1004, 128          # V not a multiple of 8 or of the 128 tile
    Vp = (V + 127) // 128 * 128
    h = rnd(M, K, seed=1, scale=0.5).bfloat16()
    wte = torch.zeros(Vp, K, dtype=torch.bfloat16, device=dev())
    wte[:V] = rnd(V, K, seed=2, scale=0.5).bfloat16()
    tgt = torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(3)).to(dev())
    tgt[0], tgt[1] = 0, V - 1
    nparts = 2 * (Vp // 128)
    smax = torch.zeros(M, nparts, device=dev())
    ssum = torch.zeros(M, nparts, device=dev())
    tval = torch.zeros(M, device=dev())
    hip.gemm(h, wte, M, V, K, 0, epilogue=hip.EPI_ROWSTATS, targets=tgt, stat_max=smax, stat_sum=ssum,
             stat_ld=nparts, target_val=tval)
    lse = torch.zeros(M, device=dev())
    lp = torch.zeros(M, device=dev())
    hip.rowstats_combine(smax, ssum, nparts, 2 * ((V + 127) // 128), tval, M, lse=lse, out_logprob=lp)
    logits = h.float() @ wte[:V].float().t()
    ref_lp = torch.log_softmax(logits, -1).gather(-1, tgt[:, None]).squeeze(-1)
    assert float((lp - ref_lp).abs().max()) <= 2e-4
    assert float((lse - torch.logsumexp(logits, -1)).abs().max()) <= 2e-4
    g = rnd(M, seed=8)
    dl = torch.full((M, Vp), 7.0, dtype=torch.bfloat16, device=dev())
    hip.gemm(h, wte, M, V, K, 0, epilogue=hip.EPI_DLOGITS, targets=tgt, row_lse=lse, row_scale=g, out_bf16=dl,
             ld_out_bf16=Vp, out_cols=Vp)
    ref = g[:, None] * (torch.softmax(logits, -1) - torch.nn.functional.one_hot(tgt, V).float())
    close(dl[:, :V], ref, 1.0 / 100, "dlogits")
    assert float(dl[:, V:].abs().max()) == 0.0  # padding columns are written as zeros


@pytest.mark.parametrize("H", [64, 512, 768, 1024, 1280, 1600])
def test_layernorm_fwd_bwd(hip, H):
    M = 37
    x = rnd(M + 5, H, seed=H)
    gamma, beta = rnd(H, seed=1) * 0.1 + 1, rnd(H, seed=2) * 0.1
    rmap = torch.randperm(M + 5, generator=torch.Generator().manual_seed(0))[:M].int().to(dev())
    yb = torch.zeros(M, H, dtype=torch.bfloat16, device=dev())
    yf = torch.zeros(M, H, device=dev())
    mean, rstd = torch.zeros(M, device=dev()), torch.zeros(M, device=dev())
    hip.layernorm_fwd(x, M, H, gamma, beta, row_map=rmap, y_bf16=yb, y_f32=yf, mean=mean, rstd=rstd)
    xs = x[rmap.long()].clone().requires_grad_()
    gp, bp = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    ref = torch.nn.functional.layer_norm(xs, (H,), gp, bp, 1e-5)
    close(yf, ref, 1e-5, "ln fwd")
    close(yb, ref, 1.0 / 128, "ln fwd bf16")
    dy = rnd(M, H, seed=3)
    ref.backward(dy)
    add = rnd(M + 5, H, seed=4)
    dx = torch.zeros(M + 5, H, device=dev())
    nb = hip.layernorm_bwd_blocks(M)
    part = torch.zeros(2, nb, H, device=dev())
    hip.layernorm_bwd(x, M, H, gamma, mean, rstd, dx, dy_f32=dy, row_map=rmap, add_to=add, part=part)
    close(dx[rmap.long()] - add[rmap.long()], xs.grad, 2e-5, "ln dx")
    dg, db = torch.zeros(H, device=dev()), torch.ones(H, device=dev())
    hip.colsum_finish(part[0], nb, H, dg)
    hip.colsum_finish(part[1], nb, H, db, accumulate=True)
    close(dg, gp.grad, 2e-5, "dgamma")
    close(db - 1, bp.grad, 2e-5, "dbeta")
    # bf16 dy path + bf16 copy of dx
    dxb = torch.zeros(M + 5, H, dtype=torch.bfloat16, device=dev())
    hip.layernorm_bwd(x, M, H, gamma, mean, rstd, dx, dy_bf16=dy.bfloat16(), row_map=rmap, dx_bf16=dxb)
    close(dxb[rmap.long()], xs.grad, 1.0 / 50, "ln dx (bf16 dy)")


def test_colsum(hip):
    M, N = 300, 1000
    x = rnd(M, N, seed=1)
    nb = hip.colsum_blocks(M)
    part = torch.zeros(nb, N, device=dev())
    hip.colsum(M, N, N, part, x_bf16=x.bfloat16())
    out = torch.zeros(N, device=dev())
    hip.colsum_finish(part, nb, N, out)
    close(out, x.bfloat16().float().sum(0), 1e-5, "colsum bf16")
    hip.colsum(M, N, N, part, x_f32=x)
    hip.colsum_finish(part, nb, N, out)
    close(out, x.sum(0), 1e-5, "colsum f32")


def attn_ref(qkv, mask, B, S, heads, causal):
    H = heads * 64
    q, k, v = qkv.float().view(B, S, 3, heads, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * 0.125
    allowed = torch.ones(S, S, dtype=torch.bool, device=qkv.device)
    if causal:
        allowed = torch.tril(allowed)
    allowed = allowed[None, None] & (mask[:, None, None, :] != 0)
    s = s.masked_fill(~allowed, float("-inf"))
    p = torch.softmax(s, -1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B * S, H), torch.logsumexp(s, -1)


@pytest.mark.parametrize("cfg", [(3, 128, 2, True), (2, 50, 3, False), (4, 37, 1, True), (2, 16, 2, True)])
def test_attention_fwd_bwd(hip, cfg):
    B, S, heads, causal = cfg
    H = heads * 64
    qkv = rnd(B * S, 3 * H, seed=S).bfloat16()
    lens = torch.tensor([S, max(1, S // 2), max(1, S - 3), 1][:B])
    mask = (torch.arange(S)[None] < lens[:, None]).int().to(dev())
    if not causal:
        mask = torch.ones_like(mask)
    out = torch.zeros(B * S, H, dtype=torch.bfloat16, device=dev())
    lse = torch.zeros(B, heads, S, device=dev())
    hip.attention_fwd(qkv, mask, B, S, heads, causal, out, lse)
    x = qkv.float().requires_grad_()
    ref, ref_lse = attn_ref(x, mask, B, S, heads, causal)
    close(out, ref, 1.0 / 64, "attn out")
    assert float((lse - ref_lse).abs().max()) <= 2e-3
    # backward only where the loss can reach: rows of valid queries
    valid = (mask.view(B * S, 1) != 0).float()
    dout = (rnd(B * S, H, seed=7) * valid).bfloat16()
    ref.backward(dout.float())
    dqkv = torch.zeros(B * S, 3 * H, dtype=torch.bfloat16, device=dev())
    hip.attention_bwd(qkv, out, dout, lse, mask, B, S, heads, causal, dqkv)
    close(dqkv, x.grad, 1.0 / 40, "attn dqkv")
    # padded keys receive exactly zero gradient
    pad_rows = (mask.view(-1) == 0)
    if bool(pad_rows.any()):
        assert float(dqkv[pad_rows][:, H:].abs().max()) == 0.0


def test_embed_fwd_bwd(hip):
    B, S, H, V = 3, 9, 128, 50
    ids = torch.randint(0, V, (B, S), generator=torch.Generator().manual_seed(0)).to(dev())
    ids[0, :3] = 7  # repeated ids -> atomics collide
    wte, wpe = rnd(V, H, seed=1), rnd(16, H, seed=2)
    att = rnd(B, H, seed=3)
    gamma, beta = rnd(H, seed=4) * 0.1 + 1, rnd(H, seed=5) * 0.1
    h0 = torch.zeros(B * S, H, device=dev())
    mean, rstd = torch.zeros(B * S, device=dev()), torch.zeros(B * S, device=dev())
    hip.embed_fwd(ids, B, S, H, wte, wpe, h0, attended=att, gamma=gamma, beta=beta, mean=mean, rstd=rstd)
    wr, pr, ar = wte.clone().requires_grad_(), wpe.clone().requires_grad_(), att.clone().requires_grad_()
    gr, br = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    ref = torch.nn.functional.layer_norm(wr[ids] + ar[:, None, :], (H,), gr, br, 1e-5) + pr[:S][None]
    close(h0, ref.view(B * S, H), 1e-5, "embed fwd")
    rmask = torch.ones(B, S, dtype=torch.int32, device=dev())
    rmask[1, 5:] = 0
    g = rnd(B * S, H, seed=6) * rmask.view(-1, 1)
    ref.view(B * S, H).backward(g)
    dwte, dwpe, datt = torch.zeros_like(wte), torch.zeros_like(wpe), torch.zeros_like(att)
    nb = hip.embed_bwd_blocks(B, S)
    part = torch.zeros(2, nb, H, device=dev())
    hip.embed_bwd(g, ids, rmask, B, S, H, dwte, dwpe, wte=wte, attended=att, gamma=gamma, mean=mean, rstd=rstd,
                  dattended=datt, part=part)
    close(dwte, wr.grad, 2e-5, "dwte")
    close(dwpe, pr.grad, 2e-5, "dwpe")
    close(datt, ar.grad, 2e-5, "dattended")
    dg = torch.zeros(H, device=dev())
    hip.colsum_finish(part[0], nb, H, dg)
    close(dg, gr.grad, 2e-5, "embed dgamma")
    # text-tower form: no cross-attention, no LN
    hip.embed_fwd(ids, B, S, H, wte, wpe, h0)
    close(h0, (wte[ids] + wpe[:S][None]).view(B * S, H), 1e-6, "embed fwd (text)")
    dwte.zero_(); dwpe.zero_()
    hip.embed_bwd(g, ids, rmask, B, S, H, dwte, dwpe)
    ref_dwte = torch.zeros_like(wte).index_add_(0, ids.view(-1), g)
    close(dwte, ref_dwte, 1e-5, "dwte (text)")


def test_patchify_and_assemble(hip):
    B, I, P, Hd = 2, 64, 32, 128
    px = rnd(B, 3, I, I, seed=1)
    G = I // P
    out = torch.zeros(B * G * G, 3 * P * P, dtype=torch.bfloat16, device=dev())
    hip.patchify(px, B, I, P, out)
    ref = px.reshape(B, 3, G, P, G, P).permute(0, 2, 4, 1, 3, 5).reshape(B * G * G, 3 * P * P)
    assert torch.equal(out, ref.bfloat16())
    pe, cls, pos = rnd(B * G * G, Hd, seed=2), rnd(Hd, seed=3), rnd(G * G + 1, Hd, seed=4)
    x = torch.zeros(B, G * G + 1, Hd, device=dev())
    hip.vit_assemble(pe, cls, pos, B, G * G + 1, Hd, x)
    ref = torch.cat([cls.expand(B, 1, Hd), pe.view(B, G * G, Hd)], 1) + pos[None]
    assert torch.equal(x, ref)


def test_patchify_patch14_pads_k(hip):
    """CLIP ViT-L/14: 3*14*14 = 588 is not a multiple of 8; the gather pads K to 640 with zero columns."""
    B, I, P, ld = 2, 224, 14, 640
    px = rnd(B, 3, I, I, seed=3)
    G = I // P
    out = torch.full((B * G * G, ld), 5.0, dtype=torch.bfloat16, device=dev())
    hip.patchify(px, B, I, P, out, ld_out=ld)
    ref = px.reshape(B, 3, G, P, G, P).permute(0, 2, 4, 1, 3, 5).reshape(B * G * G, 3 * P * P)
    assert torch.equal(out[:, :588], ref.bfloat16())
    assert float(out[:, 588:].abs().max()) == 0.0


def test_seq_batch_prepare_is_bit_exact(hip, golden):
    """The index work of the log-prob gather on the device (pgca_seq_batch_prepare) against the reference's own
    gather indices (labels[:, 1:], model.py:1070 / components.py:341) - int64 targets, row map, per-sequence counts -
    on the fixture's ragged masks, on masks with holes, on captions with a single real token, S = 2 and a large batch."""
    from pgca_amd.engine import make_seq_batch
    g = golden("logprob_dpo")
    cases = [(torch.from_numpy(g["ids_w"]), torch.from_numpy(g["mask_w"]))]
    gen = torch.Generator().manual_seed(3)
    for Bq, S in ((7, 2), (33, 129), (512, 128), (5, 300)):
        ids = torch.randint(0, 50257, (Bq, S), generator=gen)
        mask = (torch.rand(Bq, S, generator=gen) < 0.7).long()          # holes, not only right padding
        mask[0] = 1
        if Bq > 2:
            mask[1] = 0
            mask[1, 0] = 1                                               # a caption with one real token: count 0
        cases.append((ids, mask))
    for ids, mask in cases:
        sb = make_seq_batch(ids, mask, dev())
        keep = mask[:, 1:] != 0
        assert sb.targets.dtype == torch.int64 and torch.equal(sb.targets.cpu(), R.gather_indices(ids)[keep])
        assert torch.equal(sb.counts.cpu().long(), keep.sum(1))
        b, t = torch.nonzero(keep, as_tuple=True)
        assert sb.n_rows == int(keep.sum())
        assert torch.equal(sb.row_map.cpu().long(), b * ids.shape[1] + t) and torch.equal(sb.seq_of_row.cpu().long(), b)
        assert torch.equal(sb.mask.cpu().long(), (mask != 0).long()) and torch.equal(sb.ids.cpu(), ids)
    ids, mask = cases[0]
    sb = make_seq_batch(ids, mask, dev())
    logits = torch.from_numpy(g["logits_w"])
    lp = torch.log_softmax(logits.view(-1, logits.shape[-1])[sb.row_map.cpu().long()], -1).gather(1, sb.targets.cpu()[:, None])[:, 0]
    seq = torch.zeros(ids.shape[0]).index_add_(0, sb.seq_of_row.cpu().long(), lp)
    np.testing.assert_allclose(seq.numpy(), g["seq_sum_w"], atol=2e-5)
    with pytest.raises(ValueError, match="no scored token"):
        make_seq_batch(ids[:1], torch.tensor([[1] + [0] * (ids.shape[1] - 1)]), dev())


def test_prefetcher_feeds_prepared_batches_in_order(hip):
    """input.BatchPrefetcher: same prepared batches as the inline path, in loader order, errors re-raised."""
    from pgca_amd.input import BatchPrefetcher
    from pgca_amd.steps import DPOStep
    gen = torch.Generator().manual_seed(0)
    raw = []
    for i in range(5):
        ids = torch.randint(0, 509, (4, 16), generator=gen)
        mask = (torch.arange(16)[None] < torch.randint(3, 17, (4,), generator=gen)[:, None]).long()
        raw.append({"image": torch.randn(2, 3, 64, 64, generator=gen), "preferred_ids": ids[:2], "rejected_ids": ids[2:],
                    "preferred_mask": mask[:2], "rejected_mask": mask[2:]})
    got = list(BatchPrefetcher(raw, DPOStep.prepare, dev(), depth=2))
    assert len(got) == 5
    torch.cuda.synchronize()
    for r, p in zip(raw, got):
        want = DPOStep.prepare(r, dev())
        assert torch.equal(p["image"], want["image"]) and torch.equal(p["seq"].targets, want["seq"].targets)
        assert torch.equal(p["seq"].row_map, want["seq"].row_map) and p["seq"].n_rows == want["seq"].n_rows
    bad = raw[:2] + [{"image": raw[0]["image"]}]
    with pytest.raises(KeyError):
        list(BatchPrefetcher(bad, DPOStep.prepare, dev(), depth=2))


def test_seq_reduce_dpo_and_row_scale(hip, golden):
    g = golden("logprob_dpo")
    B = 8
    pc, pr, rc, rr = (torch.from_numpy(g[k]).to(dev()) for k in ("dpo_pc", "dpo_pr", "dpo_rc", "dpo_rr"))
    for name, ls, use_ref in (("std", 0.0, True), ("ls", 0.1, True), ("rf", 0.0, False)):
        loss = torch.zeros(1, device=dev())
        dw, dl, met = torch.zeros(B, device=dev()), torch.zeros(B, device=dev()), torch.zeros(4, device=dev())
        hip.dpo_loss(pc, pr, rc if use_ref else None, rr if use_ref else None, B, 0.1, ls, loss, dw, dl, met)
        assert abs(float(loss) - float(g[f"dpo_{name}_loss"])) <= 1e-5
        np.testing.assert_allclose(dw.cpu().numpy(), g[f"dpo_{name}_dpc"], rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(dl.cpu().numpy(), g[f"dpo_{name}_dpr"], rtol=1e-4, atol=1e-7)
    # compact-row reduce: rows sorted by sequence
    counts = torch.tensor([5, 0, 3, 7], dtype=torch.int32)
    seq_of_row = torch.repeat_interleave(torch.arange(4), counts.long()).int().to(dev())
    tok = rnd(int(counts.sum()), seed=1)
    for mode in (0, 1):
        out = torch.zeros(4, device=dev())
        hip.seq_reduce(tok, seq_of_row, tok.numel(), 4, counts.to(dev()), mode, out)
        ref = torch.zeros(4, device=dev()).index_add_(0, seq_of_row.long(), tok)
        if mode:
            ref = ref / counts.to(dev())
        assert torch.allclose(out, ref, atol=1e-5, equal_nan=True)  # empty sequence: 0/0 = NaN like the reference
    dseq = rnd(4, seed=2)
    rs = torch.zeros(tok.numel(), device=dev())
    hip.row_scale(dseq, seq_of_row, counts.to(dev()), tok.numel(), 1, rs)
    assert torch.allclose(rs, (dseq / counts.to(dev()))[seq_of_row.long()], atol=1e-7)
    hip.row_scale(dseq, seq_of_row, counts.to(dev()), tok.numel(), 2, rs)      # bit 2: negated (sum form)
    assert torch.equal(rs, -dseq[seq_of_row.long()])


def test_pool_and_normalise(hip):
    B, S, H = 3, 11, 96
    f = rnd(B, S, H, seed=1)
    mask = torch.tensor([[1] * 11, [1] * 4 + [0] * 7, [0] * 11], dtype=torch.int32, device=dev())
    pooled = torch.zeros(B, H, device=dev())
    hip.masked_mean_fwd(f, mask, B, S, H, pooled)
    ref = (f * mask[..., None]).sum(1) / mask.sum(1, keepdim=True).clamp(min=1)
    close(pooled, ref, 1e-6, "masked mean")
    dp = rnd(B, H, seed=2)
    df = torch.zeros(B, S, H, device=dev())
    hip.masked_mean_bwd(dp, mask, B, S, H, df)
    close(df, (dp / mask.sum(1, keepdim=True).clamp(min=1))[:, None, :] * mask[..., None], 1e-6, "masked mean bwd")
    x = rnd(B, 64, seed=3).requires_grad_()
    y, n = torch.zeros(B, 64, device=dev()), torch.zeros(B, device=dev())
    hip.l2norm_fwd(x.detach(), B, 64, y, n)
    ref = torch.nn.functional.normalize(x, p=2, dim=-1)
    close(y, ref, 1e-6, "l2norm")
    dy = rnd(B, 64, seed=4)
    ref.backward(dy)
    dx = torch.zeros(B, 64, device=dev())
    hip.l2norm_bwd(dy, y, n, B, 64, dx)
    close(dx, x.grad, 1e-5, "l2norm bwd")


def test_clip_adamw_schedule_matches_reference_optimizer(hip, golden):
    """sqnorm -> step_control -> adamw reproduces torch AdamW + clip_grad_norm_ + cosine warm-up
    (fixture from the reference's optimiser wiring, trainer.py:275-289,511-520)."""
    g = golden("optimizer")
    sizes = [33 * 17, 129]
    offs = [0, 576]           # 64-element aligned segments like ParamStore
    n = 576 + 192
    p = torch.zeros(n, device=dev())
    p[:561] = torch.from_numpy(g["p0_init"]).view(-1).to(dev())
    p[576:576 + 129] = torch.from_numpy(g["p1_init"]).to(dev())
    m, v, pb = torch.zeros(n, device=dev()), torch.zeros(n, device=dev()), torch.zeros(n, dtype=torch.bfloat16, device=dev())
    ctrl = torch.zeros(8, device=dev())
    part = torch.zeros(hip.sqnorm_blocks(n), device=dev())
    for step in range(4):
        gr = torch.zeros(n, device=dev())
        gr[:561] = torch.from_numpy(g[f"g0_{step}"]).view(-1).to(dev())
        gr[576:705] = torch.from_numpy(g[f"g1_{step}"]).to(dev())
        hip.sqnorm(gr, n, part)
        hip.step_control(part, part.numel(), 1.0, 5e-5, 2, 10, 1, 0.9, 0.999, 1.0, ctrl)
        hip.adamw(p, gr, m, v, pb, n, ctrl, 0.01, 0.9, 0.999, 1e-8)
        c = ctrl.cpu()
        assert abs(float(c[0]) - float(g[f"norm_{step}"])) <= 1e-4 * float(g[f"norm_{step}"])
        assert abs(float(c[3]) - float(g[f"lr_{step}"])) <= 1e-9
        np.testing.assert_allclose(p[:561].cpu().numpy(), g[f"p0_{step}"].reshape(-1), rtol=0, atol=1e-6)
        np.testing.assert_allclose(p[576:705].cpu().numpy(), g[f"p1_{step}"], rtol=0, atol=1e-6)
        assert torch.equal(pb, p.bfloat16())
    # non-finite gradient: the step is skipped, counters do not advance
    before, cb = p.clone(), ctrl.clone()
    gr[3] = float("nan")
    hip.sqnorm(gr, n, part)
    hip.step_control(part, part.numel(), 1.0, 5e-5, 2, 10, 1, 0.9, 0.999, 1.0, ctrl)
    hip.adamw(p, gr, m, v, pb, n, ctrl, 0.01, 0.9, 0.999, 1e-8)
    assert torch.equal(p, before) and float(ctrl[1]) == 0.0 and float(ctrl[6]) == float(cb[6])


def test_cast_axpy_gather(hip):
    x = rnd(1000, seed=1)
    y = torch.zeros(1000, dtype=torch.bfloat16, device=dev())
    hip.cast_bf16(x, y, 1000)
    assert torch.equal(y, x.bfloat16())
    z = torch.ones(1000, device=dev())
    hip.axpy(x, 0.5, z, 1000, accumulate=True)
    assert torch.allclose(z, 1 + 0.5 * x)
    src = rnd(20, 64, seed=2).bfloat16()
    rmap = torch.tensor([3, 3, 19, 0], dtype=torch.int32, device=dev())
    dst = torch.zeros(4, 64, dtype=torch.bfloat16, device=dev())
    hip.gather_rows_bf16(src, rmap, 4, 64, dst)
    assert torch.equal(dst, src[rmap.long()])


def test_bad_arguments_raise(hip):
    a = torch.zeros(8, 8, dtype=torch.bfloat16, device=dev())
    with pytest.raises(RuntimeError, match="multiple of 8"):
        hip.gemm(a, a, 8, 8, 4, 0, lda=8, ldb=8, out_f32=torch.zeros(8, 8, device=dev()))
    with pytest.raises(RuntimeError, match="S must be"):   # the backward keeps dQ of <= 4 query blocks in registers
        hip.attention_bwd(a, a, a, torch.zeros(8, device=dev()), None, 1, 600, 1, True, a)


def test_gemm_split_k_wgrad(hip):
    """Weight-gradient shape (few output tiles, long K): the 256^2 kernel splits K and adds partials with
    f32 atomics; the result must equal the accumulate semantics of the un-split kernel."""
    for (M, N, K) in ((1024, 1024, 8192), (256, 520, 4096), (1024, 3072, 16384)):
        a, b, A, B = operands(2, M, N, K, seed=K)
        ldb = None
        if N % 8:
            pytest.skip("unused")
        init = rnd(M, N, seed=3)
        out = init.clone()
        hip.gemm(A, B, M, N, K, hip.TN, out_f32=out, accumulate=True)
        ref = init + a.float() @ b.float()
        close(out, ref, 3e-4, f"split-K wgrad {M}x{N}x{K}")


# ----------------------------------------------------------------------------------------------- fused dropout
def drop_mult(seed, p, shape):
    """The oracle's restatement of the counter-based mask, on the device."""
    n = int(np.prod(shape))
    return R.dropout_multiplier(seed, p, n).view(*shape).to(dev())


def test_dropout_hash_statistics_and_gemm_epilogue(hip, tile):
    M, N, K = 320, 512, 128
    a, b, A, B = operands(1, M, N, K, seed=5)
    d = hip.drop_args(0xC0FFEE, 0.1)
    res = rnd(M, N, seed=2)
    out = torch.zeros(M, N, device=dev())
    hip.gemm(A, B, M, N, K, 1, residual=res, out_f32=out, drop=d)
    mult = drop_mult(0xC0FFEE, 0.1, (M, N))
    ref = (a.float() @ b.float()) * mult + res
    close(out, ref, 2e-4, "gemm + dropout + residual")
    frac = float((mult == 0).float().mean())
    assert abs(frac - 0.1) < 0.005 and abs(float(mult.max()) - 1 / 0.9) < 1e-6
    # activation epilogues: dropout after the activation; DRELU / DTANH replay the same mask
    h1 = torch.zeros(M, N, dtype=torch.bfloat16, device=dev())
    hip.gemm(A, B, M, N, K, 1, epilogue=hip.EPI_RELU, out_bf16=h1, drop=d)
    close(h1, torch.relu(a.float() @ b.float()) * mult, 1.0 / 100, "relu + dropout")
    g = torch.zeros(M, N, dtype=torch.bfloat16, device=dev())
    hip.gemm(A, B, M, N, K, 1, epilogue=hip.EPI_DRELU, aux_in=h1, out_bf16=g, drop=d)
    close(g, (a.float() @ b.float()) * (h1.float() > 0) * mult, 1.0 / 100, "drelu + dropout")
    y, yraw = torch.zeros_like(h1), torch.zeros_like(h1)
    hip.gemm(A, B, M, N, K, 1, epilogue=hip.EPI_TANH, alpha=0.05, out_bf16=y, aux_out=yraw, drop=d)
    th = torch.tanh(0.05 * (a.float() @ b.float()))
    close(yraw, th, 1.0 / 100, "tanh undropped copy")
    close(y, th * mult, 1.0 / 100, "tanh + dropout")


def test_attention_dropout_fwd_bwd(hip):
    B, S, heads = 3, 128, 2
    H = heads * 64
    qkv = rnd(B * S, 3 * H, seed=11).bfloat16()
    lens = torch.tensor([128, 77, 16])
    mask = (torch.arange(S)[None] < lens[:, None]).int().to(dev())
    d = hip.drop_args(12345, 0.1)
    out = torch.zeros(B * S, H, dtype=torch.bfloat16, device=dev())
    lse = torch.zeros(B, heads, S, device=dev())
    hip.attention_fwd(qkv, mask, B, S, heads, True, out, lse, drop=d)
    mult = drop_mult(12345, 0.1, (B, heads, S, S))
    x = qkv.float().requires_grad_()
    q, k, v = x.view(B, S, 3, heads, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * 0.125
    allowed = torch.tril(torch.ones(S, S, dtype=torch.bool, device=dev()))[None, None] & (mask[:, None, None, :] != 0)
    p = torch.softmax(s.masked_fill(~allowed, float("-inf")), -1) * mult
    ref = (p @ v).permute(0, 2, 1, 3).reshape(B * S, H)
    close(out, ref, 1.0 / 64, "attn + prob dropout")
    valid = (mask.view(B * S, 1) != 0).float()
    dout = (rnd(B * S, H, seed=7) * valid).bfloat16()
    ref.backward(dout.float())
    dqkv = torch.zeros(B * S, 3 * H, dtype=torch.bfloat16, device=dev())
    hip.attention_bwd(qkv, out, dout, lse, mask, B, S, heads, True, dqkv, drop=d)
    close(dqkv, x.grad, 1.0 / 40, "attn bwd with replayed dropout")


def test_layernorm_bwd_replays_neighbour_dropout(hip):
    M, H = 50, 256
    x, dy, add = rnd(M, H, seed=1), rnd(M, H, seed=2), rnd(M, H, seed=3)
    gamma = rnd(H, seed=4) * 0.1 + 1
    mean, rstd = torch.zeros(M, device=dev()), torch.zeros(M, device=dev())
    hip.layernorm_fwd(x, M, H, gamma, torch.zeros(H, device=dev()), y_f32=torch.zeros(M, H, device=dev()), mean=mean, rstd=rstd)
    da, dd = hip.drop_args(111, 0.1), hip.drop_args(222, 0.1)
    nb = hip.layernorm_bwd_blocks(M)
    part4 = torch.zeros(4, nb, H, device=dev())
    dx, dxb = torch.zeros(M, H, device=dev()), torch.zeros(M, H, dtype=torch.bfloat16, device=dev())
    hip.layernorm_bwd(x, M, H, gamma, mean, rstd, dx, dy_f32=dy, add_to=add, dx_bf16=dxb, part=part4[:2],
                      part_extra=part4[2:], drop_add=da, drop_dx=dd)
    xs = x.clone().requires_grad_()
    torch.nn.functional.layer_norm(xs, (H,), gamma, None, 1e-5).backward(dy)
    want = xs.grad + add
    close(dx, want, 2e-5, "f32 stream gradient is never masked")
    ma, md = drop_mult(111, 0.1, (M, H)), drop_mult(222, 0.1, (M, H))
    close(dxb, want * md, 1.0 / 100, "bf16 copy carries the consumer's mask")
    sa, sd = torch.zeros(H, device=dev()), torch.zeros(H, device=dev())
    hip.colsum_finish4(part4[2:], 2, nb, H, [sa, sd])
    close(sa, (add * ma).sum(0), 2e-5, "masked column sum of add_to")
    close(sd, (want * md).sum(0), 2e-5, "masked column sum of dx")


def test_embed_train_mode_heads_and_dropout(hip):
    B, S, H, V, XH = 3, 9, 128, 50, 8
    ids = torch.randint(0, V, (B, S), generator=torch.Generator().manual_seed(0)).to(dev())
    wte, wpe = rnd(V, H, seed=1), rnd(16, H, seed=2)
    U, bo = rnd(B, XH, H, seed=3), rnd(H, seed=4)
    gamma, beta = rnd(H, seed=5) * 0.1 + 1, rnd(H, seed=6) * 0.1
    dx_, de_ = hip.drop_args(77, 0.3), hip.drop_args(88, 0.1)
    h0 = torch.zeros(B * S, H, device=dev())
    mean, rstd = torch.zeros(B * S, device=dev()), torch.zeros(B * S, device=dev())
    hip.embed_fwd(ids, B, S, H, wte, wpe, h0, attended=bo, att_stride=0, gamma=gamma, beta=beta, mean=mean, rstd=rstd,
                  U=U, xheads=XH, drop_x=dx_, drop_e=de_)
    w = drop_mult(77, 0.3, (B, XH, S))                                  # head weights, 0 or 1/(1-p)
    me = drop_mult(88, 0.1, (B, S, H))
    wr, pr, Ur, br = (t.clone().requires_grad_() for t in (wte, wpe, U, bo))
    gr, ber = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    att = br[None, None, :] + torch.einsum("bhs,bhc->bsc", w, Ur)
    ref = (torch.nn.functional.layer_norm(wr[ids] + att, (H,), gr, ber, 1e-5) + pr[:S][None]) * me
    close(h0, ref.view(B * S, H), 1e-5, "embed fwd (train mode)")
    g = rnd(B * S, H, seed=9)
    ref.view(B * S, H).backward(g)
    dwte, dwpe, datt, dU = torch.zeros_like(wte), torch.zeros_like(wpe), torch.zeros(B, H, device=dev()), torch.zeros_like(U)
    nb = hip.embed_bwd_blocks(B, S)
    part = torch.zeros(2, nb, H, device=dev())
    hip.embed_bwd(g, ids, None, B, S, H, dwte, dwpe, wte=wte, attended=bo, att_stride=0, gamma=gamma, mean=mean,
                  rstd=rstd, dattended=datt, part=part, U=U, dU=dU, xheads=XH, drop_x=dx_, drop_e=de_)
    close(dwte, wr.grad, 3e-5, "dwte")
    close(dwpe, pr.grad, 3e-5, "dwpe")
    close(dU, Ur.grad, 3e-5, "dU")
    close(datt.sum(0), br.grad, 3e-5, "d b_o")


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("shape", [(2048, 1024, 1024), (1536, 768, 4096), (4096, 512, 256), (1000, 520, 192)])
def test_phase_staggered_gemm_race_screen(hip, layout, shape):
    """The phase-staggered loop (gemm_phase.hip) orders its LDS-DMA, fragment reads and barriers by count, two wave
    groups one barrier apart.  A mis-placed wait shows up as rare wrong tiles, so: 25 launches per shape, each
    compared BITWISE with the plain 2-stage kernel (same K order per output element -> identical f32 results)."""
    import os
    M, N, K = shape
    a, b, A, B = operands(layout, M, N, K, seed=layout * 7 + K)
    ldb = None
    if layout != 0 and N % 8:
        ldn = (N + 7) // 8 * 8
        Bp = torch.zeros(K, ldn, dtype=torch.bfloat16, device=dev())
        Bp[:, :N] = b
        B, ldb = Bp, ldn
    hip.set_option("gemm_tile", 256)
    try:
        hip.set_option("gemm_schedule", 0)
        want = torch.zeros(M, N, device=dev())
        hip.gemm(A, B, M, N, K, layout, ldb=ldb, out_f32=want)
        close(want, a.float() @ b.float(), 2e-4, "2-stage kernel")
        for sched in (6,):
            hip.set_option("gemm_schedule", sched)
            for it in range(25):
                got = torch.full((M, N), float("nan"), device=dev())
                hip.gemm(A, B, M, N, K, layout, ldb=ldb, out_f32=got)
                assert torch.equal(got, want), f"schedule {sched} launch {it}: {(got != want).sum().item()} elements differ"
    finally:
        hip.set_option("gemm_tile", 0)
        hip.set_option("gemm_schedule", -1)


def test_grouped_weight_gradient_launch(hip):
    """Four TN problems of different shapes in one grid (no split-K): each equals X^t dY accumulated on the old
    value; and the fallback (K % 64 != 0) gives the same through ordinary launches."""
    for K in (2048, 200):
        shapes = [(512, 256), (256, 256), (256, 768), (256, 512)]
        probs, refs = [], []
        for i, (M, N) in enumerate(shapes):
            x = rnd(K, M, seed=10 + i).bfloat16()
            dy = rnd(K, N, seed=20 + i).bfloat16()
            g0 = rnd(M, N, seed=30 + i)
            gout = g0.clone()
            probs.append((x, dy, M, N, K, gout))
            refs.append(g0 + x.float().t() @ dy.float())
        hip.gemm_wgrad_group(probs)
        for (x, dy, M, N, K, gout), ref in zip(probs, refs):
            close(gout, ref, 3e-4, f"grouped wgrad {M}x{N} K={K}")


@pytest.mark.parametrize("shape", [(512, 512, 256), (300, 200, 128), (1024, 4096, 64), (64, 72, 64), (130, 520, 192)])
def test_dgelu_gemm_leaves_bias_gradient_partials(hip, tile, shape):
    """``colsum_part`` (DGELU_NEW epilogue): row b of the partial matrix holds the column sums of rows 64b..64b+63 of the
    result; summed over b they are colsum(dpre) = the bias gradient of mlp.c_fc (HF modeling_gpt2.py:229-243 under
    trainer.py:494,606 loss.backward()) - fast path, edge blocks in M and in N, both tile sizes."""
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N)
    a = (torch.randn(M, K, generator=g) * 0.5).to(dev()).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.5).to(dev()).bfloat16()          # NT: out = a @ w^t
    pre = torch.randn(M, N, generator=g).to(dev()).bfloat16()
    nbr = (M + 63) // 64
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
    part = torch.full((nbr, N), float("nan"), device=dev())
    hip.gemm(a, w, M, N, K, hip.NT, epilogue=hip.EPI_DGELU_NEW, aux_in=pre, out_bf16=out, colsum_part=part)
    x = pre.float()
    k2 = 2.0 * 0.7978845608028654
    s = torch.sigmoid(k2 * x * (1 + 0.044715 * x * x))
    want = (a.float() @ w.float().t()) * (s + x * s * (1 - s) * k2 * (1 + 3 * 0.044715 * x * x))
    assert torch.isfinite(part).all()                                           # every (block row, column) was written
    for b in range(nbr):
        ref = want[64 * b:64 * b + 64].double().sum(0)
        close(part[b].double(), ref, 2e-3, f"block row {b}")
    bias_grad = torch.full((N,), 1.0, device=dev())
    hip.colsum_finish(part, nbr, N, bias_grad, accumulate=True)
    close(bias_grad.double() - 1.0, want.double().sum(0), 2e-3, "bias gradient")
    close(out.float(), want, 2e-2, "dpre itself")
    with pytest.raises(RuntimeError, match="colsum_part"):
        hip.gemm(a, w, M, N, K, hip.NT, out_bf16=out, colsum_part=part)


@pytest.mark.parametrize("shape", [(512, 512, 256), (300, 200, 128), (130, 520, 192)])
def test_gelu_with_saved_derivative_and_multiply_backward(hip, tile, shape):
    """The GPT-2 MLP pair that evaluates the sigmoid once: forward ``EPI_GELU_NEW_D`` writes gelu_new(pre) and, as aux_out,
    gelu_new'(pre); backward ``EPI_MUL_AUX`` multiplies the data gradient by that saved derivative and leaves the bias-gradient
    column sums - together the same numbers as GELU_NEW + DGELU_NEW (HF NewGELUActivation, activations.py:59-66)."""
    M, N, K = shape
    g = torch.Generator().manual_seed(M * 3 + N)
    x = (torch.randn(M, K, generator=g) * 0.5).to(dev()).bfloat16()
    w = (torch.randn(K, N, generator=g) * 0.2).to(dev()).bfloat16()            # NN: pre = x @ w + b
    bias = torch.randn(N, generator=g).to(dev())
    act = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
    der = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
    hip.gemm(x, w, M, N, K, hip.NN, epilogue=hip.EPI_GELU_NEW_D, bias=bias, out_bf16=act, aux_out=der)
    pre = (x.float() @ w.float() + bias).requires_grad_()
    y = R.gelu_new(pre)
    y.sum().backward()
    close(act, y.detach(), 1.0 / 100, "gelu_new")
    close(der, pre.grad, 1.0 / 100, "gelu_new' saved by the forward")
    # backward: dpre = (go @ w2^t) * gelu'(pre), column sums per 64-row block
    go = (torch.randn(M, 64, generator=g) * 0.5).to(dev()).bfloat16()
    w2 = (torch.randn(N, 64, generator=g) * 0.5).to(dev()).bfloat16()           # NT: go @ w2^t
    nbr = (M + 63) // 64
    dpre = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
    part = torch.full((nbr, N), float("nan"), device=dev())
    hip.gemm(go, w2, M, N, 64, hip.NT, epilogue=hip.EPI_MUL_AUX, aux_in=der, out_bf16=dpre, colsum_part=part)
    want = (go.float() @ w2.float().t()) * der.float()
    close(dpre.float(), want, 1.0 / 100, "dpre")
    assert torch.isfinite(part).all()
    close(part.double().sum(0), want.double().sum(0), 2e-3, "bias gradient from the same pass")
    old = torch.empty_like(dpre)                                                   # and against the two-sigmoid pair
    hip.gemm(go, w2, M, N, 64, hip.NT, epilogue=hip.EPI_DGELU_NEW, aux_in=pre.detach().bfloat16(), out_bf16=old)
    close(dpre.float(), old.float(), 3e-2, "MUL_AUX vs DGELU_NEW")
