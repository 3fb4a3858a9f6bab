"""Packed (variable-length) token rows == the padded [B, S] layout of the reference batch.

With right padding, a causal AND key-padding mask, a masked mean and a masked loss, nothing a padded position computes
reaches a loss term or a gradient (reference models/model.py:449-456,1069-1083, models/components.py:340-357; SURVEY 3.1
items 6-7), so the training steps run both GPT-2 trunks on the rows of the real tokens only (``engine.RowPack``).
Positions, key masks and every dropout index stay keyed on the padded (b, t), so the two layouts must agree token for
token: index work bit-exact, kernel outputs of the real rows bit-identical where the arithmetic order is the same,
scored log-probs / loss / gradients within f32 summation-order noise (the weight gradients sum over a different number
of rows).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def hip():
    from pgca_amd import hip as H
    H.load()
    return H


def ragged_mask(B, S, lens, holes=()):
    m = (torch.arange(S)[None] < torch.tensor(lens)[:, None]).long()
    for b, t in holes:
        m[b, t] = 0
    return m


# ------------------------------------------------------------------------------------------------- index work
@pytest.mark.parametrize("S,lens,holes", [
    (16, [16, 5, 1, 9], ()),
    (128, [128, 40, 77, 16, 2, 100], ((1, 3), (2, 50))),   # holes inside a sequence keep their rows
    (64, [64, 0, 30], ()),                                  # an empty sequence owns no row
    (256, [64] * 4, ()),                                    # row count already a multiple of the pad: no filler
])
def test_row_pack_indices_bit_exact(hip, S, lens, holes):
    from pgca_amd.engine import PACK_PAD, make_seq_batch, n_filler_seqs
    B = len(lens)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, 1000, (B, S), generator=g)
    mask = ragged_mask(B, S, lens, holes)
    if int(mask[:, 1:].sum()) == 0:
        pytest.skip("no scored token")
    sb = make_seq_batch(ids, mask, dev())
    pk = sb.pack
    mk = mask.numpy()
    want_len = np.array([0 if not r.any() else int(np.nonzero(r)[0].max()) + 1 for r in mk])
    cu = np.concatenate([[0], np.cumsum(want_len)])
    n = int(cu[-1])
    Mp = (n + PACK_PAD - 1) // PACK_PAD * PACK_PAD
    assert pk.n == n and pk.Mp == Mp and pk.Mp % PACK_PAD == 0
    assert np.array_equal(pk.lens.cpu().numpy(), want_len)
    F = n_filler_seqs(S)
    assert F == (1 if S >= 63 else -(-63 // S)) and pk.nseq == B + F
    fill = [min(n + f * S, Mp) for f in range(1, F + 1)]         # filler pseudo-sequences of at most S rows each
    assert np.array_equal(pk.cu.cpu().numpy(), np.concatenate([cu, fill]))
    rid = np.concatenate([b * S + np.arange(want_len[b]) for b in range(B)] + [np.full(Mp - n, -1)])
    assert np.array_equal(pk.row_ids.cpu().numpy(), rid)
    assert torch.equal(pk.mask[:B].cpu(), mask.int()) and bool((pk.mask[B:] == 1).all())
    # the scored rows: same (b, t), renumbered
    rm = sb.row_map.cpu().numpy()
    b_of, t_of = rm // S, rm % S
    assert np.array_equal(sb.row_map_packed.cpu().numpy(), cu[b_of] + t_of)
    assert np.array_equal(pk.row_ids.cpu().numpy()[sb.row_map_packed.cpu().numpy()], rm)


# ------------------------------------------------------------------------------------------------- attention
@pytest.mark.parametrize("S,heads,lens,p", [(128, 4, [128, 40, 77, 16, 1], 0.0), (128, 2, [100, 33, 128], 0.1),
                                            (256, 2, [256, 130, 77, 200], 0.1), (384, 1, [300, 129, 5], 0.0),
                                            (16, 2, [16, 3, 9, 1, 12], 0.1)])   # S < 63: several filler sequences
def test_packed_attention_equals_padded(hip, S, heads, lens, p):
    from pgca_amd.engine import make_row_pack
    B, H = len(lens), heads * 64
    g = torch.Generator().manual_seed(11)
    mask = ragged_mask(B, S, lens).int().to(dev())
    qkv = (torch.randn(B * S, 3 * H, generator=g)).bfloat16().to(dev())
    dout = (torch.randn(B * S, H, generator=g)).bfloat16().to(dev())
    dout = dout * mask.reshape(-1, 1).to(dout.dtype)   # a padded query has no gradient (its loss weight is zero)
    d = hip.drop_args(1234, p)
    out, lse = torch.zeros(B * S, H, dtype=torch.bfloat16, device=dev()), torch.zeros(B, heads, S, device=dev())
    dqkv = torch.zeros(B * S, 3 * H, dtype=torch.bfloat16, device=dev())
    hip.attention_fwd(qkv, mask, B, S, heads, True, out, lse, drop=d)
    hip.attention_bwd(qkv, out, dout, lse, mask, B, S, heads, True, dqkv, drop=d)
    pk = make_row_pack(mask)
    rows = pk.row_ids[:pk.n].long()
    qkv_p = torch.zeros(pk.Mp, 3 * H, dtype=torch.bfloat16, device=dev())
    dout_p = torch.zeros(pk.Mp, H, dtype=torch.bfloat16, device=dev())
    qkv_p[:pk.n], dout_p[:pk.n] = qkv[rows], dout[rows]
    out_p = torch.full((pk.Mp, H), float("nan"), dtype=torch.bfloat16, device=dev())
    lse_p = torch.zeros(pk.nseq, heads, S, device=dev())
    dqkv_p = torch.full((pk.Mp, 3 * H), float("nan"), dtype=torch.bfloat16, device=dev())
    hip.attention_fwd(qkv_p, pk.mask, pk.nseq, S, heads, True, out_p, lse_p, drop=d, cu=pk.cu)
    hip.attention_bwd(qkv_p, out_p, dout_p, lse_p, pk.mask, pk.nseq, S, heads, True, dqkv_p, drop=d, cu=pk.cu)
    assert torch.equal(out_p[:pk.n], out[rows]), "forward rows differ"
    for b, n in enumerate(lens):
        assert torch.equal(lse_p[b, :, :n], lse[b, :, :n])
    assert torch.equal(dqkv_p[:pk.n], dqkv[rows]), "backward rows differ"
    # the filler rows are an ordinary (finite) sequence of zero inputs: zero gradient, finite output
    assert bool(torch.isfinite(out_p.float()).all()) and float(dqkv_p[pk.n:].float().abs().max() if pk.Mp > pk.n else 0) == 0.0


# ------------------------------------------------------------------------------------------------- embeddings
def test_packed_embedding_equals_padded(hip):
    from pgca_amd.engine import make_row_pack
    B, S, H, V = 3, 32, 256, 97
    g = torch.Generator().manual_seed(3)
    mask = ragged_mask(B, S, [32, 7, 20]).int().to(dev())
    ids = torch.randint(0, V, (B, S), generator=g).to(dev())
    wte, wpe = torch.randn(V, H, generator=g).to(dev()), torch.randn(S, H, generator=g).to(dev())
    att = torch.randn(B, H, generator=g).to(dev())
    gamma, beta = torch.randn(H, generator=g).to(dev()), torch.randn(H, generator=g).to(dev())
    de = hip.drop_args(77, 0.1)
    M = B * S
    h0, mean, rstd = torch.zeros(M, H, device=dev()), torch.zeros(M, device=dev()), torch.zeros(M, device=dev())
    hip.embed_fwd(ids, B, S, H, wte, wpe, h0, attended=att, gamma=gamma, beta=beta, mean=mean, rstd=rstd, drop_e=de)
    pk = make_row_pack(mask)
    rows = pk.row_ids[:pk.n].long()
    h0p = torch.full((pk.Mp, H), float("nan"), device=dev())
    mp, rp = torch.zeros(pk.Mp, device=dev()), torch.zeros(pk.Mp, device=dev())
    hip.embed_fwd(ids, B, S, H, wte, wpe, h0p, attended=att, gamma=gamma, beta=beta, mean=mp, rstd=rp, drop_e=de,
                  row_ids=pk.row_ids, n_rows=pk.Mp)
    assert torch.equal(h0p[:pk.n], h0[rows]) and torch.equal(mp[:pk.n], mean[rows]) and torch.equal(rp[:pk.n], rstd[rows])
    assert float(h0p[pk.n:].abs().max()) == 0.0
    # backward: the same gradient rows, dense vs packed
    gd = torch.zeros(M, H, device=dev())
    gd[rows] = torch.randn(pk.n, H, generator=g).to(dev())
    gp = torch.zeros(pk.Mp, H, device=dev())
    gp[:pk.n] = gd[rows]
    nb = hip.embed_bwd_blocks(B, S)
    res = []
    for gg, m_, r_, cu in ((gd, mean, rstd, None), (gp, mp, rp, pk.cu)):
        dwte, dwpe = torch.zeros(V, H, device=dev()), torch.zeros(S, H, device=dev())
        datt, part = torch.zeros(B, H, device=dev()), torch.zeros(2, nb, H, device=dev())
        hip.embed_bwd(gg, ids, mask, B, S, H, dwte, dwpe, wte=wte, attended=att, gamma=gamma, mean=m_, rstd=r_,
                      dattended=datt, part=part, drop_e=de, cu=cu)
        res.append((dwte, dwpe, datt, part.sum(1)))
    for a, b in zip(*res):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)   # atomics: order differs, values do not


# ------------------------------------------------------------------------------------------------- whole steps
def _model(text_model, vision="openai/clip-vit-base-patch32", layers=2, seed=7):
    from pgca_amd.arch import make_arch, with_layers
    from pgca_amd.model import PreferenceGuidedCaptioningModel
    arch = with_layers(make_arch(vision, text_model, 512), layers, layers)
    return PreferenceGuidedCaptioningModel(freeze_vision_backbone=True, arch=arch, seed=seed, device="cuda:0"), arch


def _grads(model):
    return {n: s.grad.clone() for n, s in model.store.segments.items() if s.grad is not None}


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


@pytest.mark.parametrize("S,lens,train,ref_free,holes", [
    (128, [128, 40, 77, 16], False, True, ()), (128, [100, 33, 64, 128], True, False, ()),
    (256, [256, 130, 77, 200], True, True, ()), (256, [17, 250, 129, 128], False, False, ()),
    (128, [90, 128, 5, 61], True, False, ((0, 7), (0, 8), (1, 100), (3, 1)))])   # masks with holes inside a caption
def test_packed_dpo_step_equals_padded(S, lens, train, ref_free, holes):
    """Stage-2 step (2-forward and 4-forward), eval and train mode (dropout 0.1 at every site): scored log-probs of the
    policy and of the reference policy, loss, and every gradient tensor - packed rows vs all B*S rows."""
    from pgca_amd.engine import DropoutPlan
    from pgca_amd.steps import DPOStep, ReferencePolicy
    model, arch = _model("gpt2-medium")
    gen = torch.Generator().manual_seed(99)
    B = 2
    img = torch.randn(B, 3, 224, 224, generator=gen)
    ids = torch.randint(0, 50257, (2 * B, S), generator=gen)
    mask = ragged_mask(2 * B, S, lens, holes)
    ids = torch.where(mask.bool(), ids, torch.full_like(ids, 50257))
    batch = {"image": img, "preferred_ids": ids[:B], "rejected_ids": ids[B:], "preferred_mask": mask[:B],
             "rejected_mask": mask[B:]}
    ref = None if ref_free else ReferencePolicy(model.store, model.ws)
    if ref is not None:   # make the reference differ from the policy, or the DPO logits are identically zero
        for seg in ref.store.segments.values():
            seg.fp32.mul_(1.02)
            seg.ensure_bf16()
    out = {}
    for packed in (False, True):
        step = DPOStep(model.store, model.ws, model.vision_encoder.tower, model.vision_encoder.head,
                       model.caption_decoder.engine, beta=0.1, reference_free=ref_free, ref=ref,
                       dropout=DropoutPlan(0.1 if train else 0.0, base_seed=5), packed=packed)
        p = DPOStep.prepare(batch, model.device)
        assert p["seq"].pack is not None
        for s in model.store.trainable_segments():
            s.grad.zero_()
        loss = step.loss_and_grads(p["image"], p["seq"]).clone()
        pol = model.ws.bufs["pol.seq_lp"][:2 * B].clone()
        rlp = model.ws.bufs["ref.seq_lp"][:2 * B].clone() if ref is not None else None
        tok = model.ws.bufs["pol.tok_lp"][:p["seq"].n_rows].clone()
        out[packed] = (loss, pol, rlp, tok, _grads(model))
    (l0, p0, r0, t0, g0), (l1, p1, r1, t1, g1) = out[False], out[True]
    assert float((t0 - t1).abs().max()) <= 1e-5, "token log-probs"
    assert float((p0 - p1).abs().max()) <= 1e-4 and abs(float(l0) - float(l1)) <= 1e-6
    if r0 is not None:
        assert float((r0 - r1).abs().max()) <= 1e-4
    for name in g0:
        seg = model.store.segments[name]
        if float(g0[name].abs().max()) == 0.0:     # a segment this stage does not train
            assert float(g1[name].abs().max()) == 0.0
            continue
        # the vision head sits behind the per-sequence f32 ATOMIC sums of the embedding backward (dattended / dU: order not
        # reproducible from launch to launch in either layout) followed by a bf16 GEMM chain, so its gradient carries
        # bf16 rounding noise between any two runs; the decoder's own gradients only differ by f32 summation order
        # (the same holds for the decoder's own cross-attention / vision_projection / attention_norm tensors, which sit
        # behind those atomics too; the GPT-2 trunk, ln_f, wte and wpe do not)
        cmin = 0.9999 if name == "vision_head" else 0.99999
        assert _cos(g0[name], g1[name]) >= cmin, name
        for key in list(seg.index)[:400]:
            off, n = seg.index[key][0], int(np.prod(seg.index[key][1]))
            a, b = g0[name][off:off + n], g1[name][off:off + n]
            scale = float(a.abs().max())
            trunk = ".transformer." in key
            # trunk tensors differ by f32 summation order only (weight gradients over a different row count, LayerNorm /
            # bias partial sums over a different row partition): measured <= 2e-3 of the tensor's scale, cosine >= 0.999998
            rel = 5e-3 if trunk else 3e-2
            assert float((a - b).abs().max()) <= rel * scale + 1e-9, f"{key}: {float((a - b).abs().max())} vs {scale}"
            if trunk and scale > 0:
                assert _cos(a, b) >= 0.99999, key


def test_reference_policy_on_its_own_stream_changes_nothing():
    """The frozen reference policy's forward on a second HIP stream (DPOStep(ref_side_stream=True): its tile tails and
    epilogues interleave with the policy forward's main loops) against the single-stream order: the reference log-probs
    are bitwise equal (no atomics on that path), the loss and the policy's results too, gradients up to the f32-atomic
    summation order of the single-stream run itself."""
    from pgca_amd.engine import DropoutPlan
    from pgca_amd.steps import DPOStep, ReferencePolicy
    model, arch = _model("gpt2-medium")
    gen = torch.Generator().manual_seed(7)
    B, S = 3, 128
    img = torch.randn(B, 3, 224, 224, generator=gen)
    ids = torch.randint(0, 50257, (2 * B, S), generator=gen)
    mask = ragged_mask(2 * B, S, [128, 40, 77, 16, 99, 64], ())
    ids = torch.where(mask.bool(), ids, torch.full_like(ids, 50257))
    batch = {"image": img, "preferred_ids": ids[:B], "rejected_ids": ids[B:], "preferred_mask": mask[:B],
             "rejected_mask": mask[B:]}
    ref = ReferencePolicy(model.store, model.ws)
    for seg in ref.store.segments.values():
        seg.fp32.mul_(1.02)
        seg.ensure_bf16()
    out = {}
    for side in (False, True, True):     # twice on the side stream: the second call reuses the stream and the buffers
        step = DPOStep(model.store, model.ws, model.vision_encoder.tower, model.vision_encoder.head,
                       model.caption_decoder.engine, beta=0.1, reference_free=False, ref=ref, ref_side_stream=side,
                       dropout=DropoutPlan(0.1, base_seed=5))
        p = DPOStep.prepare(batch, model.device)
        for sg in model.store.trainable_segments():
            sg.grad.zero_()
        loss = step.loss_and_grads(p["image"], p["seq"]).clone()
        torch.cuda.synchronize()
        out[side] = (loss, model.ws.bufs["pol.seq_lp"][:2 * B].clone(), model.ws.bufs["ref.seq_lp"][:2 * B].clone(),
                     _grads(model))
    (l0, p0, r0, g0), (l1, p1, r1, g1) = out[False], out[True]
    assert torch.equal(r0, r1) and torch.equal(p0, p1) and torch.equal(l0, l1)
    for name in g0:
        if float(g0[name].abs().max()) > 0.0:
            assert _cos(g0[name], g1[name]) >= 0.99999, name


@pytest.mark.parametrize("train", [False, True])
def test_packed_contrastive_step_equals_padded(train):
    """Stage-1 step: NT-Xent loss and the gradients of the text tower and both heads, packed vs padded rows."""
    from pgca_amd.engine import DropoutPlan
    from pgca_amd.steps import ContrastiveStep
    model, arch = _model("gpt2-medium")
    gen = torch.Generator().manual_seed(21)
    B, S = 4, 128
    img = torch.randn(B, 3, 224, 224, generator=gen)
    ids = torch.randint(0, 50257, (B, S), generator=gen)
    mask = ragged_mask(B, S, [128, 9, 77, 40], holes=((2, 11),))
    out = {}
    for packed in (False, True):
        step = ContrastiveStep(model.store, model.ws, model.vision_encoder.tower, model.vision_encoder.head,
                               model.text_encoder.engine, temperature=0.5,
                               dropout=DropoutPlan(0.1 if train else 0.0, base_seed=5), packed=packed)
        p = ContrastiveStep.prepare({"image": img, "caption_ids": ids, "caption_mask": mask}, model.device)
        for s in model.store.trainable_segments():
            s.grad.zero_()
        loss = float(step.loss_and_grads(p["image"], p["ids"], p["mask"], pack=p["pack"]))
        out[packed] = (loss, model.ws.bufs["text.pooled"][:B * arch.gpt.hidden].clone(), _grads(model))
    (l0, q0, g0), (l1, q1, g1) = out[False], out[True]
    assert abs(l0 - l1) <= 1e-6
    assert float((q0 - q1).abs().max()) <= 1e-5
    for name in g0:
        if float(g0[name].abs().max()) == 0.0:
            assert float(g1[name].abs().max()) == 0.0
            continue
        assert _cos(g0[name], g1[name]) >= 0.999999, name
