"""Caption generation on the HIP kernels (reference model.py:621-678, 855-923): K/V-cache decoding,
greedy and HF-rule beam search against ids produced by the REFERENCE's own generate (tests/golden/generation.npz) and the
oracle's restatement of it; the sampling path's filtered distribution against the oracle; contracts of the rest."""
import pytest
import torch

from oracle import restatement as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def model():
    from pgca_amd.arch import tiny_arch
    from pgca_amd.model import PreferenceGuidedCaptioningModel
    return PreferenceGuidedCaptioningModel(freeze_vision_backbone=True, arch=tiny_arch(), seed=17, device=DEV)


def _fixture(model, golden):
    g = golden("generation")
    sd = {k: v.detach().cpu() for k, v in model.store.state_dict(aliases=False).items()}
    return g, sd, torch.from_numpy(g["images"]), torch.from_numpy(g["embeddings"]), int(g["pad"]), int(g["eos"])


def test_decode_logits_along_the_reference_captions(model, golden):
    """tests/golden/generation.npz holds ids from the REFERENCE's own CaptionDecoder.generate (HF generate with its K/V
    cache).  Teacher-forced along those captions, the K/V-cache decode must give the oracle's next-token logits at every
    position, and the reference's token wherever its margin over the runner-up is outside bf16 noise.  (A free-running
    comparison says little on this model: its 509 logits have a spread of 0.2 and first-step top-2 gaps of 0.002-0.014.)"""
    g, sd, img, emb, pad, eos = _fixture(model, golden)
    arch, eng = model.arch, model.caption_decoder.engine
    want = torch.from_numpy(g["greedy_norep_ids"])                       # [5, 8], no repetition penalty: raw argmax
    pv = eng.prefix_embedding(model.vision_encoder(img)["embeddings"])
    assert float((model.vision_encoder(img)["embeddings"].cpu() - emb).abs().max()) <= 4e-2
    worst, agree = 0.0, 0
    for use_graphs in (False, True, True):                               # eager, capture pass, replay pass
        eng.use_graphs = use_graphs
        logits = eng.decode_begin(pv, want.shape[1] + 1)
        for t in range(want.shape[1]):
            ref = R.generate_step_logits(sd, emb, want[:, :t], arch.gpt.heads)
            got = logits.cpu()
            worst = max(worst, float((got - ref).abs().max()))
            top2 = ref.topk(2, dim=-1).values
            sure = (top2[:, 0] - top2[:, 1]) >= 0.03
            assert torch.equal(got.argmax(-1)[sure], want[:, t][sure]), t
            agree += int(sure.sum())
            logits = eng.decode_advance(want[:, t].to(DEV))
    assert worst <= 2e-2, worst
    assert agree >= 60, agree


def test_generation_loop_rules_reproduce_the_reference_generate(model, golden, monkeypatch):
    """The token-selection logic - HF's length convention (the prefix embedding is one of the max_length positions), EOS
    and padding, repetition penalty on logits (greedy) or on log-probabilities (beams), HF's beam-search bookkeeping -
    given EXACT logits: the engine's logits are replaced by the oracle's, and every case of the fixture (greedy, beams,
    an EOS that occurs) must come out token for token as the reference's generate produced it."""
    import json
    g, sd, img, emb, pad, eos = _fixture(model, golden)
    arch, eng = model.arch, model.caption_decoder.engine

    def oracle_logits(pv, ids):
        rows = ids.shape[0] // emb.shape[0]
        return R.generate_step_logits(sd, emb.repeat_interleave(rows, 0), ids.cpu(), arch.gpt.heads).to(DEV)
    monkeypatch.setattr(eng, "next_token_logits", oracle_logits)
    for name in ("greedy", "greedy_norep", "beam4", "beam3_rep"):
        kw = json.loads(str(g[name + "_kw"]))
        got = model.generate_token_ids(img, use_cache=False, **kw).cpu()
        assert torch.equal(got, torch.from_numpy(g[name + "_ids"])), name
    e2 = int(g["eos_case_eos"])
    got = model.generate_token_ids(img, max_length=9, num_beams=1, do_sample=False, repetition_penalty=1.0,
                                   eos_token_id=e2, use_cache=False).cpu()
    assert torch.equal(got, torch.from_numpy(g["greedy_eos_ids"]))
    assert int(got[0, 0]) == e2 and bool((got[0, 1:] == pad).all())
    got = model.generate_token_ids(img, max_length=8, num_beams=4, do_sample=False, repetition_penalty=1.0,
                                   eos_token_id=e2, use_cache=False).cpu()
    assert torch.equal(got, torch.from_numpy(g["beam4_eos_ids"]))


def test_cached_and_cache_free_generation_return_the_same_captions(model):
    """Free-running greedy and beam search, K/V cache against prefix recomputation, on a model whose logits are sharp
    enough for the comparison to mean something (tied embedding x 12)."""
    arch = model.arch
    img = torch.randn(6, 3, arch.vit.image, arch.vit.image, generator=torch.Generator().manual_seed(31))
    seg = model.store.segments["decoder"]
    name = "caption_decoder.lm_model.transformer.wte.weight"
    saved = seg.w(name).clone()
    try:
        seg.w(name).mul_(12.0)
        model.sync_bf16()
        for kw in (dict(num_beams=1, do_sample=False, repetition_penalty=1.1), dict(num_beams=3, do_sample=False)):
            a = model.generate_token_ids(img, max_length=14, use_cache=True, **kw)
            b = model.generate_token_ids(img, max_length=14, use_cache=False, **kw)
            same = (a == b).float().mean()
            assert a.shape == b.shape and float(same) >= 0.9, (kw, float(same))
    finally:
        seg.w(name).copy_(saved)
        model.sync_bf16()


def test_cache_and_cache_free_decoding_agree(model):
    """K/V-cache decode (one position per step) against recomputing the prefix: the same next-token logits at every step,
    within the rounding of two different GEMM row counts."""
    arch = model.arch
    eng = model.caption_decoder.engine
    img = torch.randn(4, 3, arch.vit.image, arch.vit.image, generator=torch.Generator().manual_seed(11))
    pv = eng.prefix_embedding(model.vision_encoder(img)["embeddings"])
    gen = torch.Generator().manual_seed(12)
    toks = torch.randint(0, arch.gpt.base_vocab, (4, 10), generator=gen).to(DEV)
    lc = eng.decode_begin(pv, 11).clone()
    for t in range(11):
        lf = eng.next_token_logits(pv, toks[:, :t])
        assert float((lc - lf).abs().max()) <= 3e-2, t
        if t < 10:
            lc = eng.decode_advance(toks[:, t]).clone()
    # beam reordering: continuing sequence r from cached sequence src[r] equals decoding that prefix afresh
    src = torch.tensor([2, 2, 0, 1], device=DEV)
    eng.decode_begin(pv, 6)
    eng.decode_advance(toks[:, 0])
    eng.decode_reorder(src)
    got = eng.decode_advance(toks[:, 1]).clone()
    # (the prefix embedding of row r is NOT reordered with the cache: it was consumed at position 0)
    want = eng.next_token_logits(pv[src], torch.stack([toks[src, 0], toks[:, 1]], dim=1))
    assert float((got - want).abs().max()) <= 3e-2


def test_product_score_processing_matches_the_oracle(model):
    """Repetition penalty -> temperature -> nucleus filter as the product applies them before sampling, against the
    oracle's restatement (itself pinned to transformers' processor classes in tests/test_oracle_golden.py): the SAME
    filtered distribution, so sampling differs from HF only by the random stream."""
    gen = torch.Generator().manual_seed(21)
    scores = torch.randn(6, 509, generator=gen) * 3
    ids = torch.randint(0, 509, (6, 7), generator=gen)
    got = model.caption_decoder._process_scores(scores.to(DEV), ids.to(DEV), 1.3, True, 0.7, 0.8).cpu()
    want = R.process_scores(scores.clone(), ids, 1.3, True, 0.7, 0.8)
    assert torch.equal(torch.isinf(got), torch.isinf(want))
    keep = ~torch.isinf(want)
    assert torch.allclose(got[keep], want[keep], atol=1e-5)
    lp = torch.log_softmax(scores, dim=-1)
    got = model.caption_decoder._process_scores(lp.to(DEV), ids.to(DEV), 1.2, False, 1.0, 1.0).cpu()
    assert torch.allclose(got, R.process_scores(lp.clone(), ids, 1.2), atol=1e-6)


def test_sampling_and_beam_contracts(model):
    arch = model.arch
    img = torch.randn(3, 3, arch.vit.image, arch.vit.image, generator=torch.Generator().manual_seed(4))
    pad = arch.gpt.base_vocab
    gen = lambda s: torch.Generator(device=DEV).manual_seed(s)  # noqa: E731
    a = model.generate_token_ids(img, max_length=9, num_beams=1, do_sample=True, top_p=0.9, temperature=0.8, generator=gen(1))
    b = model.generate_token_ids(img, max_length=9, num_beams=1, do_sample=True, top_p=0.9, temperature=0.8, generator=gen(1))
    c = model.generate_token_ids(img, max_length=9, num_beams=1, do_sample=True, top_p=0.9, temperature=0.8, generator=gen(2))
    assert a.shape == (3, 8) and a.dtype == torch.int64 and torch.equal(a, b) and not torch.equal(a, c)
    assert int(a.max()) < arch.dec_vocab
    # beam search: never worse than greedy in total log-probability of the returned sequence
    beams = model.generate_token_ids(img, max_length=6, num_beams=4, do_sample=False, repetition_penalty=1.0)
    greedy = model.generate_token_ids(img, max_length=6, num_beams=1, do_sample=False, repetition_penalty=1.0)
    eng = model.caption_decoder.engine
    pv = eng.prefix_embedding(model.vision_encoder(img)["embeddings"])

    def seq_logp(ids):
        tot = torch.zeros(ids.shape[0], device=DEV)
        for t in range(ids.shape[1]):
            lp = torch.log_softmax(eng.next_token_logits(pv, ids[:, :t]), dim=-1)
            tot += lp.gather(1, ids[:, t:t + 1])[:, 0] * (ids[:, t] != pad)
        return tot
    assert bool((seq_logp(beams) >= seq_logp(greedy) - 1e-3).all())
    # eos stops a sequence and pads the rest
    first = int(greedy[0, 0])
    stopped = model.generate_token_ids(img[:1], max_length=6, num_beams=1, do_sample=False, repetition_penalty=1.0,
                                       eos_token_id=first)
    assert stopped.shape[1] == 1 and int(stopped[0, 0]) == first
    with pytest.raises(RuntimeError, match="tokenizer"):
        model.generate_captions(img)

    class Tok:
        def decode(self, ids, skip_special_tokens=True):
            return " ".join(str(i) for i in ids if i < pad)
    model.caption_decoder.tokenizer = Tok()
    caps = model.generate_captions(img, max_length=4, num_beams=1, do_sample=False)
    assert len(caps) == 3 and all(isinstance(c, str) and c for c in caps)
    del model.caption_decoder.tokenizer


@pytest.mark.parametrize("M,N,K,act,fuse_ln", [(1, 3072, 1024, 0, False), (4, 1024, 1024, 0, True), (16, 4096, 1024, 1, False),
                                               (5, 1024, 4096, 0, True), (33, 1600, 6400, 0, True), (64, 4800, 1600, 0, False),
                                               (3, 264, 200, 1, False)])
def test_skinny_gemm_against_fp32(M, N, K, act, fuse_ln):
    """pgca_gemm_skinny (the decode-step product: split-K over all CUs + per-row finish with bias / gelu_new / residual /
    fused LayerNorm, strided rows) against plain fp32 PyTorch on the same bf16 operands."""
    from pgca_amd import hip as H
    H.load()
    g = torch.Generator().manual_seed(M * 7 + N)
    lda, ld_out = K + 64, N + 24
    xb = (torch.randn(M, lda, generator=g)).bfloat16().to(DEV)
    W = (torch.randn(K, N, generator=g) * 0.05).bfloat16().to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    res = torch.randn(M, N, generator=g).to(DEV)
    gamma, beta = torch.randn(N, generator=g).to(DEV), torch.randn(N, generator=g).to(DEV)
    scratch = torch.full((H.gemm_skinny_workspace(M, N, K) // 4,), float("nan"), device=DEV)   # contents irrelevant
    out_f = torch.zeros(M, N, device=DEV)
    out_b = torch.zeros(M, ld_out, dtype=torch.bfloat16, device=DEV)
    ln_out = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    H.gemm_skinny(xb, W, M, N, K, scratch, lda=lda, bias=bias, act=H.EPI_GELU_NEW if act else H.EPI_NONE, residual=res,
                  out_f32=out_f, out_bf16=out_b, ld_out_bf16=ld_out,
                  ln=(gamma, beta, 1e-5) if fuse_ln else None, ln_out=ln_out if fuse_ln else None)
    v = xb[:, :K].float() @ W.float() + bias
    if act:
        v = 0.5 * v * (1 + torch.tanh(0.7978845608028654 * (v + 0.044715 * v ** 3)))
    v = v + res
    scale = float(v.abs().max())
    assert float((out_f - v).abs().max()) <= 2e-4 * scale + 2e-3 * bool(act)
    assert float((out_b[:, :N].float() - v).abs().max()) <= 2 ** -7 * scale
    if fuse_ln:
        want = torch.nn.functional.layer_norm(out_f, (N,), gamma, beta, 1e-5)
        assert float((ln_out.float() - want).abs().max()) <= 2 ** -7 * float(want.abs().max())


@pytest.mark.parametrize("R", [1, 5, 40])
def test_decode_step_at_real_width(R):
    """GPT-2-M width (H = 1024, 16 heads, 2 layers): the K/V-cache decode - skinny products with the LayerNorms fused into
    their finish passes for R <= 32 rows, tile GEMMs above - against the cache-free recomputation at every position, and
    against the oracle at the last one."""
    from pgca_amd.arch import make_arch, with_layers
    from pgca_amd.model import PreferenceGuidedCaptioningModel
    arch = with_layers(make_arch("openai/clip-vit-base-patch32", "gpt2-medium", 512), 1, 2)
    m = PreferenceGuidedCaptioningModel(freeze_vision_backbone=True, arch=arch, seed=23, device=DEV)
    eng = m.caption_decoder.engine
    gen = torch.Generator().manual_seed(R)
    emb = torch.randn(R, 512, generator=gen)
    toks = torch.randint(0, 50257, (R, 5), generator=gen)
    pv = eng.prefix_embedding(emb.to(DEV))
    worst = 0.0
    for graphs in (False, True, True):
        eng.use_graphs = graphs
        lc = eng.decode_begin(pv, 6).clone()
        for t in range(6):
            lf = eng.next_token_logits(pv, toks[:, :t].to(DEV))
            worst = max(worst, float((lc - lf).abs().max()))
            if t < 5:
                lc = eng.decode_advance(toks[:, t].to(DEV)).clone()
    sd = {k: v.detach().cpu() for k, v in m.store.state_dict(aliases=False).items()}
    if R <= 5:
        from oracle import restatement as Rm
        ref = Rm.generate_step_logits(sd, emb, toks, arch.gpt.heads)
        assert float((lc.cpu() - ref).abs().max()) <= 5e-2
    assert worst <= 3e-2, worst
