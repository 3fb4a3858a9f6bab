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


def _margin_equal(got, want, margins, tol=0.1):
    """identical until the first step whose decision margin is inside bf16 noise (after it the prefixes differ)"""
    assert got.shape[0] == want.shape[0]
    checked = 0
    for b in range(want.shape[0]):
        for t in range(min(got.shape[1], want.shape[1])):
            if float(margins[b, t]) < tol:
                break
            assert int(got[b, t]) == int(want[b, t]), (b, t)
            checked += 1
    return checked


def test_greedy_and_beam_match_the_reference_generate(model, golden):
    """tests/golden/generation.npz: ids from the REFERENCE's own CaptionDecoder.generate (HF generate, K/V cache) on this
    tiny model.  The HIP path (K/V-cache decode) must return them - same length convention (the prefix embedding counts
    towards max_length), same EOS / padding behaviour, HF's beam-search rules - wherever the decision margin (from the
    oracle, teacher-forced) is outside bf16 noise."""
    import json
    arch = model.arch
    g = golden("generation")
    img = torch.from_numpy(g["images"])
    sd = {k: v.detach().cpu() for k, v in model.store.state_dict(aliases=False).items()}
    emb = torch.from_numpy(g["embeddings"])
    pad, eos = int(g["pad"]), int(g["eos"])
    # first-step logits
    pv = model.caption_decoder.engine.prefix_embedding(model.vision_encoder(img)["embeddings"])
    l0 = model.caption_decoder.engine.next_token_logits(pv, torch.zeros(5, 0, dtype=torch.long, device=DEV)).cpu()
    ref0 = R.generate_step_logits(sd, emb, torch.zeros(5, 0, dtype=torch.long), arch.gpt.heads)
    assert float((l0 - ref0).abs().max()) <= 5e-2
    total = 0
    for name in ("greedy", "greedy_norep"):
        kw = json.loads(str(g[name + "_kw"]))
        want, margins = R.generate_greedy(sd, emb, kw["max_length"], arch.gpt.heads, pad, eos, kw["repetition_penalty"])
        assert torch.equal(want, torch.from_numpy(g[name + "_ids"]))
        for cache in (True, False):
            got = model.generate_token_ids(img, use_cache=cache, **kw).cpu()
            assert got.shape == want.shape, (name, got.shape, want.shape)
            total += _margin_equal(got, want, margins)
    for name in ("beam4", "beam3_rep"):
        kw = json.loads(str(g[name + "_kw"]))
        want, gaps = R.generate_beam_search(sd, emb, kw["max_length"], kw["num_beams"], arch.gpt.heads, pad, eos,
                                            kw["repetition_penalty"])
        assert torch.equal(want, torch.from_numpy(g[name + "_ids"]))
        for cache in (True, False):
            got = model.generate_token_ids(img, use_cache=cache, **kw).cpu()
            if float(gaps.min()) >= 0.05:      # no candidate ranking anywhere near a tie: the beams must be identical
                assert torch.equal(got, want), (name, cache)
                total += want.numel()
            else:                               # otherwise image by image, up to the first near-tie of that image
                total += _margin_equal(got, want, gaps.repeat_interleave(1, 0), tol=0.05)
    assert total >= 60, total
    # an EOS that occurs: the caption ends there and is padded, the batch keeps going for the others
    e2 = int(g["eos_case_eos"])
    got = model.generate_token_ids(img, max_length=9, num_beams=1, do_sample=False, repetition_penalty=1.0,
                                   eos_token_id=e2).cpu()
    want = torch.from_numpy(g["greedy_eos_ids"])
    wm, margins = R.generate_greedy(sd, emb, 9, arch.gpt.heads, pad, e2, 1.0)
    assert got.shape == want.shape
    _margin_equal(got, want, margins)
    assert int(got[0, 0]) == e2 and bool((got[0, 1:] == pad).all())


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
