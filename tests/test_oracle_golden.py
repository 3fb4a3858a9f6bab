"""The CPU restatement (oracle/restatement.py) against the reference's own outputs.

Fixtures in tests/golden/ were produced by oracle/make_golden.py from the imported
reference code (see its docstring).  Tolerances are the ones SURVEY.md §8(d) states
for restatement-vs-reference in fp32: loss |d| <= 1e-5, grads max-rel <= 1e-4,
gather indices bit-exact.
"""
import json

import numpy as np
import pytest
import torch

from oracle import restatement as R
from pgca_amd.arch import tiny_arch
from pgca_amd.params import ParamStore

T = torch.from_numpy


def maxrel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("b", [2, 8, 64])
@pytest.mark.parametrize("tau", [0.07, 0.5])
def test_nt_xent_matches_reference(golden, b, tau):
    g = golden("nt_xent")
    k = f"b{b}_t{tau}"
    img = T(g[k + "_img"]).requires_grad_()
    txt = T(g[k + "_txt"]).requires_grad_()
    loss = R.nt_xent(img, txt, tau)
    loss.backward()
    assert abs(float(loss) - float(g[k + "_loss"])) <= 1e-5
    assert maxrel(img.grad, g[k + "_dimg"]) <= 1e-4
    assert maxrel(txt.grad, g[k + "_dtxt"]) <= 1e-4
    # components.ContrastiveLoss (A5') agrees for normalised inputs when tau is inside its clamp
    assert abs(float(R.nt_xent_components(img.detach(), txt.detach(), tau)) - float(g[k + "_loss_components"])) <= 1e-5


def test_global_negatives_equal_concatenated_batch(golden):
    """sharded evaluation (each rank its rows of S and columns of S^t) == single-process loss."""
    g = golden("nt_xent")
    img, txt = T(g["b64_t0.5_img"]), T(g["b64_t0.5_txt"])
    n, world = img.shape[0], 8
    per = n // world
    total = 0.0
    for r in range(world):
        sl = slice(r * per, (r + 1) * per)
        rows = img[sl] @ txt.t() / 0.5          # this rank's rows of S
        cols = txt[sl] @ img.t() / 0.5          # this rank's rows of S^t
        lab = torch.arange(r * per, (r + 1) * per)
        total += float(torch.nn.functional.cross_entropy(rows, lab, reduction="sum")
                       + torch.nn.functional.cross_entropy(cols, lab, reduction="sum"))
    assert abs(total / (2 * n) - float(g["b64_t0.5_loss"])) <= 1e-5


def test_logprob_gather_and_preference_loss(golden):
    g = golden("logprob_dpo")
    lw, ll = T(g["logits_w"]).requires_grad_(), T(g["logits_l"]).requires_grad_()
    iw, il, mw, ml = T(g["ids_w"]), T(g["ids_l"]), T(g["mask_w"]), T(g["mask_l"])
    assert np.array_equal(R.gather_indices(iw).numpy(), g["gather_index_w"])  # bit-exact int64
    assert R.gather_indices(iw).dtype == torch.int64
    np.testing.assert_allclose(R.token_logprobs(lw, iw).detach().numpy(), g["token_lp_w"], atol=1e-5)
    np.testing.assert_allclose(R.sequence_logprob_mean(lw, iw, mw).detach().numpy(), g["seq_mean_w"], atol=1e-5)
    np.testing.assert_allclose(R.sequence_logprob_mean(ll, il, ml).detach().numpy(), g["seq_mean_l"], atol=1e-5)
    np.testing.assert_allclose(R.sequence_logprob_sum(lw, iw, mw).detach().numpy(), g["seq_sum_w"], atol=2e-5)
    np.testing.assert_allclose(R.sequence_logprob_sum(lw, iw, None).detach().numpy(), g["seq_sum_w_nomask"], atol=2e-5)
    loss = R.preference_loss(lw, ll, iw, il, mw, ml, 0.1)
    loss.backward()
    assert abs(float(loss) - float(g["pref_loss"])) <= 1e-5
    assert maxrel(lw.grad, g["dlogits_w"]) <= 1e-4
    assert maxrel(ll.grad, g["dlogits_l"]) <= 1e-4


@pytest.mark.parametrize("name,kw", [("std", {}), ("ls", {"label_smoothing": 0.1}), ("rf", {"reference_free": True})])
def test_dpo_loss(golden, name, kw):
    g = golden("logprob_dpo")
    pc, pr = T(g["dpo_pc"]).requires_grad_(), T(g["dpo_pr"]).requires_grad_()
    loss, metrics = R.dpo_loss(pc, pr, T(g["dpo_rc"]), T(g["dpo_rr"]), beta=0.1, **kw)
    loss.backward()
    assert abs(float(loss) - float(g[f"dpo_{name}_loss"])) <= 1e-5
    assert maxrel(pc.grad, g[f"dpo_{name}_dpc"]) <= 1e-4
    assert maxrel(pr.grad, g[f"dpo_{name}_dpr"]) <= 1e-4
    ref = json.loads(str(g[f"dpo_{name}_metrics"]))
    for k, v in ref.items():
        assert abs(metrics[k] - v) <= 1e-4, k


def test_identical_pairs_give_ln2():
    lp = torch.randn(5)
    loss, _ = R.dpo_loss(lp, lp.clone(), lp, lp.clone())
    assert abs(float(loss) - np.log(2.0)) < 1e-6


def test_optimizer_steps(golden):
    g = golden("optimizer")
    ps = [T(g["p0_init"]).clone(), T(g["p1_init"]).clone()]
    ms = [torch.zeros_like(p) for p in ps]
    vs = [torch.zeros_like(p) for p in ps]
    for step in range(4):
        lr = R.cosine_warmup_lr(5e-5, step, 2, 10)
        assert abs(lr - float(g[f"lr_{step}"])) <= 1e-12
        gs = [T(g[f"g0_{step}"]).clone(), T(g[f"g1_{step}"]).clone()]
        norm = float(torch.sqrt(sum((x.double() ** 2).sum() for x in gs)))
        assert abs(norm - float(g[f"norm_{step}"])) <= 1e-4 * norm
        c = R.clip_coefficient(norm, 1.0)
        for p, gr, m, v in zip(ps, gs, ms, vs):
            R.adamw_step(p, gr * c, m, v, step + 1, lr)
        np.testing.assert_allclose(ps[0].numpy(), g[f"p0_{step}"], rtol=0, atol=2e-7)
        np.testing.assert_allclose(ps[1].numpy(), g[f"p1_{step}"], rtol=0, atol=2e-7)


def golden_vit():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "tiny_vit_grads.npz"), allow_pickle=False)


def check_vit_grads(sd, gv, tag, tol=1e-4):
    """The CLIP tower left trainable (reference model.py:150-164: frozen only on request): gradients of its own
    parameters, recorded from the reference on the same weights and inputs (oracle/make_golden.py:gen_tiny_vit_grads)."""
    n = 0
    for k in gv.files:
        if not k.startswith(tag + "_grad::"):
            continue
        name = k[len(tag) + 7:]
        want = gv[k]
        got = sd[name].grad
        if np.abs(want).max() < 1e-9:      # k_proj.bias: softmax is invariant to a shift of every key score
            assert float(got.abs().max()) < 1e-8, name
        else:
            assert maxrel(got, want) <= tol, name
        n += 1
    pg = sd["vision_encoder.vision_model.embeddings.patch_embedding.weight"].grad
    assert maxrel(pg.reshape(pg.shape[0], -1)[:8], gv[tag + "_grad_patch_rows"]) <= tol
    chk = gv[tag + "_grad_patch_chk"]
    assert abs(float(pg.double().abs().sum()) - chk[1]) <= 1e-3 * chk[1]
    assert n == 17


@pytest.fixture(scope="module")
def tiny(golden):
    g = golden("tiny_e2e")
    arch = tiny_arch()
    store = ParamStore(arch, "cpu", seed=int(g["seed"]), frozen=())
    for seg in store.segments.values():  # same weights as the fixture was made with
        chk = g[f"chk_{seg.name}"]
        assert abs(float(seg.fp32.double().sum()) - chk[0]) <= 1e-6 * max(1.0, abs(chk[0]))
        assert abs(float(seg.fp32.double().abs().sum()) - chk[1]) <= 1e-6 * chk[1]
    sd = {k: v.clone().requires_grad_() for k, v in store.state_dict(aliases=False).items()}
    return g, arch, sd


def test_tiny_stage1_end_to_end(tiny):
    g, arch, sd = tiny
    out = R.model_forward(sd, T(g["images"]), T(g["ids_w"]), T(g["mask_w"]), "contrastive",
                          arch.vit.heads, arch.vit.patch, arch.gpt.heads)
    np.testing.assert_allclose(out["image_embeddings"].detach().numpy(), g["s1_image_embeddings"], atol=2e-5)
    np.testing.assert_allclose(out["text_embeddings"].detach().numpy(), g["s1_text_embeddings"], atol=2e-5)
    np.testing.assert_allclose(out["vision_features"].detach().numpy(), g["s1_vision_features"], atol=1e-4)
    valid = g["mask_w"].astype(bool)  # padded query rows are unconstrained (later masked)
    np.testing.assert_allclose(out["text_features"].detach().numpy()[valid], g["s1_text_features"][valid], atol=1e-4)
    loss = R.nt_xent(out["image_embeddings"], out["text_embeddings"], 0.5)
    assert abs(float(loss) - float(g["s1_loss"])) <= 1e-5
    loss.backward()
    for k in g.files:
        if k.startswith("s1_grad::"):
            assert maxrel(sd[k[len("s1_grad::"):]].grad, g[k]) <= 1e-4, k
    assert maxrel(sd["text_encoder.text_model.wte.weight"].grad[:64], g["s1_grad_wte_rows"]) <= 1e-4
    check_vit_grads(sd, golden_vit(), "s1")
    for k, v in sd.items():
        v.grad = None


def test_tiny_stage2_two_forward(tiny):
    g, arch, sd = tiny
    img, iw, il, mw, ml = T(g["images"]), T(g["ids_w"]), T(g["ids_l"]), T(g["mask_w"]), T(g["mask_l"])
    ow = R.model_forward(sd, img, iw, mw, "generation", arch.vit.heads, arch.vit.patch, arch.gpt.heads)["logits"]
    ol = R.model_forward(sd, img, il, ml, "generation", arch.vit.heads, arch.vit.patch, arch.gpt.heads)["logits"]
    vw, vl = g["mask_w"].astype(bool), g["mask_l"].astype(bool)
    np.testing.assert_allclose(ow.detach().numpy()[vw], g["s2_logits_w"][vw], atol=2e-4)
    np.testing.assert_allclose(ol.detach().numpy()[vl], g["s2_logits_l"][vl], atol=2e-4)
    loss = R.preference_loss(ow, ol, iw, il, mw, ml, 0.1)
    assert abs(float(loss) - float(g["s2_pref_loss"])) <= 1e-5
    loss.backward()
    for k in g.files:
        if k.startswith("s2_grad::"):
            assert maxrel(sd[k[len("s2_grad::"):]].grad, g[k]) <= 1e-4, k
    wte = sd["caption_decoder.lm_model.transformer.wte.weight"].grad
    assert maxrel(wte, g["s2_grad_wte"]) <= 1e-4
    # softmax over ONE key == 1 => q/k rows of in_proj get exactly zero gradient
    h = arch.gpt.hidden
    gi = sd["caption_decoder.cross_attention.in_proj_weight"].grad
    assert float(gi[:2 * h].abs().max()) == 0.0 and float(gi[2 * h:].abs().max()) > 0.0
    # no gradient reaches the text tower in generation mode
    assert int(g["s2_text_tower_params_with_grad"]) == 0
    assert all(v.grad is None for k, v in sd.items() if k.startswith("text_encoder."))
    check_vit_grads(sd, golden_vit(), "s2")
    for k, v in sd.items():
        v.grad = None


def test_tiny_stage2_four_forward_dpo(tiny):
    g, arch, sd = tiny
    img, iw, il, mw, ml = T(g["images"]), T(g["ids_w"]), T(g["ids_l"]), T(g["mask_w"]), T(g["mask_l"])
    ow = R.model_forward(sd, img, iw, mw, "generation", arch.vit.heads, arch.vit.patch, arch.gpt.heads)["logits"]
    ol = R.model_forward(sd, img, il, ml, "generation", arch.vit.heads, arch.vit.patch, arch.gpt.heads)["logits"]
    pw, pl = R.sequence_logprob_sum(ow, iw, mw), R.sequence_logprob_sum(ol, il, ml)
    np.testing.assert_allclose(pw.detach().numpy(), g["s2_pol_w"], rtol=1e-5, atol=1e-4)
    loss, metrics = R.dpo_loss(pw, pl, T(g["s2_ref_w"]), T(g["s2_ref_l"]), beta=0.1)
    assert abs(float(loss) - float(g["s2_dpo_loss"])) <= 1e-5
    loss.backward()
    for k in g.files:
        if k.startswith("s2dpo_grad::"):
            assert maxrel(sd[k[len("s2dpo_grad::"):]].grad, g[k]) <= 1e-4, k
    assert maxrel(sd["caption_decoder.lm_model.transformer.wte.weight"].grad, g["s2dpo_grad_wte"]) <= 1e-4
    for k, v in sd.items():
        v.grad = None


def test_cross_attention_collapse(tiny):
    """K9: with one key the attended rows are identical for every query position and equal
    W_o(W_v pv + b_v) + b_o - the identity the HIP path exploits."""
    _, arch, sd = tiny
    h = arch.gpt.hidden
    q = torch.randn(3, 7, h)
    kv = torch.randn(3, 1, h)
    full = R.cross_attention_one_key(sd, "caption_decoder.cross_attention", q, kv, arch.xattn_heads)
    w, b = sd["caption_decoder.cross_attention.in_proj_weight"], sd["caption_decoder.cross_attention.in_proj_bias"]
    v = kv[:, 0] @ w[2 * h:].t() + b[2 * h:]
    col = v @ sd["caption_decoder.cross_attention.out_proj.weight"].t() + sd["caption_decoder.cross_attention.out_proj.bias"]
    np.testing.assert_allclose(full.detach().numpy(), col[:, None, :].expand(3, 7, h).detach().numpy(), atol=1e-5)


# ------------------------------------------------------------------------------------------------ generation
def test_generation_restatement_reproduces_the_reference_generate(golden):
    """tests/golden/generation.npz holds ids produced by the REFERENCE's own CaptionDecoder.generate (model.py:621-678 ->
    HF generate with its K/V cache) on the tiny model: greedy and deterministic beam search, with and without an EOS that
    occurs.  The restatement (cache-free, HF's rules restated) must return the same ids."""
    from pgca_amd.arch import tiny_arch
    from pgca_amd.params import ParamStore
    g = golden("generation")
    arch = tiny_arch()
    store = ParamStore(arch, "cpu", seed=int(g["seed"]), frozen=())
    sd = {k: v.clone() for k, v in store.state_dict(aliases=False).items()}
    emb = torch.from_numpy(g["embeddings"])
    got_emb = R.vision_encoder_forward(sd, torch.from_numpy(g["images"]), arch.vit.heads, arch.vit.patch)["embeddings"]
    np.testing.assert_allclose(got_emb.detach().numpy(), g["embeddings"], atol=1e-4)
    pad, eos = int(g["pad"]), int(g["eos"])
    import json
    for name in ("greedy", "greedy_norep"):
        kw = json.loads(str(g[name + "_kw"]))
        ids, _ = R.generate_greedy(sd, emb, kw["max_length"], arch.gpt.heads, pad, eos, kw["repetition_penalty"])
        assert np.array_equal(ids.numpy(), g[name + "_ids"]), name
    for name in ("beam4", "beam3_rep"):
        kw = json.loads(str(g[name + "_kw"]))
        ids, _ = R.generate_beam_search(sd, emb, kw["max_length"], kw["num_beams"], arch.gpt.heads, pad, eos,
                                        kw["repetition_penalty"])
        assert np.array_equal(ids.numpy(), g[name + "_ids"]), name
    e2 = int(g["eos_case_eos"])
    ids, _ = R.generate_greedy(sd, emb, 9, arch.gpt.heads, pad, e2, 1.0)
    assert np.array_equal(ids.numpy(), g["greedy_eos_ids"])
    ids, _ = R.generate_beam_search(sd, emb, 8, 4, arch.gpt.heads, pad, e2, 1.0)
    assert np.array_equal(ids.numpy(), g["beam4_eos_ids"])


def test_score_processors_match_the_installed_transformers_classes():
    """The repetition-penalty / temperature / top-p restatement against transformers' own processor classes."""
    from transformers.generation.logits_process import (RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper,
                                                        TopPLogitsWarper)
    gen = torch.Generator().manual_seed(8)
    scores = torch.randn(6, 97, generator=gen) * 3
    ids = torch.randint(0, 97, (6, 5), generator=gen)
    want = RepetitionPenaltyLogitsProcessor(1.3)(ids, scores.clone())
    want = TopPLogitsWarper(0.8)(ids, TemperatureLogitsWarper(0.7)(ids, want))
    got = R.process_scores(scores.clone(), ids, 1.3, True, 0.7, 0.8)
    assert torch.equal(torch.isinf(got), torch.isinf(want))
    fin = ~torch.isinf(want)
    assert torch.allclose(got[fin], want[fin], atol=1e-6)
    lp = torch.log_softmax(scores, dim=-1)              # beam search feeds log-probabilities through the same classes
    assert torch.allclose(R.process_scores(lp.clone(), ids, 1.3), RepetitionPenaltyLogitsProcessor(1.3)(ids, lp.clone()))
