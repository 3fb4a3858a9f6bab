"""End-to-end parity of the HIP path (through the model surface and the step engines).

* tiny geometry: against the REFERENCE's own outputs/gradients (tests/golden/tiny_e2e.npz, produced
  by oracle/make_golden.py from the imported reference) - same seeded weights, same batch.
* config-2 geometry (ViT-B/32 + GPT-2-M, S=128): against the oracle restatement run on the host.

Tolerances (SURVEY 8d, bf16 MFMA operands / f32 accumulate): per-sequence mean log-prob |d| <= 2e-2,
loss |d| <= 5e-3, gradient cosine >= 0.99; gather indices bit-exact.
"""
import numpy as np
import pytest
import torch

from oracle import restatement as R

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def cos(a, b):
    a, b = a.double().flatten().cpu(), torch.as_tensor(b).double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


@pytest.fixture(scope="module")
def tiny(golden):
    from pgca_amd.arch import tiny_arch
    from pgca_amd.model import PreferenceGuidedCaptioningModel
    g = golden("tiny_e2e")
    model = PreferenceGuidedCaptioningModel(freeze_vision_backbone=True, arch=tiny_arch(), seed=int(g["seed"]),
                                            device="cuda:0")
    for seg in model.store.segments.values():
        chk = g[f"chk_{seg.name}"]
        assert abs(float(seg.fp32.double().sum()) - chk[0]) <= 1e-5 * max(1.0, abs(chk[0]))
    return g, model


def dpo_step(model, reference_free, ref=None):
    from pgca_amd.steps import DPOStep
    return DPOStep(model.store, model.ws, model.vision_encoder.tower, model.vision_encoder.head,
                   model.caption_decoder.engine, beta=0.1, reference_free=reference_free, ref=ref)


def batch2(g, dev):
    from pgca_amd.steps import DPOStep
    b = {"image": T(g["images"]), "preferred_ids": T(g["ids_w"]), "rejected_ids": T(g["ids_l"]),
         "preferred_mask": T(g["mask_w"]), "rejected_mask": T(g["mask_l"])}
    return DPOStep.prepare(b, dev)


def test_gather_indices_bit_exact(tiny):
    g, model = tiny
    p = batch2(g, model.device)
    sb = p["seq"]
    ids = np.concatenate([g["ids_w"], g["ids_l"]])
    mask = np.concatenate([g["mask_w"], g["mask_l"]])
    keep = mask[:, 1:] != 0
    assert sb.targets.dtype == torch.int64
    assert np.array_equal(sb.targets.cpu().numpy(), ids[:, 1:][keep])          # == labels[:, 1:] where scored
    bi, ti = np.nonzero(keep)
    assert np.array_equal(sb.row_map.cpu().numpy(), (bi * ids.shape[1] + ti).astype(np.int32))
    assert np.array_equal(sb.counts.cpu().numpy(), keep.sum(1).astype(np.int32))


def test_tiny_forward_surface_matches_reference(tiny):
    g, model = tiny
    img, iw, mw = T(g["images"]), T(g["ids_w"]), T(g["mask_w"])
    out = model(images=img, caption_ids=iw, caption_mask=mw, mode="contrastive")
    assert set(out) == {"image_embeddings", "text_embeddings", "vision_features", "text_features"}
    np.testing.assert_allclose(out["image_embeddings"].cpu().numpy(), g["s1_image_embeddings"], atol=1.5e-2)
    np.testing.assert_allclose(out["text_embeddings"].cpu().numpy(), g["s1_text_embeddings"], atol=1.5e-2)
    np.testing.assert_allclose(out["vision_features"].cpu().numpy(), g["s1_vision_features"], atol=6e-2)
    gen = model(images=img, caption_ids=iw, caption_mask=mw, labels=iw, mode="generation")
    assert set(gen) == {"logits", "generation_loss"}
    valid = g["mask_w"].astype(bool)
    np.testing.assert_allclose(gen["logits"].cpu().numpy()[valid], g["s2_logits_w"][valid], atol=5e-2)
    lp = model.sequence_logprobs(img, iw, mw, reduce="sum")
    np.testing.assert_allclose(lp.cpu().numpy(), g["s2_seq_sum_w"], atol=2e-2 * 16)


def test_tiny_input_validation(tiny):
    _, model = tiny
    long_ids = torch.zeros(1, 65, dtype=torch.long)          # tiny-gpt2 has 64 learned positions
    with pytest.raises(RuntimeError, match="learned positions"):
        model.text_encoder(long_ids, torch.ones_like(long_ids))
    with pytest.raises(ValueError, match="4D tensor"):
        model.vision_encoder(torch.zeros(3, 64, 64))
    with pytest.raises(ValueError, match="3 channels"):
        model.vision_encoder(torch.zeros(1, 1, 64, 64))
    with pytest.raises(ValueError, match="doesn't match"):
        model.text_encoder(torch.zeros(2, 8, dtype=torch.long), torch.ones(2, 9, dtype=torch.long))


def test_tiny_stage2_two_forward_matches_reference(tiny):
    g, model = tiny
    step = dpo_step(model, reference_free=True)
    p = batch2(g, model.device)
    for seg in model.store.trainable_segments():
        seg.grad.zero_()
    loss = step.loss_and_grads(p["image"], p["seq"])
    assert abs(float(loss) - float(g["s2_pref_loss"])) <= 5e-3
    for k in g.files:
        if k.startswith("s2_grad::"):
            name = k[len("s2_grad::"):]
            c = cos(model.store.g(name), g[k])
            if np.abs(g[k]).max() == 0:
                assert float(model.store.g(name).abs().max()) == 0.0, name
            else:
                assert c >= 0.99, f"{name}: cosine {c}"
    assert cos(model.store.g("caption_decoder.lm_model.transformer.wte.weight"), g["s2_grad_wte"]) >= 0.99
    # q/k rows of the collapsed cross-attention receive exactly zero (not missing) gradient
    H = model.arch.gpt.hidden
    gi = model.store.g("caption_decoder.cross_attention.in_proj_weight")
    assert float(gi[:2 * H].abs().max()) == 0.0 and float(gi[2 * H:].abs().max()) > 0
    # nothing reaches the text tower in Stage 2
    assert float(model.store.segments["text_tower"].grad.abs().max()) == 0.0


def test_tiny_stage2_four_forward_dpo_matches_reference(tiny):
    from pgca_amd.steps import ReferencePolicy
    g, model = tiny
    ref = ReferencePolicy(model.store, model.ws)
    seg = ref.store.segments["decoder"]
    for name in seg.index:  # the fixture's reference policy: snapshot with mlp.c_proj scaled by 0.9
        if ".h." in name and name.endswith("mlp.c_proj.weight"):
            seg.w(name).mul_(0.9)
    seg.ensure_bf16()
    step = dpo_step(model, reference_free=False, ref=ref)
    p = batch2(g, model.device)
    for s in model.store.trainable_segments():
        s.grad.zero_()
    loss = step.loss_and_grads(p["image"], p["seq"])
    B = 4
    pol = step.dec.ws.bufs["pol.seq_lp"][:2 * B].cpu().numpy()
    np.testing.assert_allclose(pol[:B], g["s2_pol_w"], atol=2e-2 * 16)   # length-sum: 2e-2 per token
    assert abs(float(loss) - float(g["s2_dpo_loss"])) <= 5e-3
    for k in g.files:
        if k.startswith("s2dpo_grad::"):
            name = k[len("s2dpo_grad::"):]
            assert cos(model.store.g(name), g[k]) >= 0.99, name
    assert cos(model.store.g("caption_decoder.lm_model.transformer.wte.weight"), g["s2dpo_grad_wte"]) >= 0.99
    import json
    met = step.metrics.cpu().numpy()
    refm = json.loads(str(g["s2_dpo_metrics"]))
    assert abs(met[0] - refm["reward_margin"]) <= 0.3 and abs(met[2] - refm["policy_chosen_logprob"]) <= 0.3


def test_tiny_stage1_matches_reference(tiny):
    from pgca_amd.steps import ContrastiveStep
    g, model = tiny
    step = ContrastiveStep(model.store, model.ws, model.vision_encoder.tower, model.vision_encoder.head,
                           model.text_encoder.engine, temperature=0.5)
    p = ContrastiveStep.prepare({"image": T(g["images"]), "caption_ids": T(g["ids_w"]), "caption_mask": T(g["mask_w"])},
                                model.device)
    for s in model.store.trainable_segments():
        s.grad.zero_()
    loss = step.loss_and_grads(p["image"], p["ids"], p["mask"])
    assert abs(float(loss) - float(g["s1_loss"])) <= 5e-3
    for k in g.files:
        if k.startswith("s1_grad::"):
            name = k[len("s1_grad::"):]
            assert cos(model.store.g(name), g[k]) >= 0.99, name
    assert cos(model.store.g("text_encoder.text_model.wte.weight")[:64], g["s1_grad_wte_rows"]) >= 0.99
    assert float(model.store.segments["decoder"].grad.abs().max()) == 0.0


def test_optimizer_step_changes_only_trainable_segments(tiny):
    from pgca_amd.steps import FusedOptimizer
    g, model = tiny
    step = dpo_step(model, reference_free=True)
    p = batch2(g, model.device)
    segs = [model.store.segments["vision_head"], model.store.segments["decoder"]]
    opt = FusedOptimizer(segs, lr=1e-3, max_grad_norm=1.0, warmup_steps=0, total_steps=10)
    before = {n: s.fp32.clone() for n, s in model.store.segments.items()}
    losses = []
    for _ in range(3):
        opt.zero_grad()
        losses.append(float(step.loss_and_grads(p["image"], p["seq"])))
        opt.step()
    st = opt.state()
    assert st["finite"] and st["step"] == 3 and st["clip"] <= 1.0
    assert losses[-1] < losses[0]                       # the same batch three times: the loss must fall
    assert not torch.equal(model.store.segments["decoder"].fp32, before["decoder"])
    assert torch.equal(model.store.segments["vit"].fp32, before["vit"])
    assert torch.equal(model.store.segments["text_tower"].fp32, before["text_tower"])
    assert torch.equal(model.store.segments["decoder"].bf16, model.store.segments["decoder"].fp32.bfloat16())
    for n, s in model.store.segments.items():            # restore for any later test
        s.fp32.copy_(before[n])
        s.ensure_bf16()


@pytest.mark.parametrize("text_model", ["gpt2-medium", "gpt2-large", "gpt2-xl"])
def test_config2_geometry_against_oracle(text_model):
    """ViT-B/32 + GPT-2-M widths at S = 128 with 2 layers each (full depth is the bench's job):
    fused log-probs, DPO loss and gradients vs the CPU restatement on identical weights/batch.
    gpt2-large / gpt2-xl: the decoder widths of configs C4 / C5 (1280 x 20 heads, 1600 x 25 heads) through the same
    kernels (LayerNorm NV = 5 / 7, K = 1280 / 1600 / 5120 / 6400 GEMMs, 200-wide cross-attention heads)."""
    from pgca_amd.arch import make_arch, with_layers
    from pgca_amd.model import PreferenceGuidedCaptioningModel
    from pgca_amd.steps import DPOStep
    arch = with_layers(make_arch("openai/clip-vit-base-patch32", text_model, 512), 2, 2)
    model = PreferenceGuidedCaptioningModel(freeze_vision_backbone=True, arch=arch, seed=7, device="cuda:0")
    gen = torch.Generator().manual_seed(1234)
    B, S = 2, 128
    img = torch.randn(B, 3, 224, 224, generator=gen)
    ids = torch.randint(0, 50257, (2 * B, S), generator=gen)
    lens = torch.tensor([128, 40, 77, 16])
    mask = (torch.arange(S)[None] < lens[:, None]).long()
    ids = torch.where(mask.bool(), ids, torch.full_like(ids, 50257))
    batch = {"image": img, "preferred_ids": ids[:B], "rejected_ids": ids[B:], "preferred_mask": mask[:B],
             "rejected_mask": mask[B:]}
    step = dpo_step(model, reference_free=True)
    p = DPOStep.prepare(batch, model.device)
    for s in model.store.trainable_segments():
        s.grad.zero_()
    loss = step.loss_and_grads(p["image"], p["seq"])
    pol = model.ws.bufs["pol.seq_lp"][:2 * B].cpu()
    # oracle on the host
    sd = {k: v.detach().cpu().clone().requires_grad_(k.startswith(("caption_decoder", "vision_encoder.projection")))
          for k, v in model.store.state_dict(aliases=False).items()}
    lw = R.model_forward(sd, img, ids[:B], mask[:B], "generation", arch.vit.heads, arch.vit.patch, arch.gpt.heads)["logits"]
    ll = R.model_forward(sd, img, ids[B:], mask[B:], "generation", arch.vit.heads, arch.vit.patch, arch.gpt.heads)["logits"]
    ref_w, ref_l = R.sequence_logprob_mean(lw, ids[:B], mask[:B]), R.sequence_logprob_mean(ll, ids[B:], mask[B:])
    ref_loss = R.preference_loss(lw, ll, ids[:B], ids[B:], mask[:B], mask[B:], 0.1)
    ref_loss.backward()
    assert float((pol[:B] - ref_w.detach()).abs().max()) <= 2e-2
    assert float((pol[B:] - ref_l.detach()).abs().max()) <= 2e-2
    assert abs(float(loss) - float(ref_loss)) <= 5e-3
    dec = "caption_decoder.lm_model.transformer."
    for name in (dec + "h.1.attn.c_attn.weight", dec + "h.0.mlp.c_fc.weight", dec + "h.1.mlp.c_proj.weight",
                 dec + "h.0.attn.c_proj.bias", dec + "h.0.ln_1.weight", dec + "ln_f.weight", dec + "wpe.weight",
                 dec + "wte.weight", "caption_decoder.vision_projection.0.weight",
                 "caption_decoder.cross_attention.out_proj.weight", "caption_decoder.attention_norm.weight",
                 "vision_encoder.projection.0.weight", "vision_encoder.projection.4.weight"):
        c = cos(model.store.g(name), sd[name].grad)
        assert c >= 0.99, f"{name}: cosine {c}"

@pytest.mark.parametrize("text_model", ["gpt2-large", "gpt2-xl"])
def test_config45_geometry_against_oracle(text_model):
    """BASELINE configs C4 / C5: CLIP ViT-L/14 (patch 14 -> K = 588 padded to 640, 257 tokens -> key-tiled attention,
    16 heads, MLP 4096) + GPT-2-Large / GPT-2-XL decoder at S = 256 (key-tiled causal attention forward and backward),
    2 layers each: image embeddings, fused log-probs, preference loss and gradients vs the CPU restatement."""
    from pgca_amd.arch import make_arch, with_layers
    from pgca_amd.model import PreferenceGuidedCaptioningModel
    from pgca_amd.steps import DPOStep
    arch = with_layers(make_arch("openai/clip-vit-large-patch14", text_model, 512), 2, 2)
    assert arch.vit.tokens == 257 and arch.vit.patch_dim == 588
    model = PreferenceGuidedCaptioningModel(freeze_vision_backbone=True, arch=arch, seed=9, device="cuda:0")
    gen = torch.Generator().manual_seed(4321)
    B, S = 2, 256
    img = torch.randn(B, 3, 224, 224, generator=gen)
    ids = torch.randint(0, 50257, (2 * B, S), generator=gen)
    lens = torch.tensor([256, 130, 77, 200])
    mask = (torch.arange(S)[None] < lens[:, None]).long()
    ids = torch.where(mask.bool(), ids, torch.full_like(ids, 50257))
    batch = {"image": img, "preferred_ids": ids[:B], "rejected_ids": ids[B:], "preferred_mask": mask[:B],
             "rejected_mask": mask[B:]}
    sd = {k: v.detach().cpu().clone().requires_grad_(k.startswith(("caption_decoder", "vision_encoder.projection")))
          for k, v in model.store.state_dict(aliases=False).items()}
    # vision tower on its own first (frozen, forward only)
    vis = model.vision_encoder(img)
    ref_vis = R.vision_encoder_forward(sd, img, arch.vit.heads, arch.vit.patch)
    assert vis["features"].shape == (B, 257, 1024)
    assert cos(vis["features"], ref_vis["features"].detach()) >= 0.999
    np.testing.assert_allclose(vis["embeddings"].cpu().numpy(), ref_vis["embeddings"].detach().numpy(), atol=4e-2)
    step = dpo_step(model, reference_free=True)
    p = DPOStep.prepare(batch, model.device)
    for s in model.store.trainable_segments():
        s.grad.zero_()
    loss = step.loss_and_grads(p["image"], p["seq"])
    pol = model.ws.bufs["pol.seq_lp"][:2 * B].cpu()
    lw = R.model_forward(sd, img, ids[:B], mask[:B], "generation", arch.vit.heads, arch.vit.patch, arch.gpt.heads)["logits"]
    ll = R.model_forward(sd, img, ids[B:], mask[B:], "generation", arch.vit.heads, arch.vit.patch, arch.gpt.heads)["logits"]
    ref_w, ref_l = R.sequence_logprob_mean(lw, ids[:B], mask[:B]), R.sequence_logprob_mean(ll, ids[B:], mask[B:])
    ref_loss = R.preference_loss(lw, ll, ids[:B], ids[B:], mask[:B], mask[B:], 0.1)
    ref_loss.backward()
    assert float((pol[:B] - ref_w.detach()).abs().max()) <= 2e-2
    assert float((pol[B:] - ref_l.detach()).abs().max()) <= 2e-2
    assert abs(float(loss) - float(ref_loss)) <= 5e-3
    dec = "caption_decoder.lm_model.transformer."
    for name in (dec + "h.1.attn.c_attn.weight", dec + "h.0.attn.c_attn.weight", dec + "h.0.attn.c_attn.bias",
                 dec + "h.0.mlp.c_fc.weight", dec + "h.1.mlp.c_proj.weight", dec + "h.0.attn.c_proj.bias",
                 dec + "h.0.ln_1.weight", dec + "ln_f.weight", dec + "wpe.weight", dec + "wte.weight",
                 "caption_decoder.vision_projection.0.weight", "caption_decoder.cross_attention.out_proj.weight",
                 "caption_decoder.attention_norm.weight", "vision_encoder.projection.0.weight",
                 "vision_encoder.projection.4.weight"):
        c = cos(model.store.g(name), sd[name].grad)
        assert c >= 0.99, f"{name}: cosine {c}"


@pytest.mark.parametrize("train_mode,tau", [(False, 0.5), (True, 0.5), (False, 0.07)])
def test_stage1_config_geometry_against_oracle(train_mode, tau):
    """Stage 1 at the real widths (BASELINE configs C1 / C3: ViT-B/32 + GPT-2-M, S = 128, P = 512), 2 layers each:
    masked mean, ProjHead 1024 -> 512, text-tower backward at H = 1024 and the NT-Xent backward, in eval mode and in
    train mode (dropout 0.1 with the oracle replaying the same counter-based masks).  tau = 0.5 is what the trainer
    runs (configs/default.yaml:21); tau = 0.07 is the constructor default (model.py:958).

    Conditioning: with N(0, 0.02) weights the class token of the frozen ViT is almost the same for every image
    (its image-dependent part is ~4 % of its norm, the size of the bf16 tower error), and the contrastive gradient of
    the vision head is exactly the part that survives cancellation of the common component - a comparison of noise.
    The ViT's value / output projections are therefore scaled x4 (in the model AND, through the shared state_dict, in
    the oracle) so the tower distinguishes images the way a trained one does; nothing else is touched."""
    from pgca_amd.arch import make_arch, with_layers
    from pgca_amd.engine import DropoutPlan
    from pgca_amd.model import PreferenceGuidedCaptioningModel
    from pgca_amd.steps import ContrastiveStep
    arch = with_layers(make_arch("openai/clip-vit-base-patch32", "gpt2-medium", 512), 2, 2)
    model = PreferenceGuidedCaptioningModel(freeze_vision_backbone=True, arch=arch, seed=13, device="cuda:0")
    vit = model.store.segments["vit"]
    for name in vit.index:
        if name.endswith(("self_attn.v_proj.weight", "self_attn.out_proj.weight")):
            vit.w(name).mul_(4.0)
    vit.ensure_bf16()
    gen = torch.Generator().manual_seed(77)
    B, S = 4, 128
    img = torch.randn(B, 3, 224, 224, generator=gen)
    ids = torch.randint(0, 50257, (B, S), generator=gen)
    lens = torch.tensor([128, 40, 77, 16])
    mask = (torch.arange(S)[None] < lens[:, None]).long()
    ids = torch.where(mask.bool(), ids, torch.full_like(ids, 50257))
    seed, pdrop = 991, 0.1
    plan = DropoutPlan(pdrop if train_mode else 0.0, seed)
    step = ContrastiveStep(model.store, model.ws, model.vision_encoder.tower, model.vision_encoder.head,
                           model.text_encoder.engine, temperature=tau, dropout=plan)
    p = ContrastiveStep.prepare({"image": img, "caption_ids": ids, "caption_mask": mask}, model.device)
    for s in model.store.trainable_segments():
        s.grad.zero_()
    loss = float(step.loss_and_grads(p["image"], p["ids"], p["mask"]))
    sd = {k: v.detach().cpu().clone().requires_grad_(k.startswith(("text_encoder", "vision_encoder.projection")))
          for k, v in model.store.state_dict(aliases=False).items()}
    drop = R.Dropper(seed, pdrop, step=0) if train_mode else None
    ie = R.vision_encoder_forward(sd, img, arch.vit.heads, arch.vit.patch, drop)["embeddings"]
    tx = R.text_encoder_forward(sd, ids, mask, arch.gpt.heads, drop)
    ref = R.nt_xent(torch.nn.functional.normalize(ie, dim=-1), torch.nn.functional.normalize(tx["embeddings"], dim=-1), tau)
    ref.backward()
    assert abs(loss - float(ref)) <= 5e-3, (loss, float(ref))
    tt = "text_encoder.text_model."
    for name in (tt + "h.1.attn.c_attn.weight", tt + "h.0.attn.c_proj.weight", tt + "h.0.mlp.c_fc.weight",
                 tt + "h.1.mlp.c_proj.weight", tt + "h.0.mlp.c_proj.bias", tt + "h.0.ln_1.weight", tt + "ln_f.weight",
                 tt + "wpe.weight", tt + "wte.weight", "text_encoder.projection.0.weight",
                 "text_encoder.projection.3.weight", "text_encoder.projection.4.weight",
                 "vision_encoder.projection.0.weight", "vision_encoder.projection.3.bias"):
        c = cos(model.store.g(name), sd[name].grad)
        assert c >= 0.99, f"{name}: cosine {c}"
    assert float(model.store.segments["decoder"].grad.abs().max()) == 0.0


def test_tiny_dual_mode_text_features_and_generation_loss(tiny):
    """mode="dual" returns the union of both modes' keys (reference model.py:814-853); text_features are the
    GPT-2 last_hidden_state; generation_loss is HF's shifted mean cross-entropy (modeling_gpt2.py:700-716)."""
    g, model = tiny
    img, iw, mw = T(g["images"]), T(g["ids_w"]), T(g["mask_w"])
    out = model(images=img, caption_ids=iw, caption_mask=mw, labels=iw, mode="dual")
    assert set(out) == {"image_embeddings", "text_embeddings", "vision_features", "text_features", "logits",
                        "generation_loss"}
    np.testing.assert_allclose(out["image_embeddings"].cpu().numpy(), g["s1_image_embeddings"], atol=1.5e-2)
    np.testing.assert_allclose(out["text_embeddings"].cpu().numpy(), g["s1_text_embeddings"], atol=1.5e-2)
    valid = g["mask_w"].astype(bool)
    np.testing.assert_allclose(out["logits"].cpu().numpy()[valid], g["s2_logits_w"][valid], atol=5e-2)
    sd = {k: v.detach().cpu() for k, v in model.store.state_dict(aliases=False).items()}
    arch = model.arch
    tx = R.text_encoder_forward(sd, iw, mw, arch.gpt.heads)
    tf = out["text_features"].cpu()
    assert tf.shape == tx["features"].shape
    assert cos(tf[T(g["mask_w"]).bool()], tx["features"][T(g["mask_w"]).bool()]) >= 0.999
    # HF ForCausalLMLoss on the reference's logits: mean CE over ALL shifted positions (labels = ids, no ignore index)
    lw = torch.from_numpy(g["s2_logits_w"])
    want = torch.nn.functional.cross_entropy(lw[:, :-1].reshape(-1, lw.shape[-1]), iw[:, 1:].reshape(-1))
    assert abs(float(out["generation_loss"]) - float(want)) <= 2e-2


def test_train_mode_dropout_matches_oracle_with_same_masks(tiny):
    """Train mode (p = 0.1 at every reference dropout site): the HIP path and the oracle use the same counter-based
    masks, so loss and gradients must agree exactly as in eval mode - including the backward's mask replay."""
    from pgca_amd.engine import DropoutPlan
    from pgca_amd.steps import ContrastiveStep, DPOStep
    g, model = tiny
    arch = model.arch
    B = 4
    seed, p = 4242, 0.1
    sd = {k: v.detach().cpu().clone().requires_grad_(not k.startswith("vision_encoder.vision_model"))
          for k, v in model.store.state_dict(aliases=False).items()}
    img = T(g["images"])
    ids2 = torch.cat([T(g["ids_w"]), T(g["ids_l"])])
    mask2 = torch.cat([T(g["mask_w"]), T(g["mask_l"])])

    # ---- Stage 2 (2-forward, mean log-prob), dropout step counter = 3
    plan = DropoutPlan(p, seed)
    plan.step = 3
    step = DPOStep(model.store, model.ws, model.vision_encoder.tower, model.vision_encoder.head,
                   model.caption_decoder.engine, beta=0.1, reference_free=True, dropout=plan)
    pb = batch2(g, model.device)
    for s in model.store.trainable_segments():
        s.grad.zero_()
    loss = float(step.loss_and_grads(pb["image"], pb["seq"]))
    assert plan.step == 4
    drop = R.Dropper(seed, p, step=3)
    emb = R.vision_encoder_forward(sd, img, arch.vit.heads, arch.vit.patch, drop)["embeddings"]
    logits = R.caption_decoder_logits(sd, torch.cat([emb, emb]), ids2, mask2, arch.gpt.heads, drop)
    lp = R.sequence_logprob_mean(logits, ids2, mask2)
    ref = -torch.nn.functional.logsigmoid(0.1 * (lp[:B] - lp[B:])).mean()
    ref.backward()
    assert abs(loss - float(ref)) <= 5e-3
    # eval-mode value differs: the masks really were applied
    assert abs(float(step.loss_only(pb["image"], pb["seq"])) - loss) > 1e-4
    dec = "caption_decoder.lm_model.transformer."
    for name in (dec + "h.1.attn.c_attn.weight", dec + "h.0.attn.c_proj.weight", dec + "h.0.attn.c_proj.bias",
                 dec + "h.1.mlp.c_proj.bias", dec + "h.0.mlp.c_fc.weight", dec + "h.0.ln_2.weight", dec + "ln_f.weight",
                 dec + "wpe.weight", dec + "wte.weight", "caption_decoder.cross_attention.out_proj.weight",
                 "caption_decoder.cross_attention.out_proj.bias", "caption_decoder.cross_attention.in_proj_bias",
                 "caption_decoder.vision_projection.0.weight", "caption_decoder.attention_norm.weight",
                 "vision_encoder.projection.0.weight", "vision_encoder.projection.3.weight"):
        assert cos(model.store.g(name), sd[name].grad) >= 0.99, name
    for v in sd.values():
        v.grad = None

    # ---- Stage 1, dropout step counter = 0
    plan1 = DropoutPlan(p, seed + 1)
    st1 = ContrastiveStep(model.store, model.ws, model.vision_encoder.tower, model.vision_encoder.head,
                          model.text_encoder.engine, temperature=0.5, dropout=plan1)
    p1 = ContrastiveStep.prepare({"image": img, "caption_ids": T(g["ids_w"]), "caption_mask": T(g["mask_w"])}, model.device)
    for s in model.store.trainable_segments():
        s.grad.zero_()
    loss1 = float(st1.loss_and_grads(p1["image"], p1["ids"], p1["mask"]))
    drop1 = R.Dropper(seed + 1, p, step=0)
    ie = R.vision_encoder_forward(sd, img, arch.vit.heads, arch.vit.patch, drop1)["embeddings"]
    te = R.text_encoder_forward(sd, T(g["ids_w"]), T(g["mask_w"]), arch.gpt.heads, drop1)["embeddings"]
    ref1 = R.nt_xent(torch.nn.functional.normalize(ie, dim=-1), torch.nn.functional.normalize(te, dim=-1), 0.5)
    ref1.backward()
    assert abs(loss1 - float(ref1)) <= 5e-3
    for name in ("text_encoder.text_model.h.1.attn.c_attn.weight", "text_encoder.text_model.h.0.mlp.c_proj.weight",
                 "text_encoder.text_model.h.0.mlp.c_proj.bias", "text_encoder.text_model.wpe.weight",
                 "text_encoder.projection.0.weight", "vision_encoder.projection.0.weight"):
        assert cos(model.store.g(name), sd[name].grad) >= 0.99, name


def test_loss_curve_200_steps_tracks_fp32_oracle():
    """the loss curve over >= 200 optimizer steps stays within +-2 % of the CPU / fp32 run on identical
    batches with dropout off.  Tiny geometry, Stage-2 trainer path (2-forward PreferenceLoss), clip 1.0, AdamW with
    cosine warm-up; the oracle side is autograd over the restatement + its optimizer restatement."""
    from pgca_amd.arch import tiny_arch
    from pgca_amd.model import PreferenceGuidedCaptioningModel
    from pgca_amd.steps import DPOStep, FusedOptimizer
    arch = tiny_arch()
    model = PreferenceGuidedCaptioningModel(freeze_vision_backbone=True, arch=arch, seed=11, device="cuda:0")
    steps, lr, warm, nb, Bp, S = 200, 3e-4, 20, 4, 4, 16
    gen = torch.Generator().manual_seed(5)
    batches = []
    for _ in range(nb):
        lens = torch.randint(4, S + 1, (2 * Bp,), generator=gen)
        ids = torch.randint(0, arch.gpt.base_vocab, (2 * Bp, S), generator=gen)
        mask = (torch.arange(S)[None] < lens[:, None]).long()
        batches.append({"image": torch.randn(Bp, 3, arch.vit.image, arch.vit.image, generator=gen),
                        "preferred_ids": ids[:Bp], "rejected_ids": ids[Bp:],
                        "preferred_mask": mask[:Bp], "rejected_mask": mask[Bp:]})

    # ---- fp32 oracle run on the host
    sd = {k: v.detach().cpu().clone() for k, v in model.store.state_dict(aliases=False).items()}
    train = [n for seg in ("vision_head", "decoder") for n in model.store.segments[seg].index]
    for n in train:
        sd[n].requires_grad_(True)
    m = {n: torch.zeros_like(sd[n]) for n in train}
    v = {n: torch.zeros_like(sd[n]) for n in train}
    ref_curve = []
    for t in range(steps):
        b = batches[t % nb]
        lw = R.model_forward(sd, b["image"], b["preferred_ids"], b["preferred_mask"], "generation", arch.vit.heads,
                             arch.vit.patch, arch.gpt.heads)["logits"]
        ll = R.model_forward(sd, b["image"], b["rejected_ids"], b["rejected_mask"], "generation", arch.vit.heads,
                             arch.vit.patch, arch.gpt.heads)["logits"]
        loss = R.preference_loss(lw, ll, b["preferred_ids"], b["rejected_ids"], b["preferred_mask"],
                                 b["rejected_mask"], 0.1)
        grads = torch.autograd.grad(loss, [sd[n] for n in train], allow_unused=True)
        grads = [torch.zeros_like(sd[n]) if g is None else g for n, g in zip(train, grads)]
        norm = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads)))
        c = R.clip_coefficient(norm, 1.0)
        lr_t = R.cosine_warmup_lr(lr, t, warm, steps)
        with torch.no_grad():
            for n, g in zip(train, grads):
                R.adamw_step(sd[n], g * c, m[n], v[n], t + 1, lr_t)
        ref_curve.append(float(loss.detach()))

    # ---- HIP run
    step = DPOStep(model.store, model.ws, model.vision_encoder.tower, model.vision_encoder.head,
                   model.caption_decoder.engine, beta=0.1, reference_free=True)
    segs = [model.store.segments["vision_head"], model.store.segments["decoder"]]
    opt = FusedOptimizer(segs, lr=lr, weight_decay=0.01, max_grad_norm=1.0, warmup_steps=warm, total_steps=steps)
    prepared = [DPOStep.prepare(b, model.device) for b in batches]
    curve = []
    for t in range(steps):
        p = prepared[t % nb]
        opt.zero_grad()
        curve.append(step.loss_and_grads(p["image"], p["seq"]).clone())
        opt.step()
    curve = [float(x) for x in torch.stack(curve).cpu()]
    assert opt.state()["step"] == steps
    ref_t, hip_t = np.array(ref_curve), np.array(curve)
    assert ref_t[-nb:].mean() < 0.9 * ref_t[:nb].mean(), (ref_t[:nb], ref_t[-nb:])   # the run actually learns
    # +-2 % of the fp32 curve, with the bf16 loss tolerance (5e-3) as the absolute floor once the loss is small
    err = np.abs(hip_t - ref_t)
    worst = int(np.argmax(err - 0.02 * np.abs(ref_t)))
    assert (err <= 0.02 * np.abs(ref_t) + 5e-3).all(), (worst, hip_t[worst], ref_t[worst], ref_t[:3], ref_t[-3:])
    assert np.median(err / np.maximum(np.abs(ref_t), 1e-6)) <= 0.02
