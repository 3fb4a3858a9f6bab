"""The CLIP tower left TRAINABLE (reference model.py:150-164 freezes it only on request; the constructor default is
``freeze_vision_backbone=False``): HIP backward through the whole vision transformer.

* tiny geometry: against the REFERENCE's own gradients of the tower's parameters (tests/golden/tiny_vit_grads.npz, made by
  oracle/make_golden.py:gen_tiny_vit_grads from the imported reference; same weights and batch as tiny_e2e.npz).
* ViT-B/32 width (768 / 12 heads / 50 tokens, 2 layers): against the oracle restatement run on the host.

Tolerances as everywhere (SURVEY 8d, bf16 MFMA operands / f32 accumulate): loss |d| <= 5e-3, gradient cosine >= 0.99.
"""
import numpy as np
import pytest
import torch

from oracle import restatement as R

pytestmark = pytest.mark.gpu
T = torch.from_numpy
VM = "vision_encoder.vision_model."


def cos(a, b):
    a, b = a.double().flatten().cpu(), torch.as_tensor(b).double().flatten().cpu()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


@pytest.fixture(scope="module")
def tiny(golden):
    from pgca_amd.arch import tiny_arch
    from pgca_amd.model import PreferenceGuidedCaptioningModel
    g, gv = golden("tiny_e2e"), golden("tiny_vit_grads")
    model = PreferenceGuidedCaptioningModel(arch=tiny_arch(), seed=int(g["seed"]), device="cuda:0")   # default: trainable tower
    assert model.vision_encoder.tower.trainable and not model.vision_encoder.freeze_backbone
    assert model.store.segments["vit"].grad is not None
    return g, gv, model


def check_tower_grads(model, gv, tag):
    n = 0
    qb = np.abs(gv[f"{tag}_grad::{VM}encoder.layers.1.self_attn.out_proj.bias"]).max()
    for k in gv.files:
        if not k.startswith(tag + "_grad::"):
            continue
        name = k[len(tag) + 7:]
        got, want = model.store.g(name), gv[k]
        if np.abs(want).max() < 1e-9:
            # k_proj.bias: analytically zero (softmax is invariant to a shift of every key score); on the HIP path it is
            # the rounding residue of dS rows that sum to zero
            assert float(got.abs().max()) <= 2e-2 * qb, name
        else:
            c = cos(got, want)
            assert c >= 0.99, f"{tag} {name}: cosine {c}"
            r = float(got.double().norm().cpu()) / float(np.linalg.norm(want.astype(np.float64)))
            assert 0.9 <= r <= 1.1, f"{tag} {name}: norm ratio {r}"
        n += 1
    assert n == 17
    pg = model.store.g(VM + "embeddings.patch_embedding.weight")
    assert cos(pg.reshape(pg.shape[0], -1)[:8], gv[tag + "_grad_patch_rows"]) >= 0.99
    chk = gv[tag + "_grad_patch_chk"]
    assert abs(float(pg.double().abs().sum()) - chk[1]) <= 5e-2 * chk[1]


def test_tiny_stage1_tower_gradients_match_reference(tiny):
    from pgca_amd.steps import ContrastiveStep
    g, gv, model = tiny
    step = ContrastiveStep(model.store, model.ws, model.vision_encoder.tower, model.vision_encoder.head,
                           model.text_encoder.engine, temperature=0.5)
    p = ContrastiveStep.prepare({"image": T(g["images"]), "caption_ids": T(g["ids_w"]), "caption_mask": T(g["mask_w"])},
                                model.device)
    for s in model.store.trainable_segments():
        s.grad.zero_()
    loss = float(step.loss_and_grads(p["image"], p["ids"], p["mask"]))
    assert abs(loss - float(gv["s1_loss"])) <= 5e-3
    check_tower_grads(model, gv, "s1")
    for k in g.files:          # the rest of the model is unaffected by the tower being trainable
        if k.startswith("s1_grad::"):
            assert cos(model.store.g(k[len("s1_grad::"):]), g[k]) >= 0.99, k
    # gradients ACCUMULATE over micro-batches like every other parameter's
    before = model.store.g(VM + "encoder.layers.0.mlp.fc1.weight").clone()
    step.loss_and_grads(p["image"], p["ids"], p["mask"])
    after = model.store.g(VM + "encoder.layers.0.mlp.fc1.weight")
    assert cos(after, before) >= 0.9999 and abs(float(after.norm() / before.norm()) - 2.0) <= 1e-2


def test_tiny_stage2_tower_gradients_match_reference(tiny):
    from pgca_amd.steps import DPOStep
    g, gv, model = tiny
    step = DPOStep(model.store, model.ws, model.vision_encoder.tower, model.vision_encoder.head,
                   model.caption_decoder.engine, beta=0.1, reference_free=True)
    p = DPOStep.prepare({"image": T(g["images"]), "preferred_ids": T(g["ids_w"]), "rejected_ids": T(g["ids_l"]),
                         "preferred_mask": T(g["mask_w"]), "rejected_mask": T(g["mask_l"])}, model.device)
    for s in model.store.trainable_segments():
        s.grad.zero_()
    loss = float(step.loss_and_grads(p["image"], p["seq"]))
    assert abs(loss - float(gv["s2_pref_loss"])) <= 5e-3
    check_tower_grads(model, gv, "s2")
    # evaluation forwards keep nothing and still agree
    assert abs(float(step.loss_only(p["image"], p["seq"])) - loss) <= 1e-5


def test_tower_is_updated_by_the_optimizer_and_frozen_on_request(tmp_path):
    import os
    from torch.utils.data import DataLoader
    from pgca_amd.arch import tiny_arch
    from pgca_amd.config import Config
    from pgca_amd.model import PreferenceGuidedCaptioningModel
    from pgca_amd.trainer import PreferenceGuidedTrainer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gen = torch.Generator().manual_seed(3)
    arch = tiny_arch()
    data = [{"image": torch.randn(3, arch.vit.image, arch.vit.image, generator=gen),
             "caption_ids": torch.randint(0, arch.gpt.base_vocab, (16,), generator=gen),
             "caption_mask": torch.ones(16, dtype=torch.long)} for _ in range(8)]
    for frozen in (False, True):
        cfg = Config(os.path.join(root, "configs", "default.yaml"))
        cfg.set("paths.output_dir", str(tmp_path / ("frozen" if frozen else "trainable")))
        cfg.set("training.stage1.num_epochs", 1)
        cfg.set("training.stage1.warmup_steps", 0)
        cfg.set("training.stage1.learning_rate", 1e-3)
        cfg.set("training.stage1.gradient_accumulation_steps", 1)
        cfg.set("mi355x.gpt2_pdrop", 0.0)
        model = PreferenceGuidedCaptioningModel(freeze_vision_backbone=frozen, arch=arch, dropout=0.0, seed=5,
                                                device="cuda:0")
        vit = model.store.segments["vit"]
        before = vit.fp32.clone()
        loader = DataLoader(data, batch_size=4, shuffle=False, drop_last=True)
        tr = PreferenceGuidedTrainer(model, cfg, loader, loader)
        tr.train_stage1()
        if frozen:
            assert torch.equal(vit.fp32, before) and vit.grad is None
            continue
        assert torch.equal(vit.bf16, vit.fp32.bfloat16())      # mirror refreshed by the fused AdamW pass
        for name in (VM + "embeddings.class_embedding", VM + "embeddings.patch_embedding.weight",
                     VM + "embeddings.position_embedding.weight", VM + "pre_layrnorm.weight",
                     VM + "encoder.layers.1.self_attn.q_proj.weight", VM + "encoder.layers.0.mlp.fc2.bias",
                     VM + "post_layernorm.bias"):
            off = vit.index[name][0]
            w = vit.w(name)
            assert not torch.equal(w.flatten(), before[off:off + w.numel()]), name


@pytest.mark.parametrize("vision", ["openai/clip-vit-base-patch32", "openai/clip-vit-large-patch14"])
def test_real_width_tower_backward_against_oracle(vision):
    """ViT-B/32 (768 wide, 12 heads, 50 tokens, patch K = 3072) and ViT-L/14 (1024 wide, 16 heads, 257 tokens over three
    128-query blocks, patch K = 588 padded to 640), 2 layers each, Stage-1 step with local negatives.  The value / output
    projections are scaled x4 as in test_stage1_config_geometry_against_oracle (random-init class tokens barely depend on
    the image otherwise, and the contrastive gradient is what survives the cancellation)."""
    from pgca_amd.arch import make_arch, with_layers
    from pgca_amd.model import PreferenceGuidedCaptioningModel
    from pgca_amd.steps import ContrastiveStep
    arch = with_layers(make_arch(vision, "gpt2", 512), 2, 1)
    model = PreferenceGuidedCaptioningModel(arch=arch, seed=29, device="cuda:0")
    vit = model.store.segments["vit"]
    for name in vit.index:
        if name.endswith(("self_attn.v_proj.weight", "self_attn.out_proj.weight")):
            vit.w(name).mul_(4.0)
    vit.ensure_bf16()
    gen = torch.Generator().manual_seed(78)
    B, S = 4, 32
    img = torch.randn(B, 3, arch.vit.image, arch.vit.image, generator=gen)
    ids = torch.randint(0, 50257, (B, S), generator=gen)
    mask = (torch.arange(S)[None] < torch.tensor([32, 9, 20, 5])[:, None]).long()
    ids = torch.where(mask.bool(), ids, torch.full_like(ids, 50257))
    step = ContrastiveStep(model.store, model.ws, model.vision_encoder.tower, model.vision_encoder.head,
                           model.text_encoder.engine, temperature=0.5)
    p = ContrastiveStep.prepare({"image": img, "caption_ids": ids, "caption_mask": mask}, model.device)
    for s in model.store.trainable_segments():
        s.grad.zero_()
    loss = float(step.loss_and_grads(p["image"], p["ids"], p["mask"]))
    sd = {k: v.detach().cpu().clone().requires_grad_(k.startswith("vision_encoder"))
          for k, v in model.store.state_dict(aliases=False).items()}
    ie = R.vision_encoder_forward(sd, img, arch.vit.heads, arch.vit.patch)["embeddings"]
    with torch.no_grad():
        te = R.text_encoder_forward(sd, ids, mask, arch.gpt.heads)["embeddings"]
    ref = R.nt_xent(torch.nn.functional.normalize(ie, dim=-1), torch.nn.functional.normalize(te, dim=-1), 0.5)
    ref.backward()
    assert abs(loss - float(ref)) <= 5e-3, (loss, float(ref))
    for name in ("embeddings.class_embedding", "embeddings.patch_embedding.weight", "embeddings.position_embedding.weight",
                 "pre_layrnorm.weight", "pre_layrnorm.bias", "post_layernorm.weight", "post_layernorm.bias",
                 "encoder.layers.0.self_attn.q_proj.weight", "encoder.layers.0.self_attn.k_proj.weight",
                 "encoder.layers.0.self_attn.v_proj.weight", "encoder.layers.0.self_attn.v_proj.bias",
                 "encoder.layers.1.self_attn.out_proj.weight", "encoder.layers.1.self_attn.out_proj.bias",
                 "encoder.layers.0.layer_norm1.weight", "encoder.layers.1.layer_norm2.bias",
                 "encoder.layers.0.mlp.fc1.weight", "encoder.layers.1.mlp.fc1.bias", "encoder.layers.0.mlp.fc2.weight",
                 "encoder.layers.1.mlp.fc2.bias"):
        c = cos(model.store.g(VM + name), sd[VM + name].grad)
        assert c >= 0.99, f"{name}: cosine {c}"
    assert cos(model.store.g("vision_encoder.projection.0.weight"), sd["vision_encoder.projection.0.weight"].grad) >= 0.99
