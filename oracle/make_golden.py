"""Generate the golden fixtures under ``tests/golden/`` from the REFERENCE itself.

Runs only in the build container (needs ``/root/reference``; the GPU box has no
copy).  It imports the reference's ``models/components.py`` and ``models/model.py``
(the latter behind a stub ``peft`` module - LoRA is disabled in every shipped
config and ``peft`` is not installed), instantiates the reference classes without
their hub-downloading ``__init__`` and attaches HF modules built from *local*
configs, loads OUR seeded parameter set through ``load_state_dict`` (which also
proves the key names/shapes of ``pgca_amd.params`` match the reference's), then
records inputs and the reference's outputs/gradients as ``.npz`` data.

    python oracle/make_golden.py        # rewrites tests/golden/*.npz

Versions are recorded in each fixture (``meta``).
"""
from __future__ import annotations

import importlib.util
import json
import logging
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/src/preference_guided_image_captioning_alignment/models/"
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    peft = types.ModuleType("peft")
    peft.LoraConfig = type("LoraConfig", (), {"__init__": lambda self, **kw: None})
    peft.get_peft_model = lambda m, c: m
    sys.modules.setdefault("peft", peft)
    pkg = types.ModuleType("refpkg")
    pkg.__path__ = [REF]
    sys.modules["refpkg"] = pkg

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    comp = load("refpkg.components", REF + "components.py")
    model = load("refpkg.model", REF + "model.py")
    return comp, model


def meta():
    import transformers
    return json.dumps({"torch": torch.__version__, "transformers": transformers.__version__,
                       "reference": "A-SHOJAEI/preference-guided-image-captioning-alignment @ /root/reference"})


def build_reference_model(refmodel, arch, sd):
    """Reference classes, HF innards from local configs (no hub), our weights."""
    from transformers import CLIPConfig, CLIPModel, CLIPTextConfig, CLIPVisionConfig, GPT2Config, GPT2LMHeadModel, GPT2Model

    def bare(cls):
        obj = cls.__new__(cls)
        nn.Module.__init__(obj)
        obj.logger = logging.getLogger("golden")
        return obj

    def head(in_f, proj):
        return nn.Sequential(nn.Linear(in_f, proj), nn.ReLU(), nn.Dropout(0.1), nn.Linear(proj, proj),
                             nn.LayerNorm(proj))

    v, g = arch.vit, arch.gpt
    ve = bare(refmodel.VisionEncoder)
    ve.clip_model = CLIPModel(CLIPConfig(
        vision_config=CLIPVisionConfig(hidden_size=v.hidden, intermediate_size=v.mlp, num_hidden_layers=v.layers,
                                       num_attention_heads=v.heads, image_size=v.image, patch_size=v.patch).to_dict(),
        text_config=CLIPTextConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=1,
                                   num_attention_heads=2, vocab_size=128).to_dict(),
        projection_dim=64))
    ve.vision_model = ve.clip_model.vision_model
    ve.feature_dim = v.hidden
    ve.projection_dim = arch.proj_dim
    ve.projection = head(v.hidden, arch.proj_dim)

    gcfg = dict(n_embd=g.hidden, n_layer=g.layers, n_head=g.heads, n_positions=g.n_pos, vocab_size=g.base_vocab)
    te = bare(refmodel.TextEncoder)
    te.text_model = GPT2Model(GPT2Config(**gcfg))
    te.text_model.resize_token_embeddings(arch.text_vocab, mean_resizing=False)
    te.feature_dim = g.hidden
    te.projection = head(g.hidden, arch.proj_dim)

    cd = bare(refmodel.CaptionDecoder)
    cd.lm_model = GPT2LMHeadModel(GPT2Config(**gcfg))
    cd.lm_model.resize_token_embeddings(arch.dec_vocab, mean_resizing=False)
    cd.hidden_size = g.hidden
    cd.vision_projection = nn.Sequential(nn.Linear(arch.proj_dim, g.hidden), nn.Tanh(), nn.Dropout(0.1))
    cd.cross_attention = nn.MultiheadAttention(embed_dim=g.hidden, num_heads=arch.xattn_heads, dropout=0.1,
                                               batch_first=True)
    cd.attention_norm = nn.LayerNorm(g.hidden)

    m = bare(refmodel.PreferenceGuidedCaptioningModel)
    m.projection_dim = arch.proj_dim
    m.temperature = 0.5
    m.vision_encoder, m.text_encoder, m.caption_decoder = ve, te, cd

    res = m.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys[:5]
    bad = [k for k in res.missing_keys if "clip_model.text_model" not in k and "clip_model.visual_projection" not in k
           and "clip_model.text_projection" not in k and "clip_model.logit_scale" not in k
           and "position_ids" not in k and ".attn.bias" not in k and ".attn.masked_bias" not in k]
    assert not bad, bad[:10]
    return m.eval()


def np_(t):
    return np.array(t.detach().cpu().numpy(), copy=True)  # copy: tensors are updated in place later


def gen_nt_xent(comp, refmodel):
    out = {"meta": meta()}
    g = torch.Generator().manual_seed(7)
    for b in (2, 8, 64):
        for tau in (0.07, 0.5):
            img = torch.nn.functional.normalize(torch.randn(b, 32, generator=g), dim=-1).requires_grad_()
            txt = torch.nn.functional.normalize(torch.randn(b, 32, generator=g), dim=-1).requires_grad_()
            loss = refmodel.ContrastiveLoss(temperature=tau)(img, txt)
            loss.backward()
            k = f"b{b}_t{tau}"
            out[k + "_img"], out[k + "_txt"] = np_(img), np_(txt)
            out[k + "_loss"] = np_(loss)
            out[k + "_dimg"], out[k + "_dtxt"] = np_(img.grad), np_(txt.grad)
            out[k + "_loss_components"] = np_(comp.ContrastiveLoss(temperature=tau)(img.detach(), txt.detach()))
    np.savez_compressed(os.path.join(OUT, "nt_xent.npz"), **out)


def gen_logprob_dpo(comp, refmodel):
    out = {"meta": meta()}
    g = torch.Generator().manual_seed(11)
    b, s, v = 4, 12, 509
    lens_w, lens_l = [12, 7, 3, 9], [5, 12, 8, 2]
    ids_w = torch.randint(0, v, (b, s), generator=g)
    ids_l = torch.randint(0, v, (b, s), generator=g)
    mask_w = (torch.arange(s)[None] < torch.tensor(lens_w)[:, None]).long()
    mask_l = (torch.arange(s)[None] < torch.tensor(lens_l)[:, None]).long()
    lw = (torch.randn(b, s, v, generator=g) * 2).requires_grad_()
    ll = (torch.randn(b, s, v, generator=g) * 2).requires_grad_()
    pl = refmodel.PreferenceLoss(beta=0.1)
    loss = pl(lw, ll, ids_w, ids_l, mask_w, mask_l)
    loss.backward()
    out.update(ids_w=np_(ids_w), ids_l=np_(ids_l), mask_w=np_(mask_w), mask_l=np_(mask_l),
               logits_w=np_(lw).astype(np.float32), logits_l=np_(ll).astype(np.float32),
               pref_loss=np_(loss), dlogits_w=np_(lw.grad), dlogits_l=np_(ll.grad),
               seq_mean_w=np_(pl._compute_log_probs(lw, ids_w, mask_w)),
               seq_mean_l=np_(pl._compute_log_probs(ll, ids_l, mask_l)),
               seq_sum_w=np_(comp.compute_sequence_logprobs(lw, ids_w, mask_w)),
               seq_sum_l=np_(comp.compute_sequence_logprobs(ll, ids_l, mask_l)),
               seq_sum_w_nomask=np_(comp.compute_sequence_logprobs(lw, ids_w, None)),
               gather_index_w=np_(ids_w[:, 1:].contiguous()),
               token_lp_w=np_(torch.log_softmax(lw[:, :-1], -1).gather(-1, ids_w[:, 1:, None]).squeeze(-1)))
    pc, pr = torch.randn(8, generator=g) * 5 - 40, torch.randn(8, generator=g) * 5 - 42
    rc, rr = torch.randn(8, generator=g) * 5 - 41, torch.randn(8, generator=g) * 5 - 41
    out.update(dpo_pc=np_(pc), dpo_pr=np_(pr), dpo_rc=np_(rc), dpo_rr=np_(rr))
    for name, kw in (("std", {}), ("ls", {"label_smoothing": 0.1}), ("rf", {"reference_free": True})):
        pcg = pc.clone().requires_grad_()
        prg = pr.clone().requires_grad_()
        loss, metrics = comp.DPOPreferenceLoss(beta=0.1, **kw)(pcg, prg, rc, rr)
        loss.backward()
        out[f"dpo_{name}_loss"] = np_(loss)
        out[f"dpo_{name}_dpc"], out[f"dpo_{name}_dpr"] = np_(pcg.grad), np_(prg.grad)
        out[f"dpo_{name}_metrics"] = json.dumps(metrics)
    np.savez_compressed(os.path.join(OUT, "logprob_dpo.npz"), **out)


def gen_tiny_e2e(comp, refmodel):
    from pgca_amd.arch import tiny_arch
    from pgca_amd.params import ParamStore
    arch = tiny_arch()
    store = ParamStore(arch, "cpu", seed=1234, frozen=())
    sd = {k: v.clone() for k, v in store.state_dict().items()}
    m = build_reference_model(refmodel, arch, sd)
    g = torch.Generator().manual_seed(5)
    b, s = 4, 16
    images = torch.randn(b, 3, arch.vit.image, arch.vit.image, generator=g)
    lens_w, lens_l = [16, 9, 5, 12], [7, 16, 11, 3]
    ids_w = torch.randint(0, arch.gpt.base_vocab, (b, s), generator=g)
    ids_l = torch.randint(0, arch.gpt.base_vocab, (b, s), generator=g)
    mask_w = (torch.arange(s)[None] < torch.tensor(lens_w)[:, None]).long()
    mask_l = (torch.arange(s)[None] < torch.tensor(lens_l)[:, None]).long()
    pad = arch.gpt.base_vocab
    ids_w = torch.where(mask_w.bool(), ids_w, torch.full_like(ids_w, pad))
    ids_l = torch.where(mask_l.bool(), ids_l, torch.full_like(ids_l, pad))
    out = {"meta": meta(), "seed": np.int64(1234), "images": np_(images), "ids_w": np_(ids_w), "ids_l": np_(ids_l),
           "mask_w": np_(mask_w), "mask_l": np_(mask_l)}
    for seg in store.segments.values():
        out[f"chk_{seg.name}"] = np.array([float(seg.fp32.double().sum()), float(seg.fp32.double().abs().sum())])

    # Stage 1 (reference trainer.py:467-478) ------------------------------------------------
    m.zero_grad()
    o = m(images=images, caption_ids=ids_w, caption_mask=mask_w, mode="contrastive")
    loss1 = refmodel.ContrastiveLoss(temperature=0.5)(o["image_embeddings"], o["text_embeddings"])
    loss1.backward()
    named = dict(m.named_parameters())
    out.update(s1_image_embeddings=np_(o["image_embeddings"]), s1_text_embeddings=np_(o["text_embeddings"]),
               s1_text_features=np_(o["text_features"]), s1_vision_features=np_(o["vision_features"]),
               s1_loss=np_(loss1))
    for k in ("text_encoder.text_model.h.1.attn.c_attn.weight", "text_encoder.text_model.h.0.mlp.c_fc.bias",
              "text_encoder.text_model.wpe.weight", "text_encoder.text_model.ln_f.weight",
              "text_encoder.projection.0.weight", "vision_encoder.projection.3.weight",
              "vision_encoder.projection.4.bias", "text_encoder.text_model.h.0.ln_1.weight"):
        out["s1_grad::" + k] = np_(named[k].grad)
    out["s1_grad_wte_rows"] = np_(named["text_encoder.text_model.wte.weight"].grad[:64])

    # Stage 2, trainer path: 2 forwards + PreferenceLoss (trainer.py:578-603) ---------------
    m.zero_grad()
    ow = m(images=images, caption_ids=ids_w, caption_mask=mask_w, labels=ids_w, mode="generation")
    ol = m(images=images, caption_ids=ids_l, caption_mask=mask_l, labels=ids_l, mode="generation")
    loss2 = refmodel.PreferenceLoss(beta=0.1)(ow["logits"], ol["logits"], ids_w, ids_l, mask_w, mask_l)
    loss2.backward()
    out.update(s2_logits_w=np_(ow["logits"]), s2_logits_l=np_(ol["logits"]), s2_pref_loss=np_(loss2),
               s2_seq_sum_w=np_(comp.compute_sequence_logprobs(ow["logits"], ids_w, mask_w)),
               s2_seq_sum_l=np_(comp.compute_sequence_logprobs(ol["logits"], ids_l, mask_l)))
    dec = "caption_decoder.lm_model.transformer."
    for k in (dec + "h.1.attn.c_attn.weight", dec + "h.0.attn.c_proj.weight", dec + "h.0.mlp.c_fc.weight",
              dec + "h.1.mlp.c_proj.bias", dec + "ln_f.bias", dec + "wpe.weight", dec + "h.0.ln_2.weight",
              "caption_decoder.cross_attention.in_proj_weight", "caption_decoder.cross_attention.in_proj_bias",
              "caption_decoder.cross_attention.out_proj.weight", "caption_decoder.attention_norm.weight",
              "caption_decoder.vision_projection.0.weight", "vision_encoder.projection.0.weight",
              "vision_encoder.projection.4.weight"):
        out["s2_grad::" + k] = np_(named[k].grad)
    out["s2_grad_wte"] = np_(named[dec + "wte.weight"].grad)
    unused = [k for k, p in named.items() if k.startswith("text_encoder.") and p.grad is not None
              and float(p.grad.abs().max()) > 0]
    out["s2_text_tower_params_with_grad"] = np.int64(len(unused))

    # Stage 2, 4-forward DPO (components.py:192-249,321-362) with a frozen snapshot reference
    m.zero_grad()
    with torch.no_grad():
        # reference policy == snapshot at Stage-2 start, perturbed so the terms do not cancel
        ref = build_reference_model(refmodel, arch, sd)
        for k, p in ref.named_parameters():
            if k.startswith("caption_decoder.lm_model.transformer.h.") and k.endswith("mlp.c_proj.weight"):
                p.mul_(0.9)
        rw = comp.compute_sequence_logprobs(ref(images=images, caption_ids=ids_w, caption_mask=mask_w,
                                                mode="generation")["logits"], ids_w, mask_w)
        rl = comp.compute_sequence_logprobs(ref(images=images, caption_ids=ids_l, caption_mask=mask_l,
                                                mode="generation")["logits"], ids_l, mask_l)
    ow = m(images=images, caption_ids=ids_w, caption_mask=mask_w, mode="generation")
    ol = m(images=images, caption_ids=ids_l, caption_mask=mask_l, mode="generation")
    pw = comp.compute_sequence_logprobs(ow["logits"], ids_w, mask_w)
    plg = comp.compute_sequence_logprobs(ol["logits"], ids_l, mask_l)
    loss4, metrics = comp.DPOPreferenceLoss(beta=0.1)(pw, plg, rw, rl)
    loss4.backward()
    out.update(s2_dpo_loss=np_(loss4), s2_dpo_metrics=json.dumps(metrics), s2_ref_w=np_(rw), s2_ref_l=np_(rl),
               s2_pol_w=np_(pw), s2_pol_l=np_(plg))
    for k in (dec + "h.1.attn.c_attn.weight", dec + "h.0.mlp.c_fc.weight", dec + "ln_f.weight",
              "caption_decoder.vision_projection.0.weight", "vision_encoder.projection.0.weight"):
        out["s2dpo_grad::" + k] = np_(named[k].grad)
    out["s2dpo_grad_wte"] = np_(named[dec + "wte.weight"].grad)
    np.savez_compressed(os.path.join(OUT, "tiny_e2e.npz"), **out)


def gen_tiny_vit_grads(comp, refmodel):
    """Gradients of the CLIP tower's own parameters on the tiny end-to-end case (same weights and inputs as
    ``tiny_e2e.npz``): the reference only freezes the tower on request (model.py:150-164), so with the constructor
    default every ``vision_model`` parameter trains in both stages."""
    from pgca_amd.arch import tiny_arch
    from pgca_amd.params import ParamStore
    arch = tiny_arch()
    store = ParamStore(arch, "cpu", seed=1234, frozen=())
    sd = {k: v.clone() for k, v in store.state_dict().items()}
    m = build_reference_model(refmodel, arch, sd)
    e2e = np.load(os.path.join(OUT, "tiny_e2e.npz"), allow_pickle=False)
    images = torch.from_numpy(e2e["images"])
    ids_w, ids_l = torch.from_numpy(e2e["ids_w"]), torch.from_numpy(e2e["ids_l"])
    mask_w, mask_l = torch.from_numpy(e2e["mask_w"]), torch.from_numpy(e2e["mask_l"])
    named = dict(m.named_parameters(remove_duplicate=False))   # the tower is registered twice (clip_model.vision_model)
    vm = "vision_encoder.vision_model."
    keys = [vm + k for k in (
        "embeddings.class_embedding", "embeddings.patch_embedding.weight", "embeddings.position_embedding.weight",
        "pre_layrnorm.weight", "pre_layrnorm.bias", "post_layernorm.weight", "post_layernorm.bias",
        "encoder.layers.0.self_attn.q_proj.weight", "encoder.layers.0.self_attn.k_proj.bias",
        "encoder.layers.0.self_attn.v_proj.weight", "encoder.layers.1.self_attn.out_proj.weight",
        "encoder.layers.1.self_attn.out_proj.bias", "encoder.layers.0.layer_norm1.weight",
        "encoder.layers.1.layer_norm2.bias", "encoder.layers.0.mlp.fc1.weight", "encoder.layers.1.mlp.fc1.bias",
        "encoder.layers.0.mlp.fc2.weight", "encoder.layers.1.mlp.fc2.bias")]
    out = {"meta": meta(), "seed": np.int64(1234)}

    m.zero_grad()
    o = m(images=images, caption_ids=ids_w, caption_mask=mask_w, mode="contrastive")
    loss1 = refmodel.ContrastiveLoss(temperature=0.5)(o["image_embeddings"], o["text_embeddings"])
    loss1.backward()
    assert abs(float(loss1) - float(e2e["s1_loss"])) < 1e-6
    out["s1_loss"] = np_(loss1)
    def put(tag):
        for k in keys:
            gr = named[k].grad
            if k.endswith("patch_embedding.weight"):   # 1.5 MB whole: keep 8 output channels + two checksums
                out[tag + "_grad_patch_rows"] = np_(gr.reshape(gr.shape[0], -1)[:8])
                out[tag + "_grad_patch_chk"] = np.array([float(gr.double().sum()), float(gr.double().abs().sum())])
            else:
                out[tag + "_grad::" + k] = np_(gr)

    put("s1")

    m.zero_grad()
    ow = m(images=images, caption_ids=ids_w, caption_mask=mask_w, labels=ids_w, mode="generation")
    ol = m(images=images, caption_ids=ids_l, caption_mask=mask_l, labels=ids_l, mode="generation")
    loss2 = refmodel.PreferenceLoss(beta=0.1)(ow["logits"], ol["logits"], ids_w, ids_l, mask_w, mask_l)
    loss2.backward()
    assert abs(float(loss2) - float(e2e["s2_pref_loss"])) < 1e-6
    out["s2_pref_loss"] = np_(loss2)
    put("s2")
    np.savez_compressed(os.path.join(OUT, "tiny_vit_grads.npz"), **out)


def gen_optimizer():
    """3 clipped AdamW + cosine-warm-up steps exactly as reference trainer.py:275-289,511-520 wires them."""
    from transformers import get_cosine_schedule_with_warmup
    g = torch.Generator().manual_seed(3)
    ps = [nn.Parameter(torch.randn(33, 17, generator=g)), nn.Parameter(torch.randn(129, generator=g))]
    opt = torch.optim.AdamW(ps, lr=5e-5, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-8)
    sch = get_cosine_schedule_with_warmup(opt, num_warmup_steps=2, num_training_steps=10)
    out = {"meta": meta(), "p0_init": np_(ps[0]), "p1_init": np_(ps[1])}
    for step in range(4):
        gs = [torch.randn(p.shape, generator=g) * (3.0 if step % 2 == 0 else 0.01) for p in ps]
        for p, gr in zip(ps, gs):
            p.grad = gr.clone()
        out[f"g0_{step}"], out[f"g1_{step}"] = np_(gs[0]), np_(gs[1])
        out[f"lr_{step}"] = np.float64(opt.param_groups[0]["lr"])
        norm = torch.nn.utils.clip_grad_norm_(ps, 1.0)
        out[f"norm_{step}"] = np_(norm)
        opt.step()
        sch.step()
        opt.zero_grad()
        out[f"p0_{step}"], out[f"p1_{step}"] = np_(ps[0]), np_(ps[1])
    np.savez_compressed(os.path.join(OUT, "optimizer.npz"), **out)


def gen_generation(comp, refmodel):
    """The reference's own ``CaptionDecoder.generate`` (model.py:621-678 -> HF ``generate(inputs_embeds=...)`` with its
    K/V cache) on the tiny geometry with OUR weights: greedy and deterministic beam search (``do_sample=False``; the
    sampling modes draw from torch's RNG stream and cannot be replayed), with and without an EOS that actually occurs."""
    from pgca_amd.arch import tiny_arch
    from pgca_amd.params import ParamStore
    arch = tiny_arch()
    store = ParamStore(arch, "cpu", seed=17, frozen=())
    sd = {k: v.clone() for k, v in store.state_dict().items()}
    m = build_reference_model(refmodel, arch, sd)
    g = torch.Generator().manual_seed(2)
    img = torch.randn(5, 3, arch.vit.image, arch.vit.image, generator=g)
    pad, eos = arch.gpt.base_vocab, arch.gpt.base_vocab + 2
    out = {"meta": meta(), "seed": np.int64(17), "images": np_(img), "pad": np.int64(pad), "eos": np.int64(eos)}
    with torch.no_grad():
        emb = m.vision_encoder(img)["embeddings"]
        out["embeddings"] = np_(emb)
        cd = m.caption_decoder
        cases = {"greedy": dict(max_length=12, num_beams=1, do_sample=False, repetition_penalty=1.1),
                 "greedy_norep": dict(max_length=9, num_beams=1, do_sample=False, repetition_penalty=1.0),
                 "beam4": dict(max_length=8, num_beams=4, do_sample=False, repetition_penalty=1.0),
                 "beam3_rep": dict(max_length=10, num_beams=3, do_sample=False, repetition_penalty=1.2)}
        for name, kw in cases.items():
            ids = cd.generate(emb, pad_token_id=pad, eos_token_id=eos, **kw)
            out[name + "_ids"] = np_(ids)
            out[name + "_kw"] = json.dumps(kw)
        # an EOS that occurs: the token greedy produces at step 2 for image 0 ends that caption
        first = int(out["greedy_norep_ids"][0, 2])
        out["eos_case_eos"] = np.int64(first)
        out["greedy_eos_ids"] = np_(cd.generate(emb, max_length=9, num_beams=1, do_sample=False, repetition_penalty=1.0,
                                                pad_token_id=pad, eos_token_id=first))
        out["beam4_eos_ids"] = np_(cd.generate(emb, max_length=8, num_beams=4, do_sample=False, repetition_penalty=1.0,
                                               pad_token_id=pad, eos_token_id=first))
    np.savez_compressed(os.path.join(OUT, "generation.npz"), **out)
    for k in out:
        if k.endswith("_ids"):
            print(k, out[k].shape, out[k][0].tolist())


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    gen_optimizer()  # before the stub ``peft`` exists: transformers.optimization probes for it
    comp, refmodel = import_reference()
    gen_nt_xent(comp, refmodel)
    gen_logprob_dpo(comp, refmodel)
    gen_tiny_e2e(comp, refmodel)
    gen_tiny_vit_grads(comp, refmodel)
    gen_generation(comp, refmodel)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
