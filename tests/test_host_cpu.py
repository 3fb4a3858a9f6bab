"""Host-side logic that needs no GPU: config surface, parameter store, batch index preparation."""
import os

import numpy as np
import pytest
import torch

from oracle import restatement as R
from pgca_amd import REPO_ROOT
from pgca_amd.arch import make_arch, tiny_arch
from pgca_amd.config import Config
from pgca_amd.params import ParamStore, model_specs


def test_config_reads_own_and_reference_style_yaml(tmp_path):
    c = Config(os.path.join(REPO_ROOT, "configs", "default.yaml"))
    assert c.get("model.text_model") == "gpt2-medium"
    assert c.get_stage2_config()["dpo_beta"] == 0.1
    assert c.get("mi355x.dpo.reference_free") is True
    assert c.get("does.not.exist", 7) == 7
    c.set("training.stage1.batch_size", 16)
    assert c.get("training.stage1.batch_size") == 16
    # a reference-style file without the mi355x section: defaults reproduce the reference's behaviour
    p = tmp_path / "ref.yaml"
    p.write_text("model:\n  temperature: 0.5\ntraining:\n  stage1: {learning_rate: 5.0e-5, num_epochs: 1}\n"
                 "  stage2: {learning_rate: 1.0e-5, num_epochs: 1, dpo_beta: 0.1}\n")
    r = Config(str(p))
    assert r.get("mi355x.dpo.reference_free") is True and r.get("mi355x.stage1.global_negatives") is False
    with pytest.raises(ValueError, match="Missing required"):
        bad = tmp_path / "bad.yaml"
        bad.write_text("model: {}\n")
        Config(str(bad))


def test_config_env_override(tmp_path, monkeypatch):
    monkeypatch.setenv("PGCA_CFG_TRAINING__STAGE2__DPO_BETA", "0.25")
    c = Config(os.path.join(REPO_ROOT, "configs", "default.yaml"))
    assert c.get("training.stage2.dpo_beta") == 0.25


def test_reference_environment_overrides(monkeypatch):
    """The reference's own variables (utils/config.py:91-136) with its value conversion (:151-182)."""
    for k, v in (("CAPTION_ALIGNMENT_BATCH_SIZE", "16"), ("CAPTION_ALIGNMENT_LEARNING_RATE", "3e-5"),
                 ("OUTPUT_DIR", "/tmp/run7"), ("CAPTION_ALIGNMENT_PIN_MEMORY", "off"),
                 ("CAPTION_ALIGNMENT_MIXED_PRECISION", "bf16"), ("CAPTION_ALIGNMENT_VISION_MODEL", "openai/clip-vit-large-patch14"),
                 ("ULTRAFEEDBACK_PATH", "")):
        monkeypatch.setenv(k, v)
    c = Config(os.path.join(REPO_ROOT, "configs", "default.yaml"))
    assert c.get("training.stage1.batch_size") == 16 and c.get("training.stage1.learning_rate") == 3e-5
    assert c.get("paths.output_dir") == "/tmp/run7" and c.get("data.pin_memory") is False
    assert c.get("hardware.mixed_precision") == "bf16"
    assert c.get("model.vision_model") == "openai/clip-vit-large-patch14"
    base = Config.__new__(Config)
    assert Config._convert_env_value("yes") is True and Config._convert_env_value("7") == 7
    assert Config._convert_env_value("0.5") == 0.5 and Config._convert_env_value("cuda:1") == "cuda:1"


def test_parameter_counts_match_reference_readme():
    """README.md:140,179 / SURVEY 6: 867 M total parameters, reproduced as 867 100 417 with the reference's
    modules.  Our store leaves out what the hot path never touches: the CLIP *text* tower and its projections
    (63 165 952 + 393 216 + 262 144 + 1 = 63 821 313 parameters) and registers the ViT once."""
    arch = make_arch("openai/clip-vit-base-patch32", "gpt2-medium", 512)
    n = sum(sp.numel for specs in model_specs(arch).values() for sp in specs)
    clip_text_side = 867_100_417 - n
    assert n == 803_279_104 and clip_text_side == 63_821_313
    seg_sizes = {k: sum(sp.numel for sp in v) for k, v in model_specs(arch).items()}
    assert seg_sizes["vit"] == 87_456_000
    assert seg_sizes["text_tower"] == 354_825_216 and seg_sizes["decoder"] - seg_sizes["text_tower"] > 4_000_000


def test_store_layout_alignment_and_aliases():
    st = ParamStore(tiny_arch(), "cpu", seed=0)
    for seg in st.segments.values():
        for name, (off, shape) in seg.index.items():
            assert off % 64 == 0, name
        assert seg.numel % 64 == 0
    sd = st.state_dict(aliases=True)
    assert sd["caption_decoder.lm_model.lm_head.weight"].data_ptr() == \
        sd["caption_decoder.lm_model.transformer.wte.weight"].data_ptr()
    k = "vision_encoder.vision_model.post_layernorm.weight"
    assert sd["vision_encoder.clip_model.vision_model.post_layernorm.weight"].data_ptr() == sd[k].data_ptr()
    # q,k,v of a CLIP layer are contiguous (one [3H,H] GEMM)
    seg = st.segments["vit"]
    h = st.arch.vit.hidden
    p = "vision_encoder.vision_model.encoder.layers.0.self_attn."
    assert seg.index[p + "k_proj.weight"][0] == seg.index[p + "q_proj.weight"][0] + h * h
    # decoder vocabulary is padded with zero rows to a multiple of 128 for the LM-head dgrad GEMM
    dec = st.segments["decoder"]
    pad = dec.padded(dec.fp32, "caption_decoder.lm_model.transformer.wte.weight")
    assert pad.shape[0] % 128 == 0 and float(pad[st.arch.dec_vocab:].abs().sum()) == 0.0


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO_ROOT, "preference-guided-image-captioning-alignment_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            assert "import oracle" not in src and "from oracle" not in src, f


def test_tokenised_caption_cache_against_a_real_hf_tokenizer(tmp_path):
    """N2: the tokenised-caption cache returns what the reference's ``TextProcessor.encode_caption``
    (data/preprocessing.py:206-238) returns - the HF tokenizer called with the reference's arguments - for a tokenizer
    built locally (no hub files: WordLevel vocab + the reference's added [PAD]/[BOS]/[EOS] special tokens), encodes every
    distinct caption once, and survives a save / load."""
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast

    from pgca_amd.input import TokenisedCaptionCache
    words = "a cat dog sits on the mat red blue big small runs fast [UNK]".split()
    tk = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="[UNK]"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    hf = PreTrainedTokenizerFast(tokenizer_object=tk, unk_token="[UNK]")
    hf.add_special_tokens({"pad_token": "[PAD]", "bos_token": "[BOS]", "eos_token": "[EOS]"})   # preprocessing.py:185-199
    S = 8
    cache = TokenisedCaptionCache(hf, max_length=S, capacity=2, pin=False)
    caps = ["a cat sits on the mat", "the big red dog runs fast on the small blue mat", "a zebra", "a cat sits on the mat"]
    for c in caps:
        want = hf(c, max_length=S, padding="max_length", truncation=True, add_special_tokens=True, return_tensors="pt",
                  return_attention_mask=True)
        got = cache.encode_caption(c)
        assert torch.equal(got["input_ids"], want["input_ids"].squeeze(0))
        assert torch.equal(got["attention_mask"], want["attention_mask"].squeeze(0))
        assert got["input_ids"].shape == (S,) and got["input_ids"].dtype == torch.int64
    assert len(cache) == 3 and cache.misses == 3 and cache.hits == 1          # the repeated caption was not re-tokenised
    b = cache.encode_batch(caps)
    assert b["input_ids"].shape == (4, S) and torch.equal(b["input_ids"][0], b["input_ids"][3])
    assert int(b["attention_mask"][1].sum()) == S                              # truncated to max_length
    # right padding with the pad id and mask 0: the layout the packed rows rely on
    assert bool((b["input_ids"][2][b["attention_mask"][2] == 0] == hf.pad_token_id).all())
    path = str(tmp_path / "captions.npz")
    cache.save(path)
    fresh = TokenisedCaptionCache(lambda *a, **k: (_ for _ in ()).throw(AssertionError("tokenizer called")), max_length=S,
                                  pin=False)
    fresh.load(path)
    assert torch.equal(fresh.encode_batch(caps)["input_ids"], b["input_ids"]) and fresh.misses == 0
    with pytest.raises(ValueError, match="Failed to encode caption"):
        fresh.encode_caption("never seen")
