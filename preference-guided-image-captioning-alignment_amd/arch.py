"""Tower geometries for the hot path.

The reference builds its towers with ``from_pretrained(<hub name>)``
(reference ``models/model.py:126,311-312,505-506``).  There is no hub access on
the build or GPU boxes, so geometries are tabulated here by the same names the
reference configs use (``configs/default.yaml:18-19``); weights are seeded
random-init or loaded from a checkpoint with the reference's key names.
"""
from dataclasses import dataclass, field, replace
from typing import Dict


@dataclass(frozen=True)
class VitArch:
    """CLIP vision transformer (HF ``CLIPVisionConfig`` fields that matter)."""
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    mlp: int = 3072
    patch: int = 32
    image: int = 224
    eps: float = 1e-5

    @property
    def grid(self) -> int:
        return self.image // self.patch

    @property
    def tokens(self) -> int:
        return self.grid * self.grid + 1

    @property
    def patch_dim(self) -> int:
        return 3 * self.patch * self.patch


@dataclass(frozen=True)
class GptArch:
    """GPT-2 family (HF ``GPT2Config``): pre-LN, gelu_new, Conv1D [in,out] weights."""
    hidden: int = 1024
    layers: int = 24
    heads: int = 16
    n_pos: int = 1024
    base_vocab: int = 50257
    eps: float = 1e-5

    @property
    def inner(self) -> int:
        return 4 * self.hidden

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads


@dataclass(frozen=True)
class ModelArch:
    vit: VitArch = field(default_factory=VitArch)
    gpt: GptArch = field(default_factory=GptArch)
    proj_dim: int = 512
    # reference model.py:314-324 adds [PAD],[SEP] to the text tower's tokenizer (+2)
    # and model.py:509-511 adds [PAD],[BOS],[EOS] to the decoder's (+3)
    text_vocab_extra: int = 2
    dec_vocab_extra: int = 3
    xattn_heads: int = 8  # reference model.py:528-533

    @property
    def text_vocab(self) -> int:
        return self.gpt.base_vocab + self.text_vocab_extra

    @property
    def dec_vocab(self) -> int:
        return self.gpt.base_vocab + self.dec_vocab_extra


VIT_ZOO: Dict[str, VitArch] = {
    "openai/clip-vit-base-patch32": VitArch(768, 12, 12, 3072, 32, 224),
    "openai/clip-vit-base-patch16": VitArch(768, 12, 12, 3072, 16, 224),
    "openai/clip-vit-large-patch14": VitArch(1024, 24, 16, 4096, 14, 224),
    # small geometries for parity tests (head_dim stays 64 like every real family)
    "tiny-vit": VitArch(128, 2, 2, 256, 32, 64),
}

GPT_ZOO: Dict[str, GptArch] = {
    "gpt2": GptArch(768, 12, 12),
    "gpt2-medium": GptArch(1024, 24, 16),
    "microsoft/DialoGPT-medium": GptArch(1024, 24, 16),
    "gpt2-large": GptArch(1280, 36, 20),
    "gpt2-xl": GptArch(1600, 48, 25),
    "tiny-gpt2": GptArch(128, 2, 2, n_pos=64, base_vocab=509),
}


def make_arch(vision_model: str = "openai/clip-vit-base-patch32",
              text_model: str = "gpt2-medium",
              projection_dim: int = 512) -> ModelArch:
    if vision_model not in VIT_ZOO:
        raise ValueError(f"unknown vision tower {vision_model!r}; known: {sorted(VIT_ZOO)}")
    if text_model not in GPT_ZOO:
        raise ValueError(f"unknown text tower {text_model!r}; known: {sorted(GPT_ZOO)}")
    gpt = GPT_ZOO[text_model]
    if gpt.hidden % 8 != 0:
        raise ValueError("decoder width must be divisible by the 8 cross-attention heads")
    return ModelArch(vit=VIT_ZOO[vision_model], gpt=gpt, proj_dim=projection_dim)


def tiny_arch() -> ModelArch:
    """Geometry used by the golden end-to-end fixture."""
    return ModelArch(vit=VIT_ZOO["tiny-vit"], gpt=GPT_ZOO["tiny-gpt2"], proj_dim=64)


def with_layers(arch: ModelArch, vit_layers: int, gpt_layers: int) -> ModelArch:
    return replace(arch, vit=replace(arch.vit, layers=vit_layers),
                   gpt=replace(arch.gpt, layers=gpt_layers))
