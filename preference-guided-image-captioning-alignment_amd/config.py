"""YAML configuration with dotted access, the surface the reference's trainer and CLI read
(reference utils/config.py: ``Config(path)``, ``.get("a.b.c", default)``, ``.set``, ``get_stage{1,2}_config``,
``PGCA_*``-style environment overrides).  It parses the reference's ``configs/*.yaml`` unchanged and adds an
optional ``mi355x:`` section whose defaults reproduce the reference's behaviour.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Any, Dict, Optional

import yaml

REQUIRED_SECTIONS = ("model", "training")

MI355X_DEFAULTS = {
    "dpo": {"reference_free": True, "label_smoothing": 0.0},   # True = the reference trainer's 2-forward loss
    "stage1": {"global_negatives": False},                      # False = local negatives, as the reference
    "allreduce_bucket_elems": 64 * 1024 * 1024,
    "allreduce_layer_group": 4,
    "clip_every_micro_step": False,                             # reference quirk
    "prefetch_depth": 2,                                        # batches prepared ahead on a side stream (0 = inline)
    "gpt2_pdrop": 0.1,                                          # HF GPT2Config embd/attn/resid dropout (train mode)
    "allreduce_bf16": False,                                    # bf16-compressed gradient all-reduce (halves xGMI bytes)
    "reset_best_val_between_stages": False,                     # False = as the reference: Stage 2 inherits best_val_loss
    "packed_rows": True,                                        # trunks run on the real tokens' rows only (same results)
}


class Config:
    def __init__(self, config_path: Optional[str] = None) -> None:
        if config_path is None:
            config_path = str(Path(__file__).resolve().parent.parent / "configs" / "default.yaml")
        self.config_path = Path(config_path)
        with open(self.config_path, "r", encoding="utf-8") as f:
            self.config: Dict[str, Any] = yaml.safe_load(f) or {}
        for sec in REQUIRED_SECTIONS:
            if sec not in self.config:
                raise ValueError(f"Missing required configuration section: {sec}")
        for st in ("stage1", "stage2"):
            if st not in self.config["training"]:
                raise ValueError(f"Missing required configuration section: training.{st}")
        self._apply_env_overrides()

    # the reference's own environment overrides (utils/config.py:91-136), same names, same targets
    REFERENCE_ENV = {
        "CONCEPTUAL_CAPTIONS_PATH": "data.conceptual_captions_path", "ULTRAFEEDBACK_PATH": "data.ultrafeedback_path",
        "CAPTION_ALIGNMENT_DATA_DIR": "data.conceptual_captions_path",
        "OUTPUT_DIR": "paths.output_dir", "CACHE_DIR": "paths.cache_dir",
        "CAPTION_ALIGNMENT_CACHE_DIR": "paths.cache_dir", "CAPTION_ALIGNMENT_OUTPUT_DIR": "paths.output_dir",
        "CAPTION_ALIGNMENT_LOG_DIR": "paths.log_dir",
        "CAPTION_ALIGNMENT_VISION_MODEL": "model.vision_model", "CAPTION_ALIGNMENT_TEXT_MODEL": "model.text_model",
        "CAPTION_ALIGNMENT_DEVICE": "hardware.device",
        "CAPTION_ALIGNMENT_BATCH_SIZE": "training.stage1.batch_size",
        "CAPTION_ALIGNMENT_LEARNING_RATE": "training.stage1.learning_rate",
        "CAPTION_ALIGNMENT_NUM_EPOCHS": "training.stage1.num_epochs", "CAPTION_ALIGNMENT_LOG_LEVEL": "logging.level",
        "WANDB_PROJECT": "logging.wandb_project", "WANDB_ENTITY": "logging.wandb_entity",
        "MLFLOW_EXPERIMENT": "logging.mlflow_experiment", "MLFLOW_TRACKING_URI": "logging.mlflow_tracking_uri",
        "CAPTION_ALIGNMENT_NUM_WORKERS": "data.num_workers", "CAPTION_ALIGNMENT_PIN_MEMORY": "data.pin_memory",
        "CAPTION_ALIGNMENT_MIXED_PRECISION": "hardware.mixed_precision",
    }

    @staticmethod
    def _convert_env_value(value: str) -> Any:
        """The reference's conversion (utils/config.py:151-182): booleans by name, then int, then float, else the string."""
        low = value.lower()
        if low in ("true", "1", "yes", "on"):
            return True
        if low in ("false", "0", "no", "off"):
            return False
        try:
            if "." not in value and "e" not in low:
                return int(value)
        except ValueError:
            pass
        try:
            return float(value)
        except ValueError:
            return value

    def _apply_env_overrides(self) -> None:
        """The reference's variables (``REFERENCE_ENV``: an empty value is ignored, as there), then this build's generic
        ``PGCA_CFG_<SECTION>__<KEY>[__<SUBKEY>]=value`` form (values parsed as YAML scalars)."""
        for name, path in self.REFERENCE_ENV.items():
            v = os.getenv(name)
            if v:
                self.set(path, self._convert_env_value(v))
        for k, v in os.environ.items():
            if not k.startswith("PGCA_CFG_"):
                continue
            path = k[len("PGCA_CFG_"):].lower().split("__")
            self.set(".".join(path), yaml.safe_load(v))

    def get(self, path: str, default: Any = None) -> Any:
        cur: Any = self.config
        try:
            for key in path.split("."):
                cur = cur[key]
            return cur
        except (KeyError, TypeError):
            if path.startswith("mi355x."):
                cur = MI355X_DEFAULTS
                try:
                    for key in path.split(".")[1:]:
                        cur = cur[key]
                    return cur
                except (KeyError, TypeError):
                    return default
            return default

    def set(self, path: str, value: Any) -> None:
        keys = path.split(".")
        cur = self.config
        for key in keys[:-1]:
            cur = cur.setdefault(key, {})
        cur[keys[-1]] = value

    def get_model_config(self) -> Dict[str, Any]:
        return self.config["model"]

    def get_training_config(self) -> Dict[str, Any]:
        return self.config["training"]

    def get_stage1_config(self) -> Dict[str, Any]:
        return self.config["training"]["stage1"]

    def get_stage2_config(self) -> Dict[str, Any]:
        return self.config["training"]["stage2"]

    def save(self, path: Optional[str] = None) -> None:
        with open(path or self.config_path, "w", encoding="utf-8") as f:
            yaml.safe_dump(self.config, f, default_flow_style=False, indent=2)

    def __repr__(self) -> str:
        return f"Config({self.config_path})"
