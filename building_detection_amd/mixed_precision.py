"""Precision policy, named after tf.keras.mixed_precision (the reference itself is fp32 end to end and never calls it).

    from building_detection_amd import mixed_precision
    mixed_precision.set_global_policy("mixed_bfloat16")   # models built from now on store activations in bf16
    model = Xception_DeepLabV3_Plus(...)                   # or Model(..., dtype="mixed_bfloat16") for one model

"mixed_bfloat16": every activation and activation gradient is stored in bf16 (half the HBM traffic of the
bandwidth-bound layers), convolutions multiply bf16 operands on the matrix pipe in ONE pass with fp32 accumulation,
all other kernels widen to fp32 on load; weights (fp32 master copies), BatchNormalization parameters and statistics,
the softmax head with its logits / probabilities, the loss, weight gradients and Adam stay fp32.  "float32" is the
reference's precision and the default.  BASELINE.json configs[2].
"""
from __future__ import annotations

_POLICY = "float32"
_NAMES = {"float32": "float32", "fp32": "float32", None: None, "mixed_bfloat16": "bfloat16", "bfloat16": "bfloat16",
          "bf16": "bfloat16"}


def set_global_policy(name: str) -> None:
    global _POLICY
    if name not in _NAMES or name is None:
        raise ValueError(f"unknown precision policy {name!r}: use 'float32' or 'mixed_bfloat16'")
    _POLICY = _NAMES[name]


def global_policy() -> str:
    return "mixed_bfloat16" if _POLICY == "bfloat16" else "float32"


def resolve(dtype) -> str:
    """-> "float32" | "bfloat16" (the storage dtype of a model's activations)."""
    if dtype not in _NAMES:
        raise ValueError(f"unknown dtype / policy {dtype!r}: use 'float32' or 'mixed_bfloat16'")
    return _NAMES[dtype] or _POLICY
