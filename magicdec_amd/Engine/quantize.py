"""Weight-only int8 per-channel quantisation of the linears (Engine/quantize.py of the reference, same public names:
`dynamically_quantize_per_channel`, `WeightOnlyInt8QuantHandler`, `WeightOnlyInt8Linear`,
`replace_linear_weight_only_int8_per_channel`; triggered by "int8" in the checkpoint path, Engine/utils.py:201-205).

Semantics kept from the reference: symmetric per-output-channel scales `max|w| / 127.5` (clamped at fp32 eps, stored
in the weight's dtype), `round(w / scale)` clamped to [-128, 127]; the runtime module holds an int8 `weight` and bf16
`scales` buffer and NO bias (a biased checkpoint does not load, as in the reference); forward is
`F.linear(x, weight.to(x.dtype)) * scales` -- i.e. the GEMM output is rounded to bf16 before the bf16 scale multiply.

MI355X-native underneath: in decode / verify steps (M <= 256) the int8 rows are streamed straight into the skinny GEMM
(md_linear, MD_W_INT8: 1 byte per weight from HBM, exact int8 -> bf16 conversion in registers, the scale multiply
fused into the epilogue); prefill-sized products dequantise on the fly into a library GEMM.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


def dynamically_quantize_per_channel(x, quant_min, quant_max, target_dtype):
    """Engine/quantize.py:7-41: symmetric, per row (axis 0), returns (int weights, scales, zero_points)."""
    eps = torch.finfo(torch.float32).eps
    min_val, max_val = torch.aminmax(x, dim=1)
    max_abs = torch.max(-torch.clamp(min_val, max=0), torch.clamp(max_val, min=0))
    scales = torch.clamp(max_abs / (float(quant_max - quant_min) / 2), min=eps).to(x.dtype)
    zero_points = torch.zeros(min_val.size(), dtype=torch.int64, device=x.device)
    quant = torch.clamp(torch.round(x / scales.unsqueeze(-1)) + zero_points.unsqueeze(-1), quant_min, quant_max)
    return quant.to(target_dtype), scales, zero_points


class WeightOnlyInt8Linear(nn.Module):
    """Engine/quantize.py:72-86."""
    __constants__ = ["in_features", "out_features"]

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None, dtype=None) -> None:
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.register_buffer("weight", torch.empty((out_features, in_features), dtype=torch.int8, device=device))
        self.register_buffer("scales", torch.ones(out_features, dtype=torch.bfloat16, device=device))
        self.bias = None

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        from .. import ops
        x2 = input.reshape(-1, input.shape[-1])
        if input.is_cuda and ops.linear_supported(x2.shape[0], self.out_features, self.in_features):
            ws = getattr(self, "_workspace", None)
            if ws is None:
                ws = self._workspace = ops.AttnWorkspace(input.device)
            return ops.linear(x2, self.weight, scales=self.scales, workspace=ws).view(*input.shape[:-1], -1)
        return F.linear(input, self.weight.to(dtype=input.dtype)) * self.scales


def replace_linear_weight_only_int8_per_channel(module):
    for name, child in module.named_children():
        if isinstance(child, nn.Linear):
            setattr(module, name, WeightOnlyInt8Linear(child.in_features, child.out_features))
        else:
            replace_linear_weight_only_int8_per_channel(child)


class WeightOnlyInt8QuantHandler:
    """Engine/quantize.py:51-69."""

    def __init__(self, mod):
        self.mod = mod

    @torch.no_grad()
    def create_quantized_state_dict(self):
        cur_state_dict = self.mod.state_dict()
        for fqn, mod in self.mod.named_modules():
            if isinstance(mod, torch.nn.Linear):
                int8_weight, scales, _ = dynamically_quantize_per_channel(mod.weight.float(), -128, 127, torch.int8)
                cur_state_dict[f"{fqn}.weight"] = int8_weight
                cur_state_dict[f"{fqn}.scales"] = scales.to(mod.weight.dtype)
        return cur_state_dict

    def convert_for_runtime(self):
        replace_linear_weight_only_int8_per_channel(self.mod)
        return self.mod
