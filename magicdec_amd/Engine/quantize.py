"""Weight-only int8 linears (the public names of the reference's Engine/quantize.py: `dynamically_quantize_per_channel`,
`WeightOnlyInt8QuantHandler`, `WeightOnlyInt8Linear`, `replace_linear_weight_only_int8_per_channel`; the loader
switches them in for checkpoints whose path contains "int8", Engine/utils.py:201-205).

Numerical contract (pinned byte for byte against the reference's quantiser in tests/golden/int8_quant.json):
  * one symmetric scale per output row, `max|w_row| / 127.5`, floored at float32 eps and stored in the weight's dtype;
  * `q = clamp(round(w / scale), -128, 127)` as int8, zero points all zero;
  * a runtime layer carries an int8 `weight` [out, in], bf16 `scales` [out] and no bias, and computes
    `bf16(x @ q^T) * scales` -- the product is rounded to bf16 BEFORE the bf16 scale multiply.

MI355X side: a decode / verify step (M <= 256) streams the int8 rows through `md_linear` (MD_W_INT8: 1 byte per weight
from HBM, exact int8 -> bf16 in registers, the scale multiply in the epilogue, csrc/gemm.hip); prefill-sized products
dequantise into a library GEMM.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

INT8_LO, INT8_HI = -128, 127


def dynamically_quantize_per_channel(x, quant_min, quant_max, target_dtype):
    """Symmetric per-row quantisation of a 2-d float tensor -> (q [rows, cols] target_dtype, scales [rows] x.dtype,
    zero_points [rows] int64 zeros).  Reference: Engine/quantize.py:7-41."""
    lo, hi = torch.aminmax(x, dim=1)
    # the row's largest magnitude, written as the reference does (negative part / positive part clamped at zero)
    reach = torch.max(-torch.clamp(lo, max=0), torch.clamp(hi, min=0))
    half_range = float(quant_max - quant_min) / 2
    scales = torch.clamp(reach / half_range, min=torch.finfo(torch.float32).eps).to(x.dtype)
    zero_points = torch.zeros(lo.size(), dtype=torch.int64, device=x.device)
    q = torch.round(x / scales.unsqueeze(-1)) + zero_points.unsqueeze(-1)
    return torch.clamp(q, quant_min, quant_max).to(target_dtype), scales, zero_points


class WeightOnlyInt8Linear(nn.Module):
    """Runtime layer: int8 rows + per-row bf16 scales, never a bias (Engine/quantize.py:72-86)."""
    __constants__ = ["in_features", "out_features"]

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None, dtype=None) -> None:
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.register_buffer("weight", torch.empty((out_features, in_features), dtype=torch.int8, device=device))
        self.register_buffer("scales", torch.ones(out_features, dtype=torch.bfloat16, device=device))
        self.bias = None

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        from .. import ops
        rows = input.reshape(-1, input.shape[-1])
        if input.is_cuda and ops.linear_supported(rows.shape[0], self.out_features, self.in_features):
            if getattr(self, "_workspace", None) is None:
                self._workspace = ops.AttnWorkspace(input.device)
            out = ops.linear(rows, self.weight, scales=self.scales, workspace=self._workspace)
            return out.view(*input.shape[:-1], -1)
        return F.linear(input, self.weight.to(dtype=input.dtype)) * self.scales


def _linears(module: nn.Module, prefix: str = ""):
    """(parent, attribute name, qualified name, layer) of every nn.Linear below `module`, depth first."""
    for name, child in module.named_children():
        fqn = f"{prefix}{name}"
        if isinstance(child, nn.Linear):
            yield module, name, fqn, child
        else:
            yield from _linears(child, fqn + ".")


def replace_linear_weight_only_int8_per_channel(module):
    """In place: every nn.Linear becomes an (uninitialised) WeightOnlyInt8Linear of the same shape."""
    for parent, name, _, lin in list(_linears(module)):
        setattr(parent, name, WeightOnlyInt8Linear(lin.in_features, lin.out_features))


class WeightOnlyInt8QuantHandler:
    """Offline side (quantise a float model's state dict) and runtime side (swap the modules before loading such a
    state dict) of the int8 path; Engine/quantize.py:51-69."""

    def __init__(self, mod):
        self.mod = mod

    @torch.no_grad()
    def create_quantized_state_dict(self):
        sd = self.mod.state_dict()
        for _, _, fqn, lin in _linears(self.mod):
            q, scales, _ = dynamically_quantize_per_channel(lin.weight.float(), INT8_LO, INT8_HI, torch.int8)
            sd[fqn + ".weight"] = q
            sd[fqn + ".scales"] = scales.to(lin.weight.dtype)
        return sd

    def convert_for_runtime(self):
        replace_linear_weight_only_int8_per_channel(self.mod)
        return self.mod
