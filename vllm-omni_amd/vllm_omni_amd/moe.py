"""Sparse-MoE MLP of the Qwen3-Omni talker on MI355X: the layer upstream's `SharedFusedMoE`
provides behind `Qwen3OmniMoeModel` (vllm_omni/model_executor/models/qwen3_omni/qwen3_omni_moe_talker.py:476-513).
Weight names follow the HF module tree (`gate.weight`, `experts.gate_up_proj`, `experts.down_proj`,
`shared_expert.*`, `shared_expert_gate.weight`).  The router and shared-expert projections are plain dense GEMMs
(library); routing, grouping, both expert GEMMs and the combine are the hand-written kernels of csrc/fused_moe.hip.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn.functional as F

from . import ops
from .layers import _init_weight


@dataclass
class MoEConfig:
    """``transformers`` 5.15.0 class defaults of ``Qwen3OmniMoeTalkerTextConfig`` (MoE part)."""

    hidden_size: int = 1024
    moe_intermediate_size: int = 384
    num_experts: int = 128
    num_experts_per_tok: int = 8
    norm_topk_prob: bool = False
    shared_expert_intermediate_size: Optional[int] = 2048   # None in the class default => config.intermediate_size


# VLLM_OMNI_AMD_MOE_SIDE_STREAM=0: the shared expert in line with the routed path (A/B)
_SIDE_STREAM = os.environ.get("VLLM_OMNI_AMD_MOE_SIDE_STREAM", "1") != "0"
_SIDE_STREAMS: dict = {}


def _side_stream(device) -> "torch.cuda.Stream":
    key = torch.device(device).index
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


class Qwen3OmniMoeSparseMoeBlock:
    def __init__(self, cfg: MoEConfig, dtype=torch.bfloat16, device=None, generator=None):
        self.config, self.dtype = cfg, dtype
        h, i, e = cfg.hidden_size, cfg.moe_intermediate_size, cfg.num_experts
        self.gate_weight = _init_weight(e, h, dtype, device, generator)
        self.gate_up_proj = _init_weight(e * 2 * i, h, dtype, device, generator).view(e, 2 * i, h)
        self.down_proj = _init_weight(e * h, i, dtype, device, generator).view(e, h, i)
        s = cfg.shared_expert_intermediate_size
        self.shared_gate_up = self.shared_down = self.shared_expert_gate = None
        if s:
            self.shared_gate_up = _init_weight(2 * s, h, dtype, device, generator)
            self.shared_down = _init_weight(h, s, dtype, device, generator)
            self.shared_expert_gate = _init_weight(1, h, dtype, device, generator)

    def __call__(self, x: torch.Tensor, routing=None) -> torch.Tensor:
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        shared = None
        if self.shared_gate_up is not None and x2.is_cuda and _SIDE_STREAM and torch.cuda.is_current_stream_capturing():
            # The shared expert (three small launches, ~20 us of a mostly idle GPU at a decode batch) runs beside the
            # routed path (router, grouping, 300 MB of expert weights) on a second stream and joins in front of the
            # combine launch: parallel branches of the captured hipGraph (7.03 -> 6.73 ms per step at batch 512, 3.47 ->
            # 3.33 at 32).  Only under capture: eagerly the fork / join events cost the host more than the overlap gives
            # (a layer ran 194 us against 125).
            cur = torch.cuda.current_stream()
            side = _side_stream(x2.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                sh = F.linear(ops.silu_and_mul(F.linear(x2, self.shared_gate_up)), self.shared_down)
            sh.record_stream(cur)
            x2.record_stream(side)

            def shared():   # called by fused_moe right before the combine launch
                cur.wait_stream(side)
                return sh
        elif self.shared_gate_up is not None:
            shared = F.linear(ops.silu_and_mul(F.linear(x2, self.shared_gate_up)), self.shared_down)
        logits = F.linear(x2, self.gate_weight)
        out = ops.fused_moe(x2, logits, self.gate_up_proj, self.down_proj, self.config.num_experts_per_tok,
                            self.config.norm_topk_prob, shared_out=shared,
                            shared_gate_weight=self.shared_expert_gate.view(-1) if shared is not None else None,
                            routing=routing)
        return out.view(shape)

    def load_weights(self, w: dict[str, torch.Tensor]) -> set[str]:
        loaded = set()
        pairs = [("gate.weight", self.gate_weight), ("experts.gate_up_proj", self.gate_up_proj),
                 ("experts.down_proj", self.down_proj)]
        if self.shared_gate_up is not None:
            pairs += [("shared_expert.down_proj.weight", self.shared_down), ("shared_expert_gate.weight", self.shared_expert_gate)]
            if "shared_expert.gate_proj.weight" in w and "shared_expert.up_proj.weight" in w:
                self.shared_gate_up.copy_(torch.cat([w["shared_expert.gate_proj.weight"], w["shared_expert.up_proj.weight"]], 0)
                                          .to(self.dtype))
                loaded |= {"shared_expert.gate_proj.weight", "shared_expert.up_proj.weight"}
        for name, dst in pairs:
            if name in w:
                dst.copy_(w[name].to(self.dtype))
                loaded.add(name)
        return loaded
