"""Token sampling (`compactor_vllm/layers/sampler.py:5-27`): per-row temperature, greedy where it is 0, otherwise one
draw from softmax(logits / T) by the Gumbel-max trick (logits / T minus log of an Exp(1) variate, then argmax)."""
import torch
from torch import nn


class Sampler(nn.Module):
    def forward(self, logits: torch.Tensor, temperatures: torch.Tensor) -> torch.Tensor:
        temps = temperatures.view(-1)
        scaled = logits.float()
        sample = temps != 0.0
        if bool(sample.any()):
            noise = torch.empty_like(scaled).exponential_(1).clamp_min_(1e-10).log()
            t = torch.where(sample, temps, torch.ones_like(temps)).unsqueeze(-1)
            scaled = torch.where(sample.unsqueeze(-1), scaled / t - noise, scaled)
        return scaled.argmax(dim=-1)
