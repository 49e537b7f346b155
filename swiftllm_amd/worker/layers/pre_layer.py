"""Embedding lookup. Reference: swiftllm/worker/layers/pre_layer.py:5-20."""
import torch


class LlamaPreLayer:
    def __init__(self, model_config, weights):
        self.model_config = model_config
        self.weights = weights

    def forward(self, input_ids: torch.Tensor) -> torch.Tensor:
        """input_ids int32 [num_tokens] -> [num_tokens, hidden]."""
        return torch.embedding(self.weights.wte, input_ids, padding_idx=-1)
