"""Final norm, lm_head and greedy sampling. Reference: swiftllm/worker/layers/post_layer.py:9-40."""
import torch

from ..kernels.rmsnorm import rmsnorm_inplace
from ..kernels.linear import linear
from ..kernels.sampling import argmax_rows


class LlamaPostLayer:
    def __init__(self, model_config, weights):
        self.model_config = model_config
        self.weights = weights
        self.skinny = False     # set by LlamaModel from EngineConfig.use_skinny_gemm
        # tests set this to a list: every forward appends its pre-argmax logits (the DEVICE tensor — no
        # host copy here, a forward may be under hipGraph capture; LlamaModel appends a copy after each replay)
        self.logits_tap = None
        self.last_logits = None     # the logits tensor of the most recent (eager or captured) forward

    def forward(self, input_embds: torch.Tensor, infer_state) -> torch.Tensor:
        """[num_tokens, hidden] -> next-token ids int64 [batch_size] (argmax; ties -> lowest id)."""
        idx = infer_state.last_token_indices
        if idx is None:
            # the last token of each prefill sequence, then every decoding token
            idx = torch.cat((
                infer_state.prefill_seq_start_locs + infer_state.prefill_seq_lens - 1,
                torch.arange(infer_state.num_prefill_tokens, infer_state.num_tokens,
                             device=input_embds.device, dtype=torch.int32)))
        last_input = input_embds.index_select(0, idx)    # fresh [batch, hidden] copy
        rmsnorm_inplace(last_input, self.weights.final_norm, self.model_config.rms_norm_eps)
        return self.forward_normed(last_input)

    def forward_normed(self, last_input: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        """lm_head + greedy sampling on rows that already went through the final norm (a pure-decode batch whose
        last add + norm ran fused on the split-K partials of the last down projection: every row is a last token)."""
        logits = linear(last_input, self.weights.lm_head, self.skinny)   # [batch, vocab]
        self.last_logits = logits
        if self.logits_tap is not None:
            self.logits_tap.append(logits)
        return argmax_rows(logits, out)
