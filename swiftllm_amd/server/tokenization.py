"""Tokenizer front-end. The reference runs a HuggingFace tokenizer in a Ray actor (separate process,
swiftllm/server/tokenization_engine.py:6-18); here it runs in a worker thread of the engine's process
(tokenizers releases the GIL), with the same two calls. Requests that already carry token ids skip it."""
import asyncio
from typing import List


class TokenizationEngine:
    def __init__(self, engine_config):
        self._path = engine_config.model_path
        self._tok = None

    def _tokenizer(self):
        if self._tok is None:
            from transformers import AutoTokenizer
            self._tok = AutoTokenizer.from_pretrained(self._path)
        return self._tok

    def batched_tokenize_sync(self, prompts: List[str]) -> List[List[int]]:
        return self._tokenizer()(prompts, return_attention_mask=False)["input_ids"]

    def decode_sync(self, token_ids: List[int], skip_special_tokens: bool = True) -> str:
        return self._tokenizer().decode(token_ids, skip_special_tokens=skip_special_tokens)

    async def batched_tokenize(self, prompts: List[str]) -> List[List[int]]:
        return await asyncio.get_running_loop().run_in_executor(None, self.batched_tokenize_sync, prompts)

    async def decode(self, token_ids: List[int], skip_special_tokens: bool = True) -> str:
        return await asyncio.get_running_loop().run_in_executor(None, self.decode_sync, token_ids,
                                                                skip_special_tokens)
