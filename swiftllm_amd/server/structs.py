"""Request bookkeeping shared by the scheduler, the engine and the HTTP layer.
Same names and fields as the reference's swiftllm/server/structs.py:4-63."""
import asyncio
import dataclasses
from typing import List, Optional


@dataclasses.dataclass
class StepOutput:
    """One generated token of one request."""
    token_id: int
    request: "Request"


class RawRequest:
    """What a user submits: a prompt and how many tokens to generate. `prompt_token_ids` may be given
    instead of text (tokenizer-less use: benchmarks, synthetic checkpoints)."""

    def __init__(self, prompt: str, output_len: int, prompt_token_ids: Optional[List[int]] = None):
        self.prompt = prompt
        self.output_len = output_len
        self.prompt_token_ids = prompt_token_ids


class Request:
    """A request inside the system: waiting, running (prefill or decode), swapped out, or finished."""

    def __init__(self, raw_request: RawRequest):
        self.prompt_token_ids: List[int] = list(raw_request.prompt_token_ids or [])
        self.prompt_len = len(self.prompt_token_ids)
        self.output_len = raw_request.output_len
        self.output_q: "asyncio.Queue[StepOutput]" = asyncio.Queue()    # streaming consumers read here
        self.finished_event = asyncio.Event()                           # non-streaming consumers wait here
        self.request_id = -1            # row of the block table, assigned when the request is admitted
        self.output_token_ids: List[int] = []
        self.error: Optional[str] = None    # set instead of scheduling when the request can never be served

    def is_finished(self) -> bool:
        return len(self.output_token_ids) >= self.output_len

    def get_cur_output_len(self) -> int:
        return len(self.output_token_ids)

    def is_prefill_stage(self) -> bool:
        return not self.output_token_ids

    def num_tokens(self) -> int:
        """Tokens whose KV must be resident: the prompt plus everything generated so far."""
        return self.prompt_len + len(self.output_token_ids)
