"""oracle/ref_model.py — CPU restatement of the reference's LlamaModel.forward (TEST INFRASTRUCTURE).

The checker for whole-forward parity and the CPU baseline of bench.py; never imported by
swiftllm_amd/. Follows, step by step:
  * swiftllm/worker/model.py:252-359      forward: flattening, metadata, block allocation,
                                          seq_block_size, infer state
  * swiftllm/worker/model.py:228-249      _forward: embedding, L layers, final residual add
  * swiftllm/worker/layers/transformer_layer.py:31-130   operator order and buffer aliasing
  * swiftllm/worker/layers/post_layer.py:18-40           last-token gather, norm, lm_head, argmax
  * swiftllm/worker/block_manager.py:43-86               lowest-id-first block allocation
Operators come from oracle/eager_ops.py. Weights are passed in as a plain dict of HuggingFace-named
tensors (the same dict tests write to safetensors for the product under test).
"""
import itertools
import copy
import types

import torch

from . import eager_ops as ops


class RefBlockManager:
    """swiftllm/worker/block_manager.py:5-103 on CPU tensors."""

    def __init__(self, device_name, num_blocks, max_seqs_in_block_table, max_blocks_per_seq, block_size):
        self.device_name = device_name
        self.num_free_blocks = num_blocks
        self.num_blocks = num_blocks
        self.block_size = block_size
        self.num_seq_allocated_blocks = torch.zeros(max_seqs_in_block_table, dtype=torch.int32)
        self.block_table = torch.zeros(max_seqs_in_block_table, max_blocks_per_seq, dtype=torch.int32)
        self.is_block_free = torch.ones(num_blocks, dtype=torch.bool)

    def allocate_blocks_for_seqs(self, seq_ids: torch.Tensor, target_lens: torch.Tensor):
        target = (target_lens + (self.block_size - 1)) // self.block_size
        have = self.num_seq_allocated_blocks[seq_ids.long()]
        assert (have <= target).all(), "Logic error: some sequences own more blocks than needed"
        need = target - have
        n = int(need.sum())
        if n > self.num_free_blocks:
            raise RuntimeError(
                f"No enough free blocks available on {self.device_name} ({self.num_blocks} in total, "
                f"{self.num_free_blocks} free, {n} requested)")
        picked = torch.nonzero(self.is_block_free)[:n].view(-1)     # lowest ids first (:50)
        self.num_free_blocks -= n
        self.is_block_free[picked] = False
        ops.set_block_table_and_num_seq_alloc_blocks(self.num_seq_allocated_blocks, self.block_table,
                                                     picked, seq_ids, need)
        return picked

    def free_blocks_for_seqs(self, seq_ids: torch.Tensor):
        self.num_free_blocks += int(self.num_seq_allocated_blocks[seq_ids.long()].sum())
        ops.unset_block_table_and_num_seq_alloc_blocks(self.num_seq_allocated_blocks, self.block_table,
                                                       seq_ids, self.is_block_free)

    def gather_allocated_blocks_and_free(self, seq_ids: torch.Tensor):
        ids = ops.gather_allocated_blocks_and_unset(self.num_seq_allocated_blocks, self.block_table,
                                                    seq_ids, self.is_block_free)
        self.num_free_blocks += len(ids)
        return ids

    def get_num_allocated_blocks(self, seq_ids: torch.Tensor):
        return self.num_seq_allocated_blocks[seq_ids.long()]


class RefLlamaModel:
    """The reference data plane on CPU. `score_dtype` selects how decode-attention scores are
    rounded (see eager_ops.paged_attention_phase1)."""

    def __init__(self, model_config, engine_config, state_dict: dict, dtype: torch.dtype,
                 score_dtype: str = "fp32", tied_lm_head: bool = False, dense_decode_attention: bool = False):
        self.model_config = model_config
        self.engine_config = engine_config
        self.dtype = dtype
        self.score_dtype = score_dtype
        # decode attention as one dense softmax per sequence (eager_ops.paged_attention_dense): for TIMING the oracle
        self.dense_decode_attention = dense_decode_attention
        sd = {k: v.to(dtype) for k, v in state_dict.items()}
        self.wte = sd["model.embed_tokens.weight"]
        self.lm_head = self.wte if tied_lm_head else sd["lm_head.weight"]
        self.final_norm = sd["model.norm.weight"]
        self.layers = []
        for i in range(model_config.num_layers):
            p = f"model.layers.{i}."
            self.layers.append(types.SimpleNamespace(
                attn_norm=sd[p + "input_layernorm.weight"],
                q_proj=sd[p + "self_attn.q_proj.weight"], k_proj=sd[p + "self_attn.k_proj.weight"],
                v_proj=sd[p + "self_attn.v_proj.weight"], o_proj=sd[p + "self_attn.o_proj.weight"],
                ffn_norm=sd[p + "post_attention_layernorm.weight"],
                # [up ; gate] (weight.py:133)
                up_gate_proj=torch.cat((sd[p + "mlp.up_proj.weight"], sd[p + "mlp.gate_proj.weight"]), 0),
                down_proj=sd[p + "mlp.down_proj.weight"]))
        self.cos, self.sin = ops.rope_tables(model_config, dtype)
        self.k_cache = self.v_cache = None
        self.gpu_block_manager = self.cpu_block_manager = None
        self.last_logits = None

    def init_kvcache_and_swap(self, num_blocks: int):
        cfg, ecfg = self.model_config, self.engine_config
        shape = (num_blocks, cfg.num_layers, cfg.num_kv_heads, ecfg.block_size, cfg.head_dim)
        self.k_cache = torch.zeros(shape, dtype=self.dtype)
        self.v_cache = torch.zeros(shape, dtype=self.dtype)
        swap_shape = (ecfg.num_cpu_blocks,) + shape[1:]
        self.k_swap = torch.zeros(swap_shape, dtype=self.dtype)
        self.v_swap = torch.zeros(swap_shape, dtype=self.dtype)
        mk = lambda name, n: RefBlockManager(name, n, ecfg.max_seqs_in_block_table,     # noqa: E731
                                             ecfg.max_blocks_per_seq, ecfg.block_size)
        self.gpu_block_manager = mk("GPU", num_blocks)
        self.cpu_block_manager = mk("CPU", ecfg.num_cpu_blocks)

    def fork(self, score_dtype: str = None):
        """A second oracle continuing from this one's state (KV pool, swap pool, block tables): weights shared, state
        copied. For tests that run one prompt pass and then several decode continuations with different score rounding
        (the prompt pass does not depend on it: prefill attention has one rounding, eager_ops.prefill_attention)."""
        other = copy.copy(self)
        if score_dtype is not None:
            other.score_dtype = score_dtype
        for name in ("k_cache", "v_cache", "k_swap", "v_swap", "last_logits"):
            t = getattr(self, name, None)
            setattr(other, name, None if t is None else t.clone())
        other.gpu_block_manager = copy.deepcopy(self.gpu_block_manager)
        other.cpu_block_manager = copy.deepcopy(self.cpu_block_manager)
        return other

    # ---- one transformer block (transformer_layer.py:31-130) ------------------------------------------
    def _layer(self, i, x, residual, st):
        cfg, w = self.model_config, self.layers[i]
        ops.fused_add_rmsnorm_inplace(x, residual, w.attn_norm, cfg.rms_norm_eps)
        t = x.shape[0]
        q = ops.linear(x, w.q_proj).view(t, cfg.num_q_heads, cfg.head_dim)
        k = ops.linear(x, w.k_proj).view(t, cfg.num_kv_heads, cfg.head_dim)
        v = ops.linear(x, w.v_proj).view(t, cfg.num_kv_heads, cfg.head_dim)
        ops.rotary_embedding_inplace(q, k, st)
        bt = self.gpu_block_manager.block_table if not st.ignore_kvcache else None
        if not st.ignore_kvcache:
            ops.store_kvcache(k, v, self.k_cache, self.v_cache, bt, cfg, self.engine_config, st, i)
        o = x   # attention output overwrites the normed activations (:82)
        p = st.num_prefill_tokens
        if st.num_prefill_seqs > 0:
            ops.prefill_attention(q, k, v, o[:p], cfg, self.engine_config, st)
        if st.num_decoding_seqs > 0:
            ov = o[p:].view(-1, cfg.num_q_heads, cfg.head_dim)
            if self.dense_decode_attention:
                ops.paged_attention_dense(q[p:], self.k_cache, self.v_cache, bt, cfg, self.engine_config, st, i, ov)
            else:
                ops.paged_attention(q[p:], self.k_cache, self.v_cache, bt, cfg, self.engine_config, st, i,
                                    ov, self.score_dtype)
        o = ops.linear(o, w.o_proj)
        ops.fused_add_rmsnorm_inplace(o, residual, w.ffn_norm, cfg.rms_norm_eps)
        up_gate = ops.linear(o, w.up_gate_proj)
        ops.silu_and_mul_inplace(up_gate)
        return ops.linear(up_gate[:, :cfg.ffn_inter_dim], w.down_proj)

    def forward(self, input_ids_list, seq_ids_list, decoding_seq_lens_list, ignore_kvcache=False):
        """model.py:252-359."""
        if len(input_ids_list) == 0:
            return []
        cfg = self.model_config
        num_prefill = len(input_ids_list) - len(decoding_seq_lens_list)
        flat = list(itertools.chain(*input_ids_list))
        seq_lengths_list = [len(s) for s in input_ids_list[:num_prefill]] + list(decoding_seq_lens_list)
        seq_ids = torch.tensor(seq_ids_list, dtype=torch.int32)
        seq_lengths = torch.tensor(seq_lengths_list, dtype=torch.int32)
        batch, num_tokens = len(input_ids_list), len(flat)
        plens_list = seq_lengths_list[:num_prefill]
        plens = torch.tensor(plens_list, dtype=torch.int32)
        pstarts = torch.cumsum(plens, 0, dtype=torch.int32) - plens
        dlens = torch.tensor(list(decoding_seq_lens_list), dtype=torch.int32)
        max_dec = max(decoding_seq_lens_list) if decoding_seq_lens_list else 0
        pos = torch.cat([torch.arange(n, dtype=torch.int32) for n in plens_list] + [dlens - 1]) \
            if num_tokens else torch.empty(0, dtype=torch.int32)
        if not ignore_kvcache:
            self.gpu_block_manager.allocate_blocks_for_seqs(seq_ids, seq_lengths)
        sbs = ops.select_seq_block_size(list(decoding_seq_lens_list), cfg.num_kv_heads)
        st = types.SimpleNamespace(
            batch_size=batch, num_tokens=num_tokens, seq_ids=seq_ids,
            softmax_scale=cfg.head_dim ** -0.5,
            num_prefill_seqs=num_prefill, num_prefill_tokens=num_tokens - (batch - num_prefill),
            prefill_seq_start_locs=pstarts,
            prefill_seq_start_locs_with_end=torch.cat([pstarts, torch.tensor([num_tokens - (batch - num_prefill)], dtype=torch.int32)]),
            prefill_seq_lens=plens, max_prefill_len=max(plens_list) if plens_list else 0,
            num_decoding_seqs=batch - num_prefill, decoding_seq_lens=dlens, max_decoding_len=max_dec,
            seq_block_size=sbs, num_seq_blocks=(max_dec + sbs - 1) // sbs,
            position_cos=self.cos[pos.long()], position_sin=self.sin[pos.long()],
            position_indices=None, ignore_kvcache=ignore_kvcache)
        # _forward (model.py:228-249)
        x = torch.embedding(self.wte, torch.tensor(flat, dtype=torch.int32))
        residual = torch.zeros_like(x)
        for i in range(cfg.num_layers):
            x = self._layer(i, x, residual, st)
        x = (x.float() + residual.float()).to(self.dtype)   # `input_embds += residual_buf` in fp16
        # post layer (post_layer.py:18-40)
        last = torch.cat((pstarts + plens - 1,
                          torch.arange(st.num_prefill_tokens, num_tokens, dtype=torch.int32)))
        last_input = x[last.long()].clone()
        ops.rmsnorm_inplace(last_input, self.final_norm, cfg.rms_norm_eps)
        logits = ops.linear(last_input, self.lm_head)
        self.last_logits = logits.float()
        return torch.argmax(logits, dim=1).tolist()

    def _swap(self, seq_ids_list, is_swap_in):
        """model.py:361-379."""
        src = self.cpu_block_manager if is_swap_in else self.gpu_block_manager
        dst = self.gpu_block_manager if is_swap_in else self.cpu_block_manager
        seq_ids = torch.tensor(seq_ids_list, dtype=torch.int32)
        lens = src.get_num_allocated_blocks(seq_ids) * self.engine_config.block_size
        src_ids = src.gather_allocated_blocks_and_free(seq_ids)
        dst_ids = dst.allocate_blocks_for_seqs(seq_ids, lens)
        ops.swap_blocks(src_ids.tolist(), dst_ids.tolist(), is_swap_in, self.k_cache, self.v_cache,
                        self.k_swap, self.v_swap)

    def swap_in_seqs(self, seq_ids_list):
        self._swap(seq_ids_list, True)

    def swap_out_seqs(self, seq_ids_list):
        self._swap(seq_ids_list, False)

    def free_seqs_resources(self, seq_ids_list):
        seq_ids = torch.tensor(seq_ids_list, dtype=torch.int32)
        self.gpu_block_manager.free_blocks_for_seqs(seq_ids)
        self.cpu_block_manager.free_blocks_for_seqs(seq_ids)
