"""One LLaMA transformer block on the gfx950 operators.

Operator order and buffer aliasing follow swiftllm/worker/layers/transformer_layer.py:31-130:
residual-add + attention norm, q/k/v projections, rotary, KV store, attention (prefill on the main
stream; decode — when it rides along with a prefill batch, SARATHI-style — on a side stream fenced
by events), output projection, residual-add + FFN norm, up/gate projection, SiLU-gate, down
projection. Differences: prefill attention is our own MFMA kernel (the reference calls the
third-party vllm_flash_attn), q/k/v may come from one fused GEMM, and a pure-decode batch runs on
one stream with rotary + KV store fused into a single launch.
"""
import torch

from ..kernels.linear import linear
from ..kernels.rmsnorm import fused_add_rmsnorm_inplace
from ..kernels.rotary_emb import rotary_embedding_inplace, rotary_embedding_and_store_kvcache_decode
from ..kernels.kvcache_mgmt import store_kvcache
from ..kernels.prefill_attn import prefill_attention
from ..kernels.paged_attn import paged_attention
from ..kernels.silu_and_mul import silu_and_mul_inplace


class LlamaTransformerLayer:
    def __init__(self, model_config, engine_config, weight, decoding_piggyback_stream, layer_id: int):
        self.model_config = model_config
        self.engine_config = engine_config
        self.weight = weight
        self.decoding_piggyback_stream = decoding_piggyback_stream
        self.layer_id = layer_id
        self.skinny = bool(getattr(engine_config, "use_skinny_gemm", False))

    def _project_qkv(self, x: torch.Tensor):
        cfg, w = self.model_config, self.weight
        hq, hkv = cfg.num_q_heads * cfg.head_dim, cfg.num_kv_heads * cfg.head_dim
        if w.qkv_proj is not None:
            qkv = linear(x, w.qkv_proj, self.skinny)     # [T, hq + 2*hkv]; q/k/v are column slices
            q, k, v = qkv[:, :hq], qkv[:, hq:hq + hkv], qkv[:, hq + hkv:]
        else:
            sk = self.skinny
            q, k, v = linear(x, w.q_proj, sk), linear(x, w.k_proj, sk), linear(x, w.v_proj, sk)
        t = x.shape[0]
        return (q.view(t, cfg.num_q_heads, cfg.head_dim), k.view(t, cfg.num_kv_heads, cfg.head_dim),
                v.view(t, cfg.num_kv_heads, cfg.head_dim))

    def forward(self, input_embds: torch.Tensor, residual_buf: torch.Tensor, k_cache: torch.Tensor,
                v_cache: torch.Tensor, block_table: torch.Tensor, infer_state) -> torch.Tensor:
        cfg, ecfg, w, st = self.model_config, self.engine_config, self.weight, infer_state

        # residual_buf <- input_embds + residual_buf ; input_embds <- rmsnorm(residual_buf)
        fused_add_rmsnorm_inplace(input_embds, residual_buf, w.attn_norm, cfg.rms_norm_eps)
        q, k, v = self._project_qkv(input_embds)

        pure_decode = st.num_prefill_seqs == 0 and st.num_decoding_seqs > 0
        if (pure_decode and not st.ignore_kvcache and st.position_indices is not None
                and getattr(ecfg, "fuse_rope_kvstore", False)):
            rotary_embedding_and_store_kvcache_decode(q, k, v, k_cache, v_cache, block_table, cfg,
                                                      ecfg, st, self.layer_id)
        else:
            rotary_embedding_inplace(q, k, st)
            if not st.ignore_kvcache:
                store_kvcache(k, v, k_cache, v_cache, block_table, cfg, ecfg, st, self.layer_id)

        # attention output overwrites the (already consumed) normed activations
        o = input_embds.view(-1, cfg.num_q_heads, cfg.head_dim)
        p = st.num_prefill_tokens
        if st.num_prefill_seqs > 0 and st.num_decoding_seqs > 0:
            # piggybacked decode: HBM-bound paged attention overlaps the MFMA-bound prefill attention
            assert not st.ignore_kvcache
            stored = torch.cuda.Event()
            stored.record()
            prefill_attention(q, k, v, o, cfg, ecfg, st)
            side = self.decoding_piggyback_stream
            with torch.cuda.stream(side):
                side.wait_event(stored)
                paged_attention(q[p:], k_cache, v_cache, block_table, cfg, ecfg, st, self.layer_id, o[p:])
                decoded = torch.cuda.Event()
                decoded.record()
            torch.cuda.current_stream().wait_event(decoded)
        elif st.num_prefill_seqs > 0:
            prefill_attention(q, k, v, o, cfg, ecfg, st)
        elif st.num_decoding_seqs > 0:
            assert not st.ignore_kvcache
            paged_attention(q, k_cache, v_cache, block_table, cfg, ecfg, st, self.layer_id, o)
        q = k = v = None

        attn_out = linear(input_embds, w.o_proj, self.skinny)
        fused_add_rmsnorm_inplace(attn_out, residual_buf, w.ffn_norm, cfg.rms_norm_eps)
        up_gate = linear(attn_out, w.up_gate_proj, self.skinny)
        silu_and_mul_inplace(up_gate)
        return linear(up_gate[:, :cfg.ffn_inter_dim], w.down_proj, self.skinny)
