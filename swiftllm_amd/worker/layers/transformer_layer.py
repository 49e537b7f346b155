"""One LLaMA transformer block on the gfx950 operators.

Operator order and buffer aliasing follow swiftllm/worker/layers/transformer_layer.py:31-130:
residual-add + attention norm, q/k/v projections, rotary, KV store, attention (prefill on the main
stream; decode — when it rides along with a prefill batch, SARATHI-style — on a side stream fenced
by events), output projection, residual-add + FFN norm, up/gate projection, SiLU-gate, down
projection. Differences: prefill attention is our own MFMA kernel (the reference calls the
third-party vllm_flash_attn), q/k/v may come from one fused GEMM, and a pure-decode batch runs on
one stream with rotary + KV store fused into a single launch.

Decode fast path (use_skinny_gemm): projections whose K is split across workgroups hand their fp32
partial slabs (SplitKPartials) straight to the next kernel — fused qkv -> rotary+KV-store,
o_proj -> the FFN's fused_add_rmsnorm, down_proj -> the NEXT layer's fused_add_rmsnorm — so `forward`
may return, and accept, a SplitKPartials in place of the activation tensor.

Deferred RMSNorm (defer_rmsnorm, bfloat16 only, batches of <= 32 sequences, hidden % 1024 == 0): the residual-add + norm consumers of
the o_proj / down_proj slabs only do the element-wise part, in a launch that fills the chip (add_scale_from_splitk);
the per-token 1/rms is applied by the consumer of the normalised activations — the slab-fed attention prologue for the
qkv projection, the SiLU-gate GEMM's epilogue for the FFN — in fp32, before its one rounding.

Very small decode batches (tiny_decode_batches; policy limit <= 2 sequences = kernels/linear.py _TINY_POLICY_M, the kernels
take up to 4; bfloat16 only, as it rides on the deferred norm): the two consumers above disappear altogether — the
qkv projection and the up/gate projection sum the previous projection's slabs themselves while their first weight
tiles are in flight (kernels/linear.py: linear_splitk_from_splitk / linear_silu_gate_from_splitk, csrc/gemm_tiny.hip),
the residual stream ping-pongs between two buffers: 5 launches per layer instead of 7. Since r05 this path is the FALLBACK
at these batch sizes, by measurement: with rows_decode on (the default) the row-owned layer below also runs 5 launches and
is faster — 78.5 vs 82.0 us for the projection side of a layer at one sequence, 82.9 for the consumer path
(profiles/r06b_rows_ab.jsonl, tools/gemm_rows_micro.py --layer) — so a layer only receives split-K slabs, and this path
only runs, where the row-owned kernels do not take the shape (or rows_decode is switched off, as its tests do).

Row-owned projections (rows_decode, bfloat16, on the deferred-norm path): o_proj for <= ROWS_O_MAX_M sequences and down_proj for
<= ROWS_DOWN_MAX_M add themselves into the residual buffer in their own epilogue (kernels/linear.py: linear_rows_add,
csrc/gemm_rows.hip) and the next projection normalises the raw residual rows while it stages them (linear_silu_gate_nf /
linear_splitk_nf): no slabs and no consumer launch on that side. A layer whose down projection ran that way returns a
RawResidual marker instead of an activation tensor.
"""
import torch

from ..kernels.linear import SplitKPartials, linear, linear_silu_gate, linear_splitk
from ..kernels.linear import _TINY_POLICY_M as TINY_POLICY_M
from ..kernels.linear import attn_partials_ok, linear_splitk_from_attn_partials
from ..kernels.linear import (row_scaled_silu_gate_ok, alt_residual_like, linear_silu_gate_from_splitk,
                              linear_splitk_from_splitk, tiny_from_splitk_ok)
from ..kernels.linear import RawResidual, linear_rows_add, linear_silu_gate_nf, linear_splitk_nf, nf_ok, rows_add_ok
from ..kernels.linear import linear_silu_gate_nx, linear_splitk_nx, nx_ok
from ..kernels.rmsnorm import RowScalePending
from ..kernels.rmsnorm import (add_scale_from_splitk, deferred_norm_ok, fused_add_rmsnorm_inplace,
                               fused_add_rmsnorm_from_splitk)
from ..kernels.rotary_emb import (rotary_embedding_inplace, rotary_embedding_and_store_kvcache_decode,
                                  rotary_embedding_and_store_kvcache_decode_from_splitk,
                                  rotary_embedding_and_store_kvcache_prefill)
from ..kernels.kvcache_mgmt import store_kvcache
from ..kernels.prefill_attn import prefill_attention
from ..kernels.paged_attn import paged_attention, paged_attention_from_qkv_splitk
from ..kernels.silu_and_mul import silu_and_mul_inplace


class LlamaTransformerLayer:
    # Where the row-owned projections win on MI355X (tools/gemm_rows_micro.py --layer, tools/rows_kernel_us.py): every
    # workgroup of such a projection pulls ALL of x through its own L1, so the win shrinks with the batch. With x staged as
    # full lines through LDS (r06b, csrc/gemm_rows.hip) o_proj (x = M x 8 KiB) beats the split-K pair at 32 sequences by
    # ~4 us (10.5 vs 9.6 + 5), down_proj (x = M x 28 KiB) wins by ~3 us at 16 (23.8 vs 20.5 + 5; layer chain 83.1 vs 86.3),
    # ties at 24 and loses at 32 (28.4 vs 21.4 + 5) — profiles/r06b_rows_ab.jsonl, r06b_rows_kernel_us.jsonl. (r05: 8.)
    ROWS_O_MAX_M = 32
    ROWS_DOWN_MAX_M = 16

    def __init__(self, model_config, engine_config, weight, decoding_piggyback_stream, layer_id: int):
        self.model_config = model_config
        self.engine_config = engine_config
        self.weight = weight
        self.decoding_piggyback_stream = decoding_piggyback_stream
        self.layer_id = layer_id
        self.skinny = bool(getattr(engine_config, "use_skinny_gemm", False))
        self._qkv_splits = None     # k-splits the skinny GEMM picks for the fused qkv projection (cached)
        self._qkv_even = None       # ... and whether they are even splits of whole 128-column tiles (the exact norm on the fly)
        self._tiny_ok = None        # can this layer run the tiny-batch (<= TINY_POLICY_M sequences) path (cached)

    def _split_qkv(self, qkv: torch.Tensor):
        cfg = self.model_config
        hq, hkv = cfg.num_q_heads * cfg.head_dim, cfg.num_kv_heads * cfg.head_dim
        t = qkv.shape[0]
        return (qkv[:, :hq].view(t, cfg.num_q_heads, cfg.head_dim),
                qkv[:, hq:hq + hkv].view(t, cfg.num_kv_heads, cfg.head_dim),
                qkv[:, hq + hkv:].view(t, cfg.num_kv_heads, cfg.head_dim))

    def _project_qkv(self, x: torch.Tensor):
        cfg, w, sk = self.model_config, self.weight, self.skinny
        if w.qkv_proj is not None:
            return self._split_qkv(linear(x, w.qkv_proj, sk))   # q/k/v are column slices of one output
        t = x.shape[0]
        return (linear(x, w.q_proj, sk).view(t, cfg.num_q_heads, cfg.head_dim),
                linear(x, w.k_proj, sk).view(t, cfg.num_kv_heads, cfg.head_dim),
                linear(x, w.v_proj, sk).view(t, cfg.num_kv_heads, cfg.head_dim))

    def forward(self, input_embds, residual_buf: torch.Tensor, k_cache: torch.Tensor,
                v_cache: torch.Tensor, block_table: torch.Tensor, infer_state):
        cfg, ecfg, w, st = self.model_config, self.engine_config, self.weight, infer_state

        # residual_buf <- input_embds + residual_buf ; input_embds <- rmsnorm(residual_buf)
        row_scale = None
        if isinstance(input_embds, RawResidual):        # the previous layer's down projection is already in residual_buf
            return self._forward_from_raw_residual(residual_buf, k_cache, v_cache, block_table, st, input_embds.ssq)
        if isinstance(input_embds, SplitKPartials):     # the previous layer's down projection, unreduced
            if self._tiny_decode_applies(st, input_embds, residual_buf):
                return self._forward_decode_tiny(input_embds, residual_buf, k_cache, v_cache, block_table, st)
            if self._deferred_attn_norm_ok(st):
                row_scale = add_scale_from_splitk(input_embds, residual_buf, w.attn_norm, cfg.rms_norm_eps)
                input_embds = row_scale.x
            else:
                input_embds = fused_add_rmsnorm_from_splitk(input_embds, residual_buf, w.attn_norm,
                                                            cfg.rms_norm_eps)
        else:
            fused_add_rmsnorm_inplace(input_embds, residual_buf, w.attn_norm, cfg.rms_norm_eps)
        return self._forward_after_attn_norm(input_embds, residual_buf, k_cache, v_cache, block_table, st, row_scale)

    def _slab_fed_attention_applies(self, st) -> bool:
        """Pure-decode batch on the path `fused qkv slabs -> (rotary + KV store + paged attention)` in one launch."""
        cfg, ecfg, w = self.model_config, self.engine_config, self.weight
        return (self.skinny and st.num_prefill_seqs == 0 and 0 < st.num_decoding_seqs <= 256 and not st.ignore_kvcache
                and st.position_indices is not None and getattr(ecfg, "fuse_rope_kvstore", False)
                and getattr(ecfg, "fuse_splitk_consumers", True) and w.qkv_proj is not None
                and getattr(ecfg, "fuse_rope_into_attention", True) and cfg.head_dim in (32, 64, 128))

    def _deferred_attn_norm_ok(self, st) -> bool:
        """The attention norm's 1/rms can be left to the attention prologue: slab-fed attention with <= 4 qkv slabs."""
        cfg, ecfg, w = self.model_config, self.engine_config, self.weight
        if not (getattr(ecfg, "defer_rmsnorm", False) and self._slab_fed_attention_applies(st)
                and deferred_norm_ok(st.num_decoding_seqs, cfg.hidden_size, w.qkv_proj.dtype)):
            return False
        if self._qkv_splits is None:
            from swiftllm_amd import _hip
            lib = _hip.load()
            choose = (lib.swl_gemm_skinny_packed_choose_splits if getattr(w.qkv_proj, "_swl_packed", None) is not None
                      else lib.swl_gemm_skinny_choose_splits)     # what linear_splitk will pick for this weight
            self._qkv_splits = int(choose(w.qkv_proj.shape[0], w.qkv_proj.shape[1]))
        return self._qkv_splits in (1, 2, 4)

    def _rows_applies(self, st, residual_buf, w_proj, x, max_m: int) -> bool:
        """Row-owned projection + norm on the fly for this batch: rows_decode, the deferred-norm conditions (bfloat16,
        slab-fed attention), <= max_m sequences, a packed weight the rows kernel takes."""
        ecfg = self.engine_config
        return (getattr(ecfg, "rows_decode", False) and st.num_decoding_seqs <= max_m and self._deferred_attn_norm_ok(st)
                and rows_add_ok(x, w_proj, residual_buf))

    def _forward_from_raw_residual(self, residual_buf, k_cache, v_cache, block_table, st, ssq=None):
        """residual_buf holds the layer input r (the previous down projection added itself): qkv slabs of
        round(r * attn_norm) with the 1/rms pending -> slab-fed attention -> the rest of the layer. With `ssq` (the rows'
        per-tile sums of squares: the exact path) the slabs are those of the exactly normalised rows, nothing pending."""
        cfg, ecfg, w = self.model_config, self.engine_config, self.weight
        if ssq is not None:
            qkv, pend = linear_splitk_nx(residual_buf, w.attn_norm, w.qkv_proj, cfg.rms_norm_eps, ssq), None
        else:
            qkv, pend = linear_splitk_nf(residual_buf, w.attn_norm, w.qkv_proj, cfg.rms_norm_eps)
        # (the attention output is [tokens, heads * head_dim], which need not be the residual's width)
        o = residual_buf.new_empty((residual_buf.shape[0], cfg.num_q_heads * cfg.head_dim))
        paged_attention_from_qkv_splitk(qkv, k_cache, v_cache, block_table, cfg, ecfg, st, self.layer_id, o,
                                        row_scale=pend)
        return self._forward_after_attention(o, residual_buf, True, st)

    def _rows_exact_applies(self, st, residual_buf, w_proj, x, max_m: int) -> bool:
        """Row-owned projection + EXACT norm on the fly (r06c): where the deferred norm's policy says no — float16, the
        reference's own precision — the same launches keep the reference's rounding points (csrc/gemm_rows.hip leaves the
        rows' per-tile sums of squares, the consuming projection finishes the 1/rms before its first tile)."""
        ecfg = self.engine_config
        return (getattr(ecfg, "rows_decode", False) and getattr(ecfg, "fuse_splitk_consumers", True)
                and st.num_decoding_seqs <= max_m and self._slab_fed_attention_applies(st)
                and not self._deferred_attn_norm_ok(st) and residual_buf.shape[1] % 1024 == 0
                and rows_add_ok(x, w_proj, residual_buf))

    def _qkv_even_split_exists(self) -> bool:
        """linear_splitk_nx takes even K splits of whole 128-column tiles only."""
        if self._qkv_even is None:
            from swiftllm_amd import _hip
            w = self.weight.qkv_proj
            ks = int(_hip.load().swl_gemm_skinny_packed_choose_splits(w.shape[0], w.shape[1]))
            self._qkv_even = ks >= 1 and w.shape[1] % (128 * ks) == 0
        return self._qkv_even

    def _qkv_splits_even(self) -> bool:
        """linear_splitk_nf takes even K splits only (set by _deferred_attn_norm_ok, which every caller checked first)."""
        w = self.weight
        return self._qkv_splits is not None and w.qkv_proj.shape[1] % (128 * self._qkv_splits) == 0

    def _tiny_decode_applies(self, st, partials, residual_buf) -> bool:
        """<= TINY_POLICY_M (2) decoding sequences on the deferred-norm fast path, every projection of the layer split over K (so the
        NEXT layer receives slabs again) and the two consuming projections able to rebuild their input themselves."""
        cfg, ecfg, w = self.model_config, self.engine_config, self.weight
        if not (getattr(ecfg, "tiny_decode_batches", True) and partials.shape[0] <= TINY_POLICY_M
                and self._deferred_attn_norm_ok(st)
                and residual_buf.is_contiguous() and tiny_from_splitk_ok(partials, w.qkv_proj)):
            return False
        if self._tiny_ok is None:
            from swiftllm_amd import _hip
            splits = _hip.load().swl_gemm_skinny_choose_splits
            self._tiny_ok = bool(w.up_gate_proj.shape[0] % 128 == 0 and w.up_gate_proj.shape[1] <= 4096
                                 and getattr(w.up_gate_proj, "_swl_packed", None) is not None
                                 and splits(w.o_proj.shape[0], w.o_proj.shape[1]) > 1
                                 and splits(w.down_proj.shape[0], w.down_proj.shape[1]) > 1)
        return self._tiny_ok

    def _forward_decode_tiny(self, partials, residual_buf, k_cache, v_cache, block_table, st):
        cfg, ecfg, w = self.model_config, self.engine_config, self.weight
        eps = cfg.rms_norm_eps
        alt = alt_residual_like(residual_buf)
        # residual_buf + down slabs -> alt ; qkv slabs of round(alt * attn_norm), 1/rms pending
        qkv, ssq = linear_splitk_from_splitk(partials, residual_buf, alt, w.attn_norm, w.qkv_proj)
        pend = RowScalePending(None, ssq, qkv.k_splits, eps, cfg.hidden_size)
        m = residual_buf.shape[0]
        if st.num_seq_blocks > 1 and attn_partials_ok(m, cfg.num_q_heads, cfg.head_dim, w.o_proj):
            # split sequences: no phase-2 launch — o_proj merges the partials of its K-chunk of heads itself
            scratch = paged_attention_from_qkv_splitk(qkv, k_cache, v_cache, block_table, cfg, ecfg, st, self.layer_id,
                                                      None, row_scale=pend, merge=False)
            attn_out = linear_splitk_from_attn_partials(scratch, st.decoding_seq_lens, m, cfg.num_q_heads, cfg.head_dim,
                                                        st.seq_block_size, st.num_seq_blocks, w.o_proj,
                                                        residual_buf.dtype)
        else:
            o = residual_buf.new_empty((m, cfg.num_q_heads * cfg.head_dim))
            paged_attention_from_qkv_splitk(qkv, k_cache, v_cache, block_table, cfg, ecfg, st, self.layer_id, o,
                                            row_scale=pend)
            attn_out = linear_splitk(o, w.o_proj)
        assert isinstance(attn_out, SplitKPartials)
        # alt + o_proj slabs -> residual_buf ; up * silu(gate) of rmsnorm(residual_buf)
        act = linear_silu_gate_from_splitk(attn_out, alt, residual_buf, w.ffn_norm, eps, w.up_gate_proj)
        return linear_splitk(act, w.down_proj)

    def _forward_after_attn_norm(self, input_embds, residual_buf, k_cache, v_cache, block_table, infer_state,
                                 row_scale=None):
        """`row_scale` (RowScalePending): `input_embds` is round(residual * attn_norm) with its 1/rms pending — only
        ever passed when the slab-fed attention path below is taken (_deferred_attn_norm_ok)."""
        cfg, ecfg, w, st = self.model_config, self.engine_config, self.weight, infer_state

        pure_decode = st.num_prefill_seqs == 0 and st.num_decoding_seqs > 0
        fused_rope_store = (pure_decode and not st.ignore_kvcache and st.position_indices is not None
                            and getattr(ecfg, "fuse_rope_kvstore", False))
        fast = self.skinny and pure_decode and getattr(ecfg, "fuse_splitk_consumers", True)
        qkv = None
        if (fast and fused_rope_store and w.qkv_proj is not None and getattr(ecfg, "fuse_rope_into_attention", True)
                and cfg.head_dim in (32, 64, 128) and st.num_decoding_seqs <= 256):  # (batches the slab-producing GEMMs serve)
            # fused qkv slabs -> (rotary + KV store + paged attention) in one launch
            qkv = linear_splitk(input_embds, w.qkv_proj, always=True)
            if isinstance(qkv, SplitKPartials):
                paged_attention_from_qkv_splitk(qkv, k_cache, v_cache, block_table, cfg, ecfg, st, self.layer_id,
                                                input_embds, row_scale=row_scale)
                return self._forward_after_attention(input_embds, residual_buf, fast, st)
        assert row_scale is None, "deferred RMSNorm reached a path that cannot apply it"
        if fast and fused_rope_store and w.qkv_proj is not None:
            if qkv is None:     # (else: the projection above came back as a plain tensor — batches beyond the slab
                qkv = linear_splitk(input_embds, w.qkv_proj)    # kernels; r03 computed it a second time here)
            if isinstance(qkv, SplitKPartials):
                q, k, v = rotary_embedding_and_store_kvcache_decode_from_splitk(
                    qkv, k_cache, v_cache, block_table, cfg, ecfg, st, self.layer_id)
            else:
                q, k, v = self._split_qkv(qkv)
                rotary_embedding_and_store_kvcache_decode(q, k, v, k_cache, v_cache, block_table, cfg,
                                                          ecfg, st, self.layer_id)
        else:
            q, k, v = self._project_qkv(input_embds)
            if fused_rope_store:
                rotary_embedding_and_store_kvcache_decode(q, k, v, k_cache, v_cache, block_table, cfg,
                                                          ecfg, st, self.layer_id)
            elif (st.num_prefill_seqs > 0 and not st.ignore_kvcache and st.position_indices is not None
                  and getattr(ecfg, "fuse_rope_kvstore", False)):
                # prompt tokens: rotary + KV store in one pass over k (r05); riding decodes: their fused launch
                rotary_embedding_and_store_kvcache_prefill(q, k, v, k_cache, v_cache, block_table, cfg, ecfg, st,
                                                           self.layer_id)
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
        return self._forward_after_attention(input_embds, residual_buf, fast, st)

    def _forward_after_attention(self, input_embds, residual_buf, fast: bool, st):
        cfg, w = self.model_config, self.weight
        if (fast and st is not None and self._rows_applies(st, residual_buf, w.o_proj, input_embds, self.ROWS_O_MAX_M)
                and nf_ok(residual_buf, w.up_gate_proj, w.ffn_norm) and w.up_gate_proj.shape[0] % 64 == 0):
            # o_proj adds itself into the residual; the SiLU-gate GEMM normalises the raw rows on the fly
            linear_rows_add(input_embds, w.o_proj, residual_buf)
            act = linear_silu_gate_nf(residual_buf, w.ffn_norm, cfg.rms_norm_eps, w.up_gate_proj)
            if (self._rows_applies(st, residual_buf, w.down_proj, act, self.ROWS_DOWN_MAX_M)
                    and nf_ok(residual_buf, w.qkv_proj, w.attn_norm) and self._qkv_splits_even()):
                linear_rows_add(act, w.down_proj, residual_buf)
                return RawResidual(residual_buf)
            return linear_splitk(act, w.down_proj)
        if (fast and st is not None and self._rows_exact_applies(st, residual_buf, w.o_proj, input_embds, self.ROWS_O_MAX_M)
                and nx_ok(residual_buf, w.up_gate_proj, w.ffn_norm) and w.up_gate_proj.shape[0] % 64 == 0):
            # the same two launches with the reference's rounding points: o_proj adds itself into the residual and leaves the
            # rows' sums of squares, the SiLU-gate GEMM stages round(r * rstd * w)
            ssq = linear_rows_add(input_embds, w.o_proj, residual_buf, with_ssq=True)
            act = linear_silu_gate_nx(residual_buf, w.ffn_norm, cfg.rms_norm_eps, w.up_gate_proj, ssq)
            if (st.num_decoding_seqs <= self.ROWS_DOWN_MAX_M and rows_add_ok(act, w.down_proj, residual_buf)
                    and nx_ok(residual_buf, w.qkv_proj, w.attn_norm) and self._qkv_even_split_exists()):
                return RawResidual(residual_buf, linear_rows_add(act, w.down_proj, residual_buf, with_ssq=True))
            return linear_splitk(act, w.down_proj)
        if fast:
            attn_out = linear_splitk(input_embds, w.o_proj)
            if isinstance(attn_out, SplitKPartials):
                if (getattr(self.engine_config, "defer_rmsnorm", False)
                        and deferred_norm_ok(*attn_out.shape, attn_out.dtype)
                        and row_scaled_silu_gate_ok(input_embds, w.up_gate_proj)):
                    # element-wise add + scale (fills the chip); the FFN norm's 1/rms goes into the SiLU-gate GEMM
                    pend = add_scale_from_splitk(attn_out, residual_buf, w.ffn_norm, cfg.rms_norm_eps)
                    act = linear_silu_gate(pend.x, w.up_gate_proj, row_scale=pend)
                    return linear_splitk(act, w.down_proj)
                attn_out = fused_add_rmsnorm_from_splitk(attn_out, residual_buf, w.ffn_norm, cfg.rms_norm_eps)
            else:
                fused_add_rmsnorm_inplace(attn_out, residual_buf, w.ffn_norm, cfg.rms_norm_eps)
        else:
            attn_out = linear(input_embds, w.o_proj, self.skinny)
            fused_add_rmsnorm_inplace(attn_out, residual_buf, w.ffn_norm, cfg.rms_norm_eps)
        act = linear_silu_gate(attn_out, w.up_gate_proj) if self.skinny else None
        if act is None:
            up_gate = linear(attn_out, w.up_gate_proj, self.skinny)
            silu_and_mul_inplace(up_gate)
            act = up_gate[:, :cfg.ffn_inter_dim]
        return linear_splitk(act, w.down_proj) if fast else linear(act, w.down_proj, self.skinny)
