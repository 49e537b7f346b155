"""EngineConfig — the engine-level knobs, field-compatible with the reference.

Mirrors swiftllm/engine_config.py:4-84: the nine original fields keep their names, order and
meaning so `EngineConfig(**vars(args))` (reference api_server.py:97) and the keyword construction in
examples/offline.py:22-33 keep working. Fields after `max_tokens_in_batch` are additions of this
MI355X implementation and all have defaults.

Defaults are re-sized for one MI355X (288 GB HBM3E): the reference's CLI defaults target 24-80 GB
parts (num_cpu_blocks 2048 = 4 GiB of swap for Llama-3-8B); here the swap pool default is 16 Ki
blocks and the scheduler limits admit the larger KV pool.
"""
import argparse
import dataclasses


@dataclasses.dataclass
class EngineConfig:
    # Model loading parameters
    model_path: str
    use_dummy: bool

    # PagedAttention-related parameters
    block_size: int
    gpu_mem_utilization: float
    num_cpu_blocks: int
    max_seqs_in_block_table: int
    max_blocks_per_seq: int

    # Scheduling-related parameters
    max_batch_size: int
    max_tokens_in_batch: int

    # ---- MI355X additions (all optional) -------------------------------------------------------
    # Storage/compute dtype of weights, activations and the KV pool. The reference hard-codes
    # float16 (model.py:70,147); bfloat16 is the headline precision on MI355X. NOTE: float16 keeps the reference's
    # rounding points everywhere, so the bfloat16-only deferred norm (defer_rmsnorm, tiny_decode_batches) is off for it;
    # since r06c its row-owned projections run the EXACT norm on the fly instead (rows_decode: the same 5-6 launches per
    # layer with fused_add_rmsnorm's arithmetic — 3.95 vs 4.10 ms per step at batch 32, 3.11 vs 3.25 at batch 1).
    dtype: str = "float16"
    # One [h + 2*KVH*D, h] GEMM instead of three (the reference left this commented out,
    # weight.py:131). fuse_qkv=False + use_skinny_gemm=False reproduces the reference's exact BLAS calls.
    fuse_qkv: bool = True
    # Capture pure-decode forwards into hipGraphs (one per batch size and split-geometry bucket) and replay them: one
    # graph launch per step instead of ~230 kernel launches (batch 32: 4.0 vs 4.35 ms/step on MI355X). The default;
    # False (CLI: --no-hip-graph) launches every kernel from Python.
    use_hip_graph: bool = True
    # Route decode-sized projections (<= 32 tokens) to the hand-written weight-streaming MFMA GEMM
    # instead of hipBLASLt (prefill-sized calls stay on the BLAS).
    use_skinny_gemm: bool = True
    # Keep a second copy of the projection weights in MFMA-fragment order for the decode GEMMs (+1x projection
    # weights in HBM, ~8 % faster weight streaming; the row-major copy stays for prefill).
    pack_decode_weights: bool = True
    # Overrides of the internal decode-path switches below (TUNING_DEFAULTS), e.g. tuning={"defer_rmsnorm": False}: every
    # one of them selects between two parity-tested implementations of the same arithmetic and defaults to the faster one;
    # they exist so that tests and A/B measurements can hold one path against the other, not as deployment knobs.
    tuning: dict = None

    # Internal switches (all on): set through `tuning`, read by the layer code as plain attributes.
    TUNING_DEFAULTS = dict(
        fuse_rope_kvstore=True,         # rotary + KV store in one launch (decode tokens; r05: prompt tokens too, one pass over k)
        fuse_splitk_consumers=True,     # the next kernel sums a projection's split-K slabs (no reduce launches)
        fuse_rope_into_attention=True,  # ... and rotary + KV store run in the paged-attention kernel's prologue
        # apply the RMSNorm scale AFTER the projection that consumes the normalised activations (a per-token scalar), so the
        # residual-add + norm consumers become element-wise kernels that fill the chip; bfloat16 only
        # (kernels/rmsnorm.py: deferred_norm_ok), hidden % 1024 == 0, batches of <= 32 sequences
        defer_rmsnorm=True,
        # batches of <= 2 sequences: the qkv and up/gate projections sum the previous projection's slabs themselves
        # (csrc/gemm_tiny.hip), 5 launches per layer instead of 7
        tiny_decode_batches=True,
        # decode batches of <= 32 sequences: o_proj — and down_proj up to 16 sequences — finish their rows INSIDE the
        # workgroup that owns them (csrc/gemm_rows.hip: K split across the 8 waves, residual add in the epilogue: no slabs,
        # no consumer launch), and the projection that follows applies the norm weight and the 1/rms itself while it stages
        # the raw residual rows ("norm on the fly", gemm_skinny.hip NF: bfloat16, deferred 1/rms; NX: either dtype, the exact
        # norm from the per-tile sums of squares the row-owned kernel leaves): 6 launches per layer at batch 32, 5 up to 16
        rows_decode=True,
        # ONE decoding sequence: the whole transformer stack of the step as one persistent launch (csrc/decode_engine.hip: a
        # loader wave per CU streams that CU's rows of every projection by LDS-DMA, three consumer waves compute, operator
        # boundaries are in-launch granule hand-offs; the reference's rounding points in BOTH dtypes; every wait bounded).
        # OFF by default: measured on MI355X at Llama-3-8B it is parity-green and SLOWER than the 5-6 launches it replaces
        # (3.29 vs 3.03 ms per step at context 1088: its six all-to-all hand-offs per layer cost ~40 us of which the 7-slot
        # LDS ring covers ~12, the launch seams they replace ~27 — DESIGN.md section 4.9, profiles/r06_engine_*). Costs a
        # third copy of the layer weights (13.9 GB for Llama-3-8B) when on.
        decode_engine=False,
        pin_swap_memory=True,           # host swap pool in pinned memory (falls back to pageable when the host refuses)
        # the serving loop (server/engine.py) keeps the interpreter's cyclic collector out of its busy iterations: the model's
        # long-lived objects are frozen out of the collector's working set when the loop starts, young generations are
        # collected when the loop goes idle (or every 256 busy iterations), a full collection every 4096 — what bench.py's
        # timed region does around its K steps (the benchmark measures what the server does)
        pause_gc_while_serving=True,
    )

    # Tokens per KV block the HIP kernels are built for (csrc/paged_attn.hip kBlk, kvcache.hip): one 16-token block
    # of a 128-wide head is 4 KiB = one wave-wide 16 B/lane load x 4.
    SUPPORTED_BLOCK_SIZE = 16

    def __post_init__(self):
        # The reference takes block_size as a Triton constexpr (paged_attn.py:27) and its CLI default is 16
        # (engine_config.py:37-42). Here the block is a compile-time tile; anything else is refused on the host,
        # before any memory is allocated, instead of surfacing as SWL_ERR_UNSUPPORTED from the first decode step.
        if int(self.block_size) != self.SUPPORTED_BLOCK_SIZE:
            raise ValueError(
                f"block_size={self.block_size} is not supported: the MI355X kernels are built for "
                f"{self.SUPPORTED_BLOCK_SIZE}-token KV blocks (pass --block-size {self.SUPPORTED_BLOCK_SIZE})")
        if self.dtype not in ("float16", "bfloat16"):
            raise ValueError(f"dtype must be 'float16' or 'bfloat16', got {self.dtype!r}")
        unknown = set(self.tuning or ()) - set(self.TUNING_DEFAULTS)
        if unknown:
            raise ValueError(f"unknown tuning switches {sorted(unknown)}; known: {sorted(self.TUNING_DEFAULTS)}")
        for name, default in self.TUNING_DEFAULTS.items():
            setattr(self, name, bool((self.tuning or {}).get(name, default)))

    @staticmethod
    def add_cli_args(parser: argparse.ArgumentParser):
        """Add the engine's CLI flags (same flag names as the reference, engine_config.py:25-84)."""
        g = parser.add_argument_group("swiftllm engine")
        g.add_argument("--model-path", type=str, required=True,
                       help="Directory holding config.json and the weights (no downloading)")
        g.add_argument("--use-dummy", action="store_true",
                       help="Random weights instead of loading a checkpoint (profiling)")
        g.add_argument("--block-size", type=int, default=16, help="Tokens per KV block")
        g.add_argument("--gpu-mem-utilization", type=float, default=0.97,
                       help="Fraction of HBM the weights + KV pool may occupy")
        g.add_argument("--num-cpu-blocks", type=int, default=16384,
                       help="Blocks in the host swap pool")
        g.add_argument("--max-seqs-in-block-table", type=int, default=4096,
                       help="Rows of the device block table")
        g.add_argument("--max-blocks-per-seq", type=int, default=32768,
                       help="Columns of the device block table")
        g.add_argument("--max-batch-size", type=int, default=512,
                       help="Sequences per forward, at most")
        g.add_argument("--max-tokens-in-batch", type=int, default=32768,
                       help="Tokens per forward, at most")
        g.add_argument("--dtype", type=str, default="float16", choices=["float16", "bfloat16"])
        g.add_argument("--no-fuse-qkv", dest="fuse_qkv", action="store_false")
        g.add_argument("--use-hip-graph", dest="use_hip_graph", action="store_true", default=True,
                       help="(default) replay captured hipGraphs for pure-decode steps")
        g.add_argument("--no-hip-graph", dest="use_hip_graph", action="store_false")
        g.add_argument("--no-skinny-gemm", dest="use_skinny_gemm", action="store_false")
