"""Debug probe (GPU): paged attention phase 1/2 vs the oracle over a grid of (D, G, len, split)."""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import eager_ops as ops
from swiftllm_amd.worker import kernels as K
NS = types.SimpleNamespace
torch.manual_seed(0)
for D in (32, 64, 128):
    for G in (1, 2):
        for lens in ([1], [16], [17], [40], [200]):
            for sbs in (64, 256):
                KVH, L, layer = 2, 1, 0
                H = KVH * G
                n = lens[0]
                nblk = -(-n // 16) + 1
                kc = torch.randn(nblk, L, KVH, 16, D).half(); vc = torch.randn(nblk, L, KVH, 16, D).half()
                bt = torch.arange(nblk, dtype=torch.int32).view(1, -1).contiguous()
                q = torch.randn(1, H, D).half()
                st = lambda dev: NS(num_decoding_seqs=1, num_prefill_seqs=0, seq_block_size=sbs,
                                    num_seq_blocks=-(-n // sbs), softmax_scale=D ** -0.5,
                                    decoding_seq_lens=torch.tensor(lens, dtype=torch.int32, device=dev),
                                    seq_ids=torch.tensor([0], dtype=torch.int32, device=dev))
                mc, ec = NS(num_q_heads=H, num_kv_heads=KVH, head_dim=D, num_layers=L), NS(block_size=16)
                eo = torch.zeros_like(q); ops.paged_attention(q, kc, vc, bt, mc, ec, st("cpu"), layer, eo)
                o = torch.zeros_like(q).cuda()
                K.paged_attention(q.cuda(), kc.cuda(), vc.cuda(), bt.cuda(), mc, ec, st("cuda"), layer, o)
                o = o.cpu().float(); d = (o - eo.float()).abs()
                nan = torch.isnan(o)
                msg = f"D={D} G={G} len={n} sbs={sbs}: nan={int(nan.sum())}/{o.numel()} maxerr={float(d[~nan].max()) if (~nan).any() else -1:.2e}"
                if nan.any():
                    idx = torch.nonzero(nan[0])
                    msg += f" nan heads={sorted(set(idx[:,0].tolist()))} dims[{int(idx[:,1].min())}..{int(idx[:,1].max())}]"
                print(msg, flush=True)
