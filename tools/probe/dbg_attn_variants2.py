import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_gpu_kernels as tk
from types import SimpleNamespace as NS
from swiftllm_amd.worker.kernels.linear import linear_splitk
from swiftllm_amd.worker.kernels.paged_attn import paged_attention_from_qkv_splitk
from swiftllm_amd.worker.kernels.rotary_emb import rotary_embedding_and_store_kvcache_decode_from_splitk
def run(dtype, sbs, H, KVH, D, hid, reps=30):
    g = tk.gen(H * 3 + D + hid + sbs)
    L, layer = 2, 1
    lens = [1, 15, 16, 17, 63, 64, 65, 300, 129]
    nd = len(lens)
    _, kc, vc, bt, seq_ids = tk._paged_case(g, H, KVH, D, L, lens, dtype, layer)
    n = (H + 2 * KVH) * D
    x = torch.randn(nd, hid, generator=g).to(dtype).cuda()
    wqkv = (torch.randn(n, hid, generator=g) * (hid ** -0.5)).to(dtype).cuda()
    ang = torch.rand(512, D // 2, generator=g) * 6.28
    st = tk._paged_state(lens, seq_ids, sbs, D, "cuda")
    st.position_cos, st.position_sin = torch.cos(ang).to(dtype).cuda(), torch.sin(ang).to(dtype).cuda()
    st.position_indices = torch.tensor([v - 1 for v in lens], dtype=torch.int32, device="cuda")
    mc, ec = NS(num_q_heads=H, num_kv_heads=KVH, head_dim=D, num_layers=L), NS(block_size=16)
    part = linear_splitk(x, wqkv, always=True)
    kc1, vc1, btc = kc.cuda(), vc.cuda(), bt.cuda()
    q1, _, _ = rotary_embedding_and_store_kvcache_decode_from_splitk(part, kc1, vc1, btc, mc, ec, st, layer)
    q1 = q1.clone()
    torch.cuda.synchronize()
    # fp64 reference from the pools
    G = H // KVH
    ref = torch.zeros(nd, H, D, dtype=torch.float64)
    kcc, vcc, qc, btl = kc1.cpu().double(), vc1.cpu().double(), q1.cpu().double().view(nd, H, D), btc.cpu()
    for s_, n_ in enumerate(lens):
        nb = -(-n_ // 16)
        for h in range(H):
            kk = torch.cat([kcc[btl[seq_ids[s_], b], layer, h // G] for b in range(nb)])[:n_]
            vv = torch.cat([vcc[btl[seq_ids[s_], b], layer, h // G] for b in range(nb)])[:n_]
            p_ = torch.softmax(kk @ qc[s_, h] * D ** -0.5, 0)
            ref[s_, h] = p_ @ vv
    outs = []
    for rep in range(reps):
        o1 = torch.zeros(nd, H, D, dtype=dtype, device="cuda")
        tk.K().paged_attention(q1.view(nd, H, D), kc1, vc1, btc, mc, ec, st, layer, o1)
        outs.append(o1.cpu())
    base = outs[0]
    nbad = sum(int(not torch.equal(base, o)) for o in outs)
    import collections
    classes = collections.Counter(hash(o.view(torch.int16).numpy().tobytes()) for o in outs)
    print('   distinct outputs:', len(classes), sorted(classes.values(), reverse=True)[:8], 'seqs affected:', sorted(set(int(i[0]) for o in outs for i in (base != o).nonzero().tolist())))
    print(dtype, H, KVH, D, 'sbs', sbs, 'plain kernel reps differing from rep 0:', nbad, '/', reps, ' err vs f64 of rep0:', float((base.double() - ref).abs().max()))
    for o in outs[1:]:
        if not torch.equal(base, o):
            idx = (base != o).nonzero()
            for i in idx[:5].tolist():
                a, b, r = float(base[tuple(i)]), float(o[tuple(i)]), float(ref[tuple(i)])
                print('   ', i, 'rep0', a, 'other', b, 'ref', r, 'err0', abs(a - r), 'err1', abs(b - r))
            break
run(torch.bfloat16, 1024, 32, 8, 128, 4096)
run(torch.bfloat16, 1024, 8, 4, 64, 256)
run(torch.float16, 1024, 32, 8, 128, 4096)
run(torch.bfloat16, 64, 32, 8, 128, 4096)
