import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_gpu_kernels as tk
from swiftllm_amd import _hip
dtype, sbs, H, KVH, D = torch.bfloat16, 1024, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g = tk.gen(7)
L, layer = 2, 1
lens = [1, 15, 16, 17, 63, 64, 65, 300, 129]
nd = len(lens); G = H // KVH
q, kc, vc, bt, seq_ids = tk._paged_case(g, H, KVH, D, L, lens, dtype, layer)
q, kc, vc, bt = q.cuda(), kc.cuda(), vc.cuda(), bt.cuda()
sl = torch.tensor(lens, dtype=torch.int32, device='cuda'); sid = torch.tensor(seq_ids, dtype=torch.int32, device='cuda')
dumps = []
for rep in range(30):
    o = torch.zeros(nd, H, D, dtype=dtype, device='cuda')
    scratch = torch.zeros(nd * H * 8 * D + nd * H * 8 * 2, dtype=torch.float32, device='cuda')
    _hip.call("swl_paged_attn_decode", o.data_ptr(), q.data_ptr(), kc.data_ptr(), vc.data_ptr(), bt.data_ptr(), sid.data_ptr(),
              sl.data_ptr(), scratch.data_ptr(), D ** -0.5, nd, H, KVH, D, L, 16, layer, bt.shape[1], sbs, 1, H * D, H * D,
              _hip.dtype_code(dtype), _hip.stream())
    torch.cuda.synchronize()
    acc = scratch[: nd * H * 8 * D].view(nd, KVH, 8, G, D).cpu()
    ml = scratch[nd * H * 8 * D:].view(nd, KVH, 8, G, 2).cpu()
    dumps.append((o.cpu(), acc, ml))
o0, a0, m0 = dumps[0]
for rep, (o, a, m) in enumerate(dumps[1:], 1):
    do = (o != o0).nonzero(); da = (a != a0).nonzero(); dm = (m != m0).nonzero()
    if do.shape[0] + da.shape[0] + dm.shape[0]: print('rep', rep, 'o diffs', do.shape[0], 'acc diffs', da.shape[0], 'ml diffs', dm.shape[0])
    if da.shape[0]:
        for i in da[:4].tolist():
            print('    acc', i, float(a0[tuple(i)]), float(a[tuple(i)]))
    if dm.shape[0]:
        for i in dm[:6].tolist():
            print('    ml', i, float(m0[tuple(i)]), float(m[tuple(i)]))

print('done')
