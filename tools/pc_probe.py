"""Time the pan/core kernels alone (tuning aid): python tools/pc_probe.py [G S n_iter]"""
import sys, time
sys.path.insert(0, __file__.rsplit('/tools/', 1)[0])
import numpy as np, torch
from pangenomix_amd import _native, synth
from pangenomix_amd import pangenome_analysis as pa
G0, S, n_iter = (int(x) for x in (sys.argv[1:4] + ['150000', '400', '1000'][len(sys.argv) - 1:]))
dev = torch.device('cuda', 0)
ctx = _native.Context(0)
stream = torch.cuda.current_stream().cuda_stream
row, col, G = synth.pancore_matrix(G0, S, seed=1)
np.random.seed(0)
perms = pa.draw_permutations(S, n_iter)
stride = _native.lib().pgx_bitmap_stride_words(G)
d_row, d_col = torch.from_numpy(row).to(dev), torch.from_numpy(col).to(dev)
d_bits = torch.zeros((S, stride), dtype=torch.int64, device=dev)
d_perms = torch.from_numpy(perms).to(dev)
d_pan = torch.empty((n_iter, S), dtype=torch.int32, device=dev)
d_core = torch.empty((n_iter, S), dtype=torch.int32, device=dev)
ws_bytes = _native.lib().pgx_pan_core_workspace_bytes(G, S, n_iter)
d_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
ctx.presence_bitmap_dev(d_row.data_ptr(), d_col.data_ptr(), row.size, G, S, d_bits.data_ptr(), stream)
ctx.profile(True)
for _ in range(3):
    ctx.pan_core_dev(d_bits.data_ptr(), G, S, d_perms.data_ptr(), n_iter, d_pan.data_ptr(), d_core.data_ptr(), d_ws.data_ptr(), ws_bytes, stream)
torch.cuda.synchronize(); ctx.profile_reset()
for _ in range(10):
    ctx.pan_core_dev(d_bits.data_ptr(), G, S, d_perms.data_ptr(), n_iter, d_pan.data_ptr(), d_core.data_ptr(), d_ws.data_ptr(), ws_bytes, stream)
torch.cuda.synchronize()
prof = ctx.profile_read()
ms = prof['pan_core_sweep_kernel'][0] / prof['pan_core_sweep_kernel'][1]
alg = n_iter * (S * ((G + 63) // 64) * 8 + 2 * S * 4) + n_iter * S * 4
print('G %d S %d iters %d: sweep %.4f ms (%.1f GB/s algorithmic), reduce %.4f ms' % (G, S, n_iter, ms, alg / ms / 1e6, prof['pan_core_reduce_kernel'][0] / 10))
import oracle
opan, ocore = oracle.pan_core(row, col, None, G, S, perms[:2])
assert np.array_equal(d_pan.cpu().numpy()[:2], opan) and np.array_equal(d_core.cpu().numpy()[:2], ocore)
print('parity ok')
