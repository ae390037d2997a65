"""Are torch and libpgx on the same HIP runtime in this process? Prints the loaded libamdhip64 files and
whether torch.cuda.synchronize() waits for work libpgx enqueued on the null stream."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
torch.cuda.init()
from pangenomix_amd import _native, synth
from pangenomix_amd import pangenome_analysis as pa

ctx = _native.Context(0)
libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip64' in l or 'libhsa-runtime' in l})
print('\n'.join(libs))
dev = torch.device('cuda', 0)
S, n_iter = 400, 20000
row, col, G = synth.pancore_matrix(150000, S, seed=1)
np.random.seed(0)
perms = pa.draw_permutations(S, n_iter)
stride = _native.lib().pgx_bitmap_stride_words(G)
d_row, d_col = torch.from_numpy(row).to(dev), torch.from_numpy(col).to(dev)
d_bits = torch.zeros((S, stride), dtype=torch.int64, device=dev)
d_perms = torch.from_numpy(perms).to(dev)
d_pan = torch.zeros((n_iter, S), dtype=torch.int32, device=dev)
d_core = torch.zeros((n_iter, S), dtype=torch.int32, device=dev)
ws = _native.lib().pgx_pan_core_workspace_bytes(G, S, n_iter)
d_ws = torch.empty(ws, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for rep in range(3):
    ctx.presence_bitmap_dev(d_row.data_ptr(), d_col.data_ptr(), row.size, G, S, d_bits.data_ptr(), 0)
    t0 = time.perf_counter()
    ctx.pan_core_dev(d_bits.data_ptr(), G, S, d_perms.data_ptr(), n_iter, d_pan.data_ptr(), d_core.data_ptr(),
                     d_ws.data_ptr(), ws, 0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('enqueue %.2f ms, torch.cuda.synchronize %.2f ms (kernel needs ~%.0f ms)' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, 0.7 * n_iter / 1000))
ctx.close()
