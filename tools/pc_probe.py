import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from pangenomix_amd import _native, cluster, synth
from pangenomix_amd import pangenome_analysis as pa
dev = torch.device('cuda', 0)
ctx = _native.Context(0)
stream = torch.cuda.current_stream().cuda_stream
S = 400
row, col, G = synth.pancore_matrix(150000, S, seed=1)
n_iter = 1000
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
pset = synth.protein_set('cfg-2s')
res, off, n_raw = pset.nr_arrays()
params = cluster.params_from_cdhit_args({'-n': 5, '-c': 0.8})
d_res = torch.from_numpy(res.copy()).to(dev); d_off = torch.from_numpy(off.view(np.int64)).to(dev)
def pc(tag):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.presence_bitmap_dev(d_row.data_ptr(), d_col.data_ptr(), row.size, G, S, d_bits.data_ptr(), stream)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ctx.pan_core_dev(d_bits.data_ptr(), G, S, d_perms.data_ptr(), n_iter, d_pan.data_ptr(), d_core.data_ptr(), d_ws.data_ptr(), ws_bytes, stream)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print('%s: bitmap %.3f ms, pan_core %.3f ms' % (tag, (t1-t0)*1e3, (t2-t1)*1e3), flush=True)
pc('cold'); pc('warm'); pc('warm')
for i in range(2):
    ctx.cluster_greedy_dev(d_res.data_ptr(), d_off.data_ptr(), off.size-1, res.size, params, stream)
    torch.cuda.synchronize()
    pc('after cluster %d' % i); pc('again')
ctx.profile(True)
pc('profiled'); pc('profiled')
print(ctx.profile_read())
