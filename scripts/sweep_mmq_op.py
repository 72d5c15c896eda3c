"""Time the fused op ggq_mul_mat_q (hipGraph of 50 calls). usage: [GGQ_LIB=..] python scripts/sweep_mmq_op.py type batch rows"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))
import torch
from ggq import lib as ggqlib, synth
t, batch, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]); K = int(os.environ.get("K", 4096))
L = ggqlib.hip() if not os.environ.get("GGQ_LIB") else ggqlib._bind(ctypes.CDLL(os.environ["GGQ_LIB"]), ggqlib.HIP_SYMBOLS)
vp = lambda x: ctypes.c_void_p(x.data_ptr()); st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
w = torch.from_numpy(synth.random_weight(t, N, K, seed=0)).cuda()
x = torch.randn((batch, K), generator=torch.Generator().manual_seed(0)).half().cuda()
scr = torch.empty(int(L.ggq_mmq_scratch_bytes(batch, K)), dtype=torch.uint8, device="cuda")
y = torch.empty((batch, N), dtype=torch.float16, device="cuda")
f = lambda: L.ggq_mul_mat_q(vp(w), vp(x), vp(y), t, 1, batch, K, N, vp(scr), st())
for _ in range(10): f()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(50): f()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(4): g.replay()
e1.record(); torch.cuda.synchronize()
print(f"type {t} batch {batch} rows {N}: {e0.elapsed_time(e1) * 1000 / 200:.2f} us per op", flush=True)
