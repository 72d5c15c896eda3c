"""Time ggq_mul_mat_vec_q (HIP events around a hipGraph of 50 launches, 4 replays).
usage: [GGQ_LIB=...] [COLD=1] python scripts/sweep_mmvq.py type [rows...]   (COLD=1: consecutive launches cycle 16 weight tensors)"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))
import torch
from ggq import lib as ggqlib, synth
t = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rows_list = [int(a) for a in sys.argv[2:]] or [11008]
K = int(os.environ.get("K", 4096))
L = ggqlib.hip() if not os.environ.get("GGQ_LIB") else ggqlib._bind(ctypes.CDLL(os.environ["GGQ_LIB"]), ggqlib.HIP_SYMBOLS)
vp = lambda x: ctypes.c_void_p(x.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
w = torch.from_numpy(synth.random_weight(t, max(rows_list), K, seed=0)).cuda()
COLD = os.environ.get("COLD") == "1"
ws = [w] + ([w.clone() for _ in range(15)] if COLD else [])
x = torch.randn((1, K), generator=torch.Generator().manual_seed(0)).half().cuda()
scr = torch.empty(int(L.ggq_mmvq_scratch_bytes(K)) + 64, dtype=torch.uint8, device="cuda")
for N in rows_list:
    y = torch.empty((1, N), dtype=torch.float16, device="cuda")
    f = lambda i=0: L.ggq_mul_mat_vec_q(vp(ws[i % len(ws)]), vp(x), vp(y), t, 1, K, N, vp(scr), st())
    for _ in range(20): f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(50): f(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 200
    print(f"type {t} K {K} rows {N} {'cold' if COLD else 'warm'}: {us:.2f} us", flush=True)
