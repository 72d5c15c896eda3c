"""GEMV: the fused op (ggq_mul_mat_vec_q) against quantise + ggq_mul_mat_vec_q_prequant (two launches), by shape, warm / cold.
usage: python scripts/sweep_mmvq_fused.py type rows:k [rows:k ...]"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))
import torch
from ggq import lib as ggqlib, synth
t = int(sys.argv[1]); L = ggqlib.hip()
vp = lambda x: ctypes.c_void_p(x.data_ptr()); st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(f, cold):
    for i in range(4): f(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(64): f(i if cold else 0)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / 128
for shape in sys.argv[2:]:
    N, K = (int(v) for v in shape.split(":"))
    w0 = torch.from_numpy(synth.random_weight(t, N, K, seed=0)).cuda()
    nring = max(2, (352 << 20) // w0.numel() + 2)
    ws = [w0] + [w0.clone() for _ in range(nring - 1)]
    x = torch.randn((1, K), generator=torch.Generator().manual_seed(0)).half().cuda()
    y = torch.empty((1, N), dtype=torch.float16, device="cuda"); y2 = torch.empty_like(y)
    scr = torch.empty(int(L.ggq_mmvq_scratch_bytes(K)) + 4096, dtype=torch.uint8, device="cuda")
    def fused(i): assert L.ggq_mul_mat_vec_q(vp(ws[i % nring]), vp(x), vp(y), t, 1, K, N, vp(scr), st()) == 0
    def two(i):
        assert L.ggq_quantize_q8_1(vp(x), 1, vp(scr), 1, K, st()) == 0
        assert L.ggq_mul_mat_vec_q_prequant(vp(ws[i % nring]), vp(scr), vp(y2), t, 1, K, N, st()) == 0
    a = (timeit(fused, 0), timeit(fused, 1)); b = (timeit(two, 0), timeit(two, 1))
    fused(0); two(0); torch.cuda.synchronize()
    print(f"type {t} {N:6d} x {K:5d}: fused {a[0]:6.2f} / {a[1]:6.2f}   two launches {b[0]:6.2f} / {b[1]:6.2f}   {'same bits' if torch.equal(y, y2) else 'DIFFERENT'}", flush=True)
    del ws
