"""Fused GEMV (ggq_mul_mat_vec_q) against the two-launch form (ggq_quantize_q8_1 + ggq_mul_mat_vec_q_prequant), graph-timed warm / cold.
usage: python scripts/sweep_mmvq_two_launch.py rows k types..."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))
import torch
from ggq import lib as ggqlib, synth
N, K = int(sys.argv[1]), int(sys.argv[2]); types = [int(a) for a in sys.argv[3:]] or [12]
L = ggqlib.hip()
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
for t in types:
    w0 = torch.from_numpy(synth.random_weight(t, N, K, seed=0)).cuda()
    nring = max(2, (352 << 20) // w0.numel() + 2)
    ws = [w0] + [w0.clone() for _ in range(nring - 1)]
    x = torch.randn((1, K), generator=torch.Generator().manual_seed(0)).half().cuda()
    y = torch.empty((1, N), dtype=torch.float16, device="cuda")
    scr = torch.empty(int(L.ggq_mmvq_scratch_bytes(K)) + 4096, dtype=torch.uint8, device="cuda")
    def fused(i): assert L.ggq_mul_mat_vec_q(vp(ws[i % nring]), vp(x), vp(y), t, 1, K, N, vp(scr), st()) == 0
    def two(i):
        assert L.ggq_quantize_q8_1(vp(x), 1, vp(scr), 1, K, st()) == 0
        assert L.ggq_mul_mat_vec_q_prequant(vp(ws[i % nring]), vp(scr), vp(y), t, 1, K, N, st()) == 0
    print(f"type {t} {N}x{K}: fused {timeit(fused, 0):6.2f} / {timeit(fused, 1):6.2f}   quantise + prequant {timeit(two, 0):6.2f} / {timeit(two, 1):6.2f}  (us warm / cold)", flush=True)
