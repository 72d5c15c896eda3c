"""The 64 x 64 wave-tile kernel (mmq_x64.hip) against the streamed kernel and dequantise + rocBLAS, kernel and op, warm and cold.
usage: python scripts/bench_x64.py [type=12] [rows=11008] [k=4096] [batches=64,128,256,512,2048,4096]"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))
import numpy as np
import torch
from ggq import lib as ggqlib, synth
t = int(sys.argv[1]) if len(sys.argv) > 1 else 12
N = int(sys.argv[2]) if len(sys.argv) > 2 else 11008
K = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
batches = [int(b) for b in (sys.argv[4] if len(sys.argv) > 4 else "64,128,256,512,2048,4096").split(",")]
L = ggqlib.hip() if not os.environ.get("GGQ_LIB") else ggqlib._bind(ctypes.CDLL(os.environ["GGQ_LIB"]), ggqlib.HIP_SYMBOLS)
QUICK = os.environ.get("QUICK")
vp = lambda x: ctypes.c_void_p(x.data_ptr()); st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
w0 = torch.from_numpy(synth.random_weight(t, N, K, seed=0)).cuda()
ring = [w0] + [w0.clone() for _ in range(max(1, (352 << 20) // w0.numel()))]
wd = torch.empty((N, K), dtype=torch.float16, device="cuda")

def timeit(fn, iters, graph=True):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    run = None
    if graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(iters): fn(i)
        run = g.replay
    else:
        def run():
            for i in range(iters): fn(i)
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / iters)
    return float(np.median(ts))

for b in batches:
    x = torch.randn((b, K), generator=torch.Generator().manual_seed(0)).half().cuda()
    scr = torch.empty(int(L.ggq_mmq_scratch_bytes(b, K)), dtype=torch.uint8, device="cuda")
    y = torch.empty((b, N), dtype=torch.float16, device="cuda")
    iters = 8 if b >= 2048 else (32 if b >= 512 else 104)
    ops = 2.0 * b * N * K
    def x64_op(i, r=ring):
        L.ggq_quantize_q8_1_x64(vp(x), 1, vp(scr), b, K, t, st())
        assert L.ggq_mul_mat_q_x64(vp(r[i % len(r)]), vp(scr), vp(y), t, 1, b, K, N, N, 0, None, st()) == 0
    def x64_k(i, r=ring):
        L.ggq_mul_mat_q_x64(vp(r[i % len(r)]), vp(scr), vp(y), t, 1, b, K, N, N, 0, None, st())
    def old_op(i, r=ring):
        assert L.ggq_mul_mat_q(vp(r[i % len(r)]), vp(x), vp(y), t, 1, b, K, N, vp(scr), st()) == 0
    def deq(i):
        L.ggq_dequantize_f16(vp(w0), vp(wd), t, N, K, st())
        torch.matmul(x, wd.t(), out=y)
    res = {}
    L.ggq_quantize_q8_1_x64(vp(x), 1, vp(scr), b, K, t, st())
    res["x64 kernel warm"] = timeit(lambda i: x64_k(i, [w0]), iters)
    if QUICK:
        print(f"type {t} {N}x{K} batch {b}: x64 kernel warm {res['x64 kernel warm']:.1f} us = {ops / res['x64 kernel warm'] / 1e6:.0f} TOP/s", flush=True)
        continue
    res["x64 kernel cold"] = timeit(x64_k, iters)
    res["x64 op warm"] = timeit(lambda i: x64_op(i, [w0]), iters)
    res["x64 op cold"] = timeit(x64_op, iters)
    res["old op warm"] = timeit(lambda i: old_op(i, [w0]), iters)
    res["old op cold"] = timeit(old_op, iters)
    res["dequant+rocblas"] = timeit(deq, iters, graph=False)
    print(f"type {t} {N}x{K} batch {b}: " + "  ".join(f"{k} {v:.1f}" for k, v in res.items()) +
          f"  | x64 kernel {ops / res['x64 kernel warm'] / 1e6:.0f} TOP/s warm", flush=True)
