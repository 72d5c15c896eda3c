"""x64 op (quantise + 64 x 64 wave-tile kernel) against the streamed op (its own quantiser + kernel), over shapes and batches, warm / cold.
usage: python scripts/sweep_x64.py [type=12] > profiles/r04_x64_vs_stream_*.txt"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))
import numpy as np
import torch
from ggq import lib as ggqlib, synth
t = int(sys.argv[1]) if len(sys.argv) > 1 else 12
L = ggqlib.hip() if not os.environ.get("GGQ_LIB") else ggqlib._bind(ctypes.CDLL(os.environ["GGQ_LIB"]), ggqlib.HIP_SYMBOLS)
vp = lambda x: ctypes.c_void_p(x.data_ptr()); st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = [(2048, 4096), (4096, 4096), (6144, 4096), (8192, 4096), (11008, 4096), (14336, 4096), (28672, 4096), (4096, 11008), (3584, 8192), (8192, 8192), (28672, 8192)]
BATCHES = [33, 48, 64, 96, 128, 192, 256, 512, 1024]
if len(sys.argv) > 2:
    BATCHES = [int(b) for b in sys.argv[2].split(",")]

def timeit(fn, iters):
    for i in range(2): fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters): fn(i)
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / iters)
    return float(np.median(ts))

print(f"# type {t}: op us warm / cold, streamed kernel (route 3) | x64; ratio = streamed / x64 (cold)")
for (N, K) in SHAPES:
    w0 = torch.from_numpy(synth.random_weight(t, N, K, seed=0)).cuda()
    ring = [w0] + [w0.clone() for _ in range(max(1, (352 << 20) // w0.numel()))]
    for b in BATCHES:
        x = torch.randn((b, K), generator=torch.Generator().manual_seed(0)).half().cuda()
        scr = torch.empty(int(L.ggq_mmq_scratch_bytes(b, K)), dtype=torch.uint8, device="cuda")
        y = torch.empty((b, N), dtype=torch.float16, device="cuda")
        iters = 16 if b * N * K > 3e10 else 48
        def x64_op(i, r=ring):
            L.ggq_quantize_q8_1_x64(vp(x), 1, vp(scr), b, K, t, st())
            L.ggq_mul_mat_q_x64(vp(r[i % len(r)]), vp(scr), vp(y), t, 1, b, K, N, N, 0, None, st())
        def old_op(i, r=ring):   # the streamed kernel with its own quantiser (what the route takes below the x64 threshold), whatever the route says now
            L.ggq_quantize_q8_1_tiled(vp(x), 1, vp(scr), b, K, t, st())
            L.ggq_mul_mat_q_pretiled(vp(r[i % len(r)]), vp(scr), vp(y), t, 1, b, K, N, N, st())
        if os.environ.get("X64_ONLY"):
            xw, xc = timeit(lambda i: x64_op(i, [w0]), iters), timeit(x64_op, iters)
            print(f"{N:6d} x {K:5d} batch {b:5d}: units {-(-N // 64) * -(-b // 64):5d} x64 {xw:8.1f} / {xc:8.1f}", flush=True)
            continue
        ow, oc = timeit(lambda i: old_op(i, [w0]), iters), timeit(old_op, iters)
        xw, xc = timeit(lambda i: x64_op(i, [w0]), iters), timeit(x64_op, iters)
        print(f"{N:6d} x {K:5d} batch {b:5d}: route 3 {ow:8.1f} / {oc:8.1f} | {xw:8.1f} / {xc:8.1f}   ratio {oc / xc:5.2f} (warm {ow / xw:5.2f})", flush=True)
    del ring, w0
