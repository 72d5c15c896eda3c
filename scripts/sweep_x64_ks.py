"""x64 op (quantise + kernel) by K-slice count, on a -DGGQ_TUNING build (GGQ_LIB) where GGQ_X64_KS forces 8 / 4 / 2 / 1 slices per 64-row unit.
One process per setting (the override is read once).  usage: GGQ_LIB=... GGQ_X64_KS=2 python scripts/sweep_x64_ks.py TYPE "N K" ... -- B1,B2,..."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))
import numpy as np
import torch
from ggq import lib as ggqlib, synth
t = int(sys.argv[1])
sep = sys.argv.index("--")
SHAPES = [tuple(int(v) for v in s.split()) for s in sys.argv[2:sep]]
BATCHES = [int(b) for b in sys.argv[sep + 1].split(",")]
L = ggqlib.hip() if not os.environ.get("GGQ_LIB") else ggqlib._bind(ctypes.CDLL(os.environ["GGQ_LIB"]), ggqlib.HIP_SYMBOLS)
vp = lambda x: ctypes.c_void_p(x.data_ptr()); st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

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

print(f"# type {t} GGQ_X64_KS={os.environ.get('GGQ_X64_KS')} GGQ_X64_ROWS={os.environ.get('GGQ_X64_ROWS')}: x64 op us warm / cold; checksum of y")
for (N, K) in SHAPES:
    w0 = torch.from_numpy(synth.random_weight(t, N, K, seed=0)).cuda()
    ring = [w0] + [w0.clone() for _ in range(max(1, (352 << 20) // w0.numel()))]
    for b in BATCHES:
        x = torch.randn((b, K), generator=torch.Generator().manual_seed(0)).half().cuda()
        scr = torch.empty(int(L.ggq_mmq_scratch_bytes(b, K)), dtype=torch.uint8, device="cuda")
        y = torch.empty((b, N), dtype=torch.float16, device="cuda")
        iters = 16 if b * N * K > 3e10 else 48
        def x64_op(i, r=ring):
            assert L.ggq_quantize_q8_1_x64(vp(x), 1, vp(scr), b, K, t, st()) == 0
            assert L.ggq_mul_mat_q_x64(vp(r[i % len(r)]), vp(scr), vp(y), t, 1, b, K, N, N, 0, None, st()) == 0
        xw, xc = timeit(lambda i: x64_op(i, [w0]), iters), timeit(x64_op, iters)
        x64_op(0, [w0]); torch.cuda.synchronize()
        yf = y.float()
        print(f"{N:6d} x {K:5d} batch {b:5d}: units64 {-(-N // 64) * -(-b // 64):6d}  x64 {xw:8.1f} / {xc:8.1f}   sum {yf.sum().item():.6e} abs {yf.abs().sum().item():.6e}", flush=True)
    del ring, w0
