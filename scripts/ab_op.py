"""op = ggq_mul_mat_q (as routed) warm / cold for a list of (type, rows, k, batch), graph-timed; GGQ_LIB picks the library (A/B of two builds on one box).
usage: [GGQ_LIB=...] python scripts/ab_op.py "14 11008 4096 128" "10 11008 4096 128" ..."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))
import numpy as np
import torch
from ggq import lib as ggqlib, synth
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
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / iters)
    return float(np.median(ts))

for spec in sys.argv[1:]:
    t, N, K, b = (int(v) for v in spec.split())
    w0 = torch.from_numpy(synth.random_weight(t, N, K, seed=0)).cuda()
    ring = [w0] + [w0.clone() for _ in range(max(1, (352 << 20) // w0.numel()))]
    x = torch.randn((b, K), generator=torch.Generator().manual_seed(0)).half().cuda()
    scr = torch.empty(int(L.ggq_mmq_scratch_bytes(b, K)), dtype=torch.uint8, device="cuda")
    y = torch.empty((b, N), dtype=torch.float16, device="cuda")
    def op(i, r=ring):
        assert L.ggq_mul_mat_q(vp(r[i % len(r)]), vp(x), vp(y), t, 1, b, K, N, vp(scr), st()) == 0
    print(f"type {t:2d} {N:6d} x {K:5d} batch {b:4d}: route {L.ggq_mmq_route(t, b, K, N)}  op {timeit(lambda i: op(i, [w0]), 48):7.1f} / {timeit(op, 48):7.1f} us warm / cold", flush=True)
    del ring, w0
