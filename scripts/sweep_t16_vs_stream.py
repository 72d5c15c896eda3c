"""16-token-tile op against the streamed op (each = its quantiser + its kernel) over matrix shapes, warm / cold, graph-timed.
usage: python scripts/sweep_t16_vs_stream.py type batch rows:k [rows:k ...]"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))
import torch
from ggq import lib as ggqlib, synth
t, b = int(sys.argv[1]), int(sys.argv[2])
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

for shape in sys.argv[3:]:
    N, K = (int(v) for v in shape.split(":"))
    w0 = torch.from_numpy(synth.random_weight(t, N, K, seed=0)).cuda()
    nring = max(2, (352 << 20) // w0.numel() + 2)
    ws = [w0] + [w0.clone() for _ in range(nring - 1)]
    x = torch.randn((b, K), generator=torch.Generator().manual_seed(0)).half().cuda()
    y = torch.empty((b, N), dtype=torch.float16, device="cuda")
    scr = torch.empty(int(L.ggq_mmq_scratch_bytes(b + 32, K)) + 4096, dtype=torch.uint8, device="cuda")
    def op_t16(i):
        assert L.ggq_quantize_q8_1_t16(vp(x), 1, vp(scr), b, K, t, st()) == 0
        assert L.ggq_mul_mat_q_t16(vp(ws[i % nring]), vp(scr), vp(y), t, 1, b, K, N, N, 0, None, st()) == 0
    def op_stream(i):
        assert L.ggq_quantize_q8_1_tiled(vp(x), 1, vp(scr), b, K, t, st()) == 0
        assert L.ggq_mul_mat_q_pretiled(vp(ws[i % nring]), vp(scr), vp(y), t, 1, b, K, N, N, st()) == 0
    a = (timeit(op_t16, 0), timeit(op_t16, 1)); s = (timeit(op_stream, 0), timeit(op_stream, 1))
    route = L.ggq_mmq_route(t, b, K, N)
    print(f"type {t} batch {b} {N:6d} x {K:5d}: t16 {a[0]:6.2f} / {a[1]:6.2f}   streamed {s[0]:6.2f} / {s[1]:6.2f}   (routed: {'t16' if route == 4 else 'streamed' if route == 3 else route})", flush=True)
    del ws
