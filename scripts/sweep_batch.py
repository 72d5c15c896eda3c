"""ggq_mul_mat_q (the op: quantise + matmul) across small batches, warm and cold (ring of 16 weight tensors), graph-timed.
usage: [GGQ_LIB=...] python scripts/sweep_batch.py [type] [rows] [batches...]"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))
import torch
from ggq import lib as ggqlib, synth
t = int(sys.argv[1]) if len(sys.argv) > 1 else 12
N = int(sys.argv[2]) if len(sys.argv) > 2 else 11008
batches = [int(a) for a in sys.argv[3:]] or [1, 2, 4, 5, 8, 16, 32]
K = int(os.environ.get("K", 4096))
L = ggqlib.hip() if not os.environ.get("GGQ_LIB") else ggqlib._bind(ctypes.CDLL(os.environ["GGQ_LIB"]), ggqlib.HIP_SYMBOLS)
vp = lambda x: ctypes.c_void_p(x.data_ptr()); st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
w0 = torch.from_numpy(synth.random_weight(t, N, K, seed=0)).cuda()
ws = [w0] + [w0.clone() for _ in range(15)]
for b in batches:
    x = torch.randn((b, K), generator=torch.Generator().manual_seed(0)).half().cuda()
    y = torch.empty((b, N), dtype=torch.float16, device="cuda")
    scr = (torch.zeros if os.environ.get("ZERO_SCRATCH") == "1" else torch.empty)(max(int(L.ggq_mmq_scratch_bytes(b, K)), int(L.ggq_mmvq_scratch_bytes(K))) + 4096, dtype=torch.uint8, device="cuda")
    def f(i):
        w = ws[i % len(ws)]
        if b == 1: return L.ggq_mul_mat_vec_q(vp(w), vp(x), vp(y), t, 1, K, N, vp(scr), st())
        return L.ggq_mul_mat_q(vp(w), vp(x), vp(y), t, 1, b, K, N, vp(scr), st())
    out = []
    for cold in (0, 1):
        g = torch.cuda.CUDAGraph()
        for i in range(4): assert f(i) == 0
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for i in range(64): f(i if cold else 0)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1000 / 128)
    print(f"type {t} rows {N} batch {b:3d}: op warm {out[0]:6.2f} us  cold {out[1]:6.2f} us", flush=True)
