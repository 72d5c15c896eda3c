"""Time ggq_dequantize_f16 for every format, warm (one input / output buffer) and cold (a ring of distinct inputs and
4 distinct 90 MB outputs, > the 256 MB Infinity Cache).  hipGraph of 24 calls, median of 5 replays.
usage: [GGQ_LIB=...] python scripts/sweep_dequant.py [rows] [cols] [type ids...]; also times a plain 16-byte copy"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))
import numpy as np, torch
from ggq import lib as ggqlib, synth
from ggq.formats import WEIGHT_TYPES, GGMLType, row_bytes
N = int(sys.argv[1]) if len(sys.argv) > 1 else 11008
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
types = [GGMLType(int(a)) for a in sys.argv[3:]] or WEIGHT_TYPES
L = ggqlib.hip() if not os.environ.get('GGQ_LIB') else ggqlib._bind(ctypes.CDLL(os.environ['GGQ_LIB']), ggqlib.HIP_SYMBOLS)
vp = lambda x: ctypes.c_void_p(x.data_ptr()); st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
outs = [torch.empty((N, K), dtype=torch.float16, device="cuda") for _ in range(4)]
def timeit(f, iters=24):
    for i in range(5): f(i)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters): f(i)
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1000 / iters)
    return float(np.median(ts))
# reference point: a plain device copy of the same 90 MB (read 90 + write 90), cold ring
src = [torch.empty((N, K), dtype=torch.float16, device="cuda") for _ in range(4)]
us = timeit(lambda i: outs[i % 4].copy_(src[(i + 1) % 4]))
print(f"copy  {us:7.2f} us  {2 * N * K * 2 / us / 1e3:7.1f} GB/s (torch copy_ of {N * K * 2 >> 20} MB, cold ring)", flush=True)
us = timeit(lambda i: outs[i % 4].zero_())
print(f"fill  {us:7.2f} us  {N * K * 2 / us / 1e3:7.1f} GB/s (torch zero_ of {N * K * 2 >> 20} MB, cold ring of 4: the write-only ceiling)", flush=True)
us = timeit(lambda i: src[i % 4].sum())
print(f"read  {us:7.2f} us  {N * K * 2 / us / 1e3:7.1f} GB/s (torch sum of {N * K * 2 >> 20} MB fp16, cold ring of 4: a read-only stream)", flush=True)
del src
for t in types:
    w = torch.from_numpy(synth.random_weight(t, N, K, seed=0)).cuda()
    ring = [w] + [w.clone() for _ in range(max(1, (352 << 20) // w.numel()))]
    nbytes = N * row_bytes(t, K) + N * K * 2
    warm = timeit(lambda i: L.ggq_dequantize_f16(vp(w), vp(outs[0]), int(t), N, K, st()))
    cold = timeit(lambda i: L.ggq_dequantize_f16(vp(ring[i % len(ring)]), vp(outs[i % 4]), int(t), N, K, st()))
    print(f"{t.name:5s} warm {warm:7.2f} us {100 * nbytes / warm / 1e3 / 8000:5.1f} %   cold {cold:7.2f} us {nbytes / cold / 1e3:7.1f} GB/s {100 * nbytes / cold / 1e3 / 8000:5.1f} % of 8 TB/s", flush=True)
    del ring
