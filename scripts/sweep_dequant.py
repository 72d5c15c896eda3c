"""Time ggq_dequantize_f16 for every format (hipGraph of 20 calls). usage: python scripts/sweep_dequant.py [rows] [cols]"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))
import torch
from ggq import lib as ggqlib, synth
from ggq.formats import WEIGHT_TYPES, row_bytes
N = int(sys.argv[1]) if len(sys.argv) > 1 else 11008
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
L = ggqlib.hip()
vp = lambda x: ctypes.c_void_p(x.data_ptr()); st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
out = torch.empty((N, K), dtype=torch.float16, device="cuda")
for t in WEIGHT_TYPES:
    w = torch.from_numpy(synth.random_weight(t, N, K, seed=0)).cuda()
    f = lambda: L.ggq_dequantize_f16(vp(w), vp(out), int(t), N, K, st())
    for _ in range(5): f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [g.replay() for _ in range(5)]; e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 100
    nbytes = N * row_bytes(t, K) + N * K * 2
    print(f"{t.name:5s} {us:7.2f} us  {nbytes / us / 1e3:7.1f} GB/s  {100 * nbytes / us / 1e3 / 8000:5.1f} % of 8 TB/s", flush=True)
