"""where do two launches of the x64 kernel differ?  usage: python scripts/dbg_x64.py [rows] [batch] [k]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ggq import synth
import util
if os.environ.get("GGQ_LIB"):
    import ctypes
    from ggq import lib as ggqlib
    ggqlib._hip = ggqlib._bind(ctypes.CDLL(os.environ["GGQ_LIB"]), ggqlib.HIP_SYMBOLS)
from collections import Counter
N = int(sys.argv[1]) if len(sys.argv) > 1 else 11008
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
K = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
t = 12
w = synth.random_weight(t, N, K, seed=21)
x = torch.randn((B, K), generator=torch.Generator().manual_seed(22)).half().cuda()
ys = [util.gpu_mmq_x64(w, x, t, N).float().cpu().numpy() for _ in range(4)]
for i in range(1, 4):
    d = ys[i] != ys[0]
    print(f"launch {i} vs 0: {d.sum()} of {d.size} differ")
    if d.any():
        tok, row = np.nonzero(d)
        print("  tokens mod 64:", sorted(Counter((tok % 64).tolist()).items())[:40])
        print("  rows mod 64:", sorted(Counter((row % 64).tolist()).items())[:70])
        print("  row tiles:", sorted(Counter((row // 64).tolist()).items())[:20])
        rel = np.abs(ys[i][d] - ys[0][d]) / (np.abs(ys[0][d]) + 1e-6)
        print("  rel diff: median %.3g max %.3g" % (np.median(rel), rel.max()), " abs max", np.abs(ys[i][d] - ys[0][d]).max())
