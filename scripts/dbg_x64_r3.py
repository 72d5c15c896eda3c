"""96-row units of the x64 kernel against the oracle (scripts only: the oracle is the checker): which rows / tokens are off?
usage: python scripts/dbg_x64_r3.py [rows] [batch] [k]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ggq import synth
from ggq import lib as ggqlib
import util
from oracle import oracle as O
from collections import Counter
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8230
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
K = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
t = 12
print("unit rows", ggqlib.hip().ggq_mmq_x64_unit_rows(t, B, K, N))
w = synth.random_weight(t, N, K, seed=21)
x = torch.randn((B, K), generator=torch.Generator().manual_seed(22)).half().cuda()
y = util.gpu_mmq_x64(w, x, t, N).float().cpu().numpy()
ref, yabs = O.mul_mat_q(w, x.float().cpu().numpy(), t, N)
err = np.abs(y - ref) / (yabs + 1e-6)
bad = err > 2e-3
print("bad", bad.sum(), "of", bad.size, "nan", np.isnan(y).sum())
tok, row = np.nonzero(bad | np.isnan(y))
print("rows mod 96:", sorted(Counter((row % 96).tolist()).items()))
print("tokens mod 64:", sorted(Counter((tok % 64).tolist()).items()))
print("units:", sorted(Counter((row // 96).tolist()).items())[:12])
for r in (0, 31, 32, 63, 64, 65, 80, 95):
    print("row", r, "y", y[:4, r], "ref", ref[:4, r])
# ratio statistics on the one-row-tile rows
sel = (np.arange(N) % 96) >= 64
print("one-tile rows: median |y/ref|", np.nanmedian(np.abs(y[:, sel] / ref[:, sel])), " two-tile rows:", np.nanmedian(np.abs(y[:, ~sel] / ref[:, ~sel])))
