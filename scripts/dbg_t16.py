"""Which (token, row) entries of the 16-token-tile kernel disagree with the oracle, and by what factor — the first thing to look at
when a new format's operand mapping is off.  usage: python scripts/dbg_t16.py type batch k n_rows"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import util
from ggq import synth
from oracle import oracle as O
t = int(sys.argv[1]); batch = int(sys.argv[2]); k = int(sys.argv[3]); n = int(sys.argv[4])
w = synth.random_weight(t, n, k, seed=1)
x = torch.randn((batch, k), generator=torch.Generator().manual_seed(2)).cuda()
y = util.gpu_mmq_t16(w, x, t, n).float().cpu().numpy()
ref, yabs = O.mul_mat_q(w, x.cpu().numpy(), t, n)
bad = np.abs(y - ref) > 1e-3 * np.abs(ref) + 1e-5 * yabs
np.set_printoptions(linewidth=200)
print("bad map [token, row]:"); print(bad.astype(int))
print("ratio y/ref:"); print(np.round(y / ref, 3))
