"""96-row units against 64-row units of the same launch (tuning build: GGQ_X64_ROWS is read once per process -> two subprocesses).
usage: GGQ_LIB=scripts/_variants/libggq_x64stamp.so python scripts/dbg_x64_r3b.py [rows] [batch] [k] [patch]"""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8257
B = int(sys.argv[2]) if len(sys.argv) > 2 else 100
K = int(sys.argv[3]) if len(sys.argv) > 3 else 1280
PATCH = int(sys.argv[4]) if len(sys.argv) > 4 else 1
if os.environ.get("CHILD"):
    import torch, ctypes
    from ggq import synth
    from ggq import lib as ggqlib
    from ggq.formats import BLOCK
    from ggq.synth import _F16_FIELDS
    ggqlib._hip = ggqlib._bind(ctypes.CDLL(os.environ["GGQ_LIB"]), ggqlib.HIP_SYMBOLS)
    import util
    t = int(os.environ.get("TYPE", "12"))
    w = synth.random_weight(t, N, K, seed=B + K)
    bs, m_off = BLOCK[t][1], _F16_FIELDS[t][1]
    wb = w.reshape(N, -1, bs)
    vals = np.array([6e-8, 1.0, 1023.5, 1024.5, 65504.0, -65504.0, -3.0, 0.0, -2000.0], np.float16)
    if PATCH and m_off is not None:
        for r in range(0, N, 7):
            for b in range(wb.shape[1]):
                wb[r, b, m_off:m_off + 2] = vals[(r + 5 * b) % len(vals)].reshape(1).view(np.uint8)
    w = wb.reshape(N, -1)
    x = torch.randn((B, K), generator=torch.Generator().manual_seed(17)).float().cuda()
    y = util.gpu_mmq_x64(w, x, t, N).float().cpu().numpy()
    np.save(os.environ["CHILD"], y)
    sys.exit(0)
from collections import Counter
ys = {}
RA, RB = os.environ.get("ROWS_AB", "96,64").split(",")
for rows in (RA, RB):
    f = f"/tmp/dbg_r3b_{rows}.npy"
    subprocess.run([sys.executable, __file__] + sys.argv[1:], env=dict(os.environ, CHILD=f, GGQ_X64_ROWS=rows), check=True)
    ys[rows] = np.load(f)
d = ys[RA] != ys[RB]
print(f"{N} x {K} batch {B} patch {PATCH}: {d.sum()} of {d.size} differ")
if d.any():
    tok, row = np.nonzero(d)
    print("rows mod 96:", sorted(Counter((row % 96).tolist()).items()))
    print("rows mod 7 == 0:", int((row % 7 == 0).sum()), "of", len(row))
    print("tokens:", sorted(Counter((tok % 64).tolist()).items())[:70])
    print("units:", sorted(Counter((row // 96).tolist()).items())[:10])
    rel = np.abs(ys[RA][d] - ys[RB][d]) / (np.abs(ys[RB][d]) + 1e-9)
    print("rel diff median %.3g max %.3g" % (np.median(rel), rel.max()))
    for i in range(min(8, len(row))):
        print("  tok", tok[i], "row", row[i], ys[RA][tok[i], row[i]], ys[RB][tok[i], row[i]])
