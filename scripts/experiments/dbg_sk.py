"""debug: stream-K kernel vs the reference-layout kernel on a tiny shape; prints where they differ"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))
import torch, numpy as np
from ggq import lib as ggqlib, synth
t = int(sys.argv[1]); batch = int(sys.argv[2]); N = int(sys.argv[3]); K = int(sys.argv[4])
L = ggqlib.hip()
vp = lambda x: ctypes.c_void_p(x.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
w = torch.from_numpy(synth.random_weight(t, N, K, seed=0)).cuda()
x = torch.randn((batch, K), generator=torch.Generator().manual_seed(0)).half().cuda()
scr = torch.zeros(int(L.ggq_mmq_scratch_bytes(batch, K)) + 4096, dtype=torch.uint8, device="cuda")
scr0 = torch.zeros_like(scr)
y0 = torch.zeros((batch, N), dtype=torch.float16, device="cuda"); y1 = torch.zeros_like(y0)
L.ggq_quantize_q8_1_mmq(vp(x), 1, vp(scr0), batch, K, t, st())
L.ggq_mul_mat_q_prequant(vp(w), vp(scr0), vp(y0), t, 1, batch, K, N, N, st())
L.ggq_quantize_q8_1_tiled(vp(x), 1, vp(scr), batch, K, t, st())
rc = L.ggq_mul_mat_q_pretiled(vp(w), vp(scr), vp(y1), t, 1, batch, K, N, N, st())
torch.cuda.synchronize()
d = (y0.float() - y1.float()).abs().cpu().numpy()
print("rc", rc, "max diff", d.max(), "max|y|", y0.float().abs().max().item())
bad = d > 0.05 * (1 + np.abs(y0.float().cpu().numpy()))
print("bad fraction", bad.mean())
print("bad by token (first 64):", bad.mean(axis=1)[:64].round(2))
print("bad by row   (first 64):", bad.mean(axis=0)[:64].round(2))
print("y0[0,:8]", y0[0, :8].tolist()); print("y1[0,:8]", y1[0, :8].tolist())
print("ratio", (y1[0, :8].float() / y0[0, :8].float()).tolist())
import numpy as np
yb = y1.float().cpu().numpy(); ya = y0.float().cpu().numpy()
badm = ~np.isfinite(yb) | (np.abs(yb - ya) > 0.05 * (1 + np.abs(ya)))
rows = np.where(badm.any(axis=0))[0]; toks = np.where(badm.any(axis=1))[0]
print("bad rows:", rows[:40], "... count", len(rows)); print("bad tokens:", toks[:70], "count", len(toks))
if len(rows):
    r0 = rows[0]; print("row", r0, "unit row tile", r0 // 32, "values y1:", yb[toks[:6], r0], "y0:", ya[toks[:6], r0])
    tiles = sorted(set((int(r) // 32) for r in rows)); print("bad row tiles:", tiles[:40], "n", len(tiles))
for rep in range(3):
    y1.zero_()
    L.ggq_mul_mat_q_pretiled(vp(w), vp(scr), vp(y1), t, 1, batch, K, N, N, st()); torch.cuda.synchronize()
    yb = y1.float().cpu().numpy()
    badm = ~np.isfinite(yb) | (np.abs(yb - ya) > 0.05 * (1 + np.abs(ya)))
    print("relaunch", rep, "bad elements", int(badm.sum()), "bad row tiles", sorted(set(int(r) // 32 for r in np.where(badm.any(axis=0))[0]))[:20])
print("=== detail")
yb = y1.float().cpu().numpy()
badm = ~np.isfinite(yb) | (np.abs(yb - ya) > 0.05 * (1 + np.abs(ya)))
tt, rr = np.where(badm)
import collections
cnt = collections.Counter(zip((rr // 32).tolist(), (tt // 64).tolist()))
print("bad (row tile, token tile(64)) units:", len(cnt), list(cnt.items())[:12])
r0 = rr[0]
print("row", r0, "bad tokens in that row:", np.where(badm[:, r0])[0])
print("y1", yb[np.where(badm[:, r0])[0][:8], r0]); print("y0", ya[np.where(badm[:, r0])[0][:8], r0])
u_rt, u_tt = r0 // 32, tt[0] // 64
sub = badm[u_tt * 64:(u_tt + 1) * 64, u_rt * 32:(u_rt + 1) * 32]
print("unit", u_rt, u_tt, "bad map (tokens x rows):"); 
for i in range(64): print("".join("X" if b else "." for b in sub[i]))
