"""Per-wave timeline of the 64 x 64 wave-tile kernel (variant built with -DGGQ_X64_STAMP=1).
usage: GGQ_LIB=scripts/_variants/libggq_X.so [COLD=1] python scripts/stamps_x64.py [type] [batch] [rows] [k]"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))
import torch, numpy as np
from ggq import lib as ggqlib, synth
t = int(sys.argv[1]) if len(sys.argv) > 1 else 12
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
N = int(sys.argv[3]) if len(sys.argv) > 3 else 11008
K = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
L = ggqlib._bind(ctypes.CDLL(os.environ["GGQ_LIB"]), ggqlib.HIP_SYMBOLS)
vp = lambda x: ctypes.c_void_p(x.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
w0 = torch.from_numpy(synth.random_weight(t, N, K, seed=0)).cuda()
cold = os.environ.get("COLD") == "1"
ws = [w0] + ([w0.clone() for _ in range((352 << 20) // w0.numel() + 1)] if cold else [])
x = torch.randn((batch, K), generator=torch.Generator().manual_seed(0)).half().cuda()
scr = torch.empty(int(L.ggq_mmq_scratch_bytes(batch, K)), dtype=torch.uint8, device="cuda")
y = torch.empty((batch, N), dtype=torch.float16, device="cuda")
L.ggq_quantize_q8_1_x64(vp(x), 1, vp(scr), batch, K, t, st())
# a back-to-back train of launches, as the bench's graph replays: the stamps are those of the last launch
for i in range(len(ws) + 40):
    assert L.ggq_mul_mat_q_x64(vp(ws[(i + 1) % len(ws)]), vp(scr), vp(y), t, 1, batch, K, N, N, 0, None, st()) == 0
torch.cuda.synchronize()
NS = 6
buf = np.zeros(2048 * 8 * 16, dtype=np.uint64)
L.ggq_debug_read_x64_stamps.restype = ctypes.c_int
L.ggq_debug_read_x64_stamps.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
assert L.ggq_debug_read_x64_stamps(buf.ctypes.data, buf.size) == 0
a = buf.reshape(-1, 16)
if os.environ.get("RAW"):
    np.save(os.environ["RAW"], a[: 8 * 520])
a = a[a[:, 0] > 0]
s = a[:, :NS].astype(np.float64); c = a[:, 8:8 + NS].astype(np.float64); hw = a[:, 7]
t00 = s[:, 0].min()
s = (s - t00) / 100.0   # 100 MHz -> us
print("unit rows", L.ggq_mmq_x64_unit_rows(t, batch, K, N) if not os.environ.get("GGQ_X64_ROWS") else os.environ["GGQ_X64_ROWS"], "K-slices", os.environ.get("GGQ_X64_KS", "rule"))
print(("cold" if cold else "warm"), "type", t, "batch", batch, N, "x", K, "waves", len(s), "kernel span (first start -> last end) %.2f us" % s[:, NS - 1].max())
names = ["start", "loop entered", "loop done", "all slices done", "partials in LDS", "end"]
for i, n in enumerate(names):
    v = s[:, i]
    print("%-20s min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % (n, v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max()))
for i in range(1, NS):
    d = s[:, i] - s[:, i - 1]
    print("%-40s min %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % (names[i - 1] + " -> " + names[i], d.min(), np.median(d), np.percentile(d, 90), d.max()))
loop_us = s[:, 2] - s[:, 1]; loop_cyc = c[:, 2] - c[:, 1]
mhz = loop_cyc / loop_us
print("shader clock over the K loop (s_memtime / wall): min %.0f  p50 %.0f  max %.0f MHz" % (mhz.min(), np.median(mhz), mhz.max()))
# how many waves shared a CU: group by (xcc, se, cu) of HW_ID
xcc = (hw >> 32) & 0xF; hwid = hw & 0xFFFFFFFF
cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 0x1; se = (hwid >> 13) & 0x7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
uniq, cnt = np.unique(key, return_counts=True)
print("CUs used", len(uniq), "waves per CU: " + ", ".join("%d x %d" % (n, (cnt == n).sum()) for n in sorted(set(cnt))))
for n in sorted(set(cnt)):
    sel = np.isin(key, uniq[cnt == n])
    print("  CUs with %2d waves: loop p50 %6.2f us  max %6.2f   end p50 %6.2f  max %6.2f   clock p50 %.0f MHz" %
          (n, np.median(loop_us[sel]), loop_us[sel].max(), np.median(s[sel, 5]), s[sel, 5].max(), np.median(mhz[sel])))
