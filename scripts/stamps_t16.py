"""Per-wave timeline of the 16-token-tile kernel (variant built with -DGGQ_T16_STAMP=1).
usage: GGQ_LIB=scripts/_variants/libggq_X.so [COLD=1] python scripts/stamps_t16.py [type] [batch] [rows] [k]"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))
import torch, numpy as np
from ggq import lib as ggqlib, synth
t = int(sys.argv[1]) if len(sys.argv) > 1 else 12
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N = int(sys.argv[3]) if len(sys.argv) > 3 else 11008
K = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
L = ggqlib._bind(ctypes.CDLL(os.environ["GGQ_LIB"]), ggqlib.HIP_SYMBOLS)
vp = lambda x: ctypes.c_void_p(x.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
w0 = torch.from_numpy(synth.random_weight(t, N, K, seed=0)).cuda()
cold = os.environ.get("COLD") == "1"
ws = [w0] + ([w0.clone() for _ in range((352 << 20) // w0.numel() + 1)] if cold else [])
x = torch.randn((batch, K), generator=torch.Generator().manual_seed(0)).half().cuda()
scr = torch.empty(int(L.ggq_mmq_scratch_bytes(batch + 32, K)) + 4096, dtype=torch.uint8, device="cuda")
y = torch.empty((batch, N), dtype=torch.float16, device="cuda")
L.ggq_quantize_q8_1_t16(vp(x), 1, vp(scr), batch, K, t, st())
for i in range(len(ws) + 3):
    L.ggq_mul_mat_q_t16(vp(ws[(i + 1) % len(ws)]), vp(scr), vp(y), t, 1, batch, K, N, N, 0, None, st())
torch.cuda.synchronize()
NS = 7
buf = np.zeros(4096 * 16 * 8, dtype=np.uint64)
L.ggq_debug_read_t16_stamps.restype = ctypes.c_int
L.ggq_debug_read_t16_stamps.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
assert L.ggq_debug_read_t16_stamps(buf.ctypes.data, buf.size) == 0
s = buf.reshape(-1, 8)[:, :NS].astype(np.float64)
s = s[s[:, 0] > 0]
t00 = s[:, 0].min()
s = (s - t00) / 100.0   # 100 MHz -> us
print(("cold" if cold else "warm"), "waves", len(s), "kernel span (first start -> last end) %.2f us" % s[:, NS - 1].max())
names = ["start", "requests issued", "unit 0 landed", "unit 0 computed", "K loop done", "partials published", "end"]
for i, n in enumerate(names):
    c = s[:, i]
    print("%-20s min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % (n, c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max()))
for i in range(1, NS):
    d = s[:, i] - s[:, i - 1]
    print("%-40s min %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % (names[i - 1] + " -> " + names[i], d.min(), np.median(d), np.percentile(d, 90), d.max()))
