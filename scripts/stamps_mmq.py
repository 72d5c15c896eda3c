"""Per-wave timeline of the register-direct MMQ kernel (variant built with -DGGQ_ABL=32[+..]).
usage: GGQ_LIB=scripts/_variants/libggq_X.so [TILED=1] GGQ_MMQ_REG=2 python scripts/stamps_mmq.py [type] [batch] [rows]"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))
import torch, numpy as np
from ggq import lib as ggqlib, synth
t = int(sys.argv[1]) if len(sys.argv) > 1 else 12
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
N = int(sys.argv[3]) if len(sys.argv) > 3 else 11008
K = 4096
L = ggqlib._bind(ctypes.CDLL(os.environ["GGQ_LIB"]), ggqlib.HIP_SYMBOLS)
vp = lambda x: ctypes.c_void_p(x.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
w = torch.from_numpy(synth.random_weight(t, N, K, seed=0)).cuda()
x = torch.randn((batch, K), generator=torch.Generator().manual_seed(0)).half().cuda()
scr = torch.empty(int(L.ggq_mmq_scratch_bytes(batch + 32, K)) + 4096, dtype=torch.uint8, device="cuda")
y = torch.empty((batch, N), dtype=torch.float16, device="cuda")
if os.environ.get("TILED") == "1":
    L.ggq_quantize_q8_1_tiled.restype = ctypes.c_int; L.ggq_quantize_q8_1_tiled.argtypes = L.ggq_quantize_q8_1_mmq.argtypes
    L.ggq_mul_mat_q_pretiled.restype = ctypes.c_int; L.ggq_mul_mat_q_pretiled.argtypes = L.ggq_mul_mat_q_prequant.argtypes
    L.ggq_quantize_q8_1_tiled(vp(x), 1, vp(scr), batch, K, t, st()); mm = L.ggq_mul_mat_q_pretiled
else:
    L.ggq_quantize_q8_1_mmq(vp(x), 1, vp(scr), batch, K, t, st()); mm = L.ggq_mul_mat_q_prequant
for _ in range(5):
    mm(vp(w), vp(scr), vp(y), t, 1, batch, K, N, N, st())
torch.cuda.synchronize()
buf = np.zeros(8192 * 8, dtype=np.uint64)
L.ggq_debug_read_stamps.restype = ctypes.c_int
L.ggq_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
assert L.ggq_debug_read_stamps(buf.ctypes.data, buf.size) == 0
s = buf.reshape(-1, 8)[:, :5].astype(np.float64)
s = s[s[:, 0] > 0]
t00 = s[:, 0].min()
s = (s - t00) / 100.0   # 100 MHz -> us
print("waves", len(s), "kernel span (first start -> last end) %.2f us" % s[:, 4].max())
names = ["start", "first loads done", "loop done", "after barrier", "end"]
for i, n in enumerate(names):
    c = s[:, i]
    print("%-18s min %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % (n, c.min(), np.median(c), np.percentile(c, 90), c.max()))
for i in range(1, 5):
    d = s[:, i] - s[:, i - 1]
    print("%-34s min %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % (names[i - 1] + " -> " + names[i], d.min(), np.median(d), np.percentile(d, 90), d.max()))
