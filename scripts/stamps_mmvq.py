"""Per-wave timeline of the fused MMVQ kernel (variant built with -DGGQ_VSTAMP).  COLD=1: the stamped launch reads a weight
copy that the previous launches have pushed out of L2 + Infinity Cache."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))
import torch, numpy as np
from ggq import lib as ggqlib, synth
t = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N, K = 11008, 4096
L = ggqlib._bind(ctypes.CDLL(os.environ["GGQ_LIB"]), ggqlib.HIP_SYMBOLS)
vp = lambda x: ctypes.c_void_p(x.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
w = torch.from_numpy(synth.random_weight(t, N, K, seed=0)).cuda()
x = torch.randn((1, K), generator=torch.Generator().manual_seed(0)).half().cuda()
scr = torch.empty(int(L.ggq_mmvq_scratch_bytes(K)) + 64, dtype=torch.uint8, device="cuda")
y = torch.empty((1, N), dtype=torch.float16, device="cuda")
cold = os.environ.get("COLD") == "1"
ws = [w] + ([w.clone() for _ in range((352 << 20) // w.numel() + 2)] if cold else [])
for i in range(len(ws) + 4 if cold else 5):
    L.ggq_mul_mat_vec_q(vp(ws[i % len(ws)]), vp(x), vp(y), t, 1, K, N, vp(scr), st())
torch.cuda.synchronize()
buf = np.zeros(8192 * 8, dtype=np.uint64)
L.ggq_debug_read_vstamps.restype = ctypes.c_int
L.ggq_debug_read_vstamps.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
assert L.ggq_debug_read_vstamps(buf.ctypes.data, buf.size) == 0
s = buf.reshape(-1, 8)[:, :5].astype(np.float64)
s = s[s[:, 0] > 0]
s = (s - s[:, 0].min()) / 100.0
names = ["start", "x quantised", "after barrier", "sums done", "end"]
print("cold" if cold else "warm", "waves", len(s), "span %.2f us" % s[:, 4].max())
for i, n in enumerate(names):
    c = s[:, i]; print("%-14s min %5.2f p50 %5.2f p90 %5.2f max %5.2f" % (n, c.min(), np.median(c), np.percentile(c, 90), c.max()))
