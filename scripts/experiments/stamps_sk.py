"""Per-wave timeline of the stream-K MMQ kernel (variant built with -DGGQ_SK_STAMPS):
scripts/build_variant.sh sk "-DGGQ_SK_STAMPS -DGGQ_TUNING" mmq_sk
usage: GGQ_LIB=scripts/_variants/libggq_sk.so python scripts/stamps_sk.py [type] [batch] [rows]"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))
import torch, numpy as np
from ggq import lib as ggqlib, synth
t = int(sys.argv[1]) if len(sys.argv) > 1 else 12
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
N = int(sys.argv[3]) if len(sys.argv) > 3 else 11008
K = int(os.environ.get("K", 4096))
L = ggqlib._bind(ctypes.CDLL(os.environ["GGQ_LIB"]), ggqlib.HIP_SYMBOLS)
vp = lambda x: ctypes.c_void_p(x.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
w = torch.from_numpy(synth.random_weight(t, N, K, seed=0)).cuda()
x = torch.randn((batch, K), generator=torch.Generator().manual_seed(0)).half().cuda()
scr = torch.empty(int(L.ggq_mmq_scratch_bytes(batch, K)) + 4096, dtype=torch.uint8, device="cuda")
y = torch.empty((batch, N), dtype=torch.float16, device="cuda")
L.ggq_quantize_q8_1_tiled(vp(x), 1, vp(scr), batch, K, t, st())
for _ in range(5):
    L.ggq_mul_mat_q_pretiled(vp(w), vp(scr), vp(y), t, 1, batch, K, N, N, st())
torch.cuda.synchronize()
buf = np.zeros(1024 * 4 * 16, dtype=np.uint64)
L.ggq_debug_read_sk_stamps.restype = ctypes.c_int
L.ggq_debug_read_sk_stamps.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
assert L.ggq_debug_read_sk_stamps(buf.ctypes.data, buf.size) == 0
s = buf.reshape(-1, 16).astype(np.float64)
s = s[s[:, 0] > 0]
t00 = s[:, 0].min()
s = np.where(s > 0, (s - t00) / 100.0, np.nan)   # 100 MHz -> us
print("waves", len(s), "kernel span %.2f us" % np.nanmax(s))
names = ["kernel start"] + [f"seg{k} {n}" for k in range(3) for n in ("start", "first loads landed", "K loop done", "reduced", "end")]
def row(label, c):
    c = c[~np.isnan(c)]
    if len(c): print("%-40s n %5d  min %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % (label, len(c), c.min(), np.median(c), np.percentile(c, 90), c.max()))
for i, n in enumerate(names): row(n, s[:, i])
print("--- durations")
for i in range(1, 16): row(names[i - 1] + " -> " + names[i], s[:, i] - s[:, i - 1])
