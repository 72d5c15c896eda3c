"""Time ggq_mul_mat_q_prequant / _pretiled over a sweep of row counts (HIP events around a hipGraph of 96 launches; COLD=1: ring of 16 weight tensors).
usage: python scripts/sweep_mmq.py [type] [batch] rows1 rows2 ..."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))
import torch
from ggq import lib as ggqlib, synth
t = int(sys.argv[1]) if len(sys.argv) > 1 else 12
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rows_list = [int(a) for a in sys.argv[3:]] or [4096, 8192, 11008, 16384]
K = int(os.environ.get("K", 4096))
L = ggqlib.hip() if not os.environ.get('GGQ_LIB') else ggqlib._bind(ctypes.CDLL(os.environ['GGQ_LIB']), ggqlib.HIP_SYMBOLS)
vp = lambda x: ctypes.c_void_p(x.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
NMAX = max(rows_list)
w = torch.from_numpy(synth.random_weight(t, NMAX, K, seed=0)).cuda()
x = torch.randn((batch, K), generator=torch.Generator().manual_seed(0)).half().cuda()
scr = torch.empty(int(L.ggq_mmq_scratch_bytes(batch, K)) + 4096, dtype=torch.uint8, device="cuda")
TILED = os.environ.get("TILED") == "1"
import ctypes as C
if TILED:
    for f in (L.ggq_quantize_q8_1_tiled, L.ggq_mul_mat_q_pretiled):
        f.restype = C.c_int
    L.ggq_quantize_q8_1_tiled.argtypes = L.ggq_quantize_q8_1_mmq.argtypes
    L.ggq_mul_mat_q_pretiled.argtypes = L.ggq_mul_mat_q_prequant.argtypes
    scr = torch.empty(int(L.ggq_mmq_scratch_bytes(batch + 32, K)) + 4096, dtype=torch.uint8, device="cuda")
    L.ggq_quantize_q8_1_tiled(vp(x), 1, vp(scr), batch, K, t, st())
    mm = L.ggq_mul_mat_q_pretiled
else:
    L.ggq_quantize_q8_1_mmq(vp(x), 1, vp(scr), batch, K, t, st())
    mm = L.ggq_mul_mat_q_prequant
if os.environ.get("CHECK") == "1":
    N = rows_list[0]
    y0 = torch.empty((batch, N), dtype=torch.float16, device="cuda"); y1 = torch.empty_like(y0)
    scr0 = torch.empty(int(L.ggq_mmq_scratch_bytes(batch, K)) + 4096, dtype=torch.uint8, device="cuda")
    L.ggq_quantize_q8_1_mmq(vp(x), 1, vp(scr0), batch, K, t, st())
    mm(vp(w), vp(scr), vp(y1), t, 1, batch, K, N, N, st())
    L.ggq_mul_mat_q_prequant(vp(w), vp(scr0), vp(y0), t, 1, batch, K, N, N, st())
    torch.cuda.synchronize()
    print("check: max abs diff vs prequant path", (y0.float() - y1.float()).abs().max().item(), "max |y|", y0.float().abs().max().item())
COLD = os.environ.get("COLD") == "1"   # consecutive launches cycle 16 distinct weight tensors (> L2 + Infinity Cache)
ws = [w] + ([w.clone() for _ in range(15)] if COLD else [])
for N in rows_list:
    y = torch.empty((batch, N), dtype=torch.float16, device="cuda")
    for i in range(16):
        mm(vp(ws[i % len(ws)]), vp(scr), vp(y), t, 1, batch, K, N, N, st())
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()   # graph replay: no host launch gaps in the figure
    with torch.cuda.graph(g):
        for i in range(96):
            mm(vp(ws[i % len(ws)]), vp(scr), vp(y), t, 1, batch, K, N, N, st())
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / (3 * 96)
    print(f"type {t} batch {batch} K {K} rows {N} {'cold' if COLD else 'warm'}: {us:.2f} us  ({N*K*batch*2/us/1e6:.1f} TOP/s)", flush=True)
