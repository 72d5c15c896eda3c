"""16-token-tile kernel vs what ggq_mul_mat_q runs today: kernel alone and op (quantise + kernel), warm and cold, graph-timed.
usage: python scripts/sweep_t16.py [type] [rows] [k] [batches...]"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))
import torch
from ggq import lib as ggqlib, synth
t = int(sys.argv[1]) if len(sys.argv) > 1 else 12
N = int(sys.argv[2]) if len(sys.argv) > 2 else 11008
K = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
batches = [int(a) for a in sys.argv[4:]] or [5, 8, 16, 17, 32]
L = ggqlib._bind(ctypes.CDLL(os.environ["GGQ_LIB"]), ggqlib.HIP_SYMBOLS) if os.environ.get("GGQ_LIB") else ggqlib.hip()   # (a variant from scripts/build_variant.sh)
vp = lambda x: ctypes.c_void_p(x.data_ptr()); st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
w0 = torch.from_numpy(synth.random_weight(t, N, K, seed=0)).cuda()
nring = max(2, (352 << 20) // w0.numel() + 2)
ws = [w0] + [w0.clone() for _ in range(nring - 1)]

def timeit(f, cold):
    for i in range(4): f(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(64): f(i if cold else 0)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / 128

for b in batches:
    x = torch.randn((b, K), generator=torch.Generator().manual_seed(0)).half().cuda()
    y = torch.empty((b, N), dtype=torch.float16, device="cuda")
    scr = torch.empty(int(L.ggq_mmq_scratch_bytes(b, K)) + 4096, dtype=torch.uint8, device="cuda")
    scr2 = torch.empty(int(L.ggq_mmq_scratch_bytes(b, K)) + 4096, dtype=torch.uint8, device="cuda")
    assert L.ggq_quantize_q8_1_t16(vp(x), 1, vp(scr2), b, K, t, st()) == 0
    def op_old(i): assert L.ggq_mul_mat_q(vp(ws[i % nring]), vp(x), vp(y), t, 1, b, K, N, vp(scr), st()) == 0
    def k_new(i): assert L.ggq_mul_mat_q_t16(vp(ws[i % nring]), vp(scr2), vp(y), t, 1, b, K, N, N, 0, None, st()) == 0
    def q_new(i): assert L.ggq_quantize_q8_1_t16(vp(x), 1, vp(scr2), b, K, t, st()) == 0
    def op_new(i): q_new(i); k_new(i)
    r = {n: (timeit(f, 0), timeit(f, 1)) for n, f in (("op_old", op_old), ("quant_t16", q_new), ("kernel_t16", k_new), ("op_t16", op_new))}
    print(f"type {t} {N}x{K} batch {b:3d}: " + "  ".join(f"{n} {a:6.2f}/{c:6.2f}" for n, (a, c) in r.items()) + "  (us warm/cold)", flush=True)
