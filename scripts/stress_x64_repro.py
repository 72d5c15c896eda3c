"""run-to-run bit reproducibility of the x64 kernel under a forced launch shape (tuning build: GGQ_X64_KS / GGQ_X64_ROWS), fp16 and fp32 outputs.
usage: GGQ_LIB=... GGQ_X64_KS=1 [GGQ_X64_ROWS=3] python scripts/stress_x64_repro.py TYPE "B K N" ..."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))
import numpy as np, torch
from ggq import lib as ggqlib, synth
L = ggqlib.hip() if not os.environ.get("GGQ_LIB") else ggqlib._bind(ctypes.CDLL(os.environ["GGQ_LIB"]), ggqlib.HIP_SYMBOLS)
vp = lambda x: ctypes.c_void_p(x.data_ptr()); st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
t = int(sys.argv[1])
for spec in sys.argv[2:]:
    b, K, N = (int(v) for v in spec.split())
    w = torch.from_numpy(synth.random_weight(t, N, K, seed=b + K + 2)).cuda()
    x = torch.randn((b, K), generator=torch.Generator().manual_seed(23)).half().cuda()
    scr = torch.empty(int(L.ggq_mmq_scratch_bytes(b, K)), dtype=torch.uint8, device="cuda")
    for dt, td in ((1, torch.float16), (0, torch.float32)):
        ref, bad, nbad = None, 0, 0
        for rep in range(12):
            y = torch.zeros((b, N), dtype=td, device="cuda")
            assert L.ggq_quantize_q8_1_x64(vp(x), 1, vp(scr), b, K, t, st()) == 0
            assert L.ggq_mul_mat_q_x64(vp(w), vp(scr), vp(y), t, dt, b, K, N, N, 0, None, st()) == 0
            torch.cuda.synchronize()
            if ref is None: ref = y
            else:
                d = int((ref != y).sum().item())
                if d: bad += 1; nbad = max(nbad, d)
        print(f"type {t} KS={os.environ.get('GGQ_X64_KS')} ROWS={os.environ.get('GGQ_X64_ROWS')} batch {b} K {K} N {N} {td}: {bad} of 11 repeats differ (max {nbad} elements)", flush=True)
