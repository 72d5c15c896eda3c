import sys, os, ctypes, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))
import torch, numpy as np
from ggq import lib as ggqlib, synth
sys.path.insert(0, ROOT)
from bench import time_launches, vp, cur_stream, algo_bytes_matmul
K, N = 4096, 11008
L = ggqlib.hip()
for t in (2, 12, 8):
    w = torch.from_numpy(synth.random_weight(t, N, K, seed=0)).cuda()
    x = torch.randn((1, K), generator=torch.Generator().manual_seed(0)).half().cuda()
    y = torch.empty((1, N), dtype=torch.float16, device="cuda")
    scr = torch.empty(8192, dtype=torch.uint8, device="cuda")
    us, mn = time_launches(lambda: L.ggq_mul_mat_vec_q(vp(w), vp(x), vp(y), t, 1, K, N, vp(scr), cur_stream()), 200)
    L.ggq_quantize_q8_1(vp(x), 1, vp(scr), 1, K, cur_stream())
    us2, mn2 = time_launches(lambda: L.ggq_mul_mat_vec_q_prequant(vp(w), vp(scr), vp(y), t, 1, K, N, cur_stream()), 200)
    b = algo_bytes_matmul(t, N, K, 1)
    print(f"type {t} RPW={os.environ.get('GGQ_MMVQ_RPW','auto')}: fused {us:.2f} us ({b/us/1e3:.0f} GB/s)  prequant-only {us2:.2f} us ({b/us2/1e3:.0f} GB/s)")
