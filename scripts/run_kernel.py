"""Run one hot-path kernel a few times (for rocprofv3 --pmc / --kernel-trace runs).
usage: python scripts/run_kernel.py {mmq|mmq_ref_layout|t16|x64|mmvq|dequant|quant} [type] [batch] [iters]"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))
import torch
from ggq import lib as ggqlib, synth
what = sys.argv[1] if len(sys.argv) > 1 else "mmq"
t = int(sys.argv[2]) if len(sys.argv) > 2 else 12
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 128
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
K, N = int(os.environ.get('K', 4096)), int(os.environ.get('N', 11008))
L = ggqlib.hip() if not os.environ.get("GGQ_LIB") else ggqlib._bind(ctypes.CDLL(os.environ["GGQ_LIB"]), ggqlib.HIP_SYMBOLS)
vp = lambda x: ctypes.c_void_p(x.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
w = torch.from_numpy(synth.random_weight(t, N, K, seed=0)).cuda()
x = torch.randn((batch, K), generator=torch.Generator().manual_seed(0)).half().cuda()
y = torch.empty((batch, N), dtype=torch.float16, device="cuda")
out = torch.empty((N, K), dtype=torch.float16, device="cuda")
scr = torch.empty(int(L.ggq_mmq_scratch_bytes(batch, K)) + 4096, dtype=torch.uint8, device="cuda")
if what == "mmq":   # the kernel ggq_mul_mat_q runs for this (type, batch): streamed kernel on the fragment-major scratch
    L.ggq_quantize_q8_1_tiled(vp(x), 1, vp(scr), batch, K, t, st())
elif what == "mmq_ref_layout":
    L.ggq_quantize_q8_1_mmq(vp(x), 1, vp(scr), batch, K, t, st())
elif what == "t16":   # the 16-token-tile kernel on its own scratch layout
    assert L.ggq_quantize_q8_1_t16(vp(x), 1, vp(scr), batch, K, t, st()) == 0
elif what == "x64":   # the 64 x 64 wave-tile kernel on its own scratch layout
    assert L.ggq_quantize_q8_1_x64(vp(x), 1, vp(scr), batch, K, t, st()) == 0
for _ in range(iters):
    if what == "mmq":
        L.ggq_mul_mat_q_pretiled(vp(w), vp(scr), vp(y), t, 1, batch, K, N, N, st())
    elif what == "mmq_ref_layout":
        L.ggq_mul_mat_q_prequant(vp(w), vp(scr), vp(y), t, 1, batch, K, N, N, st())
    elif what == "t16":
        assert L.ggq_mul_mat_q_t16(vp(w), vp(scr), vp(y), t, 1, batch, K, N, N, 0, None, st()) == 0
    elif what == "x64":
        assert L.ggq_mul_mat_q_x64(vp(w), vp(scr), vp(y), t, 1, batch, K, N, N, 0, None, st()) == 0
    elif what == "mmvq":
        L.ggq_mul_mat_vec_q(vp(w), vp(x), vp(y), t, 1, K, N, vp(scr), st())
    elif what == "dequant":
        L.ggq_dequantize_f16(vp(w), vp(out), t, N, K, st())
    elif what == "quant":
        L.ggq_quantize_q8_1_mmq(vp(x), 1, vp(scr), batch, K, t, st())
torch.cuda.synchronize()
print("done", what, t, batch, iters)
