import sys, os, ctypes
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))
import torch
from ggq import lib as ggqlib
L = ggqlib.hip() if not os.environ.get("GGQ_LIB") else ggqlib._bind(ctypes.CDLL(os.environ["GGQ_LIB"]), ggqlib.HIP_SYMBOLS)
vp = lambda x: ctypes.c_void_p(x.data_ptr()); st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, batch in [(n, b) for b in (128, 32, 512, 4096) for n in ("tiled", "x64")]:
    K = 4096
    x = torch.randn((batch, K), generator=torch.Generator().manual_seed(0)).half().cuda()
    scr = torch.empty(int(L.ggq_mmq_scratch_bytes(batch, K)) + 4096, dtype=torch.uint8, device="cuda")
    fn = L.ggq_quantize_q8_1_tiled if name == "tiled" else L.ggq_quantize_q8_1_x64
    f = lambda: fn(vp(x), 1, vp(scr), batch, K, 12, st())
    for _ in range(10): f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(100): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"quantize_q8_1_{name} batch {batch}: {e0.elapsed_time(e1) * 1000 / 200:.2f} us per launch (graph replay, includes the ~1.6 us launch floor)")
