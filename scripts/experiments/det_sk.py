"""debug: is the stream-K kernel deterministic / finite?  usage: GGQ_LIB=... python scripts/det_sk.py type batch rows K"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))
import torch, numpy as np
from ggq import lib as ggqlib, synth
t = int(sys.argv[1]); batch = int(sys.argv[2]); N = int(sys.argv[3]); K = int(sys.argv[4])
L = ggqlib.hip() if not os.environ.get('GGQ_LIB') else ggqlib._bind(ctypes.CDLL(os.environ['GGQ_LIB']), ggqlib.HIP_SYMBOLS)
vp = lambda x: ctypes.c_void_p(x.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
w = torch.from_numpy(synth.random_weight(t, N, K, seed=0)).cuda()
x = torch.randn((batch, K), generator=torch.Generator().manual_seed(0)).half().cuda()
scr = torch.zeros(int(L.ggq_mmq_scratch_bytes(batch, K)) + 4096, dtype=torch.uint8, device="cuda")
L.ggq_quantize_q8_1_tiled(vp(x), 1, vp(scr), batch, K, t, st())
ys = []
for i in range(4):
    y = torch.zeros((batch, N), dtype=torch.float16, device="cuda")
    L.ggq_mul_mat_q_pretiled(vp(w), vp(scr), vp(y), t, 1, batch, K, N, N, st()); torch.cuda.synchronize()
    ys.append(y.cpu().numpy().view(np.uint16))
    if os.environ.get("DUMP"):
        tb = int(os.environ.get("GGQ_SK_TB", 2))
        off = ((batch + 31) // 32 * ((K - K % 512 + 512) // 64) * 2816 + 255) // 256 * 256 + 4096
        d = scr[off:off + 256 * 4 * tb * 16 * 64 * 4].cpu().numpy().view(np.uint32).reshape(256, 4, tb * 16, 64)
        dumps = globals().setdefault("dumps", []); dumps.append(d.copy())
print(os.environ.get('GGQ_LIB', 'default'), "nonfinite per launch:", [int((~np.isfinite(a.view(np.float16))).sum()) for a in ys],
      "elements differing from launch 0:", [int((a != ys[0]).sum()) for a in ys[1:]])

if os.environ.get("DUMP"):
    for a in dumps[1:]:
        diff = a != dumps[0]
        print("pre-reduction acc words differing:", int(diff.sum()), "waves affected", int(diff.any(axis=(2, 3)).sum()),
              "lanes histogram (16-lane quarters):", [int(diff[..., 16 * i:16 * i + 16].sum()) for i in range(4)])
        print("  by wave ks:", diff.sum(axis=(0, 2, 3)).tolist(), " by register (jj*16+i):", diff.sum(axis=(0, 1, 3)).tolist())
        print("  by wg (first 40):", diff.sum(axis=(1, 2, 3))[:40].tolist())
        w0 = np.argwhere(diff.any(axis=(2, 3)))[0]; dd = diff[w0[0], w0[1]]
        print("  first bad wave", w0, "regs", np.where(dd.any(axis=1))[0].tolist(), "lanes", np.where(dd.any(axis=0))[0].tolist())
        va = a[w0[0], w0[1]].view(np.float32); vb = dumps[0][w0[0], w0[1]].view(np.float32)
        r0 = np.where(dd.any(axis=1))[0][0]; print("  reg", r0, "lanes 44..63 run:", va[r0, 44:], "run0:", vb[r0, 44:])
