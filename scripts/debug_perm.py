import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ggq import synth
from ggq.formats import GGMLType
import util
for t in [GGMLType.Q8_0, GGMLType.Q4_0]:
    n_rows, k, batch = 1024, 4096, 128
    w1 = synth.random_weight(t, 1, k, seed=21)
    w = np.repeat(w1, n_rows, axis=0)
    g = torch.Generator().manual_seed(22)
    x = torch.randn((batch, k), generator=g).cuda()
    y = util.gpu_mmq(w, x, t, n_rows)   # fp32 out
    ref = y[:, :1]
    ne = (y != ref)
    print(t.name, "cols differing from col0:", ne.any(0).sum().item(), "elements", ne.sum().item())
    cols = ne.any(0).nonzero().flatten().cpu().numpy()
    print("  col%32 histogram:", np.bincount(cols % 32, minlength=32))
    print("  tile histogram (first 10 tiles):", np.bincount(cols // 32, minlength=32)[:10])
    toks = ne.any(1).nonzero().flatten().cpu().numpy()
    print("  tok%32 hist:", np.bincount(toks % 32, minlength=32), " tok//32:", np.bincount(toks//32, minlength=4))
    d = (y - ref).abs().max().item(); print("  max abs diff", d, "rel", d / ref.abs().max().item())
