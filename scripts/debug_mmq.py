import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ggq import synth
from ggq.formats import GGMLType
from oracle import oracle as O
import util
np.set_printoptions(linewidth=200, precision=4, suppress=True)
for t, batch, k, n in [(GGMLType.Q4_0, 1, 256, 33), (GGMLType.Q4_0, 40, 512, 32), (GGMLType.Q8_0, 100, 256, 32)]:
    w = synth.random_weight(t, n, k, seed=1)
    g = torch.Generator().manual_seed(4)
    x = torch.randn((batch, k), generator=g).cuda()
    y = util.gpu_mmq(w, x, t, n).cpu().numpy()
    ref, ya = O.mul_mat_q(w, x.cpu().numpy(), t, n)
    print(t.name, batch, k, n)
    print(" gpu", y[0, :12]); print(" ref", ref[0, :12]); print(" ratio", (y[0,:12]/ref[0,:12]))
    print(" gpu tok -1", y[-1, :8]); print(" ref", ref[-1, :8])
