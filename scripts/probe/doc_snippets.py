import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "ggml-libtorch_amd"))
import torch, ctypes
import ggml as ops
from ggq import synth
from ggq.formats import GGMLType, NEED_SUM
from ggq.dist import SlabGather, shard_rows
from ggq.linear import QuantizedActivations, QuantLinear, QuantGatedFFN
from ggq.gguf_io import GGUFReader, write_sample_file
import tempfile
t, n_rows, k, batch = GGMLType.Q4_K, 512, 1024, 40
w = torch.from_numpy(synth.random_weight(t, n_rows, k, seed=0)).cuda()
x = torch.randn(batch, k).half().cuda()
y = ops.ggml_mul_mat_a8(w, x, int(t), n_rows)
sg = SlabGather(batch, n_rows, x.dtype, x.device)
QuantizedActivations(x, int(t) in {int(q) for q in NEED_SUM}).matmul(w, int(t), n_rows, out=sg.local)
sg.gather()
assert torch.equal(sg.batch_major(), y), "SlabGather example"
lin = QuantLinear(w.reshape(-1), int(t), k, n_rows, bias=torch.randn(n_rows).half().cuda())
print("QuantLinear", lin(x).shape)
d = tempfile.mkdtemp(); p = write_sample_file(d, t, 1024)
for tt in GGUFReader(p).tensors:
    m, n = map(int, tt.name.split("_")[-1].split("x"))
    ww = torch.tensor(tt.data, device="cuda")
    print(tt.name, ops.ggml_mul_mat_a8(ww, torch.randn(3, n).half().cuda(), tt.tensor_type, ww.size(0)).shape)
    break
print("doc snippets ok")
