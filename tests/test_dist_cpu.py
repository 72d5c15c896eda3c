"""CPU suite: the N > 1 path (row-sharded weights + all-gather of output slabs) with
world_size 2 and 3 over gloo.  The per-rank matmul is injected (oracle on CPU) — the
partition / gather / layout logic under test is exactly what runs over RCCL on GPUs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ggq import synth
from ggq.dist import shard_rows, gather_slabs, unpermute_gathered, RowShardedQuantLinear, SlabGather
from ggq.formats import GGMLType


def test_shard_rows_partitions_exactly():
    for n in (1, 7, 32, 11008, 28672):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_rows(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_rows(28672, 8, 3) == (3 * 3584, 4 * 3584)  # 3584 rows / GPU


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_rows, quant_type, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        k, batch = 512, 5
        w_full = synth.random_weight(quant_type, n_rows, k, seed=4)
        x = torch.randn((batch, k), generator=torch.Generator().manual_seed(1))

        def cpu_matmul(w, xx, t, rows):  # same signature as ggml.ggml_mul_mat_a8
            y, _ = O.mul_mat_q(w.numpy(), xx.numpy(), t, rows)
            return torch.from_numpy(y)

        s, e = shard_rows(n_rows, world, rank)
        layer = RowShardedQuantLinear(torch.from_numpy(w_full[s:e]), quant_type, n_rows, matmul=cpu_matmul)
        y = layer(x)
        ref, _ = O.mul_mat_q(w_full, x.numpy(), quant_type, n_rows)
        assert y.shape == (batch, n_rows)
        # each output column is computed by exactly one rank with the same code: bit-exact
        assert np.array_equal(y.numpy(), ref), f"rank {rank}: gathered result differs from the 1-process result"
        if n_rows % world == 0:  # async form used by bench.py
            buf, work = gather_slabs(layer.local(x), n_rows, async_op=True)
            work.wait()
            assert np.array_equal(unpermute_gathered(buf).numpy(), ref)
            # in-place form (bench.py's configs[4] leg): the slab is written into the rank's slot, gathered in place
            sg = SlabGather(batch, n_rows, torch.float32, "cpu")
            sg.local.copy_(layer.local(x))          # on the GPU the kernel writes here directly (row pitch = rows)
            before = sg.local.data_ptr()
            sg.gather()
            assert sg.local.data_ptr() == before and sg.buf.shape == (world, batch, n_rows // world)
            assert np.array_equal(sg.batch_major().numpy(), ref)
            for r in range(world):                  # slot r = rank r's columns, untouched layout
                s_r, e_r = shard_rows(n_rows, world, r)
                assert np.array_equal(sg.buf[r].numpy(), ref[:, s_r:e_r])
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_rows", [(2, 64), (2, 37), (3, 50)])
def test_row_sharded_allgather_gloo(tmp_path, world, n_rows):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_rows, int(GGMLType.Q4_K), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
