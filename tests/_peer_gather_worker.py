"""worker of tests/test_peer_gather.py: RANK/WORLD_SIZE processes that all use cuda:0 (gloo group for the hand-off)"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from ggq import synth                       # noqa: E402
from ggq.dist import PeerSlabGather, shard_rows   # noqa: E402
from ggq.formats import GGMLType            # noqa: E402
import util                                 # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t, batch, k, n_rows = GGMLType.Q4_K, 40, 1024, 64 * world
    w = synth.random_weight(t, n_rows, k, seed=11)
    s, e = shard_rows(n_rows, world, rank)
    ok = True
    with PeerSlabGather(batch, n_rows, torch.float16, torch.device("cuda", 0)) as pg:
        consumed = []
        for it in range(6):   # several gathers back to back, a consumer kernel of the gathered buffer in between, NO host
            # synchronisation inside the loop: a rank that runs ahead must not overwrite slabs a slower peer still reads
            x = torch.randn((batch, k), generator=torch.Generator().manual_seed(12 + it)).half().cuda()
            y_local = util.gpu_mmq(w[s:e], x, t, e - s)
            if rank == it % world:
                torch.cuda._sleep(20_000_000)   # ~10 ms of device time: this rank lags behind the others
            pg.local.copy_(y_local)
            pg.gather()
            consumed.append((x, (pg.batch_major().float() * 2.0)))   # the consumer: reads the gathered buffer on the stream
        torch.cuda.synchronize()
        assert pg.status() == 0, "a ggq_peer_wait timed out"
        for x, got in consumed:
            full = util.gpu_mmq(w, x, t, n_rows)    # the one-GPU result: every slab is computed by the same kernels
            ok = ok and torch.equal(got, full.float() * 2.0)
    # the GEMM's own multi-destination write-back (ggq_mul_mat_q_gather): batch 8 goes through the 16-token-tile kernel, whose last
    # workgroup publishes the flags; batch 40 takes the fallback (ggq_mul_mat_q_ld + scatter) inside the same call
    for b2 in (8, 40):
        with PeerSlabGather(b2, n_rows, torch.float16, torch.device("cuda", 0)) as pg:
            wd = torch.from_numpy(w[s:e]).cuda()
            seen = []
            for it in range(5):
                x = torch.randn((b2, k), generator=torch.Generator().manual_seed(40 + it)).half().cuda()
                if rank == it % world:
                    torch.cuda._sleep(10_000_000)
                pg.matmul_gather(x, wd, t)
                seen.append((x, pg.batch_major().float() + 1.0))
            torch.cuda.synchronize()
            assert pg.status() == 0, "a ggq_peer_wait timed out"
            for x, got in seen:
                ok = ok and torch.equal(got, util.gpu_mmq(w, x, t, n_rows).float() + 1.0)
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    dist.destroy_process_group()
    if not all(flags):
        print(f"rank {rank}: gathered result differs: {flags}", flush=True)
        sys.exit(1)
    if rank == 0:
        print("peer gather ok", flush=True)


if __name__ == "__main__":
    main()
