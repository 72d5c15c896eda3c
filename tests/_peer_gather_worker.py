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
    # the kernels' own multi-destination write-back: batch 8 goes through the 16-token-tile kernel (its last workgroup publishes the
    # flags), batch 40 and 128 (BASELINE config 5's batch) through the streamed kernel (every wave that stores arrives; 32- and
    # 64-token units, four K-slices), batch 1 through the fused GEMV (ggq_mul_mat_vec_q_gather); Q2_K at batch 2 is a dot4 shape:
    # the fallback (ggq_mul_mat_q_ld + scatter) inside the same call.  A taller matrix for batch 128 (several units per rank,
    # a ragged last row tile), fp32 / bf16 outputs once each (scalar peer stores).
    from ggq import lib as ggqlib
    L = ggqlib.hip()
    cases = [(t, 8, n_rows, torch.float16, 4), (t, 40, n_rows, torch.float16, 3), (t, 128, 200 * world, torch.float16, 3),
             (t, 1, n_rows, torch.float16, None), (t, 1, 1376 * world, torch.float16, None), (GGMLType.Q8_0, 128, 72 * world, torch.float32, 3),
             (GGMLType.Q6_K, 48, 40 * world, torch.bfloat16, 3), (GGMLType.Q2_K, 2, n_rows, torch.float16, 1),
             # the 64 x 64 wave-tile kernel: 64-row units (four K-slices at K = 1024), fp16 16-byte and fp32 scalar peer stores, a ragged last unit
             (GGMLType.Q4_0, 640, 616 * world, torch.float16, 5), (t, 1280, 520 * world, torch.float32, 5),
             (t, 2048, 608 * world, torch.float16, 5)]   # (96-row units: 8-byte peer stores)
    for (t2, b2, n2, dt2, want_route) in cases:
        w2 = synth.random_weight(t2, n2, k, seed=21 + b2)
        s2, e2 = shard_rows(n2, world, rank)
        if want_route is not None:
            assert L.ggq_mmq_route(int(t2), b2, k, e2 - s2) == want_route, (t2, b2, e2 - s2, L.ggq_mmq_route(int(t2), b2, k, e2 - s2))
        with PeerSlabGather(b2, n2, dt2, torch.device("cuda", 0)) as pg:
            wd = torch.from_numpy(w2[s2:e2]).cuda()
            seen = []
            for it in range(5):
                x = torch.randn((b2, k), generator=torch.Generator().manual_seed(40 + it)).to(dt2).cuda()
                if rank == it % world:
                    torch.cuda._sleep(10_000_000)
                pg.matmul_gather(x, wd, t2)
                seen.append((x, pg.batch_major().float() + 1.0))
            torch.cuda.synchronize()
            assert pg.status() == 0, "a ggq_peer_wait timed out"
            for x, got in seen:
                # the one-GPU result of the same shards: every slab is computed by the kernel its rank's shape routes to
                parts = [(util.gpu_mmvq if b2 == 1 else util.gpu_mmq)(w2[a:b], x, t2, b - a) for a, b in (shard_rows(n2, world, r) for r in range(world))]
                ok = ok and torch.equal(got, torch.cat([p_.reshape(b2, -1) for p_ in parts], dim=1).float() + 1.0)
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
