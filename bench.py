#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native ggml block-quant hot path.

Metric (BASELINE.json): "Q4_K MMQ GEMV/GEMM GB/s + % HBM roofline, 4096x11008, batch 1/128".
One *step* = one pass of the hot path over one batch of synthetic input:
    ggml_mul_mat_a8(W[Q4_K, N=11008 x K=4096], X[128, 4096] fp16)  =  quantize_mmq_q8_1 + mul_mat_q
with every input already resident in HBM.  `value` = algorithmic bytes (quantised W + X + Y,
SURVEY.md §8d) of all steps of all ranks / wall time of the timed region, in GB/s.

    python bench.py [--gpus N] [--steps K] [--warmup W]
For N > 1 the driver starts it under torch.distributed.run (one rank per GPU, RCCL): every rank
owns an 11008-row shard (weak scaling) and the [128, 11008] slabs are all-gathered each step
on a side stream, overlapped with the next step's compute.

Prints ONE JSON line on rank 0.  Besides the contract keys it carries
  "roofline":     dominant kernel (mul_mat_q) vs the 8 TB/s HBM roof, duration measured live
                  with HIP events on the launch stream,
  "cpu_baseline": the oracle (CPU port of the reference algorithm) timed on this box's host cores,
  "extra":        the other BASELINE configs (batch-1 MMVQ, dequantise, other formats).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 measured achievable
Q4_K, Q5_K, Q6_K, Q4_0, Q8_0 = 12, 13, 14, 2, 8
K_DIM, N_DIM, BATCH = 4096, 11008, 128


def algo_bytes_matmul(t, n_rows, k, batch, esz=2):
    from ggq.formats import weight_bytes
    return weight_bytes(t, n_rows, k) + batch * k * esz + batch * n_rows * esz


def algo_bytes_dequant(t, n_rows, k):
    from ggq.formats import weight_bytes
    return weight_bytes(t, n_rows, k) + n_rows * k * 2


def vp(t):
    return ctypes.c_void_p(t.data_ptr())


def cur_stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def time_launches(fn, iters, reps=5, use_graph=True):
    """Average device time of one call of fn: `iters` back-to-back launches bracketed by HIP
    events recorded on the stream the kernels run on (torch's current stream)."""
    fn()
    torch.cuda.synchronize()
    runner = None
    if use_graph:
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(iters):
                    fn()
            runner = g.replay
        except Exception:  # pragma: no cover - capture unsupported: fall back to eager launches
            runner = None
            torch.cuda.synchronize()
    if runner is None:
        def runner():
            for _ in range(iters):
                fn()
    runner()
    torch.cuda.synchronize()
    times = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        runner()
        e1.record()
        e1.synchronize()
        times.append(e0.elapsed_time(e1) * 1e3 / iters)  # us per launch
    return float(np.median(times)), float(np.min(times))


def cpu_baseline(sample_rows=N_DIM):
    """The oracle (C port of the reference's mul_mat_q algorithm) on one host core, on a bounded
    sample of the same workload: `sample_rows` weight rows (default: all of them) x all 128 tokens."""
    from oracle import oracle as O
    from ggq import synth
    w = synth.random_weight(Q4_K, sample_rows, K_DIM, seed=0)
    x = torch.randn((BATCH, K_DIM), generator=torch.Generator().manual_seed(0)).half().float().numpy()
    O.lib()
    q8 = O.quantize_q8_1_mmq(x, Q4_K)
    y = np.empty((BATCH, sample_rows), np.float32)
    t0 = time.perf_counter()
    rc = O.lib().oracle_mul_mat_q(Q4_K, O._p(np.ascontiguousarray(w)), O._p(q8), O._p(y), None, BATCH, K_DIM,
                                  sample_rows)
    dt = time.perf_counter() - t0
    assert rc == 0
    nbytes = algo_bytes_matmul(Q4_K, sample_rows, K_DIM, BATCH)
    return {"value": round(nbytes / dt / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": f"oracle_mul_mat_q (quantised X given), first {sample_rows} of {N_DIM} Q4_K rows x {BATCH} tokens, "
                      f"K={K_DIM}, 1 pass, {dt:.2f} s", "host_cores_available": os.cpu_count()}


def reference_cpu_dequant():
    """BASELINE config 1 beside it: the reference's own compiled ggml-cpu op (oracle/_ref, when
    it travelled) on Q4_0 4096x4096."""
    try:
        from oracle import oracle as O
        from ggq import synth
        ref = O.load_reference_cpu_op()
        if ref is None:
            return None
        w = torch.from_numpy(synth.random_weight(Q4_0, 4096, 4096, seed=0))
        ref.ggml_dequantize(w, Q4_0, 4096, 4096)
        t0 = time.perf_counter()
        ref.ggml_dequantize(w, Q4_0, 4096, 4096)
        dt = time.perf_counter() - t0
        nb = 4096 * 4096 // 32 * 18 + 4096 * 4096 * 4
        return {"value": round(nb / dt / 1e9, 3), "unit": "GB/s", "cores": 1, "kind": "reference",
                "sample": f"reference ggml-cpu custom_ops.ggml_dequantize Q4_0 4096x4096 -> fp32, {dt * 1e3:.1f} ms"}
    except Exception as e:  # pragma: no cover
        return {"error": str(e)[:200]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--eager", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary configs and the CPU baseline")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus > 1 and world == 1:
            sys.exit("bench.py --gpus N>1 must run under torch.distributed.run with --nproc-per-node N")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)

    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("GGQ_BENCH_BACKEND", "nccl")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)

    import ggml  # noqa: F401  (loads the native extension; raises if it is missing)
    from ggq import lib as ggqlib, synth
    L = ggqlib.hip()

    # ---- workload: per-rank shard, inputs resident in HBM ----
    w = torch.from_numpy(synth.random_weight(Q4_K, N_DIM, K_DIM, seed=rank)).to(dev)
    x = torch.randn((BATCH, K_DIM), generator=torch.Generator().manual_seed(0)).half().to(dev)
    scratch = torch.empty(int(L.ggq_mmq_scratch_bytes(BATCH, K_DIM)), dtype=torch.uint8, device=dev)
    bytes_per_step = algo_bytes_matmul(Q4_K, N_DIM, K_DIM, BATCH)

    def barrier():
        if world > 1:
            dist.barrier()

    if world == 1:
        y = torch.empty((BATCH, N_DIM), dtype=torch.float16, device=dev)

        def step():
            rc = L.ggq_mul_mat_q(vp(w), vp(x), vp(y), Q4_K, 1, BATCH, K_DIM, N_DIM, vp(scratch), cur_stream())
            assert rc == 0, rc

        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        graph = None
        if not args.eager:
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    for _ in range(args.steps):
                        step()
                graph.replay()  # instantiate / upload once, untimed
                torch.cuda.synchronize()
            except Exception as e:  # pragma: no cover
                print(f"[bench] graph capture failed ({e}); timing eager launches", file=sys.stderr)
                graph = None
                torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if graph is not None:
            graph.replay()
        else:
            for _ in range(args.steps):
                step()
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        launch_mode = "hipGraph replay of all steps" if graph is not None else "eager launches"
    else:
        # rank-local slab + all-gather on a side stream, double buffered
        comm = torch.cuda.Stream(device=dev)
        ys = [torch.empty((BATCH, N_DIM), dtype=torch.float16, device=dev) for _ in range(2)]
        gathered = [torch.empty((world * BATCH, N_DIM), dtype=torch.float16, device=dev) for _ in range(2)]
        # (events are created once and re-recorded: per-step host overhead matters at ~30 us of GPU work per step)
        ready_ev = [torch.cuda.Event(), torch.cuda.Event()]
        done_ev = [torch.cuda.Event(), torch.cuda.Event()]
        used = [False, False]
        main_stream = torch.cuda.current_stream()

        def step(i):
            b = i & 1
            if used[b]:
                main_stream.wait_event(done_ev[b])  # slab b is free again
            rc = L.ggq_mul_mat_q(vp(w), vp(x), vp(ys[b]), Q4_K, 1, BATCH, K_DIM, N_DIM, vp(scratch), cur_stream())
            assert rc == 0, rc
            ready_ev[b].record(main_stream)
            comm.wait_event(ready_ev[b])
            with torch.cuda.stream(comm):
                dist.all_gather_into_tensor(gathered[b], ys[b])
                done_ev[b].record(comm)
            used[b] = True

        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        launch_mode = "eager launches, all-gather overlapped on a side stream"
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    ms_per_step = elapsed * 1e3 / args.steps
    value = world * bytes_per_step * args.steps / elapsed / 1e9

    out = {
        "metric": "Q4_K MMQ GEMV/GEMM GB/s + % HBM roofline, 4096x11008, batch 1/128",
        "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 6), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int8 x int8 -> int32 (MFMA), fp32 scale/accumulate, fp16 in/out",
        "data": "synthetic",
        "config": {"workload": "ggml_mul_mat_a8 = quantize_mmq_q8_1 + mul_mat_q, Q4_K W[11008 x 4096] per GPU, "
                               "X[128 x 4096] fp16 (BASELINE configs[3] shape, the one the metric is quoted on)",
                   "quant_type": "Q4_K", "k": K_DIM, "n_rows_per_gpu": N_DIM, "batch": BATCH,
                   "algorithmic_bytes_per_step_per_gpu": bytes_per_step,
                   "parallelism": f"row-shard x{world}" + (" + RCCL all-gather of [128 x 11008] slabs" if world > 1 else ""),
                   "launch": launch_mode, "cache_state": "warm loop over one 25 MB weight tensor (fits L2+MALL)"},
        "pct_hbm_roofline": round(100.0 * value / world / HBM_PEAK_GBS, 2),
    }

    if rank == 0 and world == 1:
        # ---- roofline of the dominant kernel (mul_mat_q alone, activations pre-quantised) ----
        # (the fused op quantises into the fragment-major scratch and runs the streamed kernel for Q4_K:
        #  time exactly that kernel, through the exported pre-quantised entry point)
        rc = L.ggq_quantize_q8_1_tiled(vp(x), 1, vp(scratch), BATCH, K_DIM, Q4_K, cur_stream())
        assert rc == 0

        def mmq_only():
            L.ggq_mul_mat_q_pretiled(vp(w), vp(scratch), vp(y), Q4_K, 1, BATCH, K_DIM, N_DIM, N_DIM, cur_stream())

        us_med, us_min = time_launches(mmq_only, 200, use_graph=not args.eager)
        achieved = bytes_per_step / (us_med * 1e-6) / 1e9
        traffic = None  # PMC-derived HBM bytes per launch, measured offline with rocprofv3 (profiles/)
        try:
            traffic = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))["mmq_q4_k_batch128"]["hbm_bytes_per_launch"]
        except Exception:
            pass
        out["roofline"] = {"bound": "hbm", "kernel": "ggq::mmq_stream_kernel<Q4_K, f16, TB=2> (32 rows x 64 tokens x 4 K-slices per workgroup)",
                           "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                           "avg_launch_us": round(us_med, 3), "min_launch_us": round(us_min, 3),
                           "algorithmic_bytes_per_launch": bytes_per_step,
                           "int8_mfma_TOPs": round(2.0 * BATCH * N_DIM * K_DIM / (us_med * 1e-6) / 1e12, 2),
                           "note": "duration = HIP-event time of 200 back-to-back launches / 200 on the launch stream"}
        if not args.no_extra:
            out["extra"] = secondary_configs(L, dev, w, x, scratch, args)
            out["cpu_baseline"] = cpu_baseline()
            ref = reference_cpu_dequant()
            if ref:
                out["cpu_reference_dequant_q4_0_4096x4096"] = ref
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def secondary_configs(L, dev, w_q4k, x128, scratch, args):
    """The other BASELINE configs on the same GPU, kernel-level (HIP events, graph replay)."""
    from ggq import synth
    res = {}
    g = not args.eager

    def rec(name, us, nbytes, ops=None):
        r = {"us": round(us, 3), "GB/s": round(nbytes / (us * 1e-6) / 1e9, 1),
             "pct_hbm_roofline": round(100 * nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 2)}
        if ops:
            r["TOP/s"] = round(ops / (us * 1e-6) / 1e12, 2)
        res[name] = r

    out16 = torch.empty((N_DIM, K_DIM), dtype=torch.float16, device=dev)
    ws = {Q4_K: w_q4k}
    for t in (Q4_0, Q8_0, Q5_K, Q6_K):
        ws[t] = torch.from_numpy(synth.random_weight(t, N_DIM, K_DIM, seed=1)).to(dev)
    names = {Q4_K: "Q4_K", Q4_0: "Q4_0", Q8_0: "Q8_0", Q5_K: "Q5_K", Q6_K: "Q6_K"}
    # config 2: dequantise 11008 x 4096 -> fp16
    for t in (Q4_0, Q8_0, Q4_K):
        us, _ = time_launches(lambda: L.ggq_dequantize_f16(vp(ws[t]), vp(out16), t, N_DIM, K_DIM, cur_stream()), 50, use_graph=g)
        rec(f"dequantize_{names[t]}_11008x4096", us, algo_bytes_dequant(t, N_DIM, K_DIM))
    # config 3: MMVQ batch 1 (quantize_q8_1 + mul_mat_vec_q)
    x1 = x128[:1].contiguous()
    y1 = torch.empty((1, N_DIM), dtype=torch.float16, device=dev)
    sc1 = torch.empty(int(L.ggq_mmvq_scratch_bytes(K_DIM)), dtype=torch.uint8, device=dev)
    for t in (Q4_0, Q4_K):
        us, _ = time_launches(lambda: L.ggq_mul_mat_vec_q(vp(ws[t]), vp(x1), vp(y1), t, 1, K_DIM, N_DIM, vp(sc1), cur_stream()), 200, use_graph=g)
        rec(f"mmvq_{names[t]}_batch1", us, algo_bytes_matmul(t, N_DIM, K_DIM, 1), 2.0 * N_DIM * K_DIM)
    # config 4: MMQ batch 128 other formats; batch 8 for Q4_K and Q8_0
    y128 = torch.empty((BATCH, N_DIM), dtype=torch.float16, device=dev)
    for t in (Q5_K, Q6_K, Q8_0, Q4_0):
        us, _ = time_launches(lambda: L.ggq_mul_mat_q(vp(ws[t]), vp(x128), vp(y128), t, 1, BATCH, K_DIM, N_DIM, vp(scratch), cur_stream()), 100, use_graph=g)
        rec(f"mmq_{names[t]}_batch128", us, algo_bytes_matmul(t, N_DIM, K_DIM, BATCH), 2.0 * BATCH * N_DIM * K_DIM)
    x8 = x128[:8].contiguous()
    for t in (Q4_K, Q8_0):
        us, _ = time_launches(lambda: L.ggq_mul_mat_q(vp(ws[t]), vp(x8), vp(y128), t, 1, 8, K_DIM, N_DIM, vp(scratch), cur_stream()), 100, use_graph=g)
        rec(f"mmq_{names[t]}_batch8", us, algo_bytes_matmul(t, N_DIM, K_DIM, 8), 2.0 * 8 * N_DIM * K_DIM)
    us, _ = time_launches(lambda: L.ggq_quantize_q8_1_mmq(vp(x128), 1, vp(scratch), BATCH, K_DIM, Q4_K, cur_stream()), 200, use_graph=g)
    res["quantize_mmq_q8_1_batch128"] = {"us": round(us, 3)}
    us, _ = time_launches(lambda: L.ggq_quantize_q8_1_tiled(vp(x128), 1, vp(scratch), BATCH, K_DIM, Q4_K, cur_stream()), 200, use_graph=g)
    res["quantize_q8_1_tiled_batch128"] = {"us": round(us, 3)}
    # BASELINE configs[4]: the per-rank shard of Q4_K 8192 x 28672 over 8 GPUs = 3584 rows x K 8192, batch 1 / 8 / 128
    K5, N5 = 8192, 3584
    w5 = torch.from_numpy(synth.random_weight(Q4_K, N5, K5, seed=5)).to(dev)
    x5 = torch.randn((BATCH, K5), generator=torch.Generator().manual_seed(5)).half().to(dev)
    y5 = torch.empty((BATCH, N5), dtype=torch.float16, device=dev)
    sc5 = torch.empty(int(L.ggq_mmq_scratch_bytes(BATCH, K5)), dtype=torch.uint8, device=dev)
    for b in (8, BATCH):
        xb = x5[:b].contiguous()
        us, _ = time_launches(lambda: L.ggq_mul_mat_q(vp(w5), vp(xb), vp(y5), Q4_K, 1, b, K5, N5, vp(sc5), cur_stream()), 100, use_graph=g)
        rec(f"mmq_Q4_K_shard_3584x8192_batch{b}", us, algo_bytes_matmul(Q4_K, N5, K5, b), 2.0 * b * N5 * K5)
    x51 = x5[:1].contiguous()
    us, _ = time_launches(lambda: L.ggq_mul_mat_vec_q(vp(w5), vp(x51), vp(y5), Q4_K, 1, K5, N5, vp(sc5), cur_stream()), 200, use_graph=g)
    rec("mmvq_Q4_K_shard_3584x8192_batch1", us, algo_bytes_matmul(Q4_K, N5, K5, 1), 2.0 * N5 * K5)
    # FFN gate + up on one activation quantisation (ggq.linear): quantise once, two streamed matmuls
    def gate_up():
        L.ggq_quantize_q8_1_tiled(vp(x128), 1, vp(scratch), BATCH, K_DIM, Q4_K, cur_stream())
        L.ggq_mul_mat_q_pretiled(vp(w_q4k), vp(scratch), vp(y128), Q4_K, 1, BATCH, K_DIM, N_DIM, N_DIM, cur_stream())
        L.ggq_mul_mat_q_pretiled(vp(ws[Q5_K]), vp(scratch), vp(y128), Q5_K, 1, BATCH, K_DIM, N_DIM, N_DIM, cur_stream())
    us, _ = time_launches(gate_up, 100, use_graph=g)
    res["gate_up_Q4_K_Q5_K_batch128_shared_quantisation"] = {"us": round(us, 3), "note": "two separate ggml_mul_mat_a8 calls: mmq_Q4_K + mmq_Q5_K step times"}
    # the reference-layout kernel (other formats' path) on the headline shape, for comparison
    L.ggq_quantize_q8_1_mmq(vp(x128), 1, vp(scratch), BATCH, K_DIM, Q4_K, cur_stream())
    us, _ = time_launches(lambda: L.ggq_mul_mat_q_prequant(vp(w_q4k), vp(scratch), vp(y128), Q4_K, 1, BATCH, K_DIM, N_DIM, N_DIM, cur_stream()), 100, use_graph=g)
    rec("mmq_Q4_K_batch128_lds_tile_kernel_only", us, algo_bytes_matmul(Q4_K, N_DIM, K_DIM, BATCH), 2.0 * BATCH * N_DIM * K_DIM)
    return res


if __name__ == "__main__":
    main()
