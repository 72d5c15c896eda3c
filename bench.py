#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native ggml block-quant hot path.

Metric (BASELINE.json): "Q4_K MMQ GEMV/GEMM GB/s + % HBM roofline, 4096x11008, batch 1/128".
One *step* = one pass of the hot path over one batch of synthetic input:
    ggml_mul_mat_a8(W[Q4_K, N=11008 x K=4096], X[128, 4096] fp16)  =  quantize_mmq_q8_1 + mul_mat_q
with every input already resident in HBM.  `value` = algorithmic bytes (quantised W + X + Y,
SURVEY.md §8d) of all steps of all ranks / wall time of the timed region, in GB/s.

Cache state: consecutive steps use DIFFERENT weight tensors — a ring of N_COLD distinct
25 MB copies (> 256 MB Infinity Cache + 32 MB L2 in total), the way the layers of a model follow each other — so
every step streams its weights from HBM ("cold").  The same loop over ONE tensor (weights resident in
L2 + Infinity Cache, "warm") is reported beside it as `value_warm`; every `extra` entry carries both states
with median / p10 / p90 over repeated timings.

    python bench.py [--gpus N] [--steps K] [--warmup W]
For N > 1 the driver starts it under torch.distributed.run (one rank per GPU, RCCL): every rank
owns an 11008-row shard (weak scaling) and the [128, 11008] slabs are all-gathered each step
on a side stream, overlapped with the next step's compute.  Rank 0 additionally reports the
strong-scaling table of BASELINE configs[4] (Q4_K 28672 x 8192 row-sharded P ways, batch 1 / 8 / 128:
kernel, collective and end-to-end times separately) under "strong_scaling_config5".

Prints ONE JSON line on rank 0.  Besides the contract keys it carries
  "roofline":     dominant kernel (mul_mat_q) vs the 8 TB/s HBM roof, duration measured live
                  with HIP events on the launch stream,
  "cpu_baseline": the oracle (CPU port of the reference algorithm) timed on this box's host cores,
  "extra":        the other BASELINE configs (batch-1 MMVQ, dequantise, other formats).
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 measured achievable
INT8_PEAK_TOPS = 5000.0  # dense int8 MFMA peak (same guide: 2x the ~2.5 PF bf16 rate)
# What a block-quantised GEMM can reach of that peak on this SIMD: every 32x32x32 int8 tile (13.4 ns of matrix pipe at the nominal
# peak, 15.8 ns measured) is followed by two exact FMAs per (row, token, 32-group) triple = 16 v_pk_fma_f32 = 30.5 ns, and the SIMD
# does not overlap the two (scripts/ubench_mfma.hip, ubench_overlap.hip): 13.4 / (15.8 + 30.5) = 0.29.
SCALED_INT8_CEILING = 0.29
Q4_K, Q5_K, Q6_K, Q4_0, Q8_0 = 12, 13, 14, 2, 8
NAMES = {Q4_K: "Q4_K", Q4_0: "Q4_0", Q8_0: "Q8_0", Q5_K: "Q5_K", Q6_K: "Q6_K"}
K_DIM, N_DIM, BATCH = 4096, 11008, 128
COLD_BYTES = 352 << 20   # distinct bytes a "cold" ring must span: 256 MiB Infinity Cache + 32 MiB L2 + margin
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "r04_traffic.json")
ROOFLINE_KERNEL = "ggq::mmq_x64_kernel<Q4_K,f16,KS=4,R3=true>"   # 96-row units: 230 workgroups of eight waves at the headline shape
KERNEL_SOURCES = ("mmq_x64.hip", "mmq_x64_loops.inc")   # the traffic figure is refused when these changed since it was measured


def kernel_source_sha():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "ggml-libtorch_amd", "csrc", "hip", f), "rb").read())
    return h.hexdigest()
MAX_LINE_BYTES = 4096    # the driver reads the LAST stdout line; r03's 21.7 KB line did not parse (BENCH_r03.json parsed: null)
EXTRA_JSON = "bench_extra.json"


def compact_line(out, detail):
    """The ONE JSON line the driver parses: the contract keys + `roofline` + `cpu_baseline` + a few scalar extras, bounded
    by MAX_LINE_BYTES.  Everything else (`detail`: per-config warm / cold tables, strong-scaling leg, the other CPU legs) goes to
    bench_extra.json and to stderr.  Pure function of two dicts: tests/test_host_logic.py serialises a fully populated pair."""
    line = dict(out)
    ex = detail.get("extra") or {}
    summary = {}
    for name, key in (("mmvq_Q4_K_batch1", "gemv_Q4_K_b1"), ("mmq_Q4_K_batch8", "mmq_Q4_K_b8"), ("mmq_Q4_K_batch32", "mmq_Q4_K_b32"),
                      ("mmq_Q8_0_batch128", "mmq_Q8_0_b128"), ("mmq_Q6_K_batch128", "mmq_Q6_K_b128"),
                      ("dequantize_Q4_K_11008x4096", "dequant_Q4_K"), ("dequantize_Q4_0_11008x4096", "dequant_Q4_0"),
                      ("dequantize_Q8_0_11008x4096", "dequant_Q8_0")):
        e = ex.get(name)
        if isinstance(e, dict) and "cold" in e:
            summary[key] = {"us_cold": e["cold"].get("us"), "us_warm": e.get("us"), "pct_hbm_cold": e["cold"].get("pct_hbm_roofline")}
            if "pct_int8_mfma_peak" in e["cold"]:
                summary[key]["pct_int8_cold"] = e["cold"]["pct_int8_mfma_peak"]
    if summary:
        line["summary_us"] = summary
    lb = detail.get("large_batch") or {}
    if lb:   # the reference benchmark's default regime: op us + speed-up over dequantise + rocBLAS
        line["large_batch"] = {k.replace("mmq_", "").replace("batch", "b"): [v.get("us"), v.get("speedup_vs_dequantize_plus_rocblas")]
                               for k, v in lb.items() if isinstance(v, dict)}
    cb = detail.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample") if k in cb}
        if isinstance(cb.get("threads"), dict):
            line["cpu_baseline"]["threads_value"] = cb["threads"].get("value")
            line["cpu_baseline"]["threads_cores"] = cb["threads"].get("cores")
    line["extra_file"] = EXTRA_JSON
    s = json.dumps(line, separators=(",", ":"))
    if len(s) > MAX_LINE_BYTES:   # never print an unparseable line: drop the optional parts, largest first
        for k in ("summary_us", "large_batch", "value_gpu_events", "ms_per_step_gpu_events"):
            line.pop(k, None)
            s = json.dumps(line, separators=(",", ":"))
            if len(s) <= MAX_LINE_BYTES:
                break
    assert len(s) <= MAX_LINE_BYTES and "\n" not in s, len(s)
    return s


def write_extra(out, detail):
    """bench_extra.json (repo root, and gpurun_out/ when present so it travels back from the GPU box) + one stderr line"""
    full = dict(out)
    full.update(detail)
    text = json.dumps(full, indent=1)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, EXTRA_JSON), "w") as f:
                    f.write(text)
        except OSError:   # read-only checkout: the compact line is still printed
            pass
    print("[bench-extra] " + json.dumps(full), file=sys.stderr)


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


def ring_of(t, nbytes_each, min_n=2):
    """distinct device copies of tensor t whose total size exceeds the caches (cold ring)"""
    n = max(min_n, -(-COLD_BYTES // max(1, nbytes_each)) + 1)
    return [t] + [t.clone() for _ in range(n - 1)]


def time_launches(fn, iters, reps=9, use_graph=True):
    """Device time of one call of fn(i): `iters` back-to-back launches (i = 0..iters-1) bracketed by HIP
    events recorded on the stream the kernels run on (torch's current stream), repeated `reps` times.
    Returns {"us": median, "p10", "p90", "min"} of the per-launch averages."""
    fn(0)
    torch.cuda.synchronize()
    runner = None
    if use_graph:
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(iters):
                    fn(i)
            runner = g.replay
        except Exception:  # pragma: no cover - capture unsupported: fall back to eager launches
            runner = None
            torch.cuda.synchronize()
    if runner is None:
        def runner():
            for i in range(iters):
                fn(i)
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
    a = np.asarray(times)
    return {"us": round(float(np.median(a)), 3), "p10": round(float(np.percentile(a, 10)), 3),
            "p90": round(float(np.percentile(a, 90)), 3), "min": round(float(a.min()), 3)}


def rates(stats, nbytes, ops=None):
    us = stats["us"]
    r = dict(stats)
    r["GB/s"] = round(nbytes / (us * 1e-6) / 1e9, 1)
    r["pct_hbm_roofline"] = round(100 * nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 2)
    if ops:
        r["TOP/s"] = round(ops / (us * 1e-6) / 1e12, 2)
        r["pct_int8_mfma_peak"] = round(100 * ops / (us * 1e-6) / 1e12 / INT8_PEAK_TOPS, 2)
        r["pct_of_per_group_scaling_ceiling"] = round(r["pct_int8_mfma_peak"] / SCALED_INT8_CEILING, 2)   # of the 29 % such a GEMM can reach
    return r


# ------------------------------------------------------------------------------------------------ CPU baselines
def _threads_available():
    try:
        return len(os.sched_getaffinity(0))
    except Exception:  # pragma: no cover
        return os.cpu_count() or 1


def cpu_baseline(sample_rows=N_DIM):
    """The oracle (C port of the reference's mul_mat_q algorithm) on the host cores, on a bounded sample of the same
    workload: `sample_rows` weight rows x all 128 tokens — one core (the reference's CPU code is single-threaded,
    ggml-cpu/ggml-quants.hpp) and, beside it, the same loop row-partitioned over T threads."""
    from oracle import oracle as O
    from ggq import synth
    w = np.ascontiguousarray(synth.random_weight(Q4_K, sample_rows, K_DIM, seed=0))
    x = torch.randn((BATCH, K_DIM), generator=torch.Generator().manual_seed(0)).half().float().numpy()
    lib = O.lib()
    q8 = O.quantize_q8_1_mmq(x, Q4_K)

    def run(r0, r1, y):
        rc = lib.oracle_mul_mat_q(Q4_K, O._p(w[r0:r1]), O._p(q8), O._p(y), None, BATCH, K_DIM, r1 - r0)
        assert rc == 0

    y = np.empty((BATCH, sample_rows), np.float32)
    t0 = time.perf_counter()
    run(0, sample_rows, y)
    dt1 = time.perf_counter() - t0
    nbytes = algo_bytes_matmul(Q4_K, sample_rows, K_DIM, BATCH)
    out = {"value": round(nbytes / dt1 / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
           "sample": f"oracle_mul_mat_q (quantised X given), {sample_rows} of {N_DIM} Q4_K rows x {BATCH} tokens, "
                     f"K={K_DIM}, 1 pass, {dt1:.2f} s", "host_cores_available": _threads_available()}
    T = min(_threads_available(), 64)
    if T > 1:
        per = -(-sample_rows // T)
        spans = [(i * per, min(sample_rows, (i + 1) * per)) for i in range(T) if i * per < sample_rows]
        ys = [np.empty((BATCH, b - a), np.float32) for a, b in spans]
        th = [threading.Thread(target=run, args=(a, b, yy)) for (a, b), yy in zip(spans, ys)]   # ctypes drops the GIL
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dtT = time.perf_counter() - t0
        out["threads"] = {"value": round(nbytes / dtT / 1e9, 4), "unit": "GB/s", "cores": len(spans),
                          "sample": f"the same rows partitioned over {len(spans)} threads, {dtT:.2f} s"}
    # the strongest host implementation we have: the product's CPU twin (ggq_cpu_mul_mat_q: AVX-512 VNNI / AVX2 byte dots,
    # the same float sequence), checked here bit for bit against the oracle's output above
    try:
        from ggq import lib as ggqlib
        C = ggqlib.cpu()
        simd = C.ggq_cpu_mmq_simd_name().decode()
        y2 = np.empty((BATCH, sample_rows), np.float32)
        for nt in sorted({1, T}):
            y2[:] = np.nan
            t0 = time.perf_counter()
            rc = C.ggq_cpu_mul_mat_q(O._p(w), O._p(q8), O._p(y2), Q4_K, BATCH, K_DIM, sample_rows, nt, 1)
            dt = time.perf_counter() - t0
            assert rc == 0
            out[f"simd_twin_{nt}_threads"] = {
                "value": round(nbytes / dt / 1e9, 4), "unit": "GB/s", "cores": nt, "kind": "product (ggq_cpu_mul_mat_q)",
                "simd": simd, "bit_identical_to_oracle": bool(np.array_equal(y.view(np.uint32), y2.view(np.uint32))),
                "sample": f"the same {sample_rows} rows x {BATCH} tokens, {dt * 1e3:.1f} ms"}
    except Exception as e:  # pragma: no cover
        out["simd_twin"] = {"error": str(e)[:200]}
    return out


def cpu_config1():
    """BASELINE configs[0]: Q4_0 dequantise 4096 x 4096 on the host — the reference's own compiled ggml-cpu op
    (oracle/_ref, when it travelled), the product's custom_ops twin on 1 thread and on T threads."""
    res = {}
    from ggq import synth, lib as ggqlib
    m = n = 4096
    w_np = synth.random_weight(Q4_0, m, n, seed=0)
    nb = m * n // 32 * 18 + m * n * 4
    try:
        from oracle import oracle as O
        ref = O.load_reference_cpu_op()
        if ref is not None:
            w = torch.from_numpy(w_np)
            ref.ggml_dequantize(w, Q4_0, m, n)
            t0 = time.perf_counter()
            ref.ggml_dequantize(w, Q4_0, m, n)
            dt = time.perf_counter() - t0
            res["reference_ggml_cpu_op"] = {"value": round(nb / dt / 1e9, 3), "unit": "GB/s", "cores": 1, "kind": "reference",
                                            "sample": f"reference custom_ops.ggml_dequantize Q4_0 4096x4096 -> fp32, {dt * 1e3:.1f} ms"}
    except Exception as e:  # pragma: no cover
        res["reference_ggml_cpu_op"] = {"error": str(e)[:200]}
    try:
        C = ggqlib.cpu()
        out = np.empty((m, n), np.float32)
        wb = np.ascontiguousarray(w_np)
        simd = C.ggq_cpu_simd_name().decode()
        for nt, sv in ((1, 0), (1, 1), (min(_threads_available(), 64), 1)):
            C.ggq_cpu_dequantize_f32_ex(wb.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), Q4_0, m, n, nt, sv)
            t0 = time.perf_counter()
            rc = C.ggq_cpu_dequantize_f32_ex(wb.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), Q4_0, m, n, nt, sv)
            dt = time.perf_counter() - t0
            assert rc == 0
            res[f"product_custom_ops_{nt}_threads_{simd if sv else 'scalar'}"] = {
                "value": round(nb / dt / 1e9, 3), "unit": "GB/s", "cores": nt, "kind": "product",
                "sample": f"ggq_cpu_dequantize_f32_ex Q4_0 4096x4096 -> fp32, {'vector (' + simd + ')' if sv else 'scalar loops'}, {dt * 1e3:.1f} ms"}
    except Exception as e:  # pragma: no cover
        res["product_custom_ops"] = {"error": str(e)[:200]}
    return res


def cpu_torch_matmul():
    """What the reference's tests compute as ground truth (HK/tests/kernels/test_cuda_kernels.py:71-74): x @ dequant(W).T in
    fp32 torch on the host (threads = torch.get_num_threads()); the dequantised matrix is given (not timed)."""
    try:
        from oracle import oracle as O
        from ggq import synth
        rows = 2048   # bounded sample of the 11008 rows
        w = synth.random_weight(Q4_K, rows, K_DIM, seed=0)
        wd = torch.from_numpy(O.dequantize_f16(w, Q4_K, rows * K_DIM).astype(np.float32).reshape(rows, K_DIM))
        x = torch.randn((BATCH, K_DIM), generator=torch.Generator().manual_seed(0))
        x @ wd.T
        t0 = time.perf_counter()
        x @ wd.T
        dt = time.perf_counter() - t0
        return {"value": round(2.0 * BATCH * rows * K_DIM / dt / 1e9, 1), "unit": "GFLOP/s", "cores": torch.get_num_threads(),
                "sample": f"fp32 torch x[128,4096] @ dequant(W)[{rows},4096].T on the host, {dt * 1e3:.1f} ms"}
    except Exception as e:  # pragma: no cover
        return {"error": str(e)[:200]}


# ------------------------------------------------------------------------------------------------ config 5
def strong_scaling_config5(L, dev, world, rank, dist):
    """BASELINE configs[4]: Q4_K 28672 x 8192 row-sharded over P = world ranks; batch 1 / 8 / 128.
    Rank r multiplies its 28672/P rows and writes the slab straight into slot r of a [P, batch, N/P] buffer
    (ggq_mul_mat_q_ld / ggq_mul_mat_vec_q write through a row pitch, no staging copy), which is then all-gathered
    IN PLACE (input = the rank's own slot of the output).  Kernel, collective and end-to-end are timed separately
    with HIP events on the launch stream, eager launches, max over ranks."""
    from ggq import synth
    from ggq.dist import shard_rows
    N5, K5 = 28672, 8192
    s, e = shard_rows(N5, world, rank)
    rows = e - s
    w5 = torch.from_numpy(synth.random_weight(Q4_K, rows, K5, seed=100 + rank)).to(dev)
    res = {"P": world, "rows_per_rank": rows, "k": K5, "quant_type": "Q4_K",
           "backend": (dist.get_backend() if world > 1 else "none (single rank)"),
           "world_size_seen_by_torch_distributed": (dist.get_world_size() if world > 1 else 1)}
    for b in (1, 8, BATCH):
        x = torch.randn((b, K5), generator=torch.Generator().manual_seed(5)).half().to(dev)
        buf = torch.empty((world, b, rows), dtype=torch.float16, device=dev)
        mine = buf[rank]
        sc = torch.empty(max(int(L.ggq_mmq_scratch_bytes(b, K5)), int(L.ggq_mmvq_scratch_bytes(K5))), dtype=torch.uint8, device=dev)

        def kernel():
            if b == 1:
                rc = L.ggq_mul_mat_vec_q(vp(w5), vp(x), vp(mine), Q4_K, 1, K5, rows, vp(sc), cur_stream())
            else:
                rc = L.ggq_mul_mat_q_ld(vp(w5), vp(x), vp(mine), Q4_K, 1, b, K5, rows, rows, vp(sc), cur_stream())
            assert rc == 0, rc

        def collective():
            if world > 1:
                dist.all_gather_into_tensor(buf.view(world * b, rows), mine)

        def both():
            kernel()
            collective()

        entry = {}
        for name, fn in (("kernel", kernel), ("collective", collective), ("end_to_end", both)):
            if name == "collective" and world == 1:
                entry[name] = {"us": 0.0}
                continue
            if name == "end_to_end" and world == 1:   # no collective: the same launches, not timed a second time (two eager
                entry[name] = dict(entry["kernel"])   # timings of one thing differ by host jitter and contradicted each other in r03)
                continue
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            iters = 50
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            e1.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / iters
            if world > 1:
                tt = torch.tensor([us], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                us = float(tt.item())
            entry[name] = {"us": round(us, 2)}
        if world > 1 and os.environ.get("GGQ_BENCH_PEER", "0") == "1":   # opt-in until it has passed once on a real multi-GPU node: a GPU fault here is
            # not catchable and would take the RCCL columns of the default scaling run with it (advisor, round 3)
            # the peer-mapped direct-write gather (ggq.dist.PeerSlabGather: HIP IPC mapping, one scatter kernel that stores the
            # slab into every peer's buffer over xGMI and publishes a device flag, one wait kernel; no RCCL, no host barrier)
            # beside the RCCL column.  Never run across GPUs by the build that wrote it (one-GPU box): every step is guarded,
            # and the ranks exchange an ok flag so that a failure on one rank cannot desynchronise the collectives that follow.
            err, us_peer, pg = None, None, None
            try:
                from ggq.dist import PeerSlabGather
                pg = PeerSlabGather(b, N5, torch.float16, dev)

                def peer_step():
                    if b == 1:
                        rc = L.ggq_mul_mat_vec_q(vp(w5), vp(x), vp(pg.local), Q4_K, 1, K5, rows, vp(sc), cur_stream())
                        assert rc == 0, rc
                        pg.gather()
                    else:   # the GEMM's own stores to every rank's slot + flags from its last workgroup where the 16-token tiles
                        pg.matmul_gather(x, w5, Q4_K, sc)   # serve the batch (8); ggq_mul_mat_q_ld + scatter kernel otherwise (128)
                for _ in range(3):
                    peer_step()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    peer_step()
                e1.record()
                e1.synchronize()
                us_peer = e0.elapsed_time(e1) * 1e3 / 20
                if pg.status() != 0:
                    err = "ggq_peer_wait timed out on a peer"
            except Exception as ex:  # pragma: no cover
                err = repr(ex)[:300]
            errs = [None] * world
            dist.all_gather_object(errs, err)
            try:
                if pg is not None:
                    pg.close()
            except Exception as ex:  # pragma: no cover
                errs.append(repr(ex)[:200])
            if any(e is not None for e in errs):
                entry["end_to_end_peer_write"] = {"error": [e for e in errs if e is not None][:2]}
            else:
                tt = torch.tensor([us_peer], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                entry["end_to_end_peer_write"] = {"us": round(float(tt.item()), 2),
                                                  "timing": "HIP events; batch 8: the GEMM kernel stores into every rank's slot itself and publishes the flags (ggq_mul_mat_q_gather), others: kernel + scatter kernel (peer stores + device flag); + wait kernel; 20 iterations, max over ranks"}
        nbytes = algo_bytes_matmul(Q4_K, N5, K5, b)
        entry["end_to_end"]["GB/s_whole_job"] = round(nbytes / (entry["end_to_end"]["us"] * 1e-6) / 1e9, 1)
        entry["message_bytes_per_rank"] = b * rows * 2
        entry["timing"] = "eager launches, HIP events, 50 iterations, max over ranks (includes host launch overhead at batch 1)"
        res[f"batch{b}"] = entry
    if world == 1:
        # on a one-GPU box: what ONE rank of the 8-way split computes (3584 rows), kernel only, graph-timed — the
        # compute side of the P = 8 column; its collective needs the 8-GPU node
        rows8 = N5 // 8
        w8 = w5[:rows8]
        proj = {}
        for b in (1, 8, BATCH):
            x = torch.randn((b, K5), generator=torch.Generator().manual_seed(5)).half().to(dev)
            y = torch.empty((b, rows8), dtype=torch.float16, device=dev)
            sc = torch.empty(max(int(L.ggq_mmq_scratch_bytes(b, K5)), int(L.ggq_mmvq_scratch_bytes(K5))), dtype=torch.uint8, device=dev)
            if b == 1:
                fn = lambda i: L.ggq_mul_mat_vec_q(vp(w8), vp(x), vp(y), Q4_K, 1, K5, rows8, vp(sc), cur_stream())
            else:
                fn = lambda i: L.ggq_mul_mat_q(vp(w8), vp(x), vp(y), Q4_K, 1, b, K5, rows8, vp(sc), cur_stream())
            proj[f"batch{b}"] = rates(time_launches(fn, 104), algo_bytes_matmul(Q4_K, rows8, K5, b), 2.0 * b * rows8 * K5)
            proj[f"batch{b}"]["cache_state"] = "warm: one 16.5 MB shard"
        res["one_rank_of_8_shard_3584x8192_kernel_only"] = proj
    return res


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

    # ---- workload: per-rank shard, inputs resident in HBM; a ring of distinct weight copies for the cold steps ----
    w = torch.from_numpy(synth.random_weight(Q4_K, N_DIM, K_DIM, seed=rank)).to(dev)
    w_ring = ring_of(w, w.numel())
    x = torch.randn((BATCH, K_DIM), generator=torch.Generator().manual_seed(0)).half().to(dev)
    scratch = torch.empty(int(L.ggq_mmq_scratch_bytes(BATCH, K_DIM)), dtype=torch.uint8, device=dev)
    bytes_per_step = algo_bytes_matmul(Q4_K, N_DIM, K_DIM, BATCH)

    def barrier():
        if world > 1:
            dist.barrier()

    value_warm = None
    gpu_side = [None]
    if world == 1:
        y = torch.empty((BATCH, N_DIM), dtype=torch.float16, device=dev)

        def step(i, ring=w_ring):
            rc = L.ggq_mul_mat_q(vp(ring[i % len(ring)]), vp(x), vp(y), Q4_K, 1, BATCH, K_DIM, N_DIM, vp(scratch), cur_stream())
            assert rc == 0, rc

        def timed(ring):
            for i in range(args.warmup):
                step(i, ring)
            torch.cuda.synchronize()
            graph = None
            if not args.eager:
                try:
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        for i in range(args.steps):
                            step(i, ring)
                    graph.replay()  # instantiate / upload once, untimed
                    torch.cuda.synchronize()
                except Exception as e:  # pragma: no cover
                    print(f"[bench] graph capture failed ({e}); timing eager launches", file=sys.stderr)
                    graph = None
                    torch.cuda.synchronize()
            done = torch.cuda.Event(enable_timing=True)
            began = torch.cuda.Event(enable_timing=True)
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            began.record()
            if graph is not None:
                graph.replay()
            else:
                for i in range(args.steps):
                    step(i, ring)
            # poll before the blocking synchronize: a sleeping host thread is woken 10-30 us after the GPU is done, which at
            # the driver's --steps 20 (0.6 ms of GPU work) is 2-5 % of the timed region; the synchronize still brackets it
            done.record()
            while not done.query():
                pass
            torch.cuda.synchronize()
            barrier()
            wall = time.perf_counter() - t0
            gpu_side[0] = began.elapsed_time(done) * 1e-3   # the same steps by HIP events: what the host bracket adds is launch + wake-up
            return wall, graph is not None

        gpu_side = [None]
        elapsed_warm, _ = timed([w])          # untimed for the headline: the warm twin, reported beside it
        value_warm = bytes_per_step * args.steps / elapsed_warm / 1e9
        elapsed, graphed = timed(w_ring)      # the headline: EXACTLY args.steps steps, weights streamed from HBM
        launch_mode = "hipGraph replay of all steps" if graphed else "eager launches"
    else:
        # rank-local slab + all-gather on a side stream, double buffered
        comm = torch.cuda.Stream(device=dev)
        ys = [torch.empty((BATCH, N_DIM), dtype=torch.float16, device=dev) for _ in range(2)]
        # gather buffers in the [P, batch, rows] form (slot r = rank r's [batch, rows] slab; ggq.dist.unpermute_gathered
        # turns it into [batch, P * rows] where a consumer needs that layout)
        gathered = [torch.empty((world * BATCH, N_DIM), dtype=torch.float16, device=dev) for _ in range(2)]
        full = [torch.empty((BATCH, world * N_DIM), dtype=torch.float16, device=dev) for _ in range(2)]   # [batch, P * rows]: what a consumer reads
        # (events are created once and re-recorded: per-step host overhead matters at ~30 us of GPU work per step)
        ready_ev = [torch.cuda.Event(), torch.cuda.Event()]
        done_ev = [torch.cuda.Event(), torch.cuda.Event()]
        used = [False, False]
        main_stream = torch.cuda.current_stream()

        def step(i):
            b = i & 1
            if used[b]:
                main_stream.wait_event(done_ev[b])  # slab b is free again
            rc = L.ggq_mul_mat_q(vp(w_ring[i % len(w_ring)]), vp(x), vp(ys[b]), Q4_K, 1, BATCH, K_DIM, N_DIM, vp(scratch), cur_stream())
            assert rc == 0, rc
            ready_ev[b].record(main_stream)
            comm.wait_event(ready_ev[b])
            with torch.cuda.stream(comm):
                dist.all_gather_into_tensor(gathered[b], ys[b])
                # [P, batch, rows] -> [batch, P * rows] inside the timed region (one permute copy on the side stream)
                full[b].view(BATCH, world, N_DIM).copy_(gathered[b].view(world, BATCH, N_DIM).permute(1, 0, 2))
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
        launch_mode = "eager launches, all-gather + un-permute to [batch, P * rows] overlapped on a side stream"
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    ms_per_step = elapsed * 1e3 / args.steps
    value = world * bytes_per_step * args.steps / elapsed / 1e9

    detail = {}
    out = {
        "metric": "Q4_K MMQ GEMV/GEMM GB/s + % HBM roofline, 4096x11008, batch 1/128",
        "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 6), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int8",
        "data": "synthetic",
        "config": {"workload": "ggml_mul_mat_a8 = quantize_mmq_q8_1 + mul_mat_q, Q4_K W[11008 x 4096] per GPU, "
                               "X[128 x 4096] fp16 (BASELINE configs[3] shape, the one the metric is quoted on)",
                   "arithmetic": "int8 x int8 -> int32 (MFMA), fp32 scale/accumulate, fp16 in/out",
                   "quant_type": "Q4_K", "k": K_DIM, "n_rows_per_gpu": N_DIM, "batch": BATCH,
                   "algorithmic_bytes_per_step_per_gpu": bytes_per_step,
                   "parallelism": f"row-shard x{world}" + (" + RCCL all-gather of [128 x 11008] slabs" if world > 1 else ""),
                   "launch": launch_mode,
                   "cache_state": f"cold: consecutive steps cycle {len(w_ring)} distinct weight tensors "
                                  f"({len(w_ring) * w.numel() >> 20} MiB > 256 MiB Infinity Cache + 32 MiB L2)"},
        "pct_hbm_roofline": round(100.0 * value / world / HBM_PEAK_GBS, 2),
    }
    if world == 1 and gpu_side[0]:
        # the same K steps timed by HIP events on the stream: the host bracket above (which `value` uses) additionally contains one
        # graph launch and the host's wake-up, a fixed cost that weighs 5 - 9 % at --steps 20 and < 1 % at --steps 200
        out["ms_per_step_gpu_events"] = round(gpu_side[0] * 1e3 / args.steps, 6)
        out["value_gpu_events"] = round(bytes_per_step * args.steps / gpu_side[0] / 1e9, 2)
    if value_warm is not None:
        out["value_warm"] = round(value_warm, 2)
        out["config"]["cache_state_value_warm"] = "warm: every step re-reads ONE 25 MB weight tensor (resident in L2 + Infinity Cache)"
    if world > 1:
        out["config"]["rccl"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}

    if rank == 0 and world == 1:
        # ---- roofline of the dominant kernel (mul_mat_q alone, activations pre-quantised), cold weights ----
        # (the fused op quantises into the fragment-major scratch and runs the streamed kernel for Q4_K:
        #  time exactly that kernel, through the exported pre-quantised entry point)
        assert L.ggq_mmq_route(Q4_K, BATCH, K_DIM, N_DIM) == 5, "the headline shape is expected on the 64 x 64 wave-tile kernel (GGQ_MMQ_ROUTE_X64)"
        rc = L.ggq_quantize_q8_1_x64(vp(x), 1, vp(scratch), BATCH, K_DIM, Q4_K, cur_stream())
        assert rc == 0

        def mmq_only(i, ring=w_ring):
            L.ggq_mul_mat_q_x64(vp(ring[i % len(ring)]), vp(scratch), vp(y), Q4_K, 1, BATCH, K_DIM, N_DIM, N_DIM, 0, None, cur_stream())

        st_cold = time_launches(mmq_only, 208, use_graph=not args.eager)
        st_warm = time_launches(lambda i: mmq_only(i, [w]), 208, use_graph=not args.eager)
        achieved = bytes_per_step / (st_cold["us"] * 1e-6) / 1e9
        ops = 2.0 * BATCH * N_DIM * K_DIM
        traffic, traffic_note = None, "profiles/r04_traffic.json absent"
        try:   # PMC-derived HBM bytes per launch, measured offline with rocprofv3 (profiles/); refused when the kernel changed since
            tj = json.load(open(TRAFFIC_JSON))
            if tj.get("kernel_source_sha256") == kernel_source_sha():
                traffic, traffic_note = tj["mmq_q4_k_batch128"]["hbm_bytes_per_launch"], tj["mmq_q4_k_batch128"].get("how", "")
            else:
                traffic_note = "profiles/r04_traffic.json was measured on different kernel sources (sha mismatch): stale, not reported"
        except Exception:
            pass
        # `frac` prices the kernel against the HBM roof (the metric is GB/s + % of the HBM roofline) although at batch 128 it
        # is bound by the SIMD's own instruction issue (vector + matrix work, which do not overlap on this SIMD): `bound` says
        # so, and `frac_int8_mfma_peak` is the same duration against the dense int8 MFMA peak
        out["roofline"] = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                           "kernel": ROOFLINE_KERNEL, "avg_launch_us": st_cold["us"], "cache_state": "cold",
                           "warm_avg_launch_us": st_warm["us"],
                           "warm_frac": round(bytes_per_step / (st_warm["us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                           "algorithmic_bytes_per_launch": bytes_per_step,
                           "frac_int8_mfma_peak": round(ops / (st_cold["us"] * 1e-6) / 1e12 / INT8_PEAK_TOPS, 4)}
        detail["roofline"] = {"traffic_note": traffic_note, "p10_launch_us": st_cold["p10"], "p90_launch_us": st_cold["p90"],
                              "min_launch_us": st_cold["min"], "cache_state": out["config"]["cache_state"],
                              "warm": st_warm, "int8_mfma_TOPs": round(ops / (st_cold["us"] * 1e-6) / 1e12, 2),
                              "int8_ceiling_with_per_group_scales": SCALED_INT8_CEILING,
                              "frac_of_that_ceiling": round(ops / (st_cold["us"] * 1e-6) / 1e12 / INT8_PEAK_TOPS / SCALED_INT8_CEILING, 4),
                              "limiter": "at batch 128 the kernel is bound by the SIMD's own arithmetic (2 exact FMAs per (row, token, 32-group) triple "
                                         "+ the int8 MFMAs), not by HBM; `frac` prices it against the HBM roof because the metric is GB/s + % of the "
                                         "HBM roofline, `frac_int8_mfma_peak` is the same duration against the dense int8 peak (DESIGN.md 5.4)",
                              "note": "duration = HIP-event time of 208 back-to-back launches / 208 on the launch stream, median of 9 repeats"}
        if not args.no_extra:
            detail["extra"] = secondary_configs(L, dev, w, w_ring, x, scratch, args)
            detail["large_batch"] = large_batch_configs(L, dev, args)
            detail["strong_scaling_config5"] = strong_scaling_config5(L, dev, world, rank, dist)
            detail["cpu_baseline"] = cpu_baseline()
            detail["cpu_config1_q4_0_dequant_4096x4096"] = cpu_config1()
            detail["cpu_torch_matmul_on_dequantised"] = cpu_torch_matmul()
    if rank == 0:
        if world == 1:
            write_extra(out, detail)
        sys.stderr.flush()
        print(compact_line(out, detail), flush=True)
    if world > 1 and not args.no_extra and os.environ.get("GGQ_BENCH_STRONG", "1") == "1":
        # AFTER the headline line is on stdout: this leg has never run across GPUs (no multi-GPU box for this build), and a fault or a
        # stuck collective in it must not cost the scaling run its number.  Its results go to bench_extra.json and stderr only.
        ss = strong_scaling_config5(L, dev, world, rank, dist)   # collective: every rank takes part
        if rank == 0:
            detail["strong_scaling_config5"] = ss
    if rank == 0 and world > 1:
        write_extra(out, detail)
        sys.stderr.flush()
    if world > 1:
        dist.destroy_process_group()


def large_batch_configs(L, dev, args):
    """The reference benchmark's own regime (benchmarks/benchmark_mmq.py:152 defaults to 4096 tokens, its tests run 2048):
    quantise + mul_mat_q against dequantise-to-fp16 + rocBLAS fp16 GEMM at 512 / 2048 / 4096 tokens, 11008 x 4096, warm
    (one weight tensor; at these batches the op takes 0.1 - 1 ms and the 25 - 48 MB of weights are a small part of its traffic)."""
    from ggq import synth
    res = {}
    g = not args.eager
    for t in (Q4_K, Q8_0, Q6_K):
        w = torch.from_numpy(synth.random_weight(t, N_DIM, K_DIM, seed=11)).to(dev)
        wd = torch.empty((N_DIM, K_DIM), dtype=torch.float16, device=dev)
        for bb in (512, 2048, 4096):
            x = torch.randn((bb, K_DIM), generator=torch.Generator().manual_seed(bb)).half().to(dev)
            y = torch.empty((bb, N_DIM), dtype=torch.float16, device=dev)
            sc = torch.empty(int(L.ggq_mmq_scratch_bytes(bb, K_DIM)), dtype=torch.uint8, device=dev)
            ops = 2.0 * bb * N_DIM * K_DIM
            nb = algo_bytes_matmul(t, N_DIM, K_DIM, bb)
            iters = 8 if bb >= 2048 else 24

            def mmq(i):
                rc = L.ggq_mul_mat_q(vp(w), vp(x), vp(y), t, 1, bb, K_DIM, N_DIM, vp(sc), cur_stream())
                assert rc == 0, rc

            def deq(i):
                rc = L.ggq_dequantize_f16(vp(w), vp(wd), t, N_DIM, K_DIM, cur_stream())
                assert rc == 0, rc
                torch.matmul(x, wd.t(), out=y)

            e = rates(time_launches(mmq, iters, reps=5, use_graph=g), nb, ops)
            d = time_launches(deq, iters, reps=5, use_graph=False)   # rocBLAS picks its workspace outside a capture
            e["dequantize_plus_rocblas_us"] = d["us"]
            e["speedup_vs_dequantize_plus_rocblas"] = round(d["us"] / e["us"], 3)
            res[f"mmq_{NAMES[t]}_batch{bb}"] = e
            del x, y, sc
        del w, wd
    return res


def secondary_configs(L, dev, w_q4k, w_q4k_ring, x128, scratch, args):
    """The other BASELINE configs on the same GPU, kernel-level (HIP events, graph replay), each warm AND cold."""
    from ggq import synth
    res = {}
    g = not args.eager

    def rec(name, make_fn, rings, nbytes, ops=None, iters=104):
        """make_fn(bufs) -> launch using one buffer of each ring; cold cycles the rings, warm pins their first buffer"""
        cold = time_launches(lambda i: make_fn([r[i % len(r)] for r in rings]), iters, use_graph=g)
        warm = time_launches(lambda i: make_fn([r[0] for r in rings]), iters, use_graph=g)
        entry = rates(warm, nbytes, ops)
        entry["cache_state"] = "warm: one buffer re-used by every launch (L2 + Infinity Cache resident where it fits)"
        entry["cold"] = rates(cold, nbytes, ops)
        span = sum(len(r) * r[0].numel() * r[0].element_size() for r in rings) >> 20
        entry["cold"]["cache_state"] = f"cold: launches cycle distinct buffers spanning {span} MiB"
        res[name] = entry

    ws = {Q4_K: w_q4k}
    for t in (Q4_0, Q8_0, Q5_K, Q6_K):
        ws[t] = torch.from_numpy(synth.random_weight(t, N_DIM, K_DIM, seed=1)).to(dev)
    rings = {t: (w_q4k_ring if t == Q4_K else ring_of(ws[t], ws[t].numel())) for t in ws}
    # config 2: dequantise 11008 x 4096 -> fp16 (the 90 MB output cycles over 4 buffers in the cold run)
    out_ring = [torch.empty((N_DIM, K_DIM), dtype=torch.float16, device=dev) for _ in range(4)]
    for t in (Q4_0, Q8_0, Q4_K):
        rec(f"dequantize_{NAMES[t]}_11008x4096",
            lambda b, t=t: L.ggq_dequantize_f16(vp(b[0]), vp(b[1]), t, N_DIM, K_DIM, cur_stream()),
            [rings[t], out_ring], algo_bytes_dequant(t, N_DIM, K_DIM), iters=52)
    del out_ring
    # config 3: MMVQ batch 1 (quantize_q8_1 + mul_mat_vec_q)
    x1 = x128[:1].contiguous()
    y1 = torch.empty((1, N_DIM), dtype=torch.float16, device=dev)
    sc1 = torch.empty(int(L.ggq_mmvq_scratch_bytes(K_DIM)), dtype=torch.uint8, device=dev)
    for t in (Q4_0, Q4_K):
        rec(f"mmvq_{NAMES[t]}_batch1",
            lambda b, t=t: L.ggq_mul_mat_vec_q(vp(b[0]), vp(x1), vp(y1), t, 1, K_DIM, N_DIM, vp(sc1), cur_stream()),
            [rings[t]], algo_bytes_matmul(t, N_DIM, K_DIM, 1), 2.0 * N_DIM * K_DIM, iters=208)
    # config 4: MMQ batch 128 other formats; batch 8 for Q4_K and Q8_0
    y128 = torch.empty((BATCH, N_DIM), dtype=torch.float16, device=dev)
    for t in (Q5_K, Q6_K, Q8_0, Q4_0):
        rec(f"mmq_{NAMES[t]}_batch128",
            lambda b, t=t: L.ggq_mul_mat_q(vp(b[0]), vp(x128), vp(y128), t, 1, BATCH, K_DIM, N_DIM, vp(scratch), cur_stream()),
            [rings[t]], algo_bytes_matmul(t, N_DIM, K_DIM, BATCH), 2.0 * BATCH * N_DIM * K_DIM)
    x8 = x128[:8].contiguous()
    for t in (Q4_K, Q8_0):
        rec(f"mmq_{NAMES[t]}_batch8",
            lambda b, t=t: L.ggq_mul_mat_q(vp(b[0]), vp(x8), vp(y128), t, 1, 8, K_DIM, N_DIM, vp(scratch), cur_stream()),
            [rings[t]], algo_bytes_matmul(t, N_DIM, K_DIM, 8), 2.0 * 8 * N_DIM * K_DIM)
    # the HBM-bound batches between the GEMV and the 32-token tile (Q4_K: the 16-token-tile kernel, mmq_t16.hip)
    for bb in (2, 16, 32, 64):
        xb = x128[:bb].contiguous()
        rec(f"mmq_Q4_K_batch{bb}",
            lambda b, bb=bb, xb=xb: L.ggq_mul_mat_q(vp(b[0]), vp(xb), vp(y128), Q4_K, 1, bb, K_DIM, N_DIM, vp(scratch), cur_stream()),
            [rings[Q4_K]], algo_bytes_matmul(Q4_K, N_DIM, K_DIM, bb), 2.0 * bb * N_DIM * K_DIM, iters=52)
    # ... and the 32-element-block formats on the same kernel (batch <= 16; routed by shape)
    x16 = x128[:16].contiguous()
    rec("mmq_Q4_0_batch8",
        lambda b: L.ggq_mul_mat_q(vp(b[0]), vp(x8), vp(y128), Q4_0, 1, 8, K_DIM, N_DIM, vp(scratch), cur_stream()),
        [rings[Q4_0]], algo_bytes_matmul(Q4_0, N_DIM, K_DIM, 8), 2.0 * 8 * N_DIM * K_DIM, iters=52)
    rec("mmq_Q8_0_batch16",
        lambda b: L.ggq_mul_mat_q(vp(b[0]), vp(x16), vp(y128), Q8_0, 1, 16, K_DIM, N_DIM, vp(scratch), cur_stream()),
        [rings[Q8_0]], algo_bytes_matmul(Q8_0, N_DIM, K_DIM, 16), 2.0 * 16 * N_DIM * K_DIM, iters=52)
    rec("mmq_Q6_K_batch8",
        lambda b: L.ggq_mul_mat_q(vp(b[0]), vp(x8), vp(y128), Q6_K, 1, 8, K_DIM, N_DIM, vp(scratch), cur_stream()),
        [rings[Q6_K]], algo_bytes_matmul(Q6_K, N_DIM, K_DIM, 8), 2.0 * 8 * N_DIM * K_DIM, iters=52)
    # the grid-codebook IQ formats through the GEMV (codebook staged in LDS)
    IQ3_S, IQ2_XXS = 21, 16
    for t, nm in ((IQ3_S, "IQ3_S"), (IQ2_XXS, "IQ2_XXS")):
        wi = torch.from_numpy(synth.random_weight(t, N_DIM, K_DIM, seed=5)).to(dev)
        ri = ring_of(wi, wi.numel())
        rec(f"mmvq_{nm}_batch1",
            lambda b, t=t: L.ggq_mul_mat_vec_q(vp(b[0]), vp(x1), vp(y1), t, 1, K_DIM, N_DIM, vp(sc1), cur_stream()),
            [ri], algo_bytes_matmul(t, N_DIM, K_DIM, 1), 2.0 * N_DIM * K_DIM, iters=104)
        del ri, wi
    # north_star: Q8_0 at batch 1 too (MMVQ)
    rec("mmvq_Q8_0_batch1",
        lambda b: L.ggq_mul_mat_vec_q(vp(b[0]), vp(x1), vp(y1), Q8_0, 1, K_DIM, N_DIM, vp(sc1), cur_stream()),
        [rings[Q8_0]], algo_bytes_matmul(Q8_0, N_DIM, K_DIM, 1), 2.0 * N_DIM * K_DIM, iters=208)
    # the transposed (down-projection) shape: K = 11008, N = 4096
    KT, NT = N_DIM, K_DIM
    wT = torch.from_numpy(synth.random_weight(Q4_K, NT, KT, seed=3)).to(dev)
    ringT = ring_of(wT, wT.numel())
    xT = torch.randn((BATCH, KT), generator=torch.Generator().manual_seed(7)).half().to(dev)
    yT = torch.empty((BATCH, NT), dtype=torch.float16, device=dev)
    scT = torch.empty(max(int(L.ggq_mmq_scratch_bytes(BATCH, KT)), int(L.ggq_mmvq_scratch_bytes(KT))), dtype=torch.uint8, device=dev)
    for bb in (1, 8, 128):
        xb = xT[:bb].contiguous()
        if bb == 1:
            fn = lambda b, xb=xb: L.ggq_mul_mat_vec_q(vp(b[0]), vp(xb), vp(yT), Q4_K, 1, KT, NT, vp(scT), cur_stream())
        else:
            fn = lambda b, bb=bb, xb=xb: L.ggq_mul_mat_q(vp(b[0]), vp(xb), vp(yT), Q4_K, 1, bb, KT, NT, vp(scT), cur_stream())
        rec(f"transposed_4096x11008_Q4_K_batch{bb}", fn, [ringT], algo_bytes_matmul(Q4_K, NT, KT, bb), 2.0 * bb * NT * KT, iters=52)
    del ringT, wT
    st = time_launches(lambda i: L.ggq_quantize_q8_1_mmq(vp(x128), 1, vp(scratch), BATCH, K_DIM, Q4_K, cur_stream()), 208, use_graph=g)
    res["quantize_mmq_q8_1_batch128"] = dict(st, cache_state="warm: 1 MB of activations, 0.7 MB of scratch")
    st = time_launches(lambda i: L.ggq_quantize_q8_1_tiled(vp(x128), 1, vp(scratch), BATCH, K_DIM, Q4_K, cur_stream()), 208, use_graph=g)
    res["quantize_q8_1_tiled_batch128"] = dict(st, cache_state="warm: 1 MB of activations, 0.7 MB of scratch")
    # FFN gate + up on one activation quantisation (ggq.linear): quantise once, two streamed matmuls
    def gate_up(b):
        L.ggq_quantize_q8_1_tiled(vp(x128), 1, vp(scratch), BATCH, K_DIM, Q4_K, cur_stream())
        L.ggq_mul_mat_q_pretiled(vp(b[0]), vp(scratch), vp(y128), Q4_K, 1, BATCH, K_DIM, N_DIM, N_DIM, cur_stream())
        L.ggq_mul_mat_q_pretiled(vp(b[1]), vp(scratch), vp(y128), Q5_K, 1, BATCH, K_DIM, N_DIM, N_DIM, cur_stream())
    rec("gate_up_Q4_K_Q5_K_batch128_shared_quantisation", gate_up, [rings[Q4_K], rings[Q5_K]],
        algo_bytes_matmul(Q4_K, N_DIM, K_DIM, BATCH) + algo_bytes_matmul(Q5_K, N_DIM, K_DIM, BATCH), 4.0 * BATCH * N_DIM * K_DIM)
    res["gate_up_Q4_K_Q5_K_batch128_shared_quantisation"]["note"] = "compare with the two separate ops: step time + mmq_Q5_K_batch128"
    # the whole gated FFN front half, silu(x W_gate^T) * (x W_up^T): one quantisation, gate matmul, up matmul whose
    # write-back applies silu(gate) * acc (GGQ_EPI_SILU_MUL) — against two ops + torch's silu and mul kernels
    g128 = torch.empty((BATCH, N_DIM), dtype=torch.float16, device=dev)
    def ffn_fused(b):
        L.ggq_quantize_q8_1_tiled(vp(x128), 1, vp(scratch), BATCH, K_DIM, Q4_K, cur_stream())
        L.ggq_mul_mat_q_pretiled(vp(b[0]), vp(scratch), vp(g128), Q4_K, 1, BATCH, K_DIM, N_DIM, N_DIM, cur_stream())
        L.ggq_mul_mat_q_pretiled_epi(vp(b[1]), vp(scratch), vp(y128), Q4_K, 1, BATCH, K_DIM, N_DIM, N_DIM, 2, vp(g128), cur_stream())
    def ffn_unfused(b):
        L.ggq_mul_mat_q(vp(b[0]), vp(x128), vp(g128), Q4_K, 1, BATCH, K_DIM, N_DIM, vp(scratch), cur_stream())
        L.ggq_mul_mat_q(vp(b[1]), vp(x128), vp(y128), Q4_K, 1, BATCH, K_DIM, N_DIM, vp(scratch), cur_stream())
        torch.mul(torch.nn.functional.silu(g128), y128, out=y128)
    ring2 = rings[Q4_K][1:] + rings[Q4_K][:1]
    nb2 = 2 * algo_bytes_matmul(Q4_K, N_DIM, K_DIM, BATCH)
    rec("ffn_gate_up_Q4_K_batch128_fused_silu_mul_epilogue", ffn_fused, [rings[Q4_K], ring2], nb2, 4.0 * BATCH * N_DIM * K_DIM)
    rec("ffn_gate_up_Q4_K_batch128_two_ops_plus_torch_silu_mul", ffn_unfused, [rings[Q4_K], ring2], nb2, 4.0 * BATCH * N_DIM * K_DIM)
    return res


if __name__ == "__main__":
    main()
