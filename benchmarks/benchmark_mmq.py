"""The reference's benchmark protocol (benchmarks/benchmark_mmq.py:42-145) on this build, offline.

Same two columns — "Dequant" = ggml_dequantize(...).T.to(dtype) then x @ wt, "MMQ" = ggml_mul_mat_a8 —
same loop (5 warm-up + 100 iterations over every tensor of Quant_{TYPE}_{hidden}.gguf, wall clock around a
synchronised loop), same CLI.  Differences forced by the environment: the sample files are written locally
with synthetic block-valid tensors (ggq.gguf_io) instead of snapshot_download("Isotr0py/test-gguf-sample"),
gguf.GGUFReader is ours, and --hidden-size / --rows accept any size (the hub samples exist for 256 / 1024
only).  Adds a kernel-only column (HIP events around a hipGraph of the op) and algorithmic GB/s.
Like the reference it writes `benchmark_results_local_HD{hidden}xB{tokens}.csv` with the columns
`Quantization, Dequant Time (ms), MMQ Time (ms)` (benchmark_mmq.py:177, 195-197; extra columns appended) and
`--profile` brackets ONE iteration of each column with the profiler start/stop calls plus a named range
(benchmark_mmq.py:76-77, 92-93: cudaProfilerStart/Stop -> hipProfilerStart/Stop on ROCm; the range shows up as a
roctx marker in rocprofv3 --marker-trace) after `--num-warmup-iters`, instead of the timed loop.

  python benchmarks/benchmark_mmq.py --quant-dtype Q4_K --hidden-size 4096 --rows 11008 --num-tokens 128
  python benchmarks/benchmark_mmq.py --all --num-tokens 1 8 128
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))

import torch  # noqa: E402

from ggq import gguf_io  # noqa: E402
from ggq.formats import GGMLType, row_bytes  # noqa: E402

QUANT_TYPES_MAP = {t.name: t for t in (GGMLType.Q2_K, GGMLType.Q3_K, GGMLType.Q4_K, GGMLType.Q5_K, GGMLType.Q6_K,
                                       GGMLType.Q4_0, GGMLType.Q4_1, GGMLType.Q5_0, GGMLType.Q5_1, GGMLType.Q8_0)}
DTYPES_MAP = {"half": torch.float16, "bfloat16": torch.bfloat16, "float": torch.float32}


def seed_everything(seed):
    import random
    import numpy as np
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def get_gguf_sample_tensors(sample_dir, hidden_size, quant_type, rows):
    path = os.path.join(sample_dir, gguf_io.sample_filename(quant_type, hidden_size))
    if not os.path.exists(path):
        tensors = []
        from ggq import synth
        for i, r in enumerate(rows):
            tensors.append((f"tensor_{r}x{hidden_size}", synth.random_weight(quant_type, r, hidden_size, seed=i),
                            int(quant_type), (r, hidden_size)))
        gguf_io.write_gguf(path, tensors, {"general.name": f"synthetic {quant_type.name} sample"})
    return gguf_io.GGUFReader(path).tensors


def kernel_time_us(fn, iters=50):
    """HIP-event time per call of fn replayed from one hipGraph (no host launch gaps)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


class profiled:
    """the reference's `--profile` bracket (benchmark_mmq.py:76-77, 92-93) + a named range"""

    def __init__(self, name, on):
        self.name, self.on = name, on

    def __enter__(self):
        if self.on:
            torch.cuda.synchronize()
            torch.cuda.cudart().cudaProfilerStart()
            torch.cuda.nvtx.range_push(self.name)

    def __exit__(self, *a):
        if self.on:
            torch.cuda.synchronize()
            torch.cuda.nvtx.range_pop()
            torch.cuda.cudart().cudaProfilerStop()


@torch.inference_mode()
def main(num_tokens, hidden_size, quant_type, dtype, rows, sample_dir, seed=0, num_warmup_iters=5, num_iters=100, profile=False):
    import ggml as ops
    seed_everything(seed)
    x = torch.randn(num_tokens, hidden_size, dtype=dtype, device="cuda")
    tensors = get_gguf_sample_tensors(sample_dir, hidden_size, quant_type, rows)
    w = [torch.tensor(t.data, device="cuda") for t in tensors]
    shape = [tuple(map(int, t.name.split("_")[-1].split("x"))) for t in tensors]
    matmul = ops.ggml_mul_mat_vec_a8 if num_tokens == 1 else ops.ggml_mul_mat_a8

    def run_mmq(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            for tensor in w:
                matmul(tensor, x, quant_type, tensor.size(0))
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    def run_dequant(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            for tensor, tensor_shape in zip(w, shape):
                wt = ops.ggml_dequantize(tensor, quant_type, *tensor_shape).T.to(dtype)
                x @ wt
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    run_dequant(num_warmup_iters)
    if profile:
        with profiled(f"dequant+matmul {quant_type.name}", True):
            dequant_ms = run_dequant(1) * 1e3
    else:
        dequant_ms = run_dequant(num_iters) * 1e3
    run_mmq(num_warmup_iters)
    if profile:
        with profiled(f"ggml_mul_mat_a8 {quant_type.name}", True):
            mmq_ms = run_mmq(1) * 1e3
    else:
        mmq_ms = run_mmq(num_iters) * 1e3
    kern_us = sum(kernel_time_us(lambda t=t: matmul(t, x, quant_type, t.size(0))) for t in w)
    esz = x.element_size()
    nbytes = sum(m * row_bytes(quant_type, n) + num_tokens * n * esz + num_tokens * m * esz for m, n in shape)
    return {"quant": quant_type.name, "hidden_size": hidden_size, "rows": [m for m, _ in shape], "num_tokens": num_tokens,
            "dtype": str(dtype).replace("torch.", ""), "dequant_path_ms": round(dequant_ms, 4), "mmq_path_ms": round(mmq_ms, 4),
            "speedup_vs_dequant_path": round(dequant_ms / mmq_ms, 2), "mmq_kernels_us": round(kern_us, 2),
            "algorithmic_GBps": round(nbytes / (kern_us * 1e-6) / 1e9, 1)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="Benchmark the quantised matmul ops against dequantise + matmul")
    ap.add_argument("--num-tokens", type=int, nargs="+", default=[128])
    ap.add_argument("--hidden-size", type=int, default=4096)
    ap.add_argument("--rows", type=int, nargs="+", default=[11008], help="rows of the sample tensors (reference samples: a few per file)")
    ap.add_argument("--quant-dtype", type=str, choices=QUANT_TYPES_MAP.keys(), default="Q4_K")
    ap.add_argument("--all", action="store_true", help="every format")
    ap.add_argument("--dtype", type=str, choices=DTYPES_MAP.keys(), default="half")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--sample-dir", type=str, default=None, help="where Quant_{TYPE}_{hidden}.gguf are kept (default: a temp dir)")
    ap.add_argument("--profile", action="store_true", help="profile one iteration of each column instead of timing the loop")
    ap.add_argument("--num-warmup-iters", type=int, default=5)
    ap.add_argument("--num-iters", type=int, default=100)
    ap.add_argument("--csv-dir", type=str, default=".", help="where benchmark_results_local_HD{hidden}xB{tokens}.csv is written")
    args = ap.parse_args()
    sample_dir = args.sample_dir or tempfile.mkdtemp(prefix="ggq_gguf_samples_")
    os.makedirs(sample_dir, exist_ok=True)
    types = list(QUANT_TYPES_MAP.values()) if args.all else [QUANT_TYPES_MAP[args.quant_dtype]]
    import pandas as pd
    for nt in args.num_tokens:
        rows_out = []
        for qt in types:
            r = main(nt, args.hidden_size, qt, DTYPES_MAP[args.dtype], args.rows, sample_dir, seed=args.seed,
                     num_warmup_iters=args.num_warmup_iters, num_iters=args.num_iters, profile=args.profile)
            print(json.dumps(r), flush=True)
            rows_out.append({"Quantization": r["quant"], "Dequant Time (ms)": r["dequant_path_ms"], "MMQ Time (ms)": r["mmq_path_ms"],
                             "MMQ kernels (us)": r["mmq_kernels_us"], "Algorithmic GB/s": r["algorithmic_GBps"]})
        # the reference's result file (benchmark_mmq.py:177, 195-197), one per (hidden size, token count)
        os.makedirs(args.csv_dir, exist_ok=True)
        pd.DataFrame(rows_out).to_csv(os.path.join(args.csv_dir, f"benchmark_results_local_HD{args.hidden_size}xB{nt}.csv"), index=False)
