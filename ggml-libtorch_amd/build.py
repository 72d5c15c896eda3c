"""In-tree build of every native artefact (gfx950 only, no JIT cache):

  lib/libggq_hip.so   the C-ABI library: hand-written HIP kernels (csrc/hip/*.hip)
  lib/libggq_cpu.so   the C-ABI host twin (csrc/cpu/ggq_cpu.cpp)
  ggml/_ggml*.so      torch operator registration (csrc/torch/binding.cpp) -> torch.ops._ggml.*
  custom_ops*.so      the CPU python module of the reference surface (csrc/torch/custom_ops.cpp)

`python build.py` (or __graft_entry__.build()) compiles what is out of date.  The shared
objects are git-ignored but travel to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(ROOT, "csrc")
OBJ = os.path.join(ROOT, "_build")
LIB = os.path.join(ROOT, "lib")
INCLUDE = os.path.join(os.path.dirname(ROOT), "include")
ARCH = "gfx950"
EXT_SUFFIX = sysconfig.get_config_var("EXT_SUFFIX")

HIP_SOURCES = ["dequant", "quantize", "mmvq", "mmq", "mmq_t16", "mmq_x64", "peer"]
TRAITS_SRC = os.path.join(CSRC, "core", "traits.cpp")
# No implicit fused-multiply-add contraction anywhere: the fp16 dequantise sequence and the Q8_1
# quantiser must round after every operation exactly like the reference's intrinsics, and in the
# matmul kernels contraction made the rounding of an output depend on which accumulator register
# (i.e. which tile row) it landed in.  FMAs are written explicitly (__builtin_fmaf) where wanted.
NO_CONTRACT = {"dequant", "quantize", "mmvq", "mmq", "mmq_t16", "mmq_x64"}


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build the gfx950 kernels)")


def _run(cmd):
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError("build command failed:\n  " + " ".join(cmd) + "\n" + p.stdout + p.stderr)
    return p.stdout + p.stderr


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _common_deps():
    return [os.path.join(INCLUDE, "ggq.h"), os.path.join(CSRC, "hip", "ggq_common.h"),
            os.path.join(CSRC, "hip", "mmq_unpack.h"), os.path.join(CSRC, "hip", "iq_common.h"),
            os.path.join(CSRC, "hip", "iq_tables.h"), os.path.join(CSRC, "hip", "mmq_x64_loops.inc"), os.path.abspath(__file__)]


def build_hip(force=False, verbose=False):
    so = os.path.join(LIB, "libggq_hip.so")
    sources = [os.path.join(CSRC, "hip", n + ".hip") for n in HIP_SOURCES] + [TRAITS_SRC] + _common_deps()
    if not force and not _stale(so, sources):
        return so  # prebuilt library newer than every source (the object cache need not exist: GPU box)
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIB, exist_ok=True)
    hipcc = _hipcc()
    extra = os.environ.get("GGQ_HIPCC_FLAGS", "").split()   # kernel experiments: -D switches
    objs, jobs = [], []
    for name in HIP_SOURCES:
        src = os.path.join(CSRC, "hip", name + ".hip")
        obj = os.path.join(OBJ, name + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + _common_deps()):
            cmd = [hipcc, "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-fno-gpu-rdc",
                   "-Wno-unused-result", *extra, "-c", src, "-o", obj]
            if name in NO_CONTRACT:
                cmd.insert(1, "-ffp-contract=off")
            jobs.append(cmd)
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), 6)) as ex:
            for out in ex.map(_run, jobs):
                if verbose and out.strip():
                    print(out)
    tobj = os.path.join(OBJ, "traits.o")
    objs.append(tobj)
    relink = bool(jobs)
    if force or _stale(tobj, [TRAITS_SRC, os.path.join(INCLUDE, "ggq.h")]):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-c", TRAITS_SRC, "-o", tobj])
        relink = True
    if force or relink or _stale(so, objs):
        _run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-o", so] + objs)
    return so


def build_cpu(force=False):
    os.makedirs(LIB, exist_ok=True)
    src = os.path.join(CSRC, "cpu", "ggq_cpu.cpp")
    src2 = os.path.join(CSRC, "cpu", "ggq_cpu_mmq.cpp")
    so = os.path.join(LIB, "libggq_cpu.so")
    if force or _stale(so, [src, src2, TRAITS_SRC, os.path.join(INCLUDE, "ggq.h")]):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-pthread", "-o", so, src, src2,
              TRAITS_SRC])
    return so


def _torch_flags():
    import torch
    from torch.utils import cpp_extension as ce
    inc = [f"-I{p}" for p in ce.include_paths()]
    inc += [f"-I{sysconfig.get_paths()['include']}", "-I/opt/rocm/include"]
    libdir = ce.library_paths()[0]
    defs = [f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-D__HIP_PLATFORM_AMD__=1",
            "-DUSE_ROCM=1", "-DTORCH_API_INCLUDE_EXTENSION_H"]
    return inc, libdir, defs


def build_torch_ext(force=False):
    inc, libdir, defs = _torch_flags()
    hip_so = os.path.join(LIB, "libggq_hip.so")
    cpu_so = os.path.join(LIB, "libggq_cpu.so")
    out = []
    # torch.ops._ggml registration
    src = os.path.join(CSRC, "torch", "binding.cpp")
    so = os.path.join(ROOT, "ggml", "_ggml" + EXT_SUFFIX)
    if force or _stale(so, [src, os.path.join(INCLUDE, "ggq.h"), hip_so]):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DTORCH_EXTENSION_NAME=_ggml"] + defs + inc +
             ["-o", so, src, f"-L{LIB}", "-lggq_hip", f"-L{libdir}", "-ltorch", "-ltorch_cpu", "-lc10",
              "-ltorch_hip", "-lc10_hip", "-Wl,-rpath,$ORIGIN/../lib", f"-Wl,-rpath,{libdir}"])
    out.append(so)
    # custom_ops (CPU) python module
    src = os.path.join(CSRC, "torch", "custom_ops.cpp")
    so = os.path.join(ROOT, "custom_ops" + EXT_SUFFIX)
    if force or _stale(so, [src, os.path.join(INCLUDE, "ggq.h"), cpu_so]):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DTORCH_EXTENSION_NAME=custom_ops"] + defs + inc +
             ["-o", so, src, f"-L{LIB}", "-lggq_cpu", f"-L{libdir}", "-ltorch", "-ltorch_cpu", "-lc10",
              "-ltorch_python", "-Wl,-rpath,$ORIGIN/lib", f"-Wl,-rpath,{libdir}"])
    out.append(so)
    return out


def build_all(force=False, verbose=False):
    res = {"hip": build_hip(force, verbose), "cpu": build_cpu(force)}
    res["torch"] = build_torch_ext(force)
    return res


if __name__ == "__main__":
    r = build_all(force="--force" in sys.argv, verbose=True)
    for k, v in r.items():
        print(k, v)
