"""One-off generator of ggml-libtorch_amd/csrc/hip/iq_tables.h: the codebook grids of the IQ formats, transcribed as DATA
from the reference's HK/ggml/ggml-common.h (iq2xxs_grid :193-258, iq2xs_grid :260-389, iq2s_grid :391-648, iq3xxs_grid
:650-683, iq3xs_grid :685-750, iq1s_grid_gpu :754-1011).  Run in the build container only (the reference does not travel);
the generated header is committed.  ksigns_iq2xs / ksigns64 / kmask_iq2xs are not stored: ksigns_iq2xs[i] = i | (parity(i) << 7)."""
import re, sys, hashlib
src = open("/root/reference/hf-kernels/ggml-kernels/ggml/ggml-common.h").read()
want = [("iq2xxs_grid", 256, 64), ("iq2xs_grid", 512, 64), ("iq2s_grid", 1024, 64), ("iq3xxs_grid", 256, 32), ("iq3xs_grid", 512, 32),
        ("iq1s_grid_gpu", 2048, 32)]
out = ["// iq_tables.h — codebook grids of the ggml IQ formats: constant DATA transcribed from the reference's",
       "// HK/ggml/ggml-common.h:193-1011 by scripts/gen_iq_tables.py (values only; every use site is ours).",
       "// GGQ_IQ_CONST is the storage class: `static __device__ const` in the HIP kernels, `static const` in the C oracle.",
       "// One grid entry = 8 (iq2*, 64-bit) or 4 (iq3*, 32-bit) unsigned byte magnitudes, or 8 nibbles (iq1s, 32-bit).",
       "#pragma once", "#include <stdint.h>", "#ifndef GGQ_IQ_CONST", "#define GGQ_IQ_CONST static const", "#endif", ""]
for name, n, bits in want:
    m = re.search(r"%s\[(\d+)\]\s*=\s*\{(.*?)\};" % name, src, re.S)
    assert m and int(m.group(1)) == n, name
    vals = [int(v, 16) for v in re.findall(r"0x[0-9a-fA-F]+", m.group(2))]
    assert len(vals) == n, (name, len(vals))
    assert all(v < (1 << bits) for v in vals), name
    ty = "uint64_t" if bits == 64 else "uint32_t"
    w = 16 if bits == 64 else 8
    out.append(f"GGQ_IQ_CONST {ty} ggq_{name}[{n}] = {{")
    per = 4 if bits == 64 else 8
    for i in range(0, n, per):
        out.append("  " + ", ".join(f"0x{v:0{w}x}" + ("ull" if bits == 64 else "u") for v in vals[i:i + per]) + ",")
    out.append("};")
    out.append("")
    print(name, n, hashlib.sha256(repr(vals).encode()).hexdigest()[:16])
# check the ksigns identity against the reference's table
m = re.search(r"ksigns_iq2xs\[128\]\s*=\s*\{(.*?)\};", src, re.S)
ks = [int(v) for v in re.findall(r"\d+", m.group(1))]
assert len(ks) == 128 and all(ks[i] == (i | ((bin(i).count("1") & 1) << 7)) for i in range(128)), "ksigns identity"
open("/root/repo/ggml-libtorch_amd/csrc/hip/iq_tables.h", "w").write("\n".join(out))
print("ksigns_iq2xs[i] == i | parity(i) << 7: verified")
