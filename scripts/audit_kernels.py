"""Audit of the SHIPPED code objects (lib/libggq_hip.so), no GPU needed:
  * which kernels spill registers (.vgpr_spill_count / .sgpr_spill_count in the code-object metadata) — INFORMATIONAL: the
    three round-2 builds whose results changed from run to run (lanes 48-63 of single accumulator registers) all spilled
    41-70 registers, but so do 48 shipped kernels that have never shown it, so spilling alone is not the cause;
  * every v_mfma_i32_32x32x32_i8 keeps the wait states scripts/ubench_mfma_hazard.hip measured the hardware to need and
    not to interlock (12 before a VALU read of its result, 4 before a VALU write of its SrcC): scripts/check_mfma_hazards.py.
usage: python scripts/audit_kernels.py [path/to/lib.so]  -> summary on stdout, exit 1 on a wait-state violation"""
import os, re, subprocess, sys, tempfile, shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import check_mfma_hazards as H


def extract(lib, tmp):
    dst = os.path.join(tmp, os.path.basename(lib))
    shutil.copy(lib, dst)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", dst], check=True, capture_output=True)
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f)


def kernels_meta(co):
    out = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    res, cur = [], None
    for line in out.split("\n"):
        m = re.match(r"\s*(- )?\.(\w+):\s*(.*)", line)
        if not m:
            continue
        k, v = m.group(2), m.group(3).strip()
        if k == "agpr_count":   # first key of a kernel entry (keys are sorted)
            cur = {}
            res.append(cur)
        if cur is not None and k in ("name", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size"):
            cur[k] = v
    return [r for r in res if "vgpr_count" in r and "name" in r]


def audit(lib):
    tmp = tempfile.mkdtemp(prefix="ggq_audit_")
    findings, spilling, rows = 0, 0, []
    try:
        for co in extract(lib, tmp):
            for k in kernels_meta(co):
                sp = int(k.get("vgpr_spill_count", 0)) + int(k.get("sgpr_spill_count", 0))
                rows.append((k["name"], int(k["vgpr_count"]), sp, int(k.get("private_segment_fixed_size", 0))))
                if sp:
                    spilling += 1
                    print(f"spill   {k['name'][:90]}: {sp} registers")
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True).stdout
            # objdump format -> the listing format the checker parses: drop the '// addr: bytes' tails, '<name>:' labels
            lines = []
            for l in dis.split("\n"):
                l = l.split("//")[0].rstrip()
                m = re.match(r"^[0-9a-f]+ <(\w+)>:", l)
                lines.append((m.group(1) + ":") if m else l)
            names = [(i, l[:-1]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:$", l)] + [(len(lines), None)]
            for (a, n), (b, _) in zip(names, names[1:]):
                body = lines[a:b]
                if any("v_mfma_i32_32x32x32" in x for x in body):
                    findings += H.check(body, n[:70])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return findings, spilling, rows


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "ggml-libtorch_amd", "lib", "libggq_hip.so")
    f, sp, rows = audit(lib)
    print(f"{len(rows)} kernels in {os.path.basename(lib)}: max VGPRs {max(r[1] for r in rows)}, kernels that spill {sp}, with a private segment {sum(1 for r in rows if r[3])}, MFMA wait-state violations {f}")
    sys.exit(1 if f else 0)
