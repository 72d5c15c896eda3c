"""Static check of the s_waitcnt placement in compiled gfx950 kernels: is every register a memory instruction returns
into (vector loads -> vmcnt, LDS reads -> lgkmcnt, scalar loads -> lgkmcnt) waited for before ANY later instruction on ANY
path touches it, loop-carried uses included?  Written to answer one question of the round-2 review about the parked MMQ
variants whose results changed from run to run ("is a vmcnt / lgkmcnt wait placed for the first but not the loop-carried
use?") from the ISA alone, without re-running a faulty build.

Model (the in-order rules scripts/ubench_vmcnt_order.hip confirmed on the hardware):
  * vmcnt: every vector-memory instruction (buffer_/global_/scratch_/flat_ loads, stores, atomics, LDS-DMA) takes a slot;
    slots retire in issue order; `s_waitcnt vmcnt(N)` returns when at most N are outstanding, i.e. every op that has more than
    N younger vector-memory ops behind it is complete.
  * lgkmcnt: LDS instructions retire in order among themselves -> after lgkmcnt(N) all but the N youngest LDS ops are complete;
    scalar loads may return out of order -> only lgkmcnt(0) completes them.
  * LDS-DMA (`... lds`) writes LDS under vmcnt: an LDS read while one may be outstanding is reported (the kernels that use it
    wait for vmcnt(0) first).
The walk is path-sensitive: a worklist over (instruction, state) with states normalised (registers that are already safe are
dropped), so loops are followed until no new state appears.

usage: python scripts/check_waitcnt.py lib.so|code-object [kernel-name-substring ...]   -> findings on stdout, exit 1 on any"""
import os, re, shutil, subprocess, sys, tempfile
from collections import deque

LLVM = "/opt/rocm/lib/llvm/bin"
REG1 = re.compile(r"\b([vsa])(\d+)\b")
REGN = re.compile(r"\b([vsa])\[(\d+):(\d+)\]")
VM_LOAD = ("buffer_load", "global_load", "scratch_load", "flat_load", "tbuffer_load")
VM_STORE = ("buffer_store", "global_store", "scratch_store", "flat_store", "tbuffer_store")
VM_ATOMIC = ("buffer_atomic", "global_atomic", "flat_atomic")
MAX_STATES_PER_PC = 4096


def regs_of(text):
    out = []
    for m in REGN.finditer(text):
        out += [(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    text = REGN.sub(" ", text)
    out += [(m.group(1), int(m.group(2))) for m in REG1.finditer(text)]
    return out


class Ins:
    __slots__ = ("addr", "mn", "ops", "text", "dst", "src", "kind", "target", "wait")

    def __init__(self, addr, text, word=None):
        self.addr, self.text = addr, text
        parts = text.split(None, 1)
        self.mn = parts[0]
        ops = parts[1] if len(parts) > 1 else ""
        self.ops = ops
        mn = self.mn
        self.kind, self.target, self.wait = "alu", None, None
        oplist = [o.strip() for o in ops.split(",")]
        first = regs_of(oplist[0]) if oplist and oplist[0] else []
        rest = regs_of(",".join(oplist[1:])) if len(oplist) > 1 else []
        self.dst, self.src = [], first + rest
        if mn.startswith(VM_LOAD):
            if " lds" in " " + ops or "_lds_" in mn:
                self.kind = "vm_dma"
            else:
                self.kind, self.dst, self.src = "vm_load", first, rest
        elif mn.startswith(VM_STORE):
            self.kind = "vm_store"
        elif mn.startswith(VM_ATOMIC):
            ret = " sc0" in ops or " glc" in ops
            self.kind = "vm_load" if ret else "vm_store"
            if ret:
                self.dst, self.src = first, rest
        elif mn.startswith("ds_"):
            # reads, returning atomics, permutes and swizzles write their first operand; writes do not
            returns = any(k in mn for k in ("read", "_rtn", "permute", "swizzle", "consume", "append", "ds_load"))
            self.kind = "lds_ret" if returns else "lds"
            if returns:
                self.dst, self.src = first, rest
        elif mn.startswith(("s_load", "s_buffer_load", "s_scratch_load")):
            self.kind, self.dst, self.src = "smem", first, rest
        elif mn in ("s_memtime", "s_memrealtime"):
            self.kind, self.dst, self.src = "smem", first, rest
        elif mn == "s_waitcnt":
            self.kind = "wait"
            vm = re.search(r"vmcnt\((\d+)\)", ops)
            lg = re.search(r"lgkmcnt\((\d+)\)", ops)
            if vm is None and lg is None and re.fullmatch(r"\s*(0x[0-9a-fA-F]+|\d+)\s*", ops):   # raw immediate (gfx9 encoding)
                v = int(ops.strip(), 0)
                vmv, lgv = (v & 15) | (((v >> 14) & 3) << 4), (v >> 8) & 15
                self.wait = (None if vmv == 63 else vmv, None if lgv == 15 else lgv)
            else:
                self.wait = (int(vm.group(1)) if vm else None, int(lg.group(1)) if lg else None)
        elif mn.startswith(("s_cbranch", "s_branch")):
            self.kind = "cbranch" if mn.startswith("s_cbranch") else "branch"
            # the branch offset from the encoding when it is there (hand-written asm labels are printed symbolically)
            off = (word & 0xFFFF) if word is not None else int(ops.strip().split()[0], 0)
            if off >= 32768:
                off -= 65536
            self.target = addr + 4 + 4 * off
        elif mn == "s_endpgm":
            self.kind = "end"


def parse_kernels(disasm):
    """-> {kernel name: [Ins]} from `llvm-objdump -d` text"""
    kernels, cur = {}, None
    for line in disasm.split("\n"):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            if m.group(1).startswith("L_"):   # a label of a hand-written inline-asm loop (mmq_x64_loops.inc), not a function
                continue
            cur = kernels.setdefault(m.group(1), [])
            continue
        if cur is None or "//" not in line:
            continue
        body, tail = line.split("//", 1)
        body = body.strip()
        ma = re.match(r"\s*([0-9A-Fa-f]+):\s*([0-9A-Fa-f]{8})?", tail)
        if not body or not ma:
            continue
        cur.append(Ins(int(ma.group(1), 16), body, int(ma.group(2), 16) if ma.group(2) else None))
    return kernels


def norm(state):
    pv, pl, vm, lds, sm, dma = state
    vm = frozenset((r, a) for r, a in vm if a < pv)
    lds = frozenset((r, a) for r, a in lds if a < pl)
    dma = tuple(sorted(a for a in dma if a < pv))
    return (pv, pl, vm, lds, sm, dma)


def check_kernel(name, ins):
    index = {i.addr: n for n, i in enumerate(ins)}
    start = (0, 0, frozenset(), frozenset(), frozenset(), ())
    seen = [set() for _ in ins]
    work = deque([(0, start)])
    findings = {}
    truncated = False
    while work:
        pc, st = work.popleft()
        if pc >= len(ins):
            continue
        if st in seen[pc]:
            continue
        if len(seen[pc]) >= MAX_STATES_PER_PC:
            truncated = True
            continue
        seen[pc].add(st)
        I = ins[pc]
        pv, pl, vm, lds, sm, dma = st
        vmd, ldsd = dict(vm), dict(lds)
        if I.kind == "wait":
            wv, wl = I.wait
            if wv is not None:
                pv = min(pv, wv)
            if wl is not None:
                pl = min(pl, wl)
                if wl == 0:
                    sm = frozenset()
        else:
            touched = I.src + (I.dst if I.kind == "alu" or I.kind in ("lds", "vm_store", "vm_dma") else [])
            # a memory instruction's own destination: only a pending result of ANOTHER counter is a hazard (same counter returns in order)
            for r in touched:
                if r in vmd:
                    findings.setdefault((I.addr, r, "vmcnt"), (I.text, pv, vmd[r]))
                if r in ldsd:
                    findings.setdefault((I.addr, r, "lgkmcnt(lds)"), (I.text, pl, ldsd[r]))
                if r in sm:
                    findings.setdefault((I.addr, r, "lgkmcnt(smem)"), (I.text, pl, -1))
            for r in I.dst:
                if I.kind != "alu":
                    if I.kind != "vm_load" and r in vmd:
                        findings.setdefault((I.addr, r, "vmcnt(waw)"), (I.text, pv, vmd[r]))
                    if I.kind != "lds_ret" and r in ldsd:
                        findings.setdefault((I.addr, r, "lgkmcnt(lds,waw)"), (I.text, pl, ldsd[r]))
                    if r in sm and I.kind != "smem":
                        findings.setdefault((I.addr, r, "lgkmcnt(smem,waw)"), (I.text, pl, -1))
            if I.kind in ("lds", "lds_ret") and dma:
                findings.setdefault((I.addr, ("lds", 0), "vmcnt(lds-dma)"), (I.text, pv, min(dma)))
            if I.kind in ("vm_load", "vm_store", "vm_dma"):
                vmd = {r: a + 1 for r, a in vmd.items()}
                dma = tuple(a + 1 for a in dma)
                pv = min(pv + 1, 64)
                if I.kind == "vm_load":
                    for r in I.dst:
                        vmd[r] = 0
                if I.kind == "vm_dma":
                    dma = dma + (0,)
                if I.mn.startswith("flat_"):   # flat also takes an lgkmcnt slot; we never emit it
                    pl = min(pl + 1, 16)
            elif I.kind in ("lds", "lds_ret"):
                ldsd = {r: a + 1 for r, a in ldsd.items()}
                pl = min(pl + 1, 16)
                if I.kind == "lds_ret":
                    for r in I.dst:
                        ldsd[r] = 0
            elif I.kind == "smem":
                sm = sm | frozenset(I.dst)
        new = norm((pv, pl, frozenset(vmd.items()), frozenset(ldsd.items()), sm, dma))
        if I.kind == "end":
            continue
        if I.kind in ("branch", "cbranch"):
            if I.target in index:
                work.append((index[I.target], new))
            if I.kind == "branch":
                continue
        work.append((pc + 1, new))
    return findings, truncated


def disassemble(path, tmp):
    if path.endswith(".so"):
        dst = os.path.join(tmp, os.path.basename(path))
        shutil.copy(path, dst)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", dst], check=True, capture_output=True, cwd=tmp)
        cos = sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f)
    else:
        cos = [path]
    for co in cos:
        yield subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True).stdout


def main(argv):
    path, filters = argv[1], argv[2:]
    tmp = tempfile.mkdtemp(prefix="ggq_waitcnt_")
    total, nk, ntrunc, ndma, dma_kernels = 0, 0, 0, 0, set()
    try:
        for dis in disassemble(path, tmp):
            for name, ins in parse_kernels(dis).items():
                if filters and not any(f in name for f in filters):
                    continue
                if not ins or not any(i.kind == "end" for i in ins):
                    continue
                nk += 1
                f, trunc = check_kernel(name, ins)
                ntrunc += trunc
                for (addr, reg, cnt), (text, pend, age) in sorted(f.items(), key=lambda kv: kv[0][0]):
                    if cnt == "vmcnt(lds-dma)":   # double-buffered LDS-DMA kernels read one buffer while the other is filled: listed, not failed
                        ndma += 1
                        dma_kernels.add(name)
                        continue
                    total += 1
                    print(f"UNWAITED {name[:80]} @{addr:x}: {text}   <- {reg[0]}{reg[1]} may still be in flight ({cnt}: {pend} possibly outstanding, {age} younger)")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print(f"{nk} kernels checked in {os.path.basename(path)}: {total} unwaited register uses; {ndma} LDS reads with an LDS-DMA possibly outstanding "
          f"in {len(dma_kernels)} kernels (double-buffered activations of mmq_kernel: the buffer being read is not the one being filled — "
          f"not decidable from the ISA, listed only); {ntrunc} kernels with a truncated state set")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
