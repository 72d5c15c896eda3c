"""Static check of a gfx950 ISA listing (hipcc -S) against the wait states MEASURED by scripts/ubench_mfma_hazard.hip for
v_mfma_i32_32x32x32_i8 (the hardware does not interlock these; the compiler's hazard recognizer is trusted to):
  RAW  a VALU (non-MFMA) read of an MFMA result register needs >= 12 wait states after the MFMA,
  WAR  a VALU write of an MFMA's SrcC register needs >= 4 wait states after the MFMA.
usage: python scripts/check_mfma_hazards.py file.s [kernel-name-substring]   -> prints every violation, exit 1 if any"""
import re, sys

RAW_NEED, WAR_NEED = 12, 4
reg_re = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")

def regs(tok):
    out = set()
    for a, b, c in reg_re.findall(tok):
        if c: out.add(int(c))
        else: out.update(range(int(a), int(b) + 1))
    return out

def parse(line):
    line = line.split(";")[0].strip()
    if not line or line.endswith(":") or line.startswith("."): return None
    m = re.match(r"(\S+)\s*(.*)", line)
    op, rest = m.group(1), m.group(2)
    ops = [o.strip() for o in rest.split(",")] if rest else []
    return op, ops

def states(op, ops):
    if op == "s_nop": return int(ops[0]) + 1
    return 1

def check(lines, name=""):
    bad = 0
    ins = [(i, parse(l)) for i, l in enumerate(lines)]
    ins = [(i, p) for i, p in ins if p]
    for k, (ln, (op, ops)) in enumerate(ins):
        if not op.startswith("v_mfma_i32_32x32x32"): continue
        dst, srcc = regs(ops[0]), regs(ops[3])
        ws = 0
        for ln2, (op2, ops2) in ins[k + 1:]:
            if op2.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")): break   # (basic-block end: not followed)
            is_valu = op2.startswith("v_") and not op2.startswith("v_mfma")
            if is_valu and ops2:
                wr = regs(ops2[0]); rd = set().union(*[regs(o) for o in ops2[1:]]) if len(ops2) > 1 else set()
                if ws < WAR_NEED and wr & srcc and not (wr & srcc) <= dst:
                    print(f"{name}:{ln2 + 1}: WAR  {op2} writes SrcC of the MFMA at line {ln + 1} after {ws} wait states (< {WAR_NEED})"); bad += 1
                if ws < RAW_NEED and rd & dst:
                    print(f"{name}:{ln2 + 1}: RAW  {op2} reads a result of the MFMA at line {ln + 1} after {ws} wait states (< {RAW_NEED})"); bad += 1
            ws += states(op2, ops2)
            if ws >= max(RAW_NEED, WAR_NEED): break
    return bad

if __name__ == "__main__":
    txt = open(sys.argv[1]).read().split("\n")
    sub = sys.argv[2] if len(sys.argv) > 2 else None
    total, cur, start = 0, None, 0
    names = [(i, l.split(":")[0]) for i, l in enumerate(txt) if re.match(r"^_Z\w+:", l)]
    names.append((len(txt), None))
    nk = 0
    for (a, n), (b, _) in zip(names, names[1:]):
        if sub and sub not in n: continue
        body = txt[a:b]
        if not any("v_mfma_i32_32x32x32" in l for l in body): continue
        nk += 1
        total += check(body, n[:60])
    print(f"{nk} kernels with v_mfma_i32_32x32x32_i8 checked, {total} violations")
    sys.exit(1 if total else 0)
