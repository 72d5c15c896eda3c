"""Instruction mix of the largest loop of a kernel in an AMDGPU .s file.
usage: python scripts/isa_loop.py file.s mangled_kernel_name [--dump]"""
import re, sys, collections
src = open(sys.argv[1]).read()
name = sys.argv[2]
i = src.index("\n" + name + ":"); j = src.index("s_endpgm", i)
body = src[i:j].split("\n")
labels = {}
for n, l in enumerate(body):
    m = re.match(r"\s*(\.LBB\d+_\d+):", l)
    if m: labels[m.group(1)] = n
best = None
for n, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < n:
        span = (labels[m.group(1)], n)
        if best is None or span[1] - span[0] > best[1] - best[0]: best = span
if best is None:
    print("no loop found; total lines", len(body)); sys.exit()
loop = [l.strip() for l in body[best[0]:best[1] + 1] if l.strip() and not l.strip().startswith((".", ";"))]
c = collections.Counter()
for l in loop:
    op = l.split()[0]
    if op.startswith("v_mfma"): c["mfma"] += 1
    elif op.startswith("v_"): c["valu"] += 1
    elif op.startswith("s_waitcnt"): c["waitcnt"] += 1
    elif op.startswith("s_"): c["salu"] += 1
    elif op.startswith("ds_"): c["lds"] += 1
    elif op.startswith(("global_", "buffer_", "scratch_")): c["vmem:" + op] += 1
    else: c[op] += 1
print(len(loop), dict(c))
print(collections.Counter(l.split()[0] for l in loop if l.startswith("v_")).most_common(30))
if "--dump" in sys.argv: print("\n".join(loop))
