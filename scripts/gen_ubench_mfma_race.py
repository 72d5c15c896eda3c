#!/usr/bin/env python3
"""Generates scripts/ubench_mfma_race.hip: is a VALU read of an int8 MFMA's result safe after a FIXED number of the wave's own
instructions when ANOTHER wave on the same SIMD keeps the matrix pipe busy?  Victim waves (blocks 0..255) loop
    MFMA -> D filler vector instructions -> read the result, compare with the expected value (it alternates, so a stale register shows)
while hog waves (blocks 256..511, co-resident: two 256-thread blocks per CU = two waves per SIMD) issue back-to-back MFMAs,
MFMA + packed FMAs, or nothing.  Prints the number of wrong lanes x registers seen per (filler count, hog kind)."""
import sys
DIST = [8, 12, 16, 20, 24, 28, 32, 40, 48, 64]
HOGS = ["none", "mfma", "mfma_valu", "valu"]

FILL_KIND = "pk"
def fill(n, base=128):
    if FILL_KIND == "mix":
        return [f"v_fma_mix_f32 v{base + (i % 16)}, v{base + (i % 16)}, v144, v{base + (i % 16)} op_sel:[0,{i & 1},0] op_sel_hi:[0,1,0]" for i in range(n)]
    if FILL_KIND == "misc":
        ops = ["v_cvt_f32_ubyte0 v{d}, v145", "v_and_b32 v{d}, 0x0f0f0f0f, v146", "v_lshrrev_b32 v{d}, 4, v147", "v_mul_u32_u24 v{d}, 7282, v145",
               "v_cvt_f32_f16_sdwa v{d}, v144 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1", "v_cvt_pkrtz_f16_f32 v{d}, v144, v145",
               "v_mad_u32_u24 v{d}, v145, s23, v146", "v_and_or_b32 v{d}, v145, s23, v146"]
        return [ops[i % len(ops)].format(d=base + (i % 16)) for i in range(n)]
    return [f"v_pk_fma_f32 v[{base + 2 * (i % 8)}:{base + 1 + 2 * (i % 8)}], v[{base + 2 * (i % 8)}:{base + 1 + 2 * (i % 8)}], v[144:145], v[146:147]" for i in range(n)]

PK_READ = False
def check(cbase, expect_reg):
    L = []
    if PK_READ:   # as in the GEMM: the result is consumed by in-place packed FMAs  t = fma(12582912 + C, 1.0, -12582912) = float(C)
        for j in range(0, 16, 2):
            L.append(f"v_pk_fma_f32 v[{cbase + j}:{cbase + j + 1}], v[{cbase + j}:{cbase + j + 1}], v[178:179], v[180:181]")
        for r in range(16):
            L += [f"v_cmp_neq_f32 vcc, v{cbase + r}, v{expect_reg + 34}", "s_bcnt1_i32_b64 s22, vcc", "s_add_u32 s21, s21, s22"]
        return L
    for r in range(16):
        L += [f"v_cmp_ne_u32 vcc, v{cbase + r}, v{expect_reg}", "s_bcnt1_i32_b64 s22, vcc", "s_add_u32 s21, s21, s22"]
    return L

def kernel2(d, hog, k1=24):
    """two MFMAs in flight: the first one's result is read d fillers after the SECOND one was issued"""
    name = f"k2_d{d}_{hog}"
    victim = ["s_mov_b32 s20, %[iters]", f"L_v_{name}%=:"]
    for (bsel, exp) in (("v[116:119]", 148), ("v[120:123]", 149)):
        victim += [f"v_mfma_i32_32x32x32_i8 v[64:79], v[112:115], {bsel}, v[96:111]"] + fill(k1)
        victim += [f"v_mfma_i32_32x32x32_i8 v[80:95], v[112:115], {bsel}, v[96:111]"] + fill(d)
        victim += check(64, exp) + fill(16) + check(80, exp)
    victim += ["s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", f"s_cbranch_scc1 L_v_{name}%="]
    return name, victim

def memops(nv, nl):
    L = [f"global_load_dwordx4 v[{152 + 4 * (i % 4)}:{155 + 4 * (i % 4)}], v[150:151], off offset:{1024 * (i % 4)}" for i in range(nv)]
    L += [f"ds_read_b128 v[{168 + 4 * (i % 2)}:{171 + 4 * (i % 2)}], v176 offset:{4096 * (i % 4)}" for i in range(nl)]
    return L

def kernel3(d, hog, nv, nl, k1=24):
    """two MFMAs in flight + nv vector loads and nl LDS reads returning data while the MFMAs execute"""
    name = f"k3_d{d}_{hog}_v{nv}_l{nl}"
    victim = ["s_mov_b32 s20, %[iters]", f"L_v_{name}%=:"]
    for (bsel, exp) in (("v[116:119]", 148), ("v[120:123]", 149)):
        victim += memops(nv, nl)
        victim += [f"v_mfma_i32_32x32x32_i8 v[64:79], v[112:115], {bsel}, v[96:111]"] + fill(k1)
        victim += memops(nv, nl)
        victim += [f"v_mfma_i32_32x32x32_i8 v[80:95], v[112:115], {bsel}, v[96:111]"] + fill(d)
        victim += check(64, exp) + fill(16) + check(80, exp)
        victim += ["s_waitcnt vmcnt(0) lgkmcnt(0)"]
    victim += ["s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", f"s_cbranch_scc1 L_v_{name}%="]
    return name, victim

def kernel(d, hog, two=False, mem=None):
    name = f"k_d{d}_{hog}"
    if mem:
        name, victim = kernel3(d, hog, *mem)
        return finish(name, victim, hog, mem)
    if two:
        name, victim = kernel2(d, hog)
        return finish(name, victim, hog)
    victim = ["s_mov_b32 s20, %[iters]", f"L_v_{name}%=:",
              # even iteration: B = b1 (ones) -> C = magic + 32 ; odd: B = b2 (twos) -> magic + 64
              "v_mfma_i32_32x32x32_i8 v[64:79], v[112:115], v[116:119], v[96:111]"]
    for i in range(d):
        victim.append(f"v_pk_fma_f32 v[{128 + 2 * (i % 8)}:{129 + 2 * (i % 8)}], v[{128 + 2 * (i % 8)}:{129 + 2 * (i % 8)}], v[144:145], v[146:147]")
    for r in range(16):
        victim.append(f"v_cmp_ne_u32 vcc, v{64 + r}, v148")        # expected (magic + 32)
        victim.append("s_bcnt1_i32_b64 s22, vcc")
        victim.append("s_add_u32 s21, s21, s22")
    victim += ["v_mfma_i32_32x32x32_i8 v[64:79], v[112:115], v[120:123], v[96:111]"]
    for i in range(d):
        victim.append(f"v_pk_fma_f32 v[{128 + 2 * (i % 8)}:{129 + 2 * (i % 8)}], v[{128 + 2 * (i % 8)}:{129 + 2 * (i % 8)}], v[144:145], v[146:147]")
    for r in range(16):
        victim.append(f"v_cmp_ne_u32 vcc, v{64 + r}, v149")        # expected (magic + 64)
        victim.append("s_bcnt1_i32_b64 s22, vcc")
        victim.append("s_add_u32 s21, s21, s22")
    victim += ["s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", f"s_cbranch_scc1 L_v_{name}%="]
    return finish(name, victim, hog)

def finish(name, victim, hog, mem=None):
    hogl = ["s_mov_b32 s20, %[iters]", "s_lshl_b32 s20, s20, 1", f"L_h_{name}%=:"]
    for rep in range(4):
        if hog in ("mfma", "mfma_valu"):
            hogl.append(f"v_mfma_i32_32x32x32_i8 v[{64 if rep % 2 == 0 else 80}:{79 if rep % 2 == 0 else 95}], v[112:115], v[116:119], v[96:111]")
        if hog in ("valu", "mfma_valu"):
            for i in range(12):
                hogl.append(f"v_pk_fma_f32 v[{128 + 2 * (i % 8)}:{129 + 2 * (i % 8)}], v[{128 + 2 * (i % 8)}:{129 + 2 * (i % 8)}], v[144:145], v[146:147]")
        if hog == "none":
            hogl.append("s_nop 7")
        if mem:
            hogl += memops(*mem) + ["s_waitcnt vmcnt(8) lgkmcnt(4)"]
    hogl += ["s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", f"s_cbranch_scc1 L_h_{name}%="]
    va = "\n".join(f'        "{l}\\n"' for l in victim)
    ha = "\n".join(f'        "{l}\\n"' for l in hogl)
    clob = ", ".join(f'"v{i}"' for i in list(range(64, 96)) + list(range(128, 144)) + list(range(152, 176)))
    return name, f'''
__global__ void __launch_bounds__(256, 2) {name}(unsigned* out, int iters, const int* gbuf) {{
  extern __shared__ char lds[];
  const int* gp = gbuf + (blockIdx.x % 64) * 4096 + (threadIdx.x & 63) * 4;
  unsigned ldsa = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16;
  v16i magic; v4i a, b1, b2; v4f fm; unsigned e1 = 0x4B400000u + 32u, e2 = 0x4B400000u + 64u;
  v4f one_negm = v4f{{1.0f, 1.0f, -12582912.0f, -12582912.0f}}; float f1 = 32.0f, f2 = 64.0f;
  for (int i = 0; i < 16; ++i) magic[i] = 0x4B400000;
  for (int i = 0; i < 4; ++i) {{ a[i] = 0x01010101; b1[i] = 0x01010101; b2[i] = 0x02020202; }}
  fm = v4f{{1.0001f, 1.0001f, 0.5f, 0.5f}};
  unsigned err = 0;
  if (blockIdx.x < 256) {{
    asm volatile(
        "s_mov_b32 s21, 0\\n"
{va}
        "s_mov_b32 %[err], s21\\n"
        : [err] "=s"(err)
        : "{{v[96:111]}}"(magic), "{{v[112:115]}}"(a), "{{v[116:119]}}"(b1), "{{v[120:123]}}"(b2), "{{v[144:147]}}"(fm), "{{v148}}"(e1), "{{v149}}"(e2), "{{v[150:151]}}"(gp), "{{v176}}"(ldsa), "{{v[178:181]}}"(one_negm), "{{v182}}"(f1), "{{v183}}"(f2), [iters] "s"(iters)
        : "memory", "scc", "vcc", "s20", "s21", "s22", "s23", {clob});
    if ((threadIdx.x & 63) == 0) atomicAdd(out, err);
  }} else {{
    asm volatile(
{ha}
        :
        : "{{v[96:111]}}"(magic), "{{v[112:115]}}"(a), "{{v[116:119]}}"(b1), "{{v[144:147]}}"(fm), "{{v[150:151]}}"(gp), "{{v176}}"(ldsa), [iters] "s"(iters)
        : "memory", "scc", "vcc", "s20", {clob});
  }}
}}
'''

src = '''// GENERATED by scripts/gen_ubench_mfma_race.py — do not edit.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
'''
names = []
DIST2 = [0, 2, 4, 8, 12, 16, 24, 32]
for hog in HOGS:
    for d in DIST:
        n, s = kernel(d, hog)
        names.append((n, d, hog))
        src += s
names3 = []
for hog in ("mfma_valu",):
    for mem in ((4, 0), (0, 4), (4, 2), (8, 4)):
        for d in (0, 4, 8, 16, 32):
            n, s = kernel(d, hog, mem=mem)
            names3.append((n, d, hog, mem))
            src += s
names4 = []
for kind in ("mix", "misc"):
    FILL_KIND = kind
    for hog in ("mfma_valu", "mfma"):
        for d in (0, 4, 8, 16):
            n, s = kernel(d, hog, two=True)
            n2 = n + "_" + kind
            names4.append((n2, d, hog, kind))
            src += s.replace(n, n2)
FILL_KIND = "pk"
names5 = []
PK_READ = True
for kind in ("pk", "mix"):
    FILL_KIND = kind
    for hog in ("mfma_valu", "mfma"):
        for d in (0, 4, 8, 16):
            n, s = kernel(d, hog, two=True)
            n2 = n + "_pkread_" + kind
            names5.append((n2, d, hog, kind))
            src += s.replace(n, n2)
PK_READ = False
FILL_KIND = "pk"
names2 = []
for hog in HOGS:
    for d in DIST2:
        n, s = kernel(d, hog, two=True)
        names2.append((n, d, hog))
        src += s
src += '''
int main() {
  unsigned* out; hipMalloc(&out, 4);
  int* gbuf; hipMalloc(&gbuf, 64 * 4096 * 4 + 65536); hipMemset(gbuf, 1, 64 * 4096 * 4 + 65536);
  const int iters = 20000;
'''
for hog in HOGS:
    src += f'  printf("hog %-10s wrong (lane, register) reads per filler count:", "{hog}");\n'
    for n, d, h in names:
        if h != hog:
            continue
        src += f'''  {{ hipMemset(out, 0, 4); hipLaunchKernelGGL({n}, dim3(512), dim3(256), 32768, 0, out, iters, gbuf); unsigned e; hipMemcpy(&e, out, 4, hipMemcpyDeviceToHost); printf("  d={d}: %u", e); }}\n'''
    src += '  printf("\\n");\n'
for hog in HOGS:
    src += f'  printf("two in flight, hog %-10s wrong reads per filler count after the 2nd MFMA:", "{hog}");\n'
    for n, d, h in names2:
        if h != hog:
            continue
        src += f'''  {{ hipMemset(out, 0, 4); hipLaunchKernelGGL({n}, dim3(512), dim3(256), 32768, 0, out, iters, gbuf); unsigned e; hipMemcpy(&e, out, 4, hipMemcpyDeviceToHost); printf("  d={d}: %u", e); }}\n'''
    src += '  printf("\\n");\n'
for kind in ("mix", "misc"):
    for hog in ("mfma_valu", "mfma"):
        src += f'  printf("two in flight, fillers {kind}, hog {hog}:");\n'
        for n, d, h, k in names4:
            if h != hog or k != kind:
                continue
            src += f'''  {{ hipMemset(out, 0, 4); hipLaunchKernelGGL({n}, dim3(512), dim3(256), 32768, 0, out, iters, gbuf); unsigned e; hipMemcpy(&e, out, 4, hipMemcpyDeviceToHost); printf("  d={d}: %u", e); }}\n'''
        src += '  printf("\\n");\n'
for kind in ("pk", "mix"):
    for hog in ("mfma_valu", "mfma"):
        src += f'  printf("two in flight, result consumed by in-place v_pk_fma_f32, fillers {kind}, hog {hog}:");\n'
        for n, d, h, k in names5:
            if h != hog or k != kind:
                continue
            src += f'''  {{ hipMemset(out, 0, 4); hipLaunchKernelGGL({n}, dim3(512), dim3(256), 32768, 0, out, iters, gbuf); unsigned e; hipMemcpy(&e, out, 4, hipMemcpyDeviceToHost); printf("  d={d}: %u", e); }}\n'''
        src += '  printf("\\n");\n'
for mem in ((4, 0), (0, 4), (4, 2), (8, 4)):
    src += f'  printf("two in flight + {mem[0]} vector loads + {mem[1]} LDS reads per MFMA, hog mfma_valu (+ the same loads):");\n'
    for n, d, h, m in names3:
        if m != mem:
            continue
        src += f'''  {{ hipMemset(out, 0, 4); hipLaunchKernelGGL({n}, dim3(512), dim3(256), 32768, 0, out, iters, gbuf); unsigned e; hipMemcpy(&e, out, 4, hipMemcpyDeviceToHost); printf("  d={d}: %u", e); }}\n'''
    src += '  printf("\\n");\n'
src += "  return 0;\n}\n"
open(sys.argv[1] if len(sys.argv) > 1 else "scripts/ubench_mfma_race.hip", "w").write(src)
