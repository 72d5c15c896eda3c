#!/usr/bin/env python3
"""Generates scripts/ubench_tile.hip: what ONE int8 32x32x32 MFMA tile + its two exact FMAs per (row, token, 32-group) triple cost on a
gfx950 SIMD when the instruction stream is hand-placed (physical registers, two MFMA result sets: the FMAs of tile n-1 run under
the MFMA of tile n).  Variants: extra plain VALU ops per tile, LDS reads per tile, packed vs plain FMAs, 1 or 2 waves per SIMD.
This is the design probe for the round-4 big-tile kernel."""
import sys

def body(pk, extra, lds, nops, extra_pos="end", srcc0=False, salu=0, split=False, mix=False):
    """one loop iteration = 4 tiles (a 2x2 wave tile, one 32-group); registers: acc v[0:63], C0 v[64:79], C1 v[80:95], magic v[96:111],
    A frags v[112:119], B frags v[120:127], d8 v[128:159] (two token tiles x 16), dw/nm v[160:163], scratch v[164:171], lds addr v172"""
    L = []
    for t in range(4):
        cur, prev = (64, 80) if t % 2 == 0 else (80, 64)
        a = 112 + 4 * (t & 1)
        b = 120 + 4 * (t >> 1)
        L.append(f"v_mfma_i32_32x32x32_i8 v[{cur}:{cur+15}], v[{a}:{a+3}], v[{b}:{b+3}], " + ("0" if srcc0 else "v[96:111]"))
        if nops:
            L.append(f"s_nop {nops - 1}")
        for j in range(salu):
            L.append(f"s_add_u32 s{24 + (j & 3)}, s{24 + (j & 3)}, 1")
        if extra_pos == "after_mfma":
            for j in range(extra):
                L.append(f"v_and_b32 v{164 + (j & 7)}, 0x0f0f0f0f, v{112 + (j & 7)}")
        pt = (t - 1) % 4            # the tile whose results sit in `prev`
        accb = 16 * pt
        d8 = 128 + 16 * (pt >> 1)
        dw = 160 + 2 * (pt & 1)
        for j in range(lds):
            L.append(f"ds_read_b128 v[{164 + 4 * (j & 1)}:{167 + 4 * (j & 1)}], v172 offset:{256 * j}")
        if pk and split:   # interleave: first stage of register pair i, second stage of pair i - 4
            for i in range(0, 24, 2):
                if i < 16:
                    L.append(f"v_pk_fma_f32 v[{prev+i}:{prev+i+1}], v[{prev+i}:{prev+i+1}], v[{dw}:{dw+1}], v[{dw}:{dw+1}] op_sel:[0,0,1] op_sel_hi:[1,0,1]")
                if i >= 8:
                    j = i - 8
                    L.append(f"v_pk_fma_f32 v[{accb+j}:{accb+j+1}], v[{prev+j}:{prev+j+1}], v[{d8+j}:{d8+j+1}], v[{accb+j}:{accb+j+1}]")
        elif pk and mix:
            for i in range(0, 16, 2):
                L.append(f"v_pk_fma_f32 v[{prev+i}:{prev+i+1}], v[{prev+i}:{prev+i+1}], v[{dw}:{dw+1}], v[{dw}:{dw+1}] op_sel:[0,0,1] op_sel_hi:[1,0,1]")
            for i in range(16):
                L.append(f"v_fma_mix_f32 v{accb+i}, v{prev+i}, v{d8+(i>>1)}, v{accb+i} op_sel:[0,{i&1},0] op_sel_hi:[0,1,0]")
        elif pk:
            for i in range(0, 16, 2):
                L.append(f"v_pk_fma_f32 v[{prev+i}:{prev+i+1}], v[{prev+i}:{prev+i+1}], v[{dw}:{dw+1}], v[{dw}:{dw+1}] op_sel:[0,0,1] op_sel_hi:[1,0,1]")
            for i in range(0, 16, 2):
                L.append(f"v_pk_fma_f32 v[{accb+i}:{accb+i+1}], v[{prev+i}:{prev+i+1}], v[{d8+i}:{d8+i+1}], v[{accb+i}:{accb+i+1}]")
        else:
            for i in range(16):
                L.append(f"v_fma_f32 v{prev+i}, v{prev+i}, v{dw}, v{dw+1}")
            for i in range(16):
                L.append(f"v_fma_f32 v{accb+i}, v{prev+i}, v{d8+i}, v{accb+i}")
        if extra_pos == "end":
            for j in range(extra):
                L.append(f"v_and_b32 v{164 + (j & 7)}, 0x0f0f0f0f, v{112 + (j & 7)}")
        if lds:
            L.append("s_waitcnt lgkmcnt(0)")
    return L

def kernel(name, pk, extra, lds, nops, **kw):
    lines = ["s_mov_b32 s20, %[iters]", "s_memtime s[28:29]", f"L_{name}%=:"] + body(pk, extra, lds, nops, **kw) + ["s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", f"s_cbranch_scc1 L_{name}%=", "s_memtime s[30:31]", "s_waitcnt lgkmcnt(0)", "s_sub_u32 %[cyc], s30, s28"]
    asm = "\n".join(f'      "{l}\\n"' for l in lines)
    clob = ", ".join(f'"v{i}"' for i in range(64, 96)) + ", " + ", ".join(f'"v{i}"' for i in range(164, 172))
    return f'''
__global__ void __launch_bounds__(256, 2) {name}(float* out, int iters, float s) {{
  extern __shared__ char lds[];
  v32f acc0, acc1; v16i magic; v8i af, bf; v32f d8; v4f dwn; 
  for (int i = 0; i < 32; ++i) {{ acc0[i] = i; acc1[i] = -i; d8[i] = s + i; }}
  for (int i = 0; i < 16; ++i) magic[i] = 0x4B400000;
  for (int i = 0; i < 8; ++i) {{ af[i] = threadIdx.x * 0x01010101 + i; bf[i] = 0x01020304 * (i + 1); }}
  dwn = v4f{{s, -12582912.0f * s, s * 0.5f, -12582912.0f * s * 0.5f}};
  int ldsa = (threadIdx.x & 31) * 16;
  unsigned cyc;
  asm volatile(
{asm}
      : "+{{v[0:31]}}"(acc0), "+{{v[32:63]}}"(acc1), [cyc] "=s"(cyc)
      : "{{v[96:111]}}"(magic), "{{v[112:119]}}"(af), "{{v[120:127]}}"(bf), "{{v[128:159]}}"(d8), "{{v[160:163]}}"(dwn), "{{v172}}"(ldsa), [iters] "s"(iters)
      : "memory", "scc", "s20", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", {clob});
  if (blockIdx.x == 0 && threadIdx.x == 0) g_cycles = cyc;
  float r = 0;
  for (int i = 0; i < 32; ++i) r += acc0[i] + acc1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}}
'''

VARIANTS = [("pk_e0", 1, 0, 0, 0, {}), ("pl_e0", 0, 0, 0, 0, {}), ("pk_e0_n1", 1, 0, 0, 1, {}), ("pk_e0_n2", 1, 0, 0, 2, {}), ("pk_e0_n4", 1, 0, 0, 4, {}),
            ("pk_e0_n8", 1, 0, 0, 8, {}), ("pk_e0_n16", 1, 0, 0, 16, {}), ("pl_e0_n4", 0, 0, 0, 4, {}), ("pk_e0_c0", 1, 0, 0, 0, dict(srcc0=True)),
            ("pk_e0_c0n4", 1, 0, 0, 4, dict(srcc0=True)), ("pk_e0_s4", 1, 0, 0, 0, dict(salu=4)), ("pk_e4_am", 1, 4, 0, 0, dict(extra_pos="after_mfma")),
            ("pk_e8_am", 1, 8, 0, 0, dict(extra_pos="after_mfma")), ("pk_e8", 1, 8, 0, 0, {}), ("pk_e8_n4", 1, 8, 0, 4, {}), ("pk_e0_sp", 1, 0, 0, 0, dict(split=True)),
            ("pk_e0_spn4", 1, 0, 0, 4, dict(split=True)), ("pk_e8_l2n4", 1, 8, 2, 4, {}), ("pk_e8_amn4", 1, 8, 0, 4, dict(extra_pos="after_mfma")),
            ("mix_e0", 1, 0, 0, 0, dict(mix=True)), ("mix_e8_am", 1, 8, 0, 0, dict(mix=True, extra_pos="after_mfma")), ("pl_e8_am", 0, 8, 0, 0, dict(extra_pos="after_mfma")),
            ("pk_e12_am", 1, 12, 0, 0, dict(extra_pos="after_mfma")), ("pk_e16_am", 1, 16, 0, 0, dict(extra_pos="after_mfma"))]

src = '''// GENERATED by scripts/gen_ubench_tile.py — do not edit.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v32f __attribute__((ext_vector_type(32)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
__device__ unsigned g_cycles;
'''
for v in VARIANTS:
    src += kernel(*v[:5], **v[5])
src += '''
template <typename K> float timeit(K kern, int grid, float* out, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 16384, 0, out, 64, 1.0001f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 16384, 0, out, iters, 1.0001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  float* out; hipMalloc(&out, 256 * 4 * 256 * 4);
  const int iters = 4000;
'''
for name, pk, extra, lds, nops, kw in VARIANTS:
    src += f'''  printf("%-12s fma=%s extra_valu=%2d lds_reads=%d nops=%2d %-24s:", "{name}", "{'pk' if pk else 'plain'}", {extra}, {lds}, {nops}, "{' '.join(f'{k}={v}' for k, v in kw.items())}");
  for (int w = 1; w <= 2; ++w) {{
    float ms = timeit({name}, 256 * w, out, iters);
    unsigned cyc; hipMemcpyFromSymbol(&cyc, HIP_SYMBOL(g_cycles), 4);
    printf("  w%d %6.1f ns %6.1f cyc(wave)", w, ms * 1e6 / ((double)iters * 4 * w), cyc / ((double)iters * 4));
  }}
  printf("   per tile per SIMD\\n");
'''
src += "  return 0;\n}\n"
open(sys.argv[1] if len(sys.argv) > 1 else "scripts/ubench_tile.hip", "w").write(src)
