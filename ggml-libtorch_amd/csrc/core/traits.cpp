// traits.cpp — format traits / error strings of the C ABI (include/ggq.h).
// Host-only; compiled into both libggq_hip.so and libggq_cpu.so.
#include "../../../include/ggq.h"

extern "C" int ggq_abi_version(void) { return GGQ_ABI_VERSION; }

extern "C" const char* ggq_strerror(int status) {
  switch (status) {
    case GGQ_OK: return "ok";
    case GGQ_ERR_TYPE: return "unsupported ggml quantisation type";
    case GGQ_ERR_SHAPE: return "shape is not compatible with the block format";
    case GGQ_ERR_DTYPE: return "unsupported activation dtype (float32, float16, bfloat16 only)";
    case GGQ_ERR_ARG: return "invalid argument (null pointer or negative size)";
    case GGQ_ERR_LAUNCH: return "HIP kernel launch failed";
    case GGQ_ERR_ALIGN: return "pointer alignment below the ABI contract";
    default: return "unknown ggq status";
  }
}

// ggml_get_block_size, HK/ggml/mmq.cu:57-81
extern "C" int ggq_block_elems(int type) {
  switch (type) {
    case GGQ_TYPE_Q4_0: case GGQ_TYPE_Q4_1: case GGQ_TYPE_Q5_0: case GGQ_TYPE_Q5_1:
    case GGQ_TYPE_Q8_0: case GGQ_TYPE_Q8_1: case GGQ_TYPE_IQ4_NL:
      return 32;
    case GGQ_TYPE_Q2_K: case GGQ_TYPE_Q3_K: case GGQ_TYPE_Q4_K: case GGQ_TYPE_Q5_K:
    case GGQ_TYPE_Q6_K: case GGQ_TYPE_IQ4_XS:
    case GGQ_TYPE_IQ2_XXS: case GGQ_TYPE_IQ2_XS: case GGQ_TYPE_IQ2_S: case GGQ_TYPE_IQ3_XXS: case GGQ_TYPE_IQ3_S:
    case GGQ_TYPE_IQ1_S: case GGQ_TYPE_IQ1_M:
      return 256;
    default: return 0;
  }
}

extern "C" int ggq_block_bytes(int type) {
  switch (type) {
    case GGQ_TYPE_Q4_0: return 18;
    case GGQ_TYPE_Q4_1: return 20;
    case GGQ_TYPE_Q5_0: return 22;
    case GGQ_TYPE_Q5_1: return 24;
    case GGQ_TYPE_Q8_0: return 34;
    case GGQ_TYPE_Q8_1: return 36;
    case GGQ_TYPE_Q2_K: return 84;
    case GGQ_TYPE_Q3_K: return 110;
    case GGQ_TYPE_Q4_K: return 144;
    case GGQ_TYPE_Q5_K: return 176;
    case GGQ_TYPE_Q6_K: return 210;
    case GGQ_TYPE_IQ4_NL: return 18;    // block_iq4_nl, HK/ggml/ggml-common.h:176-182
    case GGQ_TYPE_IQ4_XS: return 136;   // block_iq4_xs, HK/ggml/ggml-common.h:184-191
    case GGQ_TYPE_IQ2_XXS: return 66;   // block_iq2_xxs ... block_iq1_m, HK/ggml/ggml-common.h:108-168
    case GGQ_TYPE_IQ2_XS: return 74;
    case GGQ_TYPE_IQ2_S: return 82;
    case GGQ_TYPE_IQ3_XXS: return 98;
    case GGQ_TYPE_IQ3_S: return 110;
    case GGQ_TYPE_IQ1_S: return 50;
    case GGQ_TYPE_IQ1_M: return 56;
    default: return 0;
  }
}

extern "C" int64_t ggq_row_bytes(int type, int64_t k) {
  const int qk = ggq_block_elems(type);
  if (qk == 0) return GGQ_ERR_TYPE;
  if (k < 0 || k % qk) return GGQ_ERR_SHAPE;
  return k / qk * ggq_block_bytes(type);
}

extern "C" int ggq_type_supported(int type) {
  return type != GGQ_TYPE_Q8_1 && ggq_block_elems(type) != 0;
}

// the ten cases of ggml_mul_mat_a8's switch (HK/ggml/mmq.cu:222-251): no IQ format
extern "C" int ggq_mmq_type_supported(int type) {
  return ggq_type_supported(type) && !(type >= GGQ_TYPE_IQ2_XXS && type <= GGQ_TYPE_IQ4_XS) && type != GGQ_TYPE_IQ1_M;
}

// mmq_need_sum, HK/ggml/mmq.cu:84-106
extern "C" int ggq_mmq_need_sum(int type) {
  switch (type) {
    case GGQ_TYPE_Q4_0: case GGQ_TYPE_Q4_1: case GGQ_TYPE_Q5_1: case GGQ_TYPE_Q4_K:
    case GGQ_TYPE_Q5_K:
      return 1;
    default: return 0;
  }
}

extern "C" int64_t ggq_mmvq_padded_k(int64_t k) { return (k + 511) / 512 * 512; }   // ggml_kernel.cu:84
extern "C" int64_t ggq_mmq_padded_k(int64_t k) { return k - k % 512 + 512; }         // mmq.cu:190-191

extern "C" size_t ggq_mmvq_scratch_bytes(int64_t k) {
  return (size_t)(ggq_mmvq_padded_k(k) / 32 * 36);
}
extern "C" size_t ggq_mmq_scratch_bytes(int64_t batch, int64_t k) {
  // whole 32-token tiles: the fragment-major layout (ggq_quantize_q8_1_tiled) addresses tiles
  const size_t tiled = (size_t)((batch + 31) / 32 * 32) * (size_t)(ggq_mmq_padded_k(k) / 32 * 36);
  // the x64 layout (ggq_quantize_q8_1_x64): 10240-byte records per (256 elements, 32 tokens), token tiles in pairs
  const size_t x64 = k % 256 ? 0 : (size_t)((batch + 63) / 64 * 2) * (size_t)(k / 256) * 10240;
  return tiled > x64 ? tiled : x64;
}

// the 64 x 64 wave-tile kernel (csrc/hip/mmq_x64.hip)
extern "C" int ggq_mmq_x64_type_supported(int type) {
  switch (type) {
    case GGQ_TYPE_Q4_K: case GGQ_TYPE_Q8_0: case GGQ_TYPE_Q4_0: case GGQ_TYPE_Q5_K: return 1;
    default: return 0;
  }
}
extern "C" int ggq_mmq_x64_k_slices(int64_t batch, int64_t k, int64_t n_rows) {
  const int64_t units = ((n_rows + 63) / 64) * ((batch + 63) / 64);
  if (units >= 2048) return 1;   // a full round of one-wave workgroups: no K-slicing (mmq_x64.hip launch_x64; 64-row units only: Q5_K's 32-row units keep 4)
  return units <= 256 && k >= 8 * 256 ? 8 : 4;
}

// rows of one unit: 96 (four two-row-tile waves + four one-row-tile waves, one workgroup per CU) where the launch then is ONE round of
// at most 256 workgroups while 64-row units would put two workgroups on some CUs and one on the others (the two-workgroup CUs finish
// 40-60 % later: profiles/r04_x64_stamps.txt; the sweep behind the rule: profiles/r04_x64_unit_rows.txt)
// 32 (every wave a one-row-tile wave: twice the units, half the work each) where the launch has fewer than 160 units of 64 rows — too few
// to fill the chip; against the better of the streamed kernel and 64-row units (profiles/r04b_x64_unit_rows32_*.txt, cold): Q8_0 0.75 - 0.96
// and Q4_0 0.76 - 0.93 from 32 units of 64 up, Q4_K 0.87 - 0.94 at 96 - 128 units (level with the streamed kernel below) and 1.03 - 1.2 from
// 168 units on for all three, where 64- / 96-row units stay.  Bit-identical to the other unit shapes (same loops, same slice order).
extern "C" int ggq_mmq_x64_unit_rows(int type, int64_t batch, int64_t k, int64_t n_rows) {
  if (!ggq_mmq_x64_type_supported(type)) return 64;
  if (type == GGQ_TYPE_Q5_K) return 32;   // its 176-byte super-blocks fit the LDS as 32-row stages only (mmq_x64.hip, X64Fmt)
  if (ggq_mmq_x64_tile_tokens(type, batch, k, n_rows) == 32) return 32;   // 32-token tiles: the one-tile loops (32 rows x 32 tokens per wave)
  const int64_t tt = (batch + 63) / 64;
  const int64_t u64 = ((n_rows + 63) / 64) * tt, u96 = ((n_rows + 95) / 96) * tt;
  if (u64 < 160) return 32;
  return k >= 4 * 256 && u64 > 256 && u96 <= 256 ? 96 : 64;
}
// tokens per wave tile: 32 (the one-tile loops: 32 rows x 32 tokens per wave) up to 32 tokens, and at 33 - 64 tokens while the launch then
// still has at most 256 units (4096 rows: one workgroup of eight K-slices per CU) — there two 32-token tiles per 32 rows beat one
// 64-token tile (profiles/r04b_x64_one_tile_b33_64.txt, op us cold: 4096 x 4096 Q4_K 13.5 -> 12.8, Q8_0 16.1 -> 15.0, Q5_K 14.7 -> 13.2, 3584 x 8192 Q8_0
// 24.8 -> 21.9; 2048 x 4096 -3 ... -8 %); from 6144 rows on (more than 256 units) they lose 20 - 40 %.  Else 64.
extern "C" int ggq_mmq_x64_tile_tokens(int type, int64_t batch, int64_t k, int64_t n_rows) {
  (void)k;
  if (!ggq_mmq_x64_type_supported(type)) return 64;
  if (batch <= 32) return 32;
  return batch <= 64 && ((n_rows + 31) / 32) * 2 <= 256 ? 32 : 64;
}
// fewest 32-row units from which the route takes the x64 kernel below 160 units of 64 rows (0: never)
static int64_t x64_min_units32(int type) {
  switch (type) {
    case GGQ_TYPE_Q4_K: return 192;
    case GGQ_TYPE_Q5_K: return 128;   // (32-row units only, profiles/r04b_x64_vs_stream_q5_k.txt: 1.01 - 1.26 x the streamed kernel cold from 128 units of 32)
    case GGQ_TYPE_Q8_0: case GGQ_TYPE_Q4_0: return 64;
    default: return 0;
  }
}

extern "C" int ggq_mmq_x64_supported(int type, int64_t k, int64_t batch) {
  if (!ggq_mmq_x64_type_supported(type) || k <= 0 || k % 256 || batch <= 0) return 0;
  if ((uint64_t)((batch + 63) / 64 * 2) * (uint64_t)(k / 256) * 10240 >= (1ull << 32)) return 0;   // 32-bit byte offsets into the scratch
  if ((uint64_t)ggq_row_bytes(type, k) * 64 >= (1ull << 32)) return 0;                             // ... and inside a wave's weight tile (64 rows at most)
  return 1;
}

// ---- which kernel ggq_mul_mat_q runs: the role of the reference's tile heuristic (mul_mat_q_case / get_mmq_x_max_host,
// HK/ggml/kernel_instances/mmq_kernel.cuh:21-32, mmq.cuh:155-164), decided from the format, the batch AND the shape.
// Host-only and exported so that the table can be tested without a GPU (tests/test_host_logic.py).
extern "C" int ggq_mmq_t16_type_supported(int type) {
  switch (type) {
    case GGQ_TYPE_Q4_K: case GGQ_TYPE_Q5_K: case GGQ_TYPE_Q8_0:
    case GGQ_TYPE_Q4_0: case GGQ_TYPE_Q4_1: case GGQ_TYPE_Q5_0: case GGQ_TYPE_Q5_1: case GGQ_TYPE_Q6_K: case GGQ_TYPE_Q3_K: return 1;
    default: return 0;
  }
}

extern "C" int ggq_mmq_t16_supported(int type, int64_t k, int64_t batch) {
  // 256-element units; 32-bit byte offsets into the activation scratch
  if (!ggq_mmq_t16_type_supported(type) || k <= 0 || k % 256 || batch <= 0) return 0;
  if ((ggq_block_elems(type) == 32 || type == GGQ_TYPE_Q6_K || type == GGQ_TYPE_Q3_K) && batch > 16) return 0;   // the 32-element-block formats and Q6_K have no two-token-tile instance
  if ((uint64_t)ggq_mmq_scratch_bytes(batch, k) >= (1ull << 31)) return 0;
  if ((uint64_t)ggq_row_bytes(type, k) * 16 >= (1ull << 32)) return 0;   // 32-bit byte offsets inside a 16-row weight tile
  return 1;
}

// Tokens per unit of the streamed kernel (32 rows x 32 or 64 tokens per workgroup).  64-token units read a weight tile once for two
// token blocks; 32-token units give twice as many, smaller units.  Measured (op, us warm, 32- against 64-token units, 11008 x 4096
// unless said otherwise; scripts/sweep_mmq_op.py on a -DGGQ_TUNING build, profiles/r03_stream_unit_tokens.txt):
//   batch 40 / 64   Q4_K 19.6-20.0 / 21.8 vs 20.9-21.4 / 21.6   (3584 x 8192: 17.4 / 18.3 vs 19.7 / 20.7; 4096 x 11008: 23.0 / 24.6 vs 25.9 / 27.4)
//                   Q5_K 22.4 / 23.0 vs 22.8 / 23.5     Q4_1 24.4 / 25.2 vs 26.8 / 27.6     Q5_1 26.2 / 26.8 vs 28.6 / 29.3
//                   Q4_0 24.4 / 24.9 vs 22.6 / 23.2     Q5_0 25.5 / 26.2 vs 24.2 / 24.5     Q3_K 31.9 / 32.4 vs 28.5 / 29.0     Q6_K 46.0 / 46.5 vs 35.7 / 36.0
//   (Q2_K: 32 always — its second int8 tile does not fit the registers with two token blocks; batch <= 32: one token block is the batch)
//   Q8_0 (kernel alone, scripts/route_audit.sh, profiles/r03_route_audit.txt): 3584 x 8192 batch 48 / 64 20.6 / 20.8 vs 23.6 / 23.7, 4096 x 11008
//                   26.7 / 26.3 vs 28.7 / 28.8; 11008 x 4096 30.1 / 30.4 vs 26.7 / 27.5 (where ggq_mmq_route keeps the LDS-tile kernel anyway)
//   batch 80 / 96: 64-token units win or tie for every format at every shape (same file)
//   By shape (scripts/route_audit4.sh / route_audit5.sh, profiles/r03_route_audit4.txt, _audit5.txt; kernel alone at batch 48, us warm / cold, 32- | 64-token units):
//     Q4_K K = 4096  rows 2048 8.5 / 9.1 | 10.1 / 10.6   4096 9.2 / 10.0 | 10.7 / 11.5   6144 13.3 / 14.4 | 11.1 / 12.8   8192 14.6 / 17.1 | 11.7 / 14.8
//                    14336 24.5 / 30.6 | 20.1 / 23.9   16384 26.0 / 32.4 | 21.8 / 25.1   28672 40.3 / 49.1 | 37.5 / 43.2      8192 x 8192 25.3 / 28.3 | 21.2 / 24.6
//     every other format at 4096 x 4096, 3584 x 8192, 4096 x 11008: 32-token units 8 - 20 % faster; at 8192 x 4096: 64-token units 15 - 25 % faster
//   i.e. the time follows the workgroups per CU: up to 4096 rows two 32-token units per row tile are still one workgroup per CU (and half the
//   work each); beyond that they are two or more where 64-token units are one, except in the band where 64-token units round badly
//   (256 < row tiles <= 384, e.g. 344 at 11008 rows) and the 32-token ones still fit one resident round (<= 768): there the four formats
//   measured above at 11008 x 4096 take 32, the others 64.
extern "C" int ggq_mmq_stream_unit_tokens(int type, int64_t batch, int64_t n_rows) {
  if (batch <= 32 || type == GGQ_TYPE_Q2_K) return 32;
  // one workgroup per CU at most with 32-token units: they win at every batch (route_audit6: batch 96 / 128, 1024 - 2048 rows: Q4_K 8.3 - 9.3 / 8.8 - 9.7
  // against 10.1 - 10.7 / 10.6 - 11.2, Q4_0 the same 13 - 18 %; 3072 rows, 288 - 384 units: 13.1 - 14.2 against 11.1 - 11.2) — at batch 33 - 64 that is "up to 4096 rows"
  const int64_t row_tiles = (n_rows + 31) / 32, tok_tiles = (batch + 31) / 32;
  if (row_tiles * tok_tiles <= 256) return 32;
  if (batch <= 64 && row_tiles > 256 && row_tiles <= 384 &&
      (type == GGQ_TYPE_Q4_K || type == GGQ_TYPE_Q5_K || type == GGQ_TYPE_Q4_1 || type == GGQ_TYPE_Q5_1)) return 32;
  return 64;
}

// Batch 17 - 32 (two 16-token tiles per wave) against the streamed kernel's 32-token units, by matrix shape — scripts/sweep_t16_vs_stream.py,
// profiles/r03_t16_vs_stream_b17_32.txt, op us warm / cold at batch 32, 16-token tiles | streamed:
//   Q4_K  K = 4096   rows 2048  8.7 /  9.4 | 11.7 / 12.8    4096  9.1 / 10.1 | 11.2 / 13.0    6144 16.7 / 17.9 | 11.4 / 13.9    8192 17.2 / 18.7 | 11.8 / 15.4
//                         11008 16.4 / 18.5 | 15.8 / 19.6   14336 21.8 / 24.1 | 17.2 / 20.9   16384 22.9 / 23.7 | 17.8 / 21.6   28672 36.0 / 35.0 | 30.0 / 37.3
//         K = 8192   rows 2048 12.5 / 13.8 | 16.0 / 18.6    4096 13.4 / 14.9 | 16.6 / 20.5    8192 30.3 / 31.4 | 19.5 / 22.3   16384 47.0 / 46.0 | 29.6 / 34.5
//                         28672 76.6 / 75.8 | 49.4 / 64.8      4096 x 11008 16.8 / 19.2 | 21.0 / 26.5      4096 x 14336 22.3 / 23.0 | 24.4 / 30.8
//   Q5_K  the same picture, 1 - 2 us higher on both sides (4096 x 11008: 18.0 / 20.3 | 24.1 / 28.5; 8192 x 4096: 19.3 / 20.8 | 12.6 / 15.9)
// The 16-token tiles need every workgroup resident at once: up to 4096 rows that is one workgroup per CU with 8 - 12 short K-slices, beyond it two
// or three with slices twice as long, and the kernel takes twice the time.  The streamed kernel's time goes with the units per CU, ceil(units / 256):
// it loses only where that rounding leaves a third of the CU-rounds empty (344 units on 256 CUs at 11008 rows: a tie).
static bool t16_two_tiles_pay(int64_t n_rows) {
  if (n_rows <= 4096) return true;
  const int64_t units = (n_rows + 31) / 32, per_cu = (units + 255) / 256;
  return units > 256 && units * 10 < per_cu * 256 * 7;
}

extern "C" int ggq_mmq_route(int type, int64_t batch, int64_t k, int64_t n_rows) {
  if (!ggq_mmq_type_supported(type) || batch <= 0 || k <= 0 || n_rows <= 0 || k % ggq_block_elems(type)) return GGQ_MMQ_ROUTE_NONE;
  // Batch 17 - 32 on many rows (Q4_K, Q5_K: the formats with a one-tile loop — 32 rows x 32 tokens per wave): from 256 units of 32 rows
  // (8192-row matrices) the x64 kernel is ahead of the 16-token tiles / the streamed kernel, op us warm / cold (profiles/r04b_x64_one_tile_b17_32.txt):
  // Q4_K 8192 x 4096 10.5 / 12.6 against 11.1 / 14.4, 14336 15.3 / 17.7 against 15.9 / 20.3, 28672 28.4 / 31.3 against 28.7 / 35.2, 11008 14.4 / 17.1 against
  // 15.4 / 17.6; Q5_K 8192 10.6 / 13.3 against 12.0 / 15.6; below (4096 x 4096, 3584 x 8192) it loses 15 - 25 %.
  // Q8_0 / Q4_0 (profiles/r04b_x64_one_tile_b17_32_q80_q40.txt; before: the LDS-tile / streamed kernels): ahead at all six measured shapes —
  // Q8_0 8192 x 4096 13.2 / 15.8 against 18.7 / 22.9, 4096 x 4096 10.8 / 14.1 against 14.2 / 18.5, 3584 x 8192 16.6 / 21.4 against 22.4 / 27.9, 11008 17.5 / 23.1 against
  // 21.2 / 25.2; Q4_0 8192 10.9 / 12.8 against 12.5 / 17.0, 4096 x 4096 10.6 / 12.1 against 12.3 / 16.3, 28672 level — from 112 units of 32 rows (3584 rows, the
  // smallest measured).
  if (batch >= 17 && batch <= 32 && ggq_mmq_x64_supported(type, k, batch)) {
    const int64_t u32 = (n_rows + 31) / 32;
    if ((type == GGQ_TYPE_Q4_K || type == GGQ_TYPE_Q5_K) ? u32 >= 256 : u32 >= 112) return GGQ_MMQ_ROUTE_X64;
  }
  // The HBM-bound batches.  Measured (scripts/sweep_t16.py, op = quantise + kernel, us warm / cold, old route -> 16-token tiles):
  //   Q4_K 11008 x 4096   b2  8.8/11.6 ->  9.0/12.0   b3 10.4/12.8 -> 8.9/12.0   b8 15.2/16.4 -> 9.3/12.2   b16 15.8/19.9 -> 11.6/13.9
  //                       b32 15.8/19.7 -> 15.0/17.9  b33 20.4/24.0 -> 23.0/25.1
  //   Q4_K 3584 x 8192    b2 11.8/13.9 ->  8.1/ 9.8   b8 22.5/23.8 -> 8.3/10.2   b32 16.5/20.4 -> 12.8/14.5   b64 20.1/24.0 -> 21.2/22.9
  //   Q4_K 4096 x 11008   b2 16.1/18.8 ->  9.5/12.0   b8 31.3/31.9 -> 9.8/12.2   b32 20.6/26.3 -> 15.1/17.9
  //   Q5_K 11008 x 4096   b8 15.2/18.7 -> 10.6/13.5   b16 18.0/20.2 -> 12.6/15.1   b32 15.8/20.2 -> 18.5/21.2
  // i.e. every shape from batch 2 (a tie at the one shape whose row count suits the dot4 kernel's 4096 waves) up to the
  // last batch one workgroup column covers without re-reading the weights: 32 tokens (two token tiles) for Q4_K, 16 for
  // Q5_K and Q8_0 whose two-tile instances sit at / beyond the register limit.
  // The 32-element-block formats (two request -> land -> compute rounds per wave for Q8_0, whose weights do not fit the chip's
  // LDS in one go; one token tile only: batch <= 16), same sweep:
  //   11008 x 4096   Q4_0 b2 9.4/12.3 -> 12.3/13.3   b4 10.8/13.1 -> 12.4/13.4   b5 13.6/15.9 -> 12.5/13.2   b8 14.0/16.2 -> 12.6/13.9   b16 17.2/20.9 -> 14.6/15.8
  //                  Q4_1 b2 8.3/11.6 -> 12.6/13.6   b8 14.7/15.8 -> 12.9/13.7   b16 19.6/21.5 -> 16.2/17.4
  //                  Q5_0 b8 15.1/16.8 -> 16.2/16.9   b16 18.4/23.3 -> 18.0/19.2     Q5_1 b8 15.8/17.1 -> 18.6/18.8   b16 19.7/23.1 -> 20.5/21.2
  //                  Q8_0 b2 12.8/17.0 -> 13.4/15.1   b4 13.2/15.4 -> 13.4/15.2   b8 15.5/18.9 -> 14.2/15.8   b16 22.0/25.5 -> 14.6/16.7
  //   3584 x 8192    Q4_0 b2 13.1/16.2 -> 9.6/10.0   b8 21.1/22.3 -> 9.7/10.1   b16 18.6/23.1 -> 11.5/11.9   Q5_0 b8 23.5/24.7 -> 11.9/12.3
  //                  Q5_1 b8 23.7/24.5 -> 13.5/14.3   b16 20.1/21.5 -> 12.6/13.4   Q8_0 b2 10.2/11.4 -> 10.4/11.4   b8 24.2/26.8 -> 10.5/11.5
  //   4096 x 11008   Q4_0 b2 17.2/21.1 -> 11.5/12.2   b8 28.9/29.9 -> 12.0/12.2   Q5_0 b8 32.7/32.9 -> 14.0/15.0   b16 22.7/26.2 -> 18.0/18.7
  //                  Q5_1 b8 33.2/33.4 -> 12.9/13.9   b16 25.8/27.0 -> 20.6/21.6   Q8_0 b8 32.1/35.4 -> 13.6/15.7
  // i.e. with fewer rows than the dot4 / streamed kernels need to fill the chip (< 8192) the 16-token tiles win from batch 2 for
  // every format; with many rows they win from where the dot4 kernel stops scaling: batch 5 (Q4_0 Q4_1 Q8_0), 9 (Q5_0), never (Q5_1).
  // Q6_K (210-byte super-blocks: unaligned row slices, an MFMA per 16-element sub-block; batch <= 16):
  //   11008 x 4096 b2 14.1/18.6 -> 14.9/16.8   b8 18.5/22.0 -> 14.4/16.8   b16 27.6/31.1 -> 20.2/23.5
  //   3584 x 8192  b8 29.3/31.6 -> 11.0/13.2   b16 29.7/34.9 -> 19.5/22.1     4096 x 11008 b8 41.0/42.7 -> 15.0/18.3
  // Q3_K (110-byte super-blocks, same scheme; its scale decode keeps the kernel at 15.7 us, so it only pays where the old kernels
  // do not fill the chip):  11008 x 4096 b8 19.0/21.1 -> 19.0/21.3   b16 21.0/22.9 -> 21.1/23.6
  //   3584 x 8192 b2 26.5/28.8 -> 12.8/15.3   b8 20.8/23.4 -> 13.3/15.4   b16 22.5/24.7 -> 16.5/19.1     4096 x 11008 b2 35.3/36.7 -> 18.5/20.9   b8 26.9/29.3 -> 18.7/21.2
  int64_t t16_from = 2, t16_to = 0;
  switch (type) {
    case GGQ_TYPE_Q4_K: case GGQ_TYPE_Q5_K: t16_to = t16_two_tiles_pay(n_rows) ? 32 : 16; break;
    // Q6_K on vocabulary-sized matrices (profiles/r03_vocab_rows.txt, kernel alone, warm): once its workgroups (3.4 KB of LDS image per unit) are no
    // longer all resident the 16-token-tile kernel collapses — 65536 x 4096 batch 8: 120 us against 68 on dot4; 128256 rows: 277 against 152;
    // batch 16 at 28672 rows: 59 against 44 streamed — while up to 14336 rows (batch 16) / 28672 rows (batch 8: 28.4 against 36.8) it wins.
    // (The other formats' instances hold up there: Q8_0 128256 x 4096 batch 8 119 us against 143, Q5_K 80 against 103.)
    case GGQ_TYPE_Q6_K: t16_to = n_rows <= 16384 ? 16 : n_rows <= 32768 ? 8 : 0; break;
    case GGQ_TYPE_Q4_0: case GGQ_TYPE_Q4_1: t16_to = 16; t16_from = n_rows < 8192 ? 2 : 5; break;   // (many rows: see many_rows_k below)
    case GGQ_TYPE_Q8_0: t16_to = 16; break;   // from batch 2 at every shape: with many rows a tie warm (13.4 against 12.9 - 13.7 us on dot4) and
                                              // 15.1 - 15.5 against 17.2 - 18.6 with the weights from HBM (profiles/r03_sweep_batch_all.txt)
    // Beyond ~12288 rows the dot4 kernel (sized for 4096 waves) stops scaling and the 5-bit formats' 16-token tiles overtake it from batch 5
    // (scripts/sweep_t16.py, profiles/r03_t16_many_rows.txt, op us warm / cold, routed before | 16-token tiles):
    //   Q5_1 batch 8   14336 x 4096 25.1 / 25.3 | 14.1 / 15.9   16384 25.6 / 26.1 | 14.5 / 16.2   20480 26.3 / 27.8 | 17.7 / 20.1   28672 36.4 / 37.4 | 22.7 / 25.3
    //                  14336 x 8192 41.4 / 41.9 | 23.8 / 25.9   20480 x 8192 43.7 / 44.3 | 38.9 / 40.3        batch 16 (streamed before) 14336 22.0 / 24.2 | 18.4 / 19.6
    //                  20480 28.9 / 30.6 | 20.9 / 22.9   28672 35.9 / 37.9 | 29.6 / 30.7   (K = 8192: level)
    //   Q5_0 batch 8   14336 24.3 / 25.5 | 17.2 / 18.3   16384 24.6 / 25.7 | 17.7 / 20.6   20480 25.4 / 26.5 | 22.3 / 22.7   28672 32.4 / 35.0 | 26.3 / 27.0   (batch 4: level)
    //   (Q4_0 / Q4_1 batch 2 - 4 there: mixed — ahead at K = 4096, far behind at 28672 x 8192; Q3_K: level at 8, behind at 16 — unchanged)
    case GGQ_TYPE_Q5_0: t16_to = 16; t16_from = n_rows < 8192 ? 2 : n_rows > 12288 ? 5 : 9; break;
    case GGQ_TYPE_Q5_1: t16_to = (n_rows < 8192 || n_rows > 12288) ? 16 : 0; t16_from = n_rows < 8192 ? 2 : 5; break;
    case GGQ_TYPE_Q3_K: t16_to = n_rows < 8192 ? 16 : 0; break;
    default: break;
  }
  // Many rows (> 12288), batch 2 - 4, with K AND the tensor's size as inputs — the round-4 regret pass over profiles/r03_t16_many_rows.txt and
  // r03_t16_vs_stream_b8_16.txt (tests/test_host_logic.py::test_route_regret_small_batches; op us warm / cold, dot4 | 16-token tiles):
  //   14336 x 4096, batch 3 - 4: all four nibble formats are ahead on the 16-token tiles — Q4_0 15.5 / 18.9 | 13.8 / 14.6, Q4_1 15.0 / 16.7 | 13.2 / 13.9,
  //   Q5_0 18.1 / 21.7 | 17.4 / 18.5, Q5_1 17.1 / 18.7 | 14.0 / 15.7; batch 2: Q4_0 14.1 / 18.4 | 13.4 / 14.6, Q5_0 16.3 / 20.4 | 16.9 / 17.9, Q4_1 / Q5_1 level.
  //   20480 x 4096: level cold, 5 - 25 % behind warm (the launch no longer fits one resident round): dot4 stays.
  //   Q4_0 28672 x 4096 (66 MB): 19.4 / 24.9 | 20.4 / 21.4 at batch 2, 21.6 / 25.9 | 20.8 / 21.6 at 4 — ahead with the weights from HBM, which is
  //   where a 66 MB tensor lives; 28672 x 8192 (132 MB): 31.6 / 43.7 | 51.2 / 52.2 — far behind, and at batch 16 there 61.9 / 60.0 against 48.5 / 54.0
  //   streamed: K = 8192 doubles every wave's request -> land -> compute chain while the rows already need several rounds.
  if (batch <= 4 && n_rows > 12288 && ggq_block_elems(type) == 32 && type != GGQ_TYPE_Q8_0) {
    const int64_t bytes = n_rows * ggq_row_bytes(type, k);
    if (n_rows <= 16384 && k <= 4096) t16_from = (type == GGQ_TYPE_Q4_0 || type == GGQ_TYPE_Q5_0) ? 2 : 3;
    else if (type == GGQ_TYPE_Q4_0 && k <= 4096 && bytes <= (96ll << 20)) t16_from = 2;
  }
  if (type == GGQ_TYPE_Q4_0 && batch > 8 && k > 4096 && n_rows * ggq_row_bytes(type, k) > (96ll << 20)) t16_to = 8;
  // one token through this entry point: with few rows the 16-token tiles beat the dot4 kernel there too (Q4_K 4096 x 11008 10.7 / 12.4 us
  // against 10.1 / 11.1 at batch 2; 3584 x 8192 9.3 / 10.0 against 8.5 / 9.1), with many they do not (11008 x 4096: 8.5 against 9.8)
  if (t16_to > 0 && t16_from == 2 && n_rows < 8192) t16_from = 1;
  const bool t16_shape_ok = (type != GGQ_TYPE_Q6_K && type != GGQ_TYPE_Q3_K) || (n_rows * ggq_row_bytes(type, k) >= 1024 && n_rows * ggq_row_bytes(type, k) < (1ll << 32));   // ggq_mul_mat_q_t16's own guards
  if (t16_shape_ok && ggq_mmq_t16_supported(type, k, batch) && batch >= t16_from && batch <= t16_to) return GGQ_MMQ_ROUTE_T16;
  // From 33 tokens: the 64 x 64 wave-tile kernel (mmq_x64.hip) wherever its launch has enough units.  Its time goes with
  // ceil(units / 512) resident rounds of (K / 256) / 4 super-blocks per wave plus a fixed 9 - 10 us (first stage from HBM, K-slice
  // reduction, write-back, the quantise launch), the streamed kernel's with its 32 x 32/64-token units per CU: measured
  // (scripts/sweep_x64.py, profiles/r04_x64_vs_stream_q4k_ks4.txt, eleven shapes 2048 x 4096 ... 28672 x 8192, batch 33 ... 1024, op us cold,
  // streamed / x64) the ratio is 0.63 - 0.88 below 130 units (the x64 launch leaves most CUs empty while every unit pays the fixed part),
  // 1.04 - 1.23 at 168 - 192 units and 1.2 - 1.45 from 256 units on at every shape and K — one threshold on the unit count:
  //   11008 x 4096: b64 24.5 / 21.1   b128 31.2 / 29.4   b256 54.3 / 44.3   b1024 193 / 151      4096 x 4096: b128 17.1 / 20.3   b192 25.0 / 20.8
  //   4096 x 11008: b128 33.7 / 43.2   b192 53.4 / 43.9      3584 x 8192: b128 26.2 / 33.3   b192 41.9 / 34.2      28672 x 8192: b128 127 / 100
  // Re-measured with the 96-row units and for all three formats of the kernel (profiles/r04b_x64_vs_stream_{q4_k,q8_0,q4_0}.txt, streamed /
  // x64, cold): from 168 units up Q4_K 1.15 - 1.45, Q8_0 1.43 - 1.6, Q4_0 1.37 - 1.55; at 96 - 128 units Q4_K 0.88 - 0.97, Q8_0 0.88 - 1.05
  // (ahead warm, level cold), Q4_0 0.98 - 1.13; below 96 units of 64 rows every format loses with 64-row units (0.68 - 0.95).
  // Batch 33 - 64 on at most 4096 rows: two 32-token one-tile units per 32 rows (ggq_mmq_x64_tile_tokens), ahead of the streamed kernel
  // and of the 64-token 32-row units from 2048 rows up for all four formats (profiles/r04b_x64_one_tile_b33_64.txt).
  if (batch >= 33 && batch <= 64 && ggq_mmq_x64_supported(type, k, batch) && ggq_mmq_x64_tile_tokens(type, batch, k, n_rows) == 32 &&
      (n_rows + 31) / 32 >= 64)
    return GGQ_MMQ_ROUTE_X64;
  // Below 160 units the kernel's 32-row units (ggq_mmq_x64_unit_rows) take over from the band where they beat the streamed kernel.
  const int64_t x64_units = ((n_rows + 63) / 64) * ((batch + 63) / 64), x32_units = ((n_rows + 31) / 32) * ((batch + 63) / 64);
  if (batch >= 33 && ggq_mmq_x64_supported(type, k, batch) &&
      (x64_units >= 160 || (x64_min_units32(type) > 0 && x32_units >= x64_min_units32(type))))
    return GGQ_MMQ_ROUTE_X64;
  // The other formats (and batch 1 through this entry point), thresholds measured at 11008 x 4096 (rounds 1-2, mmq.hip):
  // the dot4 kernel while it beats the streamed one with the weights coming from HBM, the barrier-coupled LDS-tile
  // kernel for the mid batches of the two formats whose streamed instance is bound by its weight copy, streamed beyond.
  const bool dot4_to_8 = type == GGQ_TYPE_Q4_0 || type == GGQ_TYPE_Q4_1 || type == GGQ_TYPE_Q5_0 || type == GGQ_TYPE_Q5_1 ||
                         type == GGQ_TYPE_Q4_K;
  // Q8_0 batch 17 - 64: the LDS-tile kernel wins where the matrix has many rows (11008 x 4096: 19.4 / 25.6 us at batch 32 / 64 against
  // 23.4 / 26.7 streamed) and loses by a third where it has few (3584 x 8192: 28.5 / 32.2 against 19.6 / 20.8; 4096 x 11008: 37.0 / 41.8
  // against 23.8 / 26.3) — kernel alone, warm, profiles/r03_route_audit.txt.  Q6_K 17 - 32 is a tie either way (+-2 us by shape).
  const int64_t stream_from = type == GGQ_TYPE_Q8_0 ? (n_rows < 8192 ? 17 : 65) : type == GGQ_TYPE_Q6_K ? 33 : dot4_to_8 ? 9 :
                              // (past 12288 rows the dot4 kernel stops scaling: streamed from batch 3 / 2 there — 14336 x 4096 Q2_K batch 3
                              // 17.1 -> 12.5 us, Q3_K batch 2 22.7 -> 17.7; 28672 rows 23.8 -> 21.6, 32.4 -> 29.4; profiles/r03_t16_many_rows.txt)
                              (type == GGQ_TYPE_Q2_K && (n_rows < 8192 || n_rows > 12288)) ? 3 :
                              (type == GGQ_TYPE_Q3_K && n_rows > 12288) ? 2 : 5;
  // Q2_K (two int8 tiles per group; profiles/r03_route_audit2.txt, _audit3.txt, kernel alone, warm): its streamed instance is good up to 16 tokens
  // (transposed operands: 15.4 / 17.1 / 22.3 us at batch 16 on 11008 x 4096 / 3584 x 8192 / 4096 x 11008 against 22.5 / 25.4 / 33.1 on the
  // LDS-tile kernel) and from 33 (26.0 / 31.0 / 42.3 against 37.6 / 35.3 / 46.4), but at 17 - 32 the LDS-tile kernel wins: 22.7 / 25.4 / 33.0
  // against 23.2 - 24.6 / 29.8 / 40.7.  With few rows the dot4 kernel is overtaken at batch 3 (op, 3584 x 8192: 13.7 / 18.1 us at batch 2 / 3 on dot4, 16.1 at batch 4 streamed), not 5.
  if (type == GGQ_TYPE_Q2_K && batch >= 17 && batch <= 32) return GGQ_MMQ_ROUTE_LDS_TILE;
  const bool streamable = ggq_row_bytes(type, k) <= (32 << 20);   // ggq_mmq_tiled_supported: 32-bit offsets in a 32-row tile
  if (streamable && batch >= stream_from) return GGQ_MMQ_ROUTE_STREAM;
  return batch <= 8 ? GGQ_MMQ_ROUTE_DOT4 : GGQ_MMQ_ROUTE_LDS_TILE;
}
