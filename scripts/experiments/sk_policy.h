// sk_policy.h — host-side geometry of the persistent stream-K quantised GEMM (csrc/hip/mmq_sk.hip):
// which (type, batch, k) it serves, the layout of its activation scratch and the size of its
// fix-up workspace.  Plain C++ (no HIP): shared by traits.cpp (scratch sizing, no GPU needed),
// quantize.hip (who writes the scratch) and mmq_sk.hip (who reads it), so the three cannot disagree.
//
// Scratch of a stream-K call (all inside the caller's one scratch buffer, ggq_mmq_scratch_bytes):
//   [ activation tiles | workgroup flags | partial-sum slots ]
//   * tile (pair p of 64 K-elements, token tile tt of 32 tokens), TILE_BYTES each, index p * n_tt + tt:
//       int8 qs[2 groups][64 lanes][16]   lane = 32 * K-half + token: one int8 MFMA operand fragment is 1 KB
//       float sc[64 lanes][3]             { d8 of group 2p, d8 of group 2p+1, s8 of group 2p + K-half }
//     (same Q8_1 values as HK/ggml/mmq.cu:109-154 — d8 / s8 are the fp16-rounded values for the
//      need_sum formats, the fp32 d8 otherwise — stored as fp32 so the kernel converts nothing)
//   * flags: one int per workgroup, zeroed by the quantiser, set by a workgroup that publishes a partial
//     sum and cleared again by the workgroup that consumes it
//   * slots: MAX_WG x SLOT_FLOATS fp32 partial sums (one 32-row x 128-token unit each)
#pragma once
#include <stdint.h>

namespace ggq {
namespace sk {

constexpr int TILE_BYTES = 2816;          // 2 * 1024 + 64 * 12
constexpr int FLAG_BYTES = 4096;          // up to 1024 workgroup flags
constexpr int MAX_WG = 768;               // 256 CUs x 3 resident workgroups
constexpr int SLOT_FLOATS = 4 * 16 * 64;  // one unit of 32 rows x 128 tokens (the widest unit)
constexpr int MIN_BATCH = 17;             // below: the transposed small-batch kernels of mmq.hip

constexpr int64_t padded_k(int64_t k) { return k - k % 512 + 512; }   // ggml_mul_mat_a8's padding, HK/ggml/mmq.cu:190-191

// (type ids as in include/ggq.h; Q4_K = 12, Q5_K = 13)
constexpr bool type_served(int type) { return false && (type == 12 || type == 13); }   // experiment parked: see DESIGN §5.5
constexpr bool applicable(int type, int64_t batch, int64_t k) {
  return type_served(type) && batch >= MIN_BATCH && k > 0 && k % 256 == 0;
}
constexpr int64_t tiles_bytes(int64_t batch, int64_t k) {
  return (((batch + 31) / 32) * (padded_k(k) / 64) * TILE_BYTES + 255) / 256 * 256;
}
constexpr int64_t scratch_bytes(int64_t batch, int64_t k) {
  return tiles_bytes(batch, k) + FLAG_BYTES + (int64_t)MAX_WG * SLOT_FLOATS * 4;
}

}  // namespace sk
}  // namespace ggq
