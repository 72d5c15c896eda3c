// Wait states around v_mfma_i32_32x32x32_i8 on gfx950, measured from inline asm with fixed physical registers (no compiler
// pass adds wait states): how many does the hardware need between
//   (a) a VALU write of a register (v_mov_b32 / v_mov_b64 / v_pk_mov_b32) and the MFMA that reads it as SrcC,
//   (b) the same for SrcB (the last operand register),
//   (c) the MFMA and a VALU read of its result (v_mov_b32 of the LAST result register),
//   (d) the MFMA and a VALU overwrite of its SrcC registers (write-after-read).
// For every spacing the kernel counts, by 16-lane quarter of the wave, the lanes that saw the OLD value.
// build: hipcc -O3 --offload-arch=gfx950 scripts/ubench_mfma_hazard.hip -o scripts/ubench_mfma_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CLOB "memory", "v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63", \
             "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", \
             "v80","v81","v82","v83","v84","v85","v86","v87"
#define SETC(V) "v_mov_b32 v48, " V "\n\tv_mov_b32 v49, " V "\n\tv_mov_b32 v50, " V "\n\tv_mov_b32 v51, " V "\n\tv_mov_b32 v52, " V "\n\tv_mov_b32 v53, " V "\n\t" \
                "v_mov_b32 v54, " V "\n\tv_mov_b32 v55, " V "\n\tv_mov_b32 v56, " V "\n\tv_mov_b32 v57, " V "\n\tv_mov_b32 v58, " V "\n\tv_mov_b32 v59, " V "\n\t" \
                "v_mov_b32 v60, " V "\n\tv_mov_b32 v61, " V "\n\tv_mov_b32 v62, " V "\n\tv_mov_b32 v63, " V "\n\t"
#define ZAB "v_mov_b32 v80, 0\n\tv_mov_b32 v81, 0\n\tv_mov_b32 v82, 0\n\tv_mov_b32 v83, 0\n\tv_mov_b32 v84, 0\n\tv_mov_b32 v85, 0\n\tv_mov_b32 v86, 0\n\tv_mov_b32 v87, 0\n\t"
#define DRAIN "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"

// MODE 0: v_mov_b32 v63 -> SrcC; 1: v_mov_b64 v[62:63] -> SrcC; 2: v_pk_mov_b32 v[62:63] -> SrcC
template <int MODE, int N>
__global__ void __launch_bounds__(256) k_valu_to_srcc(uint32_t* out, int iters) {
  uint32_t bad = 0;
  for (int it = 0; it < iters; ++it) {
    const uint32_t oldv = 0x100 + it, newv = 0x7000 + it;
    uint32_t r;
#define WR0 "v_mov_b32 v63, %2\n\t"
#define WR1 "v_mov_b32 v86, %2\n\tv_mov_b32 v87, %2\n\ts_nop 7\n\tv_mov_b64 v[62:63], v[86:87]\n\t"
#define WR2 "v_mov_b32 v86, %2\n\tv_mov_b32 v87, %2\n\ts_nop 7\n\tv_pk_mov_b32 v[62:63], v[86:87], v[86:87]\n\t"
#define TAIL "v_mfma_i32_32x32x32_i8 v[64:79], v[80:83], v[80:83], v[48:63]\n\t" DRAIN "v_mov_b32 %0, v79"
#define RUN(WR, NOPS) asm volatile(SETC("%1") ZAB DRAIN WR NOPS TAIL : "=v"(r) : "v"(oldv), "v"(newv) : CLOB)
#define RUNN(WR)                                                                 \
    if constexpr (N == 0) RUN(WR, "");                                           \
    if constexpr (N == 1) RUN(WR, "s_nop 0\n\t");                                \
    if constexpr (N == 2) RUN(WR, "s_nop 1\n\t");                                \
    if constexpr (N == 3) RUN(WR, "s_nop 2\n\t");                                \
    if constexpr (N == 4) RUN(WR, "s_nop 3\n\t");                                \
    if constexpr (N == 6) RUN(WR, "s_nop 5\n\t");
    if constexpr (MODE == 0) { RUNN(WR0) }
    if constexpr (MODE == 1) { RUNN(WR1) }
    if constexpr (MODE == 2) { RUNN(WR2) }
    bad += r != newv;
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = bad;
}

// (c) MFMA -> VALU read of the last result register after N wait states; SrcC = NEW everywhere, result regs preset OLD
template <int N>
__global__ void __launch_bounds__(256) k_mfma_to_valu(uint32_t* out, int iters) {
  uint32_t bad = 0;
  for (int it = 0; it < iters; ++it) {
    const uint32_t oldv = 0x100 + it, newv = 0x7000 + it;
    uint32_t r;
#define PRE "v_mov_b32 v79, %1\n\tv_mov_b32 v64, %1\n\t"
#define RUNC(NOPS) asm volatile(SETC("%2") ZAB PRE DRAIN "v_mfma_i32_32x32x32_i8 v[64:79], v[80:83], v[80:83], v[48:63]\n\t" NOPS "v_mov_b32 %0, v79\n\t" DRAIN \
                                : "=v"(r) : "v"(oldv), "v"(newv) : CLOB)
    if constexpr (N == 4) RUNC("s_nop 3\n\t");
    if constexpr (N == 8) RUNC("s_nop 7\n\t");
    if constexpr (N == 9) RUNC("s_nop 8\n\t");
    if constexpr (N == 10) RUNC("s_nop 9\n\t");
    if constexpr (N == 11) RUNC("s_nop 10\n\t");
    if constexpr (N == 12) RUNC("s_nop 11\n\t");
    if constexpr (N == 14) RUNC("s_nop 13\n\t");
    if constexpr (N == 18) RUNC("s_nop 15\n\ts_nop 1\n\t");
    bad += r != newv;
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = bad;
}

// (d) MFMA reads SrcC = NEW; N wait states later a VALU overwrites SrcC's last register with OLD: did the MFMA still see NEW?
template <int N>
__global__ void __launch_bounds__(256) k_srcc_war(uint32_t* out, int iters) {
  uint32_t bad = 0;
  for (int it = 0; it < iters; ++it) {
    const uint32_t oldv = 0x100 + it, newv = 0x7000 + it;
    uint32_t r;
#define RUND(NOPS) asm volatile(SETC("%2") ZAB DRAIN "v_mfma_i32_32x32x32_i8 v[64:79], v[80:83], v[80:83], v[48:63]\n\t" NOPS "v_mov_b32 v63, %1\n\t" DRAIN "v_mov_b32 %0, v79" \
                                : "=v"(r) : "v"(oldv), "v"(newv) : CLOB)
    if constexpr (N == 0) RUND("");
    if constexpr (N == 1) RUND("s_nop 0\n\t");
    if constexpr (N == 2) RUND("s_nop 1\n\t");
    if constexpr (N == 4) RUND("s_nop 3\n\t");
    if constexpr (N == 7) RUND("s_nop 6\n\t");
    if constexpr (N == 11) RUND("s_nop 10\n\t");
    bad += r != newv;
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = bad;
}

template <typename F> static void report(const char* name, F launch, uint32_t* dbuf, size_t nthreads) {
  (void)hipMemset(dbuf, 0, nthreads * 4);
  launch();
  (void)hipDeviceSynchronize();
  std::vector<uint32_t> h(nthreads);
  (void)hipMemcpy(h.data(), dbuf, h.size() * 4, hipMemcpyDeviceToHost);
  unsigned long long q[4] = {0, 0, 0, 0};
  for (size_t t = 0; t < nthreads; ++t) q[(t & 63) >> 4] += h[t];
  printf("%-58s stale lanes by quarter: %9llu %9llu %9llu %9llu\n", name, q[0], q[1], q[2], q[3]);
}

int main() {
  const int blocks = 256 * 8, threads = 256, iters = 500;
  const size_t nthreads = (size_t)blocks * threads;
  uint32_t* dbuf;
  (void)hipMalloc(&dbuf, nthreads * 4);
#define A(MODE, N, NAME) report(NAME " -> MFMA SrcC, " #N " wait state(s)", [&] { hipLaunchKernelGGL((k_valu_to_srcc<MODE, N>), dim3(blocks), dim3(threads), 0, 0, dbuf, iters); }, dbuf, nthreads)
  A(0, 0, "v_mov_b32"); A(0, 1, "v_mov_b32"); A(0, 2, "v_mov_b32"); A(0, 3, "v_mov_b32"); A(0, 4, "v_mov_b32"); A(0, 6, "v_mov_b32");
  A(1, 0, "v_mov_b64"); A(1, 1, "v_mov_b64"); A(1, 2, "v_mov_b64"); A(1, 3, "v_mov_b64"); A(1, 4, "v_mov_b64"); A(1, 6, "v_mov_b64");
  A(2, 0, "v_pk_mov_b32"); A(2, 1, "v_pk_mov_b32"); A(2, 2, "v_pk_mov_b32"); A(2, 3, "v_pk_mov_b32"); A(2, 4, "v_pk_mov_b32"); A(2, 6, "v_pk_mov_b32");
#define C(N) report("MFMA -> v_mov_b32 of its last result register, " #N " wait states", [&] { hipLaunchKernelGGL((k_mfma_to_valu<N>), dim3(blocks), dim3(threads), 0, 0, dbuf, iters); }, dbuf, nthreads)
  C(4); C(8); C(9); C(10); C(11); C(12); C(14); C(18);
#define D(N) report("MFMA (reads SrcC) -> VALU overwrite of SrcC, " #N " wait states", [&] { hipLaunchKernelGGL((k_srcc_war<N>), dim3(blocks), dim3(threads), 0, 0, dbuf, iters); }, dbuf, nthreads)
  D(0); D(1); D(2); D(4); D(7); D(11);
  return 0;
}
