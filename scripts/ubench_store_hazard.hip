// Does a vector store's data read race with a VALU write of the same registers issued right behind it?
// LLVM's hazard recognizer (GCNHazardRecognizer: createsVALUHazard / checkVALUHazardsHelper) puts wait states between a
// vector-memory store and a VALU write of its data registers only when the store data is WIDER than 64 bits.  This
// micro-benchmark issues  store(width) ; [s_nop n] ; VALU overwrite of the first data register  from inline asm (fixed
// physical registers, so that no compiler pass adds wait states or moves) for global and scratch stores of 1, 2 and 4
// dwords, from every CU at once, and counts the stored words that came out as the OVERWRITING value, by 16-lane quarter.
// build: hipcc -O3 --offload-arch=gfx950 scripts/ubench_store_hazard.hip -o scripts/ubench_store_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define JUNK 0xDEAD0000u
#define SET4(G) "v_mov_b32 v40, " G "\n\tv_mov_b32 v41, " G "\n\tv_mov_b32 v42, " G "\n\tv_mov_b32 v43, " G "\n\ts_nop 7\n\t"
#define CLOB "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47"

template <int N> struct Nop;
template <> struct Nop<0> { static constexpr const char* s = ""; };

// W = dwords per store, N = wait states between the store and the overwrite
#define BODY_G(STORE, REGS, NOPS)                                                                                   \
  asm volatile(SET4("%1") STORE " %0, " REGS ", off\n\t" NOPS "v_mov_b32 v40, %2\n\ts_waitcnt vmcnt(0)"                  \
               :: "v"(dst), "v"(good), "v"(j) : CLOB)
#define BODY_S(STORE, LOAD, REGS, RREGS, NOPS)                                                                      \
  asm volatile(SET4("%2") STORE " %1, " REGS ", off\n\t" NOPS "v_mov_b32 v40, %3\n\ts_waitcnt vmcnt(0)\n\t"               \
               LOAD " " RREGS ", %1, off\n\ts_waitcnt vmcnt(0)\n\tv_mov_b32 %0, v44"                                \
               : "=v"(r0) : "v"(off), "v"(good), "v"(j) : CLOB)

template <int W, int N>
__global__ void __launch_bounds__(256) kg(uint32_t* out, int iters) {
  uint32_t* dst = out + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  uint32_t nbad = 0;
  for (int it = 0; it < iters; ++it) {
    const uint32_t good = 0x1000u + it, j = JUNK + it;
    if constexpr (W == 1 && N == 0) BODY_G("global_store_dword", "v40", "");
    if constexpr (W == 1 && N == 1) BODY_G("global_store_dword", "v40", "s_nop 0\n\t");
    if constexpr (W == 1 && N == 2) BODY_G("global_store_dword", "v40", "s_nop 1\n\t");
    if constexpr (W == 2 && N == 0) BODY_G("global_store_dwordx2", "v[40:41]", "");
    if constexpr (W == 2 && N == 1) BODY_G("global_store_dwordx2", "v[40:41]", "s_nop 0\n\t");
    if constexpr (W == 2 && N == 2) BODY_G("global_store_dwordx2", "v[40:41]", "s_nop 1\n\t");
    if constexpr (W == 4 && N == 0) BODY_G("global_store_dwordx4", "v[40:43]", "");
    if constexpr (W == 4 && N == 1) BODY_G("global_store_dwordx4", "v[40:43]", "s_nop 0\n\t");
    if constexpr (W == 4 && N == 2) BODY_G("global_store_dwordx4", "v[40:43]", "s_nop 1\n\t");
    const uint32_t r = __builtin_nontemporal_load(dst);   // word 0 = the register the VALU overwrote
    nbad += r != good;
  }
  __builtin_nontemporal_store(nbad, dst);
}

template <int W, int N>
__global__ void __launch_bounds__(256) ks(uint32_t* out, int iters, int idx) {
  // a private array of the kernel's own: makes hipcc allocate scratch for the wave (the asm below addresses its first 64 bytes)
  volatile uint32_t keep[32];
  for (int i = 0; i < 32; ++i) keep[i] = (uint32_t)(i + idx);
  uint32_t nbad = keep[(idx + 5) & 31] == 0xFFFFFFFFu ? 1u : 0u;
  const uint32_t off = (uint32_t)(idx & 3) * 16;   // byte offset inside this lane's scratch
  for (int it = 0; it < iters; ++it) {
    const uint32_t good = 0x1000u + it, j = JUNK + it;
    uint32_t r0 = 0;
    if constexpr (W == 1 && N == 0) BODY_S("scratch_store_dword", "scratch_load_dword", "v40", "v44", "");
    if constexpr (W == 1 && N == 1) BODY_S("scratch_store_dword", "scratch_load_dword", "v40", "v44", "s_nop 0\n\t");
    if constexpr (W == 1 && N == 2) BODY_S("scratch_store_dword", "scratch_load_dword", "v40", "v44", "s_nop 1\n\t");
    if constexpr (W == 2 && N == 0) BODY_S("scratch_store_dwordx2", "scratch_load_dwordx2", "v[40:41]", "v[44:45]", "");
    if constexpr (W == 2 && N == 1) BODY_S("scratch_store_dwordx2", "scratch_load_dwordx2", "v[40:41]", "v[44:45]", "s_nop 0\n\t");
    if constexpr (W == 2 && N == 2) BODY_S("scratch_store_dwordx2", "scratch_load_dwordx2", "v[40:41]", "v[44:45]", "s_nop 1\n\t");
    if constexpr (W == 4 && N == 0) BODY_S("scratch_store_dwordx4", "scratch_load_dwordx4", "v[40:43]", "v[44:47]", "");
    if constexpr (W == 4 && N == 1) BODY_S("scratch_store_dwordx4", "scratch_load_dwordx4", "v[40:43]", "v[44:47]", "s_nop 0\n\t");
    if constexpr (W == 4 && N == 2) BODY_S("scratch_store_dwordx4", "scratch_load_dwordx4", "v[40:43]", "v[44:47]", "s_nop 1\n\t");
    nbad += r0 != good;
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = nbad;
}

template <typename F> static void report(const char* name, F launch, uint32_t* dbuf, size_t nthreads, int stride) {
  (void)hipMemset(dbuf, 0, nthreads * stride * 4);
  launch();
  (void)hipDeviceSynchronize();
  std::vector<uint32_t> h(nthreads * stride);
  (void)hipMemcpy(h.data(), dbuf, h.size() * 4, hipMemcpyDeviceToHost);
  unsigned long long q[4] = {0, 0, 0, 0};
  for (size_t t = 0; t < nthreads; ++t) q[(t & 63) >> 4] += h[t * stride];
  printf("%-46s overwritten-value words by lane quarter: %8llu %8llu %8llu %8llu\n", name, q[0], q[1], q[2], q[3]);
}

int main() {
  const int blocks = 256 * 8, threads = 256, iters = 2000;
  const size_t nthreads = (size_t)blocks * threads;
  uint32_t* dbuf;
  (void)hipMalloc(&dbuf, nthreads * 4 * 4);
#define G(W, N) report("global_store " #W " dword(s), " #N " wait state(s)", [&] { hipLaunchKernelGGL((kg<W, N>), dim3(blocks), dim3(threads), 0, 0, dbuf, iters); }, dbuf, nthreads, 4)
  G(1, 0); G(1, 1); G(1, 2); G(2, 0); G(2, 1); G(2, 2); G(4, 0); G(4, 1); G(4, 2);
#define S(W, N) report("scratch_store " #W " dword(s), " #N " wait state(s)", [&] { hipLaunchKernelGGL((ks<W, N>), dim3(blocks), dim3(threads), 0, 0, dbuf, iters, 1); }, dbuf, nthreads, 1)
  S(1, 0); S(1, 1); S(1, 2); S(2, 0); S(2, 1); S(2, 2); S(4, 0); S(4, 1); S(4, 2);
  return 0;
}
