// Do vector-memory operations of different kinds retire from vmcnt IN ISSUE ORDER on gfx950?  hipcc's wait insertion
// assumes so (on targets without a separate store counter loads and stores are one in-order event class), so a spill
// store (scratch_store) issued behind a load is covered by `s_waitcnt vmcnt(1)` when the load's result is needed.
// This micro-benchmark issues  slow load (HBM miss) ; fast younger operation ; s_waitcnt vmcnt(1) ; read the load's
// register  from inline asm and counts the lanes that still hold the sentinel, by 16-lane quarter.
// build: hipcc -O3 --offload-arch=gfx950 scripts/ubench_vmcnt_order.hip -o scripts/ubench_vmcnt_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define SENT 0xBADC0DEu

// YOUNG 0: scratch_store_dword, 1: scratch_store_dwordx2, 2: global_store_dword, 3: scratch_load_dword, 4: L2-hit global_load_dword
// OLD   0: global_load_dword, 1: buffer-style global_load_dwordx3 (12 bytes), 2: global_load_dwordx4
template <int OLD, int YOUNG>
__global__ void __launch_bounds__(256) k(const uint32_t* __restrict__ big, size_t nwords, uint32_t* out, uint32_t* sink, int iters) {
  volatile uint32_t keep[32];
  for (int i = 0; i < 32; ++i) keep[i] = (uint32_t)i;
  uint32_t bad = keep[3] == 77u;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t* mysink = sink + tid * 4;
  for (int it = 0; it < iters; ++it) {
    // a far-apart, never-reused address: every lane its own 128-byte line (slow), value = its own word index
    const size_t idx = ((tid * 2654435761ull + (size_t)it * 40503ull * 64) % (nwords / 32)) * 32;
    const uint32_t* p = big + idx;
    const uint32_t expect = (uint32_t)idx;
    uint32_t r, zero = 0, hot = 0;
    const uint32_t* hotp = big + (threadIdx.x & 63);   // L2/L1-resident line for the fast younger load
#define OLD0 "global_load_dword v40, %1, off\n\t"
#define OLD1 "global_load_dwordx3 v[40:42], %1, off\n\t"
#define OLD2 "global_load_dwordx4 v[40:43], %1, off\n\t"
#define Y0 "scratch_store_dword %3, %2, off\n\t"
#define Y1 "scratch_store_dwordx2 %3, v[44:45], off\n\t"
#define Y2 "global_store_dword %4, %2, off\n\t"
#define Y3 "scratch_load_dword v46, %3, off\n\t"
#define Y4 "global_load_dword v46, %5, off\n\t"
#define RUN(O, Y) asm volatile("v_mov_b32 v40, %6\n\tv_mov_b32 v44, 0\n\tv_mov_b32 v45, 0\n\ts_nop 4\n\t" O Y "s_waitcnt vmcnt(1)\n\tv_mov_b32 %0, v40\n\ts_waitcnt vmcnt(0)" \
                               : "=v"(r) : "v"(p), "v"(zero), "v"(hot), "v"(mysink), "v"(hotp), "v"(SENT) : "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46")
#define PICK(O)                                   \
    if constexpr (YOUNG == 0) RUN(O, Y0);         \
    if constexpr (YOUNG == 1) RUN(O, Y1);         \
    if constexpr (YOUNG == 2) RUN(O, Y2);         \
    if constexpr (YOUNG == 3) RUN(O, Y3);         \
    if constexpr (YOUNG == 4) RUN(O, Y4);
    if constexpr (OLD == 0) { PICK(OLD0) }
    if constexpr (OLD == 1) { PICK(OLD1) }
    if constexpr (OLD == 2) { PICK(OLD2) }
    bad += r != expect;
  }
  out[tid] = bad;
}

template <typename F> static void report(const char* name, F launch, uint32_t* dbuf, size_t nthreads) {
  (void)hipMemset(dbuf, 0, nthreads * 4);
  launch();
  (void)hipDeviceSynchronize();
  std::vector<uint32_t> h(nthreads);
  (void)hipMemcpy(h.data(), dbuf, h.size() * 4, hipMemcpyDeviceToHost);
  unsigned long long q[4] = {0, 0, 0, 0};
  for (size_t t = 0; t < nthreads; ++t) q[(t & 63) >> 4] += h[t];
  printf("%-70s lanes read before the load landed, by quarter: %8llu %8llu %8llu %8llu\n", name, q[0], q[1], q[2], q[3]);
}

int main() {
  const int blocks = 256 * 4, threads = 256, iters = 200;
  const size_t nthreads = (size_t)blocks * threads;
  const size_t nwords = (size_t)1 << 28;   // 1 GiB: far beyond the caches
  uint32_t *big, *dbuf, *sink;
  (void)hipMalloc(&big, nwords * 4);
  (void)hipMalloc(&dbuf, nthreads * 4);
  (void)hipMalloc(&sink, nthreads * 16);
  std::vector<uint32_t> h(nwords);
  for (size_t i = 0; i < nwords; ++i) h[i] = (uint32_t)i;
  (void)hipMemcpy(big, h.data(), nwords * 4, hipMemcpyHostToDevice);
#define T(O, Y, NAME) report(NAME, [&] { hipLaunchKernelGGL((k<O, Y>), dim3(blocks), dim3(threads), 0, 0, big, nwords, dbuf, sink, iters); }, dbuf, nthreads)
  T(0, 0, "global_load_dword  ; scratch_store_dword   ; vmcnt(1)");
  T(0, 1, "global_load_dword  ; scratch_store_dwordx2 ; vmcnt(1)");
  T(0, 2, "global_load_dword  ; global_store_dword    ; vmcnt(1)");
  T(0, 3, "global_load_dword  ; scratch_load_dword    ; vmcnt(1)");
  T(0, 4, "global_load_dword  ; global_load_dword (hit); vmcnt(1)");
  T(1, 0, "global_load_dwordx3; scratch_store_dword   ; vmcnt(1)");
  T(1, 3, "global_load_dwordx3; scratch_load_dword    ; vmcnt(1)");
  T(2, 0, "global_load_dwordx4; scratch_store_dword   ; vmcnt(1)");
  T(2, 3, "global_load_dwordx4; scratch_load_dword    ; vmcnt(1)");
  return 0;
}
