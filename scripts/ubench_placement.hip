// ubench_placement.hip — where does the dispatcher put the workgroups of a one-round launch?  (Decides whether a launch of
// 768 four-wave workgroups, 3 resident per CU, can rely on "ids b, b + 256, b + 512 share a CU"-style placement.)
// Every workgroup records XCC_ID / HW_ID and its start time, then spins ~15 us so that all of them are resident together.
// build: hipcc -O2 --offload-arch=gfx950 -o scripts/_build/ubench_placement scripts/ubench_placement.hip
// usage: ubench_placement [n_workgroups=768] [lds_bytes=53248] [threads=256]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>

__global__ void __launch_bounds__(256) probe(unsigned* out, unsigned long long* t, int spin) {
  extern __shared__ char lds[];
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg(4 | (31 << 11));
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg(20 | (31 << 11));
    t[blockIdx.x] = wall_clock64();
    lds[0] = 1;
  }
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 768, ldsb = argc > 2 ? atoi(argv[2]) : 53248, th = argc > 3 ? atoi(argv[3]) : 256;
  unsigned* d; unsigned long long* dt;
  hipMalloc(&d, n * 8); hipMalloc(&dt, n * 8);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  for (int rep = 0; rep < 2; ++rep) { probe<<<n, th, ldsb>>>(d, dt, 1500); hipDeviceSynchronize(); }
  std::vector<unsigned> h(2 * n); std::vector<unsigned long long> ht(n);
  hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), dt, n * 8, hipMemcpyDeviceToHost);
  unsigned long long t0 = *std::min_element(ht.begin(), ht.end());
  std::map<unsigned, std::vector<int>> per_cu;
  for (int b = 0; b < n; ++b) {
    const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 15;
    const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
  }
  std::map<int, int> hist;
  for (auto& kv : per_cu) hist[(int)kv.second.size()]++;
  printf("%d workgroups of %d threads, %d B LDS: %zu distinct CUs;", n, th, ldsb, per_cu.size());
  for (auto& kv : hist) printf("  %d CUs host %d", kv.second, kv.first);
  printf("\nlatest start %.2f us after the first\n", (double)(*std::max_element(ht.begin(), ht.end()) - t0) / 100.0);
  int shown = 0;
  for (auto& kv : per_cu) {
    if (shown++ >= 24) break;
    printf("xcc %u se %u sh %u cu %2u:", kv.first >> 12, (kv.first >> 8) & 15, (kv.first >> 4) & 15, kv.first & 15);
    for (int b : kv.second) printf(" %4d(+%.2f)", b, (double)(ht[b] - t0) / 100.0);
    printf("\n");
  }
  // is the set of ids on a CU an arithmetic progression with a fixed stride?
  std::map<int, int> strides;
  for (auto& kv : per_cu) for (size_t i = 1; i < kv.second.size(); ++i) strides[kv.second[i] - kv.second[i - 1]]++;
  printf("id strides between neighbours on one CU:");
  for (auto& kv : strides) printf(" %d x%d", kv.first, kv.second);
  printf("\n");
  return 0;
}
