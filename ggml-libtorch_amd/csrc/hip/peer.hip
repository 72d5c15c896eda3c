// peer.hip — peer-mapped output slabs for the row-sharded matmul (no reference counterpart: SURVEY §8e, the reference has
// no multi-device code).  One process per GPU; a rank exports the gather buffer it allocated, the others map it
// (hipIpc*, dmabuf IPC: HSA_ENABLE_IPC_MODE_LEGACY=0), and every rank then WRITES its [batch, rows] slab straight into slot
// `rank` of each peer's buffer — device-to-device stores over xGMI, no RCCL collective and no staging copy.
#include "ggq_common.h"
#include <cstring>

static_assert(sizeof(hipIpcMemHandle_t) == 64, "the C ABI carries the IPC handle as 64 opaque bytes");

extern "C" int ggq_peer_export(const void* dev_ptr, void* handle_out, int64_t* offset_out) {
  if (!dev_ptr || !handle_out || !offset_out) return GGQ_ERR_ARG;
  // the handle names the ALLOCATION a pointer lies in (a caching allocator hands out interior pointers)
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)dev_ptr) != hipSuccess) return GGQ_ERR_ARG;
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, base) != hipSuccess) return GGQ_ERR_LAUNCH;
  std::memcpy(handle_out, &h, sizeof(h));
  *offset_out = (int64_t)((const char*)dev_ptr - (const char*)base);
  return GGQ_OK;
}

extern "C" int ggq_peer_import(const void* handle, int64_t offset, void** dev_ptr_out) {
  if (!handle || !dev_ptr_out || offset < 0) return GGQ_ERR_ARG;
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle, sizeof(h));
  void* base = nullptr;
  if (hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) return GGQ_ERR_LAUNCH;
  *dev_ptr_out = (char*)base + offset;
  return GGQ_OK;
}

extern "C" int ggq_peer_close(void* dev_ptr, int64_t offset) {
  if (!dev_ptr) return GGQ_ERR_ARG;
  return hipIpcCloseMemHandle((char*)dev_ptr - offset) == hipSuccess ? GGQ_OK : GGQ_ERR_LAUNCH;
}

// rows x row_bytes bytes, source and destination row pitches in bytes (a slab of a [batch, n_rows] matrix)
extern "C" int ggq_peer_write_2d(void* dst, int64_t dst_pitch, const void* src, int64_t src_pitch, int64_t row_bytes,
                                 int64_t rows, void* stream) {
  if (rows < 0 || row_bytes < 0 || dst_pitch < row_bytes || src_pitch < row_bytes) return GGQ_ERR_ARG;
  if (rows == 0 || row_bytes == 0) return GGQ_OK;
  if (!dst || !src) return GGQ_ERR_ARG;
  return hipMemcpy2DAsync(dst, (size_t)dst_pitch, src, (size_t)src_pitch, (size_t)row_bytes, (size_t)rows,
                          hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? GGQ_OK : GGQ_ERR_LAUNCH;
}

// ---------------------------------------------------------------------------------------------------------------------
// Device-side hand-off (no host barrier, no stream drain; NOT replayable from a captured graph: `generation`, the buffer
// parity and the dst / flag pointers are host-computed kernel arguments, so a replay would wait for a generation that has
// already been published and return at once — enqueue the pair eagerly, once per gather):
//   ggq_peer_scatter  one kernel: copies the rank's [rows x row_bytes] slab into slot `rank` of every peer's buffer with
//                     plain 16-byte stores, then — every storing wave drained (s_waitcnt vmcnt(0)), workgroup barrier, one
//                     lane's SYSTEM-scope release fence — counts the workgroup in; the workgroup that arrives last
//                     publishes the generation number into each peer's flag word for this rank (one 4-byte store each,
//                     system scope).  Nothing is written after the flag.
//   ggq_peer_wait     one tiny kernel on the consumer's stream: lane p polls the rank's own flag word for peer p (relaxed
//                     system-scope loads, s_sleep between polls) until it holds the generation, then one system-scope
//                     acquire fence; kernels enqueued behind it on the stream see every peer's slab.  The spin is bounded
//                     (about two seconds of wall clock): a peer that never arrives raises the status word instead of
//                     hanging the GPU.
// Protocol after MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility" (producer: drain,
// barrier, release, flag; consumer: one relaxed poll, one acquire), at system instead of agent scope because the
// destination is another device's memory.
// ---------------------------------------------------------------------------------------------------------------------
namespace ggq {

constexpr int PEER_MAX = 8;
struct PeerDsts {
  void* dst[PEER_MAX];          // slot `rank` of peer p's gather buffer
  uint32_t* flag[PEER_MAX];     // peer p's flag word for this rank
};

__global__ void __launch_bounds__(256) peer_scatter_kernel(const uint8_t* __restrict__ src, int64_t src_pitch, PeerDsts P, int n_dst,
                                                           int64_t dst_pitch, int64_t row_bytes, int64_t rows, uint32_t generation,
                                                           uint32_t* __restrict__ arrivals) {
  // 16-byte chunks of the slab, grid-stride; the tail of a row that is not a multiple of 16 bytes goes byte by byte
  const int64_t cpr = row_bytes / 16, tail = row_bytes - cpr * 16;
  const int64_t n_chunks = rows * cpr;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = c / cpr, o = (c - r * cpr) * 16;
    const uint4 v = *(const uint4*)(src + r * src_pitch + o);
    for (int d = 0; d < n_dst; ++d) *(uint4*)((uint8_t*)P.dst[d] + r * dst_pitch + o) = v;
  }
  if (tail) {
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < rows * tail; c += (int64_t)gridDim.x * blockDim.x) {
      const int64_t r = c / tail, o = cpr * 16 + (c - r * tail);
      const uint8_t v = src[r * src_pitch + o];
      for (int d = 0; d < n_dst; ++d) ((uint8_t*)P.dst[d])[r * dst_pitch + o] = v;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its own stores
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");     // system scope: the workgroup's stores are visible to other devices
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the compiler may drop the wait behind the write-back: MI355X_MICROARCH.md)
    const uint32_t before = __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (before == gridDim.x - 1) {                     // last workgroup: every workgroup's release precedes its arrival
      __hip_atomic_store(arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call on this stream
      for (int d = 0; d < n_dst; ++d) __hip_atomic_store(P.flag[d], generation, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ void __launch_bounds__(64) peer_wait_kernel(const uint32_t* __restrict__ flags, int n_src, uint32_t generation,
                                                       uint32_t* __restrict__ status) {
  const int p = threadIdx.x;
  bool ok = true;
  if (p < n_src) {
    const uint64_t t0 = wall_clock64();   // 100 MHz
    while ((int32_t)(__hip_atomic_load(flags + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - generation) < 0) {
      __builtin_amdgcn_s_sleep(8);
      if (wall_clock64() - t0 > 200000000ull) { ok = false; break; }   // ~2 s: give up instead of hanging the device
    }
  }
  if (!ok) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // system scope: later kernels on this stream read the peers' slabs
}

}  // namespace ggq

extern "C" int ggq_peer_scatter(const void* src, int64_t src_pitch, void* const* dsts, void* const* flags, int n_dst,
                                int64_t dst_pitch, int64_t row_bytes, int64_t rows, uint32_t generation, void* arrivals,
                                void* stream) {
  using namespace ggq;
  if (n_dst < 0 || n_dst > PEER_MAX || rows < 0 || row_bytes < 0 || dst_pitch < row_bytes || src_pitch < row_bytes) return GGQ_ERR_ARG;
  if (n_dst == 0) return GGQ_OK;
  if (!src || !dsts || !flags || !arrivals) return GGQ_ERR_ARG;
  if (((uintptr_t)src | (uintptr_t)src_pitch | (uintptr_t)dst_pitch) & 15) return GGQ_ERR_ALIGN;
  PeerDsts P{};
  for (int d = 0; d < n_dst; ++d) {
    if (!dsts[d] || !flags[d]) return GGQ_ERR_ARG;
    if (((uintptr_t)dsts[d] & 15) || ((uintptr_t)flags[d] & 3)) return GGQ_ERR_ALIGN;
    P.dst[d] = dsts[d];
    P.flag[d] = (uint32_t*)flags[d];
  }
  const int64_t chunks = rows * (row_bytes / 16) + 1;
  int grid = (int)((chunks + 255) / 256);
  if (grid > 64) grid = 64;   // a slab is at most a few MB: 64 workgroups keep 7 xGMI links busy and the arrival count cheap
  if (grid < 1) grid = 1;
  GGQ_HIP_PRE_LAUNCH();
  hipLaunchKernelGGL(peer_scatter_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src, src_pitch, P, n_dst,
                     dst_pitch, row_bytes, rows, generation, (uint32_t*)arrivals);
  GGQ_HIP_CHECK_LAUNCH();
  return GGQ_OK;
}

extern "C" int ggq_peer_wait(const void* flags, int n_src, uint32_t generation, void* status, void* stream) {
  using namespace ggq;
  if (n_src < 0 || n_src > 64) return GGQ_ERR_ARG;
  if (n_src == 0) return GGQ_OK;
  if (!flags || !status) return GGQ_ERR_ARG;
  GGQ_HIP_PRE_LAUNCH();
  hipLaunchKernelGGL(peer_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const uint32_t*)flags, n_src, generation,
                     (uint32_t*)status);
  GGQ_HIP_CHECK_LAUNCH();
  return GGQ_OK;
}
