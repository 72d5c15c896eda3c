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
