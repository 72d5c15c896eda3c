"""Row-sharded (output-feature-parallel) quantised matmul across the GPUs of one node.

The reference has no multi-device code at all (zero NCCL/MPI call sites);
this is the north_star's "weight columns shard across the GPUs with an RCCL all-gather of
partial outputs": rank r owns the weight rows [r·N/P, (r+1)·N/P) — any row split is
format-safe because a row is a whole number of blocks — computes its [batch, N/P] slab with
the single-GPU kernels, and the slabs are all-gathered.  One process per GPU,
``torch.distributed`` backend "nccl" (= RCCL over xGMI); no collective touches the weights.

The partition / gather logic is plain torch.distributed code and runs on CPU tensors with
the gloo backend too (that is how tests/test_dist_cpu.py covers it without GPUs).
"""
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_rows(n_rows: int, world_size: int, rank: int) -> Tuple[int, int]:
    """[start, end) of the weight rows owned by `rank`: an even split, the first
    n_rows % world_size ranks take one extra row."""
    base, rem = divmod(n_rows, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def gather_slabs(y_local: torch.Tensor, n_rows: int, group=None, out: Optional[torch.Tensor] = None,
                 async_op: bool = False):
    """All-gather the per-rank [batch, rows_r] slabs into [batch, n_rows].

    Equal shards use one all_gather_into_tensor into a [P, batch, N/P] buffer that is
    returned as [batch, N]; ragged shards are zero-padded to the widest shard first.
    Returns (y_full, work_or_None) — with async_op the raw [P, batch, rows] buffer is returned."""
    world = dist.get_world_size(group)
    batch = y_local.shape[0]
    if world == 1:
        return y_local, None
    if n_rows % world == 0:
        rows = n_rows // world
        assert y_local.shape == (batch, rows)
        # concatenated-along-dim-0 form [P*batch, rows]: accepted by both RCCL and gloo
        buf = out if out is not None else torch.empty((world * batch, rows), dtype=y_local.dtype,
                                                      device=y_local.device)
        work = dist.all_gather_into_tensor(buf.view(world * batch, rows), y_local.contiguous(), group=group,
                                           async_op=async_op)
        buf3 = buf.view(world, batch, rows)
        return (buf3.permute(1, 0, 2).reshape(batch, n_rows) if not async_op else buf3), work
    # ragged shards: collectives need equal contributions -> pad every slab to the widest shard,
    # gather, then drop the padding columns
    rows_max = -(-n_rows // world)
    padded = torch.zeros((batch, rows_max), dtype=y_local.dtype, device=y_local.device)
    padded[:, :y_local.shape[1]] = y_local
    buf = torch.empty((world * batch, rows_max), dtype=y_local.dtype, device=y_local.device)
    work = dist.all_gather_into_tensor(buf, padded, group=group, async_op=async_op)
    if async_op:
        return buf.view(world, batch, rows_max), work
    buf3 = buf.view(world, batch, rows_max)
    parts = []
    for r in range(world):
        s, e = shard_rows(n_rows, world, r)
        parts.append(buf3[r, :, :e - s])
    return torch.cat(parts, dim=1), None


class SlabGather:
    """Gather without a staging copy: a preallocated [P, batch, rows] buffer whose slot r is rank r's output slab.

    The rank's matmul writes its [batch, rows] slab straight into `local` (a contiguous view of slot `rank`:
    `ggq_mul_mat_q_ld` / the torch op with `out=`), and `gather()` all-gathers IN PLACE — the collective's input is
    the rank's own slot of its output (the in-place form of ncclAllGather), so no slab is copied locally and no
    second buffer exists.  Consumers that want [batch, N] take `batch_major()` (one permute copy) or, better,
    consume the [P, batch, rows] form directly (a following row-parallel layer reads it shard by shard).
    Equal shards only (n_rows % P == 0): ragged splits go through `gather_slabs`."""

    def __init__(self, batch: int, n_rows: int, dtype, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if n_rows % self.world:
            raise ValueError("SlabGather needs equal shards (n_rows % world_size == 0)")
        self.batch, self.rows = batch, n_rows // self.world
        self.buf = torch.empty((self.world, batch, self.rows), dtype=dtype, device=device)

    @property
    def local(self) -> torch.Tensor:
        return self.buf[self.rank]

    def gather(self, async_op: bool = False):
        if self.world == 1:
            return None
        return dist.all_gather_into_tensor(self.buf.view(self.world * self.batch, self.rows), self.local, group=self.group,
                                           async_op=async_op)

    def batch_major(self) -> torch.Tensor:
        return unpermute_gathered(self.buf)


class PeerSlabGather:
    """Gather without a collective: every rank maps the other ranks' [P, batch, rows] buffers (ggq_peer_export / _import:
    HIP IPC) and WRITES its slab straight into slot `rank` of each of them — device-to-device stores that cross xGMI
    when the ranks own different GPUs.  Same interface as SlabGather (`local`, `gather()`, `buf`, `batch_major()`).

    `local` is the rank's slot of its own buffer: the matmul writes there (out= / ldy).  `gather()` pushes that slab to
    the peers on the current stream, drains the stream and meets the other ranks at a barrier of the (CPU-capable)
    process group, after which `buf` holds every rank's slab.  This is the first step of the direct-write path of
    SURVEY 8e: the copy still follows the kernel instead of being the kernel's own stores, and the hand-off is a host
    barrier rather than a device flag; it needs no RCCL and has been exercised with two processes sharing one GPU
    (tests/test_peer_gather.py) — no multi-GPU node was available to this build."""

    def __init__(self, batch: int, n_rows: int, dtype, device, group=None):
        import ctypes
        from . import lib as ggqlib
        self.L = ggqlib.hip()
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if n_rows % self.world:
            raise ValueError("PeerSlabGather needs equal shards (n_rows % world_size == 0)")
        self.batch, self.rows = batch, n_rows // self.world
        self.buf = torch.empty((self.world, batch, self.rows), dtype=dtype, device=device)
        handle = (ctypes.c_ubyte * 64)()
        off = ctypes.c_int64(0)
        ggqlib.check(self.L.ggq_peer_export(ctypes.c_void_p(self.buf.data_ptr()), handle, ctypes.byref(off)), "ggq_peer_export")
        mine = (bytes(handle), int(off.value))
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine, group=group)
        self._peer_ptr, self._peer_off = {}, {}
        for p, (h, o) in enumerate(everyone):
            if p == self.rank:
                continue
            ptr = ctypes.c_void_p()
            hb = (ctypes.c_ubyte * 64).from_buffer_copy(h)
            ggqlib.check(self.L.ggq_peer_import(hb, o, ctypes.byref(ptr)), "ggq_peer_import")
            self._peer_ptr[p], self._peer_off[p] = ptr.value, o
        self._slab_bytes = batch * self.rows * self.buf.element_size()

    @property
    def local(self) -> torch.Tensor:
        return self.buf[self.rank]

    def gather(self):
        import ctypes
        from . import lib as ggqlib
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.buf.device).cuda_stream)
        row_bytes = self.rows * self.buf.element_size()
        src = ctypes.c_void_p(self.local.data_ptr())
        for p, base in self._peer_ptr.items():   # slot `rank` of peer p's buffer
            dst = ctypes.c_void_p(base + self.rank * self._slab_bytes)
            ggqlib.check(self.L.ggq_peer_write_2d(dst, row_bytes, src, row_bytes, row_bytes, self.batch, stream), "ggq_peer_write_2d")
        torch.cuda.current_stream(self.buf.device).synchronize()
        dist.barrier(group=self.group)
        return None

    def batch_major(self) -> torch.Tensor:
        return unpermute_gathered(self.buf)

    def close(self):
        for p, base in list(self._peer_ptr.items()):
            self.L.ggq_peer_close(base, self._peer_off[p])
        self._peer_ptr.clear()


def unpermute_gathered(buf: torch.Tensor) -> torch.Tensor:
    """[P, batch, rows] gather buffer -> [batch, P*rows]"""
    p, b, r = buf.shape
    return buf.permute(1, 0, 2).reshape(b, p * r)


class RowShardedQuantLinear:
    """y[batch, N] = x[batch, K] · W[N, K]^T with W row-sharded over the process group.

    `matmul` defaults to the drop-in GPU op (ggml.ggml_mul_mat_a8 / ggml_mul_mat_vec_a8);
    tests inject a CPU function with the same signature."""

    def __init__(self, w_shard: torch.Tensor, quant_type: int, n_rows: int, group=None,
                 matmul: Optional[Callable] = None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.quant_type = int(quant_type)
        self.n_rows = n_rows
        self.start, self.end = shard_rows(n_rows, self.world, self.rank)
        assert w_shard.shape[0] == self.end - self.start, "w_shard must hold exactly this rank's rows"
        self.w = w_shard
        self._matmul = matmul

    @staticmethod
    def shard_weight(w_full, n_rows: int, world: int, rank: int):
        s, e = shard_rows(n_rows, world, rank)
        return w_full[s:e]

    def local(self, x: torch.Tensor) -> torch.Tensor:
        if self._matmul is not None:
            return self._matmul(self.w, x, self.quant_type, self.end - self.start)
        import ggml
        if x.dim() == 2 and x.size(0) == 1:
            return ggml.ggml_mul_mat_vec_a8(self.w, x, self.quant_type, self.end - self.start)
        return ggml.ggml_mul_mat_a8(self.w, x, self.quant_type, self.end - self.start)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        y = self.local(x)
        if self.world == 1:
            return y
        full, _ = gather_slabs(y.reshape(-1, y.shape[-1]), self.n_rows, self.group)
        return full.reshape(*y.shape[:-1], self.n_rows)
