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
    """Gather without a collective and without the host: every rank maps the other ranks' gather buffers (ggq_peer_export /
    _import: HIP IPC) and `gather()` enqueues, on the current stream, ONE kernel that stores the rank's slab straight into
    slot `rank` of each peer's buffer (device-to-device stores that cross xGMI when the ranks own different GPUs) and
    publishes a generation number into the peers' flag words once every storing workgroup has released its stores at
    system scope, followed by one tiny kernel that waits for the peers' flags (ggq_peer_scatter / ggq_peer_wait).  No
    stream drain, no host barrier, no RCCL.  NOT for HIP-graph replay: the generation number and the buffer parity are host-side
    state passed as kernel arguments, so a replayed wait would return at once on the previous generation (stale slabs) —
    call gather() / matmul_gather() eagerly, once per gather.

    Same interface as SlabGather: the matmul writes the rank's slab into `local` (out= / ldy), `gather()` makes `buf`
    ([P, batch, rows]) complete for everything enqueued behind it on the stream, `batch_major()` is the [batch, N] copy.
    TWO buffers alternate by call parity: a fast rank's gather i + 1 writes the other buffer, so it cannot overwrite slabs
    a slow peer's consumers of gather i still read; by the time it reaches gather i + 2 (the same buffer again) it has
    waited for that peer's gather-(i + 1) slab, which the peer's stream produced after those consumers.  `local` / `buf`
    always name the buffer of the NEXT / LAST `gather()` respectively.
    `close()` is collective (barrier, unmap, barrier) and is also run by a context manager; a rank that fails inside
    `gather()` still reaches the status exchange.  Exercised with two and three processes sharing one GPU
    (tests/test_peer_gather.py: several gathers in a loop with a consumer kernel in between); no multi-GPU node was
    available to this build."""

    def __init__(self, batch: int, n_rows: int, dtype, device, group=None):
        import ctypes
        from . import lib as ggqlib
        self.L = ggqlib.hip()
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if n_rows % self.world:
            raise ValueError("PeerSlabGather needs equal shards (n_rows % world_size == 0)")
        if self.world > 8:   # ggq_mul_mat_q_gather / ggq_peer_scatter take at most 8 destinations (own slot + 7 peers)
            raise ValueError("PeerSlabGather: at most 8 ranks")
        self.batch, self.rows = batch, n_rows // self.world
        esz = torch.empty((), dtype=dtype).element_size()
        if (self.rows * esz) % 16:
            raise ValueError("PeerSlabGather needs slab rows of a multiple of 16 bytes")
        self._slab_bytes = batch * self.rows * esz
        self._buf_bytes = -(-self.world * self._slab_bytes // 256) * 256
        # one allocation: [2 parities]{ [P] slabs, 256-byte aligned } | [2][P] flag words | arrivals word | status word
        self._flags_off = 2 * self._buf_bytes
        total = self._flags_off + 2 * 64 * 4 + 256
        self._mem = torch.zeros(total, dtype=torch.uint8, device=device)
        self._bufs = [self._mem[p * self._buf_bytes:p * self._buf_bytes + self.world * self._slab_bytes].view(dtype).view(self.world, batch, self.rows)
                      for p in range(2)]
        self._arrivals = self._mem.data_ptr() + self._flags_off + 2 * 64 * 4
        self._status = self._arrivals + 64
        handle = (ctypes.c_ubyte * 64)()
        off = ctypes.c_int64(0)
        ggqlib.check(self.L.ggq_peer_export(ctypes.c_void_p(self._mem.data_ptr()), handle, ctypes.byref(off)), "ggq_peer_export")
        torch.cuda.synchronize(device)   # the zeroed flags are in memory before any peer can write them
        everyone = [None] * self.world
        dist.all_gather_object(everyone, (bytes(handle), int(off.value)), group=group)
        self._peer_ptr, self._peer_off = {}, {}
        for p, (h, o) in enumerate(everyone):
            if p == self.rank:
                continue
            ptr = ctypes.c_void_p()
            hb = (ctypes.c_ubyte * 64).from_buffer_copy(h)
            ggqlib.check(self.L.ggq_peer_import(hb, o, ctypes.byref(ptr)), "ggq_peer_import")
            self._peer_ptr[p], self._peer_off[p] = ptr.value, o
        self._calls = 0
        self._closed = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    @property
    def local(self) -> torch.Tensor:
        """the rank's slot of the buffer the NEXT gather() completes"""
        return self._bufs[self._calls & 1][self.rank]

    @property
    def buf(self) -> torch.Tensor:
        """[P, batch, rows] of the LAST gather() (before the first: of the next)"""
        return self._bufs[(self._calls - 1) & 1 if self._calls else 0]

    def gather(self):
        import ctypes
        from . import lib as ggqlib
        par = self._calls & 1
        gen = (self._calls >> 1) + 1                  # generation of this parity's buffer
        dev = self._mem.device
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        peers = sorted(self._peer_ptr)
        n = len(peers)
        row_bytes = self.rows * self._mem.new_empty(0, dtype=self._bufs[0].dtype).element_size()
        ok, err = True, None
        try:
            dsts = (ctypes.c_void_p * max(n, 1))(*[self._peer_ptr[p] + par * self._buf_bytes + self.rank * self._slab_bytes for p in peers])
            # peer p's flag word for THIS rank, in peer p's memory
            flg = (ctypes.c_void_p * max(n, 1))(*[self._peer_ptr[p] + self._flags_off + (par * 64 + self.rank) * 4 for p in peers])
            src = ctypes.c_void_p(self._bufs[par][self.rank].data_ptr())
            ggqlib.check(self.L.ggq_peer_scatter(src, row_bytes, dsts, flg, n, row_bytes, row_bytes, self.batch, gen,
                                                 ctypes.c_void_p(self._arrivals), stream), "ggq_peer_scatter")
            # my own flag words, one per source rank: contiguous [P] words of parity `par`; wait for the peers' entries
            # (the rank's own word is published here so that one contiguous poll covers all P)
            my_flags = self._mem.data_ptr() + self._flags_off + par * 64 * 4
            self._mem[self._flags_off + (par * 64 + self.rank) * 4:self._flags_off + (par * 64 + self.rank) * 4 + 4].view(torch.int32).fill_(gen)
            ggqlib.check(self.L.ggq_peer_wait(ctypes.c_void_p(my_flags), self.world, gen, ctypes.c_void_p(self._status), stream), "ggq_peer_wait")
        except Exception as e:   # keep the ranks in step: the failure is reported by close() / status()
            ok, err = False, e
        self._calls += 1
        if not ok:
            raise err
        return None

    def matmul_gather(self, x: torch.Tensor, w: torch.Tensor, quant_type, scratch: torch.Tensor = None):
        """y_rank = x · w_rankᵀ written by the GEMM kernel ITSELF into slot `rank` of every rank's buffer, flags published by the
        kernel's last arrival (ggq_mul_mat_q_gather — the 16-token-tile and the streamed kernel; one token: ggq_mul_mat_vec_q_gather,
        the fused GEMV: no copy, no second launch), then the wait for the peers' flags.  Falls back to ggq_mul_mat_q_ld into `local` +
        gather() for the (format, batch, shape) the other kernels serve (dot4 / LDS-tile / 64 x 64 wave tiles).
        w: this rank's [rows, row_bytes] shard on the device; x: [batch, k]."""
        import ctypes
        from . import lib as ggqlib
        L = self.L
        t = int(quant_type)
        if x.dim() != 2 or x.shape[0] != self.batch:
            raise ValueError(f"matmul_gather: x must be [{self.batch}, k], got {tuple(x.shape)}")
        if w.dim() != 2 or w.shape[0] != self.rows or w.dtype != torch.uint8:
            raise ValueError(f"matmul_gather: w must be this rank's uint8 [{self.rows}, row_bytes] shard, got {tuple(w.shape)} {w.dtype}")
        if x.device != self._mem.device or w.device != self._mem.device:
            raise ValueError("matmul_gather: x and w must live on the gather buffer's device")
        if not (x.is_contiguous() and w.is_contiguous()):
            raise ValueError("matmul_gather: x and w must be contiguous")
        if x.dtype != self._bufs[0].dtype:
            raise ValueError("matmul_gather: x must have the gather buffer's dtype (the slab is written in x's dtype)")
        k = x.shape[1]
        if scratch is None:
            scratch = torch.empty(int(L.ggq_mmq_scratch_bytes(self.batch, k)), dtype=torch.uint8, device=x.device)
        par = self._calls & 1
        gen = (self._calls >> 1) + 1
        dev = self._mem.device
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        vp = lambda tns: ctypes.c_void_p(tns.data_ptr())
        peers = sorted(self._peer_ptr)
        slot = par * self._buf_bytes + self.rank * self._slab_bytes
        flag_word = self._flags_off + (par * 64 + self.rank) * 4
        dsts = (ctypes.c_void_p * (len(peers) + 1))(self._mem.data_ptr() + slot, *[self._peer_ptr[p] + slot for p in peers])
        flg = (ctypes.c_void_p * (len(peers) + 1))(self._mem.data_ptr() + flag_word, *[self._peer_ptr[p] + flag_word for p in peers])
        if self.batch == 1:   # the reference's dispatch: one token goes to the GEMV (HK/ggml/ggml_kernel.cu:145-189)
            rc = L.ggq_mul_mat_vec_q_gather(vp(w), vp(x), dsts, len(peers) + 1, flg, len(peers) + 1, gen, ctypes.c_void_p(self._arrivals), t,
                                            ggqlib.dtype_code(x.dtype), k, self.rows, stream)
        else:
            rc = L.ggq_mul_mat_q_gather(vp(w), vp(x), dsts, len(peers) + 1, flg, len(peers) + 1, gen, ctypes.c_void_p(self._arrivals), t,
                                        ggqlib.dtype_code(x.dtype), self.batch, k, self.rows, self.rows, vp(scratch), stream)
        if rc == -2 and self.batch > 1:   # GGQ_ERR_SHAPE: a route without its own multi-destination write-back
            ggqlib.check(L.ggq_mul_mat_q_ld(vp(w), vp(x), vp(self.local), t, ggqlib.dtype_code(x.dtype), self.batch, k, self.rows, self.rows,
                                           vp(scratch), stream), "ggq_mul_mat_q_ld")
            return self.gather()
        ggqlib.check(rc, "ggq_mul_mat_q_gather")
        my_flags = self._mem.data_ptr() + self._flags_off + par * 64 * 4
        self._calls += 1
        ggqlib.check(L.ggq_peer_wait(ctypes.c_void_p(my_flags), self.world, gen, ctypes.c_void_p(self._status), stream), "ggq_peer_wait")
        return None

    def status(self) -> int:
        """0, or 1 if a ggq_peer_wait gave up on a peer after its 2-second limit (synchronises the device).  The wait kernel
        cannot stop the stream: work enqueued behind a timed-out wait has consumed incomplete slabs, so poll this before
        trusting results whenever a peer may legitimately stall that long; close() / the context manager check it and raise."""
        return int(self._mem[self._flags_off + 2 * 64 * 4 + 64:self._flags_off + 2 * 64 * 4 + 68].view(torch.int32).item())

    def batch_major(self) -> torch.Tensor:
        return unpermute_gathered(self.buf)

    def close(self):
        """collective: every rank's outstanding writes have landed and been consumed before any mapping goes away"""
        if self._closed:
            return
        self._closed = True
        torch.cuda.synchronize(self._mem.device)
        timed_out = self.status() != 0
        dist.barrier(group=self.group)
        rcs = [self.L.ggq_peer_close(base, self._peer_off[p]) for p, base in list(self._peer_ptr.items())]
        self._peer_ptr.clear()
        dist.barrier(group=self.group)
        if any(rc != 0 for rc in rcs):
            raise RuntimeError(f"ggq_peer_close failed: {rcs}")
        if timed_out:   # raised AFTER the collective part, so the ranks stay in step
            raise RuntimeError("PeerSlabGather: a ggq_peer_wait gave up on a peer (2 s): results gathered since then are incomplete")

    def __del__(self):
        # never collective from a finaliser: only drop the mappings if close() was skipped
        try:
            if not self._closed:
                for p, base in list(self._peer_ptr.items()):
                    self.L.ggq_peer_close(base, self._peer_off[p])
        except Exception:
            pass


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
