"""Multi-GPU proving: one process per GPU, the nine commitment MSMs sharded by point range.

Every rank holds the proving key and the witness; rank r holds the SRS slice
point_range(n_points, r, world) and computes its share of every MSM over it.  The shares (the
window sums of each rank's bucket reduction) are all-gathered on the device inside the library --
ncclAllGather over xGMI with the RCCL communicator, or staged through the host and a
torch.distributed (gloo) all-gather with the callback communicator -- and combined in rank order by
every rank, so all ranks derive the same transcript and proof.
"""
import ctypes as C

from . import _lib


def point_range(n_points, rank, world):
    """Contiguous, balanced split of [0, n_points) over the ranks: the first n_points % world ranks
    get one extra point."""
    base, extra = divmod(n_points, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def group_range(n_points, rank, world, batch):
    """The rank's point range when the `batch` commitments of a round are placed on groups of ranks:
    commitment k belongs to the ranks r with r % batch == k, which split [0, n_points) among them in
    rank order (plonk_prove_3wire_grouped).  Needs world >= batch."""
    if world < batch:
        raise ValueError("placement by groups needs at least as many ranks as commitments")
    members = (world - rank % batch + batch - 1) // batch
    return point_range(n_points, rank // batch, members)


class HostAllGather:
    """All-gather of fixed-size byte records over torch.distributed with host tensors (gloo),
    callable from the C library as a plonk_allgather_fn."""

    def __init__(self, dist, rank, world):
        self.dist, self.rank, self.world = dist, rank, world
        self.calls = 0
        self.callback = _lib.ALLGATHER_FN(self._allgather)  # keep a reference: the C side holds the pointer

    def allgather_bytes(self, payload):
        import torch
        send = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
        recv = torch.empty(self.world * len(payload), dtype=torch.uint8)
        self.dist.all_gather(list(recv.chunk(self.world)), send)
        self.calls += 1
        return recv.numpy().tobytes()

    def _allgather(self, user, send, nbytes, recv):
        try:
            out = self.allgather_bytes(C.string_at(send, nbytes))
            C.memmove(recv, out, len(out))
            return 0
        except Exception as e:  # never unwind into C
            import sys
            print("plonk_amd.sharded: all-gather failed: %r" % (e,), file=sys.stderr)
            return -1


class Communicator:
    """This rank's plonk_comm.  kind "rccl": device buffers over xGMI; the 128-byte RCCL id is made by
    rank 0 and handed to the other ranks through `dist` (any torch.distributed group, gloo will do --
    it carries nothing else).  kind "callback": the records are staged through the host and
    exchanged by `dist` itself (rehearsal where RCCL cannot run: ranks sharing one GPU)."""

    def __init__(self, ctx, dist, rank, world, kind="rccl"):
        self.ctx, self.rank, self.world, self.kind = ctx, rank, world, kind
        self.host = None
        h = C.c_void_p()
        if kind == "rccl":
            ident = (C.c_uint8 * _lib.COMM_ID_BYTES)()
            rc = _lib.lib.plonk_comm_unique_id(ident) if rank == 0 else 0
            # rank 0 always reaches the broadcast, with its status in front: a failure there (RCCL not loadable)
            # is then raised by every rank together instead of leaving the others waiting in the collective
            box = [(rc, bytes(ident))]
            if world > 1:
                dist.broadcast_object_list(box, src=0)
            rc, blob = box[0]
            if rc != 0:
                raise _lib.PlonkError(rc, "plonk_comm_unique_id failed on rank 0 (is RCCL installed?)")
            ident = (C.c_uint8 * _lib.COMM_ID_BYTES).from_buffer_copy(blob)
            ctx.check(_lib.lib.plonk_comm_create_rccl(ctx.h, ident, rank, world, C.byref(h)))
        elif kind == "callback":
            self.host = HostAllGather(dist, rank, world)
            ctx.check(_lib.lib.plonk_comm_create_callback(ctx.h, rank, world, self.host.callback, None, C.byref(h)))
        else:
            raise ValueError("communicator kind must be 'rccl' or 'callback'")
        self.h = h

    def stats(self):
        """(all-gathers issued, bytes this rank contributed)."""
        calls, nbytes = C.c_uint64(), C.c_uint64()
        rc = _lib.lib.plonk_comm_stats(self.h, C.byref(calls), C.byref(nbytes))
        if rc != 0:
            raise _lib.PlonkError(rc, "plonk_comm_stats")
        return calls.value, nbytes.value

    def timing(self, enable):
        """Stream time (ms) this rank's all-gathers took since the previous call, the wait for the slowest
        peer included; then switches the event marks on or off.  Call with the context synchronised."""
        ms = C.c_double()
        rc = _lib.lib.plonk_comm_timing(self.h, 1 if enable else 0, C.byref(ms))
        if rc != 0:
            raise _lib.PlonkError(rc, "plonk_comm_timing")
        return ms.value

    def close(self):
        if getattr(self, "h", None):
            _lib.lib.plonk_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def combine_partials(points):
    """Sum of per-rank partial commitments (affine (x, y) or None), on the host via the C ABI."""
    from .api import _g1_struct, _g1_value
    arr = (_lib.G1 * max(len(points), 1))(*[_g1_struct(p) for p in points])
    out = _lib.G1()
    rc = _lib.lib.plonk_g1_sum(arr, len(points), C.byref(out))
    if rc != 0:
        raise _lib.PlonkError(rc, "plonk_g1_sum")
    return _g1_value(out)
