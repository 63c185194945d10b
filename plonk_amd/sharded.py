"""Multi-GPU proving: one process per GPU, the nine commitment MSMs sharded by point range.

Every rank holds the proving key and the witness (the NTT / quotient phases are replicated);
rank r holds the SRS slice point_range(n_points, r, world) and computes the partial MSM over
it; the 104-byte partial sums are all-gathered (RCCL over xGMI with the "nccl" backend, gloo on
CPU) and added in rank order on every rank, so all ranks derive the same transcript and proof.
"""
import ctypes as C

from . import _lib


def point_range(n_points, rank, world):
    """Contiguous, balanced split of [0, n_points) over the ranks: the first n_points % world ranks
    get one extra point."""
    base, extra = divmod(n_points, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class Communicator:
    """All-gather of fixed-size byte records over torch.distributed, callable from the C library."""

    def __init__(self, dist, rank, world, local_rank=None):
        import torch
        self.dist, self.rank, self.world = dist, rank, world
        backend = dist.get_backend()
        self.device = torch.device("cuda", local_rank if local_rank is not None else rank) if backend == "nccl" \
            else torch.device("cpu")
        self.calls = 0
        self.callback = _lib.ALLGATHER_FN(self._allgather)  # keep a reference: the C side holds the pointer

    def allgather_bytes(self, payload):
        import torch
        send = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(self.device)
        recv = torch.empty(self.world * len(payload), dtype=torch.uint8, device=self.device)
        self.dist.all_gather_into_tensor(recv, send) if self.device.type == "cuda" else \
            self.dist.all_gather(list(recv.chunk(self.world)), send)
        self.calls += 1
        return bytes(recv.cpu().numpy().tobytes())

    def _allgather(self, user, send, nbytes, recv):
        try:
            out = self.allgather_bytes(C.string_at(send, nbytes))
            C.memmove(recv, out, len(out))
            return 0
        except Exception as e:  # never unwind into C
            import sys
            print("plonk_amd.sharded: all-gather failed: %r" % (e,), file=sys.stderr)
            return -1


def combine_partials(points):
    """Sum of per-rank partial commitments (affine (x, y) or None), on the host via the C ABI."""
    from .api import _g1_struct, _g1_value
    arr = (_lib.G1 * max(len(points), 1))(*[_g1_struct(p) for p in points])
    out = _lib.G1()
    rc = _lib.lib.plonk_g1_sum(arr, len(points), C.byref(out))
    if rc != 0:
        raise _lib.PlonkError(rc, "plonk_g1_sum")
    return _g1_value(out)
