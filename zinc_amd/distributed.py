"""One proof over several GPUs: one process per GPU, key queries sharded by contiguous point range.

Every rank evaluates the rows and the H polynomial itself (that work is small next to the
multiexps), computes the multiexp sums of its own point range, and the XYZZ partial sums (640
bytes per rank) are all-gathered with torch.distributed -- RCCL over xGMI when the backend is
"nccl", gloo in CPU tests.  Rank 0 adds the partials and assembles A, B, C.  The only exchange in
the whole proof is that gather; there is no data-path collective on the multiexp inputs.
"""
import ctypes as C

from . import lib as _lib
from .groth16 import _check


def shard_range(n, index, count):
    """Point range [lo, hi) of shard `index`; the same split zg16_pk_upload applies to every query."""
    return n * index // count, n * (index + 1) // count


def gather_partials(partial, world, rank, device=None):
    """All-gather one `lib.Partial` per rank.  Returns a ctypes array of `world` partials."""
    import torch
    import torch.distributed as dist
    mine = torch.frombuffer(bytearray(bytes(partial)), dtype=torch.uint8)
    if device is not None:
        mine = mine.to(device)
    outs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(outs, mine)
    host = b"".join(t.cpu().numpy().tobytes() for t in outs)
    return (_lib.Partial * world).from_buffer_copy(host)


def finish(pk_view, partials, n, r_mont, s_mont):
    """zg16_prove_finish on a list of partials (host arithmetic only)."""
    proof = _lib.Proof()
    st = _lib.load().zg16_prove_finish(C.byref(pk_view), partials, n, _lib.as_u64p(r_mont), _lib.as_u64p(s_mont),
                                       C.byref(proof))
    _check(None, st)
    return proof


def prove_sharded(ctx, key, mat, z_in_ptr, z_aux_ptr, on_device, pk_view, r_mont, s_mont, rank, world, device=None):
    """One proof across `world` ranks; returns the lib.Proof on rank 0 and None elsewhere."""
    part = _lib.Partial()
    _check(ctx.handle, ctx._lib.zg16_prove_partial(ctx.handle, key.handle, mat.handle, C.c_void_p(z_in_ptr),
                                                   C.c_void_p(z_aux_ptr), int(on_device), C.byref(part)))
    if world == 1:
        return finish(pk_view, C.byref(part), 1, r_mont, s_mont)
    parts = gather_partials(part, world, rank, device)
    if rank != 0:
        return None
    return finish(pk_view, parts, world, r_mont, s_mont)
