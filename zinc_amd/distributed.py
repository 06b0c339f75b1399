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
    if device is not None and dist.get_backend() != "gloo":
        mine = mine.to(device)
    # one flat receive buffer, so that a single copy brings every rank's partial back to the host
    flat = torch.empty(world * mine.numel(), dtype=torch.uint8, device=mine.device)
    try:
        dist.all_gather_into_tensor(flat, mine)
    except (RuntimeError, NotImplementedError, AttributeError):
        dist.all_gather(list(flat.chunk(world)), mine)
    return (_lib.Partial * world).from_buffer_copy(flat.cpu().numpy().tobytes())


def finish(pk_view, partials, n, r_mont, s_mont):
    """zg16_prove_finish on a list of partials (host arithmetic only)."""
    proof = _lib.Proof()
    st = _lib.load().zg16_prove_finish(C.byref(pk_view), partials, n, _lib.as_u64p(r_mont), _lib.as_u64p(s_mont),
                                       C.byref(proof))
    _check(None, st)
    return proof


def chain_owner(which, world):
    """Rank that runs transform chain `which` (0 = A, 1 = B, 2 = C) when the chains are split."""
    return which % world


def h_coefficients_split(ctx, mat, z_in_ptr, z_aux_ptr, on_device, rank, world, device):
    """The H coefficients with the three row-evaluation / ifft / coset_fft chains on different ranks.

    Chain k runs on rank k % world; the B and C coset evaluations travel point to point to rank 0
    (two different peers, so two different xGMI links when world >= 3), rank 0 forms
    icoset_fft((a*b - c)/Z) and broadcasts it.  Returns a device tensor of 2^log_m Fr (int64 x 4)
    on every rank.  `device` is the torch device of this rank's GPU."""
    import torch
    import torch.distributed as dist
    L = ctx._lib
    m = 1 << L.zg16_r1cs_log_domain(mat.handle)
    staged = dist.get_backend() == "gloo"  # CPU rehearsal: gloo moves host tensors

    def move(fn, t, peer):
        if staged:
            h = t.cpu()
            fn(h, peer)
            t.copy_(h)
        else:
            fn(t, peer)
            # RCCL's wait() only orders torch's current stream after the transfer; the library's kernels
            # run on its own streams, so the host waits here before handing the buffer over.
            torch.cuda.current_stream(device).synchronize()

    bufs = {}
    for which in range(3):
        if chain_owner(which, world) == rank or rank == 0:
            bufs[which] = torch.empty((m, 4), dtype=torch.int64, device=device)
    for which in range(3):
        if chain_owner(which, world) == rank:
            _check(ctx.handle, L.zg16_eval_chain(ctx.handle, mat.handle, C.c_void_p(z_in_ptr), C.c_void_p(z_aux_ptr),
                                                 int(on_device), which, C.c_void_p(bufs[which].data_ptr())))
    for which in (1, 2):
        owner = chain_owner(which, world)
        if owner == 0:
            continue
        if rank == owner:
            move(dist.send, bufs[which], 0)
        elif rank == 0:
            move(dist.recv, bufs[which], owner)
    h = bufs[0] if rank == 0 else torch.empty((m, 4), dtype=torch.int64, device=device)
    if rank == 0:
        _check(ctx.handle, L.zg16_h_from_cosets(ctx.handle, C.c_void_p(h.data_ptr()), C.c_void_p(bufs[1].data_ptr()),
                                                C.c_void_p(bufs[2].data_ptr()), m.bit_length() - 1))
    move(dist.broadcast, h, 0)
    return h


def prove_sharded(ctx, key, mat, z_in_ptr, z_aux_ptr, on_device, pk_view, r_mont, s_mont, rank, world, device=None,
                  split_transforms=False):
    """One proof across `world` ranks; returns the lib.Proof on rank 0 and None elsewhere.

    split_transforms: run the A, B, C transform chains on different ranks (h_coefficients_split)
    instead of every rank repeating all of them; pays off when the domain is large (the vectors
    moved are 32 * 2^log_m bytes each)."""
    if world == 1:
        # one GPU: zg16_prove overlaps the host's share of the assembly with the h multiexp
        proof = _lib.Proof()
        _check(ctx.handle, ctx._lib.zg16_prove(ctx.handle, key.handle, mat.handle, C.c_void_p(z_in_ptr),
                                               C.c_void_p(z_aux_ptr), int(on_device), _lib.as_u64p(r_mont),
                                               _lib.as_u64p(s_mont), C.byref(proof)))
        return proof
    part = _lib.Partial()
    if split_transforms and world > 1:
        h = h_coefficients_split(ctx, mat, z_in_ptr, z_aux_ptr, on_device, rank, world, device)
        _check(ctx.handle, ctx._lib.zg16_prove_partial_h(ctx.handle, key.handle, mat.handle, C.c_void_p(z_in_ptr),
                                                         C.c_void_p(z_aux_ptr), int(on_device),
                                                         C.c_void_p(h.data_ptr()), C.byref(part)))
    else:
        _check(ctx.handle, ctx._lib.zg16_prove_partial(ctx.handle, key.handle, mat.handle, C.c_void_p(z_in_ptr),
                                                       C.c_void_p(z_aux_ptr), int(on_device), C.byref(part)))
    parts = gather_partials(part, world, rank, device)
    if rank != 0:
        return None
    return finish(pk_view, parts, world, r_mont, s_mont)


# ---- transforms sharded through the whole H computation ----------------------------------------------
def exchange_pack(vectors, world):
    """n natural-order (L, 4) vectors -> the (n L, 4) destination-major buffer a stage hands to its one exchange:
    chunk h holds block h of vector 0, block h of vector 1, ...  (What zg16_hshard_stage writes directly; used by the
    CPU stand-ins of the gloo rehearsal.)"""
    import torch
    n, length = len(vectors), vectors[0].shape[0]
    blk = length // world
    return torch.stack([v.reshape(world, blk, 4) for v in vectors], dim=1).reshape(n * length, 4)


def exchange_unpack(flat, n, world):
    """The (n L, 4) buffer an exchange delivered (the block from rank g for vector v at (g n + v) blk) -> n
    natural-order (L, 4) vectors."""
    blk = flat.shape[0] // (n * world)
    t = flat.reshape(world, n, blk, 4)
    return [t[:, v].reshape(world * blk, 4) for v in range(n)]


def run_sharded_stages(stage, work, got, world, device=None, after_exchange=None, stream_ordered=False):
    """The stage / exchange loop of a proof whose transforms are sharded (zg16_hshard_stage's contract): call
    stage(k) for k = 0..3; it leaves n vectors, destination-major, in the first n L elements of `work`; ONE
    equal-split all-to-all moves them into the same range of `got`.  work, got: (3, L, 4) int64 tensors; stage is the
    library call on the GPU and a CPU stand-in in the gloo rehearsal (tests/gloo_hshard_worker.py).
    stream_ordered: the stages and the collectives are queued on one stream (RCCL), nothing waits on the host."""
    length = work.shape[1]
    flat_w, flat_g = work.reshape(-1, 4), got.reshape(-1, 4)
    for k in range(4):
        n = stage(k)
        if n:
            # straight into the receiving buffer (with gloo the exchange is staged through the host)
            _all_to_all(flat_w[:n * length], world, device, out=flat_g[:n * length], host_wait=not stream_ordered)
            if after_exchange is not None:
                after_exchange()


class ShardedProver:
    """One rank of a proof whose transforms are spread over the GPUs as well (zg16_hshard_*): this rank's key
    shard (h cyclic, the other queries by point range), its stage object and its two exchange buffers.

    Over RCCL the whole proof is stream-ordered: torch is handed the context's own main stream (no new stream: a
    process has few hardware queues and a fifth stream would serialise behind one of the context's four), the
    three all-to-alls are queued on it between the stages (RCCL's own stream waits for it and it waits for RCCL's,
    on the GPU), and the host waits once, for the results.  Over gloo (the one-GPU rehearsal) every exchange is
    staged through the host."""

    def __init__(self, ctx, params, mat, rank, world, device):
        import torch
        import torch.distributed as dist
        self.ctx, self.mat, self.rank, self.world, self.device = ctx, mat, rank, world, device
        self.log_g = world.bit_length() - 1
        assert 1 << self.log_g == world and 2 <= world <= 8, "2, 4 or 8 ranks"
        L = ctx._lib
        self.pk_view = params.view()
        self.key = C.c_void_p()
        _check(ctx.handle, L.zg16_pk_upload_cyclic_h(ctx.handle, C.byref(self.pk_view), rank, world, C.byref(self.key)))
        self.hs = C.c_void_p()
        _check(ctx.handle, L.zg16_hshard_create(ctx.handle, self.key, mat.handle, self.log_g, rank, C.byref(self.hs)))
        self.rows = L.zg16_hshard_vector_bytes(self.hs) // 32
        self.work = torch.empty((3, self.rows, 4), dtype=torch.int64, device=device)
        self.got = torch.empty((3, self.rows, 4), dtype=torch.int64, device=device)
        self.stream = None
        if dist.get_backend() != "gloo" and device is not None and device.type == "cuda":
            self.stream = torch.cuda.ExternalStream(ctx.get_stream(), device=device)
            _check(ctx.handle, L.zg16_hshard_stream_ordered(self.hs, 1))

    def prove(self, z_in_ptr, z_aux_ptr, on_device, r_mont, s_mont, early_randomness=True):
        """Returns the lib.Proof on rank 0 and None elsewhere.  early_randomness=False: the ranks' partials are the
        plain query sums and rank 0 assembles them with zg16_prove_finish after the gather (the same proof, with
        the whole assembly behind the gather)."""
        import contextlib
        import torch
        L = self.ctx._lib

        def stage(k):
            n = C.c_int()
            _check(self.ctx.handle, L.zg16_hshard_stage(self.hs, k, C.c_void_p(z_in_ptr), C.c_void_p(z_aux_ptr),
                                                        int(on_device), C.c_void_p(self.work.data_ptr()),
                                                        C.c_void_p(self.got.data_ptr()), C.byref(n)))
            return n.value

        # r and s up front: every rank folds its share of s*a + r*b1 into its partial beside its h multiexp, and
        # rank 0 prepares the key's terms on a host thread of the library while the stages run
        if early_randomness:
            _check(self.ctx.handle, L.zg16_hshard_randomness(self.hs, _lib.as_u64p(r_mont), _lib.as_u64p(s_mont),
                                                             int(self.rank == 0)))
        ordered = self.stream is not None
        with torch.cuda.stream(self.stream) if ordered else contextlib.nullcontext():
            # staged (gloo): the library's kernels run on its own streams, so torch's copies must have landed before
            # the next stage; ordered (RCCL): everything is on self.stream
            run_sharded_stages(stage, self.work, self.got, self.world, self.device,
                               after_exchange=None if ordered else (lambda: torch.cuda.synchronize(self.device)),
                               stream_ordered=ordered)
            part = _lib.Partial()
            _check(self.ctx.handle, L.zg16_hshard_finish(self.hs, C.byref(part)))
            parts = gather_partials(part, self.world, self.rank, self.device)
        if self.rank != 0:
            return None
        if not early_randomness:
            return finish(self.pk_view, parts, self.world, r_mont, s_mont)
        proof = _lib.Proof()
        _check(self.ctx.handle, L.zg16_hshard_assemble(self.hs, parts, self.world, C.byref(proof)))
        return proof

    def free(self):
        L = self.ctx._lib
        if self.hs:
            L.zg16_hshard_free(self.hs)
            self.hs = C.c_void_p()
        if self.key:
            L.zg16_pk_free(self.key)
            self.key = C.c_void_p()
        self.stream = None


# ---- one transform over several GPUs --------------------------------------------------------------
def ntt_shard_input(x, rank, world):
    """Rank `rank`'s part of a natural-order vector x (rows = elements): x[rank + world * j]."""
    return x[rank::world]


def ntt_shard_output_index(n, rank, world):
    """Natural-order indices of the elements rank `rank` holds after a forward sharded transform of n
    points, in the order it holds them: block k1 (of n / world^2 elements) holds
    X[(rank * blk + i) + (n / world) * k1]."""
    import numpy as np
    length = n // world
    blk = length // world
    k1 = np.arange(world).reshape(-1, 1)
    i = np.arange(blk).reshape(1, -1)
    return (rank * blk + i + length * k1).reshape(-1)


def _all_to_all(t, world, device, out=None, host_wait=True):
    """Equal-split all-to-all of a (L, 4) int64 tensor: chunk h goes to rank h, which stores what it
    gets from rank g at chunk g.  RCCL moves device memory; with gloo (CPU rehearsal) the data is staged
    through the host.  out: a tensor of the same shape to receive into (a different one from t); without it the
    result is a new tensor (RCCL) or t itself (gloo).  host_wait = False (RCCL only): the caller's next work is
    queued on torch's current stream, which the collective already orders -- no wait on the host."""
    import torch
    import torch.distributed as dist
    staged = dist.get_backend() == "gloo"
    src = t.cpu() if staged else t
    dst = out if out is not None and not staged else torch.empty_like(src)
    try:
        dist.all_to_all_single(dst, src)
    except RuntimeError:
        # a backend without all-to-all: pairwise exchanges
        ins, outs = list(src.chunk(world)), list(dst.chunk(world))
        rank = dist.get_rank()
        outs[rank].copy_(ins[rank])
        reqs = []
        for peer in range(world):
            if peer != rank:
                reqs.append(dist.isend(ins[peer].contiguous(), peer))
                reqs.append(dist.irecv(outs[peer], peer))
        for r in reqs:
            r.wait()
    if staged:
        res = t if out is None else out
        res.copy_(dst)
        return res
    # RCCL's wait() orders only torch's current stream; unless the library works on that stream too, wait here
    if host_wait:
        torch.cuda.current_stream(device).synchronize()
    return dst


def ntt_sharded(ctx, part, log_n, rank, world, inverse=False, device=None):
    """`EvaluationDomain::fft` / `ifft` of 2^log_n points held by `world` ranks (a power of two, at most
    8), one all-to-all.  part: this rank's 2^log_n / world elements as an (L, 4) int64 device tensor
    -- x[rank + world * j] going into a forward transform (ntt_shard_input), the layout of
    ntt_shard_output_index coming out; the inverse maps that layout back.  Returns the transformed
    tensor (a new one when RCCL moved the data, `part` itself otherwise)."""
    L = ctx._lib
    log_g = world.bit_length() - 1
    assert 1 << log_g == world and world <= 8, "the shard count must be a power of two, at most 8"
    log_l = log_n - log_g
    assert part.shape[0] == 1 << log_l and part.is_contiguous()

    def local(t):
        _check(ctx.handle, L.zg16_ntt(ctx.handle, C.c_void_p(t.data_ptr()), log_l, int(inverse), 0, 1))

    def cross(t):
        _check(ctx.handle, L.zg16_ntt_cross(ctx.handle, C.c_void_p(t.data_ptr()), log_n, log_g, rank, int(inverse), 1))

    return exchange_schedule(part, world, inverse, local, cross, device)


def exchange_schedule(part, world, inverse, local, cross, device=None):
    """The order of steps of a sharded transform around its one all-to-all; `local` and `cross` transform
    a tensor in place (the library calls in ntt_sharded; CPU stand-ins in the gloo rehearsal)."""
    if world == 1:
        local(part)
        return part
    if not inverse:
        local(part)
        out = _all_to_all(part, world, device)
        cross(out)
    else:
        cross(part)
        out = _all_to_all(part, world, device)
        local(out)
    return out
