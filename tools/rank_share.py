"""What ONE rank of an N-rank proof does, timed alone on one GPU.

Replicated schedule: zg16_prove_partial on shard 0 of N of the key (every rank repeats the row evaluation and the
transforms; the multiexps cover 1/N of the points).
Sharded schedule (--schedule sharded): shard 0's four zg16_hshard_stage calls, stream-ordered as over RCCL, with each
exchange replaced by a device copy of the same bytes from `work` to `got` on that stream (what arrives is then not
the other ranks' data, but it is dense field elements like theirs: the kernels' work is the same).  The time of the
real all-to-alls is NOT in it: at 2^20 rows on 8 GPUs they move 3.5 + 2.3 + 1.2 MiB per GPU and direction.
The gather of the 640-byte partial sums is not in either; the host assembly behind it is reported separately.

    python tools/rank_share.py [--log-rows 20] [--schedule replicated|sharded]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
TOXIC = dict(k1=0x1234567, k2=0x7654321, alpha=0xA11CE5, beta=0xBEEF77, gamma=0x6A33A, delta=0xDE17A, tau=0x7A0F00D)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-rows", type=int, default=20)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--schedule", choices=("replicated", "sharded"), default="replicated")
    ap.add_argument("--ranks", type=int, nargs="*", default=None, help="rank counts to time (default: all)")
    args = ap.parse_args()
    import torch
    torch.cuda.set_device(0)
    import zinc_amd
    from zinc_amd import lib as zlib, workload
    from zinc_amd.groth16 import _check
    ctx = zinc_amd.Context(0)
    circ = workload.SyntheticCircuit(1 << args.log_rows)
    r1cs = circ.r1cs()
    params = workload.generate_parameters(ctx, r1cs, TOXIC)
    mat = r1cs.upload(ctx)
    zi, za = circ.witness()
    zi_d = torch.from_numpy(zi.view(np.int64)).cuda()
    za_d = torch.from_numpy(za.view(np.int64)).cuda()
    if args.schedule == "sharded":
        return sharded(args, ctx, params, mat, zi_d, za_d)
    for n in (args.ranks or (1, 2, 4, 8)):
        key = params.upload(ctx, 0, n)
        part = zlib.Partial()

        def step():
            _check(ctx.handle, ctx._lib.zg16_prove_partial(ctx.handle, key.handle, mat.handle, C.c_void_p(zi_d.data_ptr()),
                                                           C.c_void_p(za_d.data_ptr()), 1, C.byref(part)))
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / args.reps
        print("ranks %d: shard 0 alone %.3f ms per proof; stage ms %s" % (n, ms, {k: round(v, 3) for k, v in ctx.timings().items()}), flush=True)
        key.free()


def sharded(args, ctx, params, mat, zi_d, za_d):
    import torch
    from zinc_amd import lib as zlib
    from zinc_amd.groth16 import _check
    L = ctx._lib
    stream = torch.cuda.ExternalStream(ctx.get_stream())   # the context's own main stream, as over RCCL
    for n in (args.ranks or (2, 4, 8)):
        log_g = n.bit_length() - 1
        view = params.view()
        key, hs = C.c_void_p(), C.c_void_p()
        _check(ctx.handle, L.zg16_pk_upload_cyclic_h(ctx.handle, C.byref(view), 0, n, C.byref(key)))
        _check(ctx.handle, L.zg16_hshard_create(ctx.handle, key, mat.handle, log_g, 0, C.byref(hs)))
        rows = L.zg16_hshard_vector_bytes(hs) // 32
        work = torch.zeros((3 * rows, 4), dtype=torch.int64, device="cuda")
        got = torch.zeros((3 * rows, 4), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        _check(ctx.handle, L.zg16_hshard_stream_ordered(hs, 1))
        part = zlib.Partial()

        from zinc_amd.fields import ints_to_mont, R_MOD
        rs = ints_to_mont([pow(7, 123456789, R_MOD), pow(5, 987654321, R_MOD)], R_MOD)

        def step():
            # every rank folds s*a + r*b1 of its sums into its partial (rank 0 would also prepare the key's terms,
            # on a thread of its own)
            _check(ctx.handle, L.zg16_hshard_randomness(hs, zlib.as_u64p(rs[0]), zlib.as_u64p(rs[1]), 1))
            with torch.cuda.stream(stream):
                for k in range(4):
                    nx = C.c_int()
                    _check(ctx.handle, L.zg16_hshard_stage(hs, k, C.c_void_p(zi_d.data_ptr()), C.c_void_p(za_d.data_ptr()), 1,
                                                           C.c_void_p(work.data_ptr()), C.c_void_p(got.data_ptr()), C.byref(nx)))
                    if nx.value:
                        got[:nx.value * rows].copy_(work[:nx.value * rows])
                _check(ctx.handle, L.zg16_hshard_finish(hs, C.byref(part)))
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / args.reps
        # what rank 0 does after the gather, on n copies of this partial (the sums are then not a proof's, the work
        # is): zg16_hshard_assemble against zg16_prove_finish on partials made without zg16_hshard_randomness
        parts = (zlib.Partial * n)(*([part] * n))
        proof = zlib.Proof()
        reps, t_new = 20, 0.0
        for _ in range(reps):
            step()
            t1 = time.perf_counter()
            _check(ctx.handle, L.zg16_hshard_assemble(hs, parts, n, C.byref(proof)))
            t_new += time.perf_counter() - t1
        t1 = time.perf_counter()
        for _ in range(reps):
            _check(None, L.zg16_prove_finish(C.byref(view), parts, n, zlib.as_u64p(rs[0]), zlib.as_u64p(rs[1]), C.byref(proof)))
        t_old = time.perf_counter() - t1
        print("ranks %d, sharded schedule: shard 0 alone %.3f ms per proof (exchanges as device copies); assembly after "
              "the gather %.0f us (zg16_hshard_assemble), %.0f us (zg16_prove_finish)"
              % (n, ms, 1e6 * t_new / reps, 1e6 * t_old / reps), flush=True)
        L.zg16_hshard_free(hs)
        L.zg16_pk_free(key)


if __name__ == "__main__":
    main()
