"""What ONE rank of an N-rank proof does, timed alone on one GPU: zg16_prove_partial on shard 0 of N of the key
(every rank repeats the row evaluation and the transforms below 2^21 rows; the multiexps cover 1/N of the points).
The gather of the 640-byte partial sums and the host assembly are not in it.

    python tools/rank_share.py [--log-rows 20]
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
    for n in (1, 2, 4, 8):
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


if __name__ == "__main__":
    main()
