"""Stand-alone stage rates on one MI355X: G1 multiexp over resident bases and the Fr transform.

    python tools/bench_standalone.py [--log-n 24] [--reps 3] [--no-tables]

Bases are k_i * G built by the GPU fixed-base multiplier, scalars uniform below r; both are in HBM
before the timed region.  Prints one JSON line per stage.  Not part of bench.py's contract.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=24)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-tables", action="store_true")
    ap.add_argument("--skip-msm", action="store_true")
    ap.add_argument("--skip-ntt", action="store_true")
    args = ap.parse_args()
    import torch
    import zinc_amd
    from zinc_amd import workload
    from zinc_amd.fields import R_MOD
    from zinc_amd.groth16 import ResidentBases, _check
    ctx = zinc_amd.Context(0)
    n = 1 << args.log_n
    rng = np.random.default_rng(1)

    def rand_fr(count):
        a = rng.integers(0, 1 << 63, size=(count, 4), dtype=np.uint64)
        a[:, 3] &= (1 << 59) - 1  # below 2^251 < r: a valid Montgomery representation of some residue
        return a

    if not args.skip_ntt:
        x = torch.from_numpy(rand_fr(n).view(np.int64)).cuda()
        best = {}
        for inverse in (0, 1):
            ts = []
            for _ in range(args.reps + 1):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _check(ctx.handle, ctx._lib.zg16_ntt(ctx.handle, C.c_void_p(x.data_ptr()), args.log_n, inverse, 0, 1))
                ts.append(time.perf_counter() - t0)
            best[inverse] = min(ts[1:])
        print(json.dumps({"stage": "fr_ntt", "log_n": args.log_n, "fft_s": best[0], "ifft_s": best[1],
                          "fft_algorithmic_GBps": 64.0 * n / best[0] / 1e9, "elements_per_s": n / best[0]}), flush=True)
        del x
    if not args.skip_msm:
        g1gen, _ = workload._generators()
        tb = time.time()
        bases = workload.fixed_base_g1(ctx, g1gen, rand_fr(n))
        t1 = time.time()
        res = ResidentBases(ctx, 1, bases, window_tables=not args.no_tables)
        t2 = time.time()
        del bases
        sc = torch.from_numpy(rand_fr(n).view(np.int64)).cuda()
        ts = []
        for _ in range(args.reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = res.msm(device_ptr=sc.data_ptr())
            ts.append(time.perf_counter() - t0)
        best = min(ts[1:])
        print(json.dumps({"stage": "msm_g1", "log_n": args.log_n, "window_tables": not args.no_tables, "seconds": best,
                          "points_per_s": n / best, "bases_s": t1 - tb, "upload_and_tables_s": t2 - t1,
                          "result_limb0": int(out[0])}), flush=True)
        res.free()
    ctx.close()


if __name__ == "__main__":
    main()
