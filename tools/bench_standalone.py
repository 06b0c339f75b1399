"""Stand-alone stage rates: G1 multiexp over resident bases and the Fr transform, on one MI355X or
sharded over several.

    python tools/bench_standalone.py [--log-n 24] [--reps 3] [--no-tables]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/bench_standalone.py --log-n 26

Bases are k_i * G built by the GPU fixed-base multiplier, scalars uniform below r; both are in HBM
before the timed region.  With N ranks (one per GPU, RCCL; ZG16_DIST_BACKEND=gloo rehearses the same
code with every rank on GPU 0) the multiexp is split by point range -- each rank sums its 2^log_n / N
points and the N affine partial sums are all-gathered -- and the transform runs as one sharded
transform with a single all-to-all (zinc_amd.distributed.ntt_sharded).  Times are the max over
ranks.  Prints one JSON line per stage on rank 0.  Not part of bench.py's contract.
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
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    backend = os.environ.get("ZG16_DIST_BACKEND", "nccl")
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    import zinc_amd
    from zinc_amd import workload, lib as zlib
    from zinc_amd.distributed import ntt_sharded
    from zinc_amd.groth16 import ResidentBases, _check
    ctx = zinc_amd.Context(local_rank)
    n = 1 << args.log_n
    n_loc = n // world
    rng = np.random.default_rng(1 + rank)

    def rand_fr(count):
        a = rng.integers(0, 1 << 63, size=(count, 4), dtype=np.uint64)
        a[:, 3] &= (1 << 59) - 1  # below 2^251 < r: a valid Montgomery representation of some residue
        return a

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn):
        """Best of `reps` runs after one warm-up; each run fenced on both sides, max over ranks."""
        ts = []
        for _ in range(args.reps + 1):
            fence()
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            t = time.perf_counter() - t0
            if world > 1:
                tt = torch.tensor([t], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                t = float(tt.item())
            ts.append(t)
        return min(ts[1:]), out

    def say(**kw):
        if rank == 0:
            print(json.dumps(dict(kw, n_gpus=world, backend=backend if world > 1 else None)), flush=True)

    if not args.skip_ntt:
        x = torch.from_numpy(rand_fr(n_loc).view(np.int64)).to(dev)
        best = {}
        for inverse in (0, 1):
            if world == 1:
                best[inverse], _ = timed(lambda: _check(ctx.handle, ctx._lib.zg16_ntt(ctx.handle, C.c_void_p(x.data_ptr()),
                                                                                      args.log_n, inverse, 0, 1)))
            else:
                def run():
                    nonlocal x
                    x = ntt_sharded(ctx, x, args.log_n, rank, world, bool(inverse), dev)
                best[inverse], _ = timed(run)
        say(stage="fr_ntt", log_n=args.log_n, fft_s=best[0], ifft_s=best[1],
            fft_algorithmic_GBps=64.0 * n / best[0] / 1e9, elements_per_s=n / best[0])
        del x
    if not args.skip_msm:
        g1gen, _ = workload._generators()
        tb = time.time()
        bases = workload.fixed_base_g1(ctx, g1gen, rand_fr(n_loc))   # this rank's point range
        t1 = time.time()
        res = ResidentBases(ctx, 1, bases, window_tables=not args.no_tables)
        t2 = time.time()
        del bases
        sc = torch.from_numpy(rand_fr(n_loc).view(np.int64)).to(dev)

        def run():
            mine = np.ascontiguousarray(res.msm(device_ptr=sc.data_ptr()), dtype=np.uint64).reshape(1, 8)
            if world == 1:
                return mine[0]
            t = torch.from_numpy(mine.view(np.int64))
            t = t.to(dev) if backend == "nccl" else t
            outs = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(outs, t)
            acc = outs[0].cpu().numpy().view(np.uint64).copy()
            for o in outs[1:]:   # projective addition of two affine points through the library's group code
                q = np.ascontiguousarray(o.cpu().numpy().view(np.uint64))
                nxt = np.zeros_like(acc)
                _check(ctx.handle, ctx._lib.zg16_selftest_group(ctx.handle, 1, 2, zlib.as_u64p(acc), zlib.as_u64p(q), 1,
                                                                zlib.as_u64p(nxt)))
                acc = nxt
            return acc[0]
        best, out = timed(run)
        say(stage="msm_g1", log_n=args.log_n, window_tables=not args.no_tables, seconds=best, points_per_s=n / best,
            bases_s=t1 - tb, upload_and_tables_s=t2 - t1, result_limb0=int(out[0]))
        res.free()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
