"""One rank of the CPU rehearsal of the sharded transform (see test_host.py): gloo, no GPU.  The
exchange and the layouts are the product's (zinc_amd.distributed); the per-rank transforms are CPU
stand-ins (the oracle's FFT for the local part, big-integer arithmetic for the cross step), so
this checks the schedule, the all-to-all and the index maps, not the kernels."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import bn254 as O, cpu_binding as cb  # noqa: E402
from zinc_amd.distributed import exchange_schedule, ntt_shard_input, ntt_shard_output_index  # noqa: E402
from zinc_amd.fields import R_MOD, ints_to_mont, mont_to_ints  # noqa: E402


def main():
    rank, world, port, log_n = (int(x) for x in sys.argv[1:5])
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    n = 1 << log_n
    length, blk = n // world, n // world // world
    rng = np.random.default_rng(99)
    x = ints_to_mont([int(v) % R_MOD for v in rng.integers(1, 1 << 62, size=n)], R_MOD)
    omega = pow(pow(7, (R_MOD - 1) >> 28, R_MOD), 1 << (28 - log_n), R_MOD)

    for inverse in (False, True):
        w = pow(omega, -1, R_MOD) if inverse else omega

        def local(t):
            a = np.ascontiguousarray(t.numpy().view(np.uint64))
            t.copy_(torch.from_numpy(cb.fft(a, inverse, False, threads=1).view(np.int64)))

        def cross(t):
            v = mont_to_ints(np.ascontiguousarray(t.numpy().view(np.uint64)), R_MOD)
            out = [0] * length
            ginv = pow(world, -1, R_MOD)
            for i in range(blk):
                k2 = rank * blk + i
                col = [v[g * blk + i] for g in range(world)]
                if not inverse:
                    col = [c * pow(w, g * k2, R_MOD) % R_MOD for g, c in enumerate(col)]
                res = [sum(col[g] * pow(w, (n // world) * g * k1, R_MOD) for g in range(world)) % R_MOD for k1 in range(world)]
                if inverse:
                    res = [c * pow(w, g * k2, R_MOD) * ginv % R_MOD for g, c in enumerate(res)]
                for g in range(world):
                    out[g * blk + i] = res[g]
            t.copy_(torch.from_numpy(ints_to_mont(out, R_MOD).view(np.int64)))

        whole = cb.fft(x, False, False, threads=1)          # X = fft(x), natural order
        idx = ntt_shard_output_index(n, rank, world)
        if not inverse:
            part = torch.from_numpy(np.ascontiguousarray(ntt_shard_input(x, rank, world)).view(np.int64)).clone()
            got = exchange_schedule(part, world, False, local, cross)
            assert np.array_equal(got.numpy().view(np.uint64), whole[idx]), "forward sharded transform differs"
        else:
            part = torch.from_numpy(np.ascontiguousarray(whole[idx]).view(np.int64)).clone()
            got = exchange_schedule(part, world, True, local, cross)
            assert np.array_equal(got.numpy().view(np.uint64), ntt_shard_input(x, rank, world)), "inverse sharded transform differs"
    dist.barrier()
    if rank == 0:
        print("NTT_SHARDED_MATCH", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
