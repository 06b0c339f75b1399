"""One rank of the CPU rehearsal of a proof's H computation with the transforms sharded over the ranks (see
test_host.py): gloo, no GPU.  The stage / exchange loop, the all-to-all and the layouts are the product's
(zinc_amd.distributed.run_sharded_stages); the four stages are big-integer stand-ins for zg16_hshard_stage that
follow its documented contract step for step, so this checks the schedule and the factor bookkeeping (which 1/G,
1/L, g^i, 1/Z goes where), not the kernels.  Every rank must end with H[s + G j] at slot j, H from the oracle."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import cpu_binding as cb  # noqa: E402
from zinc_amd.distributed import run_sharded_stages, exchange_pack, exchange_unpack  # noqa: E402
from zinc_amd.fields import R_MOD, ints_to_mont, mont_to_ints  # noqa: E402

P = R_MOD


def main():
    rank, world, port, log_m = (int(x) for x in sys.argv[1:5])
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    m = 1 << log_m
    G, L = world, m // world
    blk = L // G
    rng = np.random.default_rng(7)
    # row evaluations of a satisfied system: c = a * b on the domain, the last rows zero as padding would be
    a = [int(v) % P for v in rng.integers(1, 1 << 62, size=m)]
    b = [int(v) % P for v in rng.integers(1, 1 << 62, size=m)]
    for i in range(m - 5, m):
        a[i] = b[i] = 0
    c = [x * y % P for x, y in zip(a, b)]
    want, _ = cb.h_coefficients(ints_to_mont(a, P), ints_to_mont(b, P), ints_to_mont(c, P), threads=1)
    want = mont_to_ints(want, P) + [0]      # m - 1 coefficients; the m-th is zero

    omega = pow(pow(7, (P - 1) >> 28, P), 1 << (28 - log_m), P)
    omega_l = pow(omega, G, P)              # the L-point domain of a rank's local transforms
    g, z_inv = 7, pow(pow(7, m, P) - 1, -1, P)
    inv = lambda x: pow(x, -1, P)

    def local_ntt(v, inverse):              # unnormalised L-point transform
        w = inv(omega_l) if inverse else omega_l
        return [sum(v[j] * pow(w, j * k, P) for j in range(L)) % P for k in range(L)]

    def cross(v, inverse):                  # the cross-GPU step of ntt.hip on one rank's L elements
        w = inv(omega) if inverse else omega
        out = [0] * L
        for i in range(blk):
            k2 = rank * blk + i
            col = [v[gg * blk + i] for gg in range(G)]
            if not inverse:
                col = [x * pow(w, gg * k2, P) % P for gg, x in enumerate(col)]
            res = [sum(col[gg] * pow(w, L * gg * k1, P) for gg in range(G)) % P for k1 in range(G)]
            if inverse:
                res = [x * pow(w, gg * k2, P) * inv(G) % P for gg, x in enumerate(res)]
            for gg in range(G):
                out[gg * blk + i] = res[gg]
        return out

    work = torch.zeros((3, L, 4), dtype=torch.int64)
    got = torch.zeros((3, L, 4), dtype=torch.int64)
    rd = lambda t: mont_to_ints(np.ascontiguousarray(t.numpy().view(np.uint64)), P)

    def as_t(vals):
        return torch.from_numpy(ints_to_mont([x % P for x in vals], P).view(np.int64).copy())

    flat_w, flat_g = work.reshape(-1, 4), got.reshape(-1, 4)

    def send(vectors):   # the stage's vectors, destination-major, at the front of `work`
        flat_w[:len(vectors) * L].copy_(exchange_pack([as_t(v) for v in vectors], world))
        return len(vectors)

    def received(n):     # the n vectors the last exchange delivered, natural order
        return [rd(v.contiguous()) for v in exchange_unpack(flat_g[:n * L], n, world)]

    state = {}

    def stage(k):
        if k == 0:     # this rank's blocks of rows, then the cross step of the inverse transform
            rows = [(j // blk) * L + rank * blk + (j % blk) for j in range(L)]
            return send([cross([vec[r] for r in rows], True) for vec in (a, b, c)])
        if k == 1:     # c' = ifft(c) / Z; a, b: ifft, times g^i, forward local transform
            ga, gb, gc = received(3)
            cc = local_ntt(gc, True)
            state["c"] = [x * inv(L) * z_inv % P for x in cc]
            out = []
            for vec in (ga, gb):
                co = local_ntt(vec, True)
                co = [x * inv(L) * pow(g, rank + G * j, P) % P for j, x in enumerate(co)]
                out.append(local_ntt(co, False))
            return send(out)
        if k == 2:     # coset evaluations, their product, cross step of the last inverse transform
            ga, gb = received(2)
            A, B = cross(ga, False), cross(gb, False)
            return send([cross([x * y % P for x, y in zip(A, B)], True)])
        h = local_ntt(received(1)[0], True)
        state["h"] = [(x * inv(L) * pow(inv(g), rank + G * j, P) * z_inv - state["c"][j]) % P for j, x in enumerate(h)]
        return 0

    run_sharded_stages(stage, work, got, world)
    assert state["h"] == [want[rank + G * j] for j in range(L)], "rank %d: H coefficients differ" % rank
    dist.barrier()
    if rank == 0:
        print("HSHARD_MATCH", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
