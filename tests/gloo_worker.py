"""One rank of the CPU rehearsal of the multi-GPU path (see test_host.py).  The multiexp sums of this
rank's point ranges come from the CPU oracle in place of the GPU kernels; sharding, the gather and
zg16_prove_finish are the product code."""
import ctypes as C
import sys

import numpy as np
import torch.distributed as dist

from conftest import golden, FixtureCircuit, FixtureKey, fixture_proof
from oracle import cpu_binding as cb
from zinc_amd import lib as zlib
from zinc_amd.distributed import shard_range, gather_partials, finish
from zinc_amd.fields import R_MOD, Q_MOD, ints_to_mont, mont_to_g1, mont_to_g2
from zinc_amd.groth16 import ProvingAssignment, Parameters


def xyzz(aff, width):
    """affine limbs -> XYZZ limbs (zz = zzz = 1), identity -> all zero."""
    out = np.zeros(2 * width, dtype=np.uint64)
    if aff.any():
        out[:width] = aff
        one = ints_to_mont([1], Q_MOD)[0]
        out[width:width + 4] = one
        out[width + width // 2: width + width // 2 + 4] = one
    return out


def main():
    rank, world = int(sys.argv[1]), int(sys.argv[2])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fx = golden("prove_tiny_3.json")
    circ, key = FixtureCircuit(fx), FixtureKey(fx)
    params = Parameters.from_points(key)
    pa = ProvingAssignment()
    pa.alloc_input(1)
    circ.synthesize(pa)
    pa.finalize_inputs()
    r1cs = pa.r1cs()
    zi, za = pa.witness()
    (ea, eb, ec), (da, dbi, dba) = cb.eval_rows(r1cs, zi, za)
    h, m = cb.h_coefficients(ea, eb, ec)
    z = np.concatenate([zi, za])
    n_in = r1cs.n_inputs
    a_map = list(range(n_in)) + [n_in + i for i in np.nonzero(da)[0]]
    b_map = [i for i in np.nonzero(dbi)[0]] + [n_in + i for i in np.nonzero(dba)[0]]

    def part_sum(group, bases, scalars):
        lo, hi = shard_range(bases.shape[0], rank, world)
        if hi == lo:
            return np.zeros(8 if group == "g1" else 16, dtype=np.uint64)
        st, out = cb.multiexp(group, bases[lo:hi], scalars[lo:hi])
        assert st == 0
        return out

    def add_g1(p, q):  # sum of two affine results via a 2-term multiexp with unit scalars
        pts = np.stack([x for x in (p, q) if x.any()]) if (p.any() or q.any()) else np.zeros((0, 8), np.uint64)
        if pts.shape[0] == 0:
            return np.zeros(8, np.uint64)
        st, out = cb.multiexp("g1", pts, ints_to_mont([1] * pts.shape[0], R_MOD))
        assert st == 0
        return out

    part = zlib.Partial()
    hl = add_g1(part_sum("g1", params.h[:m - 1], h), part_sum("g1", params.l, za))
    for name, val, width in (("hl", hl, 8), ("a", part_sum("g1", params.a, z[a_map]), 8),
                             ("b1", part_sum("g1", params.b_g1, z[b_map]), 8),
                             ("b2", part_sum("g2", params.b_g2, z[b_map]), 16)):
        C.memmove(getattr(part, name), xyzz(val, width).ctypes.data, 16 * width)
    parts = gather_partials(part, world, rank)
    if rank == 0:
        rs = ints_to_mont([int(fx["r"], 16), int(fx["s"], 16)], R_MOD)
        pv = params.view()
        proof = finish(pv, parts, world, rs[0], rs[1])
        got = (mont_to_g1(np.frombuffer(bytes(proof.a), dtype=np.uint64))[0],
               mont_to_g2(np.frombuffer(bytes(proof.b), dtype=np.uint64))[0],
               mont_to_g1(np.frombuffer(bytes(proof.c), dtype=np.uint64))[0])
        assert got == fixture_proof(fx), "sharded proof differs from the fixture"
        print("PROOF_MATCH")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    main()
