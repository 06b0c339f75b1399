"""One rank of the multi-process GPU test of a proof whose transforms are sharded through the whole H computation
(zinc_amd.distributed.ShardedProver; see test_gpu_parity.py).  All ranks share GPU 0 and talk over gloo, which
stands in for RCCL on a one-GPU box; rank 0 checks the assembled proof against the single-GPU zg16_prove."""
import sys

import numpy as np
import torch
import torch.distributed as dist

import zinc_amd
from zinc_amd import workload
from zinc_amd.distributed import ShardedProver
from zinc_amd.fields import R_MOD, ints_to_mont
from zinc_amd.groth16 import prove_recorded


def main():
    rank, world, port, log_rows = (int(x) for x in sys.argv[1:5])
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    torch.cuda.set_device(0)
    ctx = zinc_amd.Context(0)
    circ = workload.SyntheticCircuit(1 << log_rows, n_pub=4)
    params = workload.generate_parameters(ctx, circ, dict(k1=3, k2=5, alpha=7, beta=11, gamma=13, delta=17, tau=19))
    r1cs = circ.r1cs()
    zi, za = circ.witness()
    rs = ints_to_mont([424242, 171717], R_MOD)
    mat = r1cs.upload(ctx)
    prover = ShardedProver(ctx, params, mat, rank, world, torch.device("cuda", 0))
    z_in = torch.from_numpy(zi.view(np.int64)).cuda()
    z_aux = torch.from_numpy(za.view(np.int64)).cuda()
    for _ in range(2):   # the second proof reuses every buffer and table
        proof = prover.prove(z_in.data_ptr(), z_aux.data_ptr(), 1, rs[0], rs[1])
    # a proof given its r and s and abandoned after stage 0 on every rank: the next one, started without a new
    # zg16_hshard_randomness, must not inherit them ...
    import ctypes as C
    from zinc_amd import lib as zlib
    L = ctx._lib
    zinc_amd.groth16._check(ctx.handle, L.zg16_hshard_randomness(prover.hs, zlib.as_u64p(rs[1]), zlib.as_u64p(rs[0]), int(rank == 0)))
    n_ex = C.c_int()
    zinc_amd.groth16._check(ctx.handle, L.zg16_hshard_stage(prover.hs, 0, C.c_void_p(z_in.data_ptr()), C.c_void_p(z_aux.data_ptr()), 1,
                                                            C.c_void_p(prover.work.data_ptr()), C.c_void_p(prover.got.data_ptr()), C.byref(n_ex)))
    torch.cuda.synchronize()
    dist.barrier()
    # ... it is the same proof with plain partial sums and the whole assembly on rank 0 (zg16_prove_finish)
    plain = prover.prove(z_in.data_ptr(), z_aux.data_ptr(), 1, rs[0], rs[1], early_randomness=False)
    if rank == 0:
        from zinc_amd.groth16 import SynthesisError
        for name in ("a", "b", "c"):
            assert bytes(getattr(plain, name)) == bytes(getattr(proof, name)), "the two assemblies differ in %s" % name
        # zg16_hshard_assemble without this proof's r and s is refused
        try:
            zinc_amd.groth16._check(ctx.handle, ctx._lib.zg16_hshard_assemble(prover.hs, (zlib.Partial * world)(), world,
                                                                               C.byref(zlib.Proof())))
            raise AssertionError("zg16_hshard_assemble accepted partials without zg16_hshard_randomness")
        except SynthesisError as e:
            assert e.status_name == "InvalidArgument", e
        whole = params.upload(ctx)
        want = prove_recorded(ctx, whole, mat, zi, za, rs[0], rs[1])
        for name in ("a", "b", "c"):
            assert bytes(getattr(proof, name)) == bytes(getattr(want.raw, name)), "proof point %s differs" % name
        whole.free()
        print("sharded-transform proof over %d ranks is bit-identical to the single-GPU proof" % world, flush=True)
    dist.barrier()
    prover.free()
    mat.free()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
