"""One rank of the multi-process GPU test of the sharded transform (see test_gpu_parity.py).  All
ranks share GPU 0 and talk over gloo, which stands in for RCCL on a one-GPU box."""
import sys

import numpy as np
import torch
import torch.distributed as dist

import zinc_amd
from zinc_amd.distributed import ntt_sharded, ntt_shard_input, ntt_shard_output_index


def main():
    rank, world, port, log_n = (int(x) for x in sys.argv[1:5])
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    torch.cuda.set_device(0)
    ctx = zinc_amd.Context(0)
    n = 1 << log_n
    rng = np.random.default_rng(2024)
    x = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    x[:, 3] >>= np.uint64(3)
    whole = ctx.ntt(x)
    dev = torch.device("cuda", 0)
    part = torch.from_numpy(np.ascontiguousarray(ntt_shard_input(x, rank, world)).view(np.int64)).to(dev)
    fwd = ntt_sharded(ctx, part, log_n, rank, world, False, dev)
    assert np.array_equal(fwd.cpu().numpy().view(np.uint64), whole[ntt_shard_output_index(n, rank, world)]), "forward differs"
    back = ntt_sharded(ctx, fwd, log_n, rank, world, True, dev)
    assert np.array_equal(back.cpu().numpy().view(np.uint64), ntt_shard_input(x, rank, world)), "inverse differs"
    dist.barrier()
    if rank == 0:
        print("sharded transform over %d ranks equals the whole-vector transform" % world, flush=True)
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
