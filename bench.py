#!/usr/bin/env python3
"""Groth16 prove latency on MI355X: synthetic zinc-shaped R1CS, full prove through the C ABI.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log-rows 20] [--dense]

One "step" is one complete proof (row evaluation, 7 NTTs, the multiexps, assembly) of the same
circuit and witness, which are resident in HBM before the timed region starts; the proving key
and constraint matrices are uploaded once, as a prover service would keep them.  With N > 1
(launched by torch.distributed.run, one rank per GPU) every rank holds 1/N of each key query,
computes the partial multiexp sums of its point range, and rank 0 assembles the proof from the
partials gathered over RCCL; the reported time is the max over ranks ("strong" scaling: the
proof is the same size whatever N is).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# ---- algorithmic work per unit (stated in DESIGN.md) ------------------------------------------
MAD_PER_FQ_MUL = 136               # 8x8 + 8x8 + 8 32-bit multiply-adds in one CIOS product
FQ_MUL_PER_G1_MADD = 10            # XYZZ mixed addition: 8M + 2S
FQ_MUL_PER_G2_MADD = 8 * 3 + 2 * 2  # Fq2 mul = 3 Fq mul (Karatsuba), Fq2 sqr = 2
VALU_MAD_PEAK = 256 * 4 * 16 * 2.4e9  # v_mad_u64_u32: 16 lanes/clk/SIMD x 1024 SIMDs x 2.4 GHz
VALU_MAD_MEASURED = 26.0e12         # sustained v_mad_u64_u32 rate of tools/microbench.hip on MI355X
HBM_PEAK = 8.0e12

TOXIC = dict(k1=0x1234567, k2=0x7654321, alpha=0xA11CE5, beta=0xBEEF77, gamma=0x6A33A, delta=0xDE17A, tau=0x7A0F00D)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-rows", type=int, default=20)
    ap.add_argument("--dense", action="store_true", help="all aux values uniform (multiexp worst case)")
    ap.add_argument("--cpu-log-rows", type=int, default=20, help="circuit size of the CPU-baseline sample (2^20: seconds of CPU work on all host cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true",
                    help="leave the per-kernel HIP-event timers off (no roofline in the output): shows what they cost")
    ap.add_argument("--graph", choices=("auto", "on", "off"), default="auto",
                    help="HIP-graph replay of the proof (auto: off unless ZG16_GRAPH=1); only proofs without kernel timers replay")
    ap.add_argument("--split-transforms", choices=("auto", "on", "off"), default="auto",
                    help="run the A/B/C transform chains on different ranks (auto: 3+ ranks and 2^22+ rows)")
    return ap.parse_args()


def cpu_baseline_leg(ctx, log_rows, dense, reuse=None):
    """Time the CPU restatement of the bellman prover on a bounded sample of the same workload and
    check that the GPU produces the identical proof for it.  The only place this file touches oracle/.
    reuse: (params, r1cs, z_in, z_aux, key, mat) of the benchmark itself when the sample is the whole workload."""
    from oracle import cpu_binding as cb
    from zinc_amd import workload
    from zinc_amd.groth16 import prove_recorded
    from zinc_amd.fields import ints_to_mont, R_MOD
    if reuse:
        params, r1cs, zi, za, key, mat = reuse
    else:
        circ = workload.SyntheticCircuit(1 << log_rows, dense=dense)
        params = workload.generate_parameters(ctx, circ, TOXIC)
        r1cs = circ.r1cs()
        zi, za = circ.witness()
        key, mat = params.upload(ctx), r1cs.upload(ctx)
    rs = ints_to_mont([pow(7, 123456789, R_MOD), pow(5, 987654321, R_MOD)], R_MOD)
    threads = min(cb.hardware_threads() or 1, int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or 1 << 30)
    st, cpu_proof, tm = cb.prove(params, r1cs, zi, za, rs[0], rs[1], threads)
    assert st == 0, "CPU baseline prover failed with status %d" % st
    gpu = prove_recorded(ctx, key, mat, zi, za, rs[0], rs[1])
    if not reuse:
        key.free()
        mat.free()
    same = all(np.array_equal(np.frombuffer(bytes(g), dtype=np.uint64), c)
               for g, c in zip((gpu.raw.a, gpu.raw.b, gpu.raw.c), cpu_proof))
    return {"value": tm["total_s"], "unit": "s", "cores": threads, "kind": "port",
            "sample": "full prove of the same synthetic circuit at 2^%d rows (1/%d of the GPU workload), "
                      "restated bellman algorithms, %d threads; fft %.3fs multiexp %.3fs"
                      % (log_rows, 1, threads, tm["fft_s"], tm["multiexp_s"]),
            "log_rows": log_rows, "constraints_per_s": (1 << log_rows) / tm["total_s"],
            "gpu_proof_identical": bool(same)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # one rank per GPU over RCCL; ZG16_DIST_BACKEND=gloo rehearses the same code on a one-GPU box (every
    # rank on GPU 0, exchanges staged through the host)
    backend = os.environ.get("ZG16_DIST_BACKEND", "nccl")
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    import zinc_amd
    from zinc_amd import workload, lib as zlib
    from zinc_amd.fields import ints_to_mont, R_MOD
    from zinc_amd.groth16 import _check

    ctx = zinc_amd.Context(local_rank)
    ctx.graph({"auto": None, "on": True, "off": False}[args.graph])
    L = ctx._lib
    n_rows = 1 << args.log_rows
    t0 = time.time()
    circ = workload.SyntheticCircuit(n_rows, dense=args.dense)
    params = workload.generate_parameters(ctx, circ, TOXIC)
    r1cs = circ.r1cs()
    key = params.upload(ctx, rank, world)
    mat = r1cs.upload(ctx)
    zi, za = circ.witness()
    z_in_dev = torch.from_numpy(zi.view(np.int64)).cuda()
    z_aux_dev = torch.from_numpy(za.view(np.int64)).cuda()
    rs = ints_to_mont([pow(7, 123456789, R_MOD), pow(5, 987654321, R_MOD)], R_MOD)  # full-size r, s as Fr::rand gives
    setup_s = time.time() - t0
    pk_view = params.view()
    dens = mat.density()

    from zinc_amd.distributed import prove_sharded
    dev = torch.device("cuda", local_rank) if world > 1 else None

    split = args.split_transforms == "on" or (args.split_transforms == "auto" and world >= 3 and args.log_rows >= 22)
    split = split and world > 1

    def step():
        return prove_sharded(ctx, key, mat, z_in_dev.data_ptr(), z_aux_dev.data_ptr(), 1, pk_view, rs[0], rs[1],
                             rank, world, dev, split_transforms=split)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # timed steps carry HIP-event timers on the dominant kernel (k_accumulate) only; the full per-kernel table
    # comes from extra untimed steps below (timing every launch costs about 5 % of a step)
    ctx.profile(0 if args.no_kernel_timers else 2, reset=True)
    fence()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t_start
    stats = ctx.kernel_stats()
    stage = ctx.timings()
    table_steps = 0 if args.no_kernel_timers else 3
    ctx.profile(bool(table_steps), reset=True)
    for _ in range(table_steps):
        step()
    fence()
    stats_all = ctx.kernel_stats()
    # the same with every stage on one stream: per-kernel durations free of the contention between the
    # overlapped pipelines (the step itself is slower that way)
    stats_serial = {}
    if table_steps and world == 1:
        ctx.serial(True)
        step()
        ctx.profile(True, reset=True)
        for _ in range(table_steps):
            step()
        fence()
        stats_serial = ctx.kernel_stats()
        ctx.serial(False)
    ctx.profile(False, reset=False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        # dominant kernel: the largest uncontended device time per step (the overlapped timings count shared
        # time once per kernel that was resident)
        dom = max((stats_serial or stats).items(), key=lambda kv: kv[1][0])
        per_kernel = {k: {"ms_per_step": v[0] / table_steps, "launches_per_step": v[1] / table_steps}
                      for k, v in stats_all.items() if v[1]}

        pmc = {}
        pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_hbm_2p%d.json" % args.log_rows)
        if os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path))["kernels"]

        def traffic(pmc_name, streaming):
            """HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command
            (FETCH_SIZE doubled for streaming reads as gfx950 needs; raw for gathers), or None."""
            k = pmc.get(pmc_name)
            if not k:
                return None
            return (k["fetch_bytes_per_launch"] if streaming else k["fetch_bytes_per_launch_raw"]) + k["write_bytes_per_launch"]

        def valu_roofline(name, fq_mul_per_add, pmc_name, stats=stats):
            ms, launches, adds = stats[name]
            if not launches:
                return None
            ach = adds * fq_mul_per_add * MAD_PER_FQ_MUL / (ms * 1e-3)
            return {"kernel": name, "bound": "valu", "achieved": ach / 1e12, "peak": VALU_MAD_PEAK / 1e12,
                    "unit": "Tmad/s", "frac": ach / VALU_MAD_PEAK, "traffic": traffic(pmc_name, False),
                    "avg_launch_ms": ms / launches, "adds_per_launch": adds / launches,
                    "frac_of_measured_mad_rate": ach / VALU_MAD_MEASURED}

        def hbm_roofline(name, pmc_name, fr_mul_per_byte=None, stats=stats):
            ms, launches, nbytes = stats[name]
            if not launches:
                return None
            ach = nbytes / (ms * 1e-3)
            r = {"kernel": name, "bound": "hbm", "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                 "frac": ach / HBM_PEAK, "traffic": traffic(pmc_name, True), "avg_launch_ms": ms / launches,
                 "bytes_per_launch": nbytes / launches}
            if fr_mul_per_byte:
                # the transform is bounded by field products, not bytes: the same launches against the VALU peak
                mads = nbytes * fr_mul_per_byte * MAD_PER_FQ_MUL / (ms * 1e-3)
                r["valu_achieved_Tmad_s"] = mads / 1e12
                r["valu_frac"] = mads / VALU_MAD_PEAK
            return r

        # k_ntt_pass: a radix-2^10 pass spends 4.25 butterfly products (ten stages, half a product per element each,
        # less the trivial twiddles of the first two) + 1 factor product per element = 5.25 per 64 bytes
        def all_rooflines(st):
            return {
                "k_accumulate<G1>": valu_roofline("k_accumulate<G1>", FQ_MUL_PER_G1_MADD, "k_accumulate<Fq >", st),
                "k_accumulate<G2>": valu_roofline("k_accumulate<G2>", FQ_MUL_PER_G2_MADD, "k_accumulate<Fq2>", st),
                "k_ntt_pass": hbm_roofline("k_ntt_pass", "k_ntt_pass", 5.25 / 64.0, st),
                "k_spmv_short+long": hbm_roofline("k_spmv_short+long", "k_spmv_short", None, st),
            }

        # live (HIP events over the timed steps, pipelines overlapped) for the dominant kernel kind, k_accumulate;
        # the transform and row-evaluation figures come from the extra untimed steps that time every launch
        rooflines = all_rooflines(stats)
        if table_steps:
            extra = all_rooflines(stats_all)
            for k in ("k_ntt_pass", "k_spmv_short+long"):
                rooflines[k] = extra[k]
                if rooflines[k]:
                    rooflines[k]["from"] = "%d extra untimed steps" % table_steps
        roofline = rooflines.get(dom[0]) if dom[0].startswith("k_accumulate") else None
        roofline = roofline or rooflines["k_accumulate<G1>"]
        if stats_serial:
            un = all_rooflines(stats_serial)
            for k, v in rooflines.items():
                if v and un.get(k):
                    v["uncontended"] = {f: un[k][f] for f in ("achieved", "frac", "avg_launch_ms", "frac_of_measured_mad_rate",
                                                              "valu_achieved_Tmad_s", "valu_frac") if f in un[k]}
        out = {
            "metric": "groth16_prove_latency_2^%d_constraints" % args.log_rows,
            "value": ms_per_step / 1e3,
            "unit": "s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": False,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u32x8 Montgomery (BN254 Fr/Fq, integer)",
            "data": "synthetic",
            "config": {"workload": "synthetic zinc-shaped R1CS, 2^%d rows (%s witness), full Groth16 prove incl. G2 multiexp"
                                   % (args.log_rows, "dense" if args.dense else "80% bit"),
                       "n_inputs": circ.n_inputs, "n_aux": circ.n_aux, "nnz": [int(m[1].shape[0]) for m in r1cs.mats],
                       "a_aux_density": dens[0], "b_aux_density": dens[2], "parallelism": "msm-point-range x%d%s" % (world, ", A/B/C transform chains on 3 ranks" if split else "")},
            "constraints_per_s": n_rows / (ms_per_step / 1e3),
            "stage_ms_last_step": stage,
            "kernels": per_kernel,
            "kernels_from": "%d extra untimed steps with every launch timed" % table_steps,
            "kernels_uncontended": {k: {"ms_per_step": v[0] / table_steps, "launches_per_step": v[1] / table_steps}
                                    for k, v in stats_serial.items() if v[1]},
            "roofline": roofline,
            "roofline_other": {k: v for k, v in rooflines.items() if v and v is not roofline},
            "setup_s_untimed": setup_s,
        }
        g1 = stats["k_accumulate<G1>"]
        if g1[0]:
            out["msm_g1_accumulate_points_per_s"] = g1[2] / (g1[0] * 1e-3)
        if world == 1 and not args.no_cpu_baseline:
            cpu_rows = min(args.cpu_log_rows, args.log_rows)
            args.cpu_log_rows = cpu_rows
            reuse = (params, r1cs, zi, za, key, mat) if cpu_rows == args.log_rows else None
            out["cpu_baseline"] = cpu_baseline_leg(ctx, cpu_rows, args.dense, reuse)
            out["cpu_baseline"]["gpu_over_cpu_latency"] = (out["cpu_baseline"]["value"] / (ms_per_step / 1e3)
                                                           if cpu_rows == args.log_rows else None)
            out["cpu_baseline"]["sample"] = out["cpu_baseline"]["sample"].replace(
                "(1/1 of", "(1/%d of" % (1 << (args.log_rows - args.cpu_log_rows)))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
