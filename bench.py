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

# ---- algorithmic work per unit ---------------------------------------------------------------
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
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary measurements (2^24 prove, dense 2^20, stand-alone 2^26 multiexp / transform, cold start)")
    ap.add_argument("--secondary-log-rows", type=int, default=24, help="circuit size of the secondary large prove")
    ap.add_argument("--secondary-log-n", type=int, default=26, help="size of the stand-alone multiexp / transform")
    ap.add_argument("--split-transforms", choices=("auto", "on", "off", "sharded"), default="auto",
                    help="how N > 1 ranks share the transforms: off = every rank repeats them, on = the A/B/C chains on "
                         "three ranks, sharded = every transform over all ranks (2, 4 or 8), a, b, c sharded through H; "
                         "auto: sharded from 2^20 rows on 2/4/8 ranks, chains from 2^22 rows on 3+ ranks, else off")
    return ap.parse_args()


def cpu_model():
    """The host CPU's model string (/proc/cpuinfo), reported beside the core count of the CPU baseline."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


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
    return {"value": tm["total_s"], "unit": "s", "cores": threads, "cpu_model": cpu_model(),
            "host_cpus_online": os.cpu_count(), "kind": "port",
            "sample": "full prove of the same synthetic circuit at 2^%d rows (1/%d of the GPU workload), "
                      "restated bellman algorithms, %d threads; fft %.3fs multiexp %.3fs"
                      % (log_rows, 1, threads, tm["fft_s"], tm["multiexp_s"]),
            "log_rows": log_rows, "constraints_per_s": (1 << log_rows) / tm["total_s"],
            "gpu_proof_identical": bool(same)}


class Dist:
    """What the measurement helpers need from the process group: rank layout, a fence, a max over ranks."""

    def __init__(self, rank, world, backend, local_rank):
        import torch
        self.rank, self.world, self.backend, self.local_rank = rank, world, backend, local_rank
        self.torch = torch
        self.dev = torch.device("cuda", local_rank)

    def fence(self):
        if self.world > 1:
            self.torch.distributed.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, value):
        if self.world == 1:
            return value
        t = self.torch.tensor([value], dtype=self.torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        self.torch.distributed.all_reduce(t, op=self.torch.distributed.ReduceOp.MAX)
        return float(t.item())

    def timed(self, fn, reps, warmup=1):
        """Average seconds per call over `reps` calls after `warmup`, fenced on both sides, max over ranks."""
        for _ in range(warmup):
            fn()
        self.fence()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        self.fence()
        return self.max_over_ranks((time.perf_counter() - t0) / reps)


def rand_fr(rng, count):
    """`count` valid Montgomery representations (below 2^251 < r) as (count, 4) uint64."""
    a = rng.integers(0, 1 << 63, size=(count, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 59) - 1)
    return a


def shared_workload(ctx, d, log_rows, dense):
    """(r1cs, params, z_in, z_aux, seconds spent on key generation) of one synthetic circuit.  With several ranks,
    rank 0 generates circuit, key and witness ONCE and the others map its arrays: N ranks that each generated a
    2^24-row instance would run N GPU key generations, hold N copies of 9 GB of host arrays and share the host's
    cores among N x 16 generator threads.  The arrays travel as .npy files in a RAM-backed directory, mapped
    read-only; the directory is removed once every rank has uploaded (ProveCase)."""
    import shutil
    import torch
    from zinc_amd import workload
    from zinc_amd.groth16 import R1CS, Parameters

    def generate():
        circ = workload.SyntheticCircuit(1 << log_rows, dense=dense)
        r1cs = circ.r1cs()
        t1 = time.time()
        params = workload.generate_parameters(ctx, r1cs, TOXIC)   # zg16_generate_parameters: on the GPU
        keygen_s = time.time() - t1
        zi, za = circ.witness()
        return r1cs, params, zi, za, keygen_s

    if d.world == 1:
        return generate() + (None,)
    names = ("alpha_g1", "beta_g1", "beta_g2", "delta_g1", "delta_g2", "h", "l", "a", "b_g1", "b_g2", "gamma_g2", "ic")
    tag = "zg16_bench_%s_%d_%d" % (os.environ.get("MASTER_PORT", "0"), log_rows, int(dense))
    where = [None]
    if d.rank == 0:
        r1cs, params, zi, za, keygen_s = generate()
        arrays = {"zi": zi, "za": za}
        for k, (rp, col, co) in enumerate(r1cs.mats):
            arrays.update({"rp%d" % k: rp, "col%d" % k: col, "co%d" % k: co})
        arrays.update({n: getattr(params, n) for n in names})
        need = sum(a.nbytes for a in arrays.values())
        for base in (os.environ.get("ZG16_BENCH_SHARE_DIR"), "/dev/shm", "/tmp"):
            try:
                if base and os.path.isdir(base) and shutil.disk_usage(base).free > need + (need >> 3) + (1 << 28):
                    path = os.path.join(base, tag)
                    shutil.rmtree(path, ignore_errors=True)
                    os.makedirs(path)
                    for n, a in arrays.items():
                        np.save(os.path.join(path, n + ".npy"), a)
                    with open(os.path.join(path, "meta.json"), "w") as f:
                        json.dump({"n_inputs": int(r1cs.n_inputs), "n_aux": int(r1cs.n_aux), "keygen_s": keygen_s}, f)
                    where[0] = path
                    break
            except OSError:
                shutil.rmtree(os.path.join(base, tag), ignore_errors=True)
    torch.distributed.broadcast_object_list(where, src=0)
    path = where[0]
    if d.rank == 0:
        return r1cs, params, zi, za, keygen_s, path
    if path is None:   # nowhere to share: every rank generates its own (the seeds are fixed, so they agree)
        return generate() + (None,)
    ld = lambda n: np.load(os.path.join(path, n + ".npy"), mmap_mode="r")
    meta = json.load(open(os.path.join(path, "meta.json")))
    r1cs = R1CS(meta["n_inputs"], meta["n_aux"], [(ld("rp%d" % k), ld("col%d" % k), ld("co%d" % k)) for k in range(3)])
    params = Parameters(*(ld(n) for n in names[:10]))
    params.gamma_g2, params.ic = ld("gamma_g2"), ld("ic")
    return r1cs, params, np.ascontiguousarray(ld("zi")), np.ascontiguousarray(ld("za")), meta["keygen_s"], path


class ProveCase:
    """One synthetic circuit with its key, matrices and witness resident in HBM (this rank's key shard)."""

    def __init__(self, ctx, d, log_rows, dense, split):
        import shutil
        import torch
        from zinc_amd.fields import ints_to_mont, R_MOD
        self.ctx, self.d, self.log_rows, self.dense, self.split = ctx, d, log_rows, dense, split
        t0 = time.time()
        self.r1cs, self.params, self.zi, self.za, self.keygen_s, shared_path = shared_workload(ctx, d, log_rows, dense)
        self.generate_s = time.time() - t0
        try:
            t0 = time.time()
            self.mat = self.r1cs.upload(ctx)
            self.matrices_upload_s = time.time() - t0
            self.sharded = None
            if split == "sharded":
                from zinc_amd.distributed import ShardedProver
                self.sharded = ShardedProver(ctx, self.params, self.mat, d.rank, d.world, d.dev)
                self.key = None
            else:
                self.key = self.params.upload(ctx, d.rank, d.world)
            self.z_in_dev = torch.from_numpy(np.array(self.zi).view(np.int64)).cuda()
            self.z_aux_dev = torch.from_numpy(np.array(self.za).view(np.int64)).cuda()
            torch.cuda.synchronize()
            self.upload_s = time.time() - t0
        finally:
            if d.world > 1:
                torch.distributed.barrier()   # every rank has read what it needs from the shared arrays
                if d.rank == 0 and shared_path:
                    shutil.rmtree(shared_path, ignore_errors=True)
        # full-size r, s as Fr::rand gives
        self.rs = ints_to_mont([pow(7, 123456789, R_MOD), pow(5, 987654321, R_MOD)], R_MOD)
        self.pk_view = self.params.view()

    def step(self):
        return self.step_with(self.z_in_dev.data_ptr(), self.z_aux_dev.data_ptr(), 1)

    def step_with(self, z_in_ptr, z_aux_ptr, on_device):
        from zinc_amd.distributed import prove_sharded
        d = self.d
        if self.sharded:
            return self.sharded.prove(z_in_ptr, z_aux_ptr, on_device, self.rs[0], self.rs[1])
        return prove_sharded(self.ctx, self.key, self.mat, z_in_ptr, z_aux_ptr, on_device,
                             self.pk_view, self.rs[0], self.rs[1], d.rank, d.world, d.dev if d.world > 1 else None,
                             split_transforms=bool(self.split))

    def host_witness(self, reps):
        """The same proof with the witness where every caller of the reference has it -- host memory
        (zinc-vm/src/core/circuit/facade.rs:145-153 hands bellman a synthesizer, the assignment is a Vec<Fr>): the
        PCIe copy of (n_inputs + n_aux) * 32 bytes is inside the timed call.  Pageable memory is what a Vec is;
        page-locked memory is what a host that allocates its witness with hipHostMalloc would have."""
        import torch
        d = self.d
        out = {"witness_bytes": int(self.zi.nbytes + self.za.nbytes)}
        pageable = d.timed(lambda: self.step_with(self.zi.ctypes.data, self.za.ctypes.data, 0), reps, warmup=1)
        out["pageable_ms_per_proof"] = 1e3 * pageable
        try:
            pin_i = torch.from_numpy(self.zi.view(np.int64)).pin_memory()
            pin_a = torch.from_numpy(self.za.view(np.int64)).pin_memory()
            pinned = d.timed(lambda: self.step_with(pin_i.data_ptr(), pin_a.data_ptr(), 0), reps, warmup=1)
            out["pinned_ms_per_proof"] = 1e3 * pinned
        except RuntimeError as e:
            out["pinned_ms_per_proof"] = None
            out["pinned_error"] = str(e)
        resident = d.timed(self.step, reps, warmup=1)
        out["resident_ms_per_proof"] = 1e3 * resident
        out["workload"] = ("2^%d-row proof, witness handed over in host memory (zg16_prove, witness_on_device = 0); "
                           "resident = the same calls with the witness already in HBM" % self.log_rows)
        return out

    def free(self):
        if self.sharded:
            self.sharded.free()
        if self.key:
            self.key.free()
        self.mat.free()
        del self.z_in_dev, self.z_aux_dev


def accumulate_roofline(stats, name="k_accumulate<G1>", fq_mul=FQ_MUL_PER_G1_MADD):
    ms, launches, adds = stats[name]
    if not launches:
        return None
    ach = adds * fq_mul * MAD_PER_FQ_MUL / (ms * 1e-3)
    return {"kernel": name, "bound": "valu", "achieved": ach / 1e12, "peak": VALU_MAD_PEAK / 1e12, "unit": "Tmad/s",
            "frac": ach / VALU_MAD_PEAK, "traffic": None, "avg_launch_ms": ms / launches, "adds_per_launch": adds / launches,
            "frac_of_measured_mad_rate": ach / VALU_MAD_MEASURED}


def secondary_block(ctx, d, args, split_for):
    """Measurements beside the headline, each after its own warm-up and with inputs resident in HBM:
    the large synthetic prove, the dense-witness 2^20 prove, the stand-alone G1 multiexp and Fr transform of the
    stand-alone configuration, and a cold start (key upload + table expansion + matrices + first proof).  Every rank
    takes part; the values are maxima over ranks.  A stage that fails is reported as {"error": ...} and the others
    still run."""
    import torch
    from zinc_amd import workload
    from zinc_amd.distributed import ntt_sharded
    from zinc_amd.groth16 import ResidentBases, _check
    out = {}

    def guarded(name, fn):
        try:
            out[name] = fn()
        except Exception as e:  # noqa: BLE001 -- a secondary stage must not take the headline line down with it
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        if d.world > 1:
            torch.distributed.barrier()

    def keyfile_cold_start(case):
        """The path `zvm prove` takes (zinc-vm/src/zvm/arguments/command/prove.rs:73-88): the proving key arrives as the
        bytes of the key FILE and is read with checked = true -- every point decoded, on the curve, in the subgroup,
        no identity in a query -- once per process; then the matrices, then one proof from a host witness.  On a fresh
        context.  The same key uploaded from Montgomery arrays (no decoding, no checks) is timed beside it: the
        difference is what decoding and checking cost on top of the copy and the table expansion."""
        from zinc_amd.groth16 import DeviceKey
        t0 = time.perf_counter()
        data = workload.params_file_bytes(case.params)
        encode_s = time.perf_counter() - t0
        res = {"workload": "2^%d rows: zg16_pk_upload_bytes(checked = 1) on the %.2f GB key file + zg16_r1cs_upload + first "
                           "proof from a host witness, fresh context" % (case.log_rows, data.nbytes / 1e9),
               "key_file_bytes": int(data.nbytes), "file_encode_s_untimed": encode_s}
        cold = zinc_amd_context(d.local_rank)
        try:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            key = DeviceKey.from_bytes(cold, data, checked=True)
            t1 = time.perf_counter()
            mat = case.r1cs.upload(cold)
            t2 = time.perf_counter()
            from zinc_amd.groth16 import prove_recorded
            proof = prove_recorded(cold, key, mat, case.zi, case.za, case.rs[0], case.rs[1])
            t3 = time.perf_counter()
            key.free()
            res.update({"key_file_decode_check_tables_s": t1 - t0, "matrices_upload_s": t2 - t1, "first_proof_s": t3 - t2,
                        "total_s": t3 - t0})
            t0 = time.perf_counter()
            key = DeviceKey.from_bytes(cold, data, checked=False)
            res["key_file_unchecked_s"] = time.perf_counter() - t0
            key.free()
            t0 = time.perf_counter()
            key = case.params.upload(cold)
            res["same_key_from_montgomery_arrays_s"] = time.perf_counter() - t0
            key.free()
            mat.free()
            res["decode_and_check_share_of_key_load"] = 1.0 - res["same_key_from_montgomery_arrays_s"] / res["key_file_decode_check_tables_s"]
            want = case.step()
            res["proof_identical_to_resident_key_proof"] = bool(want is not None and bytes(proof.raw) == bytes(want))
        finally:
            cold.close()
        return res

    def prove_case(log_rows, dense, reps, host_witness_into=None, keyfile_into=None):
        case = ProveCase(ctx, d, log_rows, dense, split_for(log_rows))
        try:
            case.step()
            ctx.profile(2, reset=True)
            sec = d.timed(case.step, reps, warmup=1)
            stats = ctx.kernel_stats()
            stage = ctx.timings()
            ctx.profile(False, reset=False)
            n_rows = 1 << log_rows
            if host_witness_into:
                out[host_witness_into] = case.host_witness(reps)
            if keyfile_into and d.world == 1:
                out[keyfile_into] = keyfile_cold_start(case)
            return {"workload": "synthetic zinc-shaped R1CS, 2^%d rows (%s witness), full Groth16 prove incl. G2 multiexp"
                                % (log_rows, "dense" if dense else "80% bit"),
                    "ms_per_proof": 1e3 * sec, "constraints_per_s": n_rows / sec, "proofs_timed": reps,
                    "stage_ms_last_proof": stage, "roofline": accumulate_roofline(stats),
                    "generate_s_untimed": case.generate_s, "keygen_on_gpu_s": case.keygen_s, "key_upload_and_tables_s": case.upload_s - case.matrices_upload_s,
                    "matrices_upload_s": case.matrices_upload_s,
                    "split_transforms": case.split or False}
        finally:
            case.free()

    def cold_start():
        """What a process that proves once pays: key expansion into window tables, matrices, first proof (tables of
        the transform included), against the resident-handle latency of the headline."""
        from zinc_amd.groth16 import prove_recorded
        log_rows = min(args.log_rows, 20)
        circ = workload.SyntheticCircuit(1 << log_rows)
        params = workload.generate_parameters(ctx, circ, TOXIC)
        r1cs = circ.r1cs()
        zi, za = circ.witness()
        cold = zinc_amd_context(d.local_rank)  # a fresh context: no cached transform tables, no scratch arenas
        try:
            d.fence()
            t0 = time.perf_counter()
            key = params.upload(cold, d.rank, d.world)
            t1 = time.perf_counter()
            mat = r1cs.upload(cold)
            t2 = time.perf_counter()
            part = workload_first_proof(cold, key, mat, zi, za, params, d)
            t3 = time.perf_counter()
            key.free()
            mat.free()
        finally:
            cold.close()
        return {"workload": "2^%d rows: zg16_pk_upload (window tables) + zg16_r1cs_upload + first proof from host witness" % log_rows,
                "key_upload_and_tables_s": d.max_over_ranks(t1 - t0), "matrices_upload_s": d.max_over_ranks(t2 - t1),
                "first_proof_s": d.max_over_ranks(t3 - t2), "total_s": d.max_over_ranks(t3 - t0)}

    def ntt_standalone():
        log_n = args.secondary_log_n
        n_loc = (1 << log_n) // d.world
        rng = np.random.default_rng(1 + d.rank)
        x = torch.from_numpy(rand_fr(rng, n_loc).view(np.int64)).to(d.dev)
        res = {}
        for inverse in (0, 1):
            if d.world == 1:
                fn = lambda: _check(ctx.handle, ctx._lib.zg16_ntt(ctx.handle, C.c_void_p(x.data_ptr()), log_n, inverse, 0, 1))
            else:
                holder = [x]

                def fn():
                    holder[0] = ntt_sharded(ctx, holder[0], log_n, d.rank, d.world, bool(inverse), d.dev)
            res[inverse] = d.timed(fn, 5, warmup=2)
        nbytes = 64.0 * (1 << log_n)
        return {"workload": "Fr transform of 2^%d uniform elements, natural order in and out%s"
                            % (log_n, "" if d.world == 1 else ", sharded over %d GPUs with one all-to-all" % d.world),
                "fft_ms": 1e3 * res[0], "ifft_ms": 1e3 * res[1], "elements_per_s": (1 << log_n) / res[0],
                "roofline": {"kernel": "k_ntt_pass (whole transform)", "bound": "hbm", "achieved": nbytes / res[0] / 1e9,
                             "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": nbytes / res[0] / HBM_PEAK, "traffic": None,
                             "bytes_per_transform": nbytes}}

    def msm_standalone(log_n):
        n = 1 << log_n
        n_loc = n // d.world
        rng = np.random.default_rng(100 + d.rank)
        g1gen, _ = workload._generators()
        t0 = time.time()
        bases = workload.fixed_base_g1(ctx, g1gen, rand_fr(rng, n_loc))  # this rank's point range: k_i * G
        t1 = time.time()
        res = ResidentBases(ctx, 1, bases, window_tables=True)
        t2 = time.time()
        del bases
        sc = torch.from_numpy(rand_fr(rng, n_loc).view(np.int64)).to(d.dev)
        try:
            def run():
                mine = np.ascontiguousarray(res.msm(device_ptr=sc.data_ptr()), dtype=np.uint64).reshape(1, 8)
                if d.world == 1:
                    return mine
                t = torch.from_numpy(mine.view(np.int64))
                t = t.to(d.dev) if d.backend == "nccl" else t
                outs = [torch.empty_like(t) for _ in range(d.world)]
                torch.distributed.all_gather(outs, t)
                return outs
            run()
            ctx.profile(2, reset=True)
            sec = d.timed(run, 3 if log_n >= 26 else 10, warmup=0)
            stats = ctx.kernel_stats()
            ctx.profile(False, reset=False)
        finally:
            res.free()
        return {"workload": "G1 multiexp, 2^%d points k_i*G resident as window tables, uniform scalars%s"
                            % (log_n, "" if d.world == 1 else ", point ranges over %d GPUs, affine partial sums all-gathered" % d.world),
                "seconds": sec, "points_per_s": n / sec, "bases_s_untimed": t1 - t0, "upload_and_tables_s_untimed": t2 - t1,
                "roofline": accumulate_roofline(stats)}

    def small_circuit(circ, label):
        """The small-circuit latency floor: a hand-recorded stand-in (zinc_amd/standin.py; a few thousand rows -- the
        size of a real zinc-tester program or contract method), key generated on the GPU, proof checked by the
        library's verifier against the public inputs."""
        from zinc_amd import standin
        from zinc_amd.groth16 import prove_recorded, verify_proof
        from zinc_amd.fields import ints_to_mont, R_MOD
        pa = standin.record(circ)
        r1cs = pa.r1cs()
        t0 = time.perf_counter()
        params = workload.generate_parameters(ctx, r1cs, TOXIC)
        keygen_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        key = params.upload(ctx)
        key_upload_s = time.perf_counter() - t0
        mat = r1cs.upload(ctx)
        zi, za = pa.witness()
        rs = ints_to_mont([pow(7, 123456789, R_MOD), pow(5, 987654321, R_MOD)], R_MOD)
        try:
            proof = prove_recorded(ctx, key, mat, zi, za, rs[0], rs[1])
            sec = d.timed(lambda: prove_recorded(ctx, key, mat, zi, za, rs[0], rs[1]), 50, warmup=3)
            t0 = time.perf_counter()
            ok = verify_proof(params, proof, circ.public_inputs())
            verify_s = time.perf_counter() - t0
        finally:
            key.free()
            mat.free()
        return {"workload": "%s, %d rows, witness from host memory" % (label, r1cs.n_rows),
                "ms_per_proof": 1e3 * sec, "proofs_timed": 50, "verifies_against_public_inputs": bool(ok),
                "public_inputs": [str(v) for v in circ.public_inputs()],
                "verify_ms_host": 1e3 * verify_s, "keygen_s": keygen_s, "key_upload_and_tables_s": key_upload_s}

    guarded("prove_2p%d" % args.secondary_log_rows,
            lambda: prove_case(args.secondary_log_rows, False, 3, "prove_2p%d_host_witness" % args.secondary_log_rows,
                               "cold_start_keyfile_2p%d" % args.secondary_log_rows))
    if d.world == 1:
        from zinc_amd import standin
        guarded("small_circuit_factorial", lambda: small_circuit(
            standin.FactorialCircuit(6), "hand-recorded stand-in for zinc-tester/tests/algorithm/factorial.zn (n = 6 -> 720, the "
                                         "output the reference's test pins)"))
        guarded("small_circuit_uniswap", lambda: small_circuit(
            standin.UniswapInputPriceCircuit(10 ** 18, 5 * 10 ** 21, 7 * 10 ** 23, 0xC0FFEE << 200),
            "hand-recorded stand-in for get_eth_to_token_input_price of zinc-examples/uniswap/src/main.zn:705-712 (a contract "
            "method: price and storage root are the public inputs; the reference pins no output for it)"))
    guarded("prove_2p%d_dense" % min(args.log_rows, 20),
            lambda: prove_case(min(args.log_rows, 20), True, 10, None, "cold_start_keyfile_2p%d" % min(args.log_rows, 20)))
    # the metric's second half: G1 multiexp points per second at the two headline sizes, then the stand-alone size
    for log_n in sorted({min(args.log_rows, 20), args.secondary_log_rows, args.secondary_log_n}):
        guarded("msm_g1_2p%d" % log_n, lambda log_n=log_n: msm_standalone(log_n))
    guarded("fr_ntt_2p%d" % args.secondary_log_n, ntt_standalone)
    guarded("cold_start", cold_start)
    return out


def zinc_amd_context(device):
    import zinc_amd
    return zinc_amd.Context(device)


def workload_first_proof(ctx, key, mat, zi, za, params, d):
    """One proof from a host-resident witness on a context that has never proved (rank 0 assembles)."""
    from zinc_amd.distributed import prove_sharded
    from zinc_amd.fields import ints_to_mont, R_MOD
    rs = ints_to_mont([3, 5], R_MOD)
    return prove_sharded(ctx, key, mat, zi.ctypes.data, za.ctypes.data, 0, params.view(), rs[0], rs[1], d.rank, d.world,
                         d.dev if d.world > 1 else None)


def launch_ranks(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as fresh child processes (before
    this process has touched the GPU) and exit with their status."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:  # nothing has touched the GPU yet
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run --nproc-per-node %d "
                 "(or run `python bench.py --gpus %d` without WORLD_SIZE set)" % (args.gpus, world, args.gpus, args.gpus))
    # one rank per GPU over RCCL; ZG16_DIST_BACKEND=gloo rehearses the same code on a one-GPU box (every
    # rank on GPU 0, exchanges staged through the host)
    backend = os.environ.get("ZG16_DIST_BACKEND", "nccl")
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    if world > 1 and backend == "nccl":
        # a rank runs the context's four streams AND the collectives' stream: with the default of four hardware
        # queues per process two of them share a queue and the later one waits behind the earlier one's kernels
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import zinc_amd
    d = Dist(rank, world, backend, local_rank)
    ctx = zinc_amd.Context(local_rank)
    ctx.graph({"auto": None, "on": True, "off": False}[args.graph])
    n_rows = 1 << args.log_rows

    sharded_ok = [True]  # cleared when the headline's sharded proof fails its check below

    def split_for(log_rows):
        """False (every rank repeats the transforms), True (A/B/C chains on three ranks) or "sharded"."""
        if world == 1 or args.split_transforms == "off":
            return False
        can_shard = sharded_ok[0] and world in (2, 4, 8) and log_rows >= 2 * (world.bit_length() - 1) + 1
        if args.split_transforms == "sharded":
            return "sharded" if can_shard else False
        if args.split_transforms == "on":
            return True
        if can_shard and log_rows >= 20:
            return "sharded"
        return world >= 3 and log_rows >= 22

    t0 = time.time()
    case = ProveCase(ctx, d, args.log_rows, args.dense, split_for(args.log_rows))
    schedule_fallback = None
    if world > 1 and case.split == "sharded":
        # The sharded schedule's exchanges have only ever run between contexts of one GPU: prove once and check the
        # proof with the verifier before anything is timed; a rank that fails (or a proof that does not verify) sends
        # every rank back to the replicated schedule, and the line says so.
        ok, why = 1.0, ""
        try:
            if os.environ.get("ZG16_BENCH_FAIL_SHARDED"):  # rehearsal of the way back
                raise RuntimeError("ZG16_BENCH_FAIL_SHARDED is set")
            proof = case.step()
            if rank == 0:
                from zinc_amd.fields import mont_to_ints, R_MOD
                from zinc_amd.groth16 import verify_proof
                ok = 1.0 if verify_proof(case.params, proof, mont_to_ints(np.asarray(case.zi)[1:], R_MOD)) else 0.0
                why = "" if ok else "the sharded proof did not verify"
        except Exception as e:  # noqa: BLE001
            ok, why = 0.0, "%s: %s" % (type(e).__name__, e)
        if d.max_over_ranks(1.0 - ok) > 0.0:
            schedule_fallback = why or "another rank failed"
            sharded_ok[0] = False
            case.free()
            case = ProveCase(ctx, d, args.log_rows, args.dense, False)
    setup_s = time.time() - t0
    params, r1cs, key, mat, zi, za = case.params, case.r1cs, case.key, case.mat, case.zi, case.za
    split = case.split
    dens = mat.density()
    step, fence = case.step, d.fence

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
    elapsed = d.max_over_ranks(elapsed)
    out = None

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        # dominant kernel: the largest uncontended device time per step (the overlapped timings count shared
        # time once per kernel that was resident)
        dom = max((stats_serial or stats).items(), key=lambda kv: kv[1][0])
        per_kernel = {k: {"ms_per_step": v[0] / table_steps, "launches_per_step": v[1] / table_steps}
                      for k, v in stats_all.items() if v[1]}

        pmc = {}
        traffic_from = None
        # the newest committed counter passes of this same command (profiles/rNN_pmc_hbm_2pK.json)
        import glob
        pmc_files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                                                  "r*_pmc_hbm_2p%d.json" % args.log_rows)))
        if pmc_files:
            pmc = json.load(open(pmc_files[-1]))["kernels"]
            traffic_from = ("profiles/%s: rocprofv3 --pmc passes of this command on an earlier box (FETCH_SIZE and "
                            "WRITE_SIZE in separate runs), not counters of this run" % os.path.basename(pmc_files[-1]))

        def traffic(pmc_name, streaming):
            """HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command
            (FETCH_SIZE doubled for streaming reads as gfx950 needs; raw for gathers), or None."""
            names = pmc_name if isinstance(pmc_name, tuple) else (pmc_name,)   # a timer that brackets several kernels
            if not all(n in pmc for n in names):
                return None
            return sum((pmc[n]["fetch_bytes_per_launch"] if streaming else pmc[n]["fetch_bytes_per_launch_raw"])
                       + pmc[n]["write_bytes_per_launch"] for n in names)

        def valu_roofline(name, fq_mul_per_add, pmc_name, stats=stats):
            ms, launches, adds = stats[name]
            if not launches:
                return None
            ach = adds * fq_mul_per_add * MAD_PER_FQ_MUL / (ms * 1e-3)
            return {"kernel": name, "bound": "valu", "achieved": ach / 1e12, "peak": VALU_MAD_PEAK / 1e12,
                    "unit": "Tmad/s", "frac": ach / VALU_MAD_PEAK, "traffic": traffic(pmc_name, False),
                    "traffic_from": traffic_from,
                    "avg_launch_ms": ms / launches, "adds_per_launch": adds / launches,
                    "frac_of_measured_mad_rate": ach / VALU_MAD_MEASURED}

        def hbm_roofline(name, pmc_name, fr_mul_per_byte=None, stats=stats, streaming=True):
            ms, launches, nbytes = stats[name]
            if not launches:
                return None
            ach = nbytes / (ms * 1e-3)
            r = {"kernel": name, "bound": "hbm", "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                 "frac": ach / HBM_PEAK, "traffic": traffic(pmc_name, streaming), "traffic_from": traffic_from,
                 "avg_launch_ms": ms / launches,
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
                # (the kernel a proof runs for the short rows is k_spmv_ordered; k_spmv_short only serves sharded row blocks)
                "k_spmv_short+long": hbm_roofline("k_spmv_short+long", ("k_spmv_ordered", "k_spmv_long"), None, st, streaming=False),  # gathers: raw FETCH_SIZE
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
                       "n_inputs": int(r1cs.n_inputs), "n_aux": int(r1cs.n_aux), "nnz": [int(m[1].shape[0]) for m in r1cs.mats],
                       "a_aux_density": dens[0], "b_aux_density": dens[2], "parallelism": "msm-point-range x%d%s" % (world, ", transforms sharded over all ranks" if split == "sharded" else ", A/B/C transform chains on 3 ranks" if split else ""),
                       **({"schedule_fallback": "sharded schedule abandoned for the replicated one: " + schedule_fallback} if schedule_fallback else {})},
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
            out["msm_g1_accumulate_adds_per_s"] = g1[2] / (g1[0] * 1e-3)
        if world == 1 and not args.no_cpu_baseline:
            cpu_rows = min(args.cpu_log_rows, args.log_rows)
            args.cpu_log_rows = cpu_rows
            reuse = (params, r1cs, zi, za, key, mat) if cpu_rows == args.log_rows else None
            out["cpu_baseline"] = cpu_baseline_leg(ctx, cpu_rows, args.dense, reuse)
            out["cpu_baseline"]["gpu_over_cpu_latency"] = (out["cpu_baseline"]["value"] / (ms_per_step / 1e3)
                                                           if cpu_rows == args.log_rows else None)
            out["cpu_baseline"]["sample"] = out["cpu_baseline"]["sample"].replace(
                "(1/1 of", "(1/%d of" % (1 << (args.log_rows - args.cpu_log_rows)))
    host_witness = None
    if not args.no_secondary:
        try:
            host_witness = case.host_witness(max(3, min(args.steps, 10)))
        except Exception as e:  # noqa: BLE001 -- must not take the headline line down
            host_witness = {"error": "%s: %s" % (type(e).__name__, e)}
    case.free()
    del case, key, mat
    if not args.no_secondary:
        sec = secondary_block(ctx, d, args, split_for)
        if rank == 0:
            sec["prove_2p%d_host_witness" % args.log_rows] = host_witness
            out["secondary"] = sec
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
