/* C ABI of libzinc_groth16_hip.so -- an MI355X (gfx950) Groth16 prover backend for zinc-vm.
 *
 * What it replaces.  zinc-vm proves through one generic Rust call,
 *     franklin_crypto::bellman::groth16::create_random_proof(circuit, &params, rng)
 * (zinc-vm/src/core/circuit/facade.rs:153, zinc-vm/src/core/contract/facade.rs:387), whose
 * callee side is `Circuit::synthesize` (zinc-vm/src/core/circuit/synthesizer.rs:33-38).  The
 * Rust side keeps a recording ConstraintSystem (the role bellman's ProvingAssignment plays)
 * and hands what it recorded to the entry points below; everything after synthesis -- row
 * evaluation, the seven FFTs, the eight multiexps, final assembly -- runs on the GPU.
 *
 * Conventions
 *  - Field elements: 4 x u64 little-endian limbs, Montgomery form with R = 2^256 (the in-memory
 *    layout of ff_ce's Fr/Fq, so a Rust &[Fr] is passed as-is).  32 bytes each.
 *  - G1 affine point: x || y (64 bytes).  G2 affine point: x.c0 || x.c1 || y.c0 || y.c1
 *    (128 bytes).  All-zero bytes encode the point at infinity.
 *  - A variable index is a u32: bit 31 set = aux variable, clear = input variable; input 0 is
 *    the constant ONE (zinc-vm/src/constraint_systems/main.rs:31).
 *  - Every function returns a zg16_status; no exceptions or aborts cross this boundary.
 *    Status values mirror the bellman SynthesisError variants that can occur after synthesis
 *    (mapped by the caller at zinc-vm/src/core/circuit/facade.rs:154).
 *  - Entry points block until the result is on the host.  A context is used by one thread at a
 *    time.  The caller owns every host buffer; the library copies what it needs.
 */
#ifndef ZINC_GROTH16_H
#define ZINC_GROTH16_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  ZG16_OK = 0,
  ZG16_ERR_POLYNOMIAL_DEGREE_TOO_LARGE = 1, /* domain would need 2^28 or more rows            */
  ZG16_ERR_UNEXPECTED_IDENTITY = 2,         /* identity base consumed, or delta at infinity   */
  ZG16_ERR_ASSIGNMENT_MISSING = 3,          /* a witness vector is NULL or too short          */
  ZG16_ERR_MALFORMED_KEY = 4,               /* key vector lengths do not fit the circuit      */
  ZG16_ERR_INVALID_ARGUMENT = 5,
  ZG16_ERR_DEVICE = 6,                      /* a HIP call failed; see zg16_last_error         */
  ZG16_ERR_OUT_OF_MEMORY = 7,
  ZG16_ERR_UNCONSTRAINED_VARIABLE = 8       /* key generation: an aux variable occurs in no row */
} zg16_status;

typedef struct zg16_ctx zg16_ctx;   /* one GPU, its streams and scratch memory */
typedef struct zg16_pk zg16_pk;     /* a proving key resident in HBM           */
typedef struct zg16_r1cs zg16_r1cs; /* recorded constraint matrices in HBM     */

/* Host view of bellman's `Parameters<Bn256>` (what `Parameters::read` yields at
 * zinc-vm/src/zvm/arguments/command/prove.rs:73-76).  a, b_g1, b_g2 hold points only for the
 * variables whose polynomial is non-zero, inputs first (upstream drops identities at setup). */
typedef struct {
  const uint64_t* alpha_g1; /* 8  limbs */
  const uint64_t* beta_g1;  /* 8  */
  const uint64_t* beta_g2;  /* 16 */
  const uint64_t* delta_g1; /* 8  */
  const uint64_t* delta_g2; /* 16 */
  const uint64_t* h;    size_t n_h;    /* G1, m - 1 points          */
  const uint64_t* l;    size_t n_l;    /* G1, one per aux variable  */
  const uint64_t* a;    size_t n_a;    /* G1                        */
  const uint64_t* b_g1; size_t n_b_g1; /* G1                        */
  const uint64_t* b_g2; size_t n_b_g2; /* G2                        */
} zg16_pk_view;

/* One constraint matrix in CSR form.  Rows are in enforce() order and must already include the
 * `input_i * 0 = 0` rows bellman appends after synthesis (A has the single term input_i there).
 * Within a row: no repeated variable and no zero coefficient -- the canonical form produced by
 * zinc-vm/src/constraint_systems/dedup.rs:50-68. */
typedef struct {
  const uint64_t* row_ptr; /* n_rows + 1 offsets into col/coeff          */
  const uint32_t* col;     /* variable index, bit 31 = aux               */
  const uint64_t* coeff;   /* 4 limbs per entry, Montgomery Fr           */
} zg16_csr;

typedef struct {
  size_t n_rows;
  size_t n_inputs; /* including ONE */
  size_t n_aux;
  zg16_csr a, b, c;
} zg16_r1cs_view;

/* Groth16 proof as three affine points (bellman `Proof<Bn256>`, returned by value at
 * zinc-vm/src/core/circuit/facade.rs:153). */
typedef struct {
  uint64_t a[8];  /* G1 */
  uint64_t b[16]; /* G2 */
  uint64_t c[8];  /* G1 */
} zg16_proof;

/* Partial result of one GPU's share of a proof: the multiexp sums over that GPU's point range,
 * in extended Jacobian XYZZ coordinates (Montgomery limbs).  Partials from all shards are summed
 * coordinate-free by zg16_prove_finish, so they can be gathered with any byte-moving collective. */
typedef struct {
  uint64_t hl[16];   /* G1: h-query + l-query   */
  uint64_t a[16];    /* G1: a-query             */
  uint64_t b1[16];   /* G1: b_g1-query          */
  uint64_t b2[32];   /* G2: b_g2-query          */
} zg16_partial;

/* ---- context ------------------------------------------------------------------------------ */
zg16_status zg16_ctx_create(int device_id, zg16_ctx** out);
void        zg16_ctx_destroy(zg16_ctx* ctx);
/* Text of the most recent failure on this context (never NULL). */
const char* zg16_last_error(const zg16_ctx* ctx);
/* Make the context issue its work on a caller-owned hipStream_t (NULL = its own stream). */
zg16_status zg16_ctx_set_stream(zg16_ctx* ctx, void* hip_stream);
/* The hipStream_t the context issues its main-stream work on (its own unless zg16_ctx_set_stream changed it), so that
 * a caller can queue its own work -- an RCCL collective between two stages of a sharded proof -- in order with the
 * library's WITHOUT bringing a stream of its own: a GPU process has few hardware queues (4 by default), the context
 * already uses four streams, and a fifth stream shares a queue with one of them and serialises behind it. */
zg16_status zg16_ctx_get_stream(const zg16_ctx* ctx, void** hip_stream);

/* ---- proving key -------------------------------------------------------------------------- */
/* Upload the slice [shard_index/shard_count] of every query (contiguous point ranges) so that
 * `shard_count` GPUs each hold 1/shard_count of the key.  shard_count = 1 uploads all of it. */
zg16_status zg16_pk_upload(zg16_ctx* ctx, const zg16_pk_view* view, int shard_index,
                           int shard_count, zg16_pk** out);
void        zg16_pk_free(zg16_pk* pk);
/* The same from the bytes `Parameters::<Bn256>::write` produces -- the proving-key file that
 * `zvm prove` opens with `Parameters::read(file, checked = true)`
 * (zinc-vm/src/zvm/arguments/command/prove.rs:73-76; written at setup.rs:67-69).  The points are decoded,
 * converted to Montgomery form and (with `checked`) validated on the GPU: on the curve, in the prime-order
 * subgroup (G2: psi(P) = [6x^2] P with psi the untwist-Frobenius-twist map, which holds exactly when r P = O;
 * checked = 2 runs the plain multiplication by r instead, the slower form kept as its checker), no point at
 * infinity in a query vector.  Any violation or a
 * truncated file gives ZG16_ERR_MALFORMED_KEY.  zg16_pk_vk_view fills a view with the handle's
 * verifying-key points (host memory owned by the handle) for zg16_prove_finish. */
zg16_status zg16_pk_upload_bytes(zg16_ctx* ctx, const uint8_t* data, size_t len, int checked,
                                 int shard_index, int shard_count, zg16_pk** out);
zg16_status zg16_pk_vk_view(const zg16_pk* pk, zg16_pk_view* out);
/* How the key is held.  Every query is stored as a window table, 2^(c w) P for the windows w of a c-bit digit,
 * so that a multiexp needs no doublings; the widths c are per query role.  Full tables cost W = ceil(255 / c) times
 * the points (about 70 GB for a 2^24-row key); when they would not fit the budget -- the ZG16_TABLE_BUDGET_MB
 * environment variable, else four fifths of the GPU memory free at upload -- the upload keeps only every
 * `step`-th window (1 / step of the memory) and the multiexps use `step` bucket sets per result instead of one:
 * more bucket work, the same proof.  An upload that does not fit even with one window per query fails with
 * ZG16_ERR_OUT_OF_MEMORY.  Any pointer may be NULL. */
zg16_status zg16_pk_table_info(const zg16_pk* pk, unsigned* width_h, unsigned* width_witness, unsigned* width_g2,
                               unsigned* step, size_t* device_bytes);

/* ---- constraint matrices ------------------------------------------------------------------ */
/* Upload A, B, C once per circuit; the structure does not depend on the witness.  Also derives the
 * A/B density maps (which variables own a base in the a / b queries) from the column indices, the
 * job bellman's DensityTracker does while zinc-vm synthesizes. */
zg16_status zg16_r1cs_upload(zg16_ctx* ctx, const zg16_r1cs_view* view, zg16_r1cs** out);
void        zg16_r1cs_free(zg16_r1cs* r1cs);
/* popcounts of the density maps: a_aux, b_input, b_aux (any pointer may be NULL). */
zg16_status zg16_r1cs_density(const zg16_r1cs* r1cs, size_t* a_aux, size_t* b_input, size_t* b_aux);
/* The shape the matrices were uploaded with (any pointer may be NULL).  zg16_prove reads n_inputs + n_aux witness
 * elements through the pointers it is given and cannot see how many the caller holds: a host that proves a circuit it
 * only knows by name (zg16_r1cs_cached with view = NULL) compares its witness lengths with these first and reports
 * AssignmentMissing on a mismatch -- the Rust, C++ and Python create_proof bindings do. */
zg16_status zg16_r1cs_shape(const zg16_r1cs* r1cs, size_t* n_rows, size_t* n_inputs, size_t* n_aux);

/* ---- resident keys and circuits between calls ----------------------------------------------
 * The reference's callers cannot hold a device handle: `zvm prove` reads the key and proves once per process
 * (zinc-vm/src/zvm/arguments/command/prove.rs:73-88) and zinc-tester proves every case with `params.clone()`
 * (zinc-tester/src/runners/proof_check.rs:122, :181).  A drop-in `create_random_proof` therefore asks the library
 * for what it has already uploaded; expanding a key into window tables costs about a hundred proofs.
 *
 * zg16_default_ctx: one context per GPU, created on first use, alive until zg16_shutdown or the end of the process
 * (do not pass it to zg16_ctx_destroy while others may still ask for it).
 * zg16_pk_cached: the resident handle of the key `view` shows, uploaded (unsharded) on first sight.  A key is
 * recognised by its vector lengths, its verifying-key points, 1024 evenly spaced points of every query and a
 * position-weighted sum and an xor over every word of every query (memory-bound: milliseconds); with `full_digest`
 * every byte goes through the hash instead.  *was_cached (may be NULL) tells whether the upload was skipped.
 * zg16_r1cs_cached: the same for constraint matrices.  The structure of a circuit does not depend on its inputs,
 * so a caller that can name it -- `circuit_id`, any bytes, e.g. a digest of the bytecode and the method name
 * (the synthesis-side cost: zinc-vm/src/constraint_systems/dedup.rs:30-69, logging.rs:99-111) -- calls with
 * view = NULL first: a known circuit comes back at once and the caller synthesizes with a constraint system that
 * records only the witness; *out = NULL with ZG16_OK means unknown, record the matrices and call again with the
 * view.  With circuit_id = NULL the matrices themselves are digested (every byte).
 * Handles returned here belong to the context: zg16_pk_free / zg16_r1cs_free ignore them; they are released by
 * zg16_cache_clear, by zg16_ctx_destroy, or when newer entries push them out (least recently used first; at most
 * ZG16_CACHE_KEYS keys, default 4, and ZG16_CACHE_CIRCUITS circuits, default 16, per context; a key that does not
 * fit beside the cached ones evicts them).  A key handle stays valid at least until the next zg16_pk_cached call
 * on its context, a circuit handle until the next zg16_r1cs_cached call.
 * The default context is one object for every thread that asks, and a context is used by one thread at a time:
 * a host with several proving threads serialises each whole look-up / prove sequence on it (the Rust, C++ and
 * Python create_proof bindings hold a process-wide lock for the call), or gives each thread a context of its own. */
zg16_status zg16_default_ctx(int device_id, zg16_ctx** out);
void        zg16_shutdown(void);
zg16_status zg16_pk_cached(zg16_ctx* ctx, const zg16_pk_view* view, int full_digest, zg16_pk** out,
                           int* was_cached);
/* 1 when `pk` is a key handle this context's cache still holds (and marks it used), else 0: lets a host that
 * remembers which of its key objects maps to which handle skip building a view for zg16_pk_cached.  `id` is what
 * zg16_pk_id returned for the handle when it was handed out -- a serial number no later handle of the process
 * repeats.  The address alone does not identify a key: once the cache has evicted it, a newer key can be allocated
 * at the same address, and a host thread that still remembers the old mapping would prove with the wrong tables.
 * `pk` itself is never dereferenced here, so a stale address is safe to ask about. */
uint64_t    zg16_pk_id(const zg16_pk* pk);
int         zg16_pk_cache_holds(zg16_ctx* ctx, const zg16_pk* pk, uint64_t id);
zg16_status zg16_r1cs_cached(zg16_ctx* ctx, const uint8_t* circuit_id, size_t circuit_id_len,
                             const zg16_r1cs_view* view, zg16_r1cs** out, int* was_cached);
zg16_status zg16_cache_clear(zg16_ctx* ctx);
typedef struct {
  uint64_t keys, circuits;               /* entries held now          */
  uint64_t key_hits, key_misses, circuit_hits, circuit_misses, evictions;
  uint64_t key_bytes;                    /* HBM held by cached keys   */
} zg16_cache_counts;
zg16_status zg16_cache_stats(const zg16_ctx* ctx, zg16_cache_counts* out);

/* ---- prove -------------------------------------------------------------------------------- */
/* bellman `create_proof(circuit, params, r, s)`: z_in has n_inputs elements (z_in[0] = ONE),
 * z_aux has n_aux elements, r and s are Montgomery Fr.  Witness pointers are host memory unless
 * `witness_on_device` is non-zero, in which case they are device pointers on ctx's GPU. */
zg16_status zg16_prove(zg16_ctx* ctx, const zg16_pk* pk, const zg16_r1cs* r1cs,
                       const uint64_t* z_in, const uint64_t* z_aux, int witness_on_device,
                       const uint64_t r[4], const uint64_t s[4], zg16_proof* out);

/* The same for a host that keeps bellman's own row evaluation (`ProvingAssignment::enforce` pushes
 * a.evaluate(), b.evaluate(), c.evaluate() per constraint, including the n_inputs extra rows that
 * `create_proof` appends): a, b, c hold r1cs.n_rows Fr each, in host memory or, with
 * `evals_on_device`, device memory.  The matrices in `r1cs` are then used only for their shape and
 * density maps; the row-evaluation kernels do not run. */
zg16_status zg16_prove_from_evals(zg16_ctx* ctx, const zg16_pk* pk, const zg16_r1cs* r1cs,
                                  const uint64_t* a, const uint64_t* b, const uint64_t* c,
                                  int evals_on_device, const uint64_t* z_in, const uint64_t* z_aux,
                                  int witness_on_device, const uint64_t r[4], const uint64_t s[4],
                                  zg16_proof* out);

/* Graph replay (opt-in).  A proof is a fixed sequence of launches for a given key and set of matrices (grids are
 * sized for the worst case and read the actual counts on the device), so zg16_prove / zg16_prove_partial can capture
 * it into a HIP graph the second time they see a (key, matrices) pair and replay it afterwards: one call queues the
 * whole proof.  mode 1: on; 0: off; -1 (default): on only if the ZG16_GRAPH environment variable is 1.  It is off
 * by default because replay measured slower than plain multi-stream launches on MI355X / ROCm 7.2 (the graph's
 * branches overlap less than the streams do).  Graphs are dropped when a handle is freed.  Per-kernel timing
 * (zg16_profile_enable) needs plain launches: with profiling on no graph is used; a replayed proof reports only
 * total_ms in zg16_last_timings.  No counterpart upstream. */
zg16_status zg16_graph_mode(zg16_ctx* ctx, int mode);
/* Proofs captured, proofs replayed, captures that failed (those pairs stay on plain launches) since the context
 * was created; any pointer may be NULL. */
zg16_status zg16_graph_counts(const zg16_ctx* ctx, uint64_t* captures, uint64_t* replays, uint64_t* failures);

/* bellman `Proof::write`: the 128 bytes whose hex `zvm prove` prints
 * (zinc-vm/src/zvm/arguments/command/prove.rs:171-174).  Host arithmetic only. */
zg16_status zg16_proof_encode(const zg16_proof* proof, uint8_t out[128]);

/* Multi-GPU form: every rank calls zg16_prove_partial with its key shard (and the full
 * witness); the partials are gathered (RCCL all_gather / host copy) and any one rank calls
 * zg16_prove_finish on the full list.  The vk points come from the same zg16_pk_view. */
zg16_status zg16_prove_partial(zg16_ctx* ctx, const zg16_pk* pk, const zg16_r1cs* r1cs,
                               const uint64_t* z_in, const uint64_t* z_aux, int witness_on_device,
                               zg16_partial* out);
zg16_status zg16_prove_finish(const zg16_pk_view* vk_points, const zg16_partial* partials,
                              size_t n_partials, const uint64_t r[4], const uint64_t s[4],
                              zg16_proof* out);

/* Transform split for three or more GPUs.  bellman runs three independent chains before it can form H
 * (row evaluation of A, B or C, then `ifft`, then `coset_fft`); each chain can run on its own GPU:
 *   zg16_eval_chain      rows of matrix `which` (0 = A, 1 = B, 2 = C) times z, ifft, coset_fft -> out_dev,
 *                        2^log_m Fr of DEVICE memory owned by the caller
 *   (the caller moves the B and C vectors to the GPU that holds A: two point-to-point copies)
 *   zg16_h_from_cosets   a <- icoset_fft((a*b - c) / Z): the H coefficients, device memory, in place
 *   (the caller broadcasts H)
 *   zg16_prove_partial_h zg16_prove_partial with the H coefficients supplied (device memory, 2^log_m Fr)
 * zg16_r1cs_log_domain returns log_m.  No counterpart call exists upstream: bellman_ce does all of it inside
 * `create_proof` on one host (zinc-vm/src/core/circuit/facade.rs:153). */
zg16_status zg16_eval_chain(zg16_ctx* ctx, const zg16_r1cs* r1cs, const uint64_t* z_in,
                            const uint64_t* z_aux, int witness_on_device, int which, uint64_t* out_dev);
zg16_status zg16_h_from_cosets(zg16_ctx* ctx, uint64_t* a_dev, const uint64_t* b_dev,
                               const uint64_t* c_dev, unsigned log_m);
zg16_status zg16_prove_partial_h(zg16_ctx* ctx, const zg16_pk* pk, const zg16_r1cs* r1cs,
                                 const uint64_t* z_in, const uint64_t* z_aux, int witness_on_device,
                                 const uint64_t* h_dev, zg16_partial* out);
unsigned    zg16_r1cs_log_domain(const zg16_r1cs* r1cs);

/* ---- one proof over several GPUs, transforms sharded too ---------------------------------------
 * No counterpart upstream (bellman_ce proves on one host); the proof is the same group elements however it is split.
 * With G = 2, 4 or 8 GPUs and m = G L rows (m >= 2 G^2), GPU s evaluates the rows of its blocks, holds coefficient
 * s + G j of every polynomial at slot j, and owns the h points of those indices (zg16_pk_upload_cyclic_h: h cyclic,
 * l / a / b_g1 / b_g2 by contiguous range as zg16_pk_upload shards them).  What crosses GPUs: three all-to-all rounds
 * of 3, 2 and 1 vectors of 32 L bytes, and one zg16_partial per GPU.
 *
 * Per GPU: zg16_hshard_create once per (key, circuit); then per proof, for stage = 0, 1, 2, 3:
 *     zg16_hshard_stage(hs, stage, z_in, z_aux, witness_on_device, work, got, &n)   (witness used by stage 0 only)
 *     if n > 0: ONE exchange -- an equal-split all-to-all of the first n vectors' worth (n * zg16_hshard_vector_bytes)
 *               of `work` into the same range of `got`: chunk h (1/G of that range) goes to GPU h, which stores the
 *               chunk it gets from GPU g at chunk g.  The stage leaves its n vectors destination-major (chunk h holds
 *               block h of vector 0, block h of vector 1, ...) and the next stage reads what arrived in that order, so
 *               nothing is packed or unpacked around the collective.  The call returns with `work` ready.
 * and zg16_hshard_finish for this GPU's partial sums; zg16_prove_finish adds the partials of all GPUs.
 * work, got: device memory of the caller, 3 vectors each.  zg16_hshard_stage(.., 0, ..) restarts a proof at any time.
 * zg16_hshard_stream_ordered(hs, 1): the stages only queue their kernels on the context's stream and return; the
 * caller queues each exchange on that same stream (zg16_ctx_set_stream with the stream its collectives are ordered
 * on), so no host thread waits between the stages; zg16_hshard_finish still waits for the results. */
typedef struct zg16_hshard zg16_hshard;
zg16_status zg16_pk_upload_cyclic_h(zg16_ctx* ctx, const zg16_pk_view* view, int shard_index, int shard_count,
                                    zg16_pk** out);
zg16_status zg16_hshard_create(zg16_ctx* ctx, const zg16_pk* pk, const zg16_r1cs* r1cs, unsigned log_shards,
                               unsigned shard, zg16_hshard** out);
void        zg16_hshard_free(zg16_hshard* hs);
size_t      zg16_hshard_vector_bytes(const zg16_hshard* hs);
zg16_status zg16_hshard_stage(zg16_hshard* hs, int stage, const uint64_t* z_in, const uint64_t* z_aux,
                              int witness_on_device, uint64_t* work_dev, uint64_t* got_dev, int* n_exchange);
zg16_status zg16_hshard_stream_ordered(zg16_hshard* hs, int on);
zg16_status zg16_hshard_finish(zg16_hshard* hs, zg16_partial* out);
/* Optional: the proof's r and s (Montgomery form, as zg16_prove takes them) before stage 0, on EVERY GPU.  The
 * assembly of bellman_ce 0.3.0 `create_proof` is linear in the query sums -- C = [s*(alpha + r*delta) + r*beta_g1]
 * + s*a_query + r*b_query + h + l -- so each GPU then adds s*a + r*b1 of its own sums into the `hl` of the partial
 * zg16_hshard_finish returns (a 254-bit double scalar multiplication on the host, beside its h multiexp), and on
 * the GPU that will put the proof together (assembles != 0) a host thread prepares alpha + r*delta, beta + s*delta
 * and the bracket above while the stages run.  zg16_hshard_assemble on that GPU adds the partials of all GPUs to
 * them: a few dozen additions after the gather instead of the two fixed-base and the double scalar multiplication
 * of zg16_prove_finish.  Partials made after zg16_hshard_randomness go to zg16_hshard_assemble only
 * (zg16_prove_finish would add s*a + r*b1 a second time); r and s hold for the one proof that the next stage 0
 * starts (a proof restarted without a new call proves plainly again; a call in the middle of a proof abandons it). */
zg16_status zg16_hshard_randomness(zg16_hshard* hs, const uint64_t r_mont[4], const uint64_t s_mont[4], int assembles);
zg16_status zg16_hshard_assemble(zg16_hshard* hs, const zg16_partial* parts, size_t n_parts, zg16_proof* out);

/* The same driven from ONE process that sees all the GPUs (what a Rust host linking this library can do without a
 * launcher): a context per listed device, a host thread per GPU during a call, peer-to-peer copies for the
 * exchanges.  zg16_multi_load shards the key and uploads the matrices to every GPU (previous ones are released);
 * sharded_transforms: 1 = the schedule above (needs 2, 4 or 8 devices), 0 = every GPU repeats the row evaluation
 * and the transforms and only the multiexps are split (any device count), -1 = choose (sharded from 2^20 rows).
 * zg16_multi_prove: witness and (r, s) in host memory, as zg16_prove.  A device id may be listed more than once
 * (several contexts on one GPU), which is how a one-GPU machine rehearses the schedule. */
typedef struct zg16_multi zg16_multi;
zg16_status zg16_multi_create(const int* device_ids, int n_devices, zg16_multi** out);
void        zg16_multi_destroy(zg16_multi* mg);
const char* zg16_multi_last_error(const zg16_multi* mg);
zg16_status zg16_multi_load(zg16_multi* mg, const zg16_pk_view* key, const zg16_r1cs_view* circuit,
                            int sharded_transforms);
int         zg16_multi_sharded_transforms(const zg16_multi* mg);
zg16_status zg16_multi_prove(zg16_multi* mg, const uint64_t* z_in, const uint64_t* z_aux, const uint64_t r[4],
                             const uint64_t s[4], zg16_proof* out);

/* ---- stand-alone stages (each is one stage of bellman's prover) ------------------------------
 * data: 2^log_n Montgomery Fr, natural order in and out, transformed in place.
 * inverse = 0: `EvaluationDomain::fft`; inverse = 1: `ifft` (includes the 1/m factor);
 * coset != 0 selects `coset_fft` / `icoset_fft` (multiplicative generator 7). */
zg16_status zg16_ntt(zg16_ctx* ctx, uint64_t* data, unsigned log_n, int inverse, int coset,
                     int on_device);
/* One `EvaluationDomain::fft` / `ifft` of 2^log_n points spread over G = 2^log_shards GPUs (G <= 8) with a
 * single all-to-all.  bellman_ce has no counterpart (its transform runs on one host's cores); the values are
 * those of zg16_ntt on the whole vector.
 *   forward: GPU g holds x[g + G j], j < L = 2^log_n / G.  Each GPU runs zg16_ntt(part, log_n - log_shards, 0, 0, 1);
 *            an all-to-all with equal splits follows (chunk h of every GPU goes to GPU h, which stores the chunk
 *            it gets from GPU g at block g); then zg16_ntt_cross(.., inverse = 0) on every GPU.  Block k1 of
 *            GPU h (L / G elements) then holds X[(h L/G + i) + L k1], i < L / G.
 *   inverse: from that layout, zg16_ntt_cross(.., inverse = 1), the same all-to-all, then
 *            zg16_ntt(part, log_n - log_shards, 1, 0, 1); GPU g ends with x[g + G j].  The 1/N is included.
 * data: this GPU's L elements, transformed in place; device memory when `on_device` is non-zero (what the
 * exchange delivers), host memory otherwise. */
zg16_status zg16_ntt_cross(zg16_ctx* ctx, uint64_t* data, unsigned log_n, unsigned log_shards,
                           unsigned shard, int inverse, int on_device);
/* H coefficients from row evaluations a, b, c (n_rows each): 3 ifft, 3 coset_fft, pointwise
 * a*b-c, divide by Z on the coset, icoset_fft.  Writes m - 1 Montgomery Fr to h. */
zg16_status zg16_h_coefficients(zg16_ctx* ctx, const uint64_t* a, const uint64_t* b,
                                const uint64_t* c, size_t n_rows, uint64_t* h, size_t* m_out);
/* bellman `multiexp`: sum_i scalars[i] * bases[i].  Scalars are Montgomery Fr.  If `density` is
 * non-NULL it has one byte per scalar; bases then hold points only for the non-zero bytes, in
 * order, and n_bases must equal their count.  Result is an affine point. */
zg16_status zg16_msm_g1(zg16_ctx* ctx, const uint64_t* bases, size_t n_bases,
                        const uint64_t* scalars, size_t n_scalars, const uint8_t* density,
                        uint64_t out[8]);
zg16_status zg16_msm_g2(zg16_ctx* ctx, const uint64_t* bases, size_t n_bases,
                        const uint64_t* scalars, size_t n_scalars, const uint8_t* density,
                        uint64_t out[16]);
/* The same multiexp over a base set that stays in HBM (what a proving-key query is): upload once,
 * multiply many scalar vectors.  With `window_tables` non-zero the upload also stores 2^(c w) * P for
 * every window w, so all windows share one bucket set (the layout zg16_prove uses); it costs W times
 * the memory.  `group` is 1 (G1, 64-byte points) or 2 (G2, 128-byte points).
 * zg16_msm_resident: scalars are n Montgomery Fr in host memory, or device memory when
 * `scalars_on_device` is non-zero; out receives the affine sum (8 or 16 limbs). */
typedef struct zg16_bases zg16_bases;
zg16_status zg16_bases_upload(zg16_ctx* ctx, int group, const uint64_t* bases, size_t n,
                              int window_tables, zg16_bases** out);
void        zg16_bases_free(zg16_bases* bases);
zg16_status zg16_msm_resident(zg16_ctx* ctx, const zg16_bases* bases, const uint64_t* scalars,
                              int scalars_on_device, uint64_t* out);
/* zg16_msm_resident for callers that cannot allocate (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc): the same sum, computed in a fixed device array that is part of the
 * library image (about 0.8 MB per GPU, held until the process ends) instead of the context's grow-only
 * scratch arena.  No counterpart in the reference.  The call performs no hipMalloc / hipFree and uses no
 * allocating library routine.  Limits: the base set must have been uploaded with window_tables != 0 and hold
 * at most 16384 points (and no more than the direct-table limit of the upload, 2^14 unless
 * ZG16_DIRECT_MAX_LOG lowered it); anything else is ZG16_ERR_INVALID_ARGUMENT.  Scalars as for
 * zg16_msm_resident (host scalars are staged inside the same array).  All calls on one GPU share the array:
 * calls that overlap on one GPU -- from any context or thread -- race. */
zg16_status zg16_msm_resident_noalloc(zg16_ctx* ctx, const zg16_bases* bases, const uint64_t* scalars,
                                      int scalars_on_device, uint64_t* out);
/* Row evaluations <A_row, z>, <B_row, z>, <C_row, z> (n_rows Montgomery Fr each, host memory). */
zg16_status zg16_spmv(zg16_ctx* ctx, const zg16_r1cs* r1cs, const uint64_t* z_in,
                      const uint64_t* z_aux, uint64_t* a, uint64_t* b, uint64_t* c);

/* ---- key-generation support ------------------------------------------------------------------
 * out[i] = scalars[i] * base as affine points (scalars are Montgomery Fr, host memory).  This is the
 * group half of bellman `generate_parameters` (zinc-vm/src/core/circuit/facade.rs:126): every key
 * point is a known scalar times one generator.  Not used while proving. */
zg16_status zg16_fixed_base_g1(zg16_ctx* ctx, const uint64_t base[8], const uint64_t* scalars, size_t n,
                               uint64_t* out);
zg16_status zg16_fixed_base_g2(zg16_ctx* ctx, const uint64_t base[16], const uint64_t* scalars, size_t n,
                               uint64_t* out);

/* bellman `generate_parameters(circuit, g1, g2, alpha, beta, gamma, delta, tau)` -- what
 * `generate_random_parameters` runs after drawing those values (zinc-vm/src/core/circuit/facade.rs:126,
 * core/contract/facade.rs:299) -- on the GPU: the Lagrange values L_j(tau) by one inverse transform, the per-variable
 * sums A_v(tau), B_v(tau), C_v(tau) as a sparse product with the transposed matrices, and every key point by
 * fixed-base multiplication.  `circuit` holds the rows as for proving (the `input_i * 0 = 0` rows appended; the
 * coefficients are structure, no witness is involved).  Scalars are Montgomery Fr, g1 / g2 affine points.
 * The caller allocates the output arrays: ic n_inputs points, h m - 1 (m = the domain size, next power of two
 * >= n_rows), l n_aux, a / b_g1 / b_g2 n_inputs + n_aux (capacity; n_a and n_b receive the counts after the
 * identities are dropped, as upstream drops them).  ZG16_ERR_UNCONSTRAINED_VARIABLE mirrors
 * SynthesisError::UnconstrainedVariable. */
typedef struct {
  uint64_t g1[8], g2[16];
  uint64_t alpha[4], beta[4], gamma[4], delta[4], tau[4];
} zg16_toxic_waste;
typedef struct {
  uint64_t alpha_g1[8], beta_g1[8], beta_g2[16], gamma_g2[16], delta_g1[8], delta_g2[16];
  uint64_t* ic;
  uint64_t* h;
  uint64_t* l;
  uint64_t* a;    size_t n_a;
  uint64_t* b_g1;
  uint64_t* b_g2; size_t n_b;
} zg16_parameters_out;
zg16_status zg16_generate_parameters(zg16_ctx* ctx, const zg16_r1cs_view* circuit, const zg16_toxic_waste* toxic,
                                     zg16_parameters_out* out);

/* bellman `verify_proof(&prepare_verifying_key(&vk), &proof, &public_inputs)` (zinc-vm/src/core/facade.rs:32-38;
 * the check zinc-tester ends every proof test with, zinc-tester/src/runners/proof_check.rs:243-264).  Host
 * arithmetic only (four Miller loops and one final exponentiation, whatever the circuit size); no context needed.
 * public_inputs: n_inputs Montgomery Fr, without the leading ONE.  ic holds n_ic = n_inputs + 1 points, otherwise
 * ZG16_ERR_MALFORMED_KEY (bellman's MalformedVerifyingKey).  *accepted = 1 iff
 * e(A, B) = e(alpha, beta) * e(ic[0] + sum x_i ic[i+1], gamma) * e(C, delta). */
typedef struct {
  const uint64_t* alpha_g1; /* 8  */
  const uint64_t* beta_g2;  /* 16 */
  const uint64_t* gamma_g2; /* 16 */
  const uint64_t* delta_g2; /* 16 */
  const uint64_t* ic;  size_t n_ic;  /* G1 */
} zg16_vk_view;
zg16_status zg16_verify_proof(const zg16_vk_view* vk, const zg16_proof* proof, const uint64_t* public_inputs,
                              size_t n_inputs, int* accepted);

/* ---- timing of the last zg16_prove / zg16_prove_partial on this context -------------------- */
typedef struct {
  float upload_ms, spmv_ms, ntt_ms, msm_g1_ms, msm_g2_ms, finish_ms, total_ms;
} zg16_timings;
zg16_status zg16_last_timings(const zg16_ctx* ctx, zg16_timings* out);

/* ---- primitive self-test hooks -----------------------------------------------------------------
 * Element-wise evaluation of the device field / group primitives so tests can pin them against
 * golden vectors (pairing_ce / ff_ce semantics; those crates are not part of the reference tree).
 * field: 0 Fr, 1 Fq, 2 Fq2.  op: 0 add, 1 sub, 2 mul, 3 square, 4 negate, 5 inverse (0 -> 0).
 * group: 1 G1, 2 G2 (saturated 32-bit-limb code: key tables, host assembly); 3 G1, 4 G2 through the
 * arithmetic of the multiexp kernels (29-bit-limb field; G2 values spread over lane pairs).
 * op: 0 double p, 1 p + q (mixed), 2 p + q (projective), 3 k * p with k a Montgomery Fr; group 4 only (q unused):
 * 4 [6 x^2] p and 5 psi(p), the two sides of the subgroup test the key-file decoder runs on every b_g2 point
 * (x the BN parameter, psi the untwist-Frobenius-twist map); 6 the decoder's own kernel on each point: the first
 * limb of out[i] is 1 when it accepts p[i], else 0.  Groups 3 and 4: 7 p + q and 8 2p + 2q through the addition the
 * sums that end a multiexp use (a G1 point spread over four lanes, one coordinate each; 8 chains three of them).
 * Inputs and outputs are affine points / Montgomery limbs in host memory. */
zg16_status zg16_selftest_field(zg16_ctx* ctx, int field, int op, const uint64_t* a, const uint64_t* b,
                                size_t n, uint64_t* out);
zg16_status zg16_selftest_group(zg16_ctx* ctx, int group, int op, const uint64_t* p, const uint64_t* q,
                                size_t n, uint64_t* out);

/* ---- per-kernel timing -------------------------------------------------------------------------
 * With profiling enabled every kernel launch is bracketed by two HIP events on the context's
 * stream; totals accumulate per kernel kind until reset.  `units` is the algorithmic work the
 * launches covered: bytes for the SpMV / NTT kernels, terms for k_digits, additions for the
 * k_accumulate kernels, buckets for k_scan / k_split_sums.  on = 2 times only the kernel that
 * dominates (k_accumulate, three launches per proof), which keeps the timed run within a percent or two of an
 * untimed one; the events of on = 1 cost about 5 % of a 2^20-row prove. */
zg16_status zg16_profile_enable(zg16_ctx* ctx, int on);
/* on = 1: a prove issues every stage on one stream, so that per-kernel timings are free of the
 * contention between the overlapped pipelines (slower overall); 0: overlapped (the default);
 * -1: follow the ZG16_SERIAL environment variable. */
zg16_status zg16_profile_serial(zg16_ctx* ctx, int on);
zg16_status zg16_profile_reset(zg16_ctx* ctx);
int         zg16_profile_kernel_count(void);
zg16_status zg16_profile_get(const zg16_ctx* ctx, int kernel, const char** name, double* total_ms,
                             uint64_t* launches, double* units);

#ifdef __cplusplus
}
#endif
#endif /* ZINC_GROTH16_H */
