// One proof over several GPUs with the transforms sharded as well as the multiexps.
//
// bellman_ce 0.3.0 `create_proof` (reached from zinc-vm/src/core/circuit/facade.rs:153 and
// zinc-vm/src/core/contract/facade.rs:387) runs on one host; nothing upstream corresponds to this file.  The values
// are upstream's: every step is exact field / group arithmetic, so the proof does not depend on how it was split.
//
// Layout.  m = G L rows over G = 2^log_g GPUs.  GPU s evaluates the rows of its blocks [(s blk + i) + L k1] (blk = L / G),
// which is the layout the inverse sharded transform starts from (ntt.hip, "one transform over several GPUs"), and ends
// every inverse transform holding coefficient s + G j at slot j.  The h query is uploaded in that same cyclic order
// (zg16_pk_upload_cyclic_h), so the H coefficients never move: the only data that crosses GPUs are the three
// all-to-all rounds of the transforms -- 3 + 2 + 1 vectors of 32 L bytes per GPU -- and the 640-byte partial sums.
//
//   stage 0   rows of A, B, C (this GPU's blocks); cross-GPU step of the inverse transform      exchange 3 vectors
//   stage 1   c' = local ifft(c) / Z;  a, b: local ifft * g^i, local fft                          exchange 2 vectors
//   stage 2   cross-GPU step of the forward transform on a, b; a * b; cross-GPU inverse step      exchange 1 vector
//   stage 3   H = local ifft(a b) * g^-i / Z - c';  the h multiexp over this GPU's h points
// (icoset_fft is linear, so icoset_fft((A B - C) / Z) = icoset_fft(A B) / Z - ifft(c) / Z: c needs no coset
// evaluation.)  The multiexps over the witness (l, a, b_g1, b_g2) start in stage 0 on their own streams.
//
// One exchange per stage.  The n vectors a stage hands over leave it DESTINATION-MAJOR: chunk h of `work` (n blk
// elements: block h of vector 0, block h of vector 1, ...) is everything GPU h gets from this one, so the stage's
// whole exchange is a single equal-split all-to-all of the first n L elements, and GPU h finds the block it got from
// GPU g for vector v at (g n + v) blk of `got`.  The kernels on either side write and read those layouts directly
// (chunk strides in the cross-GPU step, an index remap in the first / last pass of a local transform): nothing is
// packed or unpacked in between.  At 2^20 rows on 8 GPUs a vector is 4 MiB per GPU and an exchange is bound by the
// collective's launch latency, not by the links, so three exchanges per proof instead of six is what counts.
//
// The stages are exported one by one so that any transport can carry the exchanges: zg16_multi_* below drives
// several GPUs from one process with peer copies; zinc_amd/distributed.py drives one process per GPU with RCCL.
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

#include <stdlib.h>

#include "r1cs.hpp"

using namespace zg;

struct zg16_hshard {
  zg16_ctx* ctx = nullptr;
  const zg16_pk* pk = nullptr;
  const zg16_r1cs* r = nullptr;
  unsigned log_g = 0, shard = 0, log_m = 0, log_l = 0;
  const HShardTables* tables = nullptr;
  Fr* c_coef = nullptr;  // L: ifft(c) / Z at this GPU's indices
  Fr* h = nullptr;       // L: the H coefficients at this GPU's indices
  Fr* tmp = nullptr;     // 3 L: vectors between the steps of a stage, natural order
  bool stream_ordered = false;  // stages only queue their work; the caller orders the exchanges on the context's stream
  MsmPending pend_g2, pend_z, pend_h;
  hipEvent_t ev_z = nullptr, ev_g2 = nullptr, ev_early = nullptr;
  int next_stage = 0;
  // Stream-ordered proofs queue the ~50 launches of the witness multiexps from a second host thread: queued by the
  // calling thread they keep it from the first exchange for about 0.4 ms, which the GPU's main stream -- the
  // critical path -- spends idle (a rank's share of a 2^20-row proof on 8 GPUs is 1.3 ms of GPU work).
  std::thread helper;
  zg16_status helper_status = ZG16_OK;
  uint64_t proofs = 0;  // the first proof sizes the scratch arenas and stays on one thread
  // zg16_hshard_randomness: r and s for the NEXT proof (rs_pending, assemble_pending); stage 0 takes them over, so
  // a proof restarted without a new call does not inherit an earlier proof's.  have_rs: zg16_hshard_finish adds
  // s*a + r*b1 of this GPU's sums to hl.  assembly_ready: `assembler` prepares the key's terms for
  // zg16_hshard_assemble; partial_has_rs: the partial zg16_hshard_finish last returned carries that share.
  Assembly as;
  bool rs_pending = false, assemble_pending = false;
  bool have_rs = false, assembly_ready = false, partial_has_rs = false;
  std::thread assembler;
  std::chrono::steady_clock::time_point t_begin;  // stage 0 entered (total_ms of zg16_last_timings: host clock, exchanges included)
};

// The multiexps over the witness (l, a, b_g1 in one launch; b_g2), queued on the context's early and aux streams
// behind hs->ev_z.
static zg16_status enqueue_witness_multiexps(zg16_hshard* hs, const Fr* z) {
  zg16_ctx* ctx = hs->ctx;
  const zg16_pk* pk = hs->pk;
  const zg16_r1cs* r = hs->r;
  char* pinned = (char*)ctx->pinned;
  const uint32_t tc = pk->table_c;
  // the witness launch sorts for b_g2 as well when the digit widths agree (see prove_partial_impl)
  MsmSortShare share;
  share.group = 2;
  share.sorted = ctx->ev[10];
  const bool try_share = tc == pk->table_c_g2 && pk->hi_b > pk->lo_b && !(getenv("ZG16_SHARE_SORT") && atoi(getenv("ZG16_SHARE_SORT")) == 0);
  {
    ZG_HIP(ctx, hipStreamWaitEvent(ctx->early_stream, hs->ev_z, 0));
    const MsmJob<Fq> jobs_z[3] = {
        {pk->l, z + r->n_in + pk->lo_l, nullptr, pk->hi_l - pk->lo_l, pk->hi_l - pk->lo_l, tc, 0, pk->l_d, pk->table_step},
        {pk->a, z, r->a_map + pk->lo_a, pk->hi_a - pk->lo_a, pk->hi_a - pk->lo_a, tc, 1, pk->a_d, pk->table_step},
        {pk->b1, z, r->b_map + pk->lo_b, pk->hi_b - pk->lo_b, pk->hi_b - pk->lo_b, tc, 2, pk->b1_d, pk->table_step}};
    const MsmStreams ms = {ctx->early_stream, nullptr, nullptr, nullptr};
    ZG_TRY(msm_launch<Fq>(ctx, ms, 6, pinned + 2 * MSM_PINNED_BYTES, jobs_z, 3, 3, &hs->pend_z, nullptr, try_share ? &share : nullptr));
    ZG_HIP(ctx, hipEventRecord(hs->ev_early, ctx->early_stream));
  }
  {
    ZG_HIP(ctx, hipStreamWaitEvent(ctx->aux_stream, hs->ev_z, 0));
    MsmJob<Fq2> jb = {pk->b2, z, r->b_map + pk->lo_b, pk->hi_b - pk->lo_b, pk->hi_b - pk->lo_b, pk->table_c_g2, 0, pk->b2_d, pk->table_step};
    const MsmStreams ms = {ctx->aux_stream, nullptr, nullptr, nullptr};
    ZG_TRY(msm_launch<Fq2>(ctx, ms, 2, pinned + MSM_PINNED_BYTES, &jb, 1, 1, &hs->pend_g2, nullptr, nullptr,
                           try_share && share.valid ? &share : nullptr));
    ZG_HIP(ctx, hipEventRecord(hs->ev_g2, ctx->aux_stream));
  }
  return ZG16_OK;
}

extern "C" {

zg16_status zg16_hshard_create(zg16_ctx* ctx, const zg16_pk* pk, const zg16_r1cs* r, unsigned log_shards, unsigned shard,
                               zg16_hshard** out) {
  if (!ctx || !pk || !r || !out) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad arguments to zg16_hshard_create");
  *out = nullptr;
  if (pk->ctx != ctx || r->ctx != ctx) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "handles belong to another context");
  if (log_shards < 1 || log_shards > 3 || shard >= (1u << log_shards) || 2 * log_shards + 1 > r->log_m)
    return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "2^%u rows over 2^%u GPUs (shard %u) is not a valid split", r->log_m, log_shards, shard);
  if (!pk->h_cyclic || pk->shard_count != (1 << log_shards) || pk->shard_index != (int)shard)
    return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "the key must come from zg16_pk_upload_cyclic_h with this shard index and count");
  if (!pk->table_c || !pk->table_c_h || !pk->table_c_g2)
    return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "a sharded proof needs the key's window tables (ZG16_MSM_TABLE_WINDOW=0 switches them off)");
  ZG_TRY(check_key_fits(ctx, pk, r));
  ZG_HIP(ctx, hipSetDevice(ctx->device));
  std::unique_ptr<zg16_hshard, void (*)(zg16_hshard*)> hs(new zg16_hshard(), zg16_hshard_free);
  hs->ctx = ctx; hs->pk = pk; hs->r = r;
  hs->log_g = log_shards; hs->shard = shard; hs->log_m = r->log_m; hs->log_l = r->log_m - log_shards;
  ZG_TRY(hshard_tables(ctx, hs->log_m, hs->log_g, shard, &hs->tables));
  const size_t L = (size_t)1 << hs->log_l;
  ZG_HIP(ctx, hipMalloc((void**)&hs->c_coef, L * sizeof(Fr)));
  ZG_HIP(ctx, hipMalloc((void**)&hs->h, L * sizeof(Fr)));
  ZG_HIP(ctx, hipMalloc((void**)&hs->tmp, 3 * L * sizeof(Fr)));
  ZG_HIP(ctx, hipEventCreateWithFlags(&hs->ev_z, hipEventDisableTiming));
  ZG_HIP(ctx, hipEventCreateWithFlags(&hs->ev_g2, hipEventDisableTiming));
  ZG_HIP(ctx, hipEventCreateWithFlags(&hs->ev_early, hipEventDisableTiming));
  *out = hs.release();
  return ZG16_OK;
}

void zg16_hshard_free(zg16_hshard* hs) {
  if (!hs) return;
  if (hs->helper.joinable()) hs->helper.join();
  if (hs->assembler.joinable()) hs->assembler.join();
  if (hs->ctx) {
    (void)hipSetDevice(hs->ctx->device);
    (void)hipDeviceSynchronize();
  }
  (void)hipFree(hs->c_coef);
  (void)hipFree(hs->h);
  (void)hipFree(hs->tmp);
  if (hs->ev_z) (void)hipEventDestroy(hs->ev_z);
  if (hs->ev_g2) (void)hipEventDestroy(hs->ev_g2);
  if (hs->ev_early) (void)hipEventDestroy(hs->ev_early);
  delete hs;
}

size_t zg16_hshard_vector_bytes(const zg16_hshard* hs) { return hs ? sizeof(Fr) << hs->log_l : 0; }

zg16_status zg16_hshard_stream_ordered(zg16_hshard* hs, int on) {
  if (!hs) return ZG16_ERR_INVALID_ARGUMENT;
  hs->stream_ordered = on != 0;
  return ZG16_OK;
}

zg16_status zg16_hshard_stage(zg16_hshard* hs, int stage, const uint64_t* z_in, const uint64_t* z_aux, int witness_on_device,
                              uint64_t* work_dev, uint64_t* got_dev, int* n_exchange) {
  if (!hs) return ZG16_ERR_INVALID_ARGUMENT;
  zg16_ctx* ctx = hs->ctx;
  if (!work_dev || !got_dev || stage < 0 || stage > 3) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad arguments to zg16_hshard_stage");
  if (stage != hs->next_stage && stage != 0)
    return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "zg16_hshard_stage: stage %d asked for, stage %d is next", stage, hs->next_stage);
  ZG_HIP(ctx, hipSetDevice(ctx->device));
  const zg16_pk* pk = hs->pk;
  const zg16_r1cs* r = hs->r;
  const size_t L = (size_t)1 << hs->log_l, m = (size_t)1 << hs->log_m;
  Fr* work = (Fr*)work_dev;
  Fr* got = (Fr*)got_dev;
  Fr* tmp = hs->tmp;
  hipStream_t st = ctx->stream;
  int n_ex = 0;
  const uint32_t log_blk = hs->log_l - hs->log_g;
  const size_t blk = (size_t)1 << log_blk;
  if (stage == 0) {
    hs->t_begin = std::chrono::steady_clock::now();
    hs->have_rs = hs->rs_pending;
    hs->assembly_ready = hs->rs_pending && hs->assemble_pending;
    hs->rs_pending = false;
    hs->partial_has_rs = false;
    Fr* z;
    ZG_TRY(stage_witness(ctx, r, z_in, z_aux, witness_on_device, &z));
    ZG_HIP(ctx, hipEventRecord(hs->ev_z, st));
    // rows first: the transforms are the critical path; the witness multiexps fill the GPU beside them
    ZG_TRY(spmv_blocks_device(ctx, r, z, tmp, tmp + L, tmp + 2 * L, log_blk, (uint32_t)L, hs->shard << log_blk, (uint32_t)L));
    ZG_TRY(ntt_cross_chunked(ctx, tmp, 0, work, 3 * blk, hs->log_m, hs->log_g, hs->shard, true, nullptr, 3, L, blk));
    if (hs->helper.joinable()) hs->helper.join();  // (a proof that was abandoned part-way)
    hs->helper_status = ZG16_OK;
    if (hs->stream_ordered && !ctx->profile && hs->proofs > 0) {
      hs->helper = std::thread([hs, z] {
        hs->helper_status = hipSetDevice(hs->ctx->device) == hipSuccess ? enqueue_witness_multiexps(hs, z) : ZG16_ERR_DEVICE;
      });
    } else {
      ZG_TRY(enqueue_witness_multiexps(hs, z));
    }
    n_ex = 3;
  } else if (stage == 1) {
    // what arrived: block g of vector v at (3 g + v) blk; what leaves: block h of vector v at (2 h + v) blk
    const ChunkLayout in3 = {log_blk, 3 * blk}, out2 = {log_blk, 2 * blk};
    // the three inverse transforms go through their passes together, then the two forward ones: four launches (a
    // pass over 2^17 elements is 128 workgroups)
    const Fr* in3v[3] = {got + 2 * blk, got, got + blk};
    Fr* out3v[3] = {hs->c_coef, tmp, tmp + L};
    const int which3[3] = {2, 1, 1};
    ZG_TRY(hshard_local_ntt_batch(ctx, hs->log_l, hs->tables, 3, in3v, out3v, true, which3, nullptr, &in3, nullptr));
    const Fr* in2v[2] = {tmp, tmp + L};
    Fr* out2v[2] = {work, work + blk};
    const int which2[2] = {0, 0};
    ZG_TRY(hshard_local_ntt_batch(ctx, hs->log_l, hs->tables, 2, in2v, out2v, false, which2, nullptr, nullptr, &out2));
    n_ex = 2;
  } else if (stage == 2) {
    // the coset evaluations of a and of b (forward cross steps), their product, the inverse cross step: one launch
    ZG_TRY(ntt_cross_product(ctx, got, got + blk, 2 * blk, work, 0, hs->log_m, hs->log_g, hs->shard));
    n_ex = 1;
  } else {
    ZG_TRY(hshard_local_ntt(ctx, hs->log_l, hs->tables, got, hs->h, true, 3, hs->c_coef));
    // slot j is coefficient shard + G j; the h query stops at coefficient m - 2
    const size_t G = (size_t)1 << hs->log_g;
    size_t n_local = m - 1 > hs->shard ? (m - 1 - hs->shard + G - 1) / G : 0;
    if (n_local > pk->hi_h) n_local = pk->hi_h;
    const MsmJob<Fq> job_h = {pk->h, hs->h, nullptr, n_local, pk->hi_h - pk->lo_h, pk->table_c_h, 0, pk->h_d, pk->table_step};
    const MsmStreams ms = {st, nullptr, nullptr, nullptr};
    ZG_TRY(msm_launch<Fq>(ctx, ms, 1, ctx->pinned, &job_h, 1, 1, &hs->pend_h));
  }
  // the caller moves `work` next (or reads the results): everything this stage queued on the main stream is done
  // first -- unless the caller queues the exchange on that same stream, where the order is the stream's
  if (!hs->stream_ordered) ZG_HIP(ctx, hipStreamSynchronize(st));
  hs->next_stage = stage + 1;
  if (n_exchange) *n_exchange = n_ex;
  return ZG16_OK;
}

zg16_status zg16_hshard_finish(zg16_hshard* hs, zg16_partial* out) {
  if (!hs || !out) return ZG16_ERR_INVALID_ARGUMENT;
  zg16_ctx* ctx = hs->ctx;
  if (hs->next_stage != 4) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "zg16_hshard_finish before stage 3");
  ZG_HIP(ctx, hipSetDevice(ctx->device));
  hs->next_stage = 0;
  hs->proofs++;
  if (hs->helper.joinable()) hs->helper.join();
  // The witness pipelines end before the h query's does: their sums are finished on the host while the GPU still
  // works on it.  Every pipeline has drained before this returns (on an error too: the caller may then reuse what
  // the kernels touch).  Polled: the runtime's blocking wait can wake a fraction of a millisecond late.
  auto drained = [](hipStream_t s) {
    hipError_t q;
    while ((q = hipStreamQuery(s)) == hipErrorNotReady) __builtin_ia32_pause();
    return q;
  };
  G1XYZZ g1[3], h_sum;
  G2XYZZ qb2;
  zg16_status early_status = hs->helper_status;
  const hipError_t e1 = drained(ctx->early_stream);
  if (early_status == ZG16_OK && e1 == hipSuccess) early_status = msm_collect<Fq>(ctx, &hs->pend_z, g1);
  if (early_status == ZG16_OK && e1 == hipSuccess && hs->have_rs)  // this GPU's share of s*a_query + r*b_query
    add_full(g1[0], double_scalar_mul_glv(g1[1], hs->as.sc.v, g1[2], hs->as.rc.v));
  hs->partial_has_rs = hs->have_rs && early_status == ZG16_OK && e1 == hipSuccess;
  hs->have_rs = false;
  const hipError_t e2 = drained(ctx->aux_stream);
  if (early_status == ZG16_OK && e2 == hipSuccess) early_status = msm_collect<Fq2>(ctx, &hs->pend_g2, &qb2);
  const hipError_t e3 = drained(ctx->stream);
  profile_collect(ctx);
  ZG_HIP(ctx, e1);
  ZG_HIP(ctx, e2);
  ZG_HIP(ctx, e3);
  ZG_TRY(early_status);
  ctx->timings = zg16_timings();
  ctx->timings.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - hs->t_begin).count();
  ZG_TRY(msm_collect<Fq>(ctx, &hs->pend_h, &h_sum));
  add_full(g1[0], h_sum);
  memcpy(out->hl, &g1[0], sizeof g1[0]);
  memcpy(out->a, &g1[1], sizeof g1[1]);
  memcpy(out->b1, &g1[2], sizeof g1[2]);
  memcpy(out->b2, &qb2, sizeof qb2);
  return ZG16_OK;
}

zg16_status zg16_hshard_randomness(zg16_hshard* hs, const uint64_t r_[4], const uint64_t s_[4], int assembles) {
  if (!hs) return ZG16_ERR_INVALID_ARGUMENT;
  zg16_ctx* ctx = hs->ctx;
  const zg16_pk* pk = hs->pk;
  if (!r_ || !s_) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad arguments to zg16_hshard_randomness");
  // in the middle of a proof: that proof is abandoned (its stages could otherwise end with the next proof's r and
  // s); stage 0 is next, as after zg16_hshard_finish
  hs->next_stage = 0;
  hs->have_rs = false;
  // bellman rejects a key whose delta is the identity before doing any work
  if (pk->delta_g1.is_inf() || pk->delta_g2.is_inf())
    return fail(ctx, ZG16_ERR_UNEXPECTED_IDENTITY, "delta is the point at infinity");
  if (hs->assembler.joinable()) hs->assembler.join();
  hs->as = Assembly();
  hs->as.alpha1 = pk->alpha_g1; hs->as.beta1 = pk->beta_g1; hs->as.delta1 = pk->delta_g1;
  hs->as.beta2 = pk->beta_g2; hs->as.delta2 = pk->delta_g2;
  hs->as.delta1_table = pk->delta1_table.get();
  hs->as.delta2_table = pk->delta2_table.get();
  Fr rm, sm;
  memcpy(rm.v, r_, 32);
  memcpy(sm.v, s_, 32);
  hs->as.rc = from_mont(rm);
  hs->as.sc = from_mont(sm);
  hs->rs_pending = true;
  hs->assemble_pending = assembles != 0;
  // needs nothing from the GPU: a host thread of its own, beside the stages
  if (assembles) hs->assembler = std::thread([hs] { hs->as.key_terms_with_c(); });
  return ZG16_OK;
}

zg16_status zg16_hshard_assemble(zg16_hshard* hs, const zg16_partial* parts, size_t n_parts, zg16_proof* out) {
  if (!hs) return ZG16_ERR_INVALID_ARGUMENT;
  zg16_ctx* ctx = hs->ctx;
  if (!parts || !n_parts || !out) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad arguments to zg16_hshard_assemble");
  if (hs->assembler.joinable()) hs->assembler.join();
  if (!hs->assembly_ready || !hs->partial_has_rs)
    return fail(ctx, ZG16_ERR_INVALID_ARGUMENT,
                "zg16_hshard_assemble needs zg16_hshard_randomness(.., assembles = 1) before the proof and zg16_hshard_finish after it");
  hs->assembly_ready = false;
  G1XYZZ hl = G1XYZZ::inf(), qa = G1XYZZ::inf();
  G2XYZZ qb2 = G2XYZZ::inf();
  for (size_t i = 0; i < n_parts; i++) {
    G1XYZZ p;
    G2XYZZ p2;
    memcpy(&p, parts[i].hl, sizeof p); add_full(hl, p);
    memcpy(&p, parts[i].a, sizeof p); add_full(qa, p);
    memcpy(&p2, parts[i].b2, sizeof p2); add_full(qb2, p2);
  }
  add_full(hs->as.A, qa);
  hs->as.with_g2(qb2);
  hs->as.finish(hl, out);  // C = the key's share + sum of the GPUs' shares
  return ZG16_OK;
}

}  // extern "C"

// ---- several GPUs driven from one process ---------------------------------------------------------------
namespace {

// All ranks wait for each other (a generation counter, so it can be reused at once).
struct Barrier {
  std::mutex mu;
  std::condition_variable cv;
  int n, waiting = 0;
  uint64_t generation = 0;
  explicit Barrier(int count) : n(count) {}
  void wait() {
    std::unique_lock<std::mutex> lock(mu);
    const uint64_t gen = generation;
    if (++waiting == n) {
      waiting = 0;
      generation++;
      cv.notify_all();
    } else {
      cv.wait(lock, [&] { return generation != gen; });
    }
  }
};

}  // namespace

struct zg16_multi {
  int n = 0;
  std::vector<int> device;
  std::vector<zg16_ctx*> ctx;
  std::vector<zg16_pk*> pk;
  std::vector<zg16_r1cs*> r1cs;
  std::vector<zg16_hshard*> hs;
  std::vector<Fr*> work, got;  // 3 L elements each, per GPU (sharded transforms only)
  bool sharded = false;
  unsigned log_g = 0;
  size_t n_in = 0, n_aux = 0;
  // the verifying-key points zg16_prove_finish needs
  uint64_t alpha_g1[8], beta_g1[8], delta_g1[8], beta_g2[16], delta_g2[16];
  std::string last_error;
};

static void multi_unload(zg16_multi* mg) {
  for (int g = 0; g < mg->n; g++) {
    if (mg->ctx[g]) (void)hipSetDevice(mg->device[g]);
    zg16_hshard_free(mg->hs[g]); mg->hs[g] = nullptr;
    zg16_pk_free(mg->pk[g]); mg->pk[g] = nullptr;
    zg16_r1cs_free(mg->r1cs[g]); mg->r1cs[g] = nullptr;
    (void)hipFree(mg->work[g]); mg->work[g] = nullptr;
    (void)hipFree(mg->got[g]); mg->got[g] = nullptr;
  }
}

static zg16_status multi_fail(zg16_multi* mg, int g, zg16_status st) {
  if (st != ZG16_OK) mg->last_error = "GPU " + std::to_string(mg->device[g]) + ": " + zg16_last_error(mg->ctx[g]);
  return st;
}

extern "C" {

zg16_status zg16_multi_create(const int* device_ids, int n_devices, zg16_multi** out) {
  if (!device_ids || !out || n_devices < 1 || n_devices > 64) return ZG16_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  std::unique_ptr<zg16_multi, void (*)(zg16_multi*)> mg(new zg16_multi(), zg16_multi_destroy);
  mg->n = n_devices;
  mg->device.assign(device_ids, device_ids + n_devices);
  mg->ctx.assign(n_devices, nullptr);
  mg->pk.assign(n_devices, nullptr);
  mg->r1cs.assign(n_devices, nullptr);
  mg->hs.assign(n_devices, nullptr);
  mg->work.assign(n_devices, nullptr);
  mg->got.assign(n_devices, nullptr);
  for (int g = 0; g < n_devices; g++) {
    const zg16_status st = zg16_ctx_create(device_ids[g], &mg->ctx[g]);
    if (st != ZG16_OK) return st;
  }
  // every GPU writes its chunks straight into its peers' buffers
  for (int g = 0; g < n_devices; g++)
    for (int h = 0; h < n_devices; h++) {
      if (device_ids[g] == device_ids[h]) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, device_ids[g], device_ids[h]) != hipSuccess || !can) continue;  // copies are staged by the runtime then
      if (hipSetDevice(device_ids[g]) != hipSuccess) return ZG16_ERR_DEVICE;
      const hipError_t e = hipDeviceEnablePeerAccess(device_ids[h], 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return ZG16_ERR_DEVICE;
      (void)hipGetLastError();
    }
  *out = mg.release();
  return ZG16_OK;
}

void zg16_multi_destroy(zg16_multi* mg) {
  if (!mg) return;
  multi_unload(mg);
  for (auto* c : mg->ctx) zg16_ctx_destroy(c);
  delete mg;
}

const char* zg16_multi_last_error(const zg16_multi* mg) { return mg ? mg->last_error.c_str() : "null handle"; }

int zg16_multi_sharded_transforms(const zg16_multi* mg) { return mg && mg->sharded ? 1 : 0; }

zg16_status zg16_multi_load(zg16_multi* mg, const zg16_pk_view* key, const zg16_r1cs_view* circuit, int sharded_transforms) {
  if (!mg || !key || !circuit) return ZG16_ERR_INVALID_ARGUMENT;
  multi_unload(mg);
  unsigned log_m = 0;
  while (((size_t)1 << log_m) < (circuit->n_rows ? circuit->n_rows : 1)) log_m++;
  unsigned log_g = 0;
  while ((1 << log_g) < mg->n) log_g++;
  const bool can_shard = mg->n > 1 && (1 << log_g) == mg->n && log_g <= 3 && 2 * log_g + 1 <= log_m;
  if (sharded_transforms > 0 && !can_shard) {
    mg->last_error = "sharded transforms need 2, 4 or 8 GPUs and at least 2 G^2 rows";
    return ZG16_ERR_INVALID_ARGUMENT;
  }
  // by default the transforms are sharded where they are a visible share of the proof (replicating them costs every
  // GPU about 1 ms at 2^20 rows and 14 ms at 2^24, against three all-to-all rounds of one collective each)
  mg->sharded = can_shard && (sharded_transforms > 0 || (sharded_transforms < 0 && log_m >= 20));
  mg->log_g = log_g;
  mg->n_in = circuit->n_inputs;
  mg->n_aux = circuit->n_aux;
  if (!key->alpha_g1 || !key->beta_g1 || !key->beta_g2 || !key->delta_g1 || !key->delta_g2) {
    mg->last_error = "verifying-key points missing";
    return ZG16_ERR_MALFORMED_KEY;
  }
  memcpy(mg->alpha_g1, key->alpha_g1, 64); memcpy(mg->beta_g1, key->beta_g1, 64); memcpy(mg->delta_g1, key->delta_g1, 64);
  memcpy(mg->beta_g2, key->beta_g2, 128); memcpy(mg->delta_g2, key->delta_g2, 128);
  // every GPU uploads its share on a thread of its own (table expansion is the long part)
  std::vector<zg16_status> st(mg->n, ZG16_OK);
  std::vector<std::thread> threads;
  for (int g = 0; g < mg->n; g++)
    threads.emplace_back([&, g] {
      zg16_ctx* ctx = mg->ctx[g];
      st[g] = mg->sharded ? zg16_pk_upload_cyclic_h(ctx, key, g, mg->n, &mg->pk[g]) : zg16_pk_upload(ctx, key, g, mg->n, &mg->pk[g]);
      if (st[g] == ZG16_OK) st[g] = zg16_r1cs_upload(ctx, circuit, &mg->r1cs[g]);
      if (st[g] == ZG16_OK && mg->sharded) {
        st[g] = zg16_hshard_create(ctx, mg->pk[g], mg->r1cs[g], log_g, (unsigned)g, &mg->hs[g]);
        if (st[g] == ZG16_OK) {
          const size_t bytes = 3 * zg16_hshard_vector_bytes(mg->hs[g]);
          if (hipMalloc((void**)&mg->work[g], bytes) != hipSuccess || hipMalloc((void**)&mg->got[g], bytes) != hipSuccess) {
            fail(ctx, ZG16_ERR_OUT_OF_MEMORY, "hipMalloc of the exchange buffers failed");
            st[g] = ZG16_ERR_OUT_OF_MEMORY;
          }
        }
      }
    });
  for (auto& t : threads) t.join();
  for (int g = 0; g < mg->n; g++)
    if (st[g] != ZG16_OK) {
      multi_fail(mg, g, st[g]);
      const zg16_status s = st[g];
      multi_unload(mg);
      return s;
    }
  return ZG16_OK;
}

zg16_status zg16_multi_prove(zg16_multi* mg, const uint64_t* z_in, const uint64_t* z_aux, const uint64_t r[4],
                             const uint64_t s[4], zg16_proof* out) {
  if (!mg || !r || !s || !out) return ZG16_ERR_INVALID_ARGUMENT;
  if (!mg->pk[0] || !mg->r1cs[0]) {
    mg->last_error = "zg16_multi_prove before zg16_multi_load";
    return ZG16_ERR_INVALID_ARGUMENT;
  }
  const int n = mg->n;
  std::vector<zg16_partial> parts(n);
  std::vector<zg16_status> st(n, ZG16_OK);
  Barrier barrier(n);
  std::vector<std::thread> threads;
  for (int g = 0; g < n; g++)
    threads.emplace_back([&, g] {
      zg16_ctx* ctx = mg->ctx[g];
      if (!mg->sharded) {
        st[g] = zg16_prove_partial(ctx, mg->pk[g], mg->r1cs[g], z_in, z_aux, 0, &parts[g]);
        return;
      }
      const size_t L = zg16_hshard_vector_bytes(mg->hs[g]) / sizeof(Fr), blk = L / (size_t)n;
      // r and s up front: every GPU's partial carries its share of s*a + r*b1, GPU 0's handle prepares the key's terms
      st[g] = zg16_hshard_randomness(mg->hs[g], r, s, g == 0);
      for (int stage = 0; stage < 4; stage++) {
        int n_ex = stage == 0 ? 3 : stage == 1 ? 2 : stage == 2 ? 1 : 0;
        // a rank that has failed keeps meeting the others at every barrier, so that nobody waits for ever
        if (st[g] == ZG16_OK)
          st[g] = zg16_hshard_stage(mg->hs[g], stage, z_in, z_aux, 0, (uint64_t*)mg->work[g], (uint64_t*)mg->got[g], &n_ex);
        if (stage == 3) break;
        barrier.wait();  // every GPU's `work` is complete and every GPU is done reading its `got`
        if (st[g] == ZG16_OK && hipSetDevice(mg->device[g]) == hipSuccess) {
          // chunk h (block h of each of the stage's vectors) goes to GPU h, which stores what it gets from GPU g at chunk g
          hipError_t e = hipSuccess;
          const size_t chunk = (size_t)n_ex * blk;
          for (int h = 0; h < n && e == hipSuccess; h++)
            e = hipMemcpyPeerAsync(mg->got[h] + (size_t)g * chunk, mg->device[h], mg->work[g] + (size_t)h * chunk,
                                   mg->device[g], chunk * sizeof(Fr), ctx->stream);
          if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
          if (e != hipSuccess) st[g] = fail(ctx, ZG16_ERR_DEVICE, "peer copy failed: %s", hipGetErrorString(e));
        }
        barrier.wait();  // every chunk has arrived
      }
      // a proof that failed part-way still has multiexps in flight on this GPU: zg16_hshard_finish waits for them
      if (st[g] == ZG16_OK) {
        st[g] = zg16_hshard_finish(mg->hs[g], &parts[g]);
      } else {
        (void)hipSetDevice(mg->device[g]);
        (void)hipDeviceSynchronize();
        mg->hs[g]->next_stage = 0;
      }
    });
  for (auto& t : threads) t.join();
  for (int g = 0; g < n; g++)
    if (st[g] != ZG16_OK) return multi_fail(mg, g, st[g]);
  if (mg->sharded) {
    const zg16_status as = zg16_hshard_assemble(mg->hs[0], parts.data(), (size_t)n, out);
    return as == ZG16_OK ? as : multi_fail(mg, 0, as);
  }
  zg16_pk_view v = {};
  v.alpha_g1 = mg->alpha_g1; v.beta_g1 = mg->beta_g1; v.beta_g2 = mg->beta_g2; v.delta_g1 = mg->delta_g1; v.delta_g2 = mg->delta_g2;
  const zg16_status fs = zg16_prove_finish(&v, parts.data(), (size_t)n, r, s, out);
  if (fs == ZG16_ERR_UNEXPECTED_IDENTITY) mg->last_error = "delta is the point at infinity";
  return fs;
}

}  // extern "C"
