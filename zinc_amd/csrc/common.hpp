// Internal plumbing shared by the library's translation units: the context object, error
// capture, device-buffer helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/zinc_groth16.h"
#include "curve.hpp"

namespace zg {

struct NttDomain;   // ntt.hip
struct ProveGraph;  // api.hip: a captured proof (hipGraphExec_t and what the host needs to read its results)
struct HandleCache; // cache.hip: keys and matrices the context keeps resident between proofs

}  // namespace zg

namespace zg {
// Kernels that per-launch timing (zg16_profile_*) distinguishes.
enum KernelId {
  K_SPMV = 0, K_NTT_PASS, K_NTT_SCALE, K_NTT_POINTWISE, K_MSM_DIGITS, K_MSM_SCAN, K_MSM_ACC0_G1, K_MSM_ACC0_G2,
  K_MSM_ACCN_G1, K_MSM_ACCN_G2, K_MSM_REDUCE_G1, K_MSM_REDUCE_G2, K_MSM_ONES_G1, K_MSM_ONES_G2, K_COUNT
};
struct KernelStat {
  double ms = 0, units = 0;
  uint64_t launches = 0;
};
struct PendingTiming {
  int id;
  double units;
  hipEvent_t a, b;
};
}  // namespace zg

struct zg16_ctx {
  int device = 0;
  int serial = -1;  // 1: every stage of a prove on the main stream, 0: overlapped pipelines, -1: as ZG16_SERIAL says
  int profile = 0;  // 0: off, 1: every kernel kind, 2: only the kinds a roofline is reported for
  zg::KernelStat kstat[zg::K_COUNT];
  std::vector<zg::PendingTiming> pending;
  std::vector<hipEvent_t> event_pool;
  hipStream_t stream = nullptr;      // stream all kernels are issued on
  hipStream_t own_stream = nullptr;  // created by the context
  hipStream_t aux_stream = nullptr;  // second stream: the G2 multiexp runs beside everything else
  hipStream_t early_stream = nullptr;  // third stream: the G1 queries that need only the witness
  hipStream_t side_stream = nullptr;   // side branch of the h multiexp (its unit-scalar sums)
  int cu_count = 256;
  void* pinned = nullptr;            // page-locked host memory for asynchronous result read-backs
  // staging of a witness handed over in pageable host memory (zg::copy_witness_from_host): STAGE_THREADS x 2
  // page-locked buffers of STAGE_CHUNK bytes, allocated at the first such proof, and the event behind each
  static constexpr int STAGE_THREADS = 8;
  static constexpr size_t STAGE_CHUNK = (size_t)1 << 20;
  void* stage = nullptr;
  hipEvent_t stage_ev[2 * STAGE_THREADS] = {};
  bool stage_busy[2 * STAGE_THREADS] = {};
  // a small circuit's whole witness (z_in || z_aux) through one page-locked block and ONE transfer
  static constexpr size_t SMALL_WITNESS_BYTES = (size_t)1 << 18;
  void* small_stage = nullptr;
  hipEvent_t small_stage_ev = nullptr;
  bool small_stage_busy = false;
  hipEvent_t ev[16] = {};
  std::string last_error = "";
  std::mutex error_mutex;            // a sharded proof queues its witness multiexps from a second host thread
  std::map<unsigned, zg::NttDomain*> domains;  // per log2 size
  std::map<uint64_t, void*> cross_tables;      // twiddles of the cross-GPU step of a sharded transform
  std::map<uint64_t, void*> hshard_tables;     // strided factor tables of an H computation spread over GPUs
  // Proofs replayed as HIP graphs, per (key, matrices) pair.  Every device buffer a proof touches belongs to the
  // context, the key or the matrices, so a captured proof stays valid until one of the scratch arenas moves
  // (scratch_gen) or a handle is freed.
  std::map<std::pair<const void*, const void*>, zg::ProveGraph*> graphs;
  uint64_t scratch_gen = 0;
  bool capturing = false;  // a stream capture is open: nothing may allocate
  uint64_t graph_captures = 0, graph_replays = 0, graph_failures = 0;
  int graph_mode = -1;     // 1: replay every proof as a graph, 0: never, -1: as ZG16_GRAPH says (default: small domains)
  // grow-only device scratch arenas, one per independent consumer
  void* scratch[10] = {};
  size_t scratch_bytes[10] = {};
  zg16_timings timings = {};
  zg::HandleCache* cache = nullptr;  // created on first use (zg16_pk_cached / zg16_r1cs_cached)
  // The affine byte-window table of the last G1 ([0]) and G2 ([1]) base a fixed-base multiplication used: key
  // generation multiplies seven scalar vectors by the same two generators, and building a table is a chain of
  // ~500 dependent group operations on 32 lanes (6 ms for G1, 30 ms for G2) whatever the vector's length.
  void* fb_table[2] = {};
  unsigned char fb_base[2][128] = {};
  bool fb_valid[2] = {};
  // When set, the table builders (msm_build_table, msm_build_direct) queue their kernels on `stream` and return
  // without waiting: their scratch allocations are appended here and the caller frees them after it has
  // synchronised.  zg16_pk_upload uses it to build the tables of a small key's five queries side by side.
  std::vector<void*>* deferred_frees = nullptr;
};

namespace zg {

inline zg16_status fail(zg16_ctx* ctx, zg16_status st, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) {
    std::lock_guard<std::mutex> lock(ctx->error_mutex);
    ctx->last_error = buf;
  }
  return st;
}

#define ZG_HIP(ctx, call)                                                                     \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return zg::fail(ctx, e_ == hipErrorOutOfMemory ? ZG16_ERR_OUT_OF_MEMORY : ZG16_ERR_DEVICE, \
                      "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

#define ZG_TRY(expr)                   \
  do {                                 \
    zg16_status s_ = (expr);           \
    if (s_ != ZG16_OK) return s_;      \
  } while (0)

// Grow-only scratch arena `slot` of at least `bytes`.
inline zg16_status scratch_get(zg16_ctx* ctx, int slot, size_t bytes, void** out) {
  if (ctx->scratch_bytes[slot] < bytes) {
    if (ctx->capturing) return fail(ctx, ZG16_ERR_DEVICE, "scratch arena %d would have to grow during a stream capture", slot);
    ctx->scratch_gen++;
    if (ctx->scratch[slot]) {
      ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
      ZG_HIP(ctx, hipFree(ctx->scratch[slot]));
      ctx->scratch[slot] = nullptr;
      ctx->scratch_bytes[slot] = 0;
    }
    size_t want = bytes + bytes / 8;
    ZG_HIP(ctx, hipMalloc(&ctx->scratch[slot], want));
    ctx->scratch_bytes[slot] = want;
  }
  *out = ctx->scratch[slot];
  return ZG16_OK;
}

// Brackets the launches issued during its lifetime with two events on the context's stream when
// profiling is on; zg::profile_collect turns the pairs into per-kernel totals after a sync.
struct KernelTimer {
  zg16_ctx* ctx;
  PendingTiming p;
  bool on;
  hipStream_t st;
  KernelTimer(zg16_ctx* c, int id, double units, hipStream_t stream = nullptr)
      : ctx(c), on(c->profile == 1 || (c->profile == 2 && roofline_kind(id))), st(stream ? stream : c->stream) {
    if (!on) return;
    p.id = id;
    p.units = units;
    p.a = grab();
    p.b = grab();
    if (!p.a || !p.b) { on = false; return; }
    (void)hipEventRecord(p.a, st);
  }
  ~KernelTimer() {
    if (!on) return;
    (void)hipEventRecord(p.b, st);
    ctx->pending.push_back(p);
  }
  static bool roofline_kind(int id) {
    return id == K_MSM_ACC0_G1 || id == K_MSM_ACC0_G2;  // the dominant kernel (three launches per proof)
  }
  hipEvent_t grab() {
    if (!ctx->event_pool.empty()) {
      hipEvent_t e = ctx->event_pool.back();
      ctx->event_pool.pop_back();
      return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
  }
};

// Call with the stream idle (after a synchronize).
inline void profile_collect(zg16_ctx* ctx) {
  for (auto& p : ctx->pending) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      ctx->kstat[p.id].ms += ms;
      ctx->kstat[p.id].units += p.units;
      ctx->kstat[p.id].launches++;
    }
    ctx->event_pool.push_back(p.a);
    ctx->event_pool.push_back(p.b);
  }
  ctx->pending.clear();
}

static inline unsigned ceil_log2(size_t n) {
  unsigned l = 0;
  while (((size_t)1 << l) < n) l++;
  return l;
}

// ---- stage entry points implemented in the other translation units -------------------------
// NTT of 2^log_n device-resident elements, natural order in and out.  `data` is overwritten.
// inverse: use omega^-1 and scale by 1/m.  coset: coset_fft (pre-scale by g^i) or icoset_fft
// (post-scale by g^-i).
zg16_status ntt_device(zg16_ctx* ctx, Fr* data, unsigned log_n, bool inverse, bool coset);
// a, b, c: device arrays of 2^log_m elements (zero-padded row evaluations); on return `a` holds
// the icoset_fft output whose first m-1 entries are the H coefficients.  b, c are clobbered.
zg16_status h_pipeline_device(zg16_ctx* ctx, Fr* a, Fr* b, Fr* c, unsigned log_m);
// The two halves of it: ifft + coset_fft of one vector in place; a <- icoset_fft((a*b - c) / Z).
zg16_status coset_evals_device(zg16_ctx* ctx, Fr* p, unsigned log_m);
zg16_status h_finish_device(zg16_ctx* ctx, Fr* a, const Fr* b, const Fr* c, unsigned log_m);

// One multiexp term list: result[group] += sum_k scalar(k) * bases[k], k < n, where
// scalar(k) = scalars[index ? index[k] : k] (Montgomery Fr, device memory).
template <class F>
struct MsmJob {
  const Affine<F>* bases;
  const Fr* scalars;
  const uint32_t* index;  // optional gather map into scalars
  size_t n;
  // Non-zero when `bases` is a window table: bases[w * table_stride + k] = 2^(table_c * table_step * w) * P_k.
  size_t table_stride;
  uint32_t table_c;
  int group;  // which of the launch's results this job is summed into
  // Optional direct table of a small query (msm_build_direct), stride table_stride: when every job of a launch
  // brings one, the multiexp is a plain sum of looked-up multiples and `bases` is not read.
  const Affine<F>* direct = nullptr;
  // A table may hold only every table_step-th window (it then costs 1 / table_step of the memory): digit window w
  // looks up table window w / table_step and goes to bucket set w mod table_step of its group; the sets are
  // recombined as sum_j 2^(c j) set_j.  1: every window, one bucket set per group.
  uint32_t table_step = 1;
};
// Window width / count used for key tables, and the kernel that fills windows 1..W-1 of a table whose
// window 0 (n points, stride n) is already in place.
uint32_t msm_table_window(size_t points, int role);  // role 0: h, 1: l / a / b_g1, 2: b_g2
uint32_t msm_table_windows(uint32_t c, uint32_t step = 1);  // windows a table of width c holds when it keeps every step-th
// Smallest step for which `bytes_full / step` (about) fits the table memory budget: the ZG16_TABLE_BUDGET_MB
// environment variable, else four fifths of the GPU memory that is free now.  sizes: (points, width c, bytes per
// point) of every query that shares the step.  max_sets bounds step * groups so that the buckets stay addressable.
struct TableQuery {
  size_t points;
  uint32_t c, point_bytes;
};
uint32_t msm_table_step(const TableQuery* queries, int count);
template <class F>
zg16_status msm_build_table(zg16_ctx* ctx, Affine<F>* table, size_t n, uint32_t c, uint32_t step = 1);
// Direct table of a small query: every multiple d * 2^(5 w) * P_k a signed 5-bit digit can ask for, as affine
// points in the multiexp kernels' packed form: direct[(w * 16 + (d - 1)) * n + k].  `points` are the n bases in the
// ABI form (call it before msm_build_table rewrites them).  *out stays null when the query is larger than
// msm_direct_max() points (or empty); the caller frees *out with hipFree.
size_t msm_direct_max();
template <class F>
zg16_status msm_build_direct(zg16_ctx* ctx, const Affine<F>* points, size_t n, Affine<F>** out);
// Blocking forms on ctx->stream: the sum of several jobs as one multiexp, one XYZZ point to `out_host`.
zg16_status msm_g1_device(zg16_ctx* ctx, const MsmJob<Fq>* jobs, int njobs, G1XYZZ* out_host);
zg16_status msm_g2_device(zg16_ctx* ctx, const MsmJob<Fq2>* jobs, int njobs, G2XYZZ* out_host);

// Asynchronous form, so that several multiexps can be in flight on different streams.  msm_launch
// enqueues every kernel of the multiexp on `ms.main` plus a copy of the results into `pinned`
// (MSM_PINNED_BYTES of page-locked host memory) and returns without waiting; once `ms.main` is idle
// msm_collect turns the block into `ngroups` points.  Jobs of one launch may form several groups
// (independent sums) only when every job brings a window table of the same width.
static constexpr size_t MSM_PINNED_BYTES = 1024 * 1024;
struct MsmPending {
  void* pinned = nullptr;
  uint32_t nsets = 0, ngroups = 0, c = 0, shared = 0, step = 1;
  uint32_t direct = 0;  // the results are plain sums (one point per group), nothing to fold
  int acc_kernel = 0;  // KernelId whose work units (additions) are known only after the run
};
struct MsmStreams {
  hipStream_t main;
  hipStream_t side;             // optional: the unit-scalar sums run here beside the bucket pipeline
  hipEvent_t ev_fork, ev_join;  // used when side is set
};
// fixed: when given, the multiexp works in this caller-owned device memory instead of the context's scratch arena and
// allocates nothing; only launches whose jobs all bring direct tables fit that mode.
struct MsmWorkspace {
  char* ptr;
  size_t bytes;
};
// One sort for two multiexps over the same scalars (b_g1 and b_g2 of a proof: same witness values through the same
// gather map, different bases).  A launch given `lend` fills it after its sort -- where the bucket-sorted entries and
// the bucket offsets lie, what shape they have, and an event recorded behind the sort; a launch given that record as
// `borrow` (one job, one group, the same digit width and table step, the terms of the lender's group `group`) skips
// its own sort and walks the lender's entries of that group instead.  valid stays false when the lender did not
// sort (no terms, or a direct-table sum): the borrower is then launched without it.
struct MsmSortShare {
  bool valid = false;
  const void* rec = nullptr;      // uint2 entries, bucket-sorted, unit-scalar entries behind them
  const uint32_t* off = nullptr;  // nb + G + 1 offsets into rec
  uint32_t nb = 0, c = 0, f = 0, B = 0;
  uint32_t group = 0;             // the lender's group the borrower sums (set by the caller before the launch)
  uint32_t term_start = 0, term_count = 0;  // that group's range of the lender's term list
  hipEvent_t sorted = nullptr;    // set by the caller; recorded by the lender
};
template <class F>
zg16_status msm_launch(zg16_ctx* ctx, const MsmStreams& ms, int scratch_slot, void* pinned, const MsmJob<F>* jobs,
                       int njobs, int ngroups, MsmPending* pend, const MsmWorkspace* fixed = nullptr,
                       MsmSortShare* lend = nullptr, const MsmSortShare* borrow = nullptr);
template <class F>
zg16_status msm_collect(zg16_ctx* ctx, const MsmPending* pend, XYZZ<F>* out);

// Looks up this GPU's copy of the static workspace behind zg16_msm_resident_noalloc (once per device; the
// context's device must be current).
zg16_status msm_noalloc_init(zg16_ctx* ctx);

void cache_free(zg16_ctx* ctx);  // cache.hip: releases every cached handle (the context is going away)
void default_ctx_forget(zg16_ctx* ctx);  // cache.hip: zg16_default_ctx stops handing this context out
void ntt_free_domains(zg16_ctx* ctx);
void ntt_free_cross_tables(zg16_ctx* ctx);
void ntt_free_hshard_tables(zg16_ctx* ctx);
struct HShardTables;
// A vector held as chunks of 2^log_blk elements, consecutive chunks `stride` elements apart (stride 0: contiguous).
struct ChunkLayout {
  uint32_t log_blk;
  size_t stride;
};
zg16_status hshard_tables(zg16_ctx* ctx, unsigned log_m, unsigned log_g, unsigned shard, const HShardTables** out);
zg16_status hshard_local_ntt_batch(zg16_ctx* ctx, unsigned log_l, const HShardTables* t, int nvec, const Fr* const* src,
                                   Fr* const* dst, bool inverse, const int* which, const Fr* const* subvec,
                                   const ChunkLayout* src_chunks, const ChunkLayout* dst_chunks);
zg16_status hshard_local_ntt(zg16_ctx* ctx, unsigned log_l, const HShardTables* t, const Fr* src, Fr* dst, bool inverse,
                             int which, const Fr* subvec, const ChunkLayout* src_chunks = nullptr,
                             const ChunkLayout* dst_chunks = nullptr);
zg16_status mul_vec_device(zg16_ctx* ctx, Fr* a, const Fr* b, size_t n);
// out[i] = c * base^i, i < count (device memory, the context's stream)
zg16_status pow_table_device(zg16_ctx* ctx, Fr* out, size_t count, const Fr& base, const Fr& c);
// The cross-GPU step of a 2^log_n-point transform sharded over 2^log_g GPUs (see ntt.hip), in place on
// this shard's 2^(log_n - log_g) elements.
zg16_status ntt_cross_device(zg16_ctx* ctx, Fr* data, unsigned log_n, unsigned log_g, unsigned shard, bool inverse);
zg16_status ntt_cross_product(zg16_ctx* ctx, const Fr* a, const Fr* b, size_t in_stride, Fr* out, size_t out_stride,
                              unsigned log_n, unsigned log_g, unsigned shard);
zg16_status ntt_cross_chunked(zg16_ctx* ctx, const Fr* in, size_t in_stride, Fr* out, size_t out_stride, unsigned log_n,
                              unsigned log_g, unsigned shard, bool inverse, const Fr* mulvec, int nvec = 1,
                              size_t in_vstride = 0, size_t out_vstride = 0);

}  // namespace zg
