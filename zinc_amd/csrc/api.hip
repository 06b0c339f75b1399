// C ABI entry points (include/zinc_groth16.h) and the prove pipeline.
//
// The pipeline follows bellman_ce 0.3.0 `create_proof` stage for stage (the reference calls it at
// zinc-vm/src/core/circuit/facade.rs:153 and zinc-vm/src/core/contract/facade.rs:387):
//   row evaluation -> H coefficients -> multiexps -> assembly of A, B, C.
// Two algebraic merges keep the launch count down without changing any result: the in/aux halves
// of the a, b_g1 and b_g2 queries are summed in one multiexp each (only their sum enters A, B, C),
// and the h and l queries share one multiexp (both only enter C).
#include <string.h>

#include <atomic>
#include <chrono>
#include <memory>
#include <thread>
#include <vector>

#include "r1cs.hpp"

using namespace zg;

// A proof captured as a HIP graph: every launch, memset and result copy of the three multiexp pipelines, the row
// evaluation and the transforms, across the context's streams; replaying it hands the whole proof (~100 short
// launches) to the GPU in one call.  Opt-in: see graph_wanted for what it measures like.
struct zg::ProveGraph {
  hipGraphExec_t exec = nullptr;
  MsmPending pend_g2, pend_z, pend_h;
  uint64_t scratch_gen = 0;
  int seen = 0;        // proofs of this (key, matrices) pair so far; the second one is captured
  bool unusable = false;  // capture or instantiation failed once: stay on plain launches
};

namespace {

void drop_graphs(zg16_ctx* ctx, const void* handle) {
  for (auto it = ctx->graphs.begin(); it != ctx->graphs.end();) {
    if (!handle || it->first.first == handle || it->first.second == handle) {
      if (it->second->exec) (void)hipGraphExecDestroy(it->second->exec);
      delete it->second;
      it = ctx->graphs.erase(it);
    } else {
      ++it;
    }
  }
}

// Off unless asked for (zg16_graph_mode, or ZG16_GRAPH=1): with ROCm 7.2 on MI355X a replayed proof is SLOWER than
// the plainly launched one at every size measured (2^8 rows: 1.22 ms against 1.02 ms; 2^20: 5.99 against 4.39) --
// the runtime runs the graph's four branches with little of the overlap the streams get, and the proof is bound by
// the GPU-side chains of dependent kernels, not by the host's queueing (0.3 ms, overlapped).
bool graph_wanted(const zg16_ctx* ctx, unsigned) {
  int mode = ctx->graph_mode;
  if (mode < 0) {
    static const int env = getenv("ZG16_GRAPH") ? atoi(getenv("ZG16_GRAPH")) : 0;
    mode = env;
  }
  return mode > 0;
}

constexpr uint32_t AUX_BIT = 0x80000000u;

// Upload `count` points as window 0 of a table of `windows` windows and fill the other windows.
template <class F>
zg16_status upload_query(zg16_ctx* ctx, Affine<F>** dst, const void* src, size_t count, uint32_t table_c,
                         Affine<F>** direct, size_t* device_bytes = nullptr, uint32_t table_step = 1) {
  *dst = nullptr;
  *direct = nullptr;
  if (!count) return ZG16_OK;
  const size_t windows = table_c ? msm_table_windows(table_c, table_step) : 1;
  ZG_HIP(ctx, hipMalloc((void**)dst, windows * count * sizeof(Affine<F>)));
  if (device_bytes) *device_bytes += windows * count * sizeof(Affine<F>);
  // (in stream order: with the builders deferred, several queries are in flight on different streams)
  ZG_HIP(ctx, hipMemcpyAsync(*dst, src, count * sizeof(Affine<F>), hipMemcpyHostToDevice, ctx->stream));
  // small queries also get a direct table of every multiple a digit can ask for (built from the ABI-form points)
  zg16_status st = table_c ? msm_build_direct<F>(ctx, *dst, count, direct) : ZG16_OK;
  // fills the other windows (if any) and rewrites every point in the multiexp kernels' packed form
  if (st == ZG16_OK) st = msm_build_table<F>(ctx, *dst, count, table_c, table_step);
  // a failure must not return while the copy above may still be reading the caller's memory
  if (st != ZG16_OK) (void)hipStreamSynchronize(ctx->stream);
  return st;
}

template <class T>
zg16_status dev_alloc_copy(zg16_ctx* ctx, T** dst, const void* src, size_t count) {
  *dst = nullptr;
  if (!count) return ZG16_OK;
  ZG_HIP(ctx, hipMalloc((void**)dst, count * sizeof(T)));
  ZG_HIP(ctx, hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
  return ZG16_OK;
}

void shard_range(size_t n, int idx, int cnt, size_t* lo, size_t* hi) {
  *lo = n * (size_t)idx / (size_t)cnt;
  *hi = n * (size_t)(idx + 1) / (size_t)cnt;
}

Fr load_fr(const uint64_t* p) {
  Fr r;
  memcpy(r.v, p, 32);
  return r;
}

}  // namespace

// Device witness z = z_in || z_aux in scratch slot 3.
zg16_status zg::stage_witness(zg16_ctx* ctx, const zg16_r1cs* r, const uint64_t* z_in, const uint64_t* z_aux,
                          int on_device, Fr** z_out) {
  if (!z_in || (r->n_aux && !z_aux)) return fail(ctx, ZG16_ERR_ASSIGNMENT_MISSING, "witness pointer is NULL");
  Fr* z;
  ZG_TRY(scratch_get(ctx, 3, (r->n_in + r->n_aux) * sizeof(Fr), (void**)&z));
  const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  // A small witness in host memory: the runtime stages each copy from pageable memory on its own (two calls that
  // do not return before their bytes are staged); here both parts go into one page-locked block and leave in one
  // transfer: a 1562-row circuit's proof 0.309 -> 0.29 ms, 2^12 rows 0.48 -> 0.45 ms on the same box; no gain from
  // 2^13 rows (256 KiB) on and a loss from 2^17, so the block is that small.  ZG16_SMALL_STAGE=0 keeps the two
  // plain copies.
  static const bool small_wanted = !(getenv("ZG16_SMALL_STAGE") && atoi(getenv("ZG16_SMALL_STAGE")) == 0);
  const size_t total = (r->n_in + r->n_aux) * sizeof(Fr);
  bool small = !on_device && small_wanted && total <= zg16_ctx::SMALL_WITNESS_BYTES && !ctx->capturing;
  if (small) {  // memory the caller has page-locked itself is copied as it is: measured slower through the block
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, r->n_aux ? z_aux : z_in) == hipSuccess && attr.type == hipMemoryTypeHost) small = false;
    (void)hipGetLastError();  // (ordinary memory is "invalid value" to that query)
  }
  if (small) {
    if (!ctx->small_stage) {
      if (hipHostMalloc(&ctx->small_stage, zg16_ctx::SMALL_WITNESS_BYTES, hipHostMallocDefault) != hipSuccess ||
          hipEventCreateWithFlags(&ctx->small_stage_ev, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        if (ctx->small_stage) (void)hipHostFree(ctx->small_stage);
        ctx->small_stage = nullptr;
      }
    }
    if (ctx->small_stage) {
      // (the block's last transfer: done long ago unless the caller never waited for that proof)
      if (ctx->small_stage_busy) ZG_HIP(ctx, hipEventSynchronize(ctx->small_stage_ev));
      memcpy(ctx->small_stage, z_in, r->n_in * sizeof(Fr));
      if (r->n_aux) memcpy((char*)ctx->small_stage + r->n_in * sizeof(Fr), z_aux, r->n_aux * sizeof(Fr));
      ZG_HIP(ctx, hipMemcpyAsync(z, ctx->small_stage, total, hipMemcpyHostToDevice, ctx->stream));
      ZG_HIP(ctx, hipEventRecord(ctx->small_stage_ev, ctx->stream));
      ctx->small_stage_busy = true;
      *z_out = z;
      return ZG16_OK;
    }
  }
  ZG_HIP(ctx, hipMemcpyAsync(z, z_in, r->n_in * sizeof(Fr), kind, ctx->stream));
  if (r->n_aux) {
    if (on_device)
      ZG_HIP(ctx, hipMemcpyAsync(z + r->n_in, z_aux, r->n_aux * sizeof(Fr), kind, ctx->stream));
    else
      ZG_TRY(copy_witness_from_host(ctx, z + r->n_in, z_aux, r->n_aux * sizeof(Fr)));
  }
  *z_out = z;
  return ZG16_OK;
}

// The witness of a reference caller is a Vec<Fr>: pageable memory, which the runtime copies through staging buffers
// of its own on one thread (32 MiB at 2^20 rows: 1.4 ms, against 0.85 ms from page-locked memory).  Here the bytes
// go through the context's page-locked staging buffers on several host threads -- thread t takes chunks t, t + T, ...
// alternating between its two buffers, copies a chunk in and queues its transfer on the context's stream -- so the
// host copies and the transfers overlap.  Memory the caller has page-locked itself is copied directly, and so are
// witnesses below 64 MiB: measured on the same box, 2^24 rows (512 MiB) 71.7 -> 70.0 ms per proof, but 2^20 rows
// (32 MiB) 5.02 -> 5.30 ms -- starting the threads costs more than the runtime's single-thread staging loses there.
// ZG16_STAGE_WITNESS=0 keeps the runtime's path at every size.
zg16_status zg::copy_witness_from_host(zg16_ctx* ctx, void* dst, const void* src, size_t bytes) {
  static const bool wanted = !(getenv("ZG16_STAGE_WITNESS") && atoi(getenv("ZG16_STAGE_WITNESS")) == 0);
  constexpr int T = zg16_ctx::STAGE_THREADS;
  constexpr size_t CH = zg16_ctx::STAGE_CHUNK;
  bool direct = !wanted || bytes < 64 * CH || ctx->capturing;
  if (!direct) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, src) == hipSuccess && attr.type == hipMemoryTypeHost) direct = true;  // page-locked already
    (void)hipGetLastError();  // (ordinary memory is "invalid value" to that query)
  }
  if (!direct && !ctx->stage) {
    if (hipHostMalloc(&ctx->stage, 2 * T * CH, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      ctx->stage = nullptr;
      direct = true;
    } else {
      for (auto& e : ctx->stage_ev) ZG_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
  }
  if (direct) {
    ZG_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return ZG16_OK;
  }
  const size_t nchunks = (bytes + CH - 1) / CH;
  std::atomic<int> failed{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < T; t++)
    pool.emplace_back([=, &failed] {
      if (hipSetDevice(ctx->device) != hipSuccess) failed = 1;
      for (size_t i = t; i < nchunks && !failed; i += T) {
        const int slot = 2 * t + (int)((i / T) & 1);
        char* buf = (char*)ctx->stage + (size_t)slot * CH;
        if (ctx->stage_busy[slot] && hipEventSynchronize(ctx->stage_ev[slot]) != hipSuccess) failed = 1;
        const size_t off = i * CH, n = bytes - off < CH ? bytes - off : CH;
        memcpy(buf, (const char*)src + off, n);
        if (hipMemcpyAsync((char*)dst + off, buf, n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipEventRecord(ctx->stage_ev[slot], ctx->stream) != hipSuccess)
          failed = 1;
        ctx->stage_busy[slot] = true;
      }
    });
  for (auto& th : pool) th.join();
  if (failed) {
    (void)hipGetLastError();
    return fail(ctx, ZG16_ERR_DEVICE, "staged copy of the witness failed");
  }
  return ZG16_OK;
}

zg16_status zg::check_key_fits(zg16_ctx* ctx, const zg16_pk* pk, const zg16_r1cs* r) {
  const size_t m = (size_t)1 << r->log_m;
  if (pk->n_h + 1 < m || pk->n_l != r->n_aux || pk->n_a != r->n_a || pk->n_b1 != r->n_b || pk->n_b2 != r->n_b)
    return fail(ctx, ZG16_ERR_MALFORMED_KEY,
                "key does not fit circuit: h %zu (need %zu), l %zu (aux %zu), a %zu (need %zu), b_g1 %zu b_g2 %zu (need %zu)",
                pk->n_h, m - 1, pk->n_l, r->n_aux, pk->n_a, r->n_a, pk->n_b1, pk->n_b2, r->n_b);
  return ZG16_OK;
}

namespace {

template <class F, class RunFn>
zg16_status msm_entry(zg16_ctx* ctx, const uint64_t* bases, size_t n_bases, const uint64_t* scalars,
                             size_t n_scalars, const uint8_t* density, Affine<F>* out, RunFn run) {
  if (!ctx || !out || (n_bases && !bases) || (n_scalars && !scalars)) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad arguments to multiexp");
  ZG_HIP(ctx, hipSetDevice(ctx->device));
  std::vector<uint32_t> map;
  if (density) {
    for (size_t i = 0; i < n_scalars; i++)
      if (density[i]) map.push_back((uint32_t)i);
    if (map.size() != n_bases) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "density selects %zu scalars for %zu bases", map.size(), n_bases);
  } else if (n_bases != n_scalars) {
    return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "%zu bases for %zu scalars", n_bases, n_scalars);
  }
  char* buf;
  const size_t b_bytes = n_bases * sizeof(Affine<F>), s_bytes = n_scalars * sizeof(Fr), m_bytes = map.size() * 4;
  ZG_TRY(scratch_get(ctx, 5, b_bytes + s_bytes + m_bytes + 768, (void**)&buf));
  Affine<F>* d_b = (Affine<F>*)buf;
  Fr* d_s = (Fr*)(buf + ((b_bytes + 255) & ~(size_t)255));
  uint32_t* d_m = (uint32_t*)((char*)d_s + ((s_bytes + 255) & ~(size_t)255));
  if (n_bases) ZG_HIP(ctx, hipMemcpyAsync(d_b, bases, b_bytes, hipMemcpyHostToDevice, ctx->stream));
  if (n_scalars) ZG_HIP(ctx, hipMemcpyAsync(d_s, scalars, s_bytes, hipMemcpyHostToDevice, ctx->stream));
  if (m_bytes) ZG_HIP(ctx, hipMemcpyAsync(d_m, map.data(), m_bytes, hipMemcpyHostToDevice, ctx->stream));
  ZG_TRY(msm_build_table<F>(ctx, d_b, n_bases, 0));  // bases -> the multiexp kernels' packed form
  MsmJob<F> job = {d_b, d_s, density ? d_m : nullptr, n_bases, 0, 0, 0};
  XYZZ<F> res;
  ZG_TRY(run(ctx, &job, 1, &res));
  *out = to_affine(res);
  return ZG16_OK;
}

}  // namespace

extern "C" {

zg16_status zg16_ctx_create(int device_id, zg16_ctx** out) {
  if (!out) return ZG16_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device_id < 0 || device_id >= count) return ZG16_ERR_DEVICE;
  if (hipSetDevice(device_id) != hipSuccess) return ZG16_ERR_DEVICE;
  zg16_ctx* ctx = new zg16_ctx();
  ctx->device = device_id;
  if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->early_stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return ZG16_ERR_DEVICE;
  }
  ctx->stream = ctx->own_stream;
  if (hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return ZG16_ERR_DEVICE;
  }
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0)
    ctx->cu_count = cus;
  if (hipHostMalloc(&ctx->pinned, 3 * MSM_PINNED_BYTES, hipHostMallocDefault) != hipSuccess) {
    delete ctx;
    return ZG16_ERR_DEVICE;
  }
  for (auto& e : ctx->ev)
    if (hipEventCreate(&e) != hipSuccess) {
      delete ctx;
      return ZG16_ERR_DEVICE;
    }
  if (msm_noalloc_init(ctx) != ZG16_OK) {
    zg16_ctx_destroy(ctx);
    return ZG16_ERR_DEVICE;
  }
  *out = ctx;
  return ZG16_OK;
}

void zg16_ctx_destroy(zg16_ctx* ctx) {
  if (!ctx) return;
  default_ctx_forget(ctx);
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  cache_free(ctx);
  drop_graphs(ctx, nullptr);
  ntt_free_domains(ctx);
  ntt_free_cross_tables(ctx);
  ntt_free_hshard_tables(ctx);
  for (auto& s : ctx->scratch)
    if (s) (void)hipFree(s);
  for (auto& t : ctx->fb_table)
    if (t) (void)hipFree(t);
  for (auto& e : ctx->ev)
    if (e) (void)hipEventDestroy(e);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
  if (ctx->early_stream) (void)hipStreamDestroy(ctx->early_stream);
  if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->stage) (void)hipHostFree(ctx->stage);
  if (ctx->small_stage) (void)hipHostFree(ctx->small_stage);
  if (ctx->small_stage_ev) (void)hipEventDestroy(ctx->small_stage_ev);
  for (auto& e : ctx->stage_ev)
    if (e) (void)hipEventDestroy(e);
  delete ctx;
}

const char* zg16_last_error(const zg16_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

zg16_status zg16_ctx_set_stream(zg16_ctx* ctx, void* hip_stream) {
  if (!ctx) return ZG16_ERR_INVALID_ARGUMENT;
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  return ZG16_OK;
}

zg16_status zg16_ctx_get_stream(const zg16_ctx* ctx, void** hip_stream) {
  if (!ctx || !hip_stream) return ZG16_ERR_INVALID_ARGUMENT;
  *hip_stream = (void*)ctx->stream;
  return ZG16_OK;
}

// ---------------------------------------------------------------------------------- proving key
zg16_status zg16_pk_upload(zg16_ctx* ctx, const zg16_pk_view* v, int shard_index, int shard_count, zg16_pk** out) {
  return zg::pk_upload_impl(ctx, v, shard_index, shard_count, false, out);
}

zg16_status zg16_pk_upload_cyclic_h(zg16_ctx* ctx, const zg16_pk_view* v, int shard_index, int shard_count, zg16_pk** out) {
  return zg::pk_upload_impl(ctx, v, shard_index, shard_count, true, out);
}
}  // extern "C"

zg16_status zg::pk_upload_impl(zg16_ctx* ctx, const zg16_pk_view* v, int shard_index, int shard_count, bool h_cyclic,
                               zg16_pk** out) {
  if (!ctx || !v || !out || shard_count < 1 || shard_index < 0 || shard_index >= shard_count)
    return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad arguments to zg16_pk_upload");
  *out = nullptr;
  if (!v->alpha_g1 || !v->beta_g1 || !v->beta_g2 || !v->delta_g1 || !v->delta_g2)
    return fail(ctx, ZG16_ERR_MALFORMED_KEY, "verifying-key points missing");
  if (v->n_b_g1 != v->n_b_g2) return fail(ctx, ZG16_ERR_MALFORMED_KEY, "b_g1 and b_g2 lengths differ");
  if ((v->n_h && !v->h) || (v->n_l && !v->l) || (v->n_a && !v->a) || (v->n_b_g1 && !v->b_g1) || (v->n_b_g2 && !v->b_g2))
    return fail(ctx, ZG16_ERR_MALFORMED_KEY, "a query vector has a length but no points");
  ZG_HIP(ctx, hipSetDevice(ctx->device));
  std::unique_ptr<zg16_pk, void (*)(zg16_pk*)> pk(new zg16_pk(), zg::pk_destroy);
  pk->ctx = ctx;
  pk->shard_index = shard_index;
  pk->shard_count = shard_count;
  pk->n_h = v->n_h; pk->n_l = v->n_l; pk->n_a = v->n_a; pk->n_b1 = v->n_b_g1; pk->n_b2 = v->n_b_g2;
  shard_range(v->n_h, shard_index, shard_count, &pk->lo_h, &pk->hi_h);
  // h_cyclic: this shard holds h[shard_index + shard_count * j] at slot j instead of a contiguous range -- the layout
  // the H coefficients come out in when the transforms are spread over the GPUs (shard.hip)
  std::vector<uint64_t> h_gathered;
  if (h_cyclic) {
    const size_t s = (size_t)shard_index, G = (size_t)shard_count;
    const size_t n_local = v->n_h > s ? (v->n_h - s + G - 1) / G : 0;
    h_gathered.resize(8 * n_local);
    for (size_t j = 0; j < n_local; j++) memcpy(&h_gathered[8 * j], v->h + 8 * (s + G * j), 64);
    pk->h_cyclic = true;
    pk->lo_h = 0;
    pk->hi_h = n_local;
  }
  shard_range(v->n_l, shard_index, shard_count, &pk->lo_l, &pk->hi_l);
  shard_range(v->n_a, shard_index, shard_count, &pk->lo_a, &pk->hi_a);
  shard_range(v->n_b_g1, shard_index, shard_count, &pk->lo_b, &pk->hi_b);
  // Window tables 2^(c*w) * P for every query: the multiexps then share one bucket set across windows.
  size_t largest = pk->hi_l - pk->lo_l;
  for (size_t q : {pk->hi_a - pk->lo_a, pk->hi_b - pk->lo_b}) largest = q > largest ? q : largest;
  pk->table_c = msm_table_window(largest, 1);
  pk->table_c_h = msm_table_window(pk->hi_h - pk->lo_h, 0);
  pk->table_c_g2 = msm_table_window(pk->hi_b - pk->lo_b, 2);
  {
    const TableQuery tq[5] = {{pk->hi_h - pk->lo_h, pk->table_c_h, 64}, {pk->hi_l - pk->lo_l, pk->table_c, 64},
                              {pk->hi_a - pk->lo_a, pk->table_c, 64}, {pk->hi_b - pk->lo_b, pk->table_c, 64},
                              {pk->hi_b - pk->lo_b, pk->table_c_g2, 128}};
    pk->table_step = msm_table_step(tq, 5);
  }
  // The tables of a small key are chains of short kernels on a handful of wavefronts (1-2 ms of latency each, ten of
  // them per query): the five queries are built side by side on the context's four streams.  Large keys fill the
  // GPU one query at a time and are built in turn.
  size_t most = pk->hi_h - pk->lo_h;
  for (size_t q : {pk->hi_l - pk->lo_l, pk->hi_a - pk->lo_a, pk->hi_b - pk->lo_b}) most = q > most ? q : most;
  const bool side_by_side = most <= ((size_t)1 << 16) && ctx->stream == ctx->own_stream && !ctx->deferred_frees;
  std::vector<void*> scratch_to_free;
  hipStream_t const own = ctx->stream;
  auto on = [&](hipStream_t s) { ctx->stream = side_by_side ? s : own; };
  auto upload_all = [&]() -> zg16_status {
    on(ctx->aux_stream);
    ZG_TRY(upload_query<Fq2>(ctx, &pk->b2, v->b_g2 + 16 * pk->lo_b, pk->hi_b - pk->lo_b, pk->table_c_g2, &pk->b2_d, &pk->device_bytes, pk->table_step));
    on(own);
    ZG_TRY(upload_query<Fq>(ctx, &pk->h, h_cyclic ? h_gathered.data() : v->h + 8 * pk->lo_h, pk->hi_h - pk->lo_h, pk->table_c_h,
                            &pk->h_d, &pk->device_bytes, pk->table_step));
    on(ctx->early_stream);
    ZG_TRY(upload_query<Fq>(ctx, &pk->l, v->l + 8 * pk->lo_l, pk->hi_l - pk->lo_l, pk->table_c, &pk->l_d, &pk->device_bytes, pk->table_step));
    on(ctx->side_stream);
    ZG_TRY(upload_query<Fq>(ctx, &pk->a, v->a + 8 * pk->lo_a, pk->hi_a - pk->lo_a, pk->table_c, &pk->a_d, &pk->device_bytes, pk->table_step));
    on(own);
    ZG_TRY(upload_query<Fq>(ctx, &pk->b1, v->b_g1 + 8 * pk->lo_b, pk->hi_b - pk->lo_b, pk->table_c, &pk->b1_d, &pk->device_bytes, pk->table_step));
    return ZG16_OK;
  };
  if (side_by_side) ctx->deferred_frees = &scratch_to_free;
  const zg16_status up = upload_all();
  ctx->stream = own;
  ctx->deferred_frees = nullptr;
  hipError_t drained = hipSuccess;
  if (side_by_side) {
    drained = hipDeviceSynchronize();  // every stream, also after a failure: the scratch below is still in use until then
    for (void* p : scratch_to_free) (void)hipFree(p);
  }
  ZG_TRY(up);
  ZG_HIP(ctx, drained);
  memcpy(&pk->alpha_g1, v->alpha_g1, 64);
  memcpy(&pk->beta_g1, v->beta_g1, 64);
  memcpy(&pk->delta_g1, v->delta_g1, 64);
  memcpy(&pk->beta_g2, v->beta_g2, 128);
  memcpy(&pk->delta_g2, v->delta_g2, 128);
  pk->build_delta_tables();
  *out = pk.release();
  return ZG16_OK;
}

extern "C" {

void zg16_pk_free(zg16_pk* pk) {
  if (!pk || pk->cached) return;  // a cached key belongs to its context (zg16_cache_clear / zg16_ctx_destroy)
  zg::pk_destroy(pk);
}
}  // extern "C"

void zg::pk_destroy(zg16_pk* pk) {
  if (!pk) return;
  if (pk->ctx) {
    (void)hipSetDevice(pk->ctx->device);
    drop_graphs(pk->ctx, pk);
  }
  (void)hipFree(pk->h); (void)hipFree(pk->l); (void)hipFree(pk->a); (void)hipFree(pk->b1); (void)hipFree(pk->b2);
  (void)hipFree(pk->h_d); (void)hipFree(pk->l_d); (void)hipFree(pk->a_d); (void)hipFree(pk->b1_d); (void)hipFree(pk->b2_d);
  delete pk;
}

extern "C" {

zg16_status zg16_pk_table_info(const zg16_pk* pk, unsigned* width_h, unsigned* width_witness, unsigned* width_g2,
                               unsigned* step, size_t* device_bytes) {
  if (!pk) return ZG16_ERR_INVALID_ARGUMENT;
  if (width_h) *width_h = pk->table_c_h;
  if (width_witness) *width_witness = pk->table_c;
  if (width_g2) *width_g2 = pk->table_c_g2;
  if (step) *step = pk->table_step;
  if (device_bytes) *device_bytes = pk->device_bytes;
  return ZG16_OK;
}

// --------------------------------------------------------------------------- constraint matrices
zg16_status zg16_r1cs_upload(zg16_ctx* ctx, const zg16_r1cs_view* v, zg16_r1cs** out) {
  if (!ctx || !v || !out) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad arguments to zg16_r1cs_upload");
  *out = nullptr;
  if (v->n_inputs < 1) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "n_inputs must include the constant ONE");
  if (v->n_inputs + v->n_aux >= 0x7fffffffull || v->n_inputs >= AUX_BIT || v->n_aux >= AUX_BIT)
    return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "too many variables");
  const unsigned log_m = ceil_log2(v->n_rows ? v->n_rows : 1);
  // bellman's EvaluationDomain::from_coeffs gives up once the exponent reaches Fr's two-adicity
  if (log_m >= 28) return fail(ctx, ZG16_ERR_POLYNOMIAL_DEGREE_TOO_LARGE, "%zu rows need a domain of 2^%u", v->n_rows, log_m);
  ZG_HIP(ctx, hipSetDevice(ctx->device));
  std::unique_ptr<zg16_r1cs, void (*)(zg16_r1cs*)> r(new zg16_r1cs(), zg::r1cs_destroy);
  r->ctx = ctx;
  r->n_rows = v->n_rows; r->n_in = v->n_inputs; r->n_aux = v->n_aux; r->log_m = log_m;
  const zg16_csr* mats[3] = {&v->a, &v->b, &v->c};
  std::vector<uint8_t> dens_a(v->n_aux, 0), dens_b_in(v->n_inputs, 0), dens_b_aux(v->n_aux, 0);
  for (int k = 0; k < 3; k++) {
    const zg16_csr& m = *mats[k];
    if (!m.row_ptr) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "matrix %d has no row_ptr", k);
    const size_t nnz = (size_t)m.row_ptr[v->n_rows];
    if (nnz >= 0xffffffffull) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "matrix %d has too many entries", k);
    if (nnz && (!m.col || !m.coeff)) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "matrix %d has no col/coeff", k);
    std::vector<uint32_t> rp(v->n_rows + 1), flat(nnz), longs;
    for (size_t i = 0; i <= v->n_rows; i++) {
      if (m.row_ptr[i] > nnz || (i && m.row_ptr[i] < m.row_ptr[i - 1]))
        return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "matrix %d row_ptr is not monotone", k);
      rp[i] = (uint32_t)m.row_ptr[i];
      if (i && rp[i] - rp[i - 1] > SPMV_LONG_ROW) longs.push_back((uint32_t)(i - 1));
    }
    // columns -> flat witness indices and the density marks, the entries split over host threads (marks are
    // single bytes that only ever become one, stored with relaxed atomics)
    std::atomic<int> bad_col(0);
    host_parallel(nnz, [&](size_t lo, size_t hi) {
      for (size_t e = lo; e < hi; e++) {
        const uint32_t c = m.col[e];
        if (c & AUX_BIT) {
          const uint32_t i = c & ~AUX_BIT;
          if (i >= v->n_aux) { bad_col.store(1); return; }
          flat[e] = (uint32_t)v->n_inputs + i;
          if (k == 0) __atomic_store_n(&dens_a[i], (uint8_t)1, __ATOMIC_RELAXED);
          if (k == 1) __atomic_store_n(&dens_b_aux[i], (uint8_t)1, __ATOMIC_RELAXED);
        } else {
          if (c >= v->n_inputs) { bad_col.store(1); return; }
          flat[e] = c;
          if (k == 1) __atomic_store_n(&dens_b_in[c], (uint8_t)1, __ATOMIC_RELAXED);
        }
      }
    });
    if (bad_col.load()) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "matrix %d names a variable the circuit does not have", k);
    R1csMatrix& d = r->m[k];
    d.nnz = nnz;
    d.n_long = (uint32_t)longs.size();
    ZG_TRY(dev_alloc_copy(ctx, &d.row_ptr, rp.data(), rp.size()));
    ZG_TRY(dev_alloc_copy(ctx, &d.col, flat.data(), nnz));
    {
      std::vector<uint16_t> codes;
      std::vector<Fr> table;
      bool needs_full = false;
      spmv_encode_coeffs(m.coeff, nnz, codes, table, &needs_full);
      d.n_table = (uint32_t)table.size();
      ZG_TRY(dev_alloc_copy(ctx, &d.code, codes.data(), nnz));
      ZG_TRY(dev_alloc_copy(ctx, &d.table, table.data(), table.size()));
      if (needs_full) ZG_TRY(dev_alloc_copy(ctx, &d.coeff, m.coeff, nnz));
      std::vector<uint32_t> order;
      spmv_row_order(rp.data(), v->n_rows, codes.data(), order);
      d.n_order = (uint32_t)order.size();
      ZG_TRY(dev_alloc_copy(ctx, &d.order, order.data(), order.size()));
    }
    ZG_TRY(dev_alloc_copy(ctx, &d.long_rows, longs.data(), longs.size()));
  }
  // query maps: A treats every input as dense (upstream tracks A density for aux only)
  std::vector<uint32_t> a_map, b_map;
  a_map.reserve(v->n_inputs + v->n_aux);
  for (size_t i = 0; i < v->n_inputs; i++) a_map.push_back((uint32_t)i);
  for (size_t i = 0; i < v->n_aux; i++)
    if (dens_a[i]) { a_map.push_back((uint32_t)(v->n_inputs + i)); r->n_a_aux++; }
  for (size_t i = 0; i < v->n_inputs; i++)
    if (dens_b_in[i]) { b_map.push_back((uint32_t)i); r->n_b_in++; }
  for (size_t i = 0; i < v->n_aux; i++)
    if (dens_b_aux[i]) { b_map.push_back((uint32_t)(v->n_inputs + i)); r->n_b_aux++; }
  r->n_a = a_map.size();
  r->n_b = b_map.size();
  ZG_TRY(dev_alloc_copy(ctx, &r->a_map, a_map.data(), a_map.size()));
  ZG_TRY(dev_alloc_copy(ctx, &r->b_map, b_map.data(), b_map.size()));
  *out = r.release();
  return ZG16_OK;
}

void zg16_r1cs_free(zg16_r1cs* r) {
  if (!r || r->cached) return;  // a cached circuit belongs to its context
  zg::r1cs_destroy(r);
}
}  // extern "C"

void zg::r1cs_destroy(zg16_r1cs* r) {
  if (!r) return;
  if (r->ctx) {
    (void)hipSetDevice(r->ctx->device);
    drop_graphs(r->ctx, r);
  }
  for (auto& m : r->m) {
    (void)hipFree(m.row_ptr); (void)hipFree(m.col); (void)hipFree(m.coeff); (void)hipFree(m.long_rows);
    (void)hipFree(m.code); (void)hipFree(m.table); (void)hipFree(m.order);
  }
  (void)hipFree(r->a_map); (void)hipFree(r->b_map);
  delete r;
}

extern "C" {

zg16_status zg16_r1cs_density(const zg16_r1cs* r, size_t* a_aux, size_t* b_input, size_t* b_aux) {
  if (!r) return ZG16_ERR_INVALID_ARGUMENT;
  if (a_aux) *a_aux = r->n_a_aux;
  if (b_input) *b_input = r->n_b_in;
  if (b_aux) *b_aux = r->n_b_aux;
  return ZG16_OK;
}

zg16_status zg16_r1cs_shape(const zg16_r1cs* r, size_t* n_rows, size_t* n_inputs, size_t* n_aux) {
  if (!r) return ZG16_ERR_INVALID_ARGUMENT;
  if (n_rows) *n_rows = r->n_rows;
  if (n_inputs) *n_inputs = r->n_in;
  if (n_aux) *n_aux = r->n_aux;
  return ZG16_OK;
}

// ---------------------------------------------------------------------------------------- prove
}  // extern "C"

// One GPU's share of a proof.  h_external: device array of the H coefficients (2^log_m entries) computed
// elsewhere, or null to evaluate the rows and run the transforms here.
// early: when given (single-GPU proof), its witness-side step runs on the host as soon as the a, b_g1
// and b_g2 sums are back, beside the h multiexp.
// evals: when given, the n_rows row evaluations of A, B and C computed by the caller (host memory, or
// device memory with evals_on_device); the row evaluation here is skipped.
struct RowEvals {
  const uint64_t* v[3];
  int on_device;
};
static zg16_status prove_partial_impl(zg16_ctx* ctx, const zg16_pk* pk, const zg16_r1cs* r, const uint64_t* z_in,
                                      const uint64_t* z_aux, int on_device, const Fr* h_external, zg16_partial* out,
                                      Assembly* early = nullptr, const RowEvals* evals = nullptr) {
  if (!ctx || !pk || !r || !out) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad arguments to zg16_prove_partial");
  if (pk->ctx != ctx || r->ctx != ctx) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "handles belong to another context");
  if (pk->h_cyclic) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "a key uploaded with zg16_pk_upload_cyclic_h proves through zg16_hshard_*");
  ZG_TRY(check_key_fits(ctx, pk, r));
  ZG_HIP(ctx, hipSetDevice(ctx->device));
  zg16_timings& tm = ctx->timings;
  tm = zg16_timings();
  // ZG16_HOST_TRACE=1: host-side timestamps (us since entry) of the enqueue and wait points, on stderr
  static const bool host_trace = getenv("ZG16_HOST_TRACE") && atoi(getenv("ZG16_HOST_TRACE")) != 0;
  const auto t_entry = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (host_trace)
      fprintf(stderr, "[zg16 host] %8.1f us  %s\n",
              std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_entry).count(), what);
  };
  // Stage boundaries are events; nothing below waits on the host until every kernel is enqueued.
  enum { E_BEGIN, E_Z, E_SPMV, E_NTT, E_G1, E_G2_BEGIN, E_G2_END, E_END, E_Z_END = 14, E_GRAPH_END = 15 };
  hipEvent_t* ev = ctx->ev;
  // ZG16_SERIAL=1 puts every stage on the main stream: slower, but per-kernel timings are then free of
  // contention between the pipelines (used for the uncontended roofline figures)
  static const bool serial_env = getenv("ZG16_SERIAL") && atoi(getenv("ZG16_SERIAL")) != 0;
  const bool serial = ctx->serial >= 0 ? ctx->serial != 0 : serial_env;
  hipStream_t st = ctx->stream, aux = serial ? st : ctx->aux_stream;
  const size_t m = (size_t)1 << r->log_m;
  const uint32_t tc = pk->table_c;
  const bool tables = tc != 0;
  auto stride = [&](size_t lo, size_t hi) { return tc ? hi - lo : (size_t)0; };

  ZG_HIP(ctx, hipEventRecord(ev[E_BEGIN], st));
  Fr* z;
  ZG_TRY(stage_witness(ctx, r, z_in, z_aux, on_device, &z));
  Fr *a = nullptr, *b = nullptr, *c = nullptr;
  if (!h_external) {
    Fr* abc;
    ZG_TRY(scratch_get(ctx, 4, 3 * m * sizeof(Fr), (void**)&abc));
    a = abc; b = abc + m; c = abc + 2 * m;
    // the padding up to the domain size: the row evaluation writes it (rows past the last evaluate to zero); vectors
    // that arrive evaluated are padded here
    if (evals && m > r->n_rows)
      for (Fr* p : {a, b, c}) ZG_HIP(ctx, hipMemsetAsync(p + r->n_rows, 0, (m - r->n_rows) * sizeof(Fr), st));
  }
  const Fr* h_coeffs = h_external ? h_external : a;
  MsmPending pend_g2, pend_z, pend_h;
  G1XYZZ g1[3];
  // Everything from here to E_END is queued without the host waiting on anything, so it can also be captured
  // into a graph once and replayed.
  auto enqueue = [&]() -> zg16_status {
  ZG_HIP(ctx, hipEventRecord(ev[E_Z], st));

  // Three multiexp pipelines.  The G2 multiexp and the G1 queries over the witness (l, a, b_g1) need
  // only z: they start now on their own streams and run beside the row evaluation and the transforms.
  // The h query needs the H coefficients and follows the transforms on the main stream; the
  // latency-bound ends of the early pipelines then hide under its accumulation.
  char* pinned = (char*)ctx->pinned;
  // h covers coefficients 0..m-2; this shard owns [lo_h, hi_h) clipped to that range
  const size_t hi_h = pk->hi_h < m - 1 ? pk->hi_h : m - 1;
  const size_t lo_h = pk->lo_h < hi_h ? pk->lo_h : hi_h;
  // the h table is laid out for the whole shard; skip the points below lo_h when the range was clipped
  const MsmJob<Fq> job_h = {pk->h + (lo_h - pk->lo_h), h_coeffs + lo_h, nullptr, hi_h - lo_h, stride(pk->lo_h, pk->hi_h), pk->table_c_h, 0,
                            pk->h_d ? pk->h_d + (lo_h - pk->lo_h) : nullptr, pk->table_step};
  const MsmJob<Fq> jobs_z[3] = {
      {pk->l, z + r->n_in + pk->lo_l, nullptr, pk->hi_l - pk->lo_l, stride(pk->lo_l, pk->hi_l), tc, 0, pk->l_d, pk->table_step},
      {pk->a, z, r->a_map + pk->lo_a, pk->hi_a - pk->lo_a, stride(pk->lo_a, pk->hi_a), tc, 1, pk->a_d, pk->table_step},
      {pk->b1, z, r->b_map + pk->lo_b, pk->hi_b - pk->lo_b, stride(pk->lo_b, pk->hi_b), tc, 2, pk->b1_d, pk->table_step}};
  // The main stream is the critical path (rows -> transforms -> h multiexp) and is enqueued first, so
  // the GPU is not left waiting while the host queues the other pipelines' launches.
  if (!h_external) {
    if (evals) {
      Fr* dst[3] = {a, b, c};
      for (int k = 0; k < 3; k++)
        ZG_HIP(ctx, hipMemcpyAsync(dst[k], evals->v[k], r->n_rows * sizeof(Fr),
                                   evals->on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    } else {
      ZG_TRY(spmv_blocks_device(ctx, r, z, a, b, c, 31, 0, 0, (uint32_t)m));  // every row in order, and the padding
    }
  }
  ZG_HIP(ctx, hipEventRecord(ev[E_SPMV], st));
  if (!h_external) ZG_TRY(h_pipeline_device(ctx, a, b, c, r->log_m));
  ZG_HIP(ctx, hipEventRecord(ev[E_NTT], st));
  mark("rows and transforms enqueued");
  if (early) early->start_key_terms();  // the critical path is on its way: alpha + r*delta, beta + s*delta beside the rest
  // b_g1 and b_g2 multiply the same scalars through the same map: when their tables have the same digit width the
  // witness launch sorts for both (its group 2) and the G2 launch walks that part of its entries.  The witness
  // launch is then queued first; ZG16_SHARE_SORT=0 keeps two sorts.
  static const bool share_wanted = !(getenv("ZG16_SHARE_SORT") && atoi(getenv("ZG16_SHARE_SORT")) == 0);
  MsmSortShare share;
  share.group = 2;
  share.sorted = ev[10];
  const bool try_share = tables && share_wanted && pk->table_c == pk->table_c_g2 && pk->hi_b > pk->lo_b;
  auto witness_g1 = [&]() -> zg16_status {
    if (!tables) return ZG16_OK;
    hipStream_t early = serial ? st : ctx->early_stream;
    ZG_HIP(ctx, hipStreamWaitEvent(early, ev[E_Z], 0));
    const MsmStreams ms = {early, nullptr, nullptr, nullptr};
    ZG_TRY(msm_launch<Fq>(ctx, ms, 6, pinned + 2 * MSM_PINNED_BYTES, jobs_z, 3, 3, &pend_z, nullptr, try_share ? &share : nullptr));
    ZG_HIP(ctx, hipEventRecord(ev[E_Z_END], early));
    return ZG16_OK;
  };
  if (try_share) ZG_TRY(witness_g1());
  {
    ZG_HIP(ctx, hipStreamWaitEvent(aux, ev[E_Z], 0));
    ZG_HIP(ctx, hipEventRecord(ev[E_G2_BEGIN], aux));
    MsmJob<Fq2> jb = {pk->b2, z, r->b_map + pk->lo_b, pk->hi_b - pk->lo_b, stride(pk->lo_b, pk->hi_b), pk->table_c_g2, 0, pk->b2_d, pk->table_step};
    const MsmStreams ms = {aux, nullptr, nullptr, nullptr};
    ZG_TRY(msm_launch<Fq2>(ctx, ms, 2, pinned + MSM_PINNED_BYTES, &jb, 1, 1, &pend_g2, nullptr, nullptr,
                           try_share && share.valid ? &share : nullptr));
    ZG_HIP(ctx, hipEventRecord(ev[E_G2_END], aux));
  }
  if (!try_share) ZG_TRY(witness_g1());

  if (tables) {
    const MsmStreams ms = {st, serial ? nullptr : ctx->side_stream, ev[8], ev[9]};
    ZG_TRY(msm_launch<Fq>(ctx, ms, 1, pinned, &job_h, 1, 1, &pend_h));
    ZG_HIP(ctx, hipEventRecord(ev[E_G1], st));
    ZG_HIP(ctx, hipStreamWaitEvent(st, ev[E_Z_END], 0));
  } else {
    // without window tables every result needs its own bucket sets: one blocking launch each
    G1XYZZ h_sum;
    ZG_TRY(msm_g1_device(ctx, &job_h, 1, &h_sum));
    for (int k = 0; k < 3; k++) ZG_TRY(msm_g1_device(ctx, &jobs_z[k], 1, &g1[k]));
    add_full(g1[0], h_sum);
    ZG_HIP(ctx, hipEventRecord(ev[E_G1], st));
  }
  ZG_HIP(ctx, hipStreamWaitEvent(st, ev[E_G2_END], 0));
  ZG_HIP(ctx, hipEventRecord(ev[E_END], st));
  return ZG16_OK;
  };  // enqueue

  // Graph replay: the first proof of a (key, matrices) pair runs as plain launches (and sizes every scratch arena),
  // the second is captured, later ones are one hipGraphLaunch.  Per-kernel timers need real events between the
  // launches, so profiling keeps the plain path.
  ProveGraph* pg = nullptr;
  bool graphed = false;
  if (tables && !h_external && !evals && !serial && !ctx->profile && st == ctx->own_stream && graph_wanted(ctx, r->log_m)) {
    ProveGraph*& slot = ctx->graphs[{pk, r}];
    if (!slot) slot = new ProveGraph();
    pg = slot;
  }
  if (pg && pg->exec && pg->scratch_gen != ctx->scratch_gen) {  // an arena moved since the capture
    (void)hipGraphExecDestroy(pg->exec);
    pg->exec = nullptr;
  }
  if (pg && !pg->exec && !pg->unusable && pg->seen >= 1) {
    ctx->capturing = true;
    hipGraph_t graph = nullptr;
    hipError_t ce = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    const zg16_status es = ce == hipSuccess ? enqueue() : ZG16_ERR_DEVICE;
    if (ce == hipSuccess) ce = hipStreamEndCapture(st, &graph);
    ctx->capturing = false;
    if (es == ZG16_OK && ce == hipSuccess && graph && hipGraphInstantiate(&pg->exec, graph, nullptr, nullptr, 0) == hipSuccess) {
      pg->pend_g2 = pend_g2; pg->pend_z = pend_z; pg->pend_h = pend_h;
      pg->scratch_gen = ctx->scratch_gen;
      ctx->graph_captures++;
    } else {
      (void)hipGetLastError();
      pg->exec = nullptr;
      pg->unusable = true;
      ctx->graph_failures++;
    }
    if (graph) (void)hipGraphDestroy(graph);
    mark("proof captured");
  }
  if (pg && pg->exec) {
    pend_g2 = pg->pend_g2; pend_z = pg->pend_z; pend_h = pg->pend_h;
    ZG_HIP(ctx, hipGraphLaunch(pg->exec, st));
    ZG_HIP(ctx, hipEventRecord(ev[E_GRAPH_END], st));
    graphed = true;
    ctx->graph_replays++;
  } else {
    if (pg) pg->seen++;
    const zg16_status es = enqueue();
    if (es != ZG16_OK) {
      // part of the proof may already be queued on several streams: let it drain before the caller can reuse
      // (or free) anything it touches
      (void)hipDeviceSynchronize();
      return es;
    }
  }
  mark("everything enqueued");
  // From here on the proof is in flight on several streams.  Any failure while collecting it (an identity base
  // reported by a multiexp, a failed event query) leaves through the single exit below, which lets the GPU drain
  // and returns the timers' events to the pool before the caller can reuse or free what the kernels touch.
  auto collect = [&]() -> zg16_status {
  if (host_trace && !graphed) {  // when the GPU passes each stage boundary, on the host's clock
    const int ids[] = {E_Z, E_SPMV, E_NTT, E_G2_END, E_Z_END, E_G1, E_END};
    const char* names[] = {"gpu: witness staged", "gpu: rows done", "gpu: transforms done", "gpu: G2 multiexp done",
                           "gpu: witness multiexps done", "gpu: h multiexp done", "gpu: all done"};
    for (int k = 0; k < 7; k++) {
      while (hipEventQuery(ev[ids[k]]) == hipErrorNotReady) __builtin_ia32_pause();
      mark(names[k]);
    }
  }
  G2XYZZ qb2;
  bool early_done = false;
  if (graphed) {
    // events inside a graph are only edges: the host cannot wait on the early pipelines one by one
    if (early) early->need_key_terms();
    hipError_t q;
    while ((q = hipEventQuery(ev[E_GRAPH_END])) == hipErrorNotReady) __builtin_ia32_pause();
    ZG_HIP(ctx, q);
    mark("GPU idle");
    G1XYZZ h_sum;
    ZG_TRY(msm_collect<Fq>(ctx, &pend_z, g1));
    ZG_TRY(msm_collect<Fq2>(ctx, &pend_g2, &qb2));
    ZG_TRY(msm_collect<Fq>(ctx, &pend_h, &h_sum));
    add_full(g1[0], h_sum);
    if (early) {
      early->with_g1(g1[1], g1[2]);
      early->with_g2(qb2);
    }
    memcpy(out->hl, &g1[0], sizeof g1[0]);
    memcpy(out->a, &g1[1], sizeof g1[1]);
    memcpy(out->b1, &g1[2], sizeof g1[2]);
    memcpy(out->b2, &qb2, sizeof qb2);
    ZG_HIP(ctx, hipEventElapsedTime(&tm.total_ms, ev[E_BEGIN], ev[E_GRAPH_END]));
    mark("multiexps collected");
    return ZG16_OK;
  }
  if (tables && early) {
    early->need_key_terms();  // host work while the GPU is busy (a thread of its own, unless the proof was replayed)
    mark("key terms");
    // poll: the runtime's blocking wait wakes up to 0.3 ms late, which is half of a small circuit's proof
    auto spin = [&](hipEvent_t e) {
      hipError_t q;
      while ((q = hipEventQuery(e)) == hipErrorNotReady) __builtin_ia32_pause();
      return q;
    };
    ZG_HIP(ctx, spin(ev[E_Z_END]));
    mark("witness multiexps seen");
    ZG_TRY(msm_collect<Fq>(ctx, &pend_z, g1));
    mark("witness sums folded");
    early->with_g1(g1[1], g1[2]);
    mark("witness-query multiexps collected");
    ZG_HIP(ctx, spin(ev[E_G2_END]));
    ZG_TRY(msm_collect<Fq2>(ctx, &pend_g2, &qb2));
    mark("G2 sums folded");
    early->with_g2(qb2);
    early_done = true;
    mark("G2 multiexp collected");
  }
  {
    // the proof waits on this one: poll instead of sleeping in the runtime's wait
    hipError_t q;
    while ((q = hipEventQuery(ev[E_END])) == hipErrorNotReady) __builtin_ia32_pause();
    ZG_HIP(ctx, q);
  }
  mark("GPU idle");
  if (tables) {
    G1XYZZ h_sum;
    if (!early_done) ZG_TRY(msm_collect<Fq>(ctx, &pend_z, g1));
    ZG_TRY(msm_collect<Fq>(ctx, &pend_h, &h_sum));
    add_full(g1[0], h_sum);  // h and l both only enter C
  }
  if (!early_done) {
    ZG_TRY(msm_collect<Fq2>(ctx, &pend_g2, &qb2));
    if (early) early->early(g1[1], g1[2], qb2);
  }
  memcpy(out->hl, &g1[0], sizeof g1[0]);
  memcpy(out->a, &g1[1], sizeof g1[1]);
  memcpy(out->b1, &g1[2], sizeof g1[2]);
  memcpy(out->b2, &qb2, sizeof qb2);
  mark("h multiexp collected");
  ZG_HIP(ctx, hipEventElapsedTime(&tm.upload_ms, ev[E_BEGIN], ev[E_Z]));
  ZG_HIP(ctx, hipEventElapsedTime(&tm.spmv_ms, ev[E_Z], ev[E_SPMV]));
  ZG_HIP(ctx, hipEventElapsedTime(&tm.ntt_ms, ev[E_SPMV], ev[E_NTT]));
  ZG_HIP(ctx, hipEventElapsedTime(&tm.msm_g1_ms, ev[E_NTT], ev[E_G1]));
  ZG_HIP(ctx, hipEventElapsedTime(&tm.msm_g2_ms, ev[E_G2_BEGIN], ev[E_G2_END]));  // overlaps the stages above
  ZG_HIP(ctx, hipEventElapsedTime(&tm.total_ms, ev[E_BEGIN], ev[E_END]));
  profile_collect(ctx);
  mark("timings read");
  return ZG16_OK;
  };  // collect
  const zg16_status cs = collect();
  if (cs != ZG16_OK) {
    (void)hipDeviceSynchronize();
    profile_collect(ctx);
  }
  return cs;
}

extern "C" {

zg16_status zg16_prove_partial(zg16_ctx* ctx, const zg16_pk* pk, const zg16_r1cs* r, const uint64_t* z_in,
                               const uint64_t* z_aux, int on_device, zg16_partial* out) {
  return prove_partial_impl(ctx, pk, r, z_in, z_aux, on_device, nullptr, out);
}

zg16_status zg16_prove_partial_h(zg16_ctx* ctx, const zg16_pk* pk, const zg16_r1cs* r, const uint64_t* z_in,
                                 const uint64_t* z_aux, int on_device, const uint64_t* h_dev, zg16_partial* out) {
  if (!h_dev) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "zg16_prove_partial_h needs the H coefficients");
  return prove_partial_impl(ctx, pk, r, z_in, z_aux, on_device, (const Fr*)h_dev, out);
}

// Coset evaluations of one of the three row-evaluation polynomials: rows of matrix `which` times z,
// ifft, coset_fft.  out_dev: 2^log_m elements of device memory.
zg16_status zg16_eval_chain(zg16_ctx* ctx, const zg16_r1cs* r, const uint64_t* z_in, const uint64_t* z_aux, int on_device,
                            int which, uint64_t* out_dev) {
  if (!ctx || !r || !out_dev || which < 0 || which > 2) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad arguments to zg16_eval_chain");
  if (r->ctx != ctx) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "handle belongs to another context");
  ZG_HIP(ctx, hipSetDevice(ctx->device));
  const size_t m = (size_t)1 << r->log_m;
  Fr* z;
  ZG_TRY(stage_witness(ctx, r, z_in, z_aux, on_device, &z));
  Fr* out = (Fr*)out_dev;
  if (m > r->n_rows) ZG_HIP(ctx, hipMemsetAsync(out + r->n_rows, 0, (m - r->n_rows) * sizeof(Fr), ctx->stream));
  ZG_TRY(spmv_one_device(ctx, r, which, z, out));
  ZG_TRY(coset_evals_device(ctx, out, r->log_m));
  ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZG16_OK;
}

// a_dev <- H coefficients from the three coset evaluation vectors (device memory, 2^log_m each):
// a*b - c, divide by Z on the coset, icoset_fft.  The first 2^log_m - 1 entries are the h-query scalars.
zg16_status zg16_h_from_cosets(zg16_ctx* ctx, uint64_t* a_dev, const uint64_t* b_dev, const uint64_t* c_dev, unsigned log_m) {
  if (!ctx || !a_dev || !b_dev || !c_dev) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad arguments to zg16_h_from_cosets");
  if (log_m >= 28) return fail(ctx, ZG16_ERR_POLYNOMIAL_DEGREE_TOO_LARGE, "domain 2^%u exceeds 2^27", log_m);
  ZG_HIP(ctx, hipSetDevice(ctx->device));
  ZG_TRY(h_finish_device(ctx, (Fr*)a_dev, (const Fr*)b_dev, (const Fr*)c_dev, log_m));
  ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZG16_OK;
}

unsigned zg16_r1cs_log_domain(const zg16_r1cs* r) { return r ? r->log_m : 0; }

zg16_status zg16_prove_finish(const zg16_pk_view* v, const zg16_partial* parts, size_t n_parts, const uint64_t r_[4],
                              const uint64_t s_[4], zg16_proof* out) {
  if (!v || !parts || !n_parts || !r_ || !s_ || !out) return ZG16_ERR_INVALID_ARGUMENT;
  if (!v->alpha_g1 || !v->beta_g1 || !v->beta_g2 || !v->delta_g1 || !v->delta_g2) return ZG16_ERR_MALFORMED_KEY;
  G1Affine alpha1, beta1, delta1;
  G2Affine beta2, delta2;
  memcpy(&alpha1, v->alpha_g1, 64);
  memcpy(&beta1, v->beta_g1, 64);
  memcpy(&delta1, v->delta_g1, 64);
  memcpy(&beta2, v->beta_g2, 128);
  memcpy(&delta2, v->delta_g2, 128);
  // bellman rejects a key whose delta is the identity before doing any work
  if (delta1.is_inf() || delta2.is_inf()) return ZG16_ERR_UNEXPECTED_IDENTITY;
  G1XYZZ hl = G1XYZZ::inf(), qa = G1XYZZ::inf(), qb1 = G1XYZZ::inf();
  G2XYZZ qb2 = G2XYZZ::inf();
  for (size_t i = 0; i < n_parts; i++) {
    G1XYZZ p;
    G2XYZZ p2;
    memcpy(&p, parts[i].hl, sizeof p); add_full(hl, p);
    memcpy(&p, parts[i].a, sizeof p); add_full(qa, p);
    memcpy(&p, parts[i].b1, sizeof p); add_full(qb1, p);
    memcpy(&p2, parts[i].b2, sizeof p2); add_full(qb2, p2);
  }
  Assembly as;
  as.alpha1 = alpha1; as.beta1 = beta1; as.delta1 = delta1; as.beta2 = beta2; as.delta2 = delta2;
  // r*delta_g1 and s*delta_g2 from fixed-base tables, as a key handle keeps them: the view has none, so the tables
  // of the last delta seen are kept here (a prover assembles many proofs under one key; building the pair costs
  // about as much as ten assemblies without them)
  struct DeltaTables {
    G1Affine d1;
    G2Affine d2;
    FixedBaseTable<Fq> t1;
    FixedBaseTable<Fq2> t2;
  };
  static std::mutex tables_mutex;
  static std::shared_ptr<const DeltaTables> last_tables;
  std::shared_ptr<const DeltaTables> tables;
  {
    std::lock_guard<std::mutex> lock(tables_mutex);
    if (last_tables && !memcmp(&last_tables->d1, &delta1, sizeof delta1) && !memcmp(&last_tables->d2, &delta2, sizeof delta2))
      tables = last_tables;
  }
  if (!tables) {
    auto fresh = std::make_shared<DeltaTables>();
    fresh->d1 = delta1;
    fresh->d2 = delta2;
    fresh->t1.build(delta1);
    fresh->t2.build(delta2);
    tables = fresh;
    std::lock_guard<std::mutex> lock(tables_mutex);
    last_tables = tables;
  }
  as.delta1_table = &tables->t1;
  as.delta2_table = &tables->t2;
  as.rc = from_mont(load_fr(r_));
  as.sc = from_mont(load_fr(s_));
  as.early(qa, qb1, qb2);
  as.finish(hl, out);
  return ZG16_OK;
}

static zg16_status prove_full(zg16_ctx* ctx, const zg16_pk* pk, const zg16_r1cs* r, const uint64_t* z_in,
                              const uint64_t* z_aux, int on_device, const uint64_t r_[4], const uint64_t s_[4],
                              zg16_proof* out, const RowEvals* evals) {
  if (!ctx || !pk) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad arguments to zg16_prove");
  if (pk->shard_count != 1) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "zg16_prove needs an unsharded key");
  if (pk->delta_g1.is_inf() || pk->delta_g2.is_inf())
    return fail(ctx, ZG16_ERR_UNEXPECTED_IDENTITY, "delta is the point at infinity");
  if (!r_ || !s_ || !out) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad arguments to zg16_prove");
  Assembly as;
  as.alpha1 = pk->alpha_g1; as.beta1 = pk->beta_g1; as.delta1 = pk->delta_g1;
  as.beta2 = pk->beta_g2; as.delta2 = pk->delta_g2;
  as.delta1_table = pk->delta1_table.get();
  as.delta2_table = pk->delta2_table.get();
  as.rc = from_mont(load_fr(r_));
  as.sc = from_mont(load_fr(s_));
  zg16_partial part;
  ZG_TRY(prove_partial_impl(ctx, pk, r, z_in, z_aux, on_device, nullptr, &part, &as, evals));
  auto t0 = std::chrono::steady_clock::now();
  G1XYZZ hl;
  memcpy(&hl, part.hl, sizeof hl);
  as.finish(hl, out);
  ctx->timings.finish_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  ctx->timings.total_ms += ctx->timings.finish_ms;
  return ZG16_OK;
}

zg16_status zg16_prove(zg16_ctx* ctx, const zg16_pk* pk, const zg16_r1cs* r, const uint64_t* z_in, const uint64_t* z_aux,
                       int on_device, const uint64_t r_[4], const uint64_t s_[4], zg16_proof* out) {
  return prove_full(ctx, pk, r, z_in, z_aux, on_device, r_, s_, out, nullptr);
}

zg16_status zg16_prove_from_evals(zg16_ctx* ctx, const zg16_pk* pk, const zg16_r1cs* r, const uint64_t* a, const uint64_t* b,
                                  const uint64_t* c, int evals_on_device, const uint64_t* z_in, const uint64_t* z_aux,
                                  int on_device, const uint64_t r_[4], const uint64_t s_[4], zg16_proof* out) {
  if (!a || !b || !c) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "zg16_prove_from_evals needs the three row-evaluation vectors");
  const RowEvals ev = {{a, b, c}, evals_on_device};
  return prove_full(ctx, pk, r, z_in, z_aux, on_device, r_, s_, out, &ev);
}

static const char* const kKernelNames[K_COUNT] = {
    "k_spmv_short+long", "k_ntt_pass", "k_scale_powers", "k_mul_sub_scale", "k_sort_*+k_fine_*+k_scan_*", "k_scan",
    "k_accumulate<G1>", "k_accumulate<G2>", "k_combine+k_heavy<G1>", "k_combine+k_heavy<G2>",
    "k_split_sums<G1>", "k_split_sums<G2>", "k_sum_range|k_direct_sum<G1>", "k_sum_range|k_direct_sum<G2>"};

zg16_status zg16_graph_mode(zg16_ctx* ctx, int mode) {
  if (!ctx) return ZG16_ERR_INVALID_ARGUMENT;
  ctx->graph_mode = mode < 0 ? -1 : (mode != 0);
  return ZG16_OK;
}

zg16_status zg16_graph_counts(const zg16_ctx* ctx, uint64_t* captures, uint64_t* replays, uint64_t* failures) {
  if (!ctx) return ZG16_ERR_INVALID_ARGUMENT;
  if (captures) *captures = ctx->graph_captures;
  if (replays) *replays = ctx->graph_replays;
  if (failures) *failures = ctx->graph_failures;
  return ZG16_OK;
}

zg16_status zg16_profile_enable(zg16_ctx* ctx, int on) {
  if (!ctx) return ZG16_ERR_INVALID_ARGUMENT;
  ctx->profile = on == 2 ? 2 : (on != 0);
  return ZG16_OK;
}

zg16_status zg16_profile_serial(zg16_ctx* ctx, int on) {
  if (!ctx) return ZG16_ERR_INVALID_ARGUMENT;
  ctx->serial = on < 0 ? -1 : (on != 0);
  return ZG16_OK;
}

zg16_status zg16_profile_reset(zg16_ctx* ctx) {
  if (!ctx) return ZG16_ERR_INVALID_ARGUMENT;
  for (auto& k : ctx->kstat) k = KernelStat();
  return ZG16_OK;
}

int zg16_profile_kernel_count(void) { return K_COUNT; }

zg16_status zg16_profile_get(const zg16_ctx* ctx, int kernel, const char** name, double* total_ms, uint64_t* launches,
                             double* units) {
  if (!ctx || kernel < 0 || kernel >= K_COUNT) return ZG16_ERR_INVALID_ARGUMENT;
  if (name) *name = kKernelNames[kernel];
  if (total_ms) *total_ms = ctx->kstat[kernel].ms;
  if (launches) *launches = ctx->kstat[kernel].launches;
  if (units) *units = ctx->kstat[kernel].units;
  return ZG16_OK;
}

zg16_status zg16_last_timings(const zg16_ctx* ctx, zg16_timings* out) {
  if (!ctx || !out) return ZG16_ERR_INVALID_ARGUMENT;
  *out = ctx->timings;
  return ZG16_OK;
}

// ---------------------------------------------------------------------------- stand-alone stages
zg16_status zg16_ntt(zg16_ctx* ctx, uint64_t* data, unsigned log_n, int inverse, int coset, int on_device) {
  if (!ctx || !data) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad arguments to zg16_ntt");
  if (log_n >= 28) return fail(ctx, ZG16_ERR_POLYNOMIAL_DEGREE_TOO_LARGE, "domain 2^%u exceeds 2^27", log_n);
  ZG_HIP(ctx, hipSetDevice(ctx->device));
  const size_t bytes = sizeof(Fr) << log_n;
  Fr* d = (Fr*)data;
  if (!on_device) {
    ZG_TRY(scratch_get(ctx, 4, bytes, (void**)&d));
    ZG_HIP(ctx, hipMemcpyAsync(d, data, bytes, hipMemcpyHostToDevice, ctx->stream));
  }
  ZG_TRY(ntt_device(ctx, d, log_n, inverse != 0, coset != 0));
  if (!on_device) ZG_HIP(ctx, hipMemcpyAsync(data, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
  ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZG16_OK;
}

zg16_status zg16_ntt_cross(zg16_ctx* ctx, uint64_t* data, unsigned log_n, unsigned log_shards, unsigned shard,
                           int inverse, int on_device) {
  if (!ctx || !data) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad arguments to zg16_ntt_cross");
  if (log_n >= 28) return fail(ctx, ZG16_ERR_POLYNOMIAL_DEGREE_TOO_LARGE, "domain 2^%u exceeds 2^27", log_n);
  if (log_shards > 3 || 2 * log_shards > log_n || shard >= (1u << log_shards))
    return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "zg16_ntt_cross: 2^%u points over 2^%u shards (shard %u) is not a valid split",
                log_n, log_shards, shard);
  ZG_HIP(ctx, hipSetDevice(ctx->device));
  const size_t bytes = sizeof(Fr) << (log_n - log_shards);
  Fr* d = (Fr*)data;
  if (!on_device) {
    ZG_TRY(scratch_get(ctx, 4, bytes, (void**)&d));
    ZG_HIP(ctx, hipMemcpyAsync(d, data, bytes, hipMemcpyHostToDevice, ctx->stream));
  }
  ZG_TRY(ntt_cross_device(ctx, d, log_n, log_shards, shard, inverse != 0));
  if (!on_device) ZG_HIP(ctx, hipMemcpyAsync(data, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
  ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZG16_OK;
}

zg16_status zg16_h_coefficients(zg16_ctx* ctx, const uint64_t* a, const uint64_t* b, const uint64_t* c, size_t n_rows,
                                uint64_t* h, size_t* m_out) {
  if (!ctx || !a || !b || !c || !h || !n_rows) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad arguments to zg16_h_coefficients");
  const unsigned log_m = ceil_log2(n_rows);
  if (log_m >= 28) return fail(ctx, ZG16_ERR_POLYNOMIAL_DEGREE_TOO_LARGE, "%zu rows need a domain of 2^%u", n_rows, log_m);
  ZG_HIP(ctx, hipSetDevice(ctx->device));
  const size_t m = (size_t)1 << log_m;
  Fr* abc;
  ZG_TRY(scratch_get(ctx, 4, 3 * m * sizeof(Fr), (void**)&abc));
  const uint64_t* src[3] = {a, b, c};
  for (int k = 0; k < 3; k++) {
    ZG_HIP(ctx, hipMemcpyAsync(abc + k * m, src[k], n_rows * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    if (m > n_rows) ZG_HIP(ctx, hipMemsetAsync(abc + k * m + n_rows, 0, (m - n_rows) * sizeof(Fr), ctx->stream));
  }
  ZG_TRY(h_pipeline_device(ctx, abc, abc + m, abc + 2 * m, log_m));
  if (m > 1) ZG_HIP(ctx, hipMemcpyAsync(h, abc, (m - 1) * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
  ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (m_out) *m_out = m;
  return ZG16_OK;
}

zg16_status zg16_msm_g1(zg16_ctx* ctx, const uint64_t* bases, size_t n_bases, const uint64_t* scalars, size_t n_scalars,
                        const uint8_t* density, uint64_t out[8]) {
  G1Affine res;
  zg16_status st = msm_entry<Fq>(ctx, bases, n_bases, scalars, n_scalars, density, &res, msm_g1_device);
  if (st == ZG16_OK) memcpy(out, &res, 64);
  return st;
}

zg16_status zg16_msm_g2(zg16_ctx* ctx, const uint64_t* bases, size_t n_bases, const uint64_t* scalars, size_t n_scalars,
                        const uint8_t* density, uint64_t out[16]) {
  G2Affine res;
  zg16_status st = msm_entry<Fq2>(ctx, bases, n_bases, scalars, n_scalars, density, &res, msm_g2_device);
  if (st == ZG16_OK) memcpy(out, &res, 128);
  return st;
}

zg16_status zg16_bases_upload(zg16_ctx* ctx, int group, const uint64_t* bases, size_t n, int window_tables,
                              zg16_bases** out) {
  if (!ctx || !out || (n && !bases) || (group != 1 && group != 2))
    return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad arguments to zg16_bases_upload");
  *out = nullptr;
  ZG_HIP(ctx, hipSetDevice(ctx->device));
  std::unique_ptr<zg16_bases, void (*)(zg16_bases*)> b(new zg16_bases(), zg16_bases_free);
  b->ctx = ctx;
  b->group = group;
  b->n = n;
  b->table_c = window_tables ? msm_table_window(n, 0) : 0;
  if (b->table_c) {
    const TableQuery tq = {n, b->table_c, group == 1 ? 64u : 128u};
    b->table_step = msm_table_step(&tq, 1);
  }
  // the handle's fields receive the arrays as they are allocated, so zg16_bases_free releases them on any failure
  if (group == 1)
    ZG_TRY(upload_query<Fq>(ctx, (Affine<Fq>**)&b->points, bases, n, b->table_c, (Affine<Fq>**)&b->direct, nullptr, b->table_step));
  else
    ZG_TRY(upload_query<Fq2>(ctx, (Affine<Fq2>**)&b->points, bases, n, b->table_c, (Affine<Fq2>**)&b->direct, nullptr, b->table_step));
  *out = b.release();
  return ZG16_OK;
}

void zg16_bases_free(zg16_bases* b) {
  if (!b) return;
  if (b->ctx) (void)hipSetDevice(b->ctx->device);
  (void)hipFree(b->points);
  (void)hipFree(b->direct);
  delete b;
}

zg16_status zg16_msm_resident(zg16_ctx* ctx, const zg16_bases* b, const uint64_t* scalars, int on_device, uint64_t* out) {
  if (!ctx || !b || !out || (b->n && !scalars)) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad arguments to zg16_msm_resident");
  if (b->ctx != ctx) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "handle belongs to another context");
  ZG_HIP(ctx, hipSetDevice(ctx->device));
  const Fr* d_s = (const Fr*)scalars;
  if (!on_device && b->n) {
    Fr* buf;
    ZG_TRY(scratch_get(ctx, 5, b->n * sizeof(Fr), (void**)&buf));
    ZG_HIP(ctx, hipMemcpyAsync(buf, scalars, b->n * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    d_s = buf;
  }
  const size_t stride = b->table_c ? b->n : 0;
  if (b->group == 1) {
    MsmJob<Fq> job = {(const Affine<Fq>*)b->points, d_s, nullptr, b->n, stride, b->table_c, 0, (const Affine<Fq>*)b->direct, b->table_step};
    G1XYZZ res;
    ZG_TRY(msm_g1_device(ctx, &job, 1, &res));
    const G1Affine a = to_affine(res);
    memcpy(out, &a, sizeof a);
  } else {
    MsmJob<Fq2> job = {(const Affine<Fq2>*)b->points, d_s, nullptr, b->n, stride, b->table_c, 0, (const Affine<Fq2>*)b->direct, b->table_step};
    G2XYZZ res;
    ZG_TRY(msm_g2_device(ctx, &job, 1, &res));
    const G2Affine a = to_affine(res);
    memcpy(out, &a, sizeof a);
  }
  return ZG16_OK;
}

zg16_status zg16_spmv(zg16_ctx* ctx, const zg16_r1cs* r, const uint64_t* z_in, const uint64_t* z_aux, uint64_t* a,
                      uint64_t* b, uint64_t* c) {
  if (!ctx || !r || !a || !b || !c) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad arguments to zg16_spmv");
  ZG_HIP(ctx, hipSetDevice(ctx->device));
  Fr* z;
  ZG_TRY(stage_witness(ctx, r, z_in, z_aux, 0, &z));
  Fr* abc;
  const size_t n = r->n_rows;
  ZG_TRY(scratch_get(ctx, 4, 3 * n * sizeof(Fr) + 32, (void**)&abc));
  ZG_TRY(spmv_device(ctx, r, z, abc, abc + n, abc + 2 * n));
  uint64_t* outs[3] = {a, b, c};
  for (int k = 0; k < 3; k++)
    if (n) ZG_HIP(ctx, hipMemcpyAsync(outs[k], abc + k * n, n * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
  ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZG16_OK;
}

}  // extern "C"
