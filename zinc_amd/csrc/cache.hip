// Keys and constraint matrices kept resident between proofs, and the process-wide default context.
//
// zinc-vm proves through `create_random_proof(circuit, &params, rng)` once per `zvm prove` run
// (zinc-vm/src/zvm/arguments/command/prove.rs:73-88) or, in zinc-tester, once per test case with
// `params.clone()` (zinc-tester/src/runners/proof_check.rs:122, :181).  Neither caller has a place to keep a
// device handle, and expanding a key into window tables costs far more than a proof (0.4 s against 4 ms at 2^20
// rows), so the library keeps what it has uploaded: a call that shows a key or a circuit it has seen before gets the
// resident handle back.  The structure of a circuit does not depend on its inputs
// (zinc-vm/src/constraint_systems/dedup.rs:30-69 canonicalises rows of a fixed shape), so a caller that names its
// circuit -- a digest of the bytecode and the method -- can also skip recording the matrices on later proofs.
#include <string.h>

#include <algorithm>
#include <mutex>

#include "r1cs.hpp"

namespace zg {

struct Digest {
  uint64_t a = 0, b = 0;
  bool operator==(const Digest& o) const { return a == o.a && b == o.b; }
};

// Two independent 64-bit multiply-rotate lanes over 8-byte words (tail bytes zero-padded); not cryptographic, the
// cache only needs accidental collisions to be out of reach.
struct Hasher {
  uint64_t a = 0x9e3779b97f4a7c15ull, b = 0xc2b2ae3d27d4eb4full;
  void word(uint64_t w) {
    a = (a ^ w) * 0xff51afd7ed558ccdull;
    a = (a << 29) | (a >> 35);
    b = (b + w) * 0xc4ceb9fe1a85ec53ull;
    b ^= b >> 31;
  }
  void bytes(const void* p, size_t n) {
    const unsigned char* c = (const unsigned char*)p;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
      uint64_t w;
      memcpy(&w, c + i, 8);
      word(w);
    }
    if (i < n) {
      uint64_t w = 0;
      memcpy(&w, c + i, n - i);
      word(w);
    }
    word(n);
  }
  Digest done() const { return {a ^ (b >> 7), b ^ (a << 11)}; }
};

struct CachedKey {
  Digest id;
  zg16_pk* pk;
  uint64_t last_use;
};
struct CachedCircuit {
  Digest id;
  zg16_r1cs* r;
  uint64_t last_use;
};

struct HandleCache {
  std::vector<CachedKey> keys;
  std::vector<CachedCircuit> circuits;
  uint64_t clock = 0;
  uint64_t key_hits = 0, key_misses = 0, circuit_hits = 0, circuit_misses = 0, evictions = 0;
};

static HandleCache* cache_of(zg16_ctx* ctx) {
  if (!ctx->cache) ctx->cache = new HandleCache();
  return ctx->cache;
}

static size_t env_limit(const char* name, size_t dflt) {
  const char* e = getenv(name);
  if (!e) return dflt;
  const long v = atol(e);
  return v < 1 ? 1 : (size_t)v;
}

// What identifies a key: the lengths of its vectors, its verifying-key points, SAMPLES evenly spaced points of
// every query and a position-weighted sum and an xor over ALL of its words (every byte through the hash when
// `full`).  Two setups of one circuit differ in every point; two circuits under one setup differ in their l, a and
// b points only where their matrices differ, which the samples can miss and the folds do not.
static constexpr size_t SAMPLES = 1024;
// Sum and xor of every 8-byte word of a query (order-free, so the pieces can be folded on several host threads):
// two keys of the same shape under the same setup that differ in a handful of points -- two circuits that differ
// in a few rows share every sampled point with high probability -- still get different digests.  Memory-bound:
// the 0.4 GB key of a 2^20-row circuit folds in a few milliseconds.
static void fold_words(const void* p, size_t bytes, uint64_t* sum, uint64_t* x) {
  const size_t words = bytes / 8;
  const uint64_t* w = (const uint64_t*)p;
  std::mutex mu;
  uint64_t s_all = 0, x_all = 0;
  host_parallel(words, [&](size_t lo, size_t hi) {
    uint64_t s0 = 0, x0 = 0;
    for (size_t i = lo; i < hi; i++) {
      s0 += w[i] * (2 * (uint64_t)(i & 0xffff) + 1);  // position-weighted (mod 2^16), so swapped points differ too
      x0 ^= w[i];
    }
    std::lock_guard<std::mutex> lock(mu);
    s_all += s0;
    x_all ^= x0;
  });
  *sum = s_all;
  *x = x_all;
}
static Digest key_digest(const zg16_pk_view* v, bool full) {
  Hasher h;
  h.bytes(v->alpha_g1, 64); h.bytes(v->beta_g1, 64); h.bytes(v->beta_g2, 128);
  h.bytes(v->delta_g1, 64); h.bytes(v->delta_g2, 128);
  const struct { const uint64_t* p; size_t n, width; } q[5] = {
      {v->h, v->n_h, 64}, {v->l, v->n_l, 64}, {v->a, v->n_a, 64}, {v->b_g1, v->n_b_g1, 64}, {v->b_g2, v->n_b_g2, 128}};
  for (const auto& x : q) {
    h.word(x.n);
    if (!x.n || !x.p) continue;
    const char* base = (const char*)x.p;
    if (full || x.n <= SAMPLES) {
      h.bytes(base, x.n * x.width);
    } else {
      for (size_t k = 0; k < SAMPLES; k++) h.bytes(base + ((x.n - 1) * k / (SAMPLES - 1)) * x.width, x.width);
      uint64_t sum, xr;
      fold_words(base, x.n * x.width, &sum, &xr);
      h.word(sum);
      h.word(xr);
    }
  }
  return h.done();
}

static Digest circuit_digest(const zg16_r1cs_view* v) {
  Hasher h;
  h.word(v->n_rows); h.word(v->n_inputs); h.word(v->n_aux);
  for (const zg16_csr* m : {&v->a, &v->b, &v->c}) {
    if (!m->row_ptr) continue;  // zg16_r1cs_upload reports it
    const size_t nnz = (size_t)m->row_ptr[v->n_rows];
    h.bytes(m->row_ptr, (v->n_rows + 1) * 8);
    if (nnz && m->col) h.bytes(m->col, nnz * 4);
    if (nnz && m->coeff) h.bytes(m->coeff, nnz * 32);
  }
  return h.done();
}

// Drop the least recently used entries until at most `keep` are left.
template <class E, class Free>
static void trim(std::vector<E>& v, size_t keep, HandleCache* hc, Free release) {
  while (v.size() > keep) {
    size_t lru = 0;
    for (size_t i = 1; i < v.size(); i++)
      if (v[i].last_use < v[lru].last_use) lru = i;
    release(v[lru]);
    v.erase(v.begin() + lru);
    hc->evictions++;
  }
}

static void release_key(CachedKey& e) {
  e.pk->cached = false;
  pk_destroy(e.pk);
}
static void release_circuit(CachedCircuit& e) {
  e.r->cached = false;
  r1cs_destroy(e.r);
}

void cache_free(zg16_ctx* ctx) {
  if (!ctx->cache) return;
  for (auto& e : ctx->cache->keys) release_key(e);
  for (auto& e : ctx->cache->circuits) release_circuit(e);
  delete ctx->cache;
  ctx->cache = nullptr;
}

// One context per GPU that lives until zg16_shutdown (or the end of the process).
static std::mutex g_default_mutex;
static std::map<int, zg16_ctx*> g_default_ctx;


// A default context destroyed by hand must not be handed out again.
void default_ctx_forget(zg16_ctx* ctx) {
  std::lock_guard<std::mutex> lock(g_default_mutex);
  for (auto it = g_default_ctx.begin(); it != g_default_ctx.end(); ++it)
    if (it->second == ctx) {
      g_default_ctx.erase(it);
      return;
    }
}

}  // namespace zg

using namespace zg;

extern "C" {

zg16_status zg16_default_ctx(int device_id, zg16_ctx** out) {
  if (!out) return ZG16_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_default_mutex);
    auto it = g_default_ctx.find(device_id);
    if (it != g_default_ctx.end()) {
      *out = it->second;
      return ZG16_OK;
    }
  }
  // Created outside the lock: zg16_ctx_create's failure path goes through zg16_ctx_destroy, which takes the lock
  // itself (default_ctx_forget).  Two threads that race here both create one; the loser's is destroyed.
  zg16_ctx* ctx = nullptr;
  const zg16_status st = zg16_ctx_create(device_id, &ctx);
  if (st != ZG16_OK) return st;
  zg16_ctx* mine = ctx;
  {
    std::lock_guard<std::mutex> lock(g_default_mutex);
    auto ins = g_default_ctx.emplace(device_id, ctx);
    ctx = ins.first->second;
  }
  if (ctx != mine) zg16_ctx_destroy(mine);
  *out = ctx;
  return ZG16_OK;
}

void zg16_shutdown(void) {
  std::map<int, zg16_ctx*> all;
  {
    std::lock_guard<std::mutex> lock(g_default_mutex);
    all.swap(g_default_ctx);
  }
  for (auto& kv : all) zg16_ctx_destroy(kv.second);  // (zg16_ctx_destroy takes the lock itself to forget the context)
}

zg16_status zg16_pk_cached(zg16_ctx* ctx, const zg16_pk_view* view, int full_digest, zg16_pk** out, int* was_cached) {
  if (!ctx || !view || !out) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad arguments to zg16_pk_cached");
  *out = nullptr;
  if (was_cached) *was_cached = 0;
  if (!view->alpha_g1 || !view->beta_g1 || !view->beta_g2 || !view->delta_g1 || !view->delta_g2)
    return fail(ctx, ZG16_ERR_MALFORMED_KEY, "verifying-key points missing");
  HandleCache* hc = cache_of(ctx);
  const Digest id = key_digest(view, full_digest != 0);
  for (auto& e : hc->keys)
    if (e.id == id) {
      e.last_use = ++hc->clock;
      hc->key_hits++;
      *out = e.pk;
      if (was_cached) *was_cached = 1;
      return ZG16_OK;
    }
  hc->key_misses++;
  const size_t limit = env_limit("ZG16_CACHE_KEYS", 4);
  trim(hc->keys, limit - 1, hc, release_key);
  zg16_pk* pk = nullptr;
  zg16_status st = zg16_pk_upload(ctx, view, 0, 1, &pk);
  if (st == ZG16_ERR_OUT_OF_MEMORY && !hc->keys.empty()) {
    // the older keys may be what fills the GPU: give their memory back and try once more
    trim(hc->keys, 0, hc, release_key);
    st = zg16_pk_upload(ctx, view, 0, 1, &pk);
  }
  if (st != ZG16_OK) return st;
  pk->cached = true;
  hc->keys.push_back({id, pk, ++hc->clock});
  *out = pk;
  return ZG16_OK;
}

uint64_t zg16_pk_id(const zg16_pk* pk) { return pk ? pk->id : 0; }

int zg16_pk_cache_holds(zg16_ctx* ctx, const zg16_pk* pk, uint64_t id) {
  if (!ctx || !pk || !id || !ctx->cache) return 0;
  // `pk` may be a stale address (its key evicted, the address reused by a newer key): only the cache's own, live
  // entries are dereferenced, and the address counts only together with the serial number it had when handed out
  for (auto& e : ctx->cache->keys)
    if (e.pk == pk && e.pk->id == id) {
      e.last_use = ++ctx->cache->clock;
      ctx->cache->key_hits++;
      return 1;
    }
  return 0;
}

zg16_status zg16_r1cs_cached(zg16_ctx* ctx, const uint8_t* circuit_id, size_t circuit_id_len, const zg16_r1cs_view* view,
                             zg16_r1cs** out, int* was_cached) {
  if (!ctx || !out || (!circuit_id && !view) || (circuit_id && !circuit_id_len))
    return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad arguments to zg16_r1cs_cached");
  *out = nullptr;
  if (was_cached) *was_cached = 0;
  HandleCache* hc = cache_of(ctx);
  Digest id;
  if (circuit_id) {
    Hasher h;
    h.bytes(circuit_id, circuit_id_len);
    id = h.done();
    id.b ^= 0x5a5a5a5a5a5a5a5aull;  // named circuits and content digests live in separate halves of the key space
  } else {
    id = circuit_digest(view);
  }
  for (auto& e : hc->circuits)
    if (e.id == id) {
      e.last_use = ++hc->clock;
      hc->circuit_hits++;
      *out = e.r;
      if (was_cached) *was_cached = 1;
      return ZG16_OK;
    }
  if (!view) return ZG16_OK;  // a look-up only: *out stays NULL, the caller records the matrices and calls again
  hc->circuit_misses++;
  trim(hc->circuits, env_limit("ZG16_CACHE_CIRCUITS", 16) - 1, hc, release_circuit);
  zg16_r1cs* r = nullptr;
  zg16_status st = zg16_r1cs_upload(ctx, view, &r);
  if (st == ZG16_ERR_OUT_OF_MEMORY && !hc->circuits.empty()) {
    trim(hc->circuits, 0, hc, release_circuit);
    st = zg16_r1cs_upload(ctx, view, &r);
  }
  if (st != ZG16_OK) return st;
  r->cached = true;
  hc->circuits.push_back({id, r, ++hc->clock});
  *out = r;
  return ZG16_OK;
}

zg16_status zg16_cache_clear(zg16_ctx* ctx) {
  if (!ctx) return ZG16_ERR_INVALID_ARGUMENT;
  if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, ZG16_ERR_DEVICE, "hipSetDevice failed");
  (void)hipDeviceSynchronize();
  cache_free(ctx);
  return ZG16_OK;
}

zg16_status zg16_cache_stats(const zg16_ctx* ctx, zg16_cache_counts* out) {
  if (!ctx || !out) return ZG16_ERR_INVALID_ARGUMENT;
  *out = zg16_cache_counts();
  const HandleCache* hc = ctx->cache;
  if (!hc) return ZG16_OK;
  out->keys = hc->keys.size();
  out->circuits = hc->circuits.size();
  out->key_hits = hc->key_hits;
  out->key_misses = hc->key_misses;
  out->circuit_hits = hc->circuit_hits;
  out->circuit_misses = hc->circuit_misses;
  out->evictions = hc->evictions;
  for (const auto& e : hc->keys) out->key_bytes += e.pk->device_bytes;
  return ZG16_OK;
}

}  // extern "C"
