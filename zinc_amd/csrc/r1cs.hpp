// Device-resident constraint matrices and proving key (internal definitions of the opaque
// handles declared in include/zinc_groth16.h).
#pragma once
#include <atomic>
#include <memory>
#include <thread>

#include "common.hpp"
#include "glv.hpp"

namespace zg {

static constexpr uint32_t SPMV_LONG_ROW = 32;  // rows with more terms get a whole wavefront

// Coefficient codes (spmv.hip): 0 = one, 1 = minus one, 2 + k / 256 + k = +-2^(k+1), SPMV_CODE_DICT + i = the i-th
// other value of the matrix; SPMV_CODE_FULL = read the term's coefficient from the full array.
static constexpr uint32_t SPMV_CODE_DICT = 512, SPMV_CODE_FULL = 0xffff;

struct R1csMatrix {
  uint32_t* row_ptr = nullptr;  // n_rows + 1
  uint32_t* col = nullptr;      // index into the flat witness z = z_in || z_aux
  uint16_t* code = nullptr;     // coefficient code per term
  Fr* table = nullptr;          // what the codes index (n_table entries)
  uint32_t n_table = 0;
  Fr* coeff = nullptr;          // full coefficients; only kept when the matrix has more distinct values than codes
  uint32_t* long_rows = nullptr;
  uint32_t n_long = 0;
  // The short rows ordered by (term count, any coefficient other than +-1), row index increasing inside a class:
  // consecutive lanes then run the same number of term rounds down the same path (spmv_row_order).  Null when the
  // matrix was built without it (the launch then takes the rows in their own order).
  uint32_t* order = nullptr;
  uint32_t n_order = 0;
  size_t nnz = 0;
};

// fn(lo, hi) over [0, n) in contiguous pieces on up to 16 host threads (one piece, inline, when n is small).
template <class Fn>
inline void host_parallel(size_t n, Fn fn) {
  unsigned t = std::thread::hardware_concurrency();
  t = t < 1 ? 1 : (t > 16 ? 16 : t);
  if (n < ((size_t)1 << 16) || t == 1) {
    fn((size_t)0, n);
    return;
  }
  std::vector<std::thread> pool;
  for (unsigned i = 0; i < t; i++) pool.emplace_back([=] { fn(n * i / t, n * (i + 1) / t); });
  for (auto& th : pool) th.join();
}

zg16_status spmv_device(zg16_ctx* ctx, const zg16_r1cs* r, const Fr* z, Fr* a, Fr* b, Fr* c);
// rows of matrix k (0 = A, 1 = B, 2 = C) only
zg16_status spmv_one_device(zg16_ctx* ctx, const zg16_r1cs* r, int k, const Fr* z, Fr* out);
// rows (j >> log_blk) * stride + base + (j mod 2^log_blk) for j < n_out (padding rows give zero)
zg16_status spmv_blocks_device(zg16_ctx* ctx, const zg16_r1cs* r, const Fr* z, Fr* a, Fr* b, Fr* c, uint32_t log_blk,
                               uint32_t stride, uint32_t base, uint32_t n_out);
// The short rows (at most SPMV_LONG_ROW terms) of a CSR matrix sorted by class; codes are spmv_encode_coeffs'.
void spmv_row_order(const uint32_t* row_ptr, size_t n_rows, const uint16_t* codes, std::vector<uint32_t>& order);
void spmv_encode_coeffs(const uint64_t* coeff, size_t nnz, std::vector<uint16_t>& codes, std::vector<Fr>& table,
                        bool* needs_full);

}  // namespace zg

namespace zg {
inline uint64_t next_handle_id() {
  static std::atomic<uint64_t> counter{0};
  return ++counter;
}
}  // namespace zg

struct zg16_r1cs {
  zg16_ctx* ctx = nullptr;
  bool cached = false;  // owned by the context's cache: zg16_r1cs_free leaves it alone
  size_t n_rows = 0, n_in = 0, n_aux = 0;
  unsigned log_m = 0;
  zg::R1csMatrix m[3];
  // Flat-z indices of the variables that own a base in the a / b queries, in query order:
  // every input then the A-dense aux variables; the B-dense inputs then the B-dense aux.
  uint32_t* a_map = nullptr;
  uint32_t* b_map = nullptr;
  size_t n_a = 0, n_b = 0;
  size_t n_a_aux = 0, n_b_in = 0, n_b_aux = 0;
};

struct zg16_pk {
  zg16_ctx* ctx = nullptr;
  bool cached = false;  // owned by the context's cache: zg16_pk_free leaves it alone
  uint64_t id = zg::next_handle_id();  // serial number, never reused in a process: tells a handle from a later one at its address
  size_t device_bytes = 0;  // HBM held by the query tables
  int shard_index = 0, shard_count = 1;
  bool h_cyclic = false;  // the h query holds h[shard_index + shard_count * j] at slot j ([lo_h, hi_h) = [0, slots))
  // totals of the full key and the [lo, hi) range this shard holds, per query
  size_t n_h = 0, n_l = 0, n_a = 0, n_b1 = 0, n_b2 = 0;
  size_t lo_h = 0, hi_h = 0, lo_l = 0, hi_l = 0, lo_a = 0, hi_a = 0, lo_b = 0, hi_b = 0;
  zg::G1Affine *h = nullptr, *l = nullptr, *a = nullptr, *b1 = nullptr;
  zg::G2Affine* b2 = nullptr;
  // Direct tables (msm_build_direct) of the queries that are small enough, else null; same point ranges.
  zg::G1Affine *h_d = nullptr, *l_d = nullptr, *a_d = nullptr, *b1_d = nullptr;
  zg::G2Affine* b2_d = nullptr;
  // When table_c != 0 each query array holds msm_table_windows(c) windows of its shard, c being the
  // query's own width below: q[w * (hi - lo) + k] = 2^(c * w) * P_(lo + k).
  uint32_t table_c = 0;       // l, a, b_g1 (one launch, one width)
  uint32_t table_c_h = 0;     // h
  uint32_t table_c_g2 = 0;    // b_g2
  // Every table keeps one window in table_step (chosen at upload so that the tables fit the memory budget:
  // msm_table_step); the multiexps then use table_step bucket sets per result.
  uint32_t table_step = 1;
  zg::G1Affine alpha_g1, beta_g1, delta_g1;
  zg::G2Affine beta_g2, delta_g2;
  // r * delta_g1 and s * delta_g2 open every proof's assembly on the host; delta is fixed per key, so the key
  // keeps fixed-base tables of both (host memory, built at upload by build_delta_tables).
  std::unique_ptr<zg::FixedBaseTable<zg::Fq>> delta1_table;
  std::unique_ptr<zg::FixedBaseTable<zg::Fq2>> delta2_table;
  void build_delta_tables() {
    if (delta_g1.is_inf() || delta_g2.is_inf()) return;  // such a key is rejected when it is used
    delta1_table.reset(new zg::FixedBaseTable<zg::Fq>());
    delta1_table->build(delta_g1);
    delta2_table.reset(new zg::FixedBaseTable<zg::Fq2>());
    delta2_table->build(delta_g2);
  }
};

namespace zg {
// Release a handle whoever owns it (zg16_pk_free / zg16_r1cs_free skip handles the cache owns).
// Device witness z = z_in || z_aux (scratch slot 3 of the context), copied on the context's stream.
// `bytes` of a witness in host memory to `dst` on the GPU, queued on the context's stream (see api.hip)
zg16_status copy_witness_from_host(zg16_ctx* ctx, void* dst, const void* src, size_t bytes);
zg16_status stage_witness(zg16_ctx* ctx, const zg16_r1cs* r, const uint64_t* z_in, const uint64_t* z_aux, int on_device,
                          Fr** z_out);
zg16_status check_key_fits(zg16_ctx* ctx, const zg16_pk* pk, const zg16_r1cs* r);
void pk_destroy(zg16_pk* pk);
zg16_status pk_upload_impl(zg16_ctx* ctx, const zg16_pk_view* v, int shard_index, int shard_count, bool h_cyclic,
                           zg16_pk** out);
void r1cs_destroy(zg16_r1cs* r);
}  // namespace zg

namespace zg {
// Final assembly (bellman_ce 0.3.0 `create_proof`, after the multiexps) in two steps, so that the part
// that needs only the witness-side sums can run on the host while the GPU still works on the h query:
//   A = alpha + r*delta + a_query            B = beta + s*delta + b_query (G2)
//   C = s*A + r*(beta_g1 + b_query_g1) + (h + l)
// which equals upstream's rs*delta + s*alpha + r*beta + s*a_query + r*b_query + h + l term by term.
struct Assembly {
  G1Affine alpha1, beta1, delta1;
  G2Affine beta2, delta2;
  Fr rc, sc;  // canonical r, s
  G1XYZZ A, c_part;
  G2XYZZ B;
  // A and B in affine form, converted as soon as they are complete (beside the h multiexp, not after it)
  G1Affine A_affine;
  G2Affine B_affine;
  bool have_A = false, have_B = false;
  const FixedBaseTable<Fq>* delta1_table = nullptr;   // optional (a key handle has them)
  const FixedBaseTable<Fq2>* delta2_table = nullptr;
  // needs nothing from the GPU: alpha + r*delta and beta + s*delta
  void key_terms() {
    A = delta1_table ? delta1_table->mul(rc.v) : scalar_mul_w4(G1XYZZ::from_affine(delta1), rc.v);
    add_mixed(A, alpha1);
    B = delta2_table ? delta2_table->mul(sc.v) : scalar_mul_w4(G2XYZZ::from_affine(delta2), sc.v);
    add_mixed(B, beta2);
    key_done = true;
  }
  // The same on a host thread of its own, started once the proof's first kernels are queued: the calling thread
  // goes on queueing launches (a small circuit's proof is otherwise bound by this thread's arithmetic, not by the
  // GPU).  need_key_terms: before the first use of A or B.  The object must stay where it is while the thread runs.
  bool key_done = false;
  std::thread key_worker;
  void start_key_terms() {
    if (!key_done && !key_worker.joinable()) key_worker = std::thread([this] { key_terms(); });
  }
  void need_key_terms() {
    if (key_worker.joinable()) key_worker.join();
    if (!key_done) key_terms();
  }
  Assembly() = default;
  Assembly(Assembly&&) = default;
  Assembly& operator=(Assembly&&) = default;
  ~Assembly() {
    if (key_worker.joinable()) key_worker.join();
  }
  // key_terms and the key's own share of C, s*(alpha + r*delta) + r*beta_g1: what is left of C is then
  // s*a_query + r*b_query + h + l, linear in the sums, so every GPU of a sharded proof can form it from its own
  // partial sums (zg16_hshard_randomness)
  void key_terms_with_c() {
    key_terms();
    c_part = double_scalar_mul_glv(A, sc.v, G1XYZZ::from_affine(beta1), rc.v);
  }
  void with_g2(const G2XYZZ& qb2) {
    add_full(B, qb2);
    B_affine = to_affine(B);
    have_B = true;
  }
  void with_g1(const G1XYZZ& qa, const G1XYZZ& qb1) {
    add_full(A, qa);
    G1XYZZ b1 = qb1;
    add_mixed(b1, beta1);
    c_part = double_scalar_mul_glv(A, sc.v, b1, rc.v);
    A_affine = to_affine(A);
    have_A = true;
  }
  void early(const G1XYZZ& qa, const G1XYZZ& qb1, const G2XYZZ& qb2) {
    need_key_terms();
    with_g2(qb2);
    with_g1(qa, qb1);
  }
  void finish(const G1XYZZ& hl, zg16_proof* out) const {
    G1XYZZ C = c_part;
    add_full(C, hl);
    const G1Affine Aa = have_A ? A_affine : to_affine(A), Ca = to_affine(C);
    const G2Affine Ba = have_B ? B_affine : to_affine(B);
    memcpy(out->a, &Aa, 64);
    memcpy(out->b, &Ba, 128);
    memcpy(out->c, &Ca, 64);
  }
};
}  // namespace zg

// A base set resident in HBM (zg16_bases_upload): `n` points, optionally as a window table.
struct zg16_bases {
  zg16_ctx* ctx = nullptr;
  int group = 1;  // 1 = G1, 2 = G2
  size_t n = 0;
  uint32_t table_c = 0;
  uint32_t table_step = 1;
  void* points = nullptr;  // Affine<Fq> or Affine<Fq2>, in the multiexp kernels' packed form
  void* direct = nullptr;  // direct table (msm_build_direct) when the set is small and has window tables
};
