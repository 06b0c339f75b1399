// R1CS row evaluation a = A z, b = B z, c = C z over Fr as CSR sparse mat-vec.
//
// Replaces the per-row `eval` that bellman_ce 0.3.0's ProvingAssignment::enforce performs on one
// host thread while zinc-vm synthesizes (the in-tree evaluator with the same semantics is
// zinc-vm/src/constraint_systems/main.rs:35-49).  zinc rows are short (1-3 terms) except bit
// packing rows of ~250 terms (zinc-vm/src/gadgets/scalar/mod.rs:298-300), so short rows get one
// lane each and long rows get one wavefront each; the split is fixed at upload time because the
// matrices do not depend on the witness.
//
// Coefficients are not kept as 32-byte field elements.  zinc's gadgets emit +-1 and +-2^k almost
// exclusively (zinc-vm/src/gadgets/scalar/mod.rs:298-300 for the powers of two), so a term is a
// 4-byte variable index and a 2-byte coefficient CODE into a small table: code 0 adds z, code 1
// subtracts it (upstream's `eval` skips the product for a coefficient of one; the result is the same
// field value either way), every other code multiplies by a table entry that stays in cache.  A term
// costs 6 bytes of matrix traffic instead of 36.
#include <string.h>

#include <unordered_map>

#include "r1cs.hpp"

namespace zg {

// ---- coefficient codes (host, at upload) ------------------------------------------------------------
namespace {
struct FrKey {
  uint64_t w[4];
  bool operator==(const FrKey& o) const { return w[0] == o.w[0] && w[1] == o.w[1] && w[2] == o.w[2] && w[3] == o.w[3]; }
};
struct FrKeyHash {
  size_t operator()(const FrKey& k) const {
    uint64_t h = k.w[0] * 0x9e3779b97f4a7c15ull;
    h ^= (k.w[1] + 0xc2b2ae3d27d4eb4full) * 0xff51afd7ed558ccdull;
    h ^= (k.w[2] << 17) ^ (k.w[3] * 0xc4ceb9fe1a85ec53ull);
    return (size_t)(h ^ (h >> 29));
  }
};
FrKey key_of(const Fr& f) {
  FrKey k;
  memcpy(k.w, f.v, 32);
  return k;
}
}  // namespace

// codes[e] for every coefficient and the table the codes index.  Table layout: [0] one, [1] minus one,
// [2 + k] = 2^(k+1) and [256 + k] = -2^(k+1) for k < 253, [SPMV_CODE_DICT + i] = the i-th other value in order of
// first appearance.  A matrix with more distinct other values than fit gets SPMV_CODE_FULL for the overflow and
// keeps its full coefficient array for those terms (*needs_full).
void spmv_encode_coeffs(const uint64_t* coeff, size_t nnz, std::vector<uint16_t>& codes, std::vector<Fr>& table,
                        bool* needs_full) {
  table.assign(SPMV_CODE_DICT, Fr::zero());
  std::unordered_map<FrKey, uint16_t, FrKeyHash> known;
  known.reserve(1024);
  const Fr one = Fr::one(), minus_one = neg(one);
  table[0] = one;
  table[1] = minus_one;
  known[key_of(one)] = 0;
  known[key_of(minus_one)] = 1;
  Fr p = one;
  for (uint32_t k = 0; k < 253; k++) {
    p = dbl(p);  // 2^(k+1) in Montgomery form; below r for every k < 253
    table[2 + k] = p;
    table[256 + k] = neg(p);
    known.emplace(key_of(p), (uint16_t)(2 + k));
    known.emplace(key_of(table[256 + k]), (uint16_t)(256 + k));
  }
  codes.resize(nnz);
  *needs_full = false;
  // Pass 1, over host threads: every coefficient with a fixed code (+-1, +-2^k -- nearly all of a zinc circuit)
  // gets it; the others are marked.  Pass 2, in order: the marked ones get dictionary codes by first appearance.
  constexpr uint16_t OTHER = 0xfffe;
  host_parallel(nnz, [&](size_t lo, size_t hi) {
    FrKey last = key_of(one);
    uint16_t last_code = 0;
    for (size_t e = lo; e < hi; e++) {
      FrKey k;
      memcpy(k.w, coeff + 4 * e, 32);
      if (!(k == last)) {  // runs of equal coefficients (the ones of boolean rows) skip the look-up
        auto it = known.find(k);
        last_code = it != known.end() ? it->second : OTHER;
        last = k;
      }
      codes[e] = last_code;
    }
  });
  for (size_t e = 0; e < nnz; e++) {
    if (codes[e] != OTHER) continue;
    FrKey k;
    memcpy(k.w, coeff + 4 * e, 32);
    auto it = known.find(k);
    if (it != known.end()) {
      codes[e] = it->second;
    } else if (table.size() < OTHER) {
      const uint16_t code = (uint16_t)table.size();
      Fr f;
      memcpy(f.v, k.w, 32);
      table.push_back(f);
      known.emplace(k, code);
      codes[e] = code;
    } else {
      codes[e] = SPMV_CODE_FULL;
      *needs_full = true;
    }
  }
}

// Rows of one shape are scattered all over a circuit (a boolean row here, an eight-term linear combination there),
// and a wavefront that takes 64 consecutive rows runs as many term rounds as its longest row and the product path
// if any of its rows needs it.  Ordered by (term count, needs a product), 64 consecutive entries are 64 rows of
// the same cost.  Counting sort, stable, so row indices rise inside a class and neighbouring lanes still touch
// neighbouring rows wherever a circuit has runs of one shape.
void spmv_row_order(const uint32_t* row_ptr, size_t n_rows, const uint16_t* codes, std::vector<uint32_t>& order) {
  constexpr uint32_t NKEY = 2 * (SPMV_LONG_ROW + 1);
  std::vector<uint8_t> key(n_rows);
  host_parallel(n_rows, [&](size_t lo, size_t hi) {
    for (size_t r = lo; r < hi; r++) {
      const uint32_t len = row_ptr[r + 1] - row_ptr[r];
      if (len > SPMV_LONG_ROW) {
        key[r] = 0xff;
        continue;
      }
      uint32_t product = 0;
      for (uint32_t e = row_ptr[r]; e < row_ptr[r + 1]; e++) product |= codes[e] > 1 ? 1u : 0u;
      key[r] = (uint8_t)(2 * len + product);
    }
  });
  size_t start[NKEY + 1] = {};
  for (size_t r = 0; r < n_rows; r++)
    if (key[r] != 0xff) start[key[r] + 1]++;
  for (uint32_t k = 0; k < NKEY; k++) start[k + 1] += start[k];
  order.resize(start[NKEY]);
  for (size_t r = 0; r < n_rows; r++)
    if (key[r] != 0xff) order[start[key[r]]++] = (uint32_t)r;
}

// ---- kernels ------------------------------------------------------------------------------------------
struct SpmvMatrix {
  const uint32_t* row_ptr;
  const uint32_t* col;
  const uint16_t* code;
  const Fr* table;
  const Fr* coeff_full;  // only read for SPMV_CODE_FULL terms
};

// Which rows a launch evaluates and where their results go: output slot j holds row
//   (j >> log_blk) * stride + base + (j & (2^log_blk - 1))
// -- every row in order (log_blk = 31, base = 0), or the blocks of rows one GPU owns when a proof's transforms are
// spread over several GPUs.  Rows at or beyond n_rows (the padding up to the domain size) evaluate to zero.
struct RowMap {
  uint32_t log_blk, stride, base, n_out, n_rows;
  ZG_D uint32_t row(uint32_t j) const { return (j >> log_blk) * stride + base + (j & ((1u << log_blk) - 1)); }
  // slot of row r, or 0xffffffff when this launch does not own it
  ZG_D uint32_t slot(uint32_t r) const {
    if (log_blk >= 31) return r;
    if (r < base) return 0xffffffffu;
    const uint32_t off = r - base, k = off / stride, rem = off - k * stride;
    if (rem >> log_blk) return 0xffffffffu;
    const uint32_t j = (k << log_blk) + rem;
    return j < n_out ? j : 0xffffffffu;
  }
};

ZG_D Fr add_term(const Fr& acc, const SpmvMatrix& m, uint32_t e, const Fr* __restrict__ z) {
  const uint32_t code = m.code[e];
  const Fr x = z[m.col[e]];
  if (code == 0) return add(acc, x);
  if (code == 1) return sub(acc, x);
  const Fr c = code == SPMV_CODE_FULL ? m.coeff_full[e] : m.table[code];
  return add(acc, mul(c, x));
}

// One launch evaluates up to three matrices (blockIdx.y picks one): A, B and C of a proof share the witness and
// the row map, and three short launches in a row cost more in launch gaps and tails than in work.
struct SpmvBatch {
  SpmvMatrix m[3];
  Fr* out[3];
  const uint32_t* long_rows[3];
  uint32_t n_long[3];
  const uint32_t* order[3];  // k_spmv_ordered: the short rows by class
  uint32_t n_order[3];
};

__global__ __launch_bounds__(256) void k_spmv_short(SpmvBatch batch, const Fr* __restrict__ z, RowMap map) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= map.n_out) return;
  const SpmvMatrix& m = batch.m[blockIdx.y];
  const uint32_t r = map.row(j);
  Fr acc = Fr::zero();
  if (r < map.n_rows) {
    const uint32_t lo = m.row_ptr[r], hi = m.row_ptr[r + 1];
    if (hi - lo > SPMV_LONG_ROW) return;  // a wavefront takes this row in k_spmv_long
    for (uint32_t e = lo; e < hi; e++) acc = add_term(acc, m, e, z);
  }
  batch.out[blockIdx.y][j] = acc;
}

// Every row in order (the whole matrix, no row map): lane j takes the j-th short row of the class order; the lanes
// behind the last one write the zero padding [n_rows, n_out).  Long rows are left to k_spmv_long.
__global__ __launch_bounds__(256) void k_spmv_ordered(SpmvBatch batch, const Fr* __restrict__ z, uint32_t n_rows,
                                                      uint32_t n_out) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  const SpmvMatrix& m = batch.m[blockIdx.y];
  const uint32_t n_order = batch.n_order[blockIdx.y];
  if (j >= n_order) {
    const uint32_t pad = n_rows + (j - n_order);
    if (pad < n_out) batch.out[blockIdx.y][pad] = Fr::zero();
    return;
  }
  const uint32_t r = batch.order[blockIdx.y][j];
  const uint32_t lo = m.row_ptr[r], hi = m.row_ptr[r + 1];
  Fr acc = Fr::zero();
  for (uint32_t e = lo; e < hi; e++) acc = add_term(acc, m, e, z);
  batch.out[blockIdx.y][r] = acc;
}

__global__ __launch_bounds__(256) void k_spmv_long(SpmvBatch batch, const Fr* __restrict__ z, RowMap map) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= batch.n_long[blockIdx.y]) return;
  const SpmvMatrix& m = batch.m[blockIdx.y];
  const uint32_t r = batch.long_rows[blockIdx.y][wave];
  const uint32_t j = map.slot(r);
  if (j == 0xffffffffu) return;
  const uint32_t lo = m.row_ptr[r], hi = m.row_ptr[r + 1];
  Fr acc = Fr::zero();
  for (uint32_t e = lo + lane; e < hi; e += 64) acc = add_term(acc, m, e, z);
  for (int d = 32; d >= 1; d >>= 1) {
    Fr o;
#pragma unroll
    for (int i = 0; i < 8; i++) o.v[i] = __shfl_down(acc.v[i], d);
    acc = add(acc, o);
  }
  if (lane == 0) batch.out[blockIdx.y][j] = acc;
}

// matrices first .. first + count - 1 into outs[0 .. count)
static zg16_status spmv_launch(zg16_ctx* ctx, const zg16_r1cs* r, int first, int count, const Fr* z, Fr* const* outs,
                               const RowMap& map) {
  hipStream_t st = ctx->stream;
  SpmvBatch batch = {};
  uint32_t max_long = 0, max_order = 0;
  double nnz = 0;
  // the class order covers whole matrices: launches over every row (and the padding behind them) use it
  bool ordered = map.log_blk >= 31 && map.base == 0 && map.n_out >= r->n_rows && map.n_rows == r->n_rows;
  for (int k = 0; k < count; k++) {
    const R1csMatrix& m = r->m[first + k];
    batch.m[k] = {m.row_ptr, m.col, m.code, m.table, m.coeff};
    batch.out[k] = outs[k];
    batch.long_rows[k] = m.long_rows;
    batch.n_long[k] = m.n_long;
    batch.order[k] = m.order;
    batch.n_order[k] = m.n_order;
    ordered = ordered && (m.order || !m.n_order) && (size_t)m.n_order + m.n_long == r->n_rows;
    max_long = m.n_long > max_long ? m.n_long : max_long;
    max_order = m.n_order > max_order ? m.n_order : max_order;
    nnz += (double)m.nnz;
  }
  // share of the matrices this launch evaluates (slots past the last row are padding: zeros, no terms)
  const double rows_here = map.log_blk >= 31 && map.n_out > r->n_rows ? (double)r->n_rows : (double)map.n_out;
  const double share = r->n_rows ? rows_here / (double)(map.log_blk >= 31 ? r->n_rows : ((size_t)1 << r->log_m)) : 0.0;
  KernelTimer kt(ctx, K_SPMV, share * (nnz * 36 + (double)count * r->n_rows * 36) + (double)count * (r->n_in + r->n_aux) * 32);
  if (ordered) {
    const size_t lanes = (size_t)max_order + (map.n_out - r->n_rows);
    if (lanes) k_spmv_ordered<<<dim3((unsigned)((lanes + 255) / 256), count), 256, 0, st>>>(batch, z, map.n_rows, map.n_out);
  } else if (map.n_out) {
    k_spmv_short<<<dim3((map.n_out + 255) / 256, count), 256, 0, st>>>(batch, z, map);
  }
  if (max_long) k_spmv_long<<<dim3((max_long + 3) / 4, count), 256, 0, st>>>(batch, z, map);
  ZG_HIP(ctx, hipGetLastError());
  return ZG16_OK;
}

zg16_status spmv_one_device(zg16_ctx* ctx, const zg16_r1cs* r, int k, const Fr* z, Fr* out) {
  const RowMap all = {31, 0, 0, (uint32_t)r->n_rows, (uint32_t)r->n_rows};
  return spmv_launch(ctx, r, k, 1, z, &out, all);
}

zg16_status spmv_device(zg16_ctx* ctx, const zg16_r1cs* r, const Fr* z, Fr* a, Fr* b, Fr* c) {
  Fr* const outs[3] = {a, b, c};
  const RowMap all = {31, 0, 0, (uint32_t)r->n_rows, (uint32_t)r->n_rows};
  return spmv_launch(ctx, r, 0, 3, z, outs, all);
}

// Rows (j >> log_blk) * stride + base + (j mod 2^log_blk), j < n_out, of all three matrices into a, b, c; padding
// rows give zero.
zg16_status spmv_blocks_device(zg16_ctx* ctx, const zg16_r1cs* r, const Fr* z, Fr* a, Fr* b, Fr* c, uint32_t log_blk,
                               uint32_t stride, uint32_t base, uint32_t n_out) {
  Fr* const outs[3] = {a, b, c};
  const RowMap map = {log_blk, stride, base, n_out, (uint32_t)r->n_rows};
  return spmv_launch(ctx, r, 0, 3, z, outs, map);
}

}  // namespace zg
