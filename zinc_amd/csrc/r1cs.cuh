// Device-resident constraint matrices and proving key (internal definitions of the opaque
// handles declared in include/zinc_groth16.h).
#pragma once
#include "common.cuh"

namespace zg {

static constexpr uint32_t SPMV_LONG_ROW = 32;  // rows with more terms get a whole wavefront

struct R1csMatrix {
  uint32_t* row_ptr = nullptr;  // n_rows + 1
  uint32_t* col = nullptr;      // index into the flat witness z = z_in || z_aux
  Fr* coeff = nullptr;
  uint32_t* long_rows = nullptr;
  uint32_t n_long = 0;
  size_t nnz = 0;
};

zg16_status spmv_device(zg16_ctx* ctx, const zg16_r1cs* r, const Fr* z, Fr* a, Fr* b, Fr* c);
// rows of matrix k (0 = A, 1 = B, 2 = C) only
zg16_status spmv_one_device(zg16_ctx* ctx, const zg16_r1cs* r, int k, const Fr* z, Fr* out);

}  // namespace zg

struct zg16_r1cs {
  zg16_ctx* ctx = nullptr;
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
  int shard_index = 0, shard_count = 1;
  // totals of the full key and the [lo, hi) range this shard holds, per query
  size_t n_h = 0, n_l = 0, n_a = 0, n_b1 = 0, n_b2 = 0;
  size_t lo_h = 0, hi_h = 0, lo_l = 0, hi_l = 0, lo_a = 0, hi_a = 0, lo_b = 0, hi_b = 0;
  zg::G1Affine *h = nullptr, *l = nullptr, *a = nullptr, *b1 = nullptr;
  zg::G2Affine* b2 = nullptr;
  // When table_c != 0 each query array holds msm_table_windows(c) windows of its shard, c being the
  // query's own width below: q[w * (hi - lo) + k] = 2^(c * w) * P_(lo + k).
  uint32_t table_c = 0;       // l, a, b_g1 (one launch, one width)
  uint32_t table_c_h = 0;     // h
  uint32_t table_c_g2 = 0;    // b_g2
  zg::G1Affine alpha_g1, beta_g1, delta_g1;
  zg::G2Affine beta_g2, delta_g2;
};

// A base set resident in HBM (zg16_bases_upload): `n` points, optionally as a window table.
struct zg16_bases {
  zg16_ctx* ctx = nullptr;
  int group = 1;  // 1 = G1, 2 = G2
  size_t n = 0;
  uint32_t table_c = 0;
  void* points = nullptr;  // Affine<Fq> or Affine<Fq2>, in the multiexp kernels' packed form
};
