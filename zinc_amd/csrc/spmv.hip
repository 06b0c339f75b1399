// R1CS row evaluation a = A z, b = B z, c = C z over Fr as CSR sparse mat-vec.
//
// Replaces the per-row `eval` that bellman_ce 0.3.0's ProvingAssignment::enforce performs on one
// host thread while zinc-vm synthesizes (the in-tree evaluator with the same semantics is
// zinc-vm/src/constraint_systems/main.rs:35-49).  zinc rows are short (1-3 terms) except bit
// packing rows of ~250 terms (zinc-vm/src/gadgets/scalar/mod.rs:298-300), so short rows get one
// lane each and long rows get one wavefront each; the split is fixed at upload time because the
// matrices do not depend on the witness.
#include "r1cs.cuh"

namespace zg {

ZG_D bool is_mont_one(const Fr& c) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= c.v[i] ^ FrParams::one(i);
  return o == 0;
}

ZG_D Fr term(const Fr& coeff, const Fr& z) {
  // multiplying by the Montgomery form of one is the identity map: skip it
  return is_mont_one(coeff) ? z : mul(coeff, z);
}

__global__ __launch_bounds__(256) void k_spmv_short(const uint32_t* __restrict__ row_ptr,
                                                    const uint32_t* __restrict__ col,
                                                    const Fr* __restrict__ coeff, const Fr* __restrict__ z,
                                                    Fr* __restrict__ out, uint32_t n_rows) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const uint32_t lo = row_ptr[r], hi = row_ptr[r + 1];
  if (hi - lo > SPMV_LONG_ROW) return;  // a wavefront takes this row in k_spmv_long
  Fr acc = Fr::zero();
  for (uint32_t k = lo; k < hi; k++) acc = add(acc, term(coeff[k], z[col[k]]));
  out[r] = acc;
}

__global__ __launch_bounds__(256) void k_spmv_long(const uint32_t* __restrict__ long_rows, uint32_t n_long,
                                                   const uint32_t* __restrict__ row_ptr,
                                                   const uint32_t* __restrict__ col,
                                                   const Fr* __restrict__ coeff, const Fr* __restrict__ z,
                                                   Fr* __restrict__ out) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= n_long) return;
  const uint32_t r = long_rows[wave];
  const uint32_t lo = row_ptr[r], hi = row_ptr[r + 1];
  Fr acc = Fr::zero();
  for (uint32_t k = lo + lane; k < hi; k += 64) acc = add(acc, term(coeff[k], z[col[k]]));
  for (int d = 32; d >= 1; d >>= 1) {
    Fr o;
#pragma unroll
    for (int i = 0; i < 8; i++) o.v[i] = __shfl_down(acc.v[i], d);
    acc = add(acc, o);
  }
  if (lane == 0) out[r] = acc;
}

zg16_status spmv_one_device(zg16_ctx* ctx, const zg16_r1cs* r, int k, const Fr* z, Fr* out) {
  hipStream_t st = ctx->stream;
  const R1csMatrix& m = r->m[k];
  KernelTimer kt(ctx, K_SPMV, (double)m.nnz * 36 + (double)(r->n_in + r->n_aux) * 32 + (double)r->n_rows * 36);
  if (r->n_rows)
    k_spmv_short<<<(unsigned)((r->n_rows + 255) / 256), 256, 0, st>>>(m.row_ptr, m.col, m.coeff, z, out,
                                                                      (uint32_t)r->n_rows);
  if (m.n_long)
    k_spmv_long<<<(unsigned)((m.n_long + 3) / 4), 256, 0, st>>>(m.long_rows, m.n_long, m.row_ptr, m.col, m.coeff, z, out);
  ZG_HIP(ctx, hipGetLastError());
  return ZG16_OK;
}

zg16_status spmv_device(zg16_ctx* ctx, const zg16_r1cs* r, const Fr* z, Fr* a, Fr* b, Fr* c) {
  Fr* outs[3] = {a, b, c};
  for (int k = 0; k < 3; k++) ZG_TRY(spmv_one_device(ctx, r, k, z, outs[k]));
  return ZG16_OK;
}

}  // namespace zg
