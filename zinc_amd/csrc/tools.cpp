// Host-side workload tools (libzg16_tools.so): seeded synthetic R1CS instances shaped like
// zinc-vm output, and the scalar half of Groth16 key generation for them.  Benchmarks and tests
// use this to make 2^20..2^24-row inputs; the prove path itself never calls it.
//
// Row shapes follow the reference's gadgets: boolean rows (b)(1-b)=0
// (zinc-vm/src/gadgets/scalar/mod.rs:170-187), product rows
// (zinc-vm/src/gadgets/arithmetic/mul.rs:38-43), bit-packing rows with power-of-two coefficients
// (zinc-vm/src/gadgets/scalar/mod.rs:298-300), output rows (out)(1)=(lc)
// (zinc-vm/src/gadgets/output.rs:20-25), and the input rows `input_i * 0 = 0` bellman appends.
// Every linear combination is emitted in the canonical form of
// zinc-vm/src/constraint_systems/dedup.rs:50-68 (inputs first, ascending, no zero coefficient).
// The generator is rand 0.4's XorShiftRng with the seed the reference's own tests use
// (zinc-vm/src/gadgets/contract/storage.rs:219) unless another seed is passed.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "field.hpp"

using namespace zg;

namespace {

constexpr uint32_t AUX = 0x80000000u;

struct XorShift {
  uint32_t x, y, z, w;
  uint32_t next_u32() {
    uint32_t t = x ^ (x << 11);
    x = y; y = z; z = w;
    w = w ^ (w >> 19) ^ (t ^ (t >> 8));
    return w;
  }
  uint64_t next_u64() {
    uint64_t hi = next_u32();
    return (hi << 32) | next_u32();
  }
  uint64_t below(uint64_t n) { return next_u64() % n; }
  // ff_ce `Fr::rand`: the drawn pattern is the Montgomery representation
  Fr fr() {
    for (;;) {
      Fr r;
      for (int i = 0; i < 4; i++) {
        uint64_t l = next_u64();
        r.v[2 * i] = (uint32_t)l;
        r.v[2 * i + 1] = (uint32_t)(l >> 32);
      }
      r.v[7] &= 0x3fffffffu;
      if (is_canonical<FrParams>(r.v)) return r;
    }
  }
};

struct Matrix {
  std::vector<uint64_t> row_ptr{0};
  std::vector<uint32_t> col;
  std::vector<Fr> coeff;
};

struct Term {
  uint32_t col;
  Fr coeff;
};

}  // namespace

struct zgt_circuit {
  Matrix m[3];
  std::vector<Fr> z_in, z_aux;
  size_t n_rows = 0;
};

namespace {

const Fr& val(const zgt_circuit& c, uint32_t col) { return (col & AUX) ? c.z_aux[col & ~AUX] : c.z_in[col]; }

// canonical form: merge repeated variables, drop zeros, inputs first then aux, ascending
void canon(std::vector<Term>& lc) {
  std::sort(lc.begin(), lc.end(), [](const Term& a, const Term& b) { return a.col < b.col; });
  std::vector<Term> out;
  for (const Term& t : lc) {
    if (!out.empty() && out.back().col == t.col)
      out.back().coeff = add(out.back().coeff, t.coeff);
    else
      out.push_back(t);
  }
  lc.clear();
  for (const Term& t : out)
    if (!t.coeff.is_zero()) lc.push_back(t);
}

Fr eval(const zgt_circuit& c, const std::vector<Term>& lc) {
  Fr acc = Fr::zero();
  for (const Term& t : lc) acc = add(acc, mul(t.coeff, val(c, t.col)));
  return acc;
}

void emit(zgt_circuit& c, std::vector<Term> a, std::vector<Term> b, std::vector<Term> cc) {
  std::vector<Term>* lcs[3] = {&a, &b, &cc};
  for (int k = 0; k < 3; k++) {
    canon(*lcs[k]);
    for (const Term& t : *lcs[k]) {
      c.m[k].col.push_back(t.col);
      c.m[k].coeff.push_back(t.coeff);
    }
    c.m[k].row_ptr.push_back(c.m[k].col.size());
  }
  c.n_rows++;
}

uint32_t new_aux(zgt_circuit& c, const Fr& v) {
  c.z_aux.push_back(v);
  return (uint32_t)(c.z_aux.size() - 1) | AUX;
}

}  // namespace

extern "C" {

// total_rows counts every row including the n_pub output rows and the 1 + n_pub input rows, so a
// power of two gives a domain of exactly that size.  dense != 0 makes every aux value uniform.
zgt_circuit* zgt_circuit_generate(uint64_t total_rows, uint32_t n_pub, const uint32_t seed[4], int dense,
                                  uint32_t pack_width) {
  const uint64_t n_in = 1 + (uint64_t)n_pub;
  if (total_rows < 2 * n_in + 2) return nullptr;
  zgt_circuit* c = new zgt_circuit();
  XorShift rng{0x5dbe6259u, 0x8d313d76u, 0x3237db17u, 0xe5bc0654u};
  if (seed) rng = XorShift{seed[0], seed[1], seed[2], seed[3]};
  const Fr one = Fr::one(), minus_one = neg(one);
  c->z_in.push_back(one);
  std::vector<uint32_t> bits;
  const uint64_t body_rows = total_rows - n_in - n_pub;
  auto pick = [&]() { return (uint32_t)rng.below(c->z_aux.size()) | AUX; };
  // a seed variable tied down by a dummy row, as zinc does for every witness allocation
  {
    uint32_t x0 = new_aux(*c, rng.fr());
    emit(*c, {{x0, one}}, {{0, one}}, {{x0, one}});
  }
  std::vector<Fr> pow2(pack_width ? pack_width : 1);
  pow2[0] = one;
  for (size_t i = 1; i < pow2.size(); i++) pow2[i] = dbl(pow2[i - 1]);
  while (c->n_rows < body_rows) {
    const uint64_t kind = rng.below(100);
    if (kind < 70 && !dense) {
      uint32_t b = new_aux(*c, rng.below(2) ? one : Fr::zero());
      bits.push_back(b);
      emit(*c, {{b, one}}, {{0, one}, {b, minus_one}}, {});
    } else if (kind < 90 || (dense && kind < 70)) {
      std::vector<Term> l{{pick(), one}}, r{{pick(), one}};
      if (rng.below(2)) l.push_back({pick(), from_u64<FrParams>(rng.below(5) + 1)});
      if (rng.below(2)) r.push_back({0, rng.fr()});
      canon(l);
      canon(r);
      uint32_t w = new_aux(*c, mul(eval(*c, l), eval(*c, r)));
      emit(*c, l, r, {{w, one}});
    } else if (kind < 99 || bits.size() < pack_width || !pack_width) {
      std::vector<Term> t;
      const int nt = 3 + (int)rng.below(6);
      for (int i = 0; i < nt; i++) t.push_back({pick(), rng.below(2) ? rng.fr() : minus_one});
      canon(t);
      uint32_t y = new_aux(*c, eval(*c, t));
      emit(*c, t, {{0, one}}, {{y, one}});
    } else {
      std::vector<Term> t;
      for (uint32_t i = 0; i < pack_width; i++) t.push_back({bits[bits.size() - pack_width + i], pow2[i]});
      uint32_t y = new_aux(*c, eval(*c, t));
      emit(*c, t, {{0, one}}, {{y, one}});
    }
  }
  for (uint32_t i = 0; i < n_pub; i++) {
    uint32_t src = pick();
    c->z_in.push_back(val(*c, src));
    emit(*c, {{(uint32_t)c->z_in.size() - 1, one}}, {{0, one}}, {{src, one}});
  }
  for (uint32_t i = 0; i < n_in; i++) emit(*c, {{i, one}}, {}, {});
  return c;
}

void zgt_circuit_free(zgt_circuit* c) { delete c; }
uint64_t zgt_n_rows(const zgt_circuit* c) { return c->n_rows; }
uint64_t zgt_n_inputs(const zgt_circuit* c) { return c->z_in.size(); }
uint64_t zgt_n_aux(const zgt_circuit* c) { return c->z_aux.size(); }
uint64_t zgt_nnz(const zgt_circuit* c, int k) { return c->m[k].col.size(); }
const uint64_t* zgt_row_ptr(const zgt_circuit* c, int k) { return c->m[k].row_ptr.data(); }
const uint32_t* zgt_col(const zgt_circuit* c, int k) { return c->m[k].col.data(); }
const uint64_t* zgt_coeff(const zgt_circuit* c, int k) { return (const uint64_t*)c->m[k].coeff.data(); }
const uint64_t* zgt_z_in(const zgt_circuit* c) { return (const uint64_t*)c->z_in.data(); }
const uint64_t* zgt_z_aux(const zgt_circuit* c) { return (const uint64_t*)c->z_aux.data(); }

// 1 if every row satisfies <A,z><B,z> = <C,z>
int zgt_circuit_satisfied(const zgt_circuit* c) {
  for (size_t r = 0; r < c->n_rows; r++) {
    Fr v[3];
    for (int k = 0; k < 3; k++) {
      Fr acc = Fr::zero();
      for (uint64_t e = c->m[k].row_ptr[r]; e < c->m[k].row_ptr[r + 1]; e++)
        acc = add(acc, mul(c->m[k].coeff[e], val(*c, c->m[k].col[e])));
      v[k] = acc;
    }
    if (!(mul(v[0], v[1]) == v[2])) return 0;
  }
  return 1;
}

// out[i] = c * base^i, i < n (Montgomery in and out)
void zgt_powers(const uint64_t base[4], const uint64_t c0[4], uint64_t n, uint64_t* out) {
  Fr b, acc;
  memcpy(&b, base, 32);
  memcpy(&acc, c0, 32);
  for (uint64_t i = 0; i < n; i++) {
    memcpy(out + 4 * i, &acc, 32);
    acc = mul(acc, b);
  }
}

// Scalar half of bellman `generate_parameters` for a recorded circuit.  lag[j] = L_j(tau) for the
// m-point domain (Montgomery).  Writes, all Montgomery Fr:
//   a_s, b_s      per variable (inputs then aux): A_v(tau), B_v(tau)
//   ic_s          per input:  (beta A_v + alpha B_v + C_v) / gamma
//   l_s           per aux:    (beta A_v + alpha B_v + C_v) / delta
void zgt_setup_scalars(const zgt_circuit* c, const uint64_t* lag, const uint64_t alpha_[4], const uint64_t beta_[4],
                       const uint64_t gamma_inv_[4], const uint64_t delta_inv_[4], uint64_t* a_s, uint64_t* b_s,
                       uint64_t* ic_s, uint64_t* l_s) {
  const size_t n_in = c->z_in.size(), n_aux = c->z_aux.size(), nv = n_in + n_aux;
  std::vector<Fr> acc[3];
  const Fr* L = (const Fr*)lag;
  for (int k = 0; k < 3; k++) {
    acc[k].assign(nv, Fr::zero());
    const Matrix& m = c->m[k];
    for (size_t r = 0; r < c->n_rows; r++)
      for (uint64_t e = m.row_ptr[r]; e < m.row_ptr[r + 1]; e++) {
        const uint32_t col = m.col[e];
        const size_t v = (col & AUX) ? n_in + (col & ~AUX) : col;
        acc[k][v] = add(acc[k][v], mul(m.coeff[e], L[r]));
      }
  }
  Fr alpha, beta, gi, di;
  memcpy(&alpha, alpha_, 32);
  memcpy(&beta, beta_, 32);
  memcpy(&gi, gamma_inv_, 32);
  memcpy(&di, delta_inv_, 32);
  memcpy(a_s, acc[0].data(), nv * 32);
  memcpy(b_s, acc[1].data(), nv * 32);
  for (size_t v = 0; v < nv; v++) {
    Fr t = add(add(mul(beta, acc[0][v]), mul(alpha, acc[1][v])), acc[2][v]);
    if (v < n_in) {
      t = mul(t, gi);
      memcpy(ic_s + 4 * v, &t, 32);
    } else {
      t = mul(t, di);
      memcpy(l_s + 4 * (v - n_in), &t, 32);
    }
  }
}

// Affine points (Montgomery limbs; all-zero = the identity) -> pairing_ce's uncompressed encoding, the form the
// vectors of a `Parameters::write` key file hold: 32-byte big-endian canonical coordinates, G1 = x || y,
// G2 = x.c1 || x.c0 || y.c1 || y.c0, the identity as 0x40 followed by zeros.  coords: Fq values per point (2 for
// G1, 4 for G2).  Split over `threads` host threads.  Benchmarks use it to make key files of 2^20 .. 2^24-row
// circuits (zinc-vm writes such a file at zinc-vm/src/zvm/arguments/command/setup.rs:67-69).
void zgt_points_encode(const uint64_t* pts, uint64_t n, int coords, uint8_t* out, int threads) {
  auto work = [=](uint64_t lo, uint64_t hi) {
    static const int order_g1[2] = {0, 1}, order_g2[4] = {1, 0, 3, 2};
    const int* order = coords == 2 ? order_g1 : order_g2;
    for (uint64_t i = lo; i < hi; i++) {
      const uint64_t* p = pts + i * 4 * coords;
      uint8_t* o = out + i * 32 * coords;
      uint64_t any = 0;
      for (int k = 0; k < 4 * coords; k++) any |= p[k];
      if (!any) {
        memset(o, 0, 32 * coords);
        o[0] = 0x40;
        continue;
      }
      for (int c = 0; c < coords; c++) {
        Fq m;
        memcpy(&m, p + 4 * order[c], 32);
        const Fq x = from_mont(m);
        for (int l = 0; l < 8; l++) {
          uint8_t* b = o + 32 * c + 28 - 4 * l;
          b[0] = (uint8_t)(x.v[l] >> 24); b[1] = (uint8_t)(x.v[l] >> 16); b[2] = (uint8_t)(x.v[l] >> 8); b[3] = (uint8_t)x.v[l];
        }
      }
    }
  };
  if (threads < 2 || n < 4096) {
    work(0, n);
    return;
  }
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; t++) pool.emplace_back(work, n * t / threads, n * (t + 1) / threads);
  for (auto& th : pool) th.join();
}

// Small Fr helpers for Python callers (Montgomery limbs in and out)
void zgt_fr_mul(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) {
  Fr x, y;
  memcpy(&x, a, 32);
  memcpy(&y, b, 32);
  Fr r = mul(x, y);
  memcpy(out, &r, 32);
}

}  // extern "C"
