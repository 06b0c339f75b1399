// CPU restatement of the bellman_ce 0.3.0 Groth16 prover algorithms (liboracle_cpu.so).
//
// TEST INFRASTRUCTURE ONLY.  Loaded by tests/, by __graft_entry__.smoke() and by the cpu_baseline
// leg of bench.py, as the checker and as the timed CPU baseline -- never by the product package.
//
// PARITY STATUS: "parity unpinned".  The code this restates (bellman_ce 0.3.0 groth16/prover.rs,
// domain.rs, multiexp.rs; pairing_ce 0.17.0 bn256; ff_ce 0.6.0 -- pinned by the reference's
// Cargo.lock:629-643, :2514-2517, :1363-1366, called from zinc-vm/src/core/circuit/facade.rs:153)
// is not on disk and cannot be built here (no Rust toolchain).  The reference holds no golden
// vectors for it.  This file follows the published algorithms: serial radix-2 FFT with bit
// reversal, the log2(cpus)-way parallel FFT split, Pippenger multiexp with window
// c = 3 (n < 32) else ceil(ln n), one task per window, the zero / one scalar fast paths, density
// maps that skip scalars without consuming a base, and Jacobian group formulas.  It is pinned by
// agreement with the big-integer Python oracle (oracle/groth16.py), which in turn is pinned by the
// Groth16 pairing equation and a closed-form derivation from the toxic waste.
//
// Arithmetic is written independently of the GPU code: 4 x 64-bit limbs with unsigned __int128
// products here, 8 x 32-bit limbs and XYZZ coordinates there.
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

typedef unsigned __int128 u128;

namespace {

// ------------------------------------------------------------------------------------ fields
struct FrP {
  static constexpr uint64_t M[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
  static constexpr uint64_t ONE[4] = {0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full};
  static constexpr uint64_t R2[4] = {0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull};
  static constexpr uint64_t INV = 0xc2e1f593efffffffull;
};
struct FqP {
  static constexpr uint64_t M[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
  static constexpr uint64_t ONE[4] = {0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull, 0x0e0a77c19a07df2full};
  static constexpr uint64_t R2[4] = {0xf32cfc5b538afa89ull, 0xb5e71911d44501fbull, 0x47ab1eff0a417ff6ull, 0x06d89f71cab8351full};
  static constexpr uint64_t INV = 0x87d20782e4866389ull;
};
constexpr uint64_t FrP::M[4], FrP::ONE[4], FrP::R2[4], FqP::M[4], FqP::ONE[4], FqP::R2[4];

template <class P>
struct Fp {
  uint64_t l[4];
  static Fp zero() { return Fp{{0, 0, 0, 0}}; }
  static Fp one() { return Fp{{P::ONE[0], P::ONE[1], P::ONE[2], P::ONE[3]}}; }
  bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
  bool operator==(const Fp& o) const { return l[0] == o.l[0] && l[1] == o.l[1] && l[2] == o.l[2] && l[3] == o.l[3]; }
  bool operator!=(const Fp& o) const { return !(*this == o); }

  static bool geq_mod(const uint64_t* a) {
    for (int i = 3; i >= 0; i--) {
      if (a[i] > P::M[i]) return true;
      if (a[i] < P::M[i]) return false;
    }
    return true;
  }
  static void sub_mod(uint64_t* a) {
    u128 b = 0;
    for (int i = 0; i < 4; i++) {
      u128 t = (u128)a[i] - P::M[i] - (uint64_t)b;
      a[i] = (uint64_t)t;
      b = (t >> 64) & 1;
    }
  }
  Fp operator+(const Fp& o) const {
    Fp r;
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (u128)l[i] + o.l[i];
      r.l[i] = (uint64_t)c;
      c >>= 64;
    }
    if (geq_mod(r.l)) sub_mod(r.l);
    return r;
  }
  Fp operator-(const Fp& o) const {
    Fp r;
    u128 b = 0;
    for (int i = 0; i < 4; i++) {
      u128 t = (u128)l[i] - o.l[i] - (uint64_t)b;
      r.l[i] = (uint64_t)t;
      b = (t >> 64) & 1;
    }
    if (b) {
      u128 c = 0;
      for (int i = 0; i < 4; i++) {
        c += (u128)r.l[i] + P::M[i];
        r.l[i] = (uint64_t)c;
        c >>= 64;
      }
    }
    return r;
  }
  Fp neg() const { return is_zero() ? *this : zero() - *this; }
  Fp dbl() const { return *this + *this; }
  Fp operator*(const Fp& o) const {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
      u128 c = 0;
      for (int j = 0; j < 4; j++) {
        c += (u128)l[j] * o.l[i] + t[j];
        t[j] = (uint64_t)c;
        c >>= 64;
      }
      c += t[4];
      t[4] = (uint64_t)c;
      t[5] = (uint64_t)(c >> 64);
      uint64_t m = t[0] * P::INV;
      c = (u128)m * P::M[0] + t[0];
      c >>= 64;
      for (int j = 1; j < 4; j++) {
        c += (u128)m * P::M[j] + t[j];
        t[j - 1] = (uint64_t)c;
        c >>= 64;
      }
      c += t[4];
      t[3] = (uint64_t)c;
      t[4] = t[5] + (uint64_t)(c >> 64);
    }
    Fp r{{t[0], t[1], t[2], t[3]}};
    if (t[4] || geq_mod(r.l)) sub_mod(r.l);
    return r;
  }
  Fp sqr() const { return *this * *this; }
  Fp pow(const uint64_t* e, int n) const {
    Fp r = one();
    for (int i = n - 1; i >= 0; i--)
      for (int b = 63; b >= 0; b--) {
        r = r.sqr();
        if ((e[i] >> b) & 1) r = r * *this;
      }
    return r;
  }
  Fp inverse() const {
    uint64_t e[4] = {P::M[0] - 2, P::M[1], P::M[2], P::M[3]};
    return pow(e, 4);
  }
  // Montgomery -> canonical limbs ("into_repr")
  void into_repr(uint64_t out[4]) const {
    Fp o{{1, 0, 0, 0}};
    Fp r = *this * o;
    memcpy(out, r.l, 32);
  }
  static Fp from_u64(uint64_t x) {
    Fp a{{x, 0, 0, 0}}, r2{{P::R2[0], P::R2[1], P::R2[2], P::R2[3]}};
    return a * r2;
  }
};
typedef Fp<FrP> Fr;
typedef Fp<FqP> Fq;

struct Fq2 {
  Fq c0, c1;
  static Fq2 zero() { return {Fq::zero(), Fq::zero()}; }
  static Fq2 one() { return {Fq::one(), Fq::zero()}; }
  bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  bool operator==(const Fq2& o) const { return c0 == o.c0 && c1 == o.c1; }
  bool operator!=(const Fq2& o) const { return !(*this == o); }
  Fq2 operator+(const Fq2& o) const { return {c0 + o.c0, c1 + o.c1}; }
  Fq2 operator-(const Fq2& o) const { return {c0 - o.c0, c1 - o.c1}; }
  Fq2 neg() const { return {c0.neg(), c1.neg()}; }
  Fq2 dbl() const { return {c0.dbl(), c1.dbl()}; }
  // schoolbook product with u^2 = -1
  Fq2 operator*(const Fq2& o) const { return {c0 * o.c0 - c1 * o.c1, c0 * o.c1 + c1 * o.c0}; }
  Fq2 sqr() const { return *this * *this; }
  Fq2 inverse() const {
    Fq n = (c0.sqr() + c1.sqr()).inverse();
    return {c0 * n, (c1 * n).neg()};
  }
};

// --------------------------------------------------------------------------------- curve points
template <class F>
struct Aff {
  F x, y;
  bool is_inf() const { return x.is_zero() && y.is_zero(); }  // ABI encoding of the identity
};

// Jacobian coordinates, formulas as in pairing_ce's `double` / `add_assign` / `add_assign_mixed`
template <class F>
struct Jac {
  F x, y, z;
  static Jac inf() { return {F::zero(), F::one(), F::zero()}; }
  bool is_inf() const { return z.is_zero(); }
  void dbl() {
    if (is_inf()) return;
    F a = x.sqr(), b = y.sqr(), c = b.sqr();
    F d = ((x + b).sqr() - a - c).dbl();
    F e = a.dbl() + a;
    F f = e.sqr();
    F z3 = (y * z).dbl();
    x = f - d.dbl();
    y = e * (d - x) - c.dbl().dbl().dbl();
    z = z3;
  }
  void add_mixed(const Aff<F>& o) {
    if (o.is_inf()) return;
    if (is_inf()) {
      x = o.x; y = o.y; z = F::one();
      return;
    }
    F z1z1 = z.sqr();
    F u2 = o.x * z1z1;
    F s2 = o.y * z * z1z1;
    if (x == u2 && y == s2) {
      dbl();
      return;
    }
    F h = u2 - x;
    F hh = h.sqr();
    F i = hh.dbl().dbl();
    F j = h * i;
    F r = (s2 - y).dbl();
    F v = x * i;
    F x3 = r.sqr() - j - v.dbl();
    F y3 = r * (v - x3) - (y * j).dbl();
    F z3 = (z + h).sqr() - z1z1 - hh;
    x = x3; y = y3; z = z3;
  }
  void add(const Jac& o) {
    if (o.is_inf()) return;
    if (is_inf()) {
      *this = o;
      return;
    }
    F z1z1 = z.sqr(), z2z2 = o.z.sqr();
    F u1 = x * z2z2, u2 = o.x * z1z1;
    F s1 = y * o.z * z2z2, s2 = o.y * z * z1z1;
    if (u1 == u2 && s1 == s2) {
      dbl();
      return;
    }
    F h = u2 - u1;
    F i = h.dbl().sqr();
    F j = h * i;
    F r = (s2 - s1).dbl();
    F v = u1 * i;
    F x3 = r.sqr() - j - v.dbl();
    F y3 = r * (v - x3) - (s1 * j).dbl();
    F z3 = ((z + o.z).sqr() - z1z1 - z2z2) * h;
    x = x3; y = y3; z = z3;
  }
  void mul_assign(const uint64_t k[4]) {  // canonical scalar, MSB first
    Jac res = inf();
    bool found = false;
    for (int i = 3; i >= 0; i--)
      for (int b = 63; b >= 0; b--) {
        if (found) res.dbl();
        if ((k[i] >> b) & 1) {
          found = true;
          res.add(*this);
        }
      }
    *this = res;
  }
  Aff<F> into_affine() const {
    if (is_inf()) return {F::zero(), F::zero()};
    F zi = z.inverse(), zi2 = zi.sqr();
    return {x * zi2, y * zi2 * zi};
  }
};

// -------------------------------------------------------------------------------- worker pool
// bellman's `Worker`: a fixed pool; `scope`-style chunked loops and future-like tasks.
struct Pool {
  unsigned n;
  explicit Pool(unsigned threads) : n(threads ? threads : 1) {}
  // run fn(task) for task in [0, tasks) on n threads
  void run(size_t tasks, const std::function<void(size_t)>& fn) const {
    if (n <= 1 || tasks <= 1) {
      for (size_t t = 0; t < tasks; t++) fn(t);
      return;
    }
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    unsigned use = (unsigned)std::min<size_t>(n, tasks);
    for (unsigned i = 0; i < use; i++)
      th.emplace_back([&] {
        for (;;) {
          size_t t = next.fetch_add(1);
          if (t >= tasks) return;
          fn(t);
        }
      });
    for (auto& t : th) t.join();
  }
  unsigned log_cpus() const {
    unsigned l = 0;
    while ((2u << l) <= n) l++;
    return l;
  }
};

// ---------------------------------------------------------------------------------------- FFT
Fr root_of_unity() {
  // 7^((r-1)/2^28)
  uint64_t e[4] = {FrP::M[0] - 1, FrP::M[1], FrP::M[2], FrP::M[3]};
  uint64_t s[4];
  for (int i = 0; i < 4; i++) s[i] = (e[i] >> 28) | (i + 1 < 4 ? e[i + 1] << 36 : 0);
  return Fr::from_u64(7).pow(s, 4);
}

uint32_t bitreverse(uint32_t n, uint32_t l) {
  uint32_t r = 0;
  for (uint32_t i = 0; i < l; i++) {
    r = (r << 1) | (n & 1);
    n >>= 1;
  }
  return r;
}

void serial_fft(Fr* a, size_t n, const Fr& omega, uint32_t log_n) {
  for (uint32_t k = 0; k < n; k++) {
    uint32_t rk = bitreverse(k, log_n);
    if (k < rk) std::swap(a[k], a[rk]);
  }
  size_t m = 1;
  for (uint32_t s = 0; s < log_n; s++) {
    uint64_t e[1] = {n / (2 * m)};
    Fr w_m = omega.pow(e, 1);
    for (size_t k = 0; k < n; k += 2 * m) {
      Fr w = Fr::one();
      for (size_t j = 0; j < m; j++) {
        Fr t = a[k + j + m] * w;
        Fr tmp = a[k + j] - t;
        a[k + j + m] = tmp;
        a[k + j] = a[k + j] + t;
        w = w * w_m;
      }
    }
    m *= 2;
  }
}

void parallel_fft(Fr* a, size_t n, const Fr& omega, uint32_t log_n, uint32_t log_cpus, const Pool& pool) {
  const size_t num_cpus = (size_t)1 << log_cpus;
  const uint32_t log_new_n = log_n - log_cpus;
  const size_t new_n = (size_t)1 << log_new_n;
  std::vector<std::vector<Fr>> tmp(num_cpus, std::vector<Fr>(new_n, Fr::zero()));
  uint64_t e1[1] = {num_cpus};
  const Fr new_omega = omega.pow(e1, 1);
  pool.run(num_cpus, [&](size_t j) {
    uint64_t ej[1] = {j}, es[1] = {j << log_new_n};
    const Fr omega_j = omega.pow(ej, 1), omega_step = omega.pow(es, 1);
    Fr elt = Fr::one();
    for (size_t i = 0; i < new_n; i++) {
      for (size_t s = 0; s < num_cpus; s++) {
        size_t idx = (i + (s << log_new_n)) % n;
        tmp[j][i] = tmp[j][i] + a[idx] * elt;
        elt = elt * omega_step;
      }
      elt = elt * omega_j;
    }
    serial_fft(tmp[j].data(), new_n, new_omega, log_new_n);
  });
  const size_t mask = num_cpus - 1;
  const size_t chunks = pool.n;
  pool.run(chunks, [&](size_t c) {
    size_t lo = n * c / chunks, hi = n * (c + 1) / chunks;
    for (size_t idx = lo; idx < hi; idx++) a[idx] = tmp[idx & mask][idx >> log_cpus];
  });
}

void best_fft(Fr* a, size_t n, const Fr& omega, uint32_t log_n, const Pool& pool) {
  uint32_t log_cpus = pool.log_cpus();
  if (log_n <= log_cpus || pool.n <= 1)
    serial_fft(a, n, omega, log_n);
  else
    parallel_fft(a, n, omega, log_n, log_cpus, pool);
}

struct Domain {
  std::vector<Fr> c;
  uint32_t exp;
  Fr omega, omegainv, geninv, minv;
  bool ok;
  Domain(const Fr* coeffs, size_t len) {
    size_t m = 1;
    exp = 0;
    ok = true;
    while (m < len) {
      m *= 2;
      exp++;
      if (exp >= 28) {
        ok = false;
        return;
      }
    }
    omega = root_of_unity();
    for (uint32_t i = exp; i < 28; i++) omega = omega.sqr();
    c.assign(coeffs, coeffs + len);
    c.resize(m, Fr::zero());
    omegainv = omega.inverse();
    geninv = Fr::from_u64(7).inverse();
    minv = Fr::from_u64(m).inverse();
  }
  void chunked(const Pool& pool, const std::function<void(size_t, size_t)>& fn) {
    const size_t chunks = pool.n, n = c.size();
    pool.run(chunks, [&](size_t k) { fn(n * k / chunks, n * (k + 1) / chunks); });
  }
  void fft(const Pool& p) { best_fft(c.data(), c.size(), omega, exp, p); }
  void ifft(const Pool& p) {
    best_fft(c.data(), c.size(), omegainv, exp, p);
    chunked(p, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) c[i] = c[i] * minv; });
  }
  void distribute_powers(const Pool& p, const Fr& g) {
    chunked(p, [&](size_t lo, size_t hi) {
      uint64_t e[1] = {lo};
      Fr u = g.pow(e, 1);
      for (size_t i = lo; i < hi; i++) {
        c[i] = c[i] * u;
        u = u * g;
      }
    });
  }
  void coset_fft(const Pool& p) {
    distribute_powers(p, Fr::from_u64(7));
    fft(p);
  }
  void icoset_fft(const Pool& p) {
    ifft(p);
    distribute_powers(p, geninv);
  }
  void divide_by_z_on_coset(const Pool& p) {
    uint64_t e[1] = {c.size()};
    Fr i = (Fr::from_u64(7).pow(e, 1) - Fr::one()).inverse();
    chunked(p, [&](size_t lo, size_t hi) { for (size_t k = lo; k < hi; k++) c[k] = c[k] * i; });
  }
};

// ------------------------------------------------------------------------------------ multiexp
uint32_t bellman_window(size_t n) { return n < 32 ? 3u : (uint32_t)std::ceil(std::log((double)n)); }

// bits [skip, skip+c) of a canonical 256-bit scalar
uint64_t window_digit(const uint64_t k[4], uint32_t skip, uint32_t c) {
  if (skip >= 256) return 0;
  uint32_t w = skip / 64, sh = skip % 64;
  uint64_t v = k[w] >> sh;
  if (sh && w + 1 < 4) v |= k[w + 1] << (64 - sh);
  return v & (((uint64_t)1 << c) - 1);
}

// One window of bellman's `multiexp_inner`.  Returns false if an identity base is consumed.
template <class F>
bool multiexp_window(const Aff<F>* bases, const uint8_t* density, const uint64_t* exps, size_t n, uint32_t skip,
                     uint32_t c, bool handle_trivial, Jac<F>* out) {
  Jac<F> acc = Jac<F>::inf();
  std::vector<Jac<F>> buckets(((size_t)1 << c) - 1, Jac<F>::inf());
  size_t bi = 0;
  for (size_t i = 0; i < n; i++) {
    if (density && !density[i]) continue;
    const Aff<F>& base = bases[bi++];
    const uint64_t* k = exps + 4 * i;
    const bool zero = (k[0] | k[1] | k[2] | k[3]) == 0;
    if (zero) continue;
    const bool one = k[0] == 1 && (k[1] | k[2] | k[3]) == 0;
    if (one) {
      if (handle_trivial) {
        if (base.is_inf()) return false;
        acc.add_mixed(base);
      }
      continue;
    }
    uint64_t d = window_digit(k, skip, c);
    if (d) {
      if (base.is_inf()) return false;
      buckets[d - 1].add_mixed(base);
    }
  }
  Jac<F> running = Jac<F>::inf();
  for (size_t b = buckets.size(); b-- > 0;) {
    running.add(buckets[b]);
    acc.add(running);
  }
  *out = acc;
  return true;
}

template <class F>
struct MexpTask {
  const Aff<F>* bases;
  const uint8_t* density;
  const uint64_t* exps;  // canonical
  size_t n;
  uint32_t c, nwin;
  std::vector<Jac<F>> win;
  bool ok = true;
  void init(const Aff<F>* b, const uint8_t* d, const uint64_t* e, size_t n_) {
    bases = b; density = d; exps = e; n = n_;
    c = bellman_window(n);
    nwin = (254 + c - 1) / c;
    win.assign(nwin, Jac<F>::inf());
  }
  void run_window(uint32_t w) {
    if (!multiexp_window(bases, density, exps, n, w * c, c, w == 0, &win[w])) ok = false;
  }
  Jac<F> combine() const {
    Jac<F> total = Jac<F>::inf();
    for (int w = (int)nwin - 1; w >= 0; w--) {
      for (uint32_t k = 0; k < c; k++) total.dbl();
      total.add(win[w]);
    }
    return total;
  }
};

template <class F>
int multiexp_entry(const uint64_t* bases, size_t n_bases, const uint64_t* scalars, size_t n, const uint8_t* density,
                   uint64_t* out, unsigned threads) {
  std::vector<uint64_t> exps(4 * n);
  for (size_t i = 0; i < n; i++) {
    Fr s;
    memcpy(s.l, scalars + 4 * i, 32);
    s.into_repr(&exps[4 * i]);
  }
  size_t need = 0;
  for (size_t i = 0; i < n; i++) need += density ? (density[i] != 0) : 1;
  if (need != n_bases) return 5;
  MexpTask<F> t;
  t.init((const Aff<F>*)bases, density, exps.data(), n);
  Pool pool(threads);
  pool.run(t.nwin, [&](size_t w) { t.run_window((uint32_t)w); });
  if (!t.ok) return 2;
  Aff<F> a = t.combine().into_affine();
  memcpy(out, &a, sizeof a);
  return 0;
}

constexpr uint32_t AUX = 0x80000000u;

struct Csr {
  const uint64_t* row_ptr;
  const uint32_t* col;
  const uint64_t* coeff;
};

}  // namespace

extern "C" {

struct orc_pk_view {
  const uint64_t *alpha_g1, *beta_g1, *beta_g2, *delta_g1, *delta_g2;
  const uint64_t* h; size_t n_h;
  const uint64_t* l; size_t n_l;
  const uint64_t* a; size_t n_a;
  const uint64_t* b_g1; size_t n_b_g1;
  const uint64_t* b_g2; size_t n_b_g2;
};
struct orc_csr { const uint64_t* row_ptr; const uint32_t* col; const uint64_t* coeff; };
struct orc_r1cs_view { size_t n_rows, n_inputs, n_aux; orc_csr a, b, c; };
struct orc_proof { uint64_t a[8], b[16], c[8]; };
struct orc_timings { double eval_s, fft_s, multiexp_s, total_s; };

unsigned orc_hardware_threads() { return std::thread::hardware_concurrency(); }

// EvaluationDomain::{fft, ifft, coset_fft, icoset_fft} on 2^log_n Montgomery Fr, in place.
int orc_fft(uint64_t* data, unsigned log_n, int inverse, int coset, unsigned threads) {
  Pool pool(threads);
  Domain d((const Fr*)data, (size_t)1 << log_n);
  if (!d.ok) return 1;
  if (!inverse) {
    if (coset) d.coset_fft(pool); else d.fft(pool);
  } else {
    if (coset) d.icoset_fft(pool); else d.ifft(pool);
  }
  memcpy(data, d.c.data(), sizeof(Fr) << log_n);
  return 0;
}

// H coefficients exactly as create_proof computes them; writes m - 1 Montgomery Fr.
int orc_h_coefficients(const uint64_t* a, const uint64_t* b, const uint64_t* c, size_t n_rows, uint64_t* h,
                       size_t* m_out, unsigned threads) {
  Pool pool(threads);
  Domain da((const Fr*)a, n_rows), db((const Fr*)b, n_rows), dc((const Fr*)c, n_rows);
  if (!da.ok) return 1;
  for (Domain* d : {&da, &db, &dc}) {
    d->ifft(pool);
    d->coset_fft(pool);
  }
  for (size_t i = 0; i < da.c.size(); i++) da.c[i] = da.c[i] * db.c[i] - dc.c[i];
  da.divide_by_z_on_coset(pool);
  da.icoset_fft(pool);
  memcpy(h, da.c.data(), (da.c.size() - 1) * sizeof(Fr));
  if (m_out) *m_out = da.c.size();
  return 0;
}

int orc_multiexp_g1(const uint64_t* bases, size_t n_bases, const uint64_t* scalars, size_t n, const uint8_t* density,
                    uint64_t out[8], unsigned threads) {
  return multiexp_entry<Fq>(bases, n_bases, scalars, n, density, out, threads);
}
int orc_multiexp_g2(const uint64_t* bases, size_t n_bases, const uint64_t* scalars, size_t n, const uint8_t* density,
                    uint64_t out[16], unsigned threads) {
  return multiexp_entry<Fq2>(bases, n_bases, scalars, n, density, out, threads);
}

// Row evaluations with bellman's `eval` (coefficient one skips the multiply) + density tracking.
int orc_eval_rows(const orc_r1cs_view* v, const uint64_t* z_in, const uint64_t* z_aux, uint64_t* a, uint64_t* b,
                  uint64_t* c, uint8_t* a_aux_density, uint8_t* b_input_density, uint8_t* b_aux_density) {
  const Fr* zi = (const Fr*)z_in;
  const Fr* za = (const Fr*)z_aux;
  const Fr one = Fr::one();
  const orc_csr* mats[3] = {&v->a, &v->b, &v->c};
  uint64_t* outs[3] = {a, b, c};
  if (a_aux_density) memset(a_aux_density, 0, v->n_aux);
  if (b_aux_density) memset(b_aux_density, 0, v->n_aux);
  if (b_input_density) memset(b_input_density, 0, v->n_inputs);
  for (size_t r = 0; r < v->n_rows; r++)
    for (int k = 0; k < 3; k++) {
      Fr acc = Fr::zero();
      for (uint64_t e = mats[k]->row_ptr[r]; e < mats[k]->row_ptr[r + 1]; e++) {
        const uint32_t col = mats[k]->col[e];
        Fr coeff;
        memcpy(coeff.l, mats[k]->coeff + 4 * e, 32);
        Fr val;
        if (col & AUX) {
          const uint32_t i = col & ~AUX;
          if (i >= v->n_aux) return 5;
          val = za[i];
          if (k == 0 && a_aux_density) a_aux_density[i] = 1;
          if (k == 1 && b_aux_density) b_aux_density[i] = 1;
        } else {
          if (col >= v->n_inputs) return 5;
          val = zi[col];
          if (k == 1 && b_input_density) b_input_density[col] = 1;
        }
        acc = acc + (coeff == one ? val : val * coeff);
      }
      memcpy(outs[k] + 4 * r, acc.l, 32);
    }
  return 0;
}

// bellman `create_proof(circuit, params, r, s)` from recorded matrices.  r, s Montgomery.
int orc_prove(const orc_pk_view* pk, const orc_r1cs_view* v, const uint64_t* z_in, const uint64_t* z_aux,
              const uint64_t r_[4], const uint64_t s_[4], orc_proof* out, unsigned threads, orc_timings* tm) {
  typedef std::chrono::steady_clock clk;
  auto t0 = clk::now();
  Pool pool(threads);
  const size_t n_rows = v->n_rows, n_in = v->n_inputs, n_aux = v->n_aux;
  Aff<Fq> delta1, alpha1, beta1;
  Aff<Fq2> delta2, beta2;
  memcpy(&delta1, pk->delta_g1, 64);
  memcpy(&alpha1, pk->alpha_g1, 64);
  memcpy(&beta1, pk->beta_g1, 64);
  memcpy(&delta2, pk->delta_g2, 128);
  memcpy(&beta2, pk->beta_g2, 128);
  if (delta1.is_inf() || delta2.is_inf()) return 2;

  std::vector<uint64_t> a(4 * n_rows), b(4 * n_rows), c(4 * n_rows);
  std::vector<uint8_t> da(n_aux), dbi(n_in), dba(n_aux);
  int st = orc_eval_rows(v, z_in, z_aux, a.data(), b.data(), c.data(), da.data(), dbi.data(), dba.data());
  if (st) return st;
  auto t1 = clk::now();

  size_t m = 0;
  std::vector<uint64_t> h(4 * (n_rows * 2 + 2));
  st = orc_h_coefficients(a.data(), b.data(), c.data(), n_rows, h.data(), &m, threads);
  if (st) return st;
  auto t2 = clk::now();

  // scalars into canonical form
  auto canon = [&](const uint64_t* src, size_t n) {
    std::vector<uint64_t> o(4 * n);
    pool.run(pool.n, [&](size_t k) {
      for (size_t i = n * k / pool.n; i < n * (k + 1) / pool.n; i++) {
        Fr s;
        memcpy(s.l, src + 4 * i, 32);
        s.into_repr(&o[4 * i]);
      }
    });
    return o;
  };
  std::vector<uint64_t> he = canon(h.data(), m - 1), zie = canon(z_in, n_in), zae = canon(z_aux, n_aux);
  size_t k_b = 0, n_a_aux = 0, n_b_aux = 0;
  for (size_t i = 0; i < n_in; i++) k_b += dbi[i];
  for (size_t i = 0; i < n_aux; i++) { n_a_aux += da[i]; n_b_aux += dba[i]; }
  if (pk->n_h + 1 < m || pk->n_l != n_aux || pk->n_a != n_in + n_a_aux || pk->n_b_g1 != k_b + n_b_aux ||
      pk->n_b_g2 != k_b + n_b_aux)
    return 4;

  MexpTask<Fq> g1[6];
  MexpTask<Fq2> g2[2];
  const Aff<Fq>* A = (const Aff<Fq>*)pk->a;
  const Aff<Fq>* B1 = (const Aff<Fq>*)pk->b_g1;
  const Aff<Fq2>* B2 = (const Aff<Fq2>*)pk->b_g2;
  g1[0].init((const Aff<Fq>*)pk->h, nullptr, he.data(), m - 1);
  g1[1].init((const Aff<Fq>*)pk->l, nullptr, zae.data(), n_aux);
  g1[2].init(A, nullptr, zie.data(), n_in);
  g1[3].init(A + n_in, da.data(), zae.data(), n_aux);
  g1[4].init(B1, dbi.data(), zie.data(), n_in);
  g1[5].init(B1 + k_b, dba.data(), zae.data(), n_aux);
  g2[0].init(B2, dbi.data(), zie.data(), n_in);
  g2[1].init(B2 + k_b, dba.data(), zae.data(), n_aux);
  // every (multiexp, window) pair is one pool task, as bellman spawns them
  std::vector<std::pair<int, uint32_t>> tasks;
  for (int q = 0; q < 6; q++) for (uint32_t w = 0; w < g1[q].nwin; w++) tasks.push_back({q, w});
  for (int q = 0; q < 2; q++) for (uint32_t w = 0; w < g2[q].nwin; w++) tasks.push_back({6 + q, w});
  pool.run(tasks.size(), [&](size_t t) {
    if (tasks[t].first < 6) g1[tasks[t].first].run_window(tasks[t].second);
    else g2[tasks[t].first - 6].run_window(tasks[t].second);
  });
  for (auto& t : g1) if (!t.ok) return 2;
  for (auto& t : g2) if (!t.ok) return 2;
  auto t3 = clk::now();

  Fr rm, sm;
  memcpy(rm.l, r_, 32);
  memcpy(sm.l, s_, 32);
  uint64_t rc[4], sc[4], rsc[4];
  rm.into_repr(rc);
  sm.into_repr(sc);
  (rm * sm).into_repr(rsc);
  Jac<Fq> h_q = g1[0].combine(), l_q = g1[1].combine();
  Jac<Fq> a_answer = g1[2].combine();
  a_answer.add(g1[3].combine());
  Jac<Fq> b1_answer = g1[4].combine();
  b1_answer.add(g1[5].combine());
  Jac<Fq2> b2_answer = g2[0].combine();
  b2_answer.add(g2[1].combine());

  Jac<Fq> d1{delta1.x, delta1.y, Fq::one()}, al1{alpha1.x, alpha1.y, Fq::one()}, be1{beta1.x, beta1.y, Fq::one()};
  Jac<Fq2> d2{delta2.x, delta2.y, Fq2::one()};
  Jac<Fq> g_a = d1;
  g_a.mul_assign(rc);
  g_a.add_mixed(alpha1);
  Jac<Fq2> g_b = d2;
  g_b.mul_assign(sc);
  g_b.add_mixed(beta2);
  Jac<Fq> g_c = d1;
  g_c.mul_assign(rsc);
  { Jac<Fq> t = al1; t.mul_assign(sc); g_c.add(t); }
  { Jac<Fq> t = be1; t.mul_assign(rc); g_c.add(t); }
  g_a.add(a_answer);
  { Jac<Fq> t = a_answer; t.mul_assign(sc); g_c.add(t); }
  g_b.add(b2_answer);
  { Jac<Fq> t = b1_answer; t.mul_assign(rc); g_c.add(t); }
  g_c.add(h_q);
  g_c.add(l_q);
  Aff<Fq> pa = g_a.into_affine(), pc = g_c.into_affine();
  Aff<Fq2> pb = g_b.into_affine();
  memcpy(out->a, &pa, 64);
  memcpy(out->b, &pb, 128);
  memcpy(out->c, &pc, 64);
  auto t4 = clk::now();
  if (tm) {
    auto sec = [](clk::time_point x, clk::time_point y) { return std::chrono::duration<double>(y - x).count(); };
    tm->eval_s = sec(t0, t1);
    tm->fft_s = sec(t1, t2);
    tm->multiexp_s = sec(t2, t4);
    tm->total_s = sec(t0, t4);
  }
  return 0;
}

}  // extern "C"
