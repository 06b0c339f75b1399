// C++ host-side mirror of the bellman Groth16 prover interface zinc-vm programs against, on top of
// the C ABI in include/zinc_groth16.h.  (The reference's host is Rust; no Rust toolchain is available
// to build it here, so the same interface is provided in C++ -- rust/zinc-groth16-sys holds the Rust
// source a maintainer would compile.)
//
// Names and meanings follow the reference's call sites:
//   Circuit::synthesize(cs)            zinc-vm/src/core/circuit/synthesizer.rs:33-38
//   cs.alloc / alloc_input / enforce   zinc-vm/src/constraint_systems/dedup.rs:79-127
//   create_random_proof / create_proof zinc-vm/src/core/circuit/facade.rs:140-153
//   SynthesisError                     zinc-vm/src/error.rs:124-128
// ProvingAssignment here *records* (CSR + witness) instead of evaluating rows on the host; the GPU
// evaluates them.  It reproduces bellman's bookkeeping exactly: ONE is input 0, one
// `input_i * 0 = 0` row is appended per input after synthesis, rows keep enforce() order.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/zinc_groth16.h"

namespace zinc_groth16 {

using Limbs = std::array<uint64_t, 4>;  // Montgomery Fr, little-endian limbs (ff_ce layout)

// R mod r: the Montgomery form of one
inline Limbs fr_one() { return {0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full}; }

struct SynthesisError : std::runtime_error {
  zg16_status status;
  SynthesisError(zg16_status s, const std::string& what) : std::runtime_error(what), status(s) {}
  // bellman variant names
  const char* variant() const {
    switch (status) {
      case ZG16_ERR_POLYNOMIAL_DEGREE_TOO_LARGE: return "PolynomialDegreeTooLarge";
      case ZG16_ERR_UNEXPECTED_IDENTITY: return "UnexpectedIdentity";
      case ZG16_ERR_ASSIGNMENT_MISSING: return "AssignmentMissing";
      case ZG16_ERR_MALFORMED_KEY: return "MalformedVerifyingKey";
      default: return "IoError";
    }
  }
};

struct Variable {
  uint32_t raw;  // bit 31 set = aux
  static Variable input(uint32_t i) { return {i}; }
  static Variable aux(uint32_t i) { return {i | 0x80000000u}; }
};

// Terms must already be in zinc-vm's canonical form (zinc-vm/src/constraint_systems/dedup.rs:50-68).
struct LinearCombination {
  std::vector<std::pair<Variable, Limbs>> terms;
  LinearCombination& add(const Limbs& coeff, Variable v) {
    terms.emplace_back(v, coeff);
    return *this;
  }
};

class ConstraintSystem {
 public:
  virtual ~ConstraintSystem() = default;
  static Variable one() { return Variable::input(0); }
  virtual Variable alloc(const Limbs* value) = 0;        // nullptr = AssignmentMissing
  virtual Variable alloc_input(const Limbs* value) = 0;
  virtual void enforce(const LinearCombination& a, const LinearCombination& b, const LinearCombination& c) = 0;
  virtual void push_namespace(const char*) {}
  virtual void pop_namespace() {}
};

class Circuit {
 public:
  virtual ~Circuit() = default;
  virtual void synthesize(ConstraintSystem& cs) = 0;
};

class ProvingAssignment final : public ConstraintSystem {
 public:
  // witness_only: the circuit's matrices are already resident (zg16_r1cs_cached knew its id), so rows are dropped
  // and synthesis pays for the witness alone.  (bellman passes `enforce` three closures; the Rust equivalent never
  // calls them, which removes the per-row cost of zinc-vm/src/constraint_systems/dedup.rs:30-69.)
  explicit ProvingAssignment(bool witness_only = false) : witness_only_(witness_only) {}
  bool witness_only() const { return witness_only_; }
  std::vector<Limbs> z_in, z_aux;
  std::vector<uint64_t> row_ptr[3] = {{0}, {0}, {0}};
  std::vector<uint32_t> col[3];
  std::vector<Limbs> coeff[3];

  Variable alloc(const Limbs* value) override {
    if (!value) throw SynthesisError(ZG16_ERR_ASSIGNMENT_MISSING, "alloc without a value");
    z_aux.push_back(*value);
    return Variable::aux((uint32_t)z_aux.size() - 1);
  }
  Variable alloc_input(const Limbs* value) override {
    if (!value) throw SynthesisError(ZG16_ERR_ASSIGNMENT_MISSING, "alloc_input without a value");
    z_in.push_back(*value);
    return Variable::input((uint32_t)z_in.size() - 1);
  }
  void enforce(const LinearCombination& a, const LinearCombination& b, const LinearCombination& c) override {
    if (witness_only_) return;
    const LinearCombination* lcs[3] = {&a, &b, &c};
    for (int k = 0; k < 3; k++) {
      for (const auto& t : lcs[k]->terms) {
        col[k].push_back(t.first.raw);
        coeff[k].push_back(t.second);
      }
      row_ptr[k].push_back(col[k].size());
    }
  }
  // the rows bellman appends after synthesis
  void finalize_inputs() {
    if (witness_only_) return;
    LinearCombination empty;
    for (uint32_t i = 0; i < z_in.size(); i++) enforce(LinearCombination().add(fr_one(), Variable::input(i)), empty, empty);
  }
  zg16_r1cs_view view() const {
    zg16_r1cs_view v{};
    v.n_rows = row_ptr[0].size() - 1;
    v.n_inputs = z_in.size();
    v.n_aux = z_aux.size();
    zg16_csr* m[3] = {&v.a, &v.b, &v.c};
    for (int k = 0; k < 3; k++) {
      m[k]->row_ptr = row_ptr[k].data();
      m[k]->col = col[k].data();
      m[k]->coeff = (const uint64_t*)coeff[k].data();
    }
    return v;
  }

 private:
  bool witness_only_;
};

// bellman `Parameters<Bn256>`: affine points as Montgomery limbs (x || y; G2: x.c0 x.c1 y.c0 y.c1).
struct Parameters {
  std::array<uint64_t, 8> alpha_g1{}, beta_g1{}, delta_g1{};
  std::array<uint64_t, 16> beta_g2{}, gamma_g2{}, delta_g2{};
  std::vector<std::array<uint64_t, 8>> ic, h, l, a, b_g1;
  std::vector<std::array<uint64_t, 16>> b_g2;
  zg16_pk_view view() const {
    zg16_pk_view v{};
    v.alpha_g1 = alpha_g1.data(); v.beta_g1 = beta_g1.data(); v.beta_g2 = beta_g2.data();
    v.delta_g1 = delta_g1.data(); v.delta_g2 = delta_g2.data();
    v.h = (const uint64_t*)h.data(); v.n_h = h.size();
    v.l = (const uint64_t*)l.data(); v.n_l = l.size();
    v.a = (const uint64_t*)a.data(); v.n_a = a.size();
    v.b_g1 = (const uint64_t*)b_g1.data(); v.n_b_g1 = b_g1.size();
    v.b_g2 = (const uint64_t*)b_g2.data(); v.n_b_g2 = b_g2.size();
    return v;
  }
};

using Proof = zg16_proof;

// A GPU plus the handles a prover service keeps alive between proofs.
class Prover {
 public:
  explicit Prover(int device = 0) {
    zg16_status st = zg16_ctx_create(device, &ctx_);
    if (st != ZG16_OK) throw SynthesisError(st, "cannot create a GPU context");
  }
  ~Prover() {
    zg16_pk_free(pk_);
    zg16_r1cs_free(r1cs_);
    if (owns_ctx_) zg16_ctx_destroy(ctx_);
  }
  Prover(const Prover&) = delete;
  Prover& operator=(const Prover&) = delete;

  // The process-wide prover of a GPU: it works on zg16_default_ctx, whose cache keeps keys and circuits resident
  // between create_proof calls.  Never destroyed (the context lives until zg16_shutdown).
  static Prover& shared(int device = 0) {
    static Prover* provers[64] = {};
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (device < 0 || device >= 64) throw SynthesisError(ZG16_ERR_INVALID_ARGUMENT, "bad device index");
    if (!provers[device]) {
      zg16_ctx* ctx = nullptr;
      zg16_status st = zg16_default_ctx(device, &ctx);
      if (st != ZG16_OK) throw SynthesisError(st, "cannot create a GPU context");
      provers[device] = new Prover(ctx);
    }
    return *provers[device];
  }
  // Resident handles from the context's cache (owned by the context, not by this object).
  bool use_cached_key(const Parameters& p) {
    release_own();
    zg16_pk_view v = p.view();
    int hit = 0;
    check(zg16_pk_cached(ctx_, &v, 0, &pk_, &hit));
    return hit != 0;
  }
  // circuit_id empty: the matrices are digested.  pa null: look-up only; false when the circuit is unknown.
  bool use_cached_circuit(const std::string& circuit_id, const ProvingAssignment* pa) {
    zg16_r1cs_free(r1cs_);
    r1cs_ = nullptr;
    zg16_r1cs_view v{};
    if (pa) v = pa->view();
    int hit = 0;
    check(zg16_r1cs_cached(ctx_, circuit_id.empty() ? nullptr : (const uint8_t*)circuit_id.data(), circuit_id.size(),
                           pa ? &v : nullptr, &r1cs_, &hit));
    return r1cs_ != nullptr;
  }

  void load_key(const Parameters& p) {
    zg16_pk_free(pk_);
    pk_ = nullptr;
    zg16_pk_view v = p.view();
    check(zg16_pk_upload(ctx_, &v, 0, 1, &pk_));
  }
  void load_circuit(const ProvingAssignment& pa) {
    zg16_r1cs_free(r1cs_);
    r1cs_ = nullptr;
    zg16_r1cs_view v = pa.view();
    check(zg16_r1cs_upload(ctx_, &v, &r1cs_));
  }
  Proof prove(const ProvingAssignment& pa, const Limbs& r, const Limbs& s) {
    Proof out;
    // zg16_prove reads as many witness elements as the resident circuit has variables: a circuit known only by its
    // id must have the shape this synthesis produced (bellman: AssignmentMissing), or the buffers would be overrun
    size_t n_in = 0, n_aux = 0;
    check(zg16_r1cs_shape(r1cs_, nullptr, &n_in, &n_aux));
    if (n_in != pa.z_in.size() || n_aux != pa.z_aux.size())
      throw SynthesisError(ZG16_ERR_ASSIGNMENT_MISSING, "the resident circuit has another number of variables than the witness");
    check(zg16_prove(ctx_, pk_, r1cs_, (const uint64_t*)pa.z_in.data(), (const uint64_t*)pa.z_aux.data(), 0, r.data(),
                     s.data(), &out));
    return out;
  }
  zg16_ctx* ctx() { return ctx_; }

 private:
  explicit Prover(zg16_ctx* shared_ctx) : ctx_(shared_ctx), owns_ctx_(false) {}
  void release_own() {
    zg16_pk_free(pk_);  // (ignores handles the cache owns)
    pk_ = nullptr;
  }
  void check(zg16_status st) {
    if (st != ZG16_OK) throw SynthesisError(st, zg16_last_error(ctx_));
  }
  zg16_ctx* ctx_ = nullptr;
  bool owns_ctx_ = true;
  zg16_pk* pk_ = nullptr;
  zg16_r1cs* r1cs_ = nullptr;
};

// bellman `create_proof(circuit, params, r, s)`
inline Proof create_proof(Circuit& circuit, const Parameters& params, const Limbs& r, const Limbs& s, Prover& prover) {
  ProvingAssignment pa;
  Limbs one = fr_one();
  pa.alloc_input(&one);
  circuit.synthesize(pa);
  pa.finalize_inputs();
  prover.load_key(params);
  prover.load_circuit(pa);
  return prover.prove(pa, r, s);
}

// The same against the process-wide prover: key and matrices stay resident between calls, so a loop like
// zinc-tester's (zinc-tester/src/runners/proof_check.rs:122) uploads each key once.  circuit_id (any bytes naming
// the circuit, e.g. a digest of bytecode + method; empty = digest the matrices) lets later proofs of a known
// circuit synthesize the witness only.  The shared context, its caches and the handles they hand out belong to the
// whole process, so one proof at a time goes through here: callers on several threads queue on the lock.
inline Proof create_proof(Circuit& circuit, const Parameters& params, const Limbs& r, const Limbs& s,
                          const std::string& circuit_id = std::string(), int device = 0) {
  static std::mutex one_proof_at_a_time;
  std::lock_guard<std::mutex> lock(one_proof_at_a_time);
  Prover& prover = Prover::shared(device);
  const bool known = !circuit_id.empty() && prover.use_cached_circuit(circuit_id, nullptr);
  ProvingAssignment pa(known);
  Limbs one = fr_one();
  pa.alloc_input(&one);
  circuit.synthesize(pa);
  pa.finalize_inputs();
  prover.use_cached_key(params);
  if (!known) prover.use_cached_circuit(circuit_id, &pa);
  return prover.prove(pa, r, s);
}

// bellman `verify_proof(&prepare_verifying_key(&vk), &proof, &public_inputs)` (zinc-vm/src/core/facade.rs:32-38).
// public_inputs without the leading ONE; throws MalformedVerifyingKey when their count does not fit `ic`.
inline bool verify_proof(const Parameters& params, const Proof& proof, const std::vector<Limbs>& public_inputs) {
  zg16_vk_view vk{};
  vk.alpha_g1 = params.alpha_g1.data(); vk.beta_g2 = params.beta_g2.data();
  vk.gamma_g2 = params.gamma_g2.data(); vk.delta_g2 = params.delta_g2.data();
  vk.ic = (const uint64_t*)params.ic.data(); vk.n_ic = params.ic.size();
  int ok = 0;
  zg16_status st = zg16_verify_proof(&vk, &proof, (const uint64_t*)public_inputs.data(), public_inputs.size(), &ok);
  if (st != ZG16_OK) throw SynthesisError(st, "verify_proof");
  return ok != 0;
}

// ff_ce `Fr::rand`: 4 x u64 (limb 0 first), clear the top two bits, retry until below r.  The pattern
// drawn is used as the Montgomery representation.  `next_u64` is the caller's RNG.
template <class Rng>
Limbs random_fr(Rng&& next_u64) {
  static const uint64_t M[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
  for (;;) {
    Limbs v;
    for (auto& l : v) l = next_u64();
    v[3] &= 0x3fffffffffffffffull;
    for (int i = 3; i >= 0; i--) {
      if (v[i] < M[i]) return v;
      if (v[i] > M[i]) break;
    }
  }
}

// bellman `create_random_proof(circuit, params, rng)`: r first, then s.
template <class Rng>
Proof create_random_proof(Circuit& circuit, const Parameters& params, Rng&& next_u64, Prover& prover) {
  const Limbs r = random_fr(next_u64);
  const Limbs s = random_fr(next_u64);
  return create_proof(circuit, params, r, s, prover);
}

}  // namespace zinc_groth16
