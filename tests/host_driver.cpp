// Drives the C++ host mirror (zinc_amd/host/zinc_groth16.hpp) the way zinc-vm drives bellman:
// a Circuit whose synthesize() replays recorded rows into the constraint system, then
// create_proof(circuit, params, r, s).  Input: a binary blob written by the pytest from a golden
// fixture; exit status 0 iff the proof equals the fixture's, byte for byte.
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "../zinc_amd/host/zinc_groth16.hpp"

using namespace zinc_groth16;

struct Blob {
  std::ifstream f;
  explicit Blob(const char* path) : f(path, std::ios::binary) {}
  uint64_t u64() { uint64_t v = 0; f.read((char*)&v, 8); return v; }
  template <class T> void fill(T& dst) { f.read((char*)&dst, sizeof dst); }
  template <class T> void vec(std::vector<T>& v, size_t n) { v.resize(n); if (n) f.read((char*)v.data(), n * sizeof(T)); }
};

struct Replay : Circuit {
  std::vector<Limbs> z_in, z_aux;
  std::vector<uint64_t> rp[3];
  std::vector<uint32_t> col[3];
  std::vector<Limbs> co[3];
  void synthesize(ConstraintSystem& cs) override {
    for (size_t i = 1; i < z_in.size(); i++) cs.alloc_input(&z_in[i]);
    for (auto& v : z_aux) cs.alloc(&v);
    for (size_t r = 0; r + 1 < rp[0].size(); r++) {
      LinearCombination lc[3];
      for (int k = 0; k < 3; k++)
        for (uint64_t e = rp[k][r]; e < rp[k][r + 1]; e++) lc[k].add(co[k][e], Variable{col[k][e]});
      cs.enforce(lc[0], lc[1], lc[2]);
    }
  }
};

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  Blob b(argv[1]);
  if (!b.f) return 2;
  Replay c;
  Parameters p;
  b.fill(p.alpha_g1); b.fill(p.beta_g1); b.fill(p.beta_g2); b.fill(p.delta_g1); b.fill(p.delta_g2);
  size_t n;
  n = b.u64(); b.vec(p.h, n);
  n = b.u64(); b.vec(p.l, n);
  n = b.u64(); b.vec(p.a, n);
  n = b.u64(); b.vec(p.b_g1, n);
  n = b.u64(); b.vec(p.b_g2, n);
  n = b.u64(); b.vec(c.z_in, n);
  n = b.u64(); b.vec(c.z_aux, n);
  for (int k = 0; k < 3; k++) {
    n = b.u64(); b.vec(c.rp[k], n);
    n = b.u64(); b.vec(c.col[k], n);
    b.vec(c.co[k], n);
  }
  Limbs r, s;
  b.fill(r); b.fill(s);
  Proof want;
  b.fill(want);
  try {
    Prover prover(0);
    Proof got = create_proof(c, p, r, s, prover);
    if (memcmp(&got, &want, sizeof got) != 0) { fprintf(stderr, "proof differs from the fixture\n"); return 1; }
    // a key with delta at infinity must surface bellman's UnexpectedIdentity
    Parameters bad = p;
    bad.delta_g1.fill(0);
    try {
      create_proof(c, bad, r, s, prover);
      fprintf(stderr, "expected UnexpectedIdentity\n");
      return 1;
    } catch (const SynthesisError& e) {
      if (e.status != ZG16_ERR_UNEXPECTED_IDENTITY) { fprintf(stderr, "wrong error: %s\n", e.variant()); return 1; }
    }
    // the resident path a zinc-tester-style loop takes: the key and the circuit are uploaded by the first call only
    for (int round = 0; round < 3; round++) {
      Proof again = create_proof(c, p, r, s, std::string("fixture/main"));
      if (memcmp(&again, &want, sizeof again) != 0) { fprintf(stderr, "cached proof %d differs from the fixture\n", round); return 1; }
    }
    zg16_cache_counts cc;
    if (zg16_cache_stats(Prover::shared().ctx(), &cc) != ZG16_OK || cc.key_misses != 1 || cc.key_hits != 2 ||
        cc.circuit_misses != 1 || cc.circuit_hits != 2 || cc.keys != 1 || cc.circuits != 1) {
      fprintf(stderr, "cache counts: keys %llu (%llu hits, %llu misses), circuits %llu (%llu hits, %llu misses)\n",
              (unsigned long long)cc.keys, (unsigned long long)cc.key_hits, (unsigned long long)cc.key_misses,
              (unsigned long long)cc.circuits, (unsigned long long)cc.circuit_hits, (unsigned long long)cc.circuit_misses);
      return 1;
    }
    zg16_shutdown();
  } catch (const SynthesisError& e) {
    fprintf(stderr, "SynthesisError::%s: %s\n", e.variant(), e.what());
    return 1;
  }
  printf("host driver ok\n");
  return 0;
}
