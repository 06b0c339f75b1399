"""Groth16 prove path restated on Python big integers (bellman_ce 0.3.0 semantics).

TEST INFRASTRUCTURE ONLY -- never imported by the product package.

PARITY STATUS: "parity unpinned".  bellman_ce 0.3.0 (Cargo.lock:629-643 of the reference) is a
crates.io dependency that is not on disk; the reference's call sites are
zinc-vm/src/core/circuit/facade.rs:126 (setup), :153 (prove) and zinc-vm/src/core/facade.rs:32-38
(verify).  The reference holds no proof/key/MSM/FFT vectors.  This file restates the published
algorithm and is pinned by (i) the Groth16 pairing equation, (ii) a closed-form derivation of
A, B, C from the toxic waste that uses neither FFT nor multiexp (`closed_form_proof`),
(iii) agreement between the bellman-style FFT / windowed multiexp here and naive sums.

Column convention (shared with the C ABI): a variable is a u32; bit 31 set means aux variable,
clear means input variable; input 0 is the constant ONE.
"""
import math

from .bn254 import (R_MOD, FR_S, FR_GEN, FR_ROOT_OF_UNITY, G1, G2, pairing_product_is_one)

AUX = 1 << 31
GEN_COSET = FR_GEN  # multiplicative generator used for the coset


class SynthesisError(Exception):
    pass


class PolynomialDegreeTooLarge(SynthesisError):
    pass


class UnexpectedIdentity(SynthesisError):
    pass


class UnconstrainedVariable(SynthesisError):
    pass


# --------------------------------------------------------------------------- domain / FFT
class EvaluationDomain:
    """bellman domain.rs `EvaluationDomain::from_coeffs`: pad to 2^exp, reject exp >= 28."""

    def __init__(self, coeffs):
        m, exp = 1, 0
        while m < len(coeffs):
            m *= 2
            exp += 1
            if exp >= FR_S:
                raise PolynomialDegreeTooLarge()
        omega = FR_ROOT_OF_UNITY
        for _ in range(exp, FR_S):
            omega = omega * omega % R_MOD
        self.coeffs = list(coeffs) + [0] * (m - len(coeffs))
        self.m, self.exp, self.omega = m, exp, omega
        self.omegainv = pow(omega, -1, R_MOD)
        self.geninv = pow(GEN_COSET, -1, R_MOD)
        self.minv = pow(m, -1, R_MOD)

    def fft(self):
        self.coeffs = serial_fft(self.coeffs, self.omega, self.exp)

    def ifft(self):
        self.coeffs = [v * self.minv % R_MOD for v in serial_fft(self.coeffs, self.omegainv, self.exp)]

    def distribute_powers(self, g):
        u = 1
        for i in range(self.m):
            self.coeffs[i] = self.coeffs[i] * u % R_MOD
            u = u * g % R_MOD

    def coset_fft(self):
        self.distribute_powers(GEN_COSET)
        self.fft()

    def icoset_fft(self):
        self.ifft()
        self.distribute_powers(self.geninv)

    def z(self, tau):
        return (pow(tau, self.m, R_MOD) - 1) % R_MOD

    def divide_by_z_on_coset(self):
        i = pow(self.z(GEN_COSET), -1, R_MOD)
        self.coeffs = [v * i % R_MOD for v in self.coeffs]

    def mul_assign(self, other):
        self.coeffs = [a * b % R_MOD for a, b in zip(self.coeffs, other.coeffs)]

    def sub_assign(self, other):
        self.coeffs = [(a - b) % R_MOD for a, b in zip(self.coeffs, other.coeffs)]


def serial_fft(a, omega, log_n):
    """Bit-reversal followed by iterative radix-2 DIT; natural order in and out."""
    n = 1 << log_n
    a = list(a)
    for k in range(n):
        rk = int(format(k, "0%db" % log_n)[::-1], 2) if log_n else 0
        if k < rk:
            a[k], a[rk] = a[rk], a[k]
    m = 1
    for _ in range(log_n):
        w_m = pow(omega, n // (2 * m), R_MOD)
        for k in range(0, n, 2 * m):
            w = 1
            for j in range(m):
                t = a[k + j + m] * w % R_MOD
                a[k + j + m] = (a[k + j] - t) % R_MOD
                a[k + j] = (a[k + j] + t) % R_MOD
                w = w * w_m % R_MOD
        m *= 2
    return a


def naive_dft(a, omega):
    n = len(a)
    return [sum(a[j] * pow(omega, j * k, R_MOD) for j in range(n)) % R_MOD for k in range(n)]


# --------------------------------------------------------------------------- multiexp
def bellman_window(n):
    """Window width rule of bellman multiexp: 3 below 32 points, else ceil(ln n)."""
    return 3 if n < 32 else int(math.ceil(math.log(n)))


def multiexp(group, bases, scalars, density=None, window=None):
    """Pippenger multiexp with bellman's skipping rules.

    `scalars` are canonical ints and cover every variable; `density` (list of bools or None for
    full density) says which variables own a base; `bases` holds affine points (None = identity)
    for exactly the dense variables, in order.  A consumed identity base raises
    UnexpectedIdentity, as the upstream source builder does.
    """
    n = len(scalars)
    c = window if window is not None else bellman_window(n)
    if density is None:
        density = [True] * n
    assert len(density) == n
    # pair every dense variable with its base
    pairs = []
    bi = 0
    for i in range(n):
        if density[i]:
            pairs.append((scalars[i], bi))
            bi += 1
    assert bi <= len(bases)
    nwin = (254 + c - 1) // c  # Fr::NUM_BITS = 254
    window_sums = []
    for w in range(nwin):
        skip = w * c
        acc = group.identity
        buckets = [group.identity] * ((1 << c) - 1)
        for s, b in pairs:
            if s == 0:
                continue
            if bases[b] is None:
                raise UnexpectedIdentity()
            P = group.from_affine(bases[b])
            if s == 1:
                if skip == 0:
                    acc = group.add(acc, P)
                continue
            d = (s >> skip) & ((1 << c) - 1)
            if d:
                buckets[d - 1] = group.add(buckets[d - 1], P)
        running = group.identity
        for bkt in reversed(buckets):
            running = group.add(running, bkt)
            acc = group.add(acc, running)
        window_sums.append(acc)
    total = group.identity
    for ws in reversed(window_sums):
        for _ in range(c):
            total = group.double(total)
        total = group.add(total, ws)
    return total


def naive_msm(group, bases, scalars):
    acc = group.identity
    for b, s in zip(bases, scalars):
        if b is not None and s:
            acc = group.add(acc, group.mul(group.from_affine(b), s))
    return acc


# --------------------------------------------------------------------------- constraint systems
def eval_lc(lc, z_in, z_aux):
    """<row, z>; follows the in-tree evaluator zinc-vm/src/constraint_systems/main.rs:35-49."""
    acc = 0
    for col, coeff in lc:
        v = z_aux[col & ~AUX] if col & AUX else z_in[col]
        acc += v * coeff
    return acc % R_MOD


class ProvingAssignment:
    """bellman groth16/prover.rs `ProvingAssignment` plus a CSR recorder.

    alloc/alloc_input/enforce keep upstream's density rules: A marks aux variables only, B marks
    inputs and aux, C marks nothing.  Coefficients equal to zero must not be passed (the
    reference removes them in zinc-vm/src/constraint_systems/dedup.rs:52-66 before bellman sees
    the row).
    """

    def __init__(self):
        self.z_in, self.z_aux = [], []
        self.a, self.b, self.c = [], [], []
        self.a_aux_density, self.b_input_density, self.b_aux_density = [], [], []
        self.rows = ([], [], [])  # recorded LCs for A, B, C

    def alloc_input(self, value):
        self.z_in.append(value % R_MOD)
        self.b_input_density.append(False)
        return len(self.z_in) - 1

    def alloc(self, value):
        self.z_aux.append(value % R_MOD)
        self.a_aux_density.append(False)
        self.b_aux_density.append(False)
        return (len(self.z_aux) - 1) | AUX

    def enforce(self, a_lc, b_lc, c_lc):
        for col, coeff in a_lc:
            assert coeff % R_MOD != 0
            if col & AUX:
                self.a_aux_density[col & ~AUX] = True
        for col, coeff in b_lc:
            assert coeff % R_MOD != 0
            if col & AUX:
                self.b_aux_density[col & ~AUX] = True
            else:
                self.b_input_density[col] = True
        self.a.append(eval_lc(a_lc, self.z_in, self.z_aux))
        self.b.append(eval_lc(b_lc, self.z_in, self.z_aux))
        self.c.append(eval_lc(c_lc, self.z_in, self.z_aux))
        for k, lc in enumerate((a_lc, b_lc, c_lc)):
            self.rows[k].append([(col, coeff % R_MOD) for col, coeff in lc])

    def finalize_inputs(self):
        """The `input_i * 0 = 0` rows bellman appends after synthesis."""
        for i in range(len(self.z_in)):
            self.enforce([(i, 1)], [], [])


class Circuit:
    """A recorded R1CS with a witness: replays itself into any constraint system."""

    def __init__(self, n_in, n_aux, rows_a, rows_b, rows_c, z_in=None, z_aux=None):
        self.n_in, self.n_aux = n_in, n_aux
        self.rows_a, self.rows_b, self.rows_c = rows_a, rows_b, rows_c
        self.z_in, self.z_aux = z_in, z_aux

    def synthesize(self, cs):
        zi = self.z_in if self.z_in is not None else [0] * self.n_in
        za = self.z_aux if self.z_aux is not None else [0] * self.n_aux
        assert zi[0] == 1 or self.z_in is None
        for i in range(1, self.n_in):
            cs.alloc_input(zi[i])
        for i in range(self.n_aux):
            cs.alloc(za[i])
        for a, b, c in zip(self.rows_a, self.rows_b, self.rows_c):
            cs.enforce(a, b, c)


def prepare_assignment(circuit):
    pa = ProvingAssignment()
    pa.alloc_input(1)
    circuit.synthesize(pa)
    pa.finalize_inputs()
    return pa


# --------------------------------------------------------------------------- setup
class Parameters:
    def __init__(self):
        self.alpha_g1 = self.beta_g1 = self.delta_g1 = None
        self.beta_g2 = self.gamma_g2 = self.delta_g2 = None
        self.ic = []
        self.h, self.l, self.a, self.b_g1, self.b_g2 = [], [], [], [], []


class _KeypairAssembly:
    def __init__(self):
        self.n_in = self.n_aux = self.n_rows = 0
        self.at_in, self.bt_in, self.ct_in = [], [], []
        self.at_aux, self.bt_aux, self.ct_aux = [], [], []

    def alloc_input(self, value=None):
        self.n_in += 1
        for t in (self.at_in, self.bt_in, self.ct_in):
            t.append([])
        return self.n_in - 1

    def alloc(self, value=None):
        self.n_aux += 1
        for t in (self.at_aux, self.bt_aux, self.ct_aux):
            t.append([])
        return (self.n_aux - 1) | AUX

    def enforce(self, a_lc, b_lc, c_lc):
        for lc, tin, taux in ((a_lc, self.at_in, self.at_aux), (b_lc, self.bt_in, self.bt_aux),
                              (c_lc, self.ct_in, self.ct_aux)):
            for col, coeff in lc:
                (taux[col & ~AUX] if col & AUX else tin[col]).append((coeff % R_MOD, self.n_rows))
        self.n_rows += 1


def generate_parameters(circuit, g1, g2, alpha, beta, gamma, delta, tau):
    """bellman groth16/generator.rs `generate_parameters` with explicit toxic waste.

    g1, g2 are Jacobian group elements (upstream draws random ones, not the standard generators).
    Returns (Parameters, toxic) where `toxic` also carries the per-variable QAP evaluations so
    the closed-form cross-check can use them.
    """
    asm = _KeypairAssembly()
    asm.alloc_input()
    circuit.synthesize(asm)
    for i in range(asm.n_in):
        asm.enforce([(i, 1)], [], [])
    dom = EvaluationDomain([0] * asm.n_rows)
    m = dom.m
    powers = [1] * m
    for i in range(1, m):
        powers[i] = powers[i - 1] * tau % R_MOD
    delta_inv = pow(delta, -1, R_MOD)
    gamma_inv = pow(gamma, -1, R_MOD)
    coeff = dom.z(tau) * delta_inv % R_MOD
    pk = Parameters()
    h_scalars = [powers[i] * coeff % R_MOD for i in range(m - 1)]
    dom.coeffs = powers
    dom.ifft()
    lag = dom.coeffs  # L_j(tau)

    def ev(entries):
        return sum(c * lag[row] for c, row in entries) % R_MOD

    a_in = [ev(e) for e in asm.at_in]
    b_in = [ev(e) for e in asm.bt_in]
    c_in = [ev(e) for e in asm.ct_in]
    a_aux = [ev(e) for e in asm.at_aux]
    b_aux = [ev(e) for e in asm.bt_aux]
    c_aux = [ev(e) for e in asm.ct_aux]

    def fixed(group, g, scalars):
        return group.batch_to_affine([group.mul(g, s) for s in scalars])

    pk.h = fixed(G1, g1, h_scalars)
    ic_s = [(beta * a + alpha * b + c) * gamma_inv % R_MOD for a, b, c in zip(a_in, b_in, c_in)]
    l_s = [(beta * a + alpha * b + c) * delta_inv % R_MOD for a, b, c in zip(a_aux, b_aux, c_aux)]
    pk.ic = fixed(G1, g1, ic_s)
    pk.l = fixed(G1, g1, l_s)
    if any(p is None for p in pk.l):
        raise UnconstrainedVariable()
    a_all = fixed(G1, g1, a_in + a_aux)
    b1_all = fixed(G1, g1, b_in + b_aux)
    b2_all = fixed(G2, g2, b_in + b_aux)
    pk.a = [p for p in a_all if p is not None]
    pk.b_g1 = [p for p in b1_all if p is not None]
    pk.b_g2 = [p for p in b2_all if p is not None]
    pk.alpha_g1 = G1.to_affine(G1.mul(g1, alpha))
    pk.beta_g1 = G1.to_affine(G1.mul(g1, beta))
    pk.delta_g1 = G1.to_affine(G1.mul(g1, delta))
    pk.beta_g2 = G2.to_affine(G2.mul(g2, beta))
    pk.gamma_g2 = G2.to_affine(G2.mul(g2, gamma))
    pk.delta_g2 = G2.to_affine(G2.mul(g2, delta))
    toxic = dict(g1=g1, g2=g2, alpha=alpha, beta=beta, gamma=gamma, delta=delta, tau=tau, m=m,
                 a_in=a_in, b_in=b_in, c_in=c_in, a_aux=a_aux, b_aux=b_aux, c_aux=c_aux)
    return pk, toxic


# --------------------------------------------------------------------------- prover
def h_coefficients(a, b, c):
    """a, b, c: row evaluations.  Returns (m, the m-1 H coefficients), canonical ints."""
    da, db, dc = EvaluationDomain(a), EvaluationDomain(b), EvaluationDomain(c)
    for d in (da, db, dc):
        d.ifft()
        d.coset_fft()
    da.mul_assign(db)
    da.sub_assign(dc)
    da.divide_by_z_on_coset()
    da.icoset_fft()
    return da.m, da.coeffs[:da.m - 1]


def create_proof(circuit, pk, r, s, window=None):
    """bellman groth16/prover.rs `create_proof` with explicit r, s (canonical ints)."""
    pa = prepare_assignment(circuit)
    return create_proof_from_assignment(pa, pk, r, s, window)


def create_proof_from_assignment(pa, pk, r, s, window=None):
    if pk.delta_g1 is None or pk.delta_g2 is None:
        raise UnexpectedIdentity()
    m, h = h_coefficients(pa.a, pa.b, pa.c)
    n_in = len(pa.z_in)
    g_h = multiexp(G1, pk.h, h, None, window)
    g_l = multiexp(G1, pk.l, pa.z_aux, None, window)
    a_in = multiexp(G1, pk.a[:n_in], pa.z_in, None, window)
    a_aux = multiexp(G1, pk.a[n_in:], pa.z_aux, pa.a_aux_density, window)
    k = sum(pa.b_input_density)
    b1_in = multiexp(G1, pk.b_g1[:k], pa.z_in, pa.b_input_density, window)
    b1_aux = multiexp(G1, pk.b_g1[k:], pa.z_aux, pa.b_aux_density, window)
    b2_in = multiexp(G2, pk.b_g2[:k], pa.z_in, pa.b_input_density, window)
    b2_aux = multiexp(G2, pk.b_g2[k:], pa.z_aux, pa.b_aux_density, window)

    d1 = G1.from_affine(pk.delta_g1)
    d2 = G2.from_affine(pk.delta_g2)
    al1 = G1.from_affine(pk.alpha_g1)
    be1 = G1.from_affine(pk.beta_g1)
    be2 = G2.from_affine(pk.beta_g2)
    a_sum = G1.add(a_in, a_aux)
    g_a = G1.add(G1.add(G1.mul(d1, r), al1), a_sum)
    g_b = G2.add(G2.add(G2.mul(d2, s), be2), G2.add(b2_in, b2_aux))
    g_c = G1.mul(d1, r * s % R_MOD)
    g_c = G1.add(g_c, G1.mul(al1, s))
    g_c = G1.add(g_c, G1.mul(be1, r))
    g_c = G1.add(g_c, G1.mul(a_sum, s))
    g_c = G1.add(g_c, G1.mul(G1.add(b1_in, b1_aux), r))
    g_c = G1.add(g_c, g_h)
    g_c = G1.add(g_c, g_l)
    return (G1.to_affine(g_a), G2.to_affine(g_b), G1.to_affine(g_c))


def closed_form_proof(pa, toxic, r, s):
    """A, B, C from the toxic waste only -- no FFT, no multiexp.  Valid for satisfied witnesses."""
    t = toxic
    z = pa.z_in + pa.z_aux
    at = sum(zv * av for zv, av in zip(z, t["a_in"] + t["a_aux"])) % R_MOD
    bt = sum(zv * bv for zv, bv in zip(z, t["b_in"] + t["b_aux"])) % R_MOD
    ct = sum(zv * cv for zv, cv in zip(z, t["c_in"] + t["c_aux"])) % R_MOD
    al, be, de = t["alpha"], t["beta"], t["delta"]
    a_s = (al + r * de + at) % R_MOD
    b_s = (be + s * de + bt) % R_MOD
    laux = sum(zv * (be * a + al * b + c) for zv, a, b, c in
               zip(pa.z_aux, t["a_aux"], t["b_aux"], t["c_aux"])) % R_MOD
    # h(tau) * Z(tau) = A(tau) B(tau) - C(tau) when every row is satisfied
    c_s = ((laux + at * bt - ct) * pow(de, -1, R_MOD) + s * a_s + r * b_s - r * s * de) % R_MOD
    return (G1.to_affine(G1.mul(t["g1"], a_s)), G2.to_affine(G2.mul(t["g2"], b_s)),
            G1.to_affine(G1.mul(t["g1"], c_s)))


# --------------------------------------------------------------------------- verifier
def verify_proof(pk, proof, public_inputs):
    """bellman groth16/verifier.rs `verify_proof`; public_inputs exclude the constant ONE."""
    if len(public_inputs) + 1 != len(pk.ic):
        raise SynthesisError("MalformedVerifyingKey")
    acc = G1.from_affine(pk.ic[0])
    for x, p in zip(public_inputs, pk.ic[1:]):
        acc = G1.add(acc, G1.mul(G1.from_affine(p), x % R_MOD))
    A, B, C = proof
    neg = lambda q: None if q is None else (q[0], ((-q[1][0]) % _Q, (-q[1][1]) % _Q))
    nA = None if A is None else (A[0], (-A[1]) % _Q)
    # e(A,B) = e(alpha,beta) e(acc,gamma) e(C,delta)  <=>  e(-A,B) e(alpha,beta) e(acc,gamma) e(C,delta) = 1
    return pairing_product_is_one([
        (nA, B), (pk.alpha_g1, pk.beta_g2), (G1.to_affine(acc), pk.gamma_g2), (C, pk.delta_g2)])


from .bn254 import Q_MOD as _Q  # noqa: E402
