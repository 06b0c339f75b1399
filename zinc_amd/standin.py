"""Hand-recorded stand-ins for the reference's real circuits (test / benchmark support; nothing here proves).

A compiled zinc program (`.znb`) cannot be produced without the reference's Rust compiler, so the constraint rows
of a zinc-tester program are recorded here by hand: a few gadgets that emit the row shapes zinc-vm's gadgets emit,
driven by a transcription of the program, through a canonicaliser that does what zinc-vm's `Dedup` wrapper does.
The public-input vector is the one the reference's test pins -- the program's expected output
(zinc-tester/tests/algorithm/factorial.zn, the `//!` header: n = 5, 6, 7 -> 120, 720, 5040; outputs are the public
inputs, zinc-vm/src/gadgets/output.rs:16-25).  Row COUNTS are not pinned: several gadgets bottom out in
franklin-crypto (`Expression::into_bits_le_fixed`, `conditionally_select`), which is not part of the reference tree;
the shapes follow what those calls are documented to enforce (one boolean row per bit, one packing row, one product
row per selection).

Row shapes and where the reference emits them:
  ONE * ONE = ONE                      zinc-vm/src/core/circuit/mod.rs:76-81
  witness: alloc + type check          zinc-vm/src/gadgets/witness.rs:38-58, gadgets/scalar/mod.rs:253-303
  bits: b * (1 - b) = 0, packing row   zinc-vm/src/gadgets/scalar/mod.rs:298-300 (into_bits_le_fixed)
  product: l * r = v                   zinc-vm/src/gadgets/arithmetic/mul.rs:38-43
  selection: c * (t - f) = v - f       zinc-vm/src/gadgets/select.rs:61-110
  output: v_in * ONE = element         zinc-vm/src/gadgets/output.rs:16-25
"""
from .fields import R_MOD

AUX = 1 << 31
ONE = 0


class DedupCS:
    """zinc-vm/src/constraint_systems/dedup.rs:30-69: every linear combination is rebuilt with one term per
    variable (coefficients of repeats added), zero coefficients dropped, inputs first then aux, ascending index."""

    def __init__(self, cs):
        self.cs = cs

    def alloc(self, value):
        return self.cs.alloc(value)

    def alloc_input(self, value):
        return self.cs.alloc_input(value)

    @staticmethod
    def dedup(lc):
        inputs, aux = {}, {}
        for var, c in lc:
            m = aux if var & AUX else inputs
            m[var] = (m.get(var, 0) + c) % R_MOD
        out = [(v, c) for v, c in sorted(inputs.items()) if c]
        out += [(v, c) for v, c in sorted(aux.items()) if c]
        return out

    def enforce(self, a, b, c):
        self.cs.enforce(self.dedup(a), self.dedup(b), self.dedup(c))


class Scalar:
    """A value with the linear combination that carries it (a constant is a multiple of ONE)."""

    def __init__(self, lc, value):
        self.lc, self.value = lc, value % R_MOD

    @classmethod
    def constant(cls, value):
        return cls([(ONE, value % R_MOD)], value)

    @classmethod
    def variable(cls, var, value):
        return cls([(var, 1)], value)

    def is_constant(self):
        return all(v == ONE for v, _ in self.lc)

    def __add__(self, o):
        return Scalar(self.lc + o.lc, self.value + o.value)

    def __sub__(self, o):
        return Scalar(self.lc + [(v, -c % R_MOD) for v, c in o.lc], self.value - o.value)

    def scaled(self, k):
        return Scalar([(v, c * k % R_MOD) for v, c in self.lc], self.value * k)


def into_bits_le_fixed(cs, x, bits):
    """`bits` boolean variables whose weighted sum is x; unsatisfiable when x does not fit."""
    assert x.value < (1 << bits), "value overflows its type (the reference raises ValueOverflow here)"
    out = []
    for i in range(bits):
        b = (x.value >> i) & 1
        var = cs.alloc(b)
        cs.enforce([(var, 1)], [(ONE, 1), (var, R_MOD - 1)], [])     # b * (1 - b) = 0
        out.append(Scalar.variable(var, b))
    cs.enforce([(s.lc[0][0], 1 << i) for i, s in enumerate(out)], [(ONE, 1)], x.lc)   # packing row
    return out


def select(cs, cond, if_true, if_false):
    """cond * (t - f) = v - f; constants fold as the reference's auto_const does."""
    if cond.is_constant():
        return if_true if cond.value else if_false
    value = if_true.value if cond.value else if_false.value
    var = cs.alloc(value)
    v = Scalar.variable(var, value)
    cs.enforce(cond.lc, (if_true - if_false).lc, (v - if_false).lc)
    return v


def conditional_type_check(cs, cond, x, bits):
    """gadgets/scalar/mod.rs:253-303: the value checked is `cond ? x : 0`, so a dead branch cannot overflow."""
    if x.is_constant():
        assert x.value < (1 << bits)
        return x
    into_bits_le_fixed(cs, select(cs, cond, x, Scalar.constant(0)), bits)
    return x


def allocate_witness(cs, value, bits):
    var = cs.alloc(value)
    return conditional_type_check(cs, Scalar.constant(1), Scalar.variable(var, value), bits)


def mul(cs, left, right):
    if left.is_constant():
        return right.scaled(left.value)
    if right.is_constant():
        return left.scaled(right.value)
    value = left.value * right.value % R_MOD
    var = cs.alloc(value)
    cs.enforce(left.lc, right.lc, [(var, 1)])
    return Scalar.variable(var, value)


def lesser_or_equals(cs, left, right, bits):
    """left <= right for `bits`-bit unsigned values: bit `bits` of right - left + 2^bits."""
    diff = right - left + Scalar.constant(1 << bits)
    return into_bits_le_fixed(cs, diff, bits + 1)[bits]


def logical_and(cs, a, b):
    return mul(cs, a, b)


def output(cs, element):
    var = cs.alloc_input(element.value)
    cs.enforce([(var, 1)], [(ONE, 1)], element.lc)
    return Scalar.variable(var, element.value)


class FactorialCircuit:
    """zinc-tester/tests/algorithm/factorial.zn:

        const INDEX: u8 = 8;
        fn main(n: u8) -> u240 {
            let mut fact: u240 = 1;
            for i in 1..INDEX while i <= n { fact *= i as u240; }
            fact
        }

    `storage_root`: when given, one more public input follows the output, as contract methods append their
    storage root hash (zinc-vm/src/core/contract/mod.rs:205-217)."""

    INDEX = 8
    CASES = {5: 120, 6: 720, 7: 5040}   # the reference's pinned (input, output) pairs

    def __init__(self, n, storage_root=None):
        self.n, self.storage_root = n, storage_root
        self.result = None

    def synthesize(self, cs):
        cs = DedupCS(cs)
        cs.enforce([(ONE, 1)], [(ONE, 1)], [(ONE, 1)])
        n = allocate_witness(cs, self.n, 8)
        fact = Scalar.constant(1)
        active = Scalar.constant(1)        # the `while` condition so far: once false it stays false
        for i in range(1, self.INDEX):
            cond = lesser_or_equals(cs, Scalar.constant(i), n, 8)
            active = logical_and(cs, active, cond)
            product = mul(cs, fact, Scalar.constant(i))
            conditional_type_check(cs, active, product, 240)
            fact = select(cs, active, product, fact)
        self.result = fact.value
        output(cs, fact)
        if self.storage_root is not None:
            root = cs.alloc(self.storage_root)
            rs = Scalar.variable(root, self.storage_root)
            # (a field-typed witness gets the reference's dummy row: gadgets/witness.rs:43-46)
            cs.enforce((rs + Scalar.constant(1)).lc, [(ONE, 1)], (rs + Scalar.constant(1)).lc)
            output(cs, rs)

    def public_inputs(self):
        """What a verifier is given (without the leading ONE)."""
        out = [self.result]
        if self.storage_root is not None:
            out.append(self.storage_root % R_MOD)
        return out


def record(circuit):
    """bellman's prover-side bookkeeping around `synthesize`: ONE is input 0, the `input_i * 0 = 0` rows follow."""
    from .groth16 import ProvingAssignment
    pa = ProvingAssignment()
    pa.alloc_input(1)
    circuit.synthesize(pa)
    pa.finalize_inputs()
    return pa
