"""Hand-recorded stand-ins for the reference's real circuits (test / benchmark support; nothing here proves).

A compiled zinc program (`.znb`) cannot be produced without the reference's Rust compiler, so the constraint rows
of a few programs are recorded here by hand: gadgets that emit the rows zinc-vm's gadgets emit, driven by a
transcription of the program, through a canonicaliser that does what zinc-vm's `Dedup` wrapper does.
The public-input vector is the one the reference's tests pin -- a program's expected output in the `//!` header of
its zinc-tester file (outputs are the public inputs, zinc-vm/src/gadgets/output.rs:16-25; contract methods append
their storage root, zinc-vm/src/core/contract/mod.rs:205-217).  Row COUNTS are not pinned: several gadgets bottom out
in franklin-crypto (`Expression::into_bits_le_fixed`, `into_bits_le_strict`, `conditionally_select`,
`AllocatedNum::equals`, `pack_bits_to_element`), which is not part of the reference tree; those are restated from
the published sapling-crypto circuits they derive from and marked so below.

Gadgets and where the reference emits their rows:
  ONE * ONE = ONE                          zinc-vm/src/core/circuit/mod.rs:76-81
  witness: alloc + type check / dummy row  zinc-vm/src/gadgets/witness.rs:18-58, gadgets/scalar/mod.rs:253-303
  add: (l + r) * ONE = v                   zinc-vm/src/gadgets/arithmetic/add.rs:13-60
  mul: l * r = v                           zinc-vm/src/gadgets/arithmetic/mul.rs:13-60
  div_rem: q * d = n - r, 0 <= r < |d|     zinc-vm/src/gadgets/arithmetic/div_rem.rs:13-105
  selection: (t - f) * c = v - f           zinc-vm/src/gadgets/select.rs:61-110
  not / and / or                           zinc-vm/src/gadgets/logical/{not,and,or}.rs
  comparisons                              zinc-vm/src/gadgets/comparison/mod.rs:21-176
  require: x * inv = ONE                   zinc-vm/src/gadgets/require.rs:14-48, instructions/require.rs:15-29
  output: v_in * ONE = element             zinc-vm/src/gadgets/output.rs:16-25
Only operations on two constants fold away (zinc-vm/src/gadgets/auto_const.rs:66-90); a constant condition picks
its branch without a row (select.rs:73-79) and a constant needs no type check (scalar/mod.rs:273-276).
"""
from .fields import R_MOD

AUX = 1 << 31
ONE = 0
FR_BITS = 254


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


TRUE, FALSE = Scalar.constant(1), Scalar.constant(0)


def _fresh(cs, value):
    value %= R_MOD
    return Scalar.variable(cs.alloc(value), value)


# ---- franklin-crypto pieces, restated (not in the reference tree) ----------------------------------------------
def alloc_bit(cs, b):
    """AllocatedBit::alloc: (1 - a) * a = 0."""
    var = cs.alloc(b)
    cs.enforce([(ONE, 1), (var, R_MOD - 1)], [(var, 1)], [])
    return Scalar.variable(var, b)


def into_bits_le_fixed(cs, x, bits):
    """Expression::into_bits_le_fixed: `bits` boolean variables whose weighted sum is x (one packing row);
    unsatisfiable when x does not fit."""
    assert x.value < (1 << bits), "value overflows its type (the reference raises ValueOverflow here)"
    out = [alloc_bit(cs, (x.value >> i) & 1) for i in range(bits)]
    cs.enforce([(s.lc[0][0], 1 << i) for i, s in enumerate(out)], [(ONE, 1)], x.lc)
    return out


def into_bits_le_strict(cs, x):
    """into_bits_le_strict: the 254 bits of x with the pattern constrained to be at most r - 1.  Walking r - 1 from
    its top bit: a bit under a one of r - 1 is a plain boolean and joins the current run; at a zero of r - 1 the run
    is AND-ed together (with the previous runs' result) and the bit may be set only if that AND is false:
    (1 - run - a) * a = 0.  One unpacking row at the end."""
    top = R_MOD - 1
    result, current_run, last_run = [], [], None
    for i in reversed(range(FR_BITS)):
        a = (x.value >> i) & 1
        if (top >> i) & 1:
            bit = alloc_bit(cs, a)
            current_run.append(bit)
        else:
            if current_run:
                if last_run is not None:
                    current_run.append(last_run)
                acc = current_run[0]
                for nxt in current_run[1:]:     # kary_and: a chain of AllocatedBit::and rows
                    v = _fresh(cs, acc.value * nxt.value)
                    cs.enforce(acc.lc, nxt.lc, v.lc)
                    acc = v
                last_run, current_run = acc, []
            var = cs.alloc(a)                   # AllocatedBit::alloc_conditionally
            cs.enforce([(ONE, 1), (last_run.lc[0][0], R_MOD - 1), (var, R_MOD - 1)], [(var, 1)], [])
            assert not (a and last_run.value), "not a canonical field element"
            bit = Scalar.variable(var, a)
        result.append(bit)
    assert not current_run          # r is odd: r - 1 ends in a zero
    result.reverse()                # little-endian
    cs.enforce([], [], [(b.lc[0][0], 1 << i) for i, b in enumerate(result)] + [(v, -c % R_MOD) for v, c in x.lc])
    return result


def pack_bits_to_element(cs, bits):
    """AllocatedNum::pack_bits_to_element: (sum 2^i b_i - num) * ONE = 0."""
    value = sum(b.value << i for i, b in enumerate(bits))
    num = _fresh(cs, value)
    cs.enforce([(b.lc[0][0], 1 << i) for i, b in enumerate(bits)] + [(num.lc[0][0], R_MOD - 1)], [(ONE, 1)], [])
    return num


def num_equals(cs, a, b):
    """AllocatedNum::equals: r boolean; (a - b) * inv = 1 - r; r * (a - b) = 0."""
    delta = (a.value - b.value) % R_MOD
    r = alloc_bit(cs, 0 if delta else 1)
    inv = _fresh(cs, pow(delta, -1, R_MOD) if delta else 0)
    d = a - b
    cs.enforce(d.lc, inv.lc, (TRUE - r).lc)
    cs.enforce(r.lc, d.lc, [])
    return r


def expression_into_number(cs, x):
    """Expression::into_number: expr * ONE = num."""
    num = _fresh(cs, x.value)
    cs.enforce(x.lc, [(ONE, 1)], num.lc)
    return num


# ---- zinc-vm gadgets ---------------------------------------------------------------------------------------------
def select(cs, cond, if_true, if_false):
    """gadgets/select.rs:61-110: (t - f) * c = v - f; a constant condition picks its branch."""
    if cond.is_constant():
        return if_true if cond.value else if_false
    v = _fresh(cs, if_true.value if cond.value else if_false.value)
    cs.enforce((if_true - if_false).lc, cond.lc, (v - if_false).lc)
    return v


def conditional_type_check(cs, cond, x, bits):
    """gadgets/scalar/mod.rs:253-303: the value checked is `cond ? x : 0`, so a dead branch cannot overflow."""
    if x.is_constant():
        assert x.value < (1 << bits)
        return x
    into_bits_le_fixed(cs, select(cs, cond, x, FALSE), bits)
    return x


def allocate_witness(cs, value, bits):
    """gadgets/witness.rs:18-58; bits None: a field element, which gets a dummy `x + 1` row instead of a check."""
    x = _fresh(cs, value)
    if bits is None:
        add(cs, x, TRUE)
        return x
    return conditional_type_check(cs, TRUE, x, bits)


def add(cs, left, right):
    if left.is_constant() and right.is_constant():
        return Scalar.constant(left.value + right.value)
    v = _fresh(cs, left.value + right.value)
    cs.enforce((left + right).lc, [(ONE, 1)], v.lc)
    return v


def mul(cs, left, right):
    if left.is_constant() and right.is_constant():
        return Scalar.constant(left.value * right.value)
    v = _fresh(cs, left.value * right.value)
    cs.enforce(left.lc, right.lc, v.lc)
    return v


def logical_not(cs, x):
    if x.is_constant():
        return Scalar.constant(1 - x.value)
    return expression_into_number(cs, TRUE - x)


def logical_and(cs, a, b):
    if a.is_constant() and b.is_constant():
        return Scalar.constant(a.value * b.value)
    v = _fresh(cs, a.value * b.value)
    cs.enforce(a.lc, b.lc, v.lc)
    return v


def logical_or(cs, a, b):
    if a.is_constant() and b.is_constant():
        return Scalar.constant(1 if a.value or b.value else 0)
    v = _fresh(cs, 1 if a.value or b.value else 0)
    cs.enforce((TRUE - a).lc, (TRUE - b).lc, (TRUE - v).lc)
    return v


def less_than_integer(cs, length, left, right):
    """comparison/mod.rs:153-175: bit `length` of (2^length - 1) - left + right."""
    expr = Scalar.constant((1 << length) - 1) - left + right
    return into_bits_le_fixed(cs, expr, length + 1)[length]


def less_than_field(cs, left, right):
    """comparison/mod.rs:93-151: strict bit decompositions, halves packed and compared as integers."""
    bits_a, bits_b = into_bits_le_strict(cs, left), into_bits_le_strict(cs, right)
    lower = FR_BITS // 2
    upper = FR_BITS - lower
    a_lower, b_lower = pack_bits_to_element(cs, bits_a[:lower]), pack_bits_to_element(cs, bits_b[:lower])
    a_upper, b_upper = pack_bits_to_element(cs, bits_a[lower:]), pack_bits_to_element(cs, bits_b[lower:])
    upper_lt = less_than_integer(cs, upper, a_upper, b_upper)
    lower_lt = less_than_integer(cs, lower, a_lower, b_lower)
    upper_eq = num_equals(cs, a_upper, b_upper)
    lower_and_eq = logical_and(cs, lower_lt, upper_eq)            # Boolean::and
    # Boolean::and(not upper_lt, not lower_and_eq).not(): (1 - u) * (1 - l) = n, result 1 - n through from_boolean
    n = _fresh(cs, (1 - upper_lt.value) * (1 - lower_and_eq.value))
    cs.enforce((TRUE - upper_lt).lc, (TRUE - lower_and_eq).lc, n.lc)
    return expression_into_number(cs, TRUE - n)


def lesser_than(cs, left, right, bits):
    """bits None: field operands."""
    if left.is_constant() and right.is_constant():
        return Scalar.constant(1 if left.value < right.value else 0)
    return less_than_field(cs, left, right) if bits is None else less_than_integer(cs, bits, left, right)


def greater_than(cs, left, right, bits):
    return lesser_than(cs, right, left, bits)


def lesser_or_equals(cs, left, right, bits):
    return logical_not(cs, greater_than(cs, left, right, bits))


def greater_or_equals(cs, left, right, bits):
    return lesser_or_equals(cs, right, left, bits)


def require(cs, condition, value):
    """instructions/require.rs:15-29 + gadgets/require.rs: (value or not condition) * inverse = ONE."""
    x = logical_or(cs, value, logical_not(cs, condition))
    assert x.value, "require failed (the reference raises RequireError here)"
    inv = _fresh(cs, pow(x.value, -1, R_MOD))
    cs.enforce(x.lc, inv.lc, [(ONE, 1)])


def div_rem(cs, condition, left, right, bits):
    """Unsigned integer division as instructions/operators/arithmetic/div.rs:41-55 runs it: div_rem_conditional
    (the denominator is `condition ? right : 1`), then the quotient's type check.  The remainder's range is
    checked on FIELD-typed values (div_rem.rs:85-101): two field comparisons, then (1 - lt) * (1 - ge) = 0."""
    denom = select(cs, condition, right, TRUE)
    if left.is_constant() and denom.is_constant():
        assert denom.value, "division by zero"
        return Scalar.constant(left.value // denom.value), Scalar.constant(left.value % denom.value)
    assert denom.value, "division by zero (the reference raises DivisionByZero here)"
    q, r = _fresh(cs, left.value // denom.value), _fresh(cs, left.value % denom.value)
    cs.enforce(q.lc, denom.lc, (left - r).lc)
    lt = lesser_than(cs, r, denom, None)                # |denominator| of an unsigned type is itself (abs.rs:25-27)
    ge = greater_or_equals(cs, r, FALSE, None)
    cs.enforce((TRUE - lt).lc, (TRUE - ge).lc, [])
    return conditional_type_check(cs, condition, q, bits), r


def output(cs, element):
    var = cs.alloc_input(element.value)
    cs.enforce([(var, 1)], [(ONE, 1)], element.lc)
    return Scalar.variable(var, element.value)


# ---- programs -------------------------------------------------------------------------------------------------------
class _Program:
    """Common shape: `result` after synthesis, the public inputs a verifier is given (without the leading ONE)."""

    result = None
    storage_root = None

    def public_inputs(self):
        out = list(self.result) if isinstance(self.result, (list, tuple)) else [self.result]
        if self.storage_root is not None:
            out.append(self.storage_root % R_MOD)
        return out

    def _finish_contract(self, cs):
        """Contract methods end with their storage root as one more public input (core/contract/mod.rs:205-217)."""
        output(cs, self._root)

    def _begin_contract(self, cs):
        """core/contract/mod.rs:166-178: the root hash enters as a field witness and takes part in one `+ 0` row."""
        self._root = _fresh(cs, self.storage_root)
        add(cs, self._root, FALSE)


class FactorialCircuit(_Program):
    """zinc-tester/tests/algorithm/factorial.zn:

        const INDEX: u8 = 8;
        fn main(n: u8) -> u240 {
            let mut fact: u240 = 1;
            for i in 1..INDEX while i <= n { fact *= i as u240; }
            fact
        }

    `storage_root`: when given, the circuit is wrapped the way a contract method is."""

    INDEX = 8
    CASES = {5: 120, 6: 720, 7: 5040}   # the reference's pinned (input, output) pairs

    def __init__(self, n, storage_root=None):
        self.n, self.storage_root = n, storage_root

    def synthesize(self, cs):
        cs = DedupCS(cs)
        cs.enforce([(ONE, 1)], [(ONE, 1)], [(ONE, 1)])
        if self.storage_root is not None:
            self._begin_contract(cs)
        n = allocate_witness(cs, self.n, 8)
        fact = Scalar.constant(1)
        active = TRUE                      # the `while` condition so far: once false it stays false
        for i in range(1, self.INDEX):
            cond = lesser_or_equals(cs, Scalar.constant(i), n, 8)
            active = logical_and(cs, active, cond)
            product = mul(cs, fact, Scalar.constant(i))
            conditional_type_check(cs, active, product, 240)
            fact = select(cs, active, product, fact)
        self.result = fact.value
        output(cs, fact)
        if self.storage_root is not None:
            self._finish_contract(cs)


class FibonacciCircuit(_Program):
    """zinc-tester/tests/algorithm/fibonacci.zn:

        const INDEX: u8 = 16;
        fn main(n: u8) -> u248 {
            let mut value_1: u248 = 0;
            let mut value_2: u248 = 1;
            let mut fibo = value_1;
            for i in 1..=INDEX while i + 2 <= n {
                fibo = value_1 + value_2;
                value_1 = value_2;
                value_2 = fibo;
            }
            fibo
        }"""

    INDEX = 16
    CASES = {12: 89, 14: 233, 16: 610}

    def __init__(self, n):
        self.n = n

    def synthesize(self, cs):
        cs = DedupCS(cs)
        cs.enforce([(ONE, 1)], [(ONE, 1)], [(ONE, 1)])
        n = allocate_witness(cs, self.n, 8)
        v1, v2, fibo = Scalar.constant(0), Scalar.constant(1), Scalar.constant(0)
        active = TRUE
        for i in range(1, self.INDEX + 1):
            cond = lesser_or_equals(cs, Scalar.constant(i + 2), n, 8)     # i + 2 folds: both constants
            active = logical_and(cs, active, cond)
            s = conditional_type_check(cs, active, add(cs, v1, v2), 248)
            fibo = select(cs, active, s, fibo)
            v1, v2 = select(cs, active, v2, v1), select(cs, active, fibo, v2)
        self.result = fibo.value
        output(cs, fibo)


class DivisionU8Circuit(_Program):
    """zinc-tester/tests/operator/arithmetic/division_u8.zn: fn main(a: u8, b: u8) -> u8 { a / b }"""

    CASES = {(42, 7): 6, (48, 7): 6, (49, 7): 7, (255, 1): 255, (255, 255): 1}

    def __init__(self, a, b):
        self.a, self.b = a, b

    def synthesize(self, cs):
        cs = DedupCS(cs)
        cs.enforce([(ONE, 1)], [(ONE, 1)], [(ONE, 1)])
        a, b = allocate_witness(cs, self.a, 8), allocate_witness(cs, self.b, 8)
        q, _ = div_rem(cs, TRUE, a, b, 8)
        self.result = q.value
        output(cs, q)


class UniswapInputPriceCircuit(_Program):
    """zinc-examples/uniswap/src/main.zn:705-712 with utils.zn:20-34 (Balance = u248, types.zn:5):

        pub fn get_eth_to_token_input_price(self, eth_sold: Balance) -> Balance {
            require(eth_sold > ZERO_BALANCE);
            let token_reserve: Balance = self.balance_of(self.token_address);
            utils::get_input_price(eth_sold, self.balance, token_reserve)
        }
        pub fn get_input_price(input_amount, input_reserve, output_reserve) -> Balance {
            require(input_reserve > ZERO_INTEGER && output_reserve > ZERO_INTEGER);
            let input_amount_with_fee: Balance = input_amount * 997 as Balance;
            let numerator: Balance = input_amount_with_fee * output_reserve;
            let denominator: Balance = (input_reserve * 1000 as Balance) + input_amount_with_fee;
            numerator / denominator
        }

    A contract method: the storage root is the last public input.  Storage fields arrive as typed witnesses -- in
    this fork the Merkle-path enforcement of a storage load is commented out
    (zinc-vm/src/gadgets/contract/storage.rs:83-109) -- and `balances` is a placeholder array upstream
    (main.zn:36), so `self.balance_of(self.token_address)` is one u248 storage value here.  The reference holds no
    expected output for this method: the public input is the formula's value."""

    def __init__(self, eth_sold, balance, token_reserve, storage_root, token_address=0x1234):
        self.eth_sold, self.balance, self.token_reserve = eth_sold, balance, token_reserve
        self.token_address, self.storage_root = token_address, storage_root

    @staticmethod
    def price(eth_sold, balance, token_reserve):
        fee = eth_sold * 997
        return fee * token_reserve // (balance * 1000 + fee)

    def synthesize(self, cs):
        cs = DedupCS(cs)
        cs.enforce([(ONE, 1)], [(ONE, 1)], [(ONE, 1)])
        self._begin_contract(cs)
        eth_sold = allocate_witness(cs, self.eth_sold, 248)
        require(cs, TRUE, greater_than(cs, eth_sold, FALSE, 248))
        allocate_witness(cs, self.token_address, 160)                    # self.token_address
        token_reserve = allocate_witness(cs, self.token_reserve, 248)    # self.balances[token_address]
        balance = allocate_witness(cs, self.balance, 248)                # self.balance
        both = logical_and(cs, greater_than(cs, balance, FALSE, 248), greater_than(cs, token_reserve, FALSE, 248))
        require(cs, TRUE, both)
        with_fee = conditional_type_check(cs, TRUE, mul(cs, eth_sold, Scalar.constant(997)), 248)
        numerator = conditional_type_check(cs, TRUE, mul(cs, with_fee, token_reserve), 248)
        scaled = conditional_type_check(cs, TRUE, mul(cs, balance, Scalar.constant(1000)), 248)
        denominator = conditional_type_check(cs, TRUE, add(cs, scaled, with_fee), 248)
        q, _ = div_rem(cs, TRUE, numerator, denominator, 248)
        self.result = q.value
        assert self.result == self.price(self.eth_sold, self.balance, self.token_reserve)
        output(cs, q)
        self._finish_contract(cs)


def record(circuit):
    """bellman's prover-side bookkeeping around `synthesize`: ONE is input 0, the `input_i * 0 = 0` rows follow."""
    from .groth16 import ProvingAssignment
    pa = ProvingAssignment()
    pa.alloc_input(1)
    circuit.synthesize(pa)
    pa.finalize_inputs()
    return pa
