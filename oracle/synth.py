"""Small synthetic R1CS instances shaped like zinc-vm output (test infrastructure only).

Row shapes follow what the reference's gadgets emit: boolean rows (b)(1-b)=0
(zinc-vm/src/gadgets/scalar/mod.rs:170-187), product rows (zinc-vm/src/gadgets/arithmetic/mul.rs:38-43),
short linear rows and bit-packing rows with power-of-two coefficients
(zinc-vm/src/gadgets/scalar/mod.rs:298-300), output rows (out)(1)=(lc)
(zinc-vm/src/gadgets/output.rs:20-25).  Every linear combination is in the canonical form
produced by zinc-vm/src/constraint_systems/dedup.rs:50-68: inputs first, then aux, ascending,
no zero coefficients, no repeated variable.
"""
from .bn254 import R_MOD
from .groth16 import AUX, Circuit
from .xorshift import XorShiftRng


def canon(lc):
    acc = {}
    for col, coeff in lc:
        acc[col] = (acc.get(col, 0) + coeff) % R_MOD
    items = [(c, v) for c, v in acc.items() if v]
    items.sort(key=lambda cv: (1 if cv[0] & AUX else 0, cv[0] & ~AUX))
    return items


def make_circuit(n_rows, n_pub=2, rng=None, pack_width=12, dense=False):
    """n_rows enforce() rows (the input rows bellman adds are extra); returns a satisfied Circuit."""
    rng = rng or XorShiftRng()
    z_in = [1]
    z_aux = []
    rows_a, rows_b, rows_c = [], [], []
    bits = []

    def new_aux(v):
        z_aux.append(v % R_MOD)
        return (len(z_aux) - 1) | AUX

    def val(col):
        return z_aux[col & ~AUX] if col & AUX else z_in[col]

    def pick():
        return rng.below(len(z_aux)) | AUX

    def emit(a, b, c):
        rows_a.append(canon(a)); rows_b.append(canon(b)); rows_c.append(canon(c))

    # seed variables: one uniform value constrained by a dummy row, as zinc does for witnesses
    x0 = new_aux(rng.gen_fr())
    emit([(x0, 1)], [(0, 1)], [(x0, 1)])
    while len(rows_a) < n_rows - n_pub:
        kind = rng.below(100)
        if kind < 70 and not dense:
            b = new_aux(rng.below(2))
            bits.append(b)
            emit([(b, 1)], [(0, 1), (b, R_MOD - 1)], [])
        elif kind < 90:
            l = [(pick(), 1)] + ([(pick(), rng.below(5) + 1)] if rng.below(2) else [])
            r_ = [(pick(), 1)] + ([(0, rng.gen_fr())] if rng.below(2) else [])
            lv = sum(val(c) * k for c, k in l) % R_MOD
            rv = sum(val(c) * k for c, k in r_) % R_MOD
            w = new_aux(lv * rv)
            emit(l, r_, [(w, 1)])
        elif kind < 99 or len(bits) < pack_width:
            terms = [(pick(), rng.gen_fr() if rng.below(2) else R_MOD - 1) for _ in range(3 + rng.below(6))]
            terms = canon(terms)
            y = new_aux(sum(val(c) * k for c, k in terms))
            emit(terms, [(0, 1)], [(y, 1)])
        else:
            chosen = bits[-pack_width:]
            terms = [(b, 1 << i) for i, b in enumerate(chosen)]
            y = new_aux(sum(val(c) * k for c, k in terms))
            emit(terms, [(0, 1)], [(y, 1)])
    # public outputs: (out)(1) = (lc)
    for _ in range(n_pub):
        src = pick()
        z_in.append(val(src))
        emit([(len(z_in) - 1, 1)], [(0, 1)], [(src, 1)])
    return Circuit(len(z_in), len(z_aux), rows_a, rows_b, rows_c, z_in, z_aux)
