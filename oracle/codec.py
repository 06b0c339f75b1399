"""pairing_ce 0.17.0 BN256 point encodings and bellman's Proof / VerifyingKey / Parameters layouts,
restated (test infrastructure only; "parity unpinned": the crates are not on disk and the reference
holds no key or proof files -- its call sites are zinc-vm/src/zvm/arguments/command/prove.rs:171-174
(`proof.write`), setup.rs:67-75 (`params.write`, `params.vk.write`), verify.rs:61,76 (`read`)).

Encoding rules restated: coordinates are big-endian canonical integers; byte 0 bit 6 marks the
point at infinity (all other bits zero); in compressed form bit 7 says y is the larger of (y, -y)
under the field's ordering (Fq2 compares c1 first, then c0); Fq2 values are written c1 then c0.
"""
from .bn254 import Q_MOD


def _be(x):
    return x.to_bytes(32, "big")


def _fq2_gt(a, b):
    return (a[1], a[0]) > (b[1], b[0])


def g1_compressed(p):
    if p is None:
        return bytes([0x40]) + bytes(31)
    x, y = p
    out = bytearray(_be(x))
    if y > (-y) % Q_MOD:
        out[0] |= 0x80
    return bytes(out)


def g2_compressed(p):
    if p is None:
        return bytes([0x40]) + bytes(63)
    (x0, x1), (y0, y1) = p
    out = bytearray(_be(x1) + _be(x0))
    if _fq2_gt((y0, y1), ((-y0) % Q_MOD, (-y1) % Q_MOD)):
        out[0] |= 0x80
    return bytes(out)


def g1_uncompressed(p):
    if p is None:
        return bytes([0x40]) + bytes(63)
    return _be(p[0]) + _be(p[1])


def g2_uncompressed(p):
    if p is None:
        return bytes([0x40]) + bytes(127)
    (x0, x1), (y0, y1) = p
    return _be(x1) + _be(x0) + _be(y1) + _be(y0)


def proof_bytes(proof):
    """bellman `Proof::write`: A (G1 compressed) || B (G2 compressed) || C (G1 compressed) = 128 bytes."""
    a, b, c = proof
    return g1_compressed(a) + g2_compressed(b) + g1_compressed(c)


def vk_bytes(pk):
    """bellman `VerifyingKey::write`."""
    out = g1_uncompressed(pk.alpha_g1) + g1_uncompressed(pk.beta_g1) + g2_uncompressed(pk.beta_g2)
    out += g2_uncompressed(pk.gamma_g2) + g1_uncompressed(pk.delta_g1) + g2_uncompressed(pk.delta_g2)
    out += len(pk.ic).to_bytes(4, "big")
    for p in pk.ic:
        out += g1_uncompressed(p)
    return out


def params_bytes(pk):
    """bellman `Parameters::write`: vk, then h, l, a, b_g1 (G1) and b_g2 (G2), each u32-BE length + points."""
    out = vk_bytes(pk)
    for vec in (pk.h, pk.l, pk.a, pk.b_g1):
        out += len(vec).to_bytes(4, "big")
        for p in vec:
            out += g1_uncompressed(p)
    out += len(pk.b_g2).to_bytes(4, "big")
    for p in pk.b_g2:
        out += g2_uncompressed(p)
    return out
