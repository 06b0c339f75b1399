"""Conversions between Python integers and the Montgomery limb arrays that cross the C ABI."""
import numpy as np

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
Q_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583
MONT_R = 1 << 256
_RINV = {R_MOD: pow(MONT_R, -1, R_MOD), Q_MOD: pow(MONT_R, -1, Q_MOD)}


def ints_to_mont(values, modulus):
    """List of canonical ints -> (n, 4) uint64 array of Montgomery limbs."""
    buf = b"".join(((v % modulus) * MONT_R % modulus).to_bytes(32, "little") for v in values)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()


def mont_to_ints(arr, modulus):
    """(n, 4) uint64 Montgomery limbs -> list of canonical ints."""
    rinv = _RINV[modulus]
    raw = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 4).tobytes()
    return [int.from_bytes(raw[i:i + 32], "little") * rinv % modulus for i in range(0, len(raw), 32)]


def g1_to_mont(points):
    """List of affine G1 points ((x, y) or None for infinity) -> (n, 8) uint64."""
    flat = []
    for p in points:
        flat.extend((0, 0) if p is None else (p[0], p[1]))
    out = ints_to_mont(flat, Q_MOD).reshape(-1, 8)
    for i, p in enumerate(points):
        if p is None:
            out[i] = 0
    return out


def g2_to_mont(points):
    """List of affine G2 points (((x0, x1), (y0, y1)) or None) -> (n, 16) uint64."""
    flat = []
    for p in points:
        flat.extend((0, 0, 0, 0) if p is None else (p[0][0], p[0][1], p[1][0], p[1][1]))
    out = ints_to_mont(flat, Q_MOD).reshape(-1, 16)
    for i, p in enumerate(points):
        if p is None:
            out[i] = 0
    return out


def mont_to_g1(arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 8)
    out = []
    for row in arr:
        if not row.any():
            out.append(None)
        else:
            x, y = mont_to_ints(row.reshape(2, 4), Q_MOD)
            out.append((x, y))
    return out


def mont_to_g2(arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 16)
    out = []
    for row in arr:
        if not row.any():
            out.append(None)
        else:
            x0, x1, y0, y1 = mont_to_ints(row.reshape(4, 4), Q_MOD)
            out.append(((x0, x1), (y0, y1)))
    return out
