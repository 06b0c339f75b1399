"""rand 0.4.6 `XorShiftRng` and ff_ce's `Fr: Rand`, restated (test infrastructure only).

The reference seeds its own randomised tests with
XorShiftRng::from_seed([0x5dbe6259, 0x8d313d76, 0x3237db17, 0xe5bc0654])
(zinc-vm/src/gadgets/contract/storage.rs:219); seeded fixtures here use the same generator so
that they can be regenerated from Rust later.  rand 0.4.6 is pinned at Cargo.lock:2825-2828 and is
not on disk: "parity unpinned" -- the recurrence below is the published xorshift128.
"""
from .bn254 import R_MOD, Q_MOD, MONT_R

REFERENCE_SEED = (0x5DBE6259, 0x8D313D76, 0x3237DB17, 0xE5BC0654)
M32 = 0xFFFFFFFF


class XorShiftRng:
    def __init__(self, seed=REFERENCE_SEED):
        self.x, self.y, self.z, self.w = seed

    def next_u32(self):
        t = (self.x ^ (self.x << 11)) & M32
        self.x, self.y, self.z = self.y, self.z, self.w
        self.w = (self.w ^ (self.w >> 19) ^ (t ^ (t >> 8))) & M32
        return self.w

    def next_u64(self):
        hi = self.next_u32()
        return (hi << 32) | self.next_u32()

    def _field_raw(self, modulus):
        # four limbs, least significant first; clear the top two bits; retry until < modulus
        while True:
            raw = 0
            for i in range(4):
                raw |= self.next_u64() << (64 * i)
            raw &= (1 << 254) - 1
            if raw < modulus:
                return raw

    def gen_fr_raw(self):
        """The 256-bit pattern drawn: this IS the Montgomery representation upstream."""
        return self._field_raw(R_MOD)

    def gen_fr(self):
        """Canonical value of a drawn Fr (raw * R^-1)."""
        return self.gen_fr_raw() * pow(MONT_R, -1, R_MOD) % R_MOD

    def gen_fq(self):
        return self._field_raw(Q_MOD) * pow(MONT_R, -1, Q_MOD) % Q_MOD

    def below(self, n):
        return self.next_u64() % n
