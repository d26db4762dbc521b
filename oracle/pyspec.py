"""Pure-Python-int restatement of SPEC.md §1-§5 (second, independent implementation of the integer work).

TEST INFRASTRUCTURE ONLY — never imported by the product package.  PARITY UNPINNED: the reference snapshot
has no code (only /root/reference/README.md:2 describes the product); this file and oracle/meepo_oracle.c are
two independent in-repo implementations of SPEC.md that must agree (tests/test_oracle_kat.py), and their
agreed outputs are the committed KAT fixture tests/golden/hash_kat.json.
"""
from __future__ import annotations

import struct

M64 = (1 << 64) - 1
EMPTY_KEY = -(1 << 63)
RECLAIMED_KEY = EMPTY_KEY + 1
BUCKET_W = 16


def u64(key: int) -> int:
    return key & M64


def mix64(x: int) -> int:  # SPEC §1, splitmix64 finaliser
    x &= M64
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & M64
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & M64
    x ^= x >> 31
    return x


def mix64b(x: int) -> int:  # SPEC §1, murmur3 fmix64
    x &= M64
    x ^= x >> 33
    x = (x * 0xFF51AFD7ED558CCD) & M64
    x ^= x >> 33
    x = (x * 0xC4CEB9FE1A85EC53) & M64
    x ^= x >> 33
    return x


def mulhi64(a: int, b: int) -> int:
    return ((a & M64) * (b & M64)) >> 64


def bucket(key: int, n_buckets: int) -> int:
    return mulhi64(mix64(u64(key)), n_buckets)


def next_prime(n: int) -> int:
    """SPEC §2: smallest prime >= n."""
    n = max(n, 2)
    while True:
        if n < 4 or (n % 2 and all(n % d for d in range(3, int(n ** 0.5) + 1, 2))):
            return n
        n += 1


def step(key: int, n_buckets: int) -> int:
    return 1 + mulhi64(mix64b(u64(key)), n_buckets - 1) if n_buckets > 1 else 1


def owner(key: int, n_shards: int) -> int:
    return mulhi64(mix64b(u64(key)), n_shards)


def _f32(x: float) -> float:
    return struct.unpack("<f", struct.pack("<f", x))[0]


def initial_row(key: int, dim: int, init_scale: float, init_seed: int) -> list[float]:
    """SPEC §3 'Initial row', initializer=UNIFORM."""
    row = []
    for j in range(dim):
        h = mix64(u64(key) ^ mix64((init_seed + j) & M64))
        u = (h >> 40) / float(1 << 24)  # exact
        row.append(_f32(_f32(init_scale) * (2.0 * u - 1.0)))  # one rounding: the final multiply
    return row


def splitmix64_stream(seed: int, i: int) -> int:
    """Synthetic key generator used by tests/bench: i-th output of splitmix64 seeded with `seed`,
    reinterpreted as int64."""
    z = mix64((seed + (i + 1) * 0x9E3779B97F4A7C15) & M64)
    return z - (1 << 64) if z >> 63 else z


class DictTable:
    """Model of SPEC §2-§3 on a Python dict (insert/assign/find/export semantics incl. last-wins and the
    bucketised placement rule, so TABLE_FULL behaviour is modelled too)."""

    def __init__(self, capacity: int, dim: int, default_value: float = 0.0):
        self.n_buckets = next_prime((capacity + BUCKET_W - 1) // BUCKET_W)
        self.capacity = self.n_buckets * BUCKET_W
        self.dim = dim
        self.default_value = default_value
        self.slots: list[int | None] = [None] * self.capacity
        self.rows: dict[int, list[float]] = {}
        self.full = False

    TOMB = "tomb"

    def _probe(self, key: int):
        """(slot of key | None, slot a new key would take | None) — SPEC §2 incl. RECLAIMED reuse."""
        b = bucket(key, self.n_buckets)
        stride = step(key, self.n_buckets)
        tomb = None
        for _ in range(self.n_buckets):
            base = b * BUCKET_W
            empty = None
            for j in range(BUCKET_W):
                v = self.slots[base + j]
                if v == key and v is not self.TOMB:
                    return base + j, None
                if v is None and empty is None:
                    empty = base + j
                if v is self.TOMB and tomb is None:
                    tomb = base + j
            if empty is not None:
                return None, (tomb if tomb is not None else empty)
            b = (b + stride) % self.n_buckets
        return None, tomb

    def remove(self, keys):
        before = [k in self.rows for k in keys]
        for k in keys:
            if k in self.rows:
                s, _ = self._probe(k)
                self.slots[s] = self.TOMB
                del self.rows[k]
        return before

    def insert(self, keys, rows):
        for k, r in zip(keys, rows):
            if k in (EMPTY_KEY, RECLAIMED_KEY):
                continue
            s, e = self._probe(k)
            if s is None:
                if e is None:
                    self.full = True
                    continue
                self.slots[e] = k
            self.rows[k] = list(r)

    def assign(self, keys, rows):
        found = []
        for k, r in zip(keys, rows):
            ok = k in self.rows
            if ok:
                self.rows[k] = list(r)
            found.append(ok)
        return found

    def find(self, keys):
        out, found = [], []
        for k in keys:
            ok = k in self.rows
            out.append(list(self.rows[k]) if ok else [self.default_value] * self.dim)
            found.append(ok)
        return out, found

    def size(self):
        return len(self.rows)

    def export_sorted(self):
        ks = sorted(self.rows)
        return ks, [self.rows[k] for k in ks]
