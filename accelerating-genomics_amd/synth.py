"""Seeded synthetic inputs in the reference's two text formats.

The reference ships one generator, smithWaterman/generator.py (unseeded,
header = NUM_OF_ALIGNMENTS but 2*NUM lines); it never ships here.
This module writes the same *formats* (SW: header line + one sequence per
line, antidiagonalSmithWaterman.c:205-227; PairHMM: repeated regions
"nr nh" / nr read lines `bases quals ins del gcp` / nh haplotype lines,
antidiagsPairHMM.c:371-418) from numpy RNGs with fixed seeds, in the flat
array layout the C-ABI takes (include/agx.h), so that bench.py, the tests and
tests/golden/make_golden.py all draw the same bytes.
"""
from __future__ import annotations

import dataclasses

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
NL = 10  # '\n' : the reference SW aligns it as a symbol


@dataclasses.dataclass
class SWBatch:
    """Flat SW batch: sequence k = bases[off[k] : off[k]+len[k]]; pair p = (2p, 2p+1)."""

    bases: np.ndarray  # uint8
    off: np.ndarray  # uint64 [2n]
    len: np.ndarray  # uint32 [2n]

    @property
    def n_pairs(self) -> int:
        return self.off.size // 2

    def seq(self, k: int) -> bytes:
        o = int(self.off[k])
        return self.bases[o : o + int(self.len[k])].tobytes()

    def cells(self, sentinel: bool = True) -> int:
        """sum(len_a*len_b); sentinel=False discounts one trailing '\\n' per sequence."""
        l = self.len.astype(np.int64)
        if not sentinel:
            l = l - self._has_newline()
        return int((l[0::2] * l[1::2]).sum())

    def _has_newline(self) -> np.ndarray:
        last = np.maximum(self.off.astype(np.int64) + self.len.astype(np.int64), 1) - 1
        if self.bases.size == 0:
            return np.zeros(self.len.size, dtype=bool)
        return (self.bases[np.minimum(last, self.bases.size - 1)] == NL) & (self.len > 0)

    def algorithmic_bytes(self) -> int:
        """len_a + len_b + 4 per pair, newline excluded."""
        l = self.len.astype(np.int64) - self._has_newline()
        return int(l.sum() + 4 * self.n_pairs)

    def subset(self, pairs) -> "SWBatch":
        idx = np.asarray(pairs, dtype=np.int64)
        k = np.stack([2 * idx, 2 * idx + 1], axis=1).reshape(-1)
        return sw_from_seqs([self.seq(int(i)) for i in k])


def sw_from_seqs(seqs) -> SWBatch:
    lens = np.array([len(s) for s in seqs], dtype=np.uint32)
    off = np.zeros(len(seqs), dtype=np.uint64)
    if len(seqs) > 1:
        off[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy() if seqs else np.zeros(0, np.uint8)
    return SWBatch(bases, off, lens)


def _mutate(rng, s: np.ndarray, sub: float, indel: float) -> np.ndarray:
    out = []
    for c in s:
        r = rng.random()
        if r < indel / 2:
            continue  # deletion
        if r < indel:
            out.append(_ACGT[rng.integers(4)])  # insertion before c
        if rng.random() < sub:
            c = _ACGT[rng.integers(4)]
        out.append(c)
    return np.array(out, dtype=np.uint8)


def sw_pairs(n_pairs: int, len_lo: int, len_hi: int, seed: int, related_frac: float = 0.0,
             newline: bool = True) -> SWBatch:
    """n_pairs pairs, both lengths iid U[len_lo, len_hi] over ACGT (generator.py-style).

    A fraction `related_frac` of pairs has b = mutate(a) (2 % substitutions,
    1 % indels, trimmed/padded to its drawn length) so that long diagonals and
    gap extension are exercised.  With newline=True every
    sequence carries the trailing '\\n' the reference CLI would align.
    """
    rng = np.random.default_rng(seed)
    la = rng.integers(len_lo, len_hi + 1, size=n_pairs)
    lb = rng.integers(len_lo, len_hi + 1, size=n_pairs)
    rel = rng.random(n_pairs) < related_frac
    extra = 1 if newline else 0
    lens = np.empty(2 * n_pairs, dtype=np.uint32)
    lens[0::2] = la + extra
    lens[1::2] = lb + extra
    off = np.zeros(2 * n_pairs, dtype=np.uint64)
    off[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
    total = int(lens.sum())
    bases = _ACGT[rng.integers(0, 4, size=total, dtype=np.uint8)]
    if newline:
        bases[(off + lens - 1).astype(np.int64)] = NL
    for p in np.nonzero(rel)[0]:
        oa, ob = int(off[2 * p]), int(off[2 * p + 1])
        a = bases[oa : oa + int(la[p])]
        m = _mutate(rng, a, 0.02, 0.01)
        n = int(lb[p])
        if m.size >= n:
            bases[ob : ob + n] = m[:n]
        else:
            bases[ob : ob + m.size] = m
    return SWBatch(bases, off, lens)


def write_sw_file(path: str, b: SWBatch, header: int | None = None, final_newline: bool = True) -> None:
    """Header = number of sequence LINES so the reference consumes every pair."""
    with open(path, "wb") as f:
        f.write(b"%d\n" % (2 * b.n_pairs if header is None else header))
        for k in range(2 * b.n_pairs):
            s = b.seq(k)
            if not s.endswith(b"\n") and (k < 2 * b.n_pairs - 1 or final_newline):
                s += b"\n"
            f.write(s)


# --------------------------------------------------------------------------- PairHMM


@dataclasses.dataclass
class PhmmBatch:
    """Flat PairHMM batch (include/agx.h agx_phmm_desc).

    read r: bases/q_base/q_ins/q_del/q_gcp[roff[r]:roff[r+1]]; hap h: hap_bases[hoff[h]:hoff[h+1]];
    region g: reads [rreg[g], rreg[g+1]) x haps [hreg[g], hreg[g+1]); results are
    region-major, read-major, hap-minor (antidiagsPairHMM.c:411,440).
    """

    read_bases: np.ndarray
    q_base: np.ndarray
    q_ins: np.ndarray
    q_del: np.ndarray
    q_gcp: np.ndarray
    roff: np.ndarray  # uint64 [nr+1]
    hap_bases: np.ndarray
    hoff: np.ndarray  # uint64 [nh+1]
    rreg: np.ndarray  # uint32 [ng+1]
    hreg: np.ndarray  # uint32 [ng+1]

    @property
    def n_regions(self) -> int:
        return self.rreg.size - 1

    @property
    def n_pairs(self) -> int:
        nr = np.diff(self.rreg.astype(np.int64))
        nh = np.diff(self.hreg.astype(np.int64))
        return int((nr * nh).sum())

    def pair_lengths(self):
        """(R, H) int64 arrays in output order."""
        rl = np.diff(self.roff.astype(np.int64))
        hl = np.diff(self.hoff.astype(np.int64))
        Rs, Hs = [], []
        for g in range(self.n_regions):
            r = rl[int(self.rreg[g]) : int(self.rreg[g + 1])]
            h = hl[int(self.hreg[g]) : int(self.hreg[g + 1])]
            Rs.append(np.repeat(r, h.size))
            Hs.append(np.tile(h, r.size))
        if not Rs:
            return np.zeros(0, np.int64), np.zeros(0, np.int64)
        return np.concatenate(Rs), np.concatenate(Hs)

    def cells(self) -> int:
        R, H = self.pair_lengths()
        return int((R * H).sum())

    def algorithmic_bytes(self) -> int:
        """5R + H + 8 per pair, every pair counted as independent."""
        R, H = self.pair_lengths()
        return int((5 * R + H + 8).sum())

    def regions(self, lo: int, hi: int) -> "PhmmBatch":
        """Sub-batch of regions [lo, hi) (whole regions stay together)."""
        r0, r1 = int(self.rreg[lo]), int(self.rreg[hi])
        h0, h1 = int(self.hreg[lo]), int(self.hreg[hi])
        b0, b1 = int(self.roff[r0]), int(self.roff[r1])
        c0, c1 = int(self.hoff[h0]), int(self.hoff[h1])
        return PhmmBatch(self.read_bases[b0:b1].copy(), self.q_base[b0:b1].copy(), self.q_ins[b0:b1].copy(),
                         self.q_del[b0:b1].copy(), self.q_gcp[b0:b1].copy(),
                         (self.roff[r0 : r1 + 1] - np.uint64(b0)).astype(np.uint64), self.hap_bases[c0:c1].copy(),
                         (self.hoff[h0 : h1 + 1] - np.uint64(c0)).astype(np.uint64),
                         (self.rreg[lo : hi + 1] - np.uint32(r0)).astype(np.uint32),
                         (self.hreg[lo : hi + 1] - np.uint32(h0)).astype(np.uint32))


def phmm_repeat(b: PhmmBatch, times: int) -> PhmmBatch:
    """The batch's regions `times` times over, each copy with reads and haplotypes of its own (a corpus replicated to
    bench size: regions are repeated, reads are not concatenated)."""
    nr, nh = int(b.rreg[-1]), int(b.hreg[-1])
    nb, nc = int(b.roff[nr]), int(b.hoff[nh])
    tile = lambda a, n: np.tile(a[:n], times)
    roff = np.concatenate([b.roff[:nr].astype(np.uint64) + np.uint64(k * nb) for k in range(times)] + [np.array([times * nb], np.uint64)])
    hoff = np.concatenate([b.hoff[:nh].astype(np.uint64) + np.uint64(k * nc) for k in range(times)] + [np.array([times * nc], np.uint64)])
    ng = b.n_regions
    rreg = np.concatenate([b.rreg[:ng].astype(np.uint32) + np.uint32(k * nr) for k in range(times)] + [np.array([times * nr], np.uint32)])
    hreg = np.concatenate([b.hreg[:ng].astype(np.uint32) + np.uint32(k * nh) for k in range(times)] + [np.array([times * nh], np.uint32)])
    return PhmmBatch(tile(b.read_bases, nb), tile(b.q_base, nb), tile(b.q_ins, nb), tile(b.q_del, nb), tile(b.q_gcp, nb), roff,
                     tile(b.hap_bases, nc), hoff, rreg, hreg)


def phmm_from_regions(regions) -> PhmmBatch:
    """regions: list of (reads, haps); read = (bases, qb, qi, qd, qg) byte strings, hap = bytes."""
    rb, qb, qi, qd, qg, hb = [], [], [], [], [], []
    roff, hoff, rreg, hreg = [0], [0], [0], [0]
    for reads, haps in regions:
        for r in reads:
            assert len({len(x) for x in r}) == 1, "the five read fields must have equal length"
            rb.append(r[0]); qb.append(r[1]); qi.append(r[2]); qd.append(r[3]); qg.append(r[4])
            roff.append(roff[-1] + len(r[0]))
        for h in haps:
            hb.append(h)
            hoff.append(hoff[-1] + len(h))
        rreg.append(rreg[-1] + len(reads))
        hreg.append(hreg[-1] + len(haps))
    u8 = lambda parts: np.frombuffer(b"".join(parts), dtype=np.uint8).copy()
    return PhmmBatch(u8(rb), u8(qb), u8(qi), u8(qd), u8(qg), np.array(roff, np.uint64), u8(hb),
                     np.array(hoff, np.uint64), np.array(rreg, np.uint32), np.array(hreg, np.uint32))


def parse_phmm_text(data: bytes) -> PhmmBatch:
    """Python mirror of the reference's reading rule (used by tests on the oracle side only)."""
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    i, regions = 0, []
    while i < len(lines):
        nr, nh = (int(x) for x in lines[i].split()[:2])
        i += 1
        reads = []
        for ln in lines[i : i + nr]:
            n = (len(ln) - 4) // 5
            f = ln.split()
            reads.append(tuple(x[:n] for x in f[:5]))
        i += nr
        haps = list(lines[i : i + nh])
        i += nh
        regions.append((reads, haps))
    return phmm_from_regions(regions)


def phmm_regions(n_regions: int, reads_per: int, haps_per: int, R: int, H: int, seed: int,
                 sub_rate: float = 0.01, jitter: int = 0) -> PhmmBatch:
    """Synthetic regions shaped like test_set/10s.in (SURVEY.md 8d, C3/C5 construction).

    Per region: one random H-mer + (haps_per-1) variants (<=3 SNPs, sometimes one
    1-bp indel, padded/trimmed back to H); reads = random R-bp windows of one of the
    region's haplotypes with `sub_rate` substitutions.  Base quality ~ clip(N(30,5),
    6, 41); insertion/deletion quality chars 'H'..'N' (Q39-45); gcp '+' (Q10).
    `jitter` > 0 draws read/hap lengths from [R-jitter, R] / [H-jitter, H].
    """
    rng = np.random.default_rng(seed)
    regions = []
    for _ in range(n_regions):
        base = _ACGT[rng.integers(0, 4, size=H + 8)]
        haps = []
        for k in range(haps_per):
            h = base.copy()
            if k:
                for _s in range(int(rng.integers(1, 4))):
                    h[rng.integers(h.size)] = _ACGT[rng.integers(4)]
                if H > 3 and rng.random() < 0.5:
                    pos = int(rng.integers(1, H - 1))
                    h = np.delete(h, pos) if rng.random() < 0.5 else np.insert(h, pos, _ACGT[rng.integers(4)])
            hl = H - (int(rng.integers(0, jitter + 1)) if jitter else 0)
            haps.append(h[:hl].tobytes())
        reads = []
        for _r in range(reads_per):
            rl = R - (int(rng.integers(0, jitter + 1)) if jitter else 0)
            src = np.frombuffer(haps[int(rng.integers(haps_per))], dtype=np.uint8)
            rl = min(rl, src.size)
            st = int(rng.integers(0, src.size - rl + 1))
            r = src[st : st + rl].copy()
            m = rng.random(rl) < sub_rate
            r[m] = _ACGT[rng.integers(0, 4, size=int(m.sum()))]
            qb = (np.clip(np.rint(rng.normal(30, 5, size=rl)), 6, 41).astype(np.uint8) + 33)
            qi = rng.integers(ord("H"), ord("N") + 1, size=rl).astype(np.uint8)
            qd = rng.integers(ord("H"), ord("N") + 1, size=rl).astype(np.uint8)
            qg = np.full(rl, ord("+"), dtype=np.uint8)
            reads.append((r.tobytes(), qb.tobytes(), qi.tobytes(), qd.tobytes(), qg.tobytes()))
        regions.append((reads, haps))
    return phmm_from_regions(regions)


def write_phmm_file(path: str, b: PhmmBatch) -> None:
    with open(path, "wb") as f:
        for g in range(b.n_regions):
            r0, r1 = int(b.rreg[g]), int(b.rreg[g + 1])
            h0, h1 = int(b.hreg[g]), int(b.hreg[g + 1])
            f.write(b"%d %d\n" % (r1 - r0, h1 - h0))
            for r in range(r0, r1):
                a, z = int(b.roff[r]), int(b.roff[r + 1])
                f.write(b" ".join(x[a:z].tobytes() for x in (b.read_bases, b.q_base, b.q_ins, b.q_del, b.q_gcp)) + b"\n")
            for h in range(h0, h1):
                f.write(b.hap_bases[int(b.hoff[h]) : int(b.hoff[h + 1])].tobytes() + b"\n")


# ---------------------------------------------------------------- substitution matrices (8f n3)
AMINO = b"ARNDCQEGHILKMFPSTWYV"
# BLOSUM62 over the 20 standard residues in AMINO order (Henikoff & Henikoff 1992, half-bit units)
BLOSUM62 = [
    [4, -1, -2, -2, 0, -1, -1, 0, -2, -1, -1, -1, -1, -2, -1, 1, 0, -3, -2, 0],
    [-1, 5, 0, -2, -3, 1, 0, -2, 0, -3, -2, 2, -1, -3, -2, -1, -1, -3, -2, -3],
    [-2, 0, 6, 1, -3, 0, 0, 0, 1, -3, -3, 0, -2, -3, -2, 1, 0, -4, -2, -3],
    [-2, -2, 1, 6, -3, 0, 2, -1, -1, -3, -4, -1, -3, -3, -1, 0, -1, -4, -3, -3],
    [0, -3, -3, -3, 9, -3, -4, -3, -3, -1, -1, -3, -1, -2, -3, -1, -1, -2, -2, -1],
    [-1, 1, 0, 0, -3, 5, 2, -2, 0, -3, -2, 1, 0, -3, -1, 0, -1, -2, -1, -2],
    [-1, 0, 0, 2, -4, 2, 5, -2, 0, -3, -3, 1, -2, -3, -1, 0, -1, -3, -2, -2],
    [0, -2, 0, -1, -3, -2, -2, 6, -2, -4, -4, -2, -3, -3, -2, 0, -2, -2, -3, -3],
    [-2, 0, 1, -1, -3, 0, 0, -2, 8, -3, -3, -1, -2, -1, -2, -1, -2, -2, 2, -3],
    [-1, -3, -3, -3, -1, -3, -3, -4, -3, 4, 2, -3, 1, 0, -3, -2, -1, -3, -1, 3],
    [-1, -2, -3, -4, -1, -2, -3, -4, -3, 2, 4, -2, 2, 0, -3, -2, -1, -2, -1, 1],
    [-1, 2, 0, -1, -3, 1, 1, -2, -1, -3, -2, 5, -1, -3, -1, 0, -1, -3, -2, -2],
    [-1, -1, -2, -3, -1, 0, -2, -3, -2, 1, 2, -1, 5, 0, -2, -1, -1, -1, -1, 1],
    [-2, -3, -3, -3, -2, -3, -3, -3, -1, 0, 0, -3, 0, 6, -4, -2, -2, 1, 3, -1],
    [-1, -2, -2, -1, -3, -1, -1, -2, -2, -3, -3, -1, -2, -4, 7, -1, -1, -4, -3, -2],
    [1, -1, 1, 0, -1, 0, 0, 0, -1, -2, -2, 0, -1, -2, -1, 4, 1, -3, -2, -2],
    [0, -1, 0, -1, -1, -1, -1, -2, -2, -1, -1, -1, -1, -2, -1, 1, 5, -2, -2, 0],
    [-3, -3, -4, -4, -2, -2, -3, -2, -2, -3, -2, -3, -1, 1, -4, -3, -2, 11, 2, -3],
    [-2, -2, -2, -3, -2, -1, -2, -3, 2, -1, -1, -2, -1, 3, -3, -2, -2, 2, 7, -1],
    [0, -3, -3, -3, -1, -2, -2, -3, -3, 3, 1, -2, 1, -1, -2, -2, 0, -3, -1, 4],
]


def protein_pairs(n_pairs: int, lmin: int, lmax: int, seed: int, related_frac: float = 0.5) -> SWBatch:
    """Random protein pairs over AMINO; a fraction are mutated copies (15 % substitutions, 3 % indels)."""
    rng = np.random.default_rng(seed)
    aa = np.frombuffer(AMINO, dtype=np.uint8)
    seqs = []
    for _ in range(n_pairs):
        a = aa[rng.integers(0, 20, size=int(rng.integers(lmin, lmax + 1)))]
        if rng.random() < related_frac:
            b = a.copy()
            b[rng.random(b.size) < 0.15] = aa[rng.integers(0, 20)]
            keep = rng.random(b.size) >= 0.03
            b = b[keep]
            if b.size == 0:
                b = a[:1]
            b = b[: max(1, min(b.size, lmax))]
        else:
            b = aa[rng.integers(0, 20, size=int(rng.integers(lmin, lmax + 1)))]
        seqs += [a.tobytes(), b.tobytes()]
    return sw_from_seqs(seqs)
