"""Seeded synthetic FASTQ generator: uniform-random genome, reads at
uniform positions / random strand, per-base substitution errors, a fraction of reads
with an N run, optional short reads. 4-line records, '\\n' endings."""
import numpy as np

_COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    _COMP[a] = b
_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_genome(G, seed):
    rng = np.random.default_rng(seed)
    return _BASES[rng.integers(0, 4, size=G)]


def make_fastq(genome, nreads, L, err, seed, n_frac=0.001, short_frac=0.0, lower_frac=0.0,
               name_prefix="r", plus_repeats_name=False, iupac_frac=0.0):
    """returns bytes of a FASTQ file. iupac_frac: share of reads with a few bytes that are neither a base nor 'N' (IUPAC
    codes, '.', '-', 'n'): seed 0 on the forward strand, seedTab[byte & 7] on the complement strand, no restart"""
    rng = np.random.default_rng(seed)
    G = len(genome)
    out = []
    pos = rng.integers(0, G - L + 1, size=nreads)
    strand = rng.integers(0, 2, size=nreads)
    for i in range(nreads):
        ln = L
        if short_frac and rng.random() < short_frac:
            ln = int(rng.integers(1, L))
        s = genome[pos[i]:pos[i] + ln].copy()
        if strand[i]:
            s = _COMP[s[::-1]]
        if err > 0:
            m = rng.random(ln) < err
            if m.any():
                s[m] = _BASES[(np.searchsorted(_BASES, s[m]) + rng.integers(1, 4, size=int(m.sum()))) % 4]
        if n_frac and rng.random() < n_frac:
            a = int(rng.integers(0, ln))
            b = min(ln, a + int(rng.integers(1, 4)))
            s[a:b] = ord("N")
        if lower_frac and rng.random() < lower_frac:
            s = np.frombuffer(bytes(s).lower(), dtype=np.uint8)
        if iupac_frac and rng.random() < iupac_frac:
            s = s.copy()
            for p in rng.integers(0, ln, size=int(rng.integers(1, 4))):
                s[p] = b"RYKMSWBDHVn.-*U"[int(rng.integers(0, 15))]
        name = f"@{name_prefix}{i}".encode()
        qual = b"I" * ln
        plus = b"+" + (name[1:] if plus_repeats_name else b"")
        out.append(name + b"\n" + bytes(s) + b"\n" + plus + b"\n" + qual + b"\n")
    return b"".join(out)
