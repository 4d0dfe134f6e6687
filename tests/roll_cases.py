"""The roll kernels' key streams over k and read shapes, shared by the emulator test and the GPU test."""
import ctypes as C
import random

import cqflibs
from fastq_util import chunks_by_records


def run(mk_ctx, pack):
    """mk_ctx(qb=, k=, max_batch_bytes=, max_batch_keys=) -> context. The thread-per-read roll kernels (k_roll_hist /
    k_roll_scatter, through shk_hash_route_chunks) give exactly the keys of reads_to_kmers with their chunk tags, over k
    around the 16-base rounds and 64-base units of the 2-bit staging (k_pack_reads), read lengths around those
    boundaries, lower case, 'N' (restart) and bytes that are neither (forward seed 0, complement seed seedTab[byte & 7]);
    the wave-per-read closed form (k_hash_reads) gives the same words in stream order"""
    O = cqflibs.oracle()
    rnd = random.Random(7)
    qb = 12
    hb = qb + 8
    for k in (5, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 79, 80, 81, 100, 128, 129):
        recs = []
        lens_ = [1, k - 1, k, k + 1, k + 15, k + 16, k + 17, 63, 64, 65, 127, 128, 129, 150, 191, 192, 193, 2 * k + 64, 300, 517]
        for i, L in enumerate(lens_ + [rnd.randrange(k, 400) for _ in range(30)]):
            L = max(1, L)
            s = [rnd.choice("ACGT") for _ in range(L)]
            m = i % 6
            if m == 1:       # lower case somewhere
                for _ in range(3):
                    p = rnd.randrange(L)
                    s[p] = s[p].lower()
            elif m == 2:     # an 'N' (restart when at an index >= k of its subread)
                s[rnd.randrange(L)] = "N"
            elif m == 3:     # bytes that are no base and no 'N'
                s[rnd.randrange(L)] = rnd.choice("nRYKSWDMB.-*")
            elif m == 4 and L > 2:   # several, at the ends
                s[0] = "N"
                s[-1] = "N"
                s[L // 2] = "n"
            seq = "".join(s)
            recs.append("@r%d\n%s\n+\n%s\n" % (i, seq, "I" * L))
        rnd.shuffle(recs)
        fq = "".join(recs).encode()
        offs, lens = chunks_by_records(fq, 7)
        ctx = mk_ctx(qb=qb, k=k, max_batch_bytes=1 << 20, max_batch_keys=1 << 16)
        ctx.profile(True)
        dp, counts, nw = ctx.hash_route_chunks(fq, offs, lens, 1)
        words = ctx.read_words(dp, nw)
        exp = []
        for ci, (a, n) in enumerate(zip(offs, lens)):
            exp += [kk | (ci << hb) for kk in O.chunk_keys(fq[a:a + n], k, hb)]
        assert nw == len(exp) == counts[0], k
        assert sorted(words) == sorted(exp), k
        assert ("k_pack_reads" in ctx.profile_get()) == pack
        if pack:
            dp2, nw2 = ctx.hash_chunks(fq, offs, lens)
            assert ctx.read_words(dp2, nw2) == exp, k
        ctx.close()
