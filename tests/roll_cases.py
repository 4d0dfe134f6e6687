"""The roll kernels' key streams over k and read shapes, shared by the emulator test and the GPU test."""
import ctypes as C
import random

import cqflibs
import synth
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


def run_slots(mk_ctx):
    """the last partition level gives every region a fixed-capacity slot instead of counting first (k_rp_slot_cursors, no
    k_rp_hist for that level); a region that gets more words than its slot holds (here: a batch of poly-A reads, one k-mer
    thousands of times) raises SHK_E_SLOT_FULL inside the library and the level is redone with exact bases; after two such
    batches in a row the context stops trying. Tables equal the oracle's throughout."""
    k, qb = 21, 14
    uni = synth.make_fastq(synth.make_genome(3000, 5), 250, 100, 0.005, seed=8, n_frac=0.02)
    poly = "".join("@p%d\n%s\n+\n%s\n" % (i, "A" * 100, "I" * 100) for i in range(60)).encode()
    ctx = mk_ctx(qb=qb, k=k, max_batch_bytes=1 << 20, max_batch_keys=1 << 16, max_level_bits=2)
    ctx.profile(True)
    O = cqflibs.oracle()
    q = O.new(qb)

    def batch(fq, per):
        offs, lens = chunks_by_records(fq, per)
        ctx.profile_reset()
        ctx.count_chunks(fq, offs, lens)
        q.reads_to_kmers(fq, k)
        assert ctx.blocks() == q.blocks()
        return ctx.profile_get()

    # three levels of 2 bits: the roll kernels' histogram pass counts the first two, the last one has slots of 2048 for
    # 64 regions x ~310 words: no counting pass over the keys at all
    p = batch(uni, 50)
    assert p["k_rp_slot_cursors"][0] == 1 and "k_rp_hist" not in p
    p = batch(poly, 20)                # 4800 times one key: its region overflows, the level runs again the exact way
    assert p["k_rp_slot_cursors"][0] == 1 and p["k_rp_hist"][0] == 1
    p = batch(uni, 50)                 # one overflow does not switch the slots off
    assert p["k_rp_slot_cursors"][0] == 1 and "k_rp_hist" not in p
    batch(poly, 20)
    batch(poly, 20)                    # the second overflow in a row does: counting passes from here on
    p = batch(uni, 50)
    assert "k_rp_slot_cursors" not in p and p["k_rp_hist"][0] == 1
    assert (ctx.totals().nelts, ctx.totals().ndistinct) == (q.nelts(), q.ndistinct())
    ctx.close()
    q.free()


def run_fused_point_with_a_crowded_region(mk_ctx):
    """the one-pass deNoise point keeps two 16-bit counters per key; a region that receives 2^15 words or more in the
    batch (poly-A reads: one k-mer 34,000 times) makes it hand the point to the general path (SHK_E_FUSED inside the
    library). Rounds, removed counts and table bytes equal the oracle's t = 1 build either way."""
    from fastq_util import oracle_t1, oracle_header
    k, qb = 21, 14
    uni = synth.make_fastq(synth.make_genome(2500, 15), 260, 100, 0.005, seed=18)
    poly = "".join("@p%d\n%s\n+\n%s\n" % (i, "A" * 100, "I" * 100) for i in range(430)).encode()
    for crowded in (False, True):
        fq = uni + (poly if crowded else b"") + synth.make_fastq(synth.make_genome(2500, 16), 260, 100, 0.005, seed=19, name_prefix="s")
        offs, lens = chunks_by_records(fq, 40)
        q, orounds, oremoved = oracle_t1(fq, offs, lens, k, qb, 3500, 2, False, 1 << 20)
        assert not q.full() and orounds >= 1
        ctx = mk_ctx(qb=qb, k=k, trigger=3500, num_denoise=2, min_denoise_len=1 << 20, max_batch_bytes=1 << 21, max_batch_keys=1 << 17,
                     max_level_bits=2)
        ctx.profile(True)
        st = ctx.count_chunks(fq, offs, lens)
        assert (st["denoise_rounds"], st["removed"]) == (orounds, oremoved)
        assert ctx.blocks() == q.blocks() and ctx.header() == oracle_header(q)
        assert "k_region_merge<fused>" in ctx.profile_get()      # the one-pass point was tried
        ctx.close()
        q.free()
