"""Contiger, whole pipeline: the device path (shk_unitigs_add_reads -> shk_unitig_set_write) against the sequential
restatement of the whole program (oracle/contiger_pipeline.cpp) on the same filter and reads. Shared by the emulator
test (tests/test_emu_kernels.py) and the GPU test (tests/test_gpu_parity.py).

Two schedules:
  * read by read -- one call per read: the device then takes seeds, walks and queued contigs in the order the sequential
    program does, and EVERYTHING order-free must be equal: canonical sequences, km, KC, the link set;
  * batched -- all chunks in one call (what bin/Contiger does by default per chunk): who finds a unitig first differs, as
    it does between two runs of the reference with more than one thread; sequences and links must still be equal, and
    every km must be a value some schedule of the reference produces (unitig_compare.admissible_km).
PARITY UNPINNED against the reference's output (see the oracle file's header): this is reading against reading."""
import numpy as np

import cqflibs
import synth
import unitig_compare as UC
from fastq_util import chunks_by_records, oracle_t1


def reads(G, nreads, L, err, plasmid=0, seed=41):
    """a genome with two repeats (real branches), sequencing errors (tips, bubbles) and optionally a circular plasmid
    whose reads run round the circle (pure circles)"""
    g = synth.make_genome(G, seed)
    a, b = G // 3, G // 12
    g = np.concatenate([g[:a], g[b:b + max(40, G // 40)], g[a:], g[2 * G // 3:2 * G // 3 + max(30, G // 80)]])
    fq = synth.make_fastq(g, nreads, L, err, seed=seed + 4, n_frac=0.02, lower_frac=0.02)
    if plasmid:
        p = synth.make_genome(plasmid, seed + 2)
        circ = np.concatenate([p, p, p, p[:plasmid // 3]])
        fq += synth.make_fastq(circ, max(8, nreads // 5), min(L, plasmid), 0.0, seed=seed + 6, name_prefix="p")
    return fq


def seed_kmers(fq, k):
    """the k-mers the reads offer as seeds under either rule (contig_assembly.cpp:2067-2071 / :1860-1863), upper-cased"""
    out = set()
    for line in fq.split(b"\n")[1::4]:
        n = len(line)
        if n < k:
            continue
        for mid in (n // 2, n // 2 - k // 2):
            if mid <= n - k:
                km = line[mid:mid + k].upper()
                if b"N" not in km:
                    out.add(km)
    return out


def run_case(mk_ctx, UnitigSet, tmp_path, k, qb, fq, chunk_reads, amin=2, xmin=2, xmax=1000000, max_len=1 << 16, per_read=True,
             rule=1):
    offs, lens = chunks_by_records(fq, chunk_reads)
    q, _, _ = oracle_t1(fq, offs, lens, k, qb)
    assert not q.full()
    ctx = mk_ctx(qb=qb, k=k, max_batch_bytes=len(fq) + 1024, max_batch_keys=max(1 << 14, fq.count(b"\n") * 40))
    ctx.count_chunks(fq, offs, lens)
    assert ctx.blocks() == q.blocks()
    u = UnitigSet(ctx)
    nseeds = 0
    if per_read:
        o1, l1 = chunks_by_records(fq, 1)
        for a, b in zip(o1, l1):
            nseeds += u.add_reads(fq[a:a + b], [0], [b], k, amin, xmin, xmax, max_len)
    else:
        nseeds = u.add_reads(fq, offs, lens, k, amin, xmin, xmax, max_len)
    out = str(tmp_path / "dev.fa")
    st = u.write(k, out)
    u.close()
    assert st["truncated"] == 0
    dev_fa = open(out, "rb").read()
    O = cqflibs.oracle()
    mask = (1 << (qb + 8)) - 1
    memo = {}

    def count(km):                 # filter count of a k-mer, from a copy whose traveled bits nobody touches
        v = memo.get(km)
        if v is None:
            fh, rh = O.nthash(km, k)
            v = memo[km] = qc.count(min(fh, rh) & mask)
        return v
    qc, _, _ = oracle_t1(fq, offs, lens, k, qb)
    orc_fa, ost = q.contiger(fq, offs, lens, k, amin, xmin, xmax, rule, True)
    dev = UC.canonical(UC.parse(dev_fa, k), k)
    orc = UC.canonical(UC.parse(orc_fa, k), k, drop_invalid=True)
    assert dev[2] == 0, "the device wrote a link whose target does not overlap"
    one_sided = set(dev[0]) ^ set(orc[0])
    ngroups = 0
    if one_sided:
        assert not per_read, "the sequential schedule must give the same set (%d, %d)" % (len(dev[0]), len(orc[0]))
        assert len(one_sided) <= max(2, len(dev[0]) // 50), (len(dev[0]), len(orc[0]))
        ngroups = UC.explain_one_sided(one_sided, set(dev[0]) & set(orc[0]), k, qb + 8, seed_kmers(fq, k), O.seq_keys)
    both = set(dev[0]) & set(orc[0])
    dl = {l for l in dev[1] if l[0] in both and l[2] in both}
    ol = {l for l in orc[1] if l[0] in both and l[2] in both}
    assert dl == ol, "link sets differ"
    diff = [c for c in both if dev[0][c] != orc[0][c]]
    if per_read:
        assert not diff, "km / KC differ under the sequential schedule: %d of %d" % (len(diff), len(dev[0]))
        assert nseeds == ost["seeds"]
    else:
        sk = seed_kmers(fq, k)
        for c in diff:
            adm = UC.admissible_km(c, k, count, sk)
            assert dev[0][c][0] in adm and orc[0][c][0] in adm, (dev[0][c], orc[0][c], sorted(adm))
    ctx.close()
    q.free()
    qc.free()
    return dict(unitigs=len(dev[0]), links=len(dev[1]), km_differ=len(diff), seeds=nseeds, oracle=ost, one_sided=len(one_sided),
                circles=sum(1 for l in dev[1] if l[1] == b"O"), stale_links=orc[2])
