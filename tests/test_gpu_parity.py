"""GPU parity tests (-m gpu): the HIP path through the C ABI (libshk.so) against the oracle
on the same seeded inputs. Bit-exact: table bytes, 128-byte header, counters."""
import os

import pytest

import cqflibs
import synth
from fastq_util import chunks_by_records, oracle_header, oracle_t1

pytestmark = pytest.mark.gpu


def _ctx(**kw):
    import shk
    return shk.Context(**kw)


@pytest.mark.parametrize("qb,k,nreads,G,L", [(13, 28, 150, 1500, 100), (17, 47, 1500, 20000, 150),
                                             (19, 31, 6000, 60000, 150), (21, 47, 20000, 300000, 150)])
def test_count_matches_oracle(qb, k, nreads, G, L):
    g = synth.make_genome(G, 1)
    fq = synth.make_fastq(g, nreads, L, 0.01, seed=3, n_frac=0.05, short_frac=0.02, lower_frac=0.02)
    offs, lens = chunks_by_records(fq, max(1, nreads // 7))
    ctx = _ctx(qb=qb, k=k, max_batch_bytes=len(fq) + 1024, max_batch_keys=nreads * L)
    st = ctx.count_chunks(fq, offs, lens)
    q, _, _ = oracle_t1(fq, offs, lens, k, qb)
    assert not q.full()
    t = ctx.totals()
    assert (t.nelts, t.ndistinct) == (q.nelts(), q.ndistinct())
    assert st["kmers"] == q.nelts() and st["new_distinct"] == q.ndistinct()
    assert ctx.blocks() == q.blocks()
    assert ctx.header() == oracle_header(q)
    ctx.close()
    q.free()


@pytest.mark.parametrize("qb,k,trigger,nd,endd,ml", [(15, 28, 9000, 3, False, 1 << 20), (15, 47, 8000, 6, True, 1 << 10),
                                                    (17, 31, 30000, 2, True, 1 << 12)])
def test_denoise_schedule_matches_oracle(qb, k, trigger, nd, endd, ml):
    g = synth.make_genome(6000 if qb == 15 else 25000, 7)
    fq = synth.make_fastq(g, 1200 if qb == 15 else 5000, 100, 0.01, seed=21, n_frac=0.03, short_frac=0.03)
    offs, lens = chunks_by_records(fq, 60)
    ctx = _ctx(qb=qb, k=k, trigger=trigger, num_denoise=nd, min_denoise_len=ml, max_batch_bytes=len(fq) + 1024,
               max_batch_keys=1 << 20)
    # two calls: the schedule must not depend on how chunks are batched
    half = len(offs) // 2
    s1 = ctx.count_chunks(fq, offs[:half], lens[:half])
    s2 = ctx.count_chunks(fq, offs[half:], lens[half:])
    removed = s1["removed"] + s2["removed"]
    rounds = s1["denoise_rounds"] + s2["denoise_rounds"]
    if endd:
        removed += ctx.denoise()
        rounds += 1
    q, orounds, oremoved = oracle_t1(fq, offs, lens, k, qb, trigger, nd, endd, ml)
    assert not q.full()
    assert (rounds, removed) == (orounds, oremoved)
    t = ctx.totals()
    assert (t.nelts, t.ndistinct) == (q.nelts(), q.ndistinct())
    assert ctx.blocks() == q.blocks()
    ctx.close()
    q.free()
