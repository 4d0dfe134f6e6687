"""Scenarios for the filter-to-filter entry points (shk_insert_counted, shk_dump, shk_merge, shk_multi_merge,
shk_import_shards), shared by the CPU emulator tests and the GPU parity tests. The checker is the REAL reference
(oracle/_ref: compiled gqf.c -- qf_insert_advance(count), the qfi_* iterator, qf_merge, qf_multi_merge) when it is
built, the C restatement otherwise (merge then = inserting both dumps)."""
import random

import cqflibs

REMS = [0, 1, 0x7F, 0x80, 0x81, 0xFF]


def pairs(rng, qb, n, max_count, cluster=None, special=0.3):
    """n distinct keys < 2^(qb+8) with counts; `cluster` = (first quotient, width) packs them densely;
    a share of the remainders and counts sit on the codec's edges"""
    seen, out = set(), []
    while len(out) < n:
        if cluster:
            q = cluster[0] + rng.randrange(cluster[1])
        else:
            q = rng.randrange(1 << qb)
        r = rng.choice(REMS) if rng.random() < special else rng.randrange(256)
        k = (q << 8) | r
        if k in seen:
            continue
        seen.add(k)
        e = rng.random()
        if e < 0.35:
            c = 1
        elif e < 0.7:
            c = rng.randrange(2, 300)
        elif e < 0.9:
            c = rng.choice([r, r + 1, 127, 128, 129, 130] + ([16384, 16385, 16386] if max_count > 16386 else [])) or 1
        else:
            c = rng.randrange(1, max_count)
        out.append((k, c))
    return out


def slots_needed(kc):
    """slots the entries take (encode_counter, gqf.c:1225-1255)"""
    tot = 0
    for k, c in kc:
        r, n = k & 0xFF, 1
        if c > 1:
            d, nd = c - 1, 1
            while d >> (7 * nd):
                nd += 1
            top = (d >> (7 * (nd - 1))) & 0x7F
            if nd > 1:
                top |= 0x80
            n += nd + (1 if top > r else 0)
        tot += n
    return tot


def fits(qb, kc):
    """keep the scenarios inside the table: the reference has no capacity check (it runs off the end of a full table).
    Its 8-byte slot accesses behind the last slots of the overflow tail are covered by the guard blocks the driver gives
    every reference table (oracle/ref_driver.cpp: guard_table; tests/test_oracle.py shows the overrun under ASan)."""
    merged = {}
    for k, c in kc:
        merged[k] = merged.get(k, 0) + c
    assert slots_needed(merged.items()) <= 0.95 * (1 << qb), "scenario too full"


def checker():
    return cqflibs.ref() if cqflibs.have_ref() else cqflibs.oracle()


def build(lib, qb, kc):
    q = lib.new(qb)
    for k, c in kc:
        q.insert(k, c)
    return q


def check_counted_and_dump(mk_ctx, qb, kc, batches=3):
    """counted inserts (in `batches` calls, keys repeated across calls add up) == the checker's insert(count);
    dump == the checker's iterator dump"""
    lib = checker()
    fits(qb, kc + [(k, 200) for k, _ in kc[::3]])
    ctx = mk_ctx(qb)
    q = lib.new(qb)
    rng = random.Random(len(kc))
    tot_new = tot_add = 0
    for b in range(batches):
        part = kc[b::batches]
        if b:   # some keys again: counts add up, no new distinct
            part = part + [(k, rng.randrange(1, 200)) for k, _ in kc[0::batches][:len(part) // 3]]
        new = sum(q.insert(k, c) for k, c in part)
        st = ctx.insert_counted([k for k, _ in part], [c for _, c in part])
        assert st["new_distinct"] == new and st["kmers"] == sum(c for _, c in part)
        tot_new += new
        tot_add += st["kmers"]
    assert ctx.blocks() == q.blocks()
    t = ctx.totals()
    assert (t.nelts, t.ndistinct) == (tot_add, tot_new) == (q.nelts(), q.ndistinct())
    d = ctx.dump()
    assert d == sorted(d) and len(d) == tot_new     # every entry, ascending key = iterator order
    # the reference's iterator ends early when it steps inside a run onto a slot behind nslots (gqf.c:2537-2539)
    assert ctx.dump(ref_iterator_end=True) == q.dump()
    if lib.p == "ref_":
        o = build(cqflibs.oracle(), qb, q.dump())
        rest = [kc_ for kc_ in d if kc_ not in set(q.dump())]
        for k, c in rest:
            o.insert(k, c)
        assert o.dump() == d       # the C restatement's dump has no such end: it lists what the table holds
        o.free()
    ctx.close()
    q.free()


def check_merge(mk_ctx, qb, kcs):
    """shk_merge / shk_multi_merge of len(kcs) filters == the reference's qf_merge / qf_multi_merge"""
    lib = checker()
    fits(qb, [x for kc in kcs for x in kc])
    ctxs, qs = [], []
    for kc in kcs:
        c = mk_ctx(qb)
        c.insert_counted([k for k, _ in kc], [x for _, x in kc])
        ctxs.append(c)
        qs.append(build(lib, qb, kc))
    # pairwise: c := a + b (qf_merge sees what the reference's iterator sees: inputs whose iteration ends early
    # are covered by test_merge_of_tail_entries)
    want = lib.new(qb)
    complete = all(len(q.dump()) == len(kc) for q, kc in zip(qs, kcs))
    if lib.p == "ref_" and complete:
        want.merge_from(qs[0], qs[1])
    else:
        for k, c in kcs[0] + kcs[1]:
            want.insert(k, c)
    dst = mk_ctx(qb)
    dst.merge(ctxs[0])
    st = dst.merge(ctxs[1])
    assert dst.blocks() == want.blocks()
    assert dst.dump() == want.dump()
    k0 = {k for k, _ in kcs[0]}
    assert st["new_distinct"] == len({k for k, _ in kcs[1]} - k0)
    assert st["kmers"] == sum(c for _, c in kcs[1])
    t = dst.totals()
    assert t.ndistinct == len(want.dump()) and t.nelts == sum(c for _, c in want.dump())
    want.free()
    if len(kcs) > 2:
        want = lib.new(qb)
        if lib.p == "ref_" and complete:
            want.multi_merge_from(qs)
        else:
            for kc in kcs:
                for k, c in kc:
                    want.insert(k, c)
        dst2 = mk_ctx(qb)
        dst2.multi_merge(ctxs)
        assert dst2.blocks() == want.blocks()
        dst2.close()
        want.free()
    # merging an empty filter in changes nothing
    before = dst.blocks()
    empty = mk_ctx(qb)
    st = dst.merge(empty)
    assert dst.blocks() == before and st["kmers"] == 0 and st["new_distinct"] == 0
    for c in ctxs + [dst, empty]:
        c.close()
    for q in qs:
        q.free()


def check_shards(mk_ctx, mk_shard, qb, kc, nshards):
    """per-shard tables (each with its own overflow tail) -> shk_import_shards == the single table"""
    lib = checker()
    fits(qb, kc)
    q = build(lib, qb, kc)
    per = (1 << qb) // nshards
    blocks = []
    for s in range(nshards):
        c = mk_shard(qb, s, nshards)
        mine = [(k, x) for k, x in kc if (k >> 8) // per == s]
        c.insert_counted([k for k, _ in mine], [x for _, x in mine])
        blocks.append(c.blocks())
        c.close()
    full = mk_ctx(qb)
    full.import_shards(blocks)
    assert full.blocks() == q.blocks()
    t = full.totals()
    assert (t.nelts, t.ndistinct) == (q.nelts(), q.ndistinct())
    assert full.dump(ref_iterator_end=True) == q.dump()
    full.close()
    q.free()
