"""CPU tests of the kernel LOGIC: the unmodified sources of sh-assembly_amd/csrc compiled
with g++ against tests/emu/hip/hip_runtime.h (every GPU thread a fiber, waves of 64 with
barrier-backed shuffles, 512-thread workgroups as on the GPU) and driven through the same C ABI.
These run without a GPU; the real parity tests are tests/test_gpu_parity.py (-m gpu) on libshk.so
built by hipcc. Sizes are small: workgroups run one after another on one core."""
import ctypes as C
import os
import subprocess
import sys

import pytest

import cqflibs
import synth
from fastq_util import chunks_by_records, oracle_header, oracle_t1

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu", "libshk_emu.so")


@pytest.fixture(scope="module")
def shk():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    import shk as m
    return m


def _ctx(shk, **kw):
    return shk.Context(threads_per_group=64, hash_groups=2, lib_path=EMU, **kw)


def test_key_stream_is_reference_order(shk):
    """k_hash_reads emits exactly the keys reads_to_kmers would insert, in its order,
    including the 'N' restart rule, short reads and lower case"""
    O = cqflibs.oracle()
    fq = synth.make_fastq(synth.make_genome(3000, 1), 36, 100, 0.01, seed=3, n_frac=0.25, short_frac=0.1,
                          lower_frac=0.05)
    offs, lens = chunks_by_records(fq, 15)
    for k, qb in ((28, 12), (100, 12)):
        ctx = _ctx(shk, qb=qb, k=k, max_batch_bytes=1 << 20, max_batch_keys=1 << 16)
        dp, nw = ctx.hash_chunks(fq, offs, lens)
        words = (C.c_uint64 * max(nw, 1)).from_address(dp)
        hb = qb + 8
        exp, chunk = [], []
        for ci, (a, n) in enumerate(zip(offs, lens)):
            ks = O.chunk_keys(fq[a:a + n], k, hb)
            exp += ks
            chunk += [ci] * len(ks)
        assert nw == len(exp)
        assert [w & ((1 << hb) - 1) for w in words[:nw]] == exp
        assert [w >> hb for w in words[:nw]] == chunk
        ctx.close()


@pytest.mark.parametrize("stride", ["0", "2"])
def test_count_and_denoise_schedule(shk, monkeypatch, stride):
    """hash -> partition (several levels forced) -> merge -> deNoise rounds where the t = 1 schedule
    fires them; table bytes, header and counters equal the oracle's. stride 2: the chunk of a deNoise point is first
    guessed from a statistics pass over every second region and the one-pass point verifies the guess (sample_locate)"""
    monkeypatch.setenv("SHK_SAMPLE_STRIDE", stride)
    fq = synth.make_fastq(synth.make_genome(300, 7), 32, 90, 0.01, seed=21, n_frac=0.05, short_frac=0.03)
    offs, lens = chunks_by_records(fq, 4)
    qb, k, trig, nd, ml = 10, 28, 350, 3, 64
    ctx = _ctx(shk, qb=qb, k=k, trigger=trig, num_denoise=nd, min_denoise_len=ml, max_batch_bytes=1 << 20,
               max_batch_keys=1 << 16, max_level_bits=2)
    half = len(offs) // 2
    s1 = ctx.count_chunks(fq, offs[:half], lens[:half])
    s2 = ctx.count_chunks(fq, offs[half:], lens[half:])
    removed = s1["removed"] + s2["removed"] + ctx.denoise()
    rounds = s1["denoise_rounds"] + s2["denoise_rounds"] + 1
    q, orounds, oremoved = oracle_t1(fq, offs, lens, k, qb, trig, nd, True, ml)
    assert not q.full()
    assert (rounds, removed) == (orounds, oremoved)
    assert orounds >= 2
    t = ctx.totals()
    assert (t.nelts, t.ndistinct) == (q.nelts(), q.ndistinct())
    assert ctx.blocks() == q.blocks()
    assert ctx.header() == oracle_header(q)
    # lookups + traveled marks against the oracle
    keys = [kc[0] for kc in q.dump()[:40]] + [12345, 1 << (qb + 2), (1 << (qb + 8)) - 1]
    cnt, _ = ctx.lookup(keys, mode=2)
    assert cnt == [q.count(x) for x in keys]
    _, t1 = ctx.lookup(keys, mode=1)
    _, t2 = ctx.lookup(keys, mode=1)
    exp1 = [q.count_set_traveled(x)[0] for x in keys]
    exp2 = [q.count_set_traveled(x)[0] for x in keys]
    assert (t1, t2) == (exp1, exp2)
    assert ctx.blocks() == q.blocks()
    ctx.close()
    q.free()


def test_import_then_count(shk, tmp_path):
    """a .cqf written by the oracle is imported, more reads are counted on top"""
    fq = synth.make_fastq(synth.make_genome(500, 5), 40, 80, 0.01, seed=9)
    offs, lens = chunks_by_records(fq, 10)
    qb, k = 11, 31
    q, _, _ = oracle_t1(fq, offs[:2], lens[:2], k, qb)
    p = str(tmp_path / "a.cqf")
    q.serialize(p)
    ctx = _ctx(shk, qb=qb, k=k, max_batch_bytes=1 << 20, max_batch_keys=1 << 16)
    ctx.import_cqf(p)
    assert ctx.blocks() == q.blocks()
    ctx.count_chunks(fq, offs[2:], lens[2:])
    for a, n in zip(offs[2:], lens[2:]):
        q.reads_to_kmers(fq[a:a + n], k)
    assert ctx.blocks() == q.blocks()
    p2 = str(tmp_path / "b.cqf")
    p3 = str(tmp_path / "c.cqf")
    ctx.export_cqf(p2)
    q.serialize(p3)
    assert open(p2, "rb").read() == open(p3, "rb").read()
    ctx.close()
    q.free()


def test_shards_stitch_to_single_table(shk):
    """two quotient-range shards (own tables, own tails) stitched by host/stitch.cpp give the
    bytes of the single filter, including a cluster that spills across the shard boundary"""
    import random
    from cqf_canon import build_blocks
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "sh-assembly_amd"),
                           os.path.join(ROOT, "sh-assembly_amd", "libshkhost.so")])
    H = C.CDLL(os.path.join(ROOT, "sh-assembly_amd", "libshkhost.so"))
    H.shkh_stitch.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint64]
    rnd = random.Random(11)
    qb, G = 11, 2
    half = (1 << qb) // G
    tot = {}
    for _ in range(700):
        key = (rnd.randrange(1 << qb) << 8) | rnd.randrange(256)
        tot[key] = tot.get(key, 0) + rnd.choice([1, 1, 2, 130])
    for _ in range(90):   # a clump right below the boundary: its cluster spills into shard 1
        key = ((half - 1 - rnd.randrange(20)) << 8) | rnd.randrange(256)
        tot[key] = tot.get(key, 0) + rnd.choice([1, 2, 3])
    shards = []
    for g in range(G):
        keys = [k for k, c in tot.items() if (k >> 8) // half == g for _ in range(min(c, 200))]
        ctx = _ctx(shk, qb=qb, k=21, max_batch_bytes=64, max_batch_keys=len(keys) + 16, shard_index=g, num_shards=G)
        arr = (C.c_uint64 * len(keys))(*keys)
        ctx.count_words(C.addressof(arr), len(keys), 1)
        shards.append(ctx.blocks())
        ctx.close()
    small = {k: min(c, 200) for k, c in tot.items()}
    want = build_blocks(qb, qb + 8, small)
    out = C.create_string_buffer(len(want))
    ptrs = (C.c_char_p * G)(*shards)
    nb = (C.c_uint64 * G)(*[len(s) // 89 for s in shards])
    assert H.shkh_stitch(ptrs, nb, G, qb, out, len(want)) == 0
    assert out.raw == want
    # the spill really crossed the boundary in the single table
    assert want[(half // 64) * 89] > 0


def test_long_cluster_retries_with_big_image(shk):
    """a cluster longer than the small LDS image (1536 slots) makes the first pass stop with an
    extent flag before anything is committed; the library rebuilds with the big image and the
    result is still the canonical table. Lookups walk the long cluster too."""
    import random
    from cqf_canon import build_blocks
    rnd = random.Random(5)
    qb = 12
    tot = {}
    for _ in range(560):
        key = ((1800 + rnd.randrange(0, 64)) << 8) | rnd.randrange(256)
        tot[key] = tot.get(key, 0) + rnd.choice([2, 3, 200, 20000])
    for _ in range(100):
        key = (rnd.randrange(1 << qb) << 8) | rnd.randrange(256)
        tot[key] = tot.get(key, 0) + 1
    small = {k: min(c, 300) for k, c in tot.items()}
    want = build_blocks(qb, qb + 8, small)
    from cqf_canon import layout_used
    runs = layout_used(qb, small)
    # the clump's cluster really is longer than the small image
    clump_end = max(e for q, s, e in runs if 1800 <= q < 1864)
    assert clump_end - 1800 > 1536
    # most of the table arrives as an imported image (cheap in the emulator); a last batch of keys -- new ones and
    # increments inside the long cluster -- is then counted on top of it
    keys = list(small)
    late = {k: min(small[k], 25) for k in keys[:40] + keys[-40:]}     # 40 inside the clump, 40 elsewhere
    early = {k: small[k] - late.get(k, 0) for k in keys}
    early = {k: c for k, c in early.items() if c > 0}
    ctx = _ctx(shk, qb=qb, k=21, max_batch_bytes=64, max_batch_keys=1 << 14)
    ctx.import_blocks(build_blocks(qb, qb + 8, early), nelts=sum(early.values()), ndistinct=len(early))
    part = [k for k, c in late.items() for _ in range(c)]
    rnd.shuffle(part)
    arr = (C.c_uint64 * len(part))(*part)
    ctx.count_words(C.addressof(arr), len(part), 1)
    assert ctx.L.shk_last_error_bits(ctx.h) == 0      # the extent flag of the first attempt was consumed by the retry
    assert ctx.blocks() == want
    ks = list(small)[:300]
    cnt, _ = ctx.lookup(ks, mode=2)
    assert cnt == [small[k] for k in ks]
    ctx.close()


@pytest.mark.parametrize("qb", [6, 8, 9])
def test_tiny_filters_single_region(shk, qb):
    """filters of one or two regions (no partition digits at all for qb <= 8): the last partition level
    still converts the key words to the rebuild kernel's 32-bit records"""
    fq = synth.make_fastq(synth.make_genome(60, 3), 4, 40, 0.0, seed=2)
    offs, lens = chunks_by_records(fq, 2)
    k = 28
    q, _, _ = oracle_t1(fq, offs, lens, k, qb)
    assert not q.full()
    ctx = _ctx(shk, qb=qb, k=k, max_batch_bytes=1 << 16, max_batch_keys=1 << 12)
    ctx.count_chunks(fq, offs, lens)
    t = ctx.totals()
    assert (t.nelts, t.ndistinct) == (q.nelts(), q.ndistinct())
    assert ctx.blocks() == q.blocks()
    ctx.close()
    q.free()


def _unitig_case(shk_mod, ctx_factory, qb, k, G, nreads, L, err, nseeds):
    """filter from synthetic reads; seeds = middle k-mers of the reads that pass Contiger's seed filter
    (count in [2, 1e6], contig_assembly.cpp:1860-1876); device unitigs vs the oracle's restatement"""
    g = synth.make_genome(G, 11)
    fq = synth.make_fastq(g, nreads, L, err, seed=13)
    offs, lens = chunks_by_records(fq, max(1, nreads // 3))
    q, _, _ = oracle_t1(fq, offs, lens, k, qb)
    assert not q.full()
    ctx = ctx_factory(qb=qb, k=k, max_batch_bytes=len(fq) + 1024, max_batch_keys=nreads * L)
    ctx.count_chunks(fq, offs, lens)
    assert ctx.blocks() == q.blocks()
    O = cqflibs.oracle()
    hb = qb + 8
    seeds, counts = [], []
    for line in fq.split(b"\n")[1::4]:
        if len(line) < k:
            continue
        mid = len(line) // 2 - k // 2
        km = line[mid:mid + k].upper()
        if b"N" in km or km in seeds:
            continue
        fh, rh = O.nthash(km, k)
        c = q.count(min(fh, rh) & ((1 << hb) - 1))
        if 2 <= c <= 1000000:
            seeds.append(km)
            counts.append(c)
        if len(seeds) >= nseeds:
            break
    assert len(seeds) >= 3
    max_len = 4 * G + k
    got = ctx.unitigs_from_seeds(seeds, counts, k, 2, max_len)
    stops = set()
    for s, c, (seq, med, st) in zip(seeds, counts, got):
        eseq, emed, est = q.unitig_from_seed(s, c, k, 2, max_len)
        assert (seq, med, st) == (eseq, emed, est)
        stops.update(st)
        assert len(seq) >= k
    ctx.close()
    q.free()
    return stops


def _rc(s):
    return s[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))


def _oracle_unitig_set(q, seeds, counts, k, amin, max_len):
    """independent restatement of the closure (seeds both ways, branch neighbours forward, each unitig once) on the
    oracle's get_unitig_forward; returns {canonical sequence: median}"""
    ends, queued, units = set(), set(), {}
    work = []
    for s, c in zip(seeds, counts):
        if s not in queued:
            queued.add(s)
            work.append((s, c, 2))
    while work:
        nxt, done = [], []
        for seq, med, passes in work:
            for p in range(passes):
                if p == 1:
                    seq = _rc(seq)
                seq, med, st, br, nc = q.extend_forward(seq, med, k, amin, max_len)
                if st == 1:
                    last = seq[-k:]
                    for x in range(4):
                        if br & (1 << x):
                            nxt.append((last[1:] + b"ACGT"[x:x + 1], nc[x]))
                    for z in range(4):
                        if br & (16 << z):
                            nxt.append((_rc(b"ACGT"[z:z + 1] + last[1:]), nc[4 + z]))
            done.append((seq, med))
        for seq, med in done:
            f, e = seq[:k], _rc(seq[-k:])
            if f in ends or e in ends:
                continue
            ends.update((f, e))
            units[min(seq, _rc(seq))] = med
        work = []
        for s, c in nxt:
            if s not in ends and s not in queued:
                queued.add(s)
                work.append((s, c, 1))
    return units


def _read_unitigs(path, k):
    """{canonical sequence: median}; checks the record grammar (LN, KC, km) and every L: link against the sequences:
    `L:+:j:+` <=> unitig j starts with my last k-1 bases + x, `L:+:j:-` <=> RC(j) does; `L:-:...` the same for RC(me)
    (build_graph_worker, contig_assembly.cpp:1012-1084), and no such neighbour is left unlinked"""
    out, recs = {}, []
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    for h, s in zip(lines[0::2], lines[1::2]):
        if not h:
            continue
        parts = h.split()
        assert parts[0] == b">%d" % len(recs)
        fields = dict(x.split(b":", 2)[0::2] for x in parts[1:4])
        assert int(fields[b"LN"]) == len(s)
        med = int(fields[b"km"])
        assert int(fields[b"KC"]) == med * (len(s) - k + 1)
        links = [tuple(x.split(b":")[1:]) for x in parts[4:]]
        recs.append((s, links))
        out[min(s, _rc(s))] = med
    starts = {}
    for j, (s, _) in enumerate(recs):
        f, e = s[:k], _rc(s[-k:])
        if f != e:
            starts[e] = (j, b"-")
        starts[f] = (j, b"+")
    for s, links in recs:
        want = []
        for side, fix, order in ((b"+", s[-(k - 1):], b"ACGT"), (b"-", _rc(s[:k - 1]), b"TGCA")):
            for x in order:
                hit = starts.get(fix + bytes([x]))
                if hit:
                    want.append((side, b"%d" % hit[0], hit[1]))
        assert links == want
    return out


def _find_unitigs_case(ctx_factory, tmp_path, qb, k, G, nreads, L, err, seed_every=1, repeat=0):
    g = synth.make_genome(G, 17)
    if repeat:      # a segment longer than k occurring twice: real branches in the solid graph
        import numpy as np
        g = np.concatenate([g[:3 * G // 5], g[G // 5:G // 5 + repeat], g[3 * G // 5:]])
    fq = synth.make_fastq(g, nreads, L, err, seed=19)
    offs, lens = chunks_by_records(fq, max(1, nreads // 3))
    q, _, _ = oracle_t1(fq, offs, lens, k, qb)
    assert not q.full()
    ctx = ctx_factory(qb=qb, k=k, max_batch_bytes=len(fq) + 1024, max_batch_keys=nreads * L)
    ctx.count_chunks(fq, offs, lens)
    O = cqflibs.oracle()
    hb = qb + 8
    seeds, counts = [], []
    for line in fq.split(b"\n")[1::4][::seed_every]:
        mid = len(line) // 2 - k // 2
        km = line[mid:mid + k]
        if len(km) < k or b"N" in km or km in seeds:
            continue
        fh, rh = O.nthash(km, k)
        c = q.count(min(fh, rh) & ((1 << hb) - 1))
        if 2 <= c <= 1000000:
            seeds.append(km)
            counts.append(c)
    max_len = 2 * len(g) + k
    # per-seed parity (sequence, median abundance, both stop reasons) ...
    for sd, c, (seq, med, stp) in zip(seeds, counts, ctx.unitigs_from_seeds(seeds, counts, k, 2, max_len)):
        assert (seq, med, stp) == q.unitig_from_seed(sd, c, k, 2, max_len)
    # ... and the closure
    path = str(tmp_path / "unitigs.fa")
    st = ctx.find_unitigs(seeds, counts, k, 2, max_len, path)
    got = _read_unitigs(path, k)
    exp = _oracle_unitig_set(q, seeds, counts, k, 2, max_len)
    assert got == exp
    assert st["unitigs"] == len(got) and st["total_len"] == sum(len(s) for s in got) and st["truncated"] == 0
    ctx.close()
    q.free()
    return g, got, st


def test_find_unitigs_matches_oracle_closure(shk, tmp_path, monkeypatch):
    monkeypatch.setenv("SHK_WALK_STEP", "17")       # walks continue over several launches
    g, got, st = _find_unitigs_case(lambda **kw: _ctx(shk, **kw), tmp_path, qb=12, k=21, G=240, nreads=90, L=60, err=0.004, repeat=32, seed_every=45)
    assert st["rounds"] >= 2 and len(got) >= 3      # branches were followed


def test_find_unitigs_rebuilds_an_error_free_genome(shk, tmp_path):
    """no sequencing errors, a random genome without repeated (k-1)-mers: the solid graph is one path, and the one
    unitig is the covered part of the genome"""
    g, got, st = _find_unitigs_case(lambda **kw: _ctx(shk, **kw), tmp_path, qb=12, k=21, G=200, nreads=80, L=60, err=0.0)
    gs = g.tobytes()
    for s in got:
        assert s in gs or _rc(s) in gs
    assert max(len(s) for s in got) >= 150



def test_counted_insert_dump_merge_shards(shk):
    """f-4 / a-9 on the emulator build: counted inserts, iterator dump, qf_merge / qf_multi_merge and the shard stitch
    against the compiled reference (tests/f4_scenarios.py); tiny tables incl. dense clusters and saturated offsets"""
    import random
    import f4_scenarios as F

    def mk(qb):
        return _ctx(shk, qb=qb, k=21, max_batch_keys=1 << 14)

    def mk_shard(qb, s, n):
        return _ctx(shk, qb=qb, k=21, max_batch_keys=1 << 14, shard_index=s, num_shards=n)
    rng = random.Random(11)
    # (a few small cases; tests/test_gpu_parity.py runs the full set)
    F.check_counted_and_dump(mk, 10, F.pairs(rng, 10, 120, 1 << 14, cluster=(700, 90)))      # a long cluster: offsets saturate
    F.check_counted_and_dump(mk, 10, F.pairs(rng, 10, 60, 300, cluster=(1000, 24)), batches=2)  # runs spill into the tail
    a = F.pairs(rng, 10, 100, 400, cluster=(300, 60))
    b = [(k, c + 1) for k, c in a[:50]] + F.pairs(rng, 10, 60, 400, cluster=(310, 60))   # shared keys: counts add
    b = list({k: c for k, c in b}.items())
    F.check_merge(mk, 10, [a, b, F.pairs(rng, 10, 90, 1 << 12)])
    F.check_shards(mk, mk_shard, 10, F.pairs(rng, 10, 140, 300, cluster=(480, 64)), 2)   # a cluster across the shard border


@pytest.mark.parametrize("mark", [0, 1])
def test_unitig_set_invariants(shk, tmp_path, mark):
    """the device-resident unitig engine (k_ug_walk and the finishing kernels) on the emulator: a genome with a repeat and a
    small circle, seeds fed in two calls; the output is the compacted graph of the solid k-mers (tests/unitig_invariants.py)
    with a well-formed record and link grammar; mark = 1 runs the traveled-bit / known-node protocol"""
    import numpy as np
    import unitig_invariants as UI
    k, qb = 21, 12
    g = synth.make_genome(220, 17)
    g = np.concatenate([g[:130], g[40:72], g[130:]])
    plasmid = synth.make_genome(70, 19)
    fq = synth.make_fastq(g, 70, 60, 0.004, seed=19) + synth.make_fastq(np.concatenate([plasmid, plasmid, plasmid[:40]]), 24, 50, 0.0, seed=23, name_prefix="p")
    offs, lens = chunks_by_records(fq, 40)
    q, _, _ = oracle_t1(fq, offs, lens, k, qb)
    ctx = _ctx(shk, qb=qb, k=k, max_batch_bytes=len(fq) + 1024, max_batch_keys=1 << 14)
    ctx.count_chunks(fq, offs, lens)
    O = cqflibs.oracle()

    def count(km):
        fh, rh = O.nthash(km, k)
        return q.count(min(fh, rh) & ((1 << (qb + 8)) - 1))
    out = str(tmp_path / "u.fa")
    u = shk.UnitigSet(ctx)
    if mark:
        n = u.add_reads(fq, offs, lens, k, 2, 2, 1000000, 4000)      # seeds chosen on the device, batch of all chunks
        assert n >= 3
        seeds = None
    else:
        seeds, counts = [], []
        for line in fq.split(b"\n")[1::4]:
            km = line[len(line) // 2 - k // 2:][:k]
            if len(km) == k and b"N" not in km and count(km) >= 2:
                seeds.append(km)
                counts.append(count(km))
        half = len(seeds) // 2
        u.add_seeds(seeds[:half], counts[:half], k, 2, 4000, mark_traveled=False)
        u.add_seeds(seeds[half:], counts[half:], k, 2, 4000, mark_traveled=False)
    st = u.write(k, out)
    u.close()
    got = _read_unitigs(out, k)
    seqs = [ln for ln in open(out, "rb").read().split(b"\n")[1::2] if ln]
    assert st["unitigs"] == len(seqs) == len(got) and st["truncated"] == 0
    if seeds is None:       # what the device took as seeds: every read's middle k-mer that is solid is in some unitig
        seeds = [ln[len(ln) // 2 - k // 2:][:k] for ln in fq.split(b"\n")[1::4]]
        seeds = [s for s in seeds if len(s) == k and b"N" not in s and count(s) >= 2]
    UI.check(seqs, UI.Graph(count, k, 2), seeds=seeds)
    assert any(len(x) == 70 + k - 1 and x[-(k - 1):] == x[:k - 1] for x in seqs)      # the plasmid came out as one pure circle
    ctx.close()
    q.free()


@pytest.mark.parametrize("per_read,rule", [(True, 1), (False, 1)])
def test_contiger_whole_pipeline_against_the_sequential_restatement(shk, tmp_path, per_read, rule):
    """seeds from reads -> walks -> queued contigs -> duplicate removal -> renumbering -> links -> unitigs.fa on the
    emulator build of the device code == the sequential restatement of the whole of Contiger
    (oracle/contiger_pipeline.cpp): canonical sequences, km, KC and the canonical link set under the read-by-read
    schedule; sequences, links and admissible km when all chunks are one batch (tests/contiger_cases.py)"""
    import contiger_cases as CC
    fq = CC.reads(G=420, nreads=150, L=60, err=0.006, plasmid=70, seed=65)
    r = CC.run_case(lambda **kw: _ctx(shk, **kw), shk.UnitigSet, tmp_path, k=21, qb=13, fq=fq, chunk_reads=50,
                    per_read=per_read, rule=rule, max_len=4000)
    assert r["unitigs"] >= 8 and r["links"] >= 6 and r["circles"] >= 1, r


def test_contiger_randomised_against_the_sequential_restatement(shk):
    """tools/fuzz_contiger.py on the emulator build: random small genomes (repeats, plasmids, errors), k 21 .. 64,
    thresholds incl. x < s (sequential schedule only) and x > s, both schedules"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_contiger.py"), "--emu", "--cases", "16", "--seed", "2"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "16 cases, 0 mismatches" in r.stdout, r.stdout[-2000:] + r.stderr[-500:]


@pytest.mark.parametrize("flow", ["single", "sharded"])
def test_randomised_configurations_on_the_emulator(shk, flow):
    """tools/fuzz_gpu.py on the CPU build of the kernels: random filter sizes, k, read mixes, chunking, batching, deNoise
    trigger / rounds / range length, sampled-guess strides and rebuild schemes; table bytes, header, counters, rounds and
    removed counts equal the oracle's t = 1 build in every case (the GPU suite runs hundreds of these per round)"""
    cmd = [sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py"), "--emu", "--max-qb", "13", "--cases", "9", "--seed", "7"]
    if flow == "sharded":
        cmd += ["--sharded"]
    env = dict(os.environ, MASTER_PORT=str(29800 + os.getpid() % 150))
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-500:]
    assert "fuzz: 9 cases (" in r.stdout and ", 0 mismatches" in r.stdout, r.stdout[-500:]


def test_prepared_batches_give_the_same_filter(shk):
    """shk_prepare_chunks / shk_count_prepared (on the emulator the front end runs on the calling thread; the GPU suite
    runs it on its helper thread and second stream): shadow buffers, two alternating slots, two batches prepared ahead,
    deNoise rounds inside -- same table, header, rounds and removed counts as the oracle's t = 1 build"""
    fq = synth.make_fastq(synth.make_genome(800, 1), 90, 100, 0.01, seed=3, n_frac=0.25, short_frac=0.1, lower_frac=0.05)
    offs, lens = chunks_by_records(fq, 10)
    k, qb, trig, nd, ml = 47, 12, 2500, 3, 64
    ctx = _ctx(shk, qb=qb, k=k, trigger=trig, num_denoise=nd, min_denoise_len=ml, max_batch_bytes=1 << 20, max_batch_keys=1 << 16,
               max_level_bits=2)
    third = len(offs) // 3
    parts = [(0, third), (third, 2 * third), (2 * third, len(offs))]
    rounds = removed = 0
    ctx.prepare_chunks(fq, offs[:third], lens[:third])
    for i in range(3):
        if i + 1 < 3:
            a, b = parts[i + 1]
            ctx.prepare_chunks(fq, offs[a:b], lens[a:b])
        st = ctx.count_prepared()
        rounds += st["denoise_rounds"]
        removed += st["removed"]
    q, orounds, oremoved = oracle_t1(fq, offs, lens, k, qb, trig, nd, False, ml)
    assert orounds >= 1 and (rounds, removed) == (orounds, oremoved)
    assert ctx.blocks() == q.blocks() and ctx.header() == oracle_header(q)
    ctx.close()
    q.free()


@pytest.mark.parametrize("pack", ["1", "0"])
def test_roll_kernels_key_multiset_over_k_and_read_shapes(shk, monkeypatch, pack):
    """roll_cases.run on the emulator build; pack = 0: every read on the text path (SHK_NO_PACK)"""
    import roll_cases
    if pack == "0":
        monkeypatch.setenv("SHK_NO_PACK", "1")

    def mk(**kw):
        ctx = _ctx(shk, **kw)
        ctx.read_words = lambda dp, n: list((C.c_uint64 * max(n, 1)).from_address(dp)[:n])
        return ctx
    roll_cases.run(mk, pack == "1")


def test_last_partition_level_with_region_slots_and_its_exact_fallback(shk):
    """roll_cases.run_slots on the emulator build"""
    import roll_cases
    roll_cases.run_slots(lambda **kw: _ctx(shk, **kw))


def test_one_pass_denoise_point_with_a_crowded_region(shk, monkeypatch):
    """roll_cases.run_fused_point_with_a_crowded_region on the emulator build (sampled guess from every 2nd region)"""
    import roll_cases
    monkeypatch.setenv("SHK_SAMPLE_STRIDE", "2")
    roll_cases.run_fused_point_with_a_crowded_region(lambda **kw: _ctx(shk, **kw))
