"""CPU tests: the plain-C oracle (oracle/cqf_oracle.c) against
 (a) the golden fixtures generated from the real reference code (tests/golden/), and
 (b) the real reference itself (oracle/_ref) when that build is present.
Also pins the two facts the GPU design rests on:
  * the filter's bytes are a function of the key multiset only (canonical layout);
  * one deNoise round keeps count>=2 entries plus the range-end singletons the
    reference's walk skips.
"""
import hashlib
import json
import os
import random

import pytest

import cqflibs
import synth
from cqf_canon import build_blocks, denoise_survivors

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
O = cqflibs.oracle()
needs_ref = pytest.mark.skipif(not cqflibs.have_ref(), reason="oracle/_ref not built (no /root/reference)")


def sha(b):
    return hashlib.sha256(b).hexdigest()


def test_nthash_golden():
    for kat in json.load(open(os.path.join(G, "nthash_kat.json"))):
        s = kat["seq"].encode("latin1")
        k = kat["k"]
        fh, rh = O.nthash(s, k)
        assert (fh, rh) == (kat["fh"], kat["rh"])
        for i, (f, r) in zip(range(k, len(s)), kat["rolls"]):
            fh, rh = O.nthash_roll(s[i - k], s[i], k, fh, rh)
            assert (fh, rh) == (f, r)
            assert O.nthash(s[i - k + 1:i + 1], k) == (f, r)


def test_counter_codec_golden():
    from cqf_canon import encode_counter
    for row in json.load(open(os.path.join(G, "counter_codec.json"))):
        assert O.encode_counter(row["rem"], row["count"]) == row["slots"]
        assert encode_counter(row["rem"], row["count"]) == row["slots"]


def test_insert_scenarios_golden():
    for sc in json.load(open(os.path.join(G, "insert_scenarios.json"))):
        qb = sc["qb"]
        q = O.new(qb)
        cnt = {}
        for (key, c), isnew in zip(sc["ops"], sc["isnew"]):
            assert q.insert(key, c) == isnew
            cnt[key] = cnt.get(key, 0) + c
        b = q.blocks()
        assert sha(b) == sc["blocks_sha256"]
        if sc["blocks_hex"]:
            assert b.hex() == sc["blocks_hex"]
        assert b == build_blocks(qb, qb + 8, cnt)  # canonical in the multiset
        assert [q.count(k) for k in sc["probe"]] == sc["counts"]
        assert [q.find_first_empty_slot(x) for x in range(0, 1 << qb, 13)] == sc["ffe"]
        assert [q.find_first_nonempty_slot(x) for x in range(0, 1 << qb, 13)] == sc["ffn"]
        assert q.dump() == sorted(cnt.items())
        assert q.check_offset()
        assert [list(q.count_set_traveled(k)) for k in sc["probe"][:40]] == sc["trav"]
        assert [list(q.count_set_traveled(k)) for k in sc["probe"][:40]] == sc["trav2"]
        assert sha(q.blocks()) == sc["blocks_trav_sha256"]
        q.free()
        q = O.new(qb)
        for key, c in cnt.items():  # different insertion order and grouping: same bytes
            q.insert(key, c)
        assert q.blocks() == b
        assert q.denoise_round(sc["min_len"]) == sc["removed"]
        assert (q.nelts(), q.ndistinct()) == (sc["nelts_after"], sc["ndistinct_after"])
        assert sha(q.blocks()) == sc["blocks_after_sha256"]
        surv = denoise_survivors(qb, cnt, sc["min_len"])
        assert q.blocks() == build_blocks(qb, qb + 8, surv)
        q.free()


def test_fastq_builds_golden(tmp_path):
    fx = json.load(open(os.path.join(G, "fastq_builds.json")))
    for key, sizes in fx["chunks"].items():
        f, ps, ov = key.split(":")
        assert O.chunk_sizes(os.path.join(G, f), int(ps), int(ov)) == sizes
    for b in fx["builds"]:
        c = b["cfg"]
        q = O.new(c["qb"])
        st = q.build_t1([os.path.join(G, f) for f in c["files"]], c["k"], c["trigger"], c["nd"], c["end"],
                        part_size=c["ps"], overhead=c["ov"], min_len=c["ml"])
        assert not q.full()
        assert st == b["stats"]
        assert (q.nelts(), q.ndistinct()) == (b["nelts"], b["ndistinct"])
        p = str(tmp_path / "o.cqf")
        q.serialize(p)
        data = open(p, "rb").read()
        assert sha(data) == b["sha256"]
        assert data == open(os.path.join(G, b["cqf"]), "rb").read()
        q2 = O.load(p)
        assert q2.blocks() == q.blocks() and q2.nelts() == q.nelts()
        q.free()
        q2.free()


def test_chunk_keys_match_reads_to_kmers():
    """orc_chunk_keys (the key stream the GPU hash kernel is checked against) is the
    same walk as orc_reads_to_kmers"""
    data = open(os.path.join(G, "reads0.fq"), "rb").read()
    for k, qb in ((28, 17), (47, 17)):
        keys = O.chunk_keys(data, k, qb + 8)
        q = O.new(qb)
        q.reads_to_kmers(data, k)
        cnt = {}
        for x in keys:
            cnt[x] = cnt.get(x, 0) + 1
        assert q.dump() == sorted(cnt.items())
        assert q.nelts() == len(keys)
        q.free()


def test_sizing_celegans():
    """src/CQF-deNoise.cpp:96-161 on the README example (README.md:98). boost's Poisson
    CDF is restated, so the round count is pinned only to +-1."""
    import ctypes as C

    class S(C.Structure):
        _fields_ = [("num_true", C.c_uint64), ("num_false", C.c_uint64), ("qb", C.c_uint64), ("hb", C.c_uint64),
                    ("nd", C.c_int), ("trigger", C.c_uint64), ("lb", C.c_int), ("ub", C.c_int)]
    s = S()
    O.L.orc_size_filter.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_double, C.c_double, C.c_int, C.c_double,
                                    C.POINTER(S)]
    O.L.orc_size_filter(47, 119157843, 16506371070, 0.00234, 0.0, -1, 0.0, C.byref(s))
    assert (s.qb, s.hb) == (29, 37)
    assert 6 <= s.nd <= 10
    assert s.trigger == 119157843 + s.num_false // (s.nd + 1)


@needs_ref
def test_oracle_vs_reference_random():
    R = cqflibs.ref()
    rnd = random.Random(6)
    for trial in range(40):
        qb = rnd.choice([6, 8, 9, 11])  # even qb >= 10: the reference reads past its table in deNoise
        load = rnd.choice([0.2, 0.5, 0.85])
        ins = []
        tot = {}
        for _ in range(int((1 << qb) * load)):
            key = (rnd.randrange(1 << qb) << 8) | rnd.choice([0, 1, 0x7f, 0x80, 0x81, 0xff, rnd.randrange(256)])
            c = rnd.choice([1, 1, 1, 2, 3, 130, 16385])
            ins.append((key, c))
            tot[key] = tot.get(key, 0) + c
        try:
            canon = build_blocks(qb, qb + 8, tot)
        except OverflowError:
            continue
        o, r = O.new(qb), R.new(qb)
        for key, c in ins:
            assert o.insert(key, c) == r.insert(key, c)
        assert o.blocks() == r.blocks() == canon
        for x in range(0, (1 << qb) + 50, 7):
            assert o.find_first_empty_slot(x) == r.find_first_empty_slot(x)
            assert o.find_first_nonempty_slot(x) == r.find_first_nonempty_slot(x)
        for key in list(tot)[:60] + [rnd.randrange(1 << (qb + 8)) for _ in range(60)]:
            assert o.count(key) == r.count(key) == tot.get(key, 0)
            assert o.count_is_traveled(key) == r.count_is_traveled(key)
            assert o.count_set_traveled(key) == r.count_set_traveled(key)
            assert o.count_set_traveled(key) == r.count_set_traveled(key)
        assert o.blocks() == r.blocks()
        o.free(), r.free()
        o, r = O.new(qb), R.new(qb)
        for key, c in tot.items():
            o.insert(key, c), r.insert(key, c)
        ml = rnd.choice([1 << 20, 64, 300])
        assert o.denoise_round(ml) == r.denoise_round(ml)
        assert o.blocks() == r.blocks() == build_blocks(qb, qb + 8, denoise_survivors(qb, tot, ml))
        assert (o.nelts(), o.ndistinct()) == (r.nelts(), r.ndistinct())
        o.free(), r.free()


@needs_ref
def test_oracle_vs_reference_saturated_offsets():
    """clusters long enough to saturate the 8-bit block offset at 255"""
    R = cqflibs.ref()
    rnd = random.Random(3)
    done = 0
    for trial in range(12):
        qb = 11
        tot = {}
        base = rnd.randrange(0, 600)
        for _ in range(rnd.choice([400, 600])):
            key = ((base + rnd.randrange(0, 64)) << 8) | rnd.randrange(256)
            tot[key] = tot.get(key, 0) + rnd.choice([1, 1, 2, 3, 200, 20000])
        for _ in range(200):
            key = (rnd.randrange(1 << qb) << 8) | rnd.randrange(256)
            tot[key] = tot.get(key, 0) + rnd.choice([1, 1, 1, 2])
        try:
            canon = build_blocks(qb, qb + 8, tot)
        except OverflowError:
            continue
        assert max(canon[b * 89] for b in range(len(canon) // 89)) == 255
        done += 1
        o, r = O.new(qb), R.new(qb)
        items = list(tot.items())
        rnd.shuffle(items)
        for key, c in items:
            if c <= 3:
                for _ in range(c):
                    assert o.insert(key, 1) == r.insert(key, 1)
            else:
                o.insert(key, c - 1), r.insert(key, c - 1)
                o.insert(key, 1), r.insert(key, 1)
        assert o.blocks() == r.blocks() == canon
        for key, c in items[:150]:
            assert o.count(key) == r.count(key) == c
        assert o.denoise_round() == r.denoise_round()
        assert o.blocks() == r.blocks() == build_blocks(qb, qb + 8, denoise_survivors(qb, tot))
        o.free(), r.free()
    assert done >= 4


@needs_ref
def test_oracle_vs_reference_high_load_with_long_saturated_chains():
    """99 % of the slots in use (beyond the 95 % the bench runs the filter to): clusters of thousands of slots, chains of dozens of
    blocks whose stored offset is saturated at 255 -- the oracle evaluates those by a loop where the reference recurses
    (gqf.c:580-591 <-> :655-704); inserts through reads_to_kmers, then a deNoise round, then lookups"""
    R = cqflibs.ref()
    qb, k = 18, 63
    g = synth.make_genome(60000, 17)
    fq = synth.make_fastq(g, 5650, 110, 0.02, seed=5)      # ~270 k k-mers, most of them distinct
    o, r = O.new(qb), R.new(qb)
    o.reads_to_kmers(fq, k), r.reads_to_kmers(fq, k)
    assert not o.full()
    blocks = o.blocks()
    assert blocks == r.blocks() and (o.nelts(), o.ndistinct()) == (r.nelts(), r.ndistinct())
    from f4_scenarios import slots_needed
    assert slots_needed(o.dump()) > 0.98 * (1 << qb)
    offs = [blocks[b * 89] for b in range(len(blocks) // 89)]
    run, best = 0, 0
    for x in offs:
        run = run + 1 if x == 255 else 0
        best = max(best, run)
    assert best >= 12                                           # chains of saturated blocks
    assert o.check_offset() and r.check_offset()
    rnd = random.Random(1)
    for key, c in rnd.sample(o.dump(), 300):
        assert o.count(key) == r.count(key) == c
    assert o.denoise_round(1 << 12) == r.denoise_round(1 << 12)
    assert o.blocks() == r.blocks()
    o.free(), r.free()


def test_oracle_survives_an_overfull_table():
    """round 2's fuzz crash: more distinct keys than slots -> one cluster over the whole table, every block's offset
    saturated; the reference's block_offset/run_end recursion (one level per saturated block) overran the stack in the
    restatement too. Now a loop; the oracle reports `full` (sticky) and stays inside its allocation."""
    qb, k = 22, 100
    fq = synth.make_fastq(synth.make_genome(600000, 5), 110000, 150, 0.03, seed=7)   # 5.6 M k-mers, nearly all distinct
    o = O.new(qb)
    o.reads_to_kmers(fq, k)
    assert o.full() and o.ndistinct() > 0.98 * (1 << qb)
    o.free()


@needs_ref
def test_reference_slot_access_overruns_its_allocation_without_the_guard():
    """the reference's gqf.c under AddressSanitizer (oracle/Makefile `asan`, oracle/asan_probe.cpp): with the last slots
    of the overflow tail in use its 8-byte slot accesses (gqf.c:542-574) pass the calloc of qf_init. ref_driver.cpp
    therefore gives every reference table guard blocks: with them the same scenario is clean, so test scenarios
    may fill the tail without corrupting the checker's heap."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists("/root/reference/cqf/gqf.c"):
        pytest.skip("needs the reference sources to build the ASan probe")
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "asan"])
    probe = os.path.join(root, "oracle", "_ref", "asan_probe")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")       # qf_destroy leaks the lock array (gqf.c:2306-2317)
    env.pop("LD_PRELOAD", None)
    ok = subprocess.run([probe], env=env, capture_output=True, text=True)
    assert ok.returncode == 0 and "probe ok" in ok.stdout, ok.stderr[-2000:]
    bad = subprocess.run([probe], env=dict(env, REF_NO_GUARD="1"), capture_output=True, text=True)
    assert bad.returncode != 0 and "heap-buffer-overflow" in bad.stderr and "gqf.c" in bad.stderr


def test_contiger_roll_sequence_equals_scratch_hashes():
    """get_unitig_forward never hashes a k-mer from scratch inside its loop: it rolls with the REAL NTPC64, partly with the
    forward/reverse arguments swapped (src/contig_assembly.cpp:3069, 3090, 3104, 3188-3189). Replayed here call for call
    on the compiled base/nthash.hpp (oracle/_ref): every look-up key it forms -- the four successors, the siblings in
    reverse orientation -- and the state it carries to the next step equal the canonical from-scratch hashes. This is
    what lets the oracle (and the kernels' closed forms) hash from scratch / roll their own way."""
    import random
    if not cqflibs.have_ref():
        pytest.skip("needs oracle/_ref")
    R = cqflibs.ref()
    rng = random.Random(3)
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    bases = b"ACGT"

    def rc(s):
        return bytes(comp[c] for c in reversed(s))

    def canon(s, k):
        fh, rh = R.nthash(s, k)
        return min(fh, rh)
    for trial in range(300):
        k = rng.randrange(21, 65)
        cur = bytes(rng.choice(bases) for _ in range(k))
        fh, rh = R.nthash(cur, k)                                   # :3052
        for step in range(4):
            fix = cur[1:]
            # :3067-3070  kmer_hash = NTPC64(current_kmer[0], DNA_bases[x], K, kmer_hash, kmer_RC_hash) on copies
            for x in bases:
                f2, r2 = R.nthash_roll(cur[0], x, k, fh, rh)
                assert min(f2, r2) == canon(fix + bytes([x]), k)
            # :3090  NTPC64(current_kmer[0], 'A', K, current_kmer_hash, current_kmer_RC_hash): the state moves to fix + 'A'
            fh, rh = R.nthash_roll(cur[0], ord("A"), k, fh, rh)
            assert (fh, rh) == R.nthash(fix + b"A", k)
            # :3091-3104  kmer = RC(current) with its last base replaced; NTPC64('T', x, K, kmer_RC_hash, kmer_hash): swapped
            rcur = rc(cur)
            for x in bases:
                if x == rcur[k - 1]:
                    continue
                a, b = R.nthash_roll(ord("T"), x, k, rh, fh)          # (fhVal, rhVal) := (reverse, forward) of the state
                assert min(a, b) == canon(rcur[:k - 1] + bytes([x]), k)
            # :3184-3189  extension by base x: two swapped / unswapped rolls bring the state to the new current k-mer
            x = rng.choice(bases)
            rh, fh = R.nthash_roll(ord("T"), ord("A"), k, rh, fh)     # NTPC64('T', 'A', K, current_kmer_RC_hash, current_kmer_hash)
            fh, rh = R.nthash_roll(ord("T"), x, k, fh, rh)            # NTPC64('T', x, K, current_kmer_hash, current_kmer_RC_hash)
            cur = fix + bytes([x])
            assert (fh, rh) == R.nthash(cur, k)


@needs_ref
def test_oracle_vs_reference_reads_with_bytes_that_are_no_base():
    """reads_to_kmers over reads holding IUPAC codes and other bytes: a byte that is no base hashes as seed 0 on the
    forward strand but as seedTab[byte & 7] on the complement strand (nthash.hpp:15,299: not 0 for Y K S W D ...), only
    an upper-case 'N' restarts the window (CQF_mt.h:672-676). Oracle and compiled reference insert the same keys."""
    R = cqflibs.ref()
    rnd = random.Random(31)
    for k in (21, 47):
        recs = []
        for i in range(60):
            L = rnd.randrange(k, 220)
            s = [rnd.choice("ACGT") for _ in range(L)]
            for _ in range(rnd.choice([1, 2, 5])):
                s[rnd.randrange(L)] = rnd.choice("NnRYKMSWBDHVacgt.-*U")
            recs.append("@r%d\n%s\n+\n%s\n" % (i, "".join(s), "I" * L))
        fq = "".join(recs).encode()
        o, r = O.new(14), R.new(14)
        o.reads_to_kmers(fq, k), r.reads_to_kmers(fq, k)
        assert o.blocks() == r.blocks() and o.nelts() == r.nelts() > 0
        keys = O.chunk_keys(fq, k, 22)
        assert len(keys) == o.nelts()
        o.free(), r.free()
