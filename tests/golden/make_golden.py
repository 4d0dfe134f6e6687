#!/usr/bin/env python3
"""Generates tests/golden/*.json|*.cqf|*.fq from the REAL reference code
(oracle/_ref/libshk_ref.so = /root/reference/cqf/gqf.c + base/nthash.hpp compiled as
they lie, driven by oracle/ref_driver.cpp). Run in the build container only:

    make -C oracle ref && python3 tests/golden/make_golden.py

The fixtures are data (inputs + expected outputs); no reference source is stored.
"""
import hashlib
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import cqflibs  # noqa: E402
import synth  # noqa: E402

R = cqflibs.ref()


def nthash_kats():
    rnd = random.Random(1234)
    out = []
    for k in (28, 31, 47, 63, 64, 65, 100):
        for alphabet in (b"ACGT", b"ACGTN", b"ACGTacgtNn\r"):
            s = bytes(rnd.choice(alphabet) for _ in range(k + 20))
            fh, rh = R.nthash(s, k)
            rolls = []
            f, r = fh, rh
            for i in range(k, len(s)):
                f, r = R.nthash_roll(s[i - k], s[i], k, f, r)
                rolls.append([f, r])
            out.append({"k": k, "seq": s.decode("latin1"), "fh": fh, "rh": rh, "rolls": rolls})
    return out


def counter_table():
    q = R.new(8)
    rows = []
    for rem in (0, 1, 2, 0x7f, 0x80, 0x81, 0xfe, 0xff):
        for cnt in (1, 2, 3, rem, rem + 1, rem + 2, 127, 128, 129, 130, 255, 256, 257, 2 ** 14, 2 ** 14 + 1,
                    2 ** 21, 2 ** 21 + 1, 2 ** 28 + 5, 2 ** 35 + 77):
            if cnt < 1:
                continue
            rows.append({"rem": rem, "count": cnt, "slots": R.encode_counter(rem, cnt, q)})
    q.free()
    return rows


def insert_scenarios():
    """random multisets -> block bytes (sha256) + per-key lookups + one deNoise round"""
    rnd = random.Random(99)
    out = []
    for qb, load, minlen in [(6, 0.5, 1 << 20), (7, 0.8, 64), (8, 0.3, 100), (9, 0.85, 1 << 20),
                             (11, 0.6, 256), (11, 0.9, 1 << 20), (13, 0.7, 1000)]:
        q = R.new(qb)
        ops = []
        n = int((1 << qb) * load)
        for _ in range(n):
            key = (rnd.randrange(1 << qb) << 8) | rnd.choice([0, 1, 0x7f, 0x80, 0x81, 0xff, rnd.randrange(256)])
            c = rnd.choice([1, 1, 1, 1, 2, 3, 129, 16385])
            if rnd.random() < 0.5 and qb > 8:
                c = 1
            ops.append([key, c])
        # keep the table below xnslots (the reference does not detect a full table)
        from cqf_canon import build_blocks
        tot = {}
        for key, c in ops:
            tot[key] = tot.get(key, 0) + c
        while True:
            try:
                build_blocks(qb, qb + 8, tot)
                break
            except OverflowError:
                key, c = ops.pop()
                tot[key] -= c
                if tot[key] == 0:
                    del tot[key]
        isnew = []
        for key, c in ops:
            isnew.append(q.insert(key, c))
        blocks = q.blocks()
        probe = [ops[i][0] for i in range(0, len(ops), 5)] + [rnd.randrange(1 << (qb + 8)) for _ in range(50)]
        counts = [q.count(k) for k in probe]
        ffe = [q.find_first_empty_slot(x) for x in range(0, 1 << qb, 13)]
        ffn = [q.find_first_nonempty_slot(x) for x in range(0, 1 << qb, 13)]
        # traveled marks on a copy of the state: do them after the snapshot, then rebuild
        trav = [list(q.count_set_traveled(k)) for k in probe[:40]]
        trav2 = [list(q.count_set_traveled(k)) for k in probe[:40]]
        blocks_trav = q.blocks()
        q.free()
        q = R.new(qb)
        for key, c in ops:
            q.insert(key, c)
        removed = q.denoise_round(minlen)
        out.append({
            "qb": qb, "ops": ops, "isnew": isnew, "min_len": minlen,
            "blocks_sha256": hashlib.sha256(blocks).hexdigest(),
            "blocks_hex": blocks.hex() if qb <= 8 else None,
            "probe": probe, "counts": counts, "ffe": ffe, "ffn": ffn,
            "trav": trav, "trav2": trav2,
            "blocks_trav_sha256": hashlib.sha256(blocks_trav).hexdigest(),
            "removed": removed, "nelts_after": q.nelts(), "ndistinct_after": q.ndistinct(),
            "blocks_after_sha256": hashlib.sha256(q.blocks()).hexdigest(),
        })
        q.free()
    return out


def fastq_builds():
    """small FASTQ files (committed) -> chunk sizes, t=1 build stats, .cqf bytes"""
    g = synth.make_genome(6000, 7)
    files = []
    specs = [dict(nreads=700, L=100, err=0.01, seed=21, n_frac=0.03, short_frac=0.03, lower_frac=0.02),
             dict(nreads=500, L=80, err=0.02, seed=22, n_frac=0.05, plus_repeats_name=True)]
    for i, sp in enumerate(specs):
        fq = synth.make_fastq(g, **sp)
        p = os.path.join(HERE, f"reads{i}.fq")
        open(p, "wb").write(fq)
        files.append(p)
    # a CRLF variant of a few records (reference keeps the '\r' in the read, CQF_mt.h:617-621)
    crlf = synth.make_fastq(g, 40, 60, 0.0, 23, n_frac=0.0).replace(b"\n", b"\r\n")
    p = os.path.join(HERE, "reads_crlf.fq")
    open(p, "wb").write(crlf)
    out = {"files": [os.path.basename(f) for f in files], "chunks": {}, "builds": []}
    for f in files:
        for ps, ov in [(1 << 23, 65535), (20000, 4095), (9000, 2047)]:
            out["chunks"][f"{os.path.basename(f)}:{ps}:{ov}"] = R.chunk_sizes(f, ps, ov)
    cfgs = [
        dict(k=28, qb=17, trigger=10 ** 9, nd=0, end=False, ps=20000, ov=4095, ml=1 << 20, files=[0, 1]),
        dict(k=28, qb=15, trigger=9000, nd=3, end=False, ps=20000, ov=4095, ml=1 << 20, files=[0, 1]),
        dict(k=47, qb=15, trigger=8000, nd=6, end=True, ps=9000, ov=2047, ml=1 << 10, files=[0, 1]),
        dict(k=31, qb=17, trigger=7000, nd=2, end=True, ps=20000, ov=4095, ml=1 << 12, files=[1, 0]),
        dict(k=21, qb=13, trigger=10 ** 9, nd=0, end=False, ps=1 << 23, ov=65535, ml=1 << 20, files=[2]),
    ]
    allf = files + [p]
    for ci, c in enumerate(cfgs):
        o = cqflibs.oracle().new(c["qb"])  # the restatement detects a full table; the reference corrupts memory
        o.build_t1([allf[i] for i in c["files"]], c["k"], c["trigger"], c["nd"], c["end"],
                   part_size=c["ps"], overhead=c["ov"], min_len=c["ml"])
        assert not o.full(), ("table too small for fixture", c)
        o.free()
        q = R.new(c["qb"])
        st = q.build_t1([allf[i] for i in c["files"]], c["k"], c["trigger"], c["nd"], c["end"],
                        part_size=c["ps"], overhead=c["ov"], min_len=c["ml"])
        name = f"build{ci}.cqf"
        q.serialize(os.path.join(HERE, name))
        c2 = dict(c)
        c2["files"] = [os.path.basename(allf[i]) for i in c["files"]]
        out["builds"].append({"cfg": c2, "stats": st, "nelts": q.nelts(), "ndistinct": q.ndistinct(),
                              "cqf": name,
                              "sha256": hashlib.sha256(open(os.path.join(HERE, name), "rb").read()).hexdigest()})
        q.free()
    return out


def main():
    json.dump(nthash_kats(), open(os.path.join(HERE, "nthash_kat.json"), "w"))
    json.dump(counter_table(), open(os.path.join(HERE, "counter_codec.json"), "w"))
    json.dump(insert_scenarios(), open(os.path.join(HERE, "insert_scenarios.json"), "w"))
    json.dump(fastq_builds(), open(os.path.join(HERE, "fastq_builds.json"), "w"), indent=1)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
