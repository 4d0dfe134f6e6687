#!/usr/bin/env python3
"""debug: device Contiger output vs the sequential restatement on the fuzz cases (GPU)"""
import os, sys, pathlib, tempfile, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sh-assembly_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import shk, contiger_cases as CC, unitig_compare as UC, cqflibs
from fastq_util import chunks_by_records, oracle_t1
tmp = pathlib.Path(tempfile.mkdtemp())
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 9)
only = int(sys.argv[2]) if len(sys.argv) > 2 else -1
for i in range(40):
    k = rnd.choice([21, 25, 31, 47, 63, 64])
    L = rnd.choice([2 * k + 9, 3 * k])
    G = rnd.choice([300, 500, 900]) * 12
    nreads = G * rnd.choice([8, 14, 25]) // L
    err = rnd.choice([0.0, 0.004, 0.01, 0.02])
    plasmid = rnd.choice([0, k + 30, 2 * k + 11])
    amin = rnd.choice([2, 2, 3, 1])
    per_read = rnd.random() < 0.15
    xmin = rnd.choice([2, 3, 1]) if per_read else max(amin, rnd.choice([2, 3, 1]))
    qb = 13
    while (1 << qb) < 3 * (G + nreads * L * err * k):
        qb += 1
    fq = CC.reads(G, nreads, L, err, plasmid, seed=rnd.randrange(1 << 20))
    cr = rnd.choice([20, 50]) * 12
    if per_read or (only >= 0 and i != only):
        continue
    offs, lens = chunks_by_records(fq, cr)
    ctx = shk.Context(qb=qb, k=k, max_batch_bytes=len(fq) + 1024, max_batch_keys=max(1 << 14, fq.count(b"\n") * 40))
    ctx.count_chunks(fq, offs, lens)
    u = shk.UnitigSet(ctx)
    u.add_reads(fq, offs, lens, k, amin, xmin, 1000000, 8 * G + 1000)
    out = str(tmp / "d.fa")
    st = u.write(k, out)
    u.close(); ctx.close()
    dev = UC.canonical(UC.parse(open(out, "rb").read(), k), k)
    q, _, _ = oracle_t1(fq, offs, lens, k, qb)
    fa, ost = q.contiger(fq, offs, lens, k, amin, xmin, 1000000, 1, True)
    orc = UC.canonical(UC.parse(fa, k), k, drop_invalid=True)
    a, b = set(dev[0]), set(orc[0])
    if a == b:
        continue
    print("case", i, dict(k=k, G=G, plasmid=plasmid, amin=amin, xmin=xmin, err=err), st, flush=True)
    for x in sorted(a - b, key=len):
        host = [len(y) for y in b - a if x in y + y or UC.rc(x) in y + y]
        print("   only dev", len(x), dev[0][x], "inside only-orc of len", host, "links", [(l[1], len(l[2]), l[3]) for l in dev[1] if l[0] == x])
    for x in sorted(b - a, key=len):
        print("   only orc", len(x), orc[0][x], "links", [(l[1], len(l[2]), l[3]) for l in orc[1] if l[0] == x])
