import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "sh-assembly_amd"))
import numpy as np
import torch
import shk, synth, cqflibs
import unitig_invariants as UI
from fastq_util import chunks_by_records, oracle_t1
k, qb, mark = 64, 18, 1
G = 24000
g = synth.make_genome(G, 41)
g = np.concatenate([g[:9000], g[2000:2600], g[9000:], g[15000:15300]])
plasmid = synth.make_genome(700, 43)
circ = np.concatenate([plasmid, plasmid, plasmid, plasmid[:200]])
fq = synth.make_fastq(g, 2600, 150, 0.004, seed=45) + synth.make_fastq(circ, 500, 120, 0.0, seed=47, name_prefix="p")
offs, lens = chunks_by_records(fq, 600)
ctx = shk.Context(qb=qb, k=k, max_batch_bytes=len(fq) + 1024, max_batch_keys=1 << 20)
ctx.count_chunks(fq, offs, lens)
q, _, _ = oracle_t1(fq, offs, lens, k, qb)
O = cqflibs.oracle()
def count(km):
    fh, rh = O.nthash(km, k)
    return q.count(min(fh, rh) & ((1 << (qb + 8)) - 1))
seeds, counts = [], []
for line in fq.split(b"\n")[1::4]:
    mid = len(line) // 2 - k // 2
    km = line[mid:mid + k]
    if len(km) < k or b"N" in km: continue
    c = count(km)
    if 2 <= c <= 1000000:
        seeds.append(km); counts.append(c)
u = shk.UnitigSet(ctx)
third = len(seeds) // 3
for a, b in ((0, third), (third, 2 * third), (2 * third, len(seeds))):
    u.add_seeds(seeds[a:b], counts[a:b], k, 2, 4 * len(g), mark_traveled=bool(mark))
st = u.write(k, "gpurun_out/r2c/u.fa"); print(st)
seqs = [ln for ln in open("gpurun_out/r2c/u.fa", "rb").read().split(b"\n")[1::2] if ln]
gr = UI.Graph(count, k, 2)
want = gr.reachable(seeds)
for si, s in enumerate(seqs):
    for i in range(len(s) - k + 1):
        c = UI.canon(s[i:i+k])
        if c not in want:
            km = s[i:i+k]
            print("extra", si, i, len(s), km, count(km), "succ", [(x, count(x)) for x in gr.succ(km)], "pred", [(x, count(x)) for x in gr.pred(km)], "is_seed", km in seeds or UI.rc(km) in seeds, "palin", km == UI.rc(km))
