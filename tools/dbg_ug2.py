import sys, os, subprocess, struct
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "sh-assembly_amd"))
import numpy as np
import synth, cqflibs
import unitig_invariants as UI
tmp = "gpurun_out/r2c/cli"; os.makedirs(tmp, exist_ok=True)
bind = os.path.join(ROOT, "sh-assembly_amd", "bin")
k, G = 47, 30000
g = synth.make_genome(G, 51)
g = np.concatenate([g[:14000], g[5000:5500], g[14000:]])
open(tmp + "/a.fq", "wb").write(synth.make_fastq(g, 2400, 150, 0.003, seed=53))
open(tmp + "/b.fq", "wb").write(synth.make_fastq(g, 2400, 150, 0.003, seed=55, name_prefix="s"))
open(tmp + "/files.txt", "w").write("a.fq\nb.fq\n")
cqf = tmp + "/k47.cqf"
r = subprocess.run([bind + "/CQF-deNoise", "-k", str(k), "-N", "500000", "-n", "30000", "-e", "0.003", "-f", "f", "-i", tmp + "/files.txt", "-o", cqf, "--part-size", "100000", "--overhead", "4000"], capture_output=True, text=True)
print(r.stderr[-400:])
out = tmp + "/unitigs.fa"
r = subprocess.run([bind + "/Contiger", "-k", str(k), "-i", tmp + "/files.txt", "-c", cqf, "-o", out, "--part-size", "100000", "--overhead", "4000", "--batch-chunks", "3"], capture_output=True, text=True)
print(r.stderr[-300:])
q = cqflibs.oracle().load(cqf)
hdr = open(cqf, "rb").read(128)
nslots = struct.unpack_from("<Q", hdr, 16)[0]; qb = nslots.bit_length() - 1
O = cqflibs.oracle(); mask = (1 << (qb + 8)) - 1
def key(km):
    fh, rh = O.nthash(km, k); return min(fh, rh) & mask
def count(km): return q.count(key(km))
seqs = [ln for ln in open(out, "rb").read().split(b"\n")[1::2] if ln]
fq = open(tmp + "/a.fq", "rb").read() + open(tmp + "/b.fq", "rb").read()
seeds = []
for line in fq.split(b"\n")[1::4]:
    km = line[len(line) // 2 - k // 2:][:k]
    if len(km) == k and b"N" not in km and 2 <= count(km) <= 1000000: seeds.append(km)
gr = UI.Graph(count, k, 2)
want = gr.reachable(seeds)
seen = set()
for s in seqs:
    for i in range(len(s) - k + 1): seen.add(UI.canon(s[i:i+k]))
keys = {}
for c in want: keys.setdefault(key(c), []).append(c)
print("qb", qb, len(seen), len(want))
for c in want - seen:
    print("missing", c, count(c), "succ", len(gr.succ(c)), "pred", len(gr.pred(c)), "is seed", c in seeds or UI.rc(c) in seeds, "same key as", [x for x in keys[key(c)] if x != c])
for c in seen - want:
    print("extra", c)
