"""GPU parity tests (-m gpu): the HIP path through the C ABI (libshk.so) against the oracle
on the same seeded inputs. Bit-exact: table bytes, 128-byte header, counters."""
import ctypes as C
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


def test_cli_reproduces_golden_cqf(tmp_path):
    """sh-assembly_amd/bin/CQF-deNoise (reference flags + test hooks for the small part
    geometry of the fixtures) writes the .cqf files the REFERENCE build wrote for
    tests/golden (fastq_builds.json)"""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    G = os.path.join(root, "tests", "golden")
    exe = os.path.join(root, "sh-assembly_amd", "bin", "CQF-deNoise")
    assert os.path.exists(exe), "build with make -C sh-assembly_amd"
    fx = json.load(open(os.path.join(G, "fastq_builds.json")))
    for b in fx["builds"]:
        c = b["cfg"]
        lst = tmp_path / "files.txt"
        # the list is resolved relative to its own directory (src/CQF-deNoise.cpp:59-81)
        for f in c["files"]:
            dst = tmp_path / f
            if not dst.exists():
                dst.write_bytes(open(os.path.join(G, f), "rb").read())
        lst.write_text("\n".join(c["files"]) + "\n")
        out = str(tmp_path / "out.cqf")
        cmd = [exe, "-k", str(c["k"]), "-n", "6000", "-N", "100000", "-e", "0.01", "-f", "f", "-i", str(lst), "-o", out,
               "--deNoise", str(c["nd"]), "--rounds", str(c["nd"]), "--qb", str(c["qb"]), "--trigger", str(min(c["trigger"], 1 << 62)),
               "--part-size", str(c["ps"]), "--overhead", str(c["ov"]), "--min-denoise-len", str(c["ml"])]
        if c["end"]:
            cmd.append("--endDeNoise")
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert open(out, "rb").read() == open(os.path.join(G, b["cqf"]), "rb").read(), (c, r.stderr[-400:])


def test_lookup_and_traveled_marks():
    """k_lookup against the oracle: counts, is_traveled, set_traveled (twice), table bytes after"""
    import random
    qb, k = 17, 31
    g = synth.make_genome(20000, 3)
    fq = synth.make_fastq(g, 3000, 100, 0.01, seed=5, n_frac=0.02)
    offs, lens = chunks_by_records(fq, 500)
    ctx = _ctx(qb=qb, k=k, max_batch_bytes=len(fq) + 1024, max_batch_keys=3000 * 100)
    ctx.count_chunks(fq, offs, lens)
    q, _, _ = oracle_t1(fq, offs, lens, k, qb)
    rnd = random.Random(1)
    present = [kc[0] for kc in q.dump()]
    keys = rnd.sample(present, 2000) + [rnd.randrange(1 << (qb + 8)) for _ in range(2000)]
    cnt, _ = ctx.lookup(keys, mode=2)
    assert cnt == [q.count(x) for x in keys]
    c0, t0 = ctx.lookup(keys, mode=0)
    assert t0 == [q.count_is_traveled(x)[0] for x in keys]
    sub = keys[:1500:3] + keys[2000:2300]
    _, t1 = ctx.lookup(sub, mode=1)
    exp1 = [q.count_set_traveled(x)[0] for x in sub]
    _, t2 = ctx.lookup(sub, mode=1)
    exp2 = [q.count_set_traveled(x)[0] for x in sub]
    assert (t1, t2) == (exp1, exp2)
    assert ctx.blocks() == q.blocks()
    ctx.close()
    q.free()


def _upload(keys):
    import torch
    return torch.tensor(keys, dtype=torch.int64, device="cuda")


def test_dense_clusters_and_saturated_offsets():
    """shk_count_words on crafted keys: one clump of ~400 entries with large counts inside 64
    quotients (block offsets saturate at 255, runs spill over many blocks), plus background.
    Table bytes equal the canonical layout model and the oracle; lookups agree."""
    import random
    import torch
    from cqf_canon import build_blocks
    rnd = random.Random(7)
    qb = 13
    for trial in range(4):
        tot = {}
        base = rnd.randrange(0, 5000)
        for _ in range(rnd.choice([250, 400])):
            key = ((base + rnd.randrange(0, 64)) << 8) | rnd.randrange(256)
            tot[key] = tot.get(key, 0) + rnd.choice([1, 1, 2, 3, 200, 20000])
        for _ in range(1500):
            key = (rnd.randrange(1 << qb) << 8) | rnd.randrange(256)
            tot[key] = tot.get(key, 0) + rnd.choice([1, 1, 1, 2])
        canon = build_blocks(qb, qb + 8, tot)
        assert max(canon[b * 89] for b in range(len(canon) // 89)) == 255
        # occurrences capped per key so the word list stays small; large counts via two batches of repeats
        words = []
        for key, c in tot.items():
            words += [key] * min(c, 300)
        small = {k: min(c, 300) for k, c in tot.items()}
        rnd.shuffle(words)
        ctx = _ctx(qb=qb, k=21, max_batch_bytes=64, max_batch_keys=len(words) + 16)
        half = len(words) // 2
        for part in (words[:half], words[half:]):
            t = _upload(part)
            torch.cuda.synchronize()
            ctx.count_words(t.data_ptr(), t.numel(), 1)
        assert ctx.blocks() == build_blocks(qb, qb + 8, small)
        ks = list(small)[:500] + [rnd.randrange(1 << (qb + 8)) for _ in range(200)]
        cnt, _ = ctx.lookup(ks, mode=2)
        assert cnt == [small.get(k, 0) for k in ks]
        ctx.close()


def test_full_table_is_an_error_not_corruption():
    """the reference overruns a full table; the library reports SHK_ERR_TABLE_FULL and leaves the table as it was"""
    import random
    import shk
    import torch
    rnd = random.Random(3)
    qb = 10
    ctx = _ctx(qb=qb, k=21, max_batch_bytes=64, max_batch_keys=1 << 14)
    first = [(rnd.randrange(1 << qb) << 8) | rnd.randrange(256) for _ in range(300)]
    t = _upload(first)
    torch.cuda.synchronize()
    ctx.count_words(t.data_ptr(), t.numel(), 1)
    before = ctx.blocks()
    many = [(rnd.randrange(1 << qb) << 8) | rnd.randrange(256) for _ in range(4000)]
    t = _upload(many)
    torch.cuda.synchronize()
    with pytest.raises(shk.ShkError) as e:
        ctx.count_words(t.data_ptr(), t.numel(), 1)
    assert e.value.code in (-3, -4)
    assert ctx.blocks() == before
    ctx.close()


def test_large_scale_properties(tmp_path):
    """size-independent properties at a scale the oracle cannot replay insert by insert
    (qb 26, ~100 M k-mers): the table does not depend on how the reads are cut into chunks
    and calls; header counters equal what a decode of the exported .cqf finds; a second deNoise
    round right after the first removes nothing; export -> import -> export is the identity."""
    import hashlib
    import importlib.util
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    dev = torch.device("cuda", 0)
    qb, K, L, R = 26, 47, 150, 1_000_000
    genome = torch.randint(0, 4, (20_000_000,), device=dev, dtype=torch.uint8,
                           generator=torch.Generator(device=dev).manual_seed(5))
    text = bench.gen_batch_torch(torch, genome, R, L, 0.003, 0, 77, dev)
    torch.cuda.synchronize()
    rec = 2 * L + bench.NAME_W + 6
    digests = []
    for part, calls in ((1 << 23, 1), (1 << 21, 3)):
        offs, lens = bench.chunk_table(R, rec, part, 65535 if part == 1 << 23 else 8191)
        ctx = _ctx(qb=qb, k=K, max_batch_bytes=64, max_batch_keys=R * (L - K + 1) + 4096, max_batch_reads=R + 1024)
        per = (len(offs) + calls - 1) // calls
        tot = 0
        for i in range(0, len(offs), per):
            st = ctx.count_chunks(text.data_ptr(), offs[i:i + per], lens[i:i + per], on_device=True, text_bytes=text.numel())
            tot += st["kmers"]
        t = ctx.totals()
        assert tot == t.nelts
        digests.append((hashlib.sha256(ctx.blocks()).hexdigest(), t.nelts, t.ndistinct))
        if calls == 1:
            p = str(tmp_path / "big.cqf")
            ctx.export_cqf(p)
            o = cqflibs.oracle().load(p)     # decode with the CPU checker
            assert o.check_offset()
            d = o.dump()
            assert len(d) == t.ndistinct and sum(c for _, c in d) == t.nelts
            o.free()
            r1 = ctx.denoise()
            h1 = hashlib.sha256(ctx.blocks()).hexdigest()
            r2 = ctx.denoise()
            assert r1 > 0 and r2 == 0 and hashlib.sha256(ctx.blocks()).hexdigest() == h1
            p2 = str(tmp_path / "big2.cqf")
            ctx.export_cqf(p2)
            ctx2 = _ctx(qb=qb, k=K, max_batch_bytes=64, max_batch_keys=4096)
            ctx2.import_cqf(p2)
            assert hashlib.sha256(ctx2.blocks()).hexdigest() == h1
            ctx2.close()
        ctx.close()
    assert digests[0] == digests[1]


@pytest.mark.parametrize("two_sources", [False, True])
def test_sharded_flow_one_rank_matches_oracle(monkeypatch, two_sources):
    """The multi-GPU flow (hash -> route -> all-to-all -> stage -> collective try/accept, shk/dist.py) with
    one rank over RCCL: one shard is the whole filter, so the table must equal the oracle's byte for byte,
    deNoise rounds included. two_sources: the words are staged from two buffers (shk_stage_words_pair: a rank's own words
    and the received ones are never copied together), cut at an arbitrary place."""
    import torch
    import torch.distributed as dist
    from shk import dist as shkdist
    qb, k, trigger, nd, ml = 16, 31, 12000, 3, 1 << 11
    g = synth.make_genome(12000, 5)
    fq = synth.make_fastq(g, 2400, 100, 0.01, seed=33, n_frac=0.03, short_frac=0.02)
    offs, lens = chunks_by_records(fq, 100)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
    own_pg = not dist.is_initialized()
    if own_pg:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        dev = torch.device("cuda:0")
        ctx = _ctx(qb=qb, k=k, min_denoise_len=ml, max_batch_bytes=len(fq) + 1024, max_batch_keys=1 << 20,
                   shard_index=0, num_shards=1)
        st = shkdist.ShardState(trigger, nd, dev)
        rounds = removed = 0
        third = len(offs) // 3
        for a, b in ((0, third), (third, 2 * third), (2 * third, len(offs))):   # three batches
            _, nw = ctx.hash_chunks(fq, offs[a:b], lens[a:b])
            recv = shkdist.route_words(ctx, nw, qb + 8, 1, 0, dev)
            if two_sources:
                m = recv.numel() // 3 + 17
                pa, pb = recv[:m].clone(), recv[m:].clone()
                ctx.stage_words_pair(pa.data_ptr(), pa.numel(), pb.data_ptr(), pb.numel())
            else:
                ctx.stage_words(recv.data_ptr(), recv.numel())
            out = shkdist.sharded_count(ctx, st, b - a)
            rounds += out["denoise_rounds"]
            removed += out["removed"]
        q, orounds, oremoved = oracle_t1(fq, offs, lens, k, qb, trigger, nd, False, ml)
        assert not q.full()
        assert (rounds, removed) == (orounds, oremoved) and rounds >= 1
        t = ctx.totals()
        assert (t.nelts, t.ndistinct) == (q.nelts(), q.ndistinct()) == (st.nelts, st.ndistinct)
        assert ctx.blocks() == q.blocks()
        ctx.close()
        q.free()
    finally:
        if own_pg:
            dist.destroy_process_group()


@pytest.mark.parametrize("qb,k,G,nreads,L,err,nseeds", [(17, 47, 20000, 2500, 150, 0.003, 300), (16, 31, 9000, 1500, 100, 0.01, 200)])
def test_unitig_extension_matches_oracle(qb, k, G, nreads, L, err, nseeds):
    """Contiger, first slice: maximal unitigs from seed k-mers (k_extend_forward, rolled hashes, one thread per
    open end) against the oracle's restatement of get_unitig_forward (hashes from scratch)"""
    import shk
    from test_emu_kernels import _unitig_case
    stops = _unitig_case(shk, lambda **kw: _ctx(**kw), qb=qb, k=k, G=G, nreads=nreads, L=L, err=err, nseeds=nseeds)
    assert {1, 2} & stops      # branches (sequencing errors) or dead ends (genome ends) are met


@pytest.mark.parametrize("qb,k,G,nreads,L,err,repeat,every", [(17, 47, 20000, 2500, 150, 0.003, 300, 40), (16, 31, 9000, 2000, 100, 0.002, 120, 25)])
def test_find_unitigs_matches_oracle_closure(tmp_path, monkeypatch, qb, k, G, nreads, L, err, repeat, every):
    if k == 31:
        monkeypatch.setenv("SHK_WALK_STEP", "100")  # walks continue over several launches
    """Contiger, set level: all unitigs reachable from sparse seeds (shk_find_unitigs: batched device extensions,
    branch neighbours queued, each unitig kept once) = an independent closure over the oracle's get_unitig_forward"""
    from test_emu_kernels import _find_unitigs_case
    g, got, st = _find_unitigs_case(lambda **kw: _ctx(**kw), tmp_path, qb=qb, k=k, G=G, nreads=nreads, L=L, err=err,
                                    repeat=repeat, seed_every=every)
    assert st["rounds"] >= 2 and st["unitigs"] >= 3


def test_contiger_cli_unitig_set(tmp_path):
    """sh-assembly_amd/bin/Contiger (reference flags) on a .cqf + two FASTQ files: the sequences in unitigs.fa are the
    closure the oracle computes from ALL seeds (the command line prunes seeds with the traveled bits batch by batch, as
    the reference does; which seed finds a unitig changes its median, not the set of sequences)"""
    import subprocess
    from test_emu_kernels import _oracle_unitig_set, _read_unitigs
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "sh-assembly_amd", "bin", "Contiger")
    assert os.path.exists(exe), "build with make -C sh-assembly_amd"
    qb, k, G = 17, 47, 20000
    g = synth.make_genome(G, 23)
    g = np.concatenate([g[:12000], g[3000:3400], g[12000:]])          # a 400-base repeat: real branches
    fq1 = synth.make_fastq(g, 1400, 150, 0.003, seed=29)
    fq2 = synth.make_fastq(g, 1400, 150, 0.003, seed=31, name_prefix="s")
    (tmp_path / "a.fq").write_bytes(fq1)
    (tmp_path / "b.fq").write_bytes(fq2)
    (tmp_path / "files.txt").write_text("a.fq\nb.fq\n")
    fq = fq1 + fq2
    offs, lens = chunks_by_records(fq, 700)
    q, _, _ = oracle_t1(fq, offs, lens, k, qb)
    assert not q.full()
    cqf = str(tmp_path / "k47.cqf")
    q.serialize(cqf)
    out = str(tmp_path / "unitigs.fa")
    # Contiger opens the list's entries as written, relative to the working directory (contig_assembly.cpp:248-258): the
    # list lies elsewhere, its names are relative to cwd
    (tmp_path / "lists").mkdir()
    (tmp_path / "lists" / "files.txt").write_text("a.fq\nb.fq\n")
    r = subprocess.run([exe, "-k", str(k), "-i", str(tmp_path / "lists" / "files.txt"), "-c", cqf, "-o", out, "--part-size", "60000",
                        "--overhead", "4000", "--batch-chunks", "3", "--max-len", str(2 * len(g) + k)],
                       capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    got = _read_unitigs(out, k)
    O = cqflibs.oracle()
    seeds, counts = [], []
    for line in fq.split(b"\n")[1::4]:
        mid = len(line) // 2 - k // 2
        km = line[mid:mid + k]
        if len(km) < k or b"N" in km or km in seeds:
            continue
        fh, rh = O.nthash(km, k)
        c = q.count(min(fh, rh) & ((1 << (qb + 8)) - 1))
        if 2 <= c <= 1000000:
            seeds.append(km)
            counts.append(c)
    exp = _oracle_unitig_set(q, seeds, counts, k, 2, 2 * len(g) + k)
    assert set(got) == set(exp), r.stderr[-300:]
    assert "truncated: 0" in r.stderr and len(got) >= 3
    q.free()


def _oracle_counter(q, k, qb):
    O = cqflibs.oracle()
    mask = (1 << (qb + 8)) - 1

    def count(km):
        fh, rh = O.nthash(km, k)
        return q.count(min(fh, rh) & mask)
    return count


def _fasta_seqs(path):
    with open(path, "rb") as f:
        return [ln for ln in f.read().split(b"\n")[1::2] if ln]


@pytest.mark.parametrize("k,qb,mark", [(31, 17, 0), (47, 17, 1), (21, 17, 1), (64, 18, 1)])
def test_unitigs_are_the_compacted_graph_of_the_filter(tmp_path, k, qb, mark):
    """definition-level invariants (tests/unitig_invariants.py) of the device-built unitig set: a genome with repeats (real
    branches), a circular plasmid seeded many times (pure circles), sequencing errors (tips, bubbles); with and without the
    traveled-bit protocol. The filter the checker reads is the oracle's copy of the same table."""
    import numpy as np
    import shk
    import unitig_invariants as UI
    G = 24000
    g = synth.make_genome(G, 41)
    g = np.concatenate([g[:9000], g[2000:2600], g[9000:], g[15000:15300]])     # two repeats
    plasmid = synth.make_genome(700, 43)
    circ = np.concatenate([plasmid, plasmid, plasmid, plasmid[:200]])          # a tandem: reads run round the circle
    fq = synth.make_fastq(g, 2600, 150, 0.004, seed=45) + synth.make_fastq(circ, 500, 120, 0.0, seed=47, name_prefix="p")
    offs, lens = chunks_by_records(fq, 600)
    ctx = _ctx(qb=qb, k=k, max_batch_bytes=len(fq) + 1024, max_batch_keys=1 << 20)
    ctx.count_chunks(fq, offs, lens)
    q, _, _ = oracle_t1(fq, offs, lens, k, qb)
    assert ctx.blocks() == q.blocks()
    count = _oracle_counter(q, k, qb)
    seeds, counts = [], []
    for line in fq.split(b"\n")[1::4]:
        mid = len(line) // 2 - k // 2
        km = line[mid:mid + k]
        if len(km) < k or b"N" in km:
            continue
        c = count(km)
        if 2 <= c <= 1000000:
            seeds.append(km)
            counts.append(c)
    out = str(tmp_path / "u.fa")
    u = shk.UnitigSet(ctx)
    third = len(seeds) // 3
    for a, b in ((0, third), (third, 2 * third), (2 * third, len(seeds))):     # several calls: the set accumulates
        u.add_seeds(seeds[a:b], counts[a:b], k, 2, 4 * len(g), mark_traveled=bool(mark))
    st = u.write(k, out)
    u.close()
    seqs = _fasta_seqs(out)
    assert st["unitigs"] == len(seqs) and st["total_len"] == sum(len(s) for s in seqs) and len(seqs) >= 5 and st["truncated"] == 0
    from test_emu_kernels import _read_unitigs
    _read_unitigs(out, k)                       # record grammar and every L: link
    UI.check(seqs, UI.Graph(count, k, 2), seeds=seeds)
    ctx.close()
    q.free()


def test_contiger_cli_from_gpu_built_cqf(tmp_path):
    """the two command lines chained as in the README (README.md:98, 131): bin/CQF-deNoise builds the .cqf on the GPU,
    bin/Contiger loads it and writes unitigs.fa; the unitigs satisfy the compacted-graph invariants against the filter
    (read back by the oracle from the same .cqf) and cover every seed's component"""
    import subprocess
    import numpy as np
    import unitig_invariants as UI
    from test_emu_kernels import _read_unitigs
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bind = os.path.join(root, "sh-assembly_amd", "bin")
    k, G = 47, 30000
    g = synth.make_genome(G, 51)
    g = np.concatenate([g[:14000], g[5000:5500], g[14000:]])
    (tmp_path / "a.fq").write_bytes(synth.make_fastq(g, 2400, 150, 0.003, seed=53))
    (tmp_path / "b.fq").write_bytes(synth.make_fastq(g, 2400, 150, 0.003, seed=55, name_prefix="s"))
    (tmp_path / "files.txt").write_text("a.fq\nb.fq\n")
    cqf = str(tmp_path / "k47.cqf")
    r = subprocess.run([os.path.join(bind, "CQF-deNoise"), "-k", str(k), "-N", "500000", "-n", "30000", "-e", "0.003", "-f", "f",
                        "-i", str(tmp_path / "files.txt"), "-o", cqf, "--part-size", "100000", "--overhead", "4000"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = str(tmp_path / "unitigs.fa")
    r = subprocess.run([os.path.join(bind, "Contiger"), "-k", str(k), "-i", str(tmp_path / "files.txt"), "-c", cqf, "-o", out,
                        "--part-size", "100000", "--overhead", "4000", "--batch-chunks", "3"], capture_output=True, text=True, timeout=300,
                       cwd=str(tmp_path))      # Contiger opens the names as written: relative to the working directory
    assert r.returncode == 0, r.stderr
    q = cqflibs.oracle().load(cqf)
    qb = 0
    while (q.L.orc_qf_size(q.h) // 89) * 64 > (1 << (qb + 1)):
        qb += 1
    hdr = open(cqf, "rb").read(128)
    import struct
    nslots = struct.unpack_from("<Q", hdr, 16)[0]
    qb = nslots.bit_length() - 1
    count = _oracle_counter(q, k, qb)
    seqs = _fasta_seqs(out)
    _read_unitigs(out, k)
    fq = (tmp_path / "a.fq").read_bytes() + (tmp_path / "b.fq").read_bytes()
    seeds = []
    for line in fq.split(b"\n")[1::4]:
        km = line[len(line) // 2 - k // 2:][:k]
        if len(km) == k and b"N" not in km and 2 <= count(km) <= 1000000:
            seeds.append(km)
    O = cqflibs.oracle()

    def key(km):
        fh, rh = O.nthash(km, k)
        return min(fh, rh) & ((1 << (qb + 8)) - 1)
    UI.check(seqs, UI.Graph(count, k, 2), seeds=seeds, key=key)
    assert len(seqs) >= 5 and "truncated: 0" in r.stderr
    q.free()


def test_contiger_at_celegans_table_size(tmp_path):
    """BASELINE config 3 at one-GPU scale: a C. elegans-sized filter (qb 29, 0.70 GiB) built on the GPU from 8 M reads
    (60x of a 20 Mb genome), then Contiger on the device over the same reads (seeds chosen and walked batch by batch,
    -s 4 / -x 4 as suits 60x). Checked: almost every genome k-mer lies in a unitig, unitigs are long, and a sample of
    unitigs satisfies the compacted-graph invariants against the filter itself (device lookups)."""
    import importlib.util
    import random
    import unitig_invariants as UI
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_contiger", os.path.join(root, "tools", "bench_contiger.py"))
    bc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bc)

    class A:
        genome, reads, err, k, qb, amin, xmin, max_len, batch_chunks = 20_000_000, 8_000_000, 0.00234, 47, 29, 4, 4, 1 << 26, 64
    torch, shk, ctx, text, offs, lens, _ = bc.build(A)
    out = str(tmp_path / "unitigs.fa")
    nseeds, st, t_walk, t_write, walk_ms, prof = bc.walk(shk, ctx, text, offs, lens, A, out)
    del text
    seqs = _fasta_seqs(out)
    k = A.k
    assert st["unitigs"] == len(seqs) and st["truncated"] == 0
    nk = sum(len(s) - k + 1 for s in seqs)
    assert 0.98 * A.genome <= nk <= 1.10 * A.genome            # the genome's k-mers (plus the solid error k-mers), each once
    assert max(len(s) for s in seqs) > 5000
    # sampled invariants against the filter on the device
    O = cqflibs.oracle()
    mask = (1 << (A.qb + 8)) - 1
    sample = random.Random(2).sample(seqs, 80)
    need = set()
    W = 300
    for s in sample:
        idx = list(range(len(s) - k + 1))
        if len(idx) > 2 * W:
            idx = idx[:W + 1] + idx[-W - 1:]
        for i in idx:
            km = s[i:i + k]
            for x in b"ACGT":
                need.add(UI.canon(km[1:] + bytes([x])))
                need.add(UI.canon(bytes([x]) + km[:-1]))
            need.add(UI.canon(km))
        for end in (s[-k:], UI.rc(s[:k])):
            for x in b"ACGT":
                nx = end[1:] + bytes([x])
                for z in b"ACGT":
                    need.add(UI.canon(bytes([z]) + nx[:-1]))
    need = list(need)
    keys = []
    for km in need:
        fh, rh = O.nthash(km, k)
        keys.append(min(fh, rh) & mask)
    cnt, _ = ctx.lookup(keys, mode=2)
    table = dict(zip(need, cnt))
    UI.check(sample, UI.Graph(lambda km: table[km], k, A.amin), sample=None, seeds=None, window=W)
    ctx.close()


@pytest.mark.parametrize("k,qb,per_read", [(21, 17, False), (31, 17, True), (47, 17, False), (64, 18, False)])
def test_contiger_whole_pipeline_against_the_sequential_restatement(tmp_path, k, qb, per_read):
    """the whole of Contiger -- seeds taken from the reads, walks, queued branch contigs, duplicate removal, numbering,
    links, unitigs.fa -- on the GPU against the sequential restatement of the whole program
    (oracle/contiger_pipeline.cpp, tests/contiger_cases.py): canonical sequences and the canonical link set are equal;
    km / KC are equal under the read-by-read schedule and a value some schedule of the reference gives otherwise.
    A genome with repeats, a plasmid (pure circles), errors, N and lower-case reads; default thresholds -s 2 -x 2."""
    import shk
    import contiger_cases as CC
    fq = CC.reads(G=24000 if not per_read else 6000, nreads=2600 if not per_read else 700, L=150, err=0.004, plasmid=700, seed=41)
    r = CC.run_case(_ctx, shk.UnitigSet, tmp_path, k=k, qb=qb, fq=fq, chunk_reads=600, per_read=per_read, max_len=1 << 17)
    assert r["unitigs"] >= 8 and r["links"] >= 8, r


def test_contiger_randomised_against_the_sequential_restatement():
    """tools/fuzz_contiger.py on the GPU: random genomes, k (21 .. 64), read lengths, thresholds (incl. x < s and x > s),
    schedules"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_contiger.py"), "--cases", "40", "--seed", "9", "--scale", "12"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "40 cases, 0 mismatches" in r.stdout, r.stdout[-2000:] + r.stderr[-500:]
    assert "'circles': 0" not in r.stdout        # pure circles were among the cases


def test_contiger_at_celegans_table_size_default_thresholds_against_the_restatement(tmp_path):
    """BASELINE config 3 with the command line's DEFAULT thresholds (-s 2 -x 2): a C. elegans-sized filter (qb 29) built
    on the GPU from 8 M reads of a 20 Mb genome, Contiger on the device over the same reads, and the sequential
    restatement of the whole program on the CPU over the same .cqf and the same chunks. Equal: the canonical sequence set
    and the canonical link set (hundreds of thousands of unitigs: 60x coverage leaves ~10^5 error bubbles above count 2);
    km equal or -- where the batched schedule lets another seed find a unitig first -- admissible (sampled)."""
    import importlib.util
    import random
    import unitig_compare as UC
    import contiger_cases as CC
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_contiger", os.path.join(root, "tools", "bench_contiger.py"))
    bc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bc)

    class A:
        genome, reads, err, k, qb, amin, xmin, max_len, batch_chunks = 20_000_000, 8_000_000, 0.00234, 47, 29, 2, 2, 1 << 26, 64
    torch, shk, ctx, text, offs, lens, _ = bc.build(A)
    cqf = str(tmp_path / "k47.cqf")
    ctx.export_cqf(cqf)                       # before the walk sets traveled bits
    out = str(tmp_path / "unitigs.fa")
    nseeds, st, t_walk, t_write, walk_ms, prof = bc.walk(shk, ctx, text, offs, lens, A, out)
    assert st["truncated"] == 0
    host = text.cpu().numpy().tobytes()
    del text
    ctx.close()
    k = A.k
    q = cqflibs.oracle().load(cqf)
    orc_fa, ost = q.contiger(host, offs, lens, k, A.amin, A.xmin, 1000000, 1, True)
    q.free()
    dev = UC.canonical(UC.parse(open(out, "rb").read(), k), k)
    orc = UC.canonical(UC.parse(orc_fa, k), k, drop_invalid=True)
    del orc_fa
    assert dev[2] == 0
    O = cqflibs.oracle()
    sk = CC.seed_kmers(host, k)
    # unitigs one side only reports: seeds whose filter key another k-mer shares (see unitig_compare.explain_one_sided)
    one_sided = set(dev[0]) ^ set(orc[0])
    both = set(dev[0]) & set(orc[0])
    assert len(one_sided) <= len(both) // 500, (len(dev[0]), len(orc[0]), len(one_sided))
    ngroups = UC.explain_one_sided(one_sided, both, k, A.qb + 8, sk, O.seq_keys)
    dl = {l for l in dev[1] if l[0] in both and l[2] in both}
    ol = {l for l in orc[1] if l[0] in both and l[2] in both}
    assert dl == ol, (len(dl), len(ol), len(dl ^ ol))
    assert len(both) > 50000 and sum(len(c) - k + 1 for c in both) >= 0.98 * A.genome
    diff = [c for c in both if dev[0][c] != orc[0][c]]
    print("unitigs", len(dev[0]), "links", len(dev[1]), "one-sided", len(one_sided), "in", ngroups, "groups; km differ", len(diff),
          "stale links in the restatement", orc[2], "oracle", ost)
    assert len(diff) <= 0.25 * len(both)
    qc = cqflibs.oracle().load(cqf)
    mask = (1 << (A.qb + 8)) - 1

    def count(km):
        fh, rh = O.nthash(km, k)
        return qc.count(min(fh, rh) & mask)
    small = [c for c in diff if len(c) <= 1500]
    for c in random.Random(3).sample(small, min(150, len(small))):
        adm = UC.admissible_km(c, k, count, sk)
        assert dev[0][c][0] in adm and orc[0][c][0] in adm, (dev[0][c], orc[0][c], sorted(adm))
    qc.free()


def _sha256_file(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def test_full_size_schedule_independent_of_batching(tmp_path):
    """BASELINE size (C. elegans sizing: qb 29, one bench batch of 8 M reads = 832 M k-mers) with deNoise points inside
    the batch: table, counters, rounds and removed counts do not depend on how the 302 chunks are split into calls (the
    schedule is a property of the chunk sequence, CQF_mt.h:837); the exported .cqf decodes consistently on the CPU."""
    import hashlib
    import importlib.util
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    dev = torch.device("cuda", 0)
    qb, K, L, R = 29, 47, 150, 8_000_000
    genome = torch.randint(0, 4, (100_000_000,), device=dev, dtype=torch.uint8,
                           generator=torch.Generator(device=dev).manual_seed(9))
    text = bench.gen_batch_torch(torch, genome, R, L, 0.00234, 0, 91, dev)
    torch.cuda.synchronize()
    rec = 2 * L + bench.NAME_W + 6
    offs, lens = bench.chunk_table(R, rec)
    results = []
    for calls in (1, 3):
        ctx = _ctx(qb=qb, k=K, trigger=60_000_000, num_denoise=2, max_batch_bytes=64, max_batch_keys=R * (L - K + 1) + 4096,
                   max_batch_reads=R + 1024)
        per = (len(offs) + calls - 1) // calls
        rounds = removed = kmers = 0
        for i in range(0, len(offs), per):
            st = ctx.count_chunks(text.data_ptr(), offs[i:i + per], lens[i:i + per], on_device=True, text_bytes=text.numel())
            rounds += st["denoise_rounds"]
            removed += st["removed"]
            kmers += st["kmers"]
        t = ctx.totals()
        assert t.nelts == kmers - removed and rounds == 2 and removed > 0
        results.append((hashlib.sha256(ctx.blocks()).hexdigest(), t.nelts, t.ndistinct, rounds, removed))
        if calls == 1:
            p = str(tmp_path / "full.cqf")
            ctx.export_cqf(p)
            o = cqflibs.oracle().load(p)
            assert o.check_offset()
            assert o.L.orc_qf_dump(o.h, None, None, 0) == t.ndistinct      # entries found by the CPU decode
            assert (o.nelts(), o.ndistinct()) == (t.nelts, t.ndistinct)    # header counters
            o.free()
            file_sha = _sha256_file(p)
            os.remove(p)
        ctx.close()
    assert results[0] == results[1]
    # BYTE FOR BYTE at this size: the same 302 chunks through the t = 1 schedule on the CPU checker -- the compiled reference
    # (oracle/_ref: the real gqf.c + nthash.hpp; ~2 min at ~7 M k-mers/s) when its library travelled, else the C restatement
    lib = cqflibs.ref() if cqflibs.have_ref() else cqflibs.oracle()
    host = text.cpu().numpy()
    del text, genome
    q = lib.new(qb)
    left, trigger, rounds, removed = 2, 60_000_000, 0, 0
    for a, n in zip(offs, lens):
        getattr(lib.L, lib.p + "reads_to_kmers")(q.h, C.cast(host.ctypes.data + a, C.c_char_p), n, K)
        if left and q.ndistinct() >= trigger:                      # CQF_mt.h:837
            left -= 1
            removed += q.denoise_round()
            rounds += 1
    assert (rounds, removed, q.nelts(), q.ndistinct()) == (results[0][3], results[0][4], results[0][1], results[0][2])
    assert hashlib.sha256(q.blocks()).hexdigest() == results[0][0], "qb-29 table bytes differ from the CPU reference"
    p = str(tmp_path / "cpu.cqf")
    q.serialize(p)                                                 # header (qfmetadata) + blocks as the reference writes them
    q.free()
    assert _sha256_file(p) == file_sha, ".cqf file differs from the CPU reference's"
    os.remove(p)


def test_region_hash_overflow_halves_the_chunk_range():
    """more than 512 distinct new keys for one 256-quotient region in one pass: the pass reports it before anything
    is written, the host takes fewer chunks at once; a single chunk that overflows is a hard error that leaves the
    table untouched"""
    import torch
    import shk
    from cqf_canon import build_blocks
    qb, hb = 11, 19
    keys = [((300 + (i % 200)) << 8) | (i * 7 % 256) for i in range(700)]      # 700 distinct keys, quotients 300..499 (region 1)
    keys = list(dict.fromkeys(keys))
    assert len(keys) > 600
    ctx = _ctx(qb=qb, k=21, max_batch_bytes=64, max_batch_keys=1 << 12)
    # two chunks of about half the keys each: every half fits, both together do not
    half = len(keys) // 2
    words = [k | (0 << hb) for k in keys[:half]] + [k | (1 << hb) for k in keys[half:]]
    t = _upload(words)
    torch.cuda.synchronize()
    st = ctx.count_words(t.data_ptr(), t.numel(), 2)
    assert st["kmers"] == len(keys)
    want = build_blocks(qb, hb, {k: 1 for k in keys})
    assert ctx.blocks() == want
    # one chunk with another 600 new keys for the same region cannot be split
    more = [((300 + (i % 200)) << 8) | ((i * 7 + 3) % 256) for i in range(1500)]
    more = [k for k in dict.fromkeys(more) if k not in set(keys)][:600]
    t2 = _upload(more)
    torch.cuda.synchronize()
    with pytest.raises(shk.ShkError):
        ctx.count_words(t2.data_ptr(), t2.numel(), 1)
    assert ctx.blocks() == want
    ctx.close()


def test_overlapped_upload_matches_oracle():
    """shk_upload_text: batches copied on the context's copy stream while the previous batch is counted; three
    batches through the two alternating buffers give the oracle's table"""
    import ctypes as C
    qb, k = 19, 47
    g = synth.make_genome(20000, 3)
    fqs = [synth.make_fastq(g, 900, 150, 0.01, seed=40 + i, name_prefix="b%d_" % i) for i in range(3)]
    tabs = [chunks_by_records(fq, 300) for fq in fqs]
    bufs = [C.create_string_buffer(fq, len(fq)) for fq in fqs]
    ctx = _ctx(qb=qb, k=k, max_batch_bytes=max(len(fq) for fq in fqs) + 1024, max_batch_keys=900 * 150)
    nxt = ctx.upload_text(C.addressof(bufs[0]), len(fqs[0]))
    for i in range(3):
        cur = nxt
        if i + 1 < 3:
            nxt = ctx.upload_text(C.addressof(bufs[i + 1]), len(fqs[i + 1]))
        ctx.count_chunks(cur, tabs[i][0], tabs[i][1], on_device=True, text_bytes=len(fqs[i]))
    whole = b"".join(fqs)
    offs, lens, base = [], [], 0
    for fq, (o, l) in zip(fqs, tabs):
        offs += [base + x for x in o]
        lens += l
        base += len(fq)
    q, _, _ = oracle_t1(whole, offs, lens, k, qb)
    assert not q.full()
    assert ctx.blocks() == q.blocks()
    ctx.close()
    q.free()


def test_randomised_configurations_match_oracle():
    """tools/fuzz_gpu.py: 200 random small configurations (filter size, k, read mix, chunking, batching, deNoise trigger /
    rounds / minimum range length, --endDeNoise) through shk_count_chunks: table bytes, header, counters, rounds and removed
    counts equal the oracle's t = 1 build in every one (10,500 cases were run this way in round 1, none differed)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_gpu.py"), "--cases", "200", "--seed", "3"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-500:]
    assert "0 mismatches" in r.stdout


# ---------------------------------------------------------------- f-4 / a-9: counted inserts, dump, merge, shard stitch
def _f4_makers():
    def mk(qb):
        return _ctx(qb=qb, k=21, max_batch_keys=1 << 22)

    def mk_shard(qb, s, n):
        return _ctx(qb=qb, k=21, max_batch_keys=1 << 22, shard_index=s, num_shards=n)
    return mk, mk_shard


def test_counted_insert_and_dump_match_reference():
    """shk_insert_counted == qf_insert_advance(count) of the compiled reference (gqf.c:2024-2136), byte for byte, incl.
    counts up to 2^35 (several passes of 2^22 occurrences), codec edge remainders, dense clusters, the overflow tail;
    shk_dump == the reference's iterator (gqf.c:2474-2601), incl. where it stops early"""
    import random
    import f4_scenarios as F
    mk, _ = _f4_makers()
    rng = random.Random(5)
    F.check_counted_and_dump(mk, 12, F.pairs(rng, 12, 1000, 1 << 20))
    F.check_counted_and_dump(mk, 14, F.pairs(rng, 14, 4000, 1 << 24))
    F.check_counted_and_dump(mk, 12, F.pairs(rng, 12, 900, 1 << 16, cluster=(2000, 300)))     # offsets saturate at 255
    F.check_counted_and_dump(mk, 12, F.pairs(rng, 12, 150, 1 << 16, cluster=(4000, 96)))      # runs end in the tail (not near its end:
    # the reference's 8-byte slot accesses overrun its buffer there)
    F.check_counted_and_dump(mk, 10, F.pairs(rng, 10, 40, 1 << 35), batches=2)                # > 2^32 per key
    # more distinct new keys in one 256-quotient region than its LDS hash holds: the call splits its pairs
    dense = [((1024 + i % 256) << 8 | (i * 37 + (i // 256) * 11) % 256, 1 + i % 2) for i in range(0, 1300)]
    dense = list({k: c for k, c in dense}.items())
    assert len(dense) > 1000
    F.check_counted_and_dump(mk, 12, dense, batches=1)


def test_merge_matches_reference_qf_merge():
    """shk_merge / shk_multi_merge == the compiled qf_merge / qf_multi_merge (gqf.c:2614-2704): table bytes, dump, counters"""
    import random
    import f4_scenarios as F
    mk, _ = _f4_makers()
    rng = random.Random(6)
    F.check_merge(mk, 12, [F.pairs(rng, 12, 400, 1 << 16), F.pairs(rng, 12, 500, 1 << 16), F.pairs(rng, 12, 300, 1 << 16)])
    a = F.pairs(rng, 14, 1500, 1 << 12, cluster=(3000, 4000))
    b = [(k, c + 3) for k, c in a[::2]] + F.pairs(rng, 14, 800, 1 << 12, cluster=(3500, 4000))   # one long cluster: the big LDS image
    b = list({k: c for k, c in b}.items())
    F.check_merge(mk, 14, [a, b])
    F.check_merge(mk, 16, [F.pairs(rng, 16, 8000, 1 << 10), F.pairs(rng, 16, 7000, 1 << 10), F.pairs(rng, 16, 4000, 300),
                           F.pairs(rng, 16, 2000, 300)])


def test_merge_of_tail_entries():
    """The reference's qfi_next reports "end" when it steps inside a run onto a slot behind nslots (gqf.c:2537-2539);
    qf_merge's drain loops stop there, so it can lose entries of its inputs' overflow tails. shk_merge adds everything
    the source holds: its result is the canonical table of the union, and the reference's result is contained in it,
    differing only by tail entries."""
    import random
    import f4_scenarios as F
    if not cqflibs.have_ref():
        pytest.skip("needs oracle/_ref")
    lib = cqflibs.ref()
    mk, _ = _f4_makers()
    rng = random.Random(8)
    qb = 10
    a = F.pairs(rng, qb, 80, 300, cluster=(1000, 24))          # quotients 1000..1023: the last runs lie in the tail
    b = F.pairs(rng, qb, 60, 300, cluster=(0, 900))
    qa, qb_ = F.build(lib, qb, a), F.build(lib, qb, b)
    assert len(qa.dump()) < len(a), "scenario must produce entries the reference iterator does not reach"
    want = lib.new(qb)
    want.merge_from(qa, qb_)
    ca, cb, dst, rc = mk(qb), mk(qb), mk(qb), mk(qb)
    ca.insert_counted([k for k, _ in a], [c for _, c in a])
    cb.insert_counted([k for k, _ in b], [c for _, c in b])
    dst.merge(ca)
    dst.merge(cb)
    union = F.build(cqflibs.oracle(), qb, a + b)
    assert dst.blocks() == union.blocks()
    rc.import_blocks(want.blocks())
    ref_has, ours = dict(rc.dump()), dict(dst.dump())
    assert all(ours.get(k) == c for k, c in ref_has.items())          # what the reference merged is there, same counts
    lost = [k for k in ours if k not in ref_has]
    assert lost and all((k >> 8) >= 1000 for k in lost)               # and it lost only entries of the tail cluster
    for x in (ca, cb, dst, rc):
        x.close()
    for x in (qa, qb_, want, union):
        x.free()


def test_import_shards_equals_single_table():
    """quotient-range shards (own overflow tails) stitched on the device == the single table, incl. clusters that cross
    shard borders and a shard whose last runs lie in its tail"""
    import random
    import f4_scenarios as F
    mk, mk_shard = _f4_makers()
    rng = random.Random(9)
    F.check_shards(mk, mk_shard, 12, F.pairs(rng, 12, 1200, 1 << 12), 2)
    F.check_shards(mk, mk_shard, 14, F.pairs(rng, 14, 5000, 1 << 12), 8)
    F.check_shards(mk, mk_shard, 12, F.pairs(rng, 12, 400, 300, cluster=(1900, 300)), 2)     # cluster across the border at 2048
    F.check_shards(mk, mk_shard, 12, F.pairs(rng, 12, 900, 300, cluster=(500, 3000)), 4)
    F.check_shards(mk, mk_shard, 13, F.pairs(rng, 13, 2200, 300, cluster=(1000, 7000)), 8)


def test_build_then_dump_then_rebuild_is_identity():
    """a filter built from reads, dumped and re-inserted with counts gives the same bytes (qf_insert_advance(count)
    == count single inserts), and its dump equals the oracle's"""
    g = synth.make_genome(30000, 3)
    fq = synth.make_fastq(g, 4000, 120, 0.01, seed=4, n_frac=0.02)
    offs, lens = chunks_by_records(fq, 500)
    ctx = _ctx(qb=18, k=31, max_batch_bytes=len(fq) + 1024, max_batch_keys=1 << 20)
    ctx.count_chunks(fq, offs, lens)
    q, _, _ = oracle_t1(fq, offs, lens, 31, 18)
    d = ctx.dump()
    assert d == q.dump()
    again = _ctx(qb=18, k=31, max_batch_keys=1 << 20)
    st = again.insert_counted([k for k, _ in d], [c for _, c in d])
    assert st["new_distinct"] == len(d) and st["kmers"] == q.nelts()
    assert again.blocks() == ctx.blocks() == q.blocks()
    for x in (ctx, again):
        x.close()
    q.free()


# ---------------------------------------------------------------- BASELINE configs 4 and 5 at one-GPU size
def _torch_reads(nreads, genome_len, err, seed):
    import importlib.util
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    dev = torch.device("cuda", 0)
    genome = torch.randint(0, 4, (genome_len,), device=dev, dtype=torch.uint8, generator=torch.Generator(device=dev).manual_seed(seed))
    text = bench.gen_batch_torch(torch, genome, nreads, 150, err, 0, seed + 1, dev)
    offs, lens = bench.chunk_table(nreads, 2 * 150 + bench.NAME_W + 6)
    return torch, dev, text, offs, lens


def _memory_report(ctx, nregions, batch_keys):
    t = ctx.totals()
    # (csrc/cqf_kernels.hip: SHK_SPILL_STRIDE = 768, SHK_NC_CAP = 128 entries of 2 bytes)
    m = {"table_bytes_x2": 2 * t.table_bytes, "spill_records": nregions * 768, "first_chunk_records": nregions * 256,
         "key_words_x2": 2 * 8 * batch_keys}
    assert m["spill_records"] + m["first_chunk_records"] <= 1.5 * m["table_bytes_x2"]
    return m


def test_config4_one_shard_of_eight_human_sized(tmp_path):
    """BASELINE config 4 (human 30x, k = 31, quotient range sharded over 8 GPUs) as ONE rank sees it: shard 3 of a qb-34
    filter (2^31 of its 2^34 slots + its own tail, a 3.0 GB table, 8.4 M regions) receives the keys it owns out of a
    routed batch, through the collective flow with a one-rank group. Size-independent properties: counters equal the
    routed words; the dump adds up; sampled lookups equal the words' multiplicities; inserting the batch again doubles
    every count (linearity); a deNoise round removes exactly the singletons (up to the protected range ends); an
    export / import round trip is the identity. Also prints the memory the shard needs."""
    import torch.distributed as dist
    from shk import dist as shkdist
    torch, dev, text, offs, lens = _torch_reads(2_000_000, 40_000_000, 0.005, 61)
    qb, k, S, me = 34, 31, 8, 3
    nkeys = 2_000_000 * (150 - k + 1)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29800 + os.getpid() % 100))
    own_pg = not dist.is_initialized()
    if own_pg:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        ctx = _ctx(qb=qb, k=k, max_batch_bytes=64, max_batch_keys=nkeys + 4096, max_batch_reads=2_000_000 + 1024, shard_index=me, num_shards=S,
                   min_denoise_len=0)
        t = ctx.totals()
        assert t.nslots == 1 << 31 and t.table_bytes > 2.9e9
        print("config 4 shard memory:", _memory_report(ctx, (1 << 31) // 256, nkeys))
        hb = qb + 8
        _, nw = ctx.hash_chunks(text.data_ptr(), offs, lens, on_device=True, text_bytes=text.numel())
        dp, cnts = ctx.route_words(nw, S)
        assert sum(cnts) == nw and min(cnts) > 0.11 * nw                 # uniform hash: every owner gets about an eighth
        words = shkdist.wrap_words(dp, nw, dev)
        mine = words[sum(cnts[:me]): sum(cnts[:me + 1])].clone()
        keys = mine & ((1 << hb) - 1)
        assert int((keys >> (hb - 3)).min()) == me == int((keys >> (hb - 3)).max())     # the top three quotient bits name the owner
        st = shkdist.ShardState(1 << 62, 0, dev)
        ctx.stage_words(mine.data_ptr(), mine.numel())
        out = shkdist.sharded_count(ctx, st, len(offs) * S)
        uniq, mult = torch.unique(keys, return_counts=True)
        t = ctx.totals()
        assert out["kmers"] == mine.numel() == t.nelts and t.ndistinct == uniq.numel() == out["new_distinct"]
        n = ctypes_dump_count(ctx)
        assert n == t.ndistinct
        pick = torch.randperm(uniq.numel(), device=dev)[:200000]
        cnt, _ = ctx.lookup(uniq[pick].tolist(), mode=2)
        assert cnt == mult[pick].tolist()
        absent = [(me << (hb - 3)) | (int(x) & ((1 << (hb - 3)) - 1)) for x in torch.randint(0, 1 << 40, (20000,)).tolist()]
        picked = set(uniq[pick].tolist())
        absent = [a for a in absent if a not in picked]
        cnt0, _ = ctx.lookup(absent[:5000], mode=2)
        present = set(uniq.tolist()) if uniq.numel() < 3_000_000 else None
        if present is not None:
            assert all(c == 0 or a in present for a, c in zip(absent[:5000], cnt0))
        # linearity: the same words again
        ctx.stage_words(mine.data_ptr(), mine.numel())
        shkdist.sharded_count(ctx, st, len(offs) * S)
        t2 = ctx.totals()
        assert t2.nelts == 2 * t.nelts and t2.ndistinct == t.ndistinct
        cnt2, _ = ctx.lookup(uniq[pick][:50000].tolist(), mode=2)
        assert cnt2 == (2 * mult[pick][:50000]).tolist()
        # round trip through the host layout
        blocks = ctx.blocks()
        import hashlib
        h1 = hashlib.sha256(blocks).hexdigest()
        other = _ctx(qb=qb, k=k, max_batch_bytes=64, max_batch_keys=4096, shard_index=me, num_shards=S)
        other.import_blocks(blocks, t2.nelts, t2.ndistinct)
        del blocks
        assert ctypes_dump_count(other) == t2.ndistinct
        assert hashlib.sha256(other.blocks()).hexdigest() == h1
        other.close()
        ctx.close()
    finally:
        if own_pg:
            dist.destroy_process_group()


def ctypes_dump_count(ctx):
    import ctypes as C
    n = C.c_uint64()
    assert ctx.L.shk_dump(ctx.h, None, None, 0, 0, 0, C.byref(n)) == 0
    return n.value


def test_config5_error_profile_sizing_and_qb33_filter(tmp_path):
    """BASELINE config 5 (human 100x, k = 31, per-base error profile, deNoise rounds): the sizing of src/CQF-deNoise.cpp with
    --errorProfile (true2falseKmer_DP) gives a qb-33 class filter; a whole qb-33 filter (2^33 slots, 11.9 GB table,
    33.5 M regions) on ONE GPU takes batches with deNoise rounds firing at the t = 1 schedule's points. Properties:
    counters add up across batches and rounds (nelts = presented - removed, every removed entry was a singleton), a
    round leaves no removable singleton behind except the protected range ends, lookups match multiplicities."""
    import ctypes as C
    import torch
    from test_host_logic import _sizing_api
    rates = [0.001 + 0.019 * i / 149 for i in range(150)]
    pf = tmp_path / "profile.txt"
    pf.write_text("".join("%.17g\n" % r for r in rates))
    L = _sizing_api()
    out = (C.c_uint64 * 8)()
    K, n_true, N = 31, 2_900_000_000, 300_000_000_000
    L.shkh_size_filter_profile(K, n_true, N, str(pf).encode(), -1, 0.0, out)
    qb_formula, rounds_formula = out[0], out[2]
    assert qb_formula in (33, 34) and rounds_formula >= 1
    qb = 33
    torch_, dev, text, offs, lens = _torch_reads(3_000_000, 30_000_000, 0.01, 71)
    nkeys = 3_000_000 * (150 - K + 1)
    # a trigger that fires twice inside these reads (the data set is a sliver of the one the sizing is for)
    ctx = _ctx(qb=qb, k=K, trigger=45_000_000, num_denoise=2, max_batch_bytes=64, max_batch_keys=nkeys + 4096, max_batch_reads=3_000_000 + 1024)
    t = ctx.totals()
    assert t.nslots == 1 << 33 and t.table_bytes > 11.8e9
    print("config 5 filter memory:", _memory_report(ctx, (1 << 33) // 256, nkeys))
    third = len(offs) // 3
    presented = removed = rounds = 0
    for a, b in ((0, third), (third, 2 * third), (2 * third, len(offs))):
        st = ctx.count_chunks(text.data_ptr(), offs[a:b], lens[a:b], on_device=True, text_bytes=text.numel())
        presented += st["kmers"]
        removed += st["removed"]
        rounds += st["denoise_rounds"]
    t = ctx.totals()
    assert rounds == 2 and t.rounds_left == 0 and removed > 0
    assert t.nelts == presented - removed and t.ndistinct == ctypes_dump_count(ctx)
    # an extra round now removes the singletons that arrived after the last one; a second extra round finds nothing more
    # than the range-end singletons the first extra round protected (at most one per 2^20-slot range)
    r1 = ctx.denoise()
    r2 = ctx.denoise()
    assert r1 > 0 and r2 <= (1 << 33) // (1 << 20) + 1
    t2 = ctx.totals()
    assert t2.nelts == t.nelts - r1 - r2 and t2.ndistinct == t.ndistinct - r1 - r2 == ctypes_dump_count(ctx)
    ctx.close()


# ---------------------------------------------------------------- the driver's bench contract
@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_fields():
    """`python bench.py --gpus 1 --steps K --warmup W` as the driver runs it (shorter): stdout is exactly one JSON line with
    the metric fields, the path-level `roofline` and the `cpu_baseline` of the compiled reference"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["unit"] == "k-mers/s" and d["value"] > 1e9 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert (d["n_gpus"], d["steps"], d["warmup"]) == (1, 3, 1) and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert abs(d["ms_per_step"] * d["steps"] / 1e3 - d["build_time_s"]) < 1e-6
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0 < rf["frac"] < 1
    cb = d["cpu_baseline"]
    assert cb["kind"] == "reference" and cb["unit"] == "k-mers/s" and cb["cores"] >= 1 and cb["value"] > 1e6 and cb["sample"]
    # every k-mer of the build was counted: steps x reads x (L - k + 1), minus the reads' N restarts
    assert d["build_kmers"] > 0.99 * 3 * 8_000_000 * 104


def test_overlapped_front_end_matches_oracle():
    """shk_prepare_chunks / shk_count_prepared: the front end of batch s+1 (helper thread, second stream, shadow buffers)
    runs while batch s is rebuilt; table, header, counters, rounds and removed counts equal the oracle's t = 1 build --
    two batches prepared ahead, deNoise rounds inside, and the same chunks through shk_count_chunks give the same bytes"""
    qb, k, trig, nd, ml = 19, 47, 60000, 4, 1 << 11
    fq = synth.make_fastq(synth.make_genome(30000, 5), 4200, 150, 0.01, seed=9, n_frac=0.05, short_frac=0.02, lower_frac=0.02)
    offs, lens = chunks_by_records(fq, 150)
    q, orounds, oremoved = oracle_t1(fq, offs, lens, k, qb, trig, nd, False, ml)
    assert not q.full() and orounds >= 2
    ctx = _ctx(qb=qb, k=k, trigger=trig, num_denoise=nd, min_denoise_len=ml, max_batch_bytes=len(fq) + 1024, max_batch_keys=4200 * 150)
    nb = 7
    per = (len(offs) + nb - 1) // nb
    parts = [(a, min(a + per, len(offs))) for a in range(0, len(offs), per)]
    rounds = removed = 0
    ctx.prepare_chunks(fq, offs[parts[0][0]:parts[0][1]], lens[parts[0][0]:parts[0][1]])
    for i in range(len(parts)):
        if i + 1 < len(parts):
            a, b = parts[i + 1]
            ctx.prepare_chunks(fq, offs[a:b], lens[a:b])          # two batches are prepared ahead at this moment
        st = ctx.count_prepared()
        rounds += st["denoise_rounds"]
        removed += st["removed"]
    t = ctx.totals()
    assert (rounds, removed) == (orounds, oremoved) and (t.nelts, t.ndistinct) == (q.nelts(), q.ndistinct())
    assert ctx.blocks() == q.blocks() and ctx.header() == oracle_header(q)
    import shk
    with pytest.raises(shk.ShkError):
        ctx.count_prepared()                                      # nothing is prepared
    ctx.close()
    q.free()


def test_multi_gpu_front_end_one_rank_over_rccl(tmp_path):
    """python -m shk.count (the multi-GPU CQF-deNoise: parts round-robin, all-to-all, collective rebuild, every rank
    pwriting its blocks) with ONE rank over RCCL on the real kernels: the golden .cqf files come out byte for byte.
    (2 and 4 ranks run over gloo on the emulator in tests/test_dist_gloo.py; more GPUs are the driver's to launch.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    G = os.path.join(root, "tests", "golden")
    fx = json.load(open(os.path.join(G, "fastq_builds.json")))
    for bi, b in enumerate(fx["builds"][:4]):
        c = b["cfg"]
        lst = tmp_path / ("files%d.txt" % bi)
        lst.write_text("\n".join(os.path.join(G, f) for f in c["files"]) + "\n")
        out = str(tmp_path / ("out%d.cqf" % bi))
        args = ["-k", str(c["k"]), "-n", "6000", "-N", "100000", "-e", "0.01", "-f", "f", "-i", str(lst), "-o", out,
                "--deNoise", str(c["nd"]), "--rounds", str(c["nd"]), "--qb", str(c["qb"]), "--trigger", str(min(c["trigger"], 1 << 62)),
                "--part-size", str(c["ps"]), "--overhead", str(c["ov"]), "--min-denoise-len", str(c["ml"]), "--parts-per-call", "3"]
        if c["end"]:
            args.append("--endDeNoise")
        env = dict(os.environ, PYTHONPATH=os.path.join(root, "sh-assembly_amd"), SHK_SAMPLE_STRIDE="2")
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                            "--master-port", str(29650 + bi), "-m", "shk.count"] + args, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        assert open(out, "rb").read() == open(os.path.join(G, b["cqf"]), "rb").read(), (c, r.stderr[-600:])


def test_qb34_whole_filter_context_fits_one_gpu():
    """with 1 KiB of per-region records instead of 2 a WHOLE human-sized filter (qb 34: two 47.8 GB tables,
    2^26 regions, 64 GiB of spill + first-chunk records) is created on one 288 GB MI355X, takes a batch and a deNoise round"""
    import torch
    free, total = torch.cuda.mem_get_info()
    if free < 200 * (1 << 30):
        pytest.skip("needs ~170 GB of free HBM")
    qb, k = 34, 31
    fq = synth.make_fastq(synth.make_genome(50000, 3), 4000, 150, 0.01, seed=5, n_frac=0.0)
    offs, lens = chunks_by_records(fq, 500)
    ctx = _ctx(qb=qb, k=k, trigger=1 << 40, num_denoise=2, max_batch_bytes=len(fq) + 1024, max_batch_keys=4000 * 150)
    t = ctx.totals()
    assert t.table_bytes == ((1 << qb) + int(10 * 2 ** (qb / 2)) + 63) // 64 * 89
    st = ctx.count_chunks(fq, offs, lens)
    assert st["kmers"] == 4000 * (150 - k + 1)
    nd = ctx.totals().ndistinct
    removed = ctx.denoise()
    assert 0 < removed < nd and ctx.totals().ndistinct == nd - removed
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pack", [True, False])
def test_roll_kernels_key_multiset_over_k_and_read_shapes(monkeypatch, pack):
    """roll_cases.run on the GPU: the roll kernels' (2-bit staged and text) and the wave-per-read kernel's key streams over
    k = 5..129, read lengths around the 16-base rounds and 64-base units, lower case, 'N' and IUPAC bytes"""
    import torch
    import roll_cases
    from shk import dist as shkdist
    if not pack:
        monkeypatch.setenv("SHK_NO_PACK", "1")
    dev = torch.device("cuda", 0)

    def mk(**kw):
        ctx = _ctx(**kw)
        ctx.read_words = lambda dp, n: [x & 0xFFFFFFFFFFFFFFFF for x in shkdist.wrap_words(dp, n, dev).cpu().tolist()]
        return ctx
    roll_cases.run(mk, pack)


@pytest.mark.gpu
def test_last_partition_level_with_region_slots_and_its_exact_fallback():
    """roll_cases.run_slots on the GPU: fixed-capacity region slots at the last partition level, the exact fallback when a
    region overflows its slot, and the switch-off after two overflowing batches in a row"""
    import roll_cases
    roll_cases.run_slots(_ctx)


@pytest.mark.gpu
def test_one_pass_denoise_point_with_a_crowded_region(monkeypatch):
    """roll_cases.run_fused_point_with_a_crowded_region on the GPU: a region with 2^15 words or more in the batch of a deNoise
    point sends the one-pass point (16-bit counters) to the general path; same rounds, removed counts and bytes"""
    import roll_cases
    monkeypatch.setenv("SHK_SAMPLE_STRIDE", "2")
    roll_cases.run_fused_point_with_a_crowded_region(_ctx)
