"""world_size-2 gloo test of the sharded path (CPU): two processes, each owning half of the
quotient range, hash their own reads, exchange key words (all-to-all), stage them and take
the deNoise decisions together (shk/dist.py). The kernels run in the CPU emulator build.
Checked against the oracle over the chunks in the interleaved global order."""
import ctypes as C
import os
import subprocess
import sys

import pytest

import cqflibs
import synth
from fastq_util import chunks_by_records

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu", "libshk_emu.so")
QB, K, TRIG, ND, ML = 11, 28, 650, 2, 1 << 20
TRIG_ONE_PASS = {2: 420, 4: 880, 8: 1500}   # (chosen so that no batch holds two points: every point goes the one-pass way)


def _data(rank, shape=(24, 6)):
    g = synth.make_genome(300, 7)
    fq = synth.make_fastq(g, shape[0], 90, 0.01, seed=50 + rank, n_frac=0.05)
    return fq, chunks_by_records(fq, shape[1])


def _worker(rank, world, port, q, nd=ND, out_path=None, ml=ML, trig=TRIG, shape=(24, 6)):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "sh-assembly_amd"))
    import shk
    from shk import dist as shkdist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    fq, (offs, lens) = _data(rank, shape)
    ctx = shk.Context(qb=QB, k=K, min_denoise_len=ml, max_batch_bytes=1 << 20, max_batch_keys=1 << 16,
                      shard_index=rank, num_shards=world, threads_per_group=64, hash_groups=2, lib_path=EMU)
    st = shkdist.ShardState(trig, nd, dev)
    hb = QB + 8
    # two batches, the second one's exchange started before the first is staged (the pipelined form bench.py uses)
    half = len(offs) // 2
    out = {"kmers": 0, "new_distinct": 0, "removed": 0, "denoise_rounds": 0}
    _, nw = ctx.hash_chunks(fq, offs[:half], lens[:half])
    ex = shkdist.Exchange(ctx, nw, hb, world, rank, dev)
    _, nw2 = ctx.hash_chunks(fq, offs[half:], lens[half:])
    ex2 = shkdist.Exchange(ctx, nw2, hb, world, rank, dev)
    allw = []
    for e, n in ((ex, half), (ex2, len(offs) - half)):
        recv = e.wait()
        allw.append(recv.clone())
        if rank % 2 and recv.numel() > 40:      # (odd ranks: from two buffers, shk_stage_words_pair)
            m = recv.numel() // 3 + 17
            pa, pb = recv[:m].clone(), recv[m:].clone()
            ctx.stage_words_pair(pa.data_ptr(), pa.numel(), pb.data_ptr(), pb.numel())
        else:
            ctx.stage_words(recv.data_ptr(), recv.numel())
        o = shkdist.sharded_count(ctx, st, n * world)
        for kk in out:
            out[kk] += o[kk]
    shkdist.check(ctx, st)
    recv = torch.cat(allw)
    if out_path:
        shkdist.export_cqf(ctx, st, out_path, world, rank, dev, QB, K)
    # (key, count) content of this shard through lookups of every key it received
    keys = sorted(set(int(x) & ((1 << hb) - 1) for x in recv.tolist()))
    cnt, _ = ctx.lookup(keys, mode=2)
    t = ctx.totals()
    q.put((rank, out, st.ndistinct, st.nelts, {k: c for k, c in zip(keys, cnt) if c}, t.ndistinct,
           (st.one_pass_points, st.other_points, st.guesses, st.guesses_right, st.inexact_rounds)))
    ctx.close()
    dist.destroy_process_group()


def _run(world, nd, out_path, ml=ML, trig=TRIG, shape=(24, 6)):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    import torch.multiprocessing as mp
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29600 + (os.getpid() * 7 + world * 3 + nd) % 300
    procs = [ctxm.Process(target=_worker, args=(r, world, port, q, nd, out_path, ml, trig, shape)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    # oracle: one filter, chunks in the interleaved order rank0.c0, rank1.c0, rank0.c1, ...
    O = cqflibs.oracle()
    o = O.new(QB)
    d = [_data(r, shape) for r in range(world)]
    nch = len(d[0][1][0])
    left, rounds, removed = nd, 0, 0
    for j in range(nch):
        for r in range(world):
            fq, (offs, lens) = d[r]
            o.reads_to_kmers(fq[offs[j]:offs[j] + lens[j]], K)
            if left and o.ndistinct() >= trig:
                left -= 1
                removed += o.denoise_round(ml)
                rounds += 1
    assert not o.full()
    return res, o, rounds


@pytest.mark.parametrize("world", [2, 4])
def test_shards_match_single_filter(world, tmp_path):
    out_path = str(tmp_path / "stitched.cqf")
    res, o, rounds = _run(world, ND, out_path)
    out0 = res[0][1]
    assert all(r[1] == out0 for r in res)               # every rank took the same decisions
    assert out0["denoise_rounds"] == rounds and rounds >= 1
    merged = {}
    for _, _, _, _, kc, _, _ in res:
        assert not (set(kc) & set(merged))
        merged.update(kc)
    truth = dict(o.dump())
    # every round's range walk ran over the single table's layout, shard after shard (st.inexact_rounds counts rounds that
    # had to fall back to one walk per shard): the shards together ARE the single filter
    assert res[0][6][4] == 0
    assert merged == truth
    assert (res[0][2], res[0][3]) == (o.ndistinct(), o.nelts())
    assert res[0][2] == sum(r[5] for r in res)           # global count = sum of the shards' counts
    # the stitched .cqf rank 0 wrote (shk_import_shards on the gathered tables): holds exactly the shards' entries, in
    # one canonical table, with the filter-wide counters in its header
    f = cqflibs.oracle().load(out_path)
    assert dict(f.dump()) == merged and f.check_offset()
    assert (f.nelts(), f.ndistinct()) == (res[0][3], res[0][2])
    canon = cqflibs.oracle().new(QB)
    for k, c in sorted(merged.items()):
        canon.insert(k, c)
    assert f.blocks() == canon.blocks()
    for x in (f, canon, o):
        x.free()


def test_stitched_export_equals_single_table_bytes(tmp_path):
    """without deNoise rounds nothing depends on the shards' walks: the stitched file is the single-filter .cqf, byte for byte"""
    out_path = str(tmp_path / "stitched.cqf")
    res, o, rounds = _run(2, 0, out_path)
    assert rounds == 0
    single = str(tmp_path / "single.cqf")
    o.serialize(single)
    assert open(out_path, "rb").read() == open(single, "rb").read()
    o.free()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_one_pass_points_make_the_shards_the_single_table(world, tmp_path, monkeypatch):
    """deNoise points taken in one rebuild per shard (shk/dist.py _one_pass_point): the point's chunk guessed from a sample of
    the regions and verified, the round's range walk (short ranges here: many range ends, some next to a shard border)
    continued from shard to shard over the layout of the single table (with 8 ranks a shard is ONE 256-quotient region). Then NOTHING differs from the single filter: rounds,
    removed counts, counters, and the stitched file is the oracle's .cqf byte for byte"""
    monkeypatch.setenv("SHK_SAMPLE_STRIDE", "2")
    out_path = str(tmp_path / "stitched.cqf")
    res, o, rounds = _run(world, 2, out_path, ml=64, trig=TRIG_ONE_PASS[world], shape=(24, 2))
    out0 = res[0][1]
    assert all(r[1] == out0 for r in res)
    one_pass, other, guesses, right, inexact = res[0][6]
    assert (one_pass, other) == (2, 0) and rounds == out0["denoise_rounds"] == 2
    assert (res[0][2], res[0][3]) == (o.ndistinct(), o.nelts())
    single = str(tmp_path / "single.cqf")
    o.serialize(single)
    assert open(out_path, "rb").read() == open(single, "rb").read()
    o.free()


def _failing_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "sh-assembly_amd"))
    import shk
    from shk import dist as shkdist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    # rank 1's batch holds more k-mers than its context can take (max_batch_keys): shk_hash_chunks fails THERE only
    fq, (offs, lens) = _data(rank, (24, 6) if rank == 0 else (90, 30))
    ctx = shk.Context(qb=QB, k=K, min_denoise_len=ML, max_batch_bytes=1 << 20, max_batch_keys=2048,
                      shard_index=rank, num_shards=world, threads_per_group=64, hash_groups=2, lib_path=EMU)
    got = None
    try:
        ex = shkdist.hash_and_exchange(ctx, fq, offs, lens, QB + 8, world, rank, dev)
        ex.wait()
    except shk.ShkError as e:
        got = e.code
    q.put((rank, got))
    ctx.close()
    dist.destroy_process_group()


def test_a_rank_local_failure_in_front_of_the_exchange_is_raised_on_every_rank():
    """a rank whose hash / route call fails must still enter the exchange's all-gather, and every rank must see
    the failure -- otherwise the peers wait in the collective until the backend's timeout"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    import torch.multiprocessing as mp
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29600 + (os.getpid() * 7 + 11) % 300
    procs = [ctxm.Process(target=_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] is not None and res[0][1] == res[1][1] and res[0][1] < 0, res


@pytest.mark.parametrize("world", [2, 4])
def test_multi_gpu_front_end_reproduces_the_golden_cqf_files(world, tmp_path):
    """python -m shk.count, the multi-GPU CQF-deNoise (parts round-robin over the ranks, all-to-all, collective rebuild and
    deNoise rounds, every rank pwriting its own blocks of the .cqf), on the golden FASTQ files with 2 and 4 gloo ranks on
    the emulator build: the files the REFERENCE build wrote (tests/golden/build0..3.cqf) come out byte for byte -- without
    rounds, with 3 rounds, and with 6 rounds + --endDeNoise at a range length of 1024 slots"""
    import json
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "sh-assembly_amd"), os.path.join(ROOT, "sh-assembly_amd", "libshkhost.so")])
    G = os.path.join(ROOT, "tests", "golden")
    fx = json.load(open(os.path.join(G, "fastq_builds.json")))
    done = 0
    for bi, b in enumerate(fx["builds"][:4]):
        c = b["cfg"]
        if (1 << c["qb"]) // world < 256 or (world == 4 and bi in (0, 3)):     # (4 ranks: the two builds with rounds)
            continue
        lst = tmp_path / ("files%d.txt" % bi)
        lst.write_text("\n".join(os.path.join(G, f) for f in c["files"]) + "\n")     # (absolute names: the prefix rule leaves them alone)
        out = str(tmp_path / ("out%d.cqf" % bi))
        args = ["-k", str(c["k"]), "-n", "6000", "-N", "100000", "-e", "0.01", "-f", "f", "-i", str(lst), "-o", out,
                "--deNoise", str(c["nd"]), "--rounds", str(c["nd"]), "--qb", str(c["qb"]), "--trigger", str(min(c["trigger"], 1 << 62)),
                "--part-size", str(c["ps"]), "--overhead", str(c["ov"]), "--min-denoise-len", str(c["ml"]),
                "--parts-per-call", "2", "--backend", "gloo", "--lib", EMU]
        if c["end"]:
            args.append("--endDeNoise")
        port = 29600 + (os.getpid() * 7 + world * 13 + bi) % 300
        env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "sh-assembly_amd"), SHK_SAMPLE_STRIDE="2")
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                            "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "shk.count"] + args,
                           capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr[-3000:]
        assert open(out, "rb").read() == open(os.path.join(G, b["cqf"]), "rb").read(), (c, r.stderr[-600:])
        done += 1
    assert done >= 2
