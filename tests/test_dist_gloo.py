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


def _data(rank):
    g = synth.make_genome(300, 7)
    fq = synth.make_fastq(g, 24, 90, 0.01, seed=50 + rank, n_frac=0.05)
    return fq, chunks_by_records(fq, 6)


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "sh-assembly_amd"))
    import shk
    from shk import dist as shkdist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    fq, (offs, lens) = _data(rank)
    ctx = shk.Context(qb=QB, k=K, min_denoise_len=ML, max_batch_bytes=1 << 20, max_batch_keys=1 << 16,
                      shard_index=rank, num_shards=world, threads_per_group=64, hash_groups=2, lib_path=EMU)
    st = shkdist.ShardState(TRIG, ND, dev)
    hb = QB + 8
    dp, nw = ctx.hash_chunks(fq, offs, lens)
    recv = shkdist.route_words(ctx, nw, hb, world, rank, dev)
    ctx.stage_words(recv.data_ptr(), recv.numel())
    out = shkdist.sharded_count(ctx, st, len(offs) * world)
    # (key, count) content of this shard through lookups of every key it received
    keys = sorted(set(int(x) & ((1 << hb) - 1) for x in recv.tolist()))
    cnt, _ = ctx.lookup(keys, mode=2)
    t = ctx.totals()
    q.put((rank, out, st.ndistinct, st.nelts, {k: c for k, c in zip(keys, cnt) if c}, t.ndistinct))
    ctx.close()
    dist.destroy_process_group()


def test_two_shards_match_single_filter():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    import torch.multiprocessing as mp
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29600 + os.getpid() % 300
    procs = [ctxm.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    # oracle: one filter, chunks in the interleaved order rank0.c0, rank1.c0, rank0.c1, ...
    O = cqflibs.oracle()
    o = O.new(QB)
    d = [_data(r) for r in range(2)]
    nch = len(d[0][1][0])
    left, rounds, removed = ND, 0, 0
    for j in range(nch):
        for r in range(2):
            fq, (offs, lens) = d[r]
            o.reads_to_kmers(fq[offs[j]:offs[j] + lens[j]], K)
            if left and o.ndistinct() >= TRIG:
                left -= 1
                removed += o.denoise_round(ML)
                rounds += 1
    assert not o.full()
    out0 = res[0][1]
    assert res[0][1] == res[1][1]                       # every rank took the same decisions
    assert out0["denoise_rounds"] == rounds and rounds >= 1
    merged = {}
    for _, _, _, _, kc, _ in res:
        assert not (set(kc) & set(merged))
        merged.update(kc)
    truth = dict(o.dump())
    # every entry the single filter holds with count >= 2 is identical; singletons may differ only by
    # the range-end singletons of the deNoise walk, which restarts per shard
    assert {k: c for k, c in merged.items() if c >= 2} == {k: c for k, c in truth.items() if c >= 2}
    diff = set(k for k, c in merged.items() if c == 1) ^ set(k for k, c in truth.items() if c == 1)
    assert len(diff) <= 2 * rounds * 2
    assert abs(res[0][2] - o.ndistinct()) <= 2 * rounds * 2
    assert res[0][2] == sum(r[5] for r in res)           # global count = sum of the shards' counts
