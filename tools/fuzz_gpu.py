#!/usr/bin/env python3
"""Randomised parity run on the GPU: many small random configurations (filter size, k, read mix, chunking, batching,
deNoise trigger/rounds/min length) through shk_count_chunks against the oracle's t = 1 build. Prints failures."""
import argparse, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sh-assembly_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--sharded", action="store_true", help="drive the collective flow (shk/dist.py) with one rank over RCCL")
    ap.add_argument("--emu", action="store_true", help="run the kernels in the CPU emulator build (tests/emu/libshk_emu.so; one rank over gloo with --sharded)")
    ap.add_argument("--max-qb", type=int, default=17)
    ap.add_argument("--qbs", default="", help="comma-separated filter sizes to draw from (default 11..max-qb)")
    ap.add_argument("--max-reads", type=int, default=4000)
    args = ap.parse_args()
    lib = os.path.join(ROOT, "tests", "emu", "libshk_emu.so") if args.emu else None
    import torch  # noqa: F401  (torch's HIP runtime first)
    import shk, synth
    from fastq_util import chunks_by_records, oracle_t1, oracle_header
    if args.sharded:
        import torch.distributed as dist
        from shk import dist as shkdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("gloo" if args.emu else "nccl", rank=0, world_size=1)
        dev = torch.device("cpu" if args.emu else "cuda:0")
    rnd = random.Random(args.seed)
    bad = skipped = 0
    t0 = time.time()
    for case in range(args.cases):
        qb = rnd.choice([int(x) for x in args.qbs.split(",")] if args.qbs else [q for q in (11, 12, 13, 14, 15, 16, 17) if q <= args.max_qb])
        k = rnd.choice([21, 28, 31, 47, 63, 64, 65, 100])
        L = rnd.choice([max(k + 5, 60), 100, 150])
        cap = int((1 << qb) * 0.45)
        nreads = max(8, min(args.max_reads, cap // max(1, (L - k + 1)) * rnd.choice([1, 2, 3])))
        G = max(200, nreads * L // rnd.choice([8, 20, 40]))
        err = rnd.choice([0.0, 0.003, 0.01, 0.03])
        # keep most big cases inside the table: expected slots ~ genome k-mers (with their counters) + one per erroneous
        # k-mer occurrence; an over-full draw costs the oracle tens of seconds before it reports `full` and is skipped
        while nreads > 64 and 1.3 * min(G, nreads * (L - k + 1)) + nreads * (L - k + 1) * (1 - (1 - err) ** k) > 0.9 * (1 << qb):
            nreads = nreads * 3 // 4
        fq = synth.make_fastq(synth.make_genome(G, rnd.randrange(1 << 30)), nreads, L, err, seed=rnd.randrange(1 << 30),
                              n_frac=rnd.choice([0.0, 0.02, 0.2]), short_frac=rnd.choice([0.0, 0.05]), lower_frac=rnd.choice([0.0, 0.05]),
                              iupac_frac=rnd.choice([0.0, 0.0, 0.1]))
        per = max(1, nreads // rnd.choice([1, 3, 7, 20]))
        offs, lens = chunks_by_records(fq, per)
        nd = rnd.choice([0, 1, 2, 3, 6])
        trig = rnd.choice([cap // 8, cap // 4, cap // 2, cap])
        endd = rnd.random() < 0.3
        ml = rnd.choice([1 << 20, 1 << 10, 64])
        if os.environ.get("FUZZ_VERBOSE"):
            print("case", case, dict(qb=qb, k=k, L=L, nreads=nreads, G=G, err=err, per=per, nd=nd, trig=trig, endd=endd, ml=ml), flush=True)
        q, orounds, oremoved = oracle_t1(fq, offs, lens, k, qb, trig, nd, endd, ml)
        if q.full():
            q.free()
            skipped += 1
            continue
        if os.environ.get("FUZZ_ORACLE_ONLY"):
            q.free()
            continue
        mlb = rnd.choice([0, 0, 2, 3, 4])      # partition levels of 2-4 bits: several levels on small filters
        if mlb and (qb - 8 + mlb - 1) // mlb > 4:
            mlb = (qb - 8 + 3) // 4             # (at most four levels)
        # sampled location of the deNoise point (read at context creation): off, or every 2nd / 4th / 8th region -- on these
        # small tables the guess is often wrong, which is the point
        os.environ["SHK_SAMPLE_STRIDE"] = str(rnd.choice([0, 2, 2, 4, 8]))
        # the rebuild schemes behind the default one (fallbacks and diagnostics) must stay byte-exact too: the library reads
        # these switches when a context is created / a pass is planned
        for v in ("SHK_SINGLE", "SHK_TWO_LAUNCH", "SHK_COARSE_HIST", "SHK_NO_FUSED_POINT", "SHK_RP_NO_GROUPS"):
            os.environ.pop(v, None)
        scheme = rnd.choice([None] * 5 + ["SHK_SINGLE", "SHK_TWO_LAUNCH", "SHK_COARSE_HIST", "SHK_NO_FUSED_POINT", "SHK_RP_NO_GROUPS"])
        if scheme:
            os.environ[scheme] = "1"
        if args.sharded:
            ctx = shk.Context(qb=qb, k=k, min_denoise_len=ml, max_batch_bytes=len(fq) + 1024, max_batch_keys=nreads * L + 64,
                              shard_index=0, num_shards=1, max_level_bits=mlb, lib_path=lib)
            sst = shkdist.ShardState(trig, nd, dev)
        else:
            ctx = shk.Context(qb=qb, k=k, trigger=trig, num_denoise=nd, min_denoise_len=ml, max_batch_bytes=len(fq) + 1024,
                              max_batch_keys=nreads * L + 64, max_level_bits=mlb, lib_path=lib)
        ncalls = rnd.choice([1, 2, 3, len(offs)])
        step = max(1, (len(offs) + ncalls - 1) // ncalls)
        rounds = removed = 0
        try:
            for i in range(0, len(offs), step):
                if args.sharded:
                    if rnd.random() < 0.5:      # hash + bin by owner in one pass (roll kernels) ...
                        if rnd.random() < 0.5:  # ... its own words staying in the send buffer (shk_stage_words_pair reads them there) ...
                            ex = shkdist.hash_and_exchange(ctx, fq, offs[i:i + step], lens[i:i + step], qb + 8, 1, 0, dev, async_op=False, keep_own=True)
                            recv = ex.wait()
                            if ex.own[1]:
                                ctx.stage_words_pair(ex.own[0], ex.own[1], 0, 0)
                            else:
                                ctx.stage_words(0, 0)
                            st = shkdist.sharded_count(ctx, sst, len(offs[i:i + step]))
                            rounds += st["denoise_rounds"]; removed += st["removed"]
                            continue
                        recv = shkdist.hash_and_exchange(ctx, fq, offs[i:i + step], lens[i:i + step], qb + 8, 1, 0, dev, async_op=False).wait()
                    else:                       # ... or the wave-scan hash kernel and the routing pass
                        _, nw = ctx.hash_chunks(fq, offs[i:i + step], lens[i:i + step])
                        recv = shkdist.route_words(ctx, nw, qb + 8, 1, 0, dev)
                    ctx.stage_words(recv.data_ptr(), recv.numel())
                    st = shkdist.sharded_count(ctx, sst, len(offs[i:i + step]))
                else:
                    st = ctx.count_chunks(fq, offs[i:i + step], lens[i:i + step])
                rounds += st["denoise_rounds"]; removed += st["removed"]
            if endd:
                removed += ctx.denoise(); rounds += 1
            t = ctx.totals()
            ok = ((rounds, removed) == (orounds, oremoved) and (t.nelts, t.ndistinct) == (q.nelts(), q.ndistinct())
                  and ctx.blocks() == q.blocks() and ctx.header() == oracle_header(q))
        except shk.ShkError as e:
            ok = False
            print("case", case, "error", e)
        if not ok:
            bad += 1
            print("MISMATCH case", case, scheme, dict(qb=qb, k=k, L=L, nreads=nreads, G=G, err=err, per=per, nd=nd, trig=trig, endd=endd, ml=ml,
                                              ncalls=ncalls, rounds=(rounds, orounds), removed=(removed, oremoved)))
        ctx.close()
        q.free()
    print(f"fuzz: {args.cases} cases ({skipped} skipped: the oracle's table was full), {bad} mismatches, {time.time() - t0:.0f} s")
    if args.sharded:
        dist.destroy_process_group()
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
