"""CQF-deNoise on all GPUs of a node: FASTQ files -> ONE .cqf, the filter sharded by quotient range.

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m shk.count \\
        -k 47 -N 16506371070 -n 119157843 -e 0.00234 -f f -i files.txt -o k47.cqf

(with PYTHONPATH=sh-assembly_amd). The counterpart of the reference's single entry CQF_mt::build_KmerSpectrum + save
(cqf/CQF_mt.h:959-995, 521) behind src/CQF-deNoise.cpp:212-218, one process per GPU:

  * every rank runs the reference's chunker over the file list (host/fastq_chunker.cpp: the same parts, in the same order,
    as fastq_read_parts, cqf/CQF_mt.h:735-816) and keeps the parts c with c mod G = its rank -- the global part order
    j * G + r IS the reference's round-robin file queue, so the deNoise trigger is tested after the same parts as by a
    single context (cqf/CQF_mt.h:837);
  * a batch = `--parts-per-call` parts per rank: hash (shk_hash_chunks), bin by owner and exchange (shk_route_words +
    one all-to-all, the next batch's started before this one is counted), stage, and the collective form of the rebuild
    with its deNoise rounds (shk.dist.sharded_count);
  * the .cqf is written by all ranks, each placing its own blocks (shk.dist.export_cqf): no rank ever holds another's
    table.

The flags and the sizing are those of sh-assembly_amd/bin/CQF-deNoise (src/CQF-deNoise.cpp:18-51, 96-161). Extra:
--backend gloo --lib PATH run the same flow on the CPU emulator build of the kernels (tests)."""
import argparse
import ctypes as C
import os
import sys
import time


def _host():
    L = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libshkhost.so"))
    L.shkh_batch_open.restype = C.c_void_p
    L.shkh_batch_open.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_uint64, C.c_uint32]
    L.shkh_batch_next.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.shkh_batch_close.argtypes = [C.c_void_p]
    L.shkh_free.argtypes = [C.c_void_p]
    L.shkh_size_filter.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_double, C.c_int, C.c_double, C.POINTER(C.c_uint64)]
    L.shkh_size_filter_profile.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_char_p, C.c_int, C.c_double, C.POINTER(C.c_uint64)]
    return L


def parse_args(argv):
    ap = argparse.ArgumentParser(prog="shk.count", description="CQF-deNoise over all ranks of a torch.distributed job")
    ap.add_argument("-k", type=int, required=True, help="k-mer size")
    ap.add_argument("-n", "--trueKmer", type=int, required=True, help="number of unique true k-mers")
    ap.add_argument("-N", type=int, required=True, help="total number of k-mers")
    ap.add_argument("-e", "--alpha", type=float, default=-1.0, help="average base error rate")
    ap.add_argument("--errorProfile", default="", help="error profile file (when -e is not given)")
    ap.add_argument("--fr", type=float, default=0.0, help="tolerable rate of wrongly removed true k-mers (default 1/n)")
    ap.add_argument("--deNoise", type=int, default=-1, help="number of deNoise rounds (default: from the sizing)")
    ap.add_argument("--endDeNoise", action="store_true", help="one more round after the last k-mer")
    ap.add_argument("-t", type=int, default=16, help="kept for compatibility")
    ap.add_argument("-f", "--format", required=True, choices=["g", "b", "f"], help="g(gzip); b(bzip2); f(plain fastq)")
    ap.add_argument("-i", "--input", required=True, help="a file containing the list of read files (relative to its directory)")
    ap.add_argument("-o", "--output", default="", help="output .cqf")
    # hooks, not in the reference (the same as bin/CQF-deNoise has)
    ap.add_argument("--qb", type=int, default=-1)
    ap.add_argument("--trigger", type=int, default=-1)
    ap.add_argument("--rounds", type=int, default=-1)
    ap.add_argument("--part-size", type=int, default=1 << 23)
    ap.add_argument("--overhead", type=int, default=65535)
    ap.add_argument("--min-denoise-len", type=int, default=0)
    ap.add_argument("--parts-per-call", type=int, default=16, help="parts per rank and batch")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--lib", default=None, help="libshk to load (the CPU emulator build with --backend gloo)")
    return ap.parse_args(argv)


def main(argv=None):
    a = parse_args(argv)
    if a.alpha == -1.0 and not a.errorProfile:
        sys.exit("Please specify either <alpha> or <errorProfile>")
    import torch
    import torch.distributed as dist
    import shk
    from shk import dist as shkdist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    if a.backend == "nccl":
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    else:
        device = torch.device("cpu")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    H = _host()
    # sizing as src/CQF-deNoise.cpp:96-161
    out = (C.c_uint64 * 8)()
    if a.alpha != -1.0:
        H.shkh_size_filter(a.k, a.trueKmer, a.N, a.alpha, a.deNoise, a.fr, out)
    else:
        H.shkh_size_filter_profile(a.k, a.trueKmer, a.N, a.errorProfile.encode(), a.deNoise, a.fr, out)
    qb = a.qb if a.qb > 0 else int(out[0])
    rounds = a.rounds if a.rounds >= 0 else int(out[2])
    trigger = a.trigger if a.trigger >= 0 else int(out[3])
    output = a.output or "k%d.t%d.s%d.ser" % (a.k, a.t, qb)
    # the list's entries are relative to its directory (src/CQF-deNoise.cpp:59-81)
    prefix = os.path.dirname(a.input)
    files = [os.path.join(prefix, ln.strip()) if prefix else ln.strip() for ln in open(a.input) if ln.strip()]
    mode = {"f": 0, "g": 1, "b": 2}[a.format]
    if rank == 0:
        print("CQF-deNoise settings (x%d ranks):\nqb: %d\nhb: %d\nK: %d\nnumber of true k-mers: %d\nnumber of deNoise rounds: %d\n"
              "deNoise after processing all k-mers: %s\nnumber of unique k-mers triggering deNoise: %d\n"
              % (world, qb, qb + 8, a.k, a.trueKmer, rounds, "true" if a.endDeNoise else "false", trigger), file=sys.stderr, flush=True)
    B = max(1, min(a.parts_per_call, shk.MAX_CHUNKS // world - 1))
    cap_bytes = B * (a.part_size + a.overhead + 64)
    ctx = shk.Context(qb=qb, k=a.k, min_denoise_len=a.min_denoise_len, max_batch_bytes=cap_bytes,
                      max_batch_keys=max(1 << 16, int(1.5 * cap_bytes)), max_batch_reads=cap_bytes // 16 + 1024,
                      device=local_rank if a.backend == "nccl" else 0, shard_index=rank, num_shards=world, lib_path=a.lib,
                      **({"threads_per_group": 64, "hash_groups": 2} if a.backend == "gloo" else {}))
    shkdist.reserve_exchange(ctx, device)
    st = shkdist.ShardState(trigger, rounds, device)
    arr = (C.c_char_p * len(files))(*[f.encode() for f in files])
    bh = H.shkh_batch_open(arr, len(files), mode, a.part_size, a.overhead)
    t0 = time.time()
    totals = {"kmers": 0, "new_distinct": 0, "removed": 0, "denoise_rounds": 0}

    def next_batch(first_global):
        """this rank's parts among the next B * world parts of the global order; (text, offs, lens, parts seen, error)"""
        text, offs, lens, seen, err = bytearray(), [], [], 0, 0
        for c in range(first_global, first_global + B * world):
            p, n = C.c_void_p(), C.c_uint64()
            r = H.shkh_batch_next(bh, C.byref(p), C.byref(n))
            if r <= 0:
                err = -1 if r < 0 else 0
                break
            seen += 1
            if c % world == rank:
                offs.append(len(text))
                lens.append(n.value)
                text += C.string_at(p.value, n.value)
            H.shkh_free(p)
        return bytes(text), offs, lens, seen, err

    def start(first_global):
        text, offs, lens, seen, err = next_batch(first_global)
        if err:                                    # SHK_ERR_IO: "Error: Wrong input file!"
            return shkdist.Exchange(ctx, 0, qb + 8, world, rank, device, local_rc=-6, keep_own=True), seen
        # hash + bin by owner + start the exchange (a rank without a part in the last, ragged batch takes part with no
        # words; a local failure is raised on every rank)
        return shkdist.hash_and_exchange(ctx, text, offs, lens, qb + 8, world, rank, device, keep_own=True), seen

    hb = qb + 8
    first = 0
    ex, seen = start(first)
    while True:
        # the number of parts in this batch is the same on every rank: they all walk the same files
        nxt = None
        if seen == B * world:
            nxt = start(first + seen)              # the next batch's exchange runs while this one is counted
        if seen:
            recv = ex.wait()
            shkdist.stage_received(ctx, st, recv, own=ex.own)
            o = shkdist.sharded_count(ctx, st, seen)
            for kk in totals:
                totals[kk] += o[kk]
        else:
            ex.wait()                              # (an empty exchange: the parts were a whole number of batches)
        if nxt is None:
            break
        first += seen
        ex, seen = nxt
    H.shkh_batch_close(bh)
    if a.endDeNoise:
        removed = shkdist.sharded_denoise(ctx, st)
        totals["removed"] += removed
        totals["denoise_rounds"] += 1
    shkdist.check(ctx, st)
    shkdist.export_cqf(ctx, st, output, world, rank, device, qb, a.k)
    if rank == 0:
        print("Finished building K-mer spectrum!\nnelts: %d ndistinct_elts: %d deNoise rounds: %d removed: %d\n"
              "Time for building K-mer spectrum: %.0f seconds." % (st.nelts, st.ndistinct, totals["denoise_rounds"], totals["removed"],
                                                                  time.time() - t0), file=sys.stderr, flush=True)
    ctx.close()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
