#!/usr/bin/env python3
"""bench.py -- k-mers counted per second into the CQF on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (shk_count_chunks: FASTQ text resident in HBM ->
ntHash keys -> region partition -> CQF rebuild, deNoise rounds included where the
reference's t = 1 schedule fires them) over one batch of synthetic reads.
Workload at N = 1: BASELINE.json configs[1] scaled to a few steps -- C. elegans-like
100 Mbp uniform-random genome, 150 bp reads, e = 0.00234, k = 47, filter sized exactly as
the README example (qb = 29, hb = 37, 0.70 GiB table, trigger ~3.1e8 distinct k-mers).

N > 1 (driver launches via torch.distributed.run): the filter is sharded by quotient
range, every rank hashes its own batch, bins the key words by owner and exchanges them
with one RCCL all-to-all before its local insert. Weak scaling: reads per rank fixed.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "sh-assembly_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALGO_BYTES_PER_KMER = 179.0       # 1 B input base + read and write of one 89-B block
HBM_PEAK = 8.0e12                 # MI355X_MICROARCH.md: 8 TB/s spec
NAME_W = 11                       # digits in the read name
PART = 1 << 23                    # the reference's part size (CQF_mt.h:743)
OVERHEAD = 65535                  # CQF_mt.h:742


def sizing(K, n_true, N_total, alpha, fr=0.0):
    """src/CQF-deNoise.cpp:96-161 (host arithmetic; Poisson CDF from scipy instead of boost)"""
    from scipy.stats import poisson
    num_true = int(N_total * (1 - alpha) ** K)
    num_false = N_total - num_true
    if not fr:
        fr = 1.0 / n_true
    mean = float(num_true // n_true)
    cdf0 = poisson.cdf(0, mean)

    def cdfp(x):
        return (poisson.cdf(x, mean) - cdf0) / (1 - cdf0)
    start, end = 0, int(mean + 1)
    while cdfp(end) < fr:
        end *= 2
    nd = None
    while start <= end:
        if start == end:
            nd = start
            break
        if start + 1 == end:
            t1, t2 = cdfp(start), cdfp(end)
            nd = end if t2 <= fr else (start if t1 <= fr else max(start - 1, 0))
            break
        mid = (start + end) // 2
        c = cdfp(mid)
        if c < fr:
            start = mid + 1
        elif c > fr:
            end = mid - 1
        else:
            nd = start
            break
    if nd is None:
        nd = start
    enc, tmp = 0, num_true // n_true + 1
    while tmp:
        tmp >>= 7
        enc += 1

    def nslots(d):
        return int(n_true * (enc + 1.5) + num_false * 10 // ((d + 1) * 9))
    num_slots = nslots(nd)
    qb, base = 1, 2
    while base < num_slots:
        qb += 1
        base <<= 1
    st = num_slots
    while nd and st < (1 << qb):
        nd -= 1
        st = nslots(nd)
    if st >= (1 << qb):
        nd += 1
    trigger = n_true + num_false // (nd + 1)
    return qb, nd, trigger


def chunk_table(nrec, rec, part=PART, overhead=OVERHEAD):
    """fastq_read_parts (CQF_mt.h:735-816) on a stream of fixed-size records: a part is cut
    at the first record start behind (bytes so far - overhead/2); the tail is carried over.
    Checked against the oracle's chunker in tests/test_host_logic.py."""
    total_bytes = nrec * rec
    offs, lens, pos, carry, rp, eof = [], [], 0, 0, 0, False
    while not eof:
        readed = min(part, total_bytes - rp)
        rp += readed
        eof = readed < part          # feof() is raised only by a short read
        total = carry + readed
        if eof:
            if total:
                offs.append(pos)
                lens.append(total)
            break
        i = total - overhead // 2
        cut = (i // rec + 1) * rec   # first header line that starts behind position i
        offs.append(pos)
        lens.append(cut)
        pos += cut
        carry = total - cut
    return offs, lens


def gen_batch_torch(torch, genome, nreads, L, err, first_id, seed, device):
    """synthetic FASTQ text on the device: [nreads, 2L+NAME_W+6] bytes"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    G = genome.numel()
    pos = torch.randint(0, G - L + 1, (nreads,), generator=g, device=device)
    idx = pos[:, None] + torch.arange(L, device=device)[None, :]
    seq = genome[idx]                                          # codes 0..3
    strand = torch.randint(0, 2, (nreads, 1), generator=g, device=device, dtype=torch.uint8)
    seq = torch.where(strand.bool(), (3 - seq).flip(1), seq)   # reverse complement
    e = torch.rand((nreads, L), generator=g, device=device) < err
    sub = torch.randint(1, 4, (nreads, L), generator=g, device=device, dtype=torch.uint8)
    seq = torch.where(e, (seq + sub) % 4, seq)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    bases = lut[seq.long()]
    nmask = torch.rand((nreads, 1), generator=g, device=device) < 0.001   # 0.1 % of reads carry an N
    npos = torch.randint(0, L, (nreads, 1), generator=g, device=device)
    ar = torch.arange(L, device=device)[None, :]
    bases = torch.where(nmask & (ar >= npos) & (ar < npos + 2), torch.tensor(ord("N"), dtype=torch.uint8, device=device), bases)
    rec = 2 * L + NAME_W + 6
    out = torch.empty((nreads, rec), dtype=torch.uint8, device=device)
    out[:, 0] = ord("@")
    ids = torch.arange(first_id, first_id + nreads, device=device, dtype=torch.int64)
    for d in range(NAME_W):
        out[:, 1 + d] = ((ids // (10 ** (NAME_W - 1 - d))) % 10 + ord("0")).to(torch.uint8)
    out[:, 1 + NAME_W] = 10
    out[:, 2 + NAME_W:2 + NAME_W + L] = bases
    o = 2 + NAME_W + L
    out[:, o] = 10
    out[:, o + 1] = ord("+")
    out[:, o + 2] = 10
    out[:, o + 3:o + 3 + L] = ord("I")
    out[:, o + 3 + L] = 10
    return out.reshape(-1)


def cpu_baseline(torch, text_cpu, offs, lens, k, qb, budget_s=12.0):
    """reference gqf.c + nthash.hpp (oracle/_ref, kind 'reference') or the C restatement (kind 'port') on the
    host cores, over a bounded prefix of the same FASTQ batch: first one thread (the t = 1 path the parity tests
    pin), then -- with the reference -- as many threads as the box gives this job (at most 16, the reference's
    default -t), all inserting into one filter under the reference's region locks"""
    import cqflibs
    kind = "reference" if cqflibs.have_ref() else "port"
    lib = cqflibs.ref() if kind == "reference" else cqflibs.oracle()
    q = lib.new(qb)
    t0 = time.time()
    used = 0
    for a, n in zip(offs, lens):
        q.reads_to_kmers(text_cpu[a:a + n], k)
        used += 1
        if time.time() - t0 > budget_s:
            break
    dt = time.time() - t0
    kmers = q.nelts()
    q.free()
    out = {"value": kmers / dt, "unit": "k-mers/s", "cores": 1, "kind": kind,
           "sample": f"first {used} chunk(s) of one bench batch: {kmers} k-mers inserted into an empty qb={qb} filter in {dt:.1f} s (t=1)"}
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    nt = max(1, min(16, ncpu))
    if kind == "reference" and nt > 1 and hasattr(lib.L, "ref_time_chunks_mt"):
        q = lib.new(qb)
        dtm, km, ch = q.time_chunks_mt(text_cpu, offs, lens, k, nt, budget_s)
        q.free()
        out = {"value": km / dtm, "unit": "k-mers/s", "cores": nt, "kind": kind,
               "sample": f"first {ch} chunk(s) of one bench batch: {km} k-mers inserted into an empty qb={qb} filter in "
                         f"{dtm:.1f} s by {nt} threads under the reference's region locks (CQF_mt -t {nt})",
               "single_core_value": kmers / dt}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads-per-step", type=int, default=8_000_000)
    ap.add_argument("--genome", type=int, default=119_157_843, help="synthetic genome length = distinct true k-mers n (README.md:91)")
    ap.add_argument("--qb", type=int, default=0, help="override the filter size (default: README sizing)")
    ap.add_argument("--threads", type=int, default=0, help="threads per workgroup (0 = library default)")
    ap.add_argument("--ablate", type=int, default=0, help="diagnostics: SHK_ABLATE bits applied to the timed steps only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="run the sharded/all-to-all code path even with one rank")
    ap.add_argument("--full-build", action="store_true", help="the whole C. elegans-sized build (README.md:90-91: 16.5 G k-mers = 20 steps, no warm-up); adds build_time_s; the filter is sized with 5 %% head room on N (the README numbers sit exactly at the edge of 8 rounds)")
    ap.add_argument("--host-text", action="store_true", help="hand the FASTQ text over in (pinned) host memory: PCIe-inclusive rate, never the headline value")
    args = ap.parse_args()
    if args.full_build:
        args.steps, args.warmup = 20, 0
    default_workload = (args.reads_per_step == 8_000_000 and args.genome == 119_157_843 and args.qb == 0 and not args.ablate
                        and args.gpus == 1 and not args.host_text and not args.full_build)

    import torch
    import shk
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus must equal WORLD_SIZE")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    sharded = world > 1 or args.force_dist
    if sharded:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group("nccl", device_id=device)

    K, L, ERR = 47, 150, 0.00234
    N_README, n_README = 16506371070, 119157843            # README.md:90-91
    R = args.reads_per_step
    # README sizing; when the planned run presents more k-mers than the README's N (weak
    # scaling over several GPUs) the same formulas are applied to the planned total
    # Weak scaling: with G GPUs the data set is G times larger (G x the genome, G x the distinct true
    # k-mers, G x the reads per step) and so is the filter, so every GPU keeps a C. elegans-sized shard.
    N_plan = max(N_README * world, world * (args.steps + args.warmup) * R * (L - K + 1))
    if args.full_build:
        N_plan = int(N_plan * 1.05)
    qb, nd, trigger = sizing(K, n_README * world, N_plan, ERR)
    if args.qb:
        qb = args.qb
    rec = 2 * L + NAME_W + 6
    kmers_per_read = L - K + 1
    offs, lens = chunk_table(R, rec)
    assert len(offs) <= shk.MAX_CHUNKS

    ctx = shk.Context(qb=qb, k=K, trigger=(trigger if not sharded else (1 << 62)), num_denoise=(nd if not sharded else 0),
                      max_batch_bytes=(R * (2 * L + NAME_W + 6) + 4096 if args.host_text else 64), max_batch_keys=int(R * kmers_per_read * (1.5 if sharded else 1.0)) + 4096,
                      max_batch_reads=R + 1024, threads_per_group=args.threads, device=local_rank, shard_index=rank, num_shards=world)
    tot = ctx.totals()

    genome = torch.randint(0, 4, (args.genome * world,), device=device, dtype=torch.uint8,
                           generator=torch.Generator(device=device).manual_seed(2))
    nsteps = args.steps + args.warmup
    texts = [gen_batch_torch(torch, genome, R, L, ERR, (s * world + rank) * R, 1000 + s * world + rank, device)
             for s in range(nsteps)]
    torch.cuda.synchronize()
    if args.host_text:
        texts = [t.cpu().pin_memory() for t in texts]

    hb = qb + 8
    from shk import dist as shkdist
    sstate = shkdist.ShardState(trigger, nd, device) if sharded else None
    counted_global = [0]
    uploaded = {}
    shard_shift = (qb - int(math.log2(world))) + 8
    rounds_left = nd
    counted = 0
    removed_total = 0
    rounds_fired = 0

    def step(s):
        nonlocal rounds_left, counted, removed_total, rounds_fired
        t = texts[s]
        if not sharded:
            if args.host_text:
                # overlapped ingest: this batch's copy was started before the previous batch was counted
                if s not in uploaded:
                    uploaded[s] = ctx.upload_text(t.data_ptr(), t.numel())
                dptr = uploaded.pop(s)
                if s + 1 < nsteps:
                    uploaded[s + 1] = ctx.upload_text(texts[s + 1].data_ptr(), texts[s + 1].numel())
                st = ctx.count_chunks(dptr, offs, lens, on_device=True, text_bytes=t.numel())
            else:
                st = ctx.count_chunks(t.data_ptr(), offs, lens, on_device=True, text_bytes=t.numel())
            counted += st["kmers"]
            removed_total += st["removed"]
            rounds_fired += st["denoise_rounds"]
            return
        dp, nw = ctx.hash_chunks(t.data_ptr(), offs, lens, on_device=True, text_bytes=t.numel())
        recv = shkdist.route_words(ctx, nw, hb, world, rank, device)
        torch.cuda.synchronize()
        ctx.stage_words(recv.data_ptr(), recv.numel())
        r = shkdist.sharded_count(ctx, sstate, len(offs) * world)
        counted_global[0] += r["kmers"]
        removed_total += r["removed"]
        rounds_fired += r["denoise_rounds"]

    for s in range(args.warmup):
        step(s)
    counted = 0
    counted_global[0] = 0
    removed_total = 0
    rounds_fired = 0
    if args.ablate:
        os.environ["SHK_ABLATE"] = str(args.ablate)
    ctx.profile(True)
    ctx.profile_reset()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.warmup, nsteps):
        step(s)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    prof = ctx.profile_get()
    if dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        counted = counted_global[0]      # already the whole-job count (all-reduced inside sharded_count)

    if rank == 0:
        # dominant kernel by accumulated device time
        name, (launches, ms) = max(prof.items(), key=lambda kv: kv[1][1]) if prof else ("none", (1, 1.0))
        avg_s = ms / 1e3 / max(launches, 1)
        per_rank_kmers = counted / world
        units_per_launch = per_rank_kmers / max(launches, 1)
        achieved = ALGO_BYTES_PER_KMER * units_per_launch / avg_s / 1e9
        # HBM traffic of the dominant kernel: PMC counters cannot run inside the timed loop; the number comes
        # from the committed separate --pmc passes over this same default workload (null for any other workload)
        traffic, traffic_src = None, None
        try:
            pt = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")))
            pk = {"k_region_merge<spill>": "k_region_merge<3, 24>", "k_region_place": "k_region_place<24>"}.get(name, name)
            if default_workload and pk in pt["kernels"]:
                traffic = pt["kernels"][pk]["fetch_bytes_per_launch"] + pt["kernels"][pk]["write_bytes_per_launch"]
                traffic_src = pt["source"]
        except (OSError, ValueError, KeyError):
            pass
        kern_ms = {k: round(v[1], 3) for k, v in prof.items()}
        kern_n = {k: int(v[0]) for k, v in prof.items()}
        table_bytes = tot.table_bytes
        path_bytes = ALGO_BYTES_PER_KMER * per_rank_kmers + rounds_fired * 2 * table_bytes
        out = {
            "metric": "k-mers counted/sec (whole node), CQF build, C.elegans-like k=47",
            "value": counted / dt, "unit": "k-mers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "C.elegans-like synthetic reads (G=%d x n_gpus, L=150, e=0.00234), k=47, CQF qb=%d hb=%d "
                                   "(README.md:98 sizing, N=%d), %d reads/step/GPU, 8 MiB chunks, deNoise rounds=%d trigger=%d"
                                   % (args.genome, qb, hb, N_plan, R, nd, trigger),
                       "kmers_per_step_per_gpu": R * kmers_per_read, "denoise_rounds_fired": rounds_fired,
                       "removed": removed_total, "parallelism": "quotient-range shards x%d" % world},
            "roofline": {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / (HBM_PEAK / 1e9), "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src,
                         "launches": launches, "avg_launch_ms": avg_s * 1e3,
                         "algorithmic_bytes_per_kmer": ALGO_BYTES_PER_KMER},
            "roofline_path": {"achieved": path_bytes / dt / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                              "frac": path_bytes / dt / HBM_PEAK,
                              "formula": "(kmers*179 + rounds*2*table_bytes) / t / 8e12 (SURVEY.md 8d)"},
            "kernel_ms": kern_ms, "kernel_launches": kern_n,
            **({"build_time_s": dt, "build_kmers": counted, "build_rounds": rounds_fired} if args.full_build else {}),
        }
        if not args.no_cpu_baseline and world == 1:   # reported on rank 0 at N = 1 only
            text_cpu = texts[args.warmup].cpu().numpy().tobytes()
            out["cpu_baseline"] = cpu_baseline(torch, text_cpu, offs, lens, K, qb)
        print(json.dumps(out))
    ctx.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
