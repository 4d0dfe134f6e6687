#!/usr/bin/env python3
"""bench.py -- k-mers counted per second into the CQF on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (shk_count_chunks: FASTQ text resident in HBM ->
ntHash keys -> region partition -> CQF rebuild, deNoise rounds included where the
reference's t = 1 schedule fires them) over one batch of synthetic reads.
The timed region is ONE CLEAN BUILD: `--steps` batches into an empty filter, so
`build_time_s` is the BASELINE "CQF build time" for steps x 832 M k-mers (20 steps =
16.6 G k-mers = the README's C. elegans data set, BASELINE.json configs[1]); warm-up
steps run on a scratch filter of the same geometry. The build's command line follows the
README recipe (-N = the k-mers presented, -n = the genome's k-mers, -e = the generator's
error rate, sizing as src/CQF-deNoise.cpp:96-161) and is checked against a model of the
filter's occupancy before it runs (sh-assembly_amd/shk/plan.py, tests/test_plan.py).

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
    """src/CQF-deNoise.cpp:96-161 (host arithmetic; Poisson CDF from scipy instead of boost): shk/plan.py"""
    from shk import plan
    return plan.sizing(K, n_true, N_total, alpha, fr)


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
    # err: one rate, or a tensor of L per-position rates (an --errorProfile-like read model)
    e = torch.rand((nreads, L), generator=g, device=device) < (err if isinstance(err, float) else err.to(device)[None, :])
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


def csrc_sha256():
    """hash of the kernel sources: the committed PMC traffic figures are only quoted for the code they were measured on"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "sh-assembly_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def secondary_host_text(torch, shk, new_ctx, texts, offs, lens, steps=4):
    """PCIe-inclusive rate: the same batches handed over in pinned HOST memory, uploads overlapped with the previous
    batch's counting (shk_upload_text). Never the headline value. A scratch filter, `steps` batches."""
    n = min(steps, len(texts))
    host = [t.cpu().pin_memory() for t in texts[:n]]
    ctx = new_ctx(host_text_bytes=int(host[0].numel()) + 4096)
    up = {0: ctx.upload_text(host[0].data_ptr(), host[0].numel())}
    kmers = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(n):
        dptr = up.pop(s)
        if s + 1 < n:
            up[s + 1] = ctx.upload_text(host[s + 1].data_ptr(), host[s + 1].numel())
        kmers += ctx.count_chunks(dptr, offs, lens, on_device=True, text_bytes=host[s].numel())["kmers"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ctx.close()
    return {"what": "same batches from pinned host memory, upload of batch s+1 overlapped with counting batch s (PCIe-inclusive; never `value`)",
            "steps": n, "value": kmers / dt, "unit": "k-mers/s", "ms_per_step": dt / n * 1e3}


def secondary_contiger(torch, shk, ctx, text, offs, lens, k):
    """Contiger (BASELINE config 3) on the filter the timed build has just produced, over the reads of one batch: seeds
    from the reads, walks on the device, then duplicate removal, numbering, links and the FASTA text."""
    import tempfile
    u = shk.UnitigSet(ctx)
    ctx.profile_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nseeds = 0
    B = 64
    for a in range(0, len(offs), B):
        nseeds += u.add_reads(text.data_ptr(), offs[a:a + B], lens[a:a + B], k, 2, 2, 1000000, 1 << 26, text_bytes=int(text.numel()))
    torch.cuda.synchronize()
    t_walk = time.perf_counter() - t0
    walk_ms = ctx.profile_get().get("k_ug_walk", (0, 0.0))[1]
    out = os.path.join(tempfile.mkdtemp(), "unitigs.fa")
    t0 = time.perf_counter()
    st = u.write(k, out)
    t_write = time.perf_counter() - t0
    fin_ms = ctx.profile_get().get("k_ug_check/emit/median/links", (0, 0.0))[1]
    u.close()
    try:
        os.remove(out)
    except OSError:
        pass
    ext = st["extensions"]
    return {"what": "Contiger -s 2 -x 2 on the filter just built, seeds from one batch of reads (8 M), unitigs.fa written",
            "seeds": nseeds, "unitigs": st["unitigs"], "total_len": st["total_len"], "walk_wall_s": t_walk,
            "walk_kernel_s": walk_ms / 1e3, "walk_rounds": st.get("rounds"), "extended_bases": ext,
            "bases_per_s_walk_kernel": ext / (walk_ms / 1e3) if walk_ms else None,
            "lookups_per_s_walk_kernel": 7 * ext / (walk_ms / 1e3) if walk_ms else None,
            "GBps_at_97B_per_lookup": 97 * 7 * ext / (walk_ms / 1e3) / 1e9 if walk_ms else None,
            "finish_and_write_s": t_write, "finish_kernels_s": fin_ms / 1e3}


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
    ap.add_argument("--genome", type=int, default=119_157_843, help="synthetic genome length (its G-K+1 k-mers are the distinct true k-mers, README.md:91)")
    ap.add_argument("--qb", type=int, default=0, help="override the filter size (default: the reference's sizing)")
    ap.add_argument("--max-load", type=float, default=0.95, help="predicted peak load above which deNoise rounds are added to the formula's (shk/plan.py)")
    ap.add_argument("--threads", type=int, default=0, help="threads per workgroup (0 = library default)")
    ap.add_argument("--ablate", type=int, default=0, help="diagnostics: SHK_ABLATE bits applied to the timed steps only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial", action="store_true", help="one batch after the other on one stream (shk_count_chunks) instead of the overlapped front end")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements (PCIe-inclusive rate, Contiger on the built filter)")
    ap.add_argument("--trace", action="store_true", help="diagnostics: print the filter's counters after every step (adds a sync per step)")
    ap.add_argument("--force-dist", action="store_true", help="run the sharded/all-to-all code path even with one rank")
    ap.add_argument("--full-build", action="store_true", help="the whole C. elegans-sized build (README.md:90-91: 16.5 G k-mers = 20 steps, no warm-up)")
    ap.add_argument("--host-text", action="store_true", help="hand the FASTQ text over in (pinned) host memory: PCIe-inclusive rate, never the headline value")
    args = ap.parse_args()
    if args.full_build:
        args.steps, args.warmup = 20, 0
    default_workload = (args.reads_per_step == 8_000_000 and args.genome == 119_157_843 and args.qb == 0 and not args.ablate
                        and args.gpus == 1 and not args.host_text)

    # stdout carries exactly ONE line (the JSON): libraries that chat on file descriptor 1 (RCCL prints a version
    # banner there when a process group starts) are sent to stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import shk
    from shk import plan
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
    R = args.reads_per_step
    rec = 2 * L + NAME_W + 6
    kmers_per_read = L - K + 1
    offs, lens = chunk_table(R, rec)
    assert len(offs) * world <= shk.MAX_CHUNKS
    # The timed region is ONE CLEAN BUILD: `steps` batches into an empty filter (warm-up runs on a scratch
    # filter of the same geometry). Its command line follows the README recipe (README.md:84-98): -N = the
    # k-mers the build presents, -n = the genome's k-mers, -e = the generator's error rate; sizing as
    # src/CQF-deNoise.cpp:96-161; rounds added when the predicted peak load exceeds --max-load (shk/plan.py).
    # Weak scaling: with G GPUs the data set is G times larger (G x the genome, G x the reads per step) and
    # so is the filter: every GPU keeps a C. elegans-sized shard.
    G_total = args.genome * world
    pl = plan.plan_build(K, G_total, L, ERR, R * kmers_per_read * world / len(offs), len(offs) * args.steps,
                         world=world, max_load=args.max_load)
    qb, nd, trigger = pl["qb"], pl["rounds"], pl["trigger"]
    if args.qb:
        qb = args.qb

    def new_ctx(host_text_bytes=0):
        return shk.Context(qb=qb, k=K, trigger=(trigger if not sharded else (1 << 62)), num_denoise=(nd if not sharded else 0),
                           max_batch_bytes=(R * rec + 4096 if args.host_text else max(64, host_text_bytes)),
                           max_batch_keys=int(R * kmers_per_read * (1.5 if sharded else 1.0)) + 4096,
                           max_batch_reads=R + 1024, threads_per_group=args.threads, device=local_rank, shard_index=rank, num_shards=world)

    genome = torch.randint(0, 4, (G_total,), device=device, dtype=torch.uint8,
                           generator=torch.Generator(device=device).manual_seed(2))
    texts = [gen_batch_torch(torch, genome, R, L, ERR, (s * world + rank) * R, 1000 + s * world + rank, device)
             for s in range(args.steps)]
    torch.cuda.synchronize()
    if args.host_text:
        texts = [t.cpu().pin_memory() for t in texts]

    hb = qb + 8
    from shk import dist as shkdist

    class Run:
        """one filter being built: the context plus the counters of the run"""

        def __init__(self):
            self.ctx = new_ctx()
            if not sharded and not args.serial and not args.host_text:
                self.ctx.prepare_reserve()       # (the overlapped front end's buffers: context set-up, like the table's own)
            if sharded:                          # (the exchange's buffers too: two send buffers in the library, two receive tensors)
                shkdist.reserve_exchange(self.ctx, device)
            self.sstate = shkdist.ShardState(trigger, nd, device) if sharded else None
            self.counted = self.removed = self.rounds = 0
            self.uploaded = {}
            self.inflight = {}
            self.prepared = set()

        def exchange(self, s):
            t = texts[s]
            # (a rank-local failure of the hash or the routing is raised on every rank after the exchange's all-gather)
            return shkdist.hash_and_exchange(self.ctx, t.data_ptr(), offs, lens, hb, world, rank, device, on_device=True, text_bytes=t.numel(),
                                             keep_own=True)

        def step(self, s, nsteps):
            ctx, t = self.ctx, texts[s]
            if not sharded:
                if args.host_text:
                    # overlapped ingest: this batch's copy was started before the previous batch was counted
                    if s not in self.uploaded:
                        self.uploaded[s] = ctx.upload_text(t.data_ptr(), t.numel())
                    dptr = self.uploaded.pop(s)
                    if s + 1 < nsteps:
                        self.uploaded[s + 1] = ctx.upload_text(texts[s + 1].data_ptr(), texts[s + 1].numel())
                    st = ctx.count_chunks(dptr, offs, lens, on_device=True, text_bytes=t.numel())
                elif args.serial:
                    st = ctx.count_chunks(t.data_ptr(), offs, lens, on_device=True, text_bytes=t.numel())
                else:
                    # overlapped: the front end (parse, hash, partition) of batch s+1 runs on the context's second
                    # stream while batch s is rebuilt into the table (shk_prepare_chunks / shk_count_prepared)
                    if s not in self.prepared:
                        ctx.prepare_chunks(t.data_ptr(), offs, lens, on_device=True, text_bytes=t.numel())
                        self.prepared.add(s)
                    if s + 1 < nsteps:
                        t2 = texts[s + 1]
                        ctx.prepare_chunks(t2.data_ptr(), offs, lens, on_device=True, text_bytes=t2.numel())
                        self.prepared.add(s + 1)
                    st = ctx.count_prepared()
                    self.prepared.discard(s)
            else:
                # pipelined: the all-to-all of batch s was started one step ago; before waiting for it, batch s+1 is
                # hashed, binned by owner (second send buffer) and ITS exchange started -- xGMI moves key words while
                # the CUs stage and insert batch s
                tm = [time.perf_counter()]

                def lap():          # diagnostics (--trace): where a sharded step spends its time
                    if args.trace:
                        if not os.environ.get("SHK_TRACE_NOSYNC"):
                            torch.cuda.synchronize()
                        tm.append(time.perf_counter())
                    elif os.environ.get("SHK_DIST_TIMING"):
                        tm.append(time.perf_counter())
                        if len(tm) == 5:
                            ph = getattr(self, "phase_s", [0.0] * 4)
                            self.phase_s = [ph[i] + tm[i + 1] - tm[i] for i in range(4)]
                if s not in self.inflight:
                    self.inflight[s] = self.exchange(s)
                ex = self.inflight.pop(s)
                if s + 1 < nsteps:
                    self.inflight[s + 1] = self.exchange(s + 1)
                lap()
                recv = ex.wait()
                lap()
                shkdist.stage_received(ctx, self.sstate, recv, own=ex.own)     # (a local failure surfaces on every rank in sharded_count)
                lap()
                st = shkdist.sharded_count(ctx, self.sstate, len(offs) * world)   # counts are whole-job (all-reduced)
                lap()
                if args.trace and rank == 0:
                    print("step %d: hash+route+start exchange %.1f ms, wait %.1f ms, stage %.1f ms, count %.1f ms (%d collectives so far)" % (
                        (s,) + tuple(1e3 * (tm[i + 1] - tm[i]) for i in range(4)) + (self.sstate.collectives,)), file=sys.stderr, flush=True)
                if s + 1 == nsteps:
                    shkdist.check(ctx, self.sstate)
            self.counted += st["kmers"]
            self.removed += st["removed"]
            self.rounds += st["denoise_rounds"]
            if args.trace and rank == 0:
                tt = ctx.totals()
                now = time.perf_counter()
                print("step %d: %.1f ms since the last one; nelts %d ndistinct %d rounds %d free_pointer %d / %d" % (
                    s, 1e3 * (now - getattr(self, "t_last", now)), tt.nelts, tt.ndistinct, self.rounds, tt.free_pointer, tt.xnslots), file=sys.stderr, flush=True)
                self.t_last = now

    if args.warmup:
        # untimed passes over the first batches into a scratch filter (same geometry, same schedule): loads the code
        # objects, sizes the lazily allocated buffers' pools, warms RCCL
        w = Run()
        w.ctx.profile(True)     # (the timed run records HIP events per kernel: let the runtime grow its event / signal pools here)
        for s in range(args.warmup):
            w.step(s % args.steps, args.steps)
        for ex in w.inflight.values():     # (the exchange a pipelined warm-up step started for a batch it never counted)
            ex.wait()
        torch.cuda.synchronize()
        w.ctx.close()
        del w
    run = Run()
    ctx = run.ctx
    tot = ctx.totals()
    if args.ablate:
        os.environ["SHK_ABLATE"] = str(args.ablate)
    ctx.profile(True)
    ctx.profile_reset()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    if os.environ.get("SHK_DIST_STACKS"):      # diagnostics: where the threads are, every 100 ms of the timed region
        import faulthandler
        import threading
        print("threads:", [t.name for t in threading.enumerate()], file=sys.stderr, flush=True)
        faulthandler.dump_traceback_later(0.1, repeat=True, file=sys.stderr)
    t0 = time.perf_counter()
    for s in range(args.steps):
        run.step(s, args.steps)
    torch.cuda.synchronize()
    if os.environ.get("SHK_DIST_STACKS"):
        faulthandler.cancel_dump_traceback_later()
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if sharded and rank == 0 and os.environ.get("SHK_DIST_TIMING"):
        print("small collectives: %d, %.1f ms in all (build %.1f ms); phases hash+route+exchange / wait / stage / count: %s ms" % (
            shkdist.COLLECTIVE_SECONDS[1], 1e3 * shkdist.COLLECTIVE_SECONDS[0], 1e3 * dt, ["%.0f" % (1e3 * x) for x in getattr(run, "phase_s", [])] + ["lib %.0f" % (1e3 * shkdist.HASH_EXCHANGE_SECONDS[0]), "exch %.0f" % (1e3 * shkdist.HASH_EXCHANGE_SECONDS[1])]),
              file=sys.stderr, flush=True)
    prof = ctx.profile_get()
    end = ctx.totals()
    if dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    counted, removed_total, rounds_fired = run.counted, run.removed, run.rounds

    # In the overlapped build the front end of batch s+1 and the rebuild of batch s share the chip: a kernel's HIP-event
    # span then includes the time it waits for CUs the other stream holds (k_roll_scatter: 4.6 ms on its own, ~19 ms
    # next to the rebuild). What each kernel COSTS is therefore taken from a second, untimed build of the same batches,
    # one after the other on one stream (shk_count_chunks), with the same events.
    serial_prof, serial_dt = None, None
    if rank == 0 and world == 1 and not sharded and not args.serial and not args.host_text and not args.ablate:
        r2 = Run()
        r2.ctx.profile(True)
        r2.ctx.profile_reset()
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for s in range(args.steps):
            r2.ctx.count_chunks(texts[s].data_ptr(), offs, lens, on_device=True, text_bytes=texts[s].numel())
        torch.cuda.synchronize()
        serial_dt = time.perf_counter() - ts
        serial_prof = r2.ctx.profile_get()
        r2.ctx.close()
        del r2

    if rank == 0:
        # dominant kernel by accumulated device time (HIP events on the library's streams)
        cost = serial_prof or prof
        name, (launches, ms) = max(cost.items(), key=lambda kv: kv[1][1]) if cost else ("none", (1, 1.0))
        avg_s = ms / 1e3 / max(launches, 1)
        per_rank_kmers = counted / world
        # measured HBM traffic: PMC counters cannot run inside the timed loop; the numbers come from the committed
        # separate --pmc passes over this same default workload (null for any other workload)
        traffic, traffic_src, step_traffic = None, None, None
        try:
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            pk = {"k_region_merge<spill>": "k_region_merge<3, 24, false>", "k_region_merge<fused>": "k_region_merge<3, 24, true>",
                  "k_roll_scatter": "k_roll_scatter<1024, 1>", "k_rp_scatter": "k_rp_scatter<12, 512>",
                  "k_region_merge<sample>": "k_region_merge<0, 24, false>", "k_region_place": "k_region_place<24>"}.get(name, name)
            fresh = pt.get("csrc_sha256") == csrc_sha256()     # measured on exactly these kernel sources?
            if default_workload and not fresh:
                traffic_src = "stale: profiles/pmc_traffic.json was measured on other kernel sources (csrc_sha256 differs); re-run tools/profile_round.sh"
            if default_workload and fresh and pk in pt["kernels"]:
                e = pt["kernels"][pk]
                traffic = e["fetch_bytes_per_launch"] + e["write_bytes_per_launch"]
                if pk == "k_region_merge<3, 24, true>":
                    # a deNoise point is two launches of this kernel (all regions, then the few that hold a protected
                    # singleton); the HIP events here count the first: bytes per POINT = both
                    traffic *= 2
                traffic_src = pt["source"]
            if default_workload and fresh:
                step_traffic = pt.get("bytes_per_step")
        except (OSError, ValueError, KeyError):
            pass
        kern_ms = {k: round(v[1], 3) for k, v in prof.items()}
        kern_n = {k: int(v[0]) for k, v in prof.items()}
        table_bytes = tot.table_bytes
        # algorithmic bytes of the whole path = 179 B per k-mer presented + 2 x table per deNoise round
        path_bytes = ALGO_BYTES_PER_KMER * per_rank_kmers + rounds_fired * 2 * table_bytes
        out = {
            "metric": "k-mers counted/sec (whole node), CQF build, C.elegans-like k=47",
            "value": counted / dt, "unit": "k-mers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "build_time_s": dt, "build_kmers": counted, "build_rounds": rounds_fired,
            "config": {"workload": "clean CQF build from an empty filter: C.elegans-like synthetic reads (genome %d x n_gpus, L=150, e=0.00234), "
                                   "CQF-deNoise -k 47 -N %d -n %d -e 0.00234 -> qb=%d hb=%d rounds=%d trigger=%d (src/CQF-deNoise.cpp:96-161); "
                                   "run with rounds=%d trigger=%d (rounds added until the predicted peak load %.3f <= %.2f, shk/plan.py: the formula "
                                   "budgets every false k-mer as removable, but error k-mers seen twice between two rounds stay -- at 100x "
                                   "that is ~0.16 per genome k-mer, whatever the shape of the error profile (tests/test_plan.py), and the "
                                   "README's own data has them too: f2 = 26.1 M -- so the recipe as written fills the table in its last step); "
                                   "%d steps x %d reads/step/GPU, 8 MiB chunks; warm-up on a scratch filter"
                                   % (args.genome, pl["N"], pl["n"], pl["qb"], pl["qb"] + 8, pl["formula_rounds"], pl["formula_trigger"],
                                      nd, trigger, pl["predicted_peak_load"], args.max_load, args.steps, R),
                       "kmers_per_step_per_gpu": R * kmers_per_read, "denoise_rounds_fired": rounds_fired,
                       "removed": removed_total, "parallelism": "quotient-range shards x%d" % world,
                       "final_nelts": (run.sstate.nelts if sharded else end.nelts), "final_ndistinct": (run.sstate.ndistinct if sharded else end.ndistinct),
                       "predicted_peak_load": pl["predicted_peak_load"]},
            # the path-level figure; `traffic` = measured HBM bytes per step (separate --pmc passes)
            "roofline": {"bound": "hbm", "scope": "whole insert path: hash + partition + rebuild + deNoise rounds",
                         "achieved": path_bytes / dt / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": path_bytes / dt / HBM_PEAK, "traffic": step_traffic, "traffic_unit": "bytes per step, mean over the 20-step build with its 12 deNoise points (measured, all kernels; profiles/pmc_traffic.json)",
                         "formula": "(kmers*179 + rounds*2*table_bytes) / t / 8e12 (SURVEY.md 8d)",
                         "algorithmic_bytes_per_kmer": ALGO_BYTES_PER_KMER, "table_bytes": table_bytes},
            # the kernel with the most device time, on its MEASURED bytes (no algorithmic figure applies to one stage)
            "dominant_kernel": {"kernel": name, "launches": launches, "avg_launch_ms": avg_s * 1e3,
                                "share_of_step": ms / 1e3 / (serial_dt if serial_prof else dt),
                                "measured_in": ("a second, untimed build of the same batches on one stream (HIP events); in the timed, overlapped build "
                                                "the kernels of the two streams wait for each other's CUs" if serial_prof else "the timed region (HIP events)"),
                                "avg_launch_ms_in_timed_region": (prof[name][1] / max(prof[name][0], 1)) if name in prof else None,
                                "traffic": traffic, "traffic_unit": "bytes/launch (measured)", "traffic_source": traffic_src,
                                "achieved_real_GBps": (traffic / avg_s / 1e9) if traffic else None},
            "kernel_ms": kern_ms, "kernel_launches": kern_n,
        }
        if serial_prof:
            out["serial_build"] = {"what": "the same batches through shk_count_chunks, one after the other on one stream (untimed second build)",
                                   "ms_per_step": serial_dt / args.steps * 1e3,
                                   "kernel_ms": {k: round(v[1], 3) for k, v in serial_prof.items()},
                                   "kernel_launches": {k: int(v[0]) for k, v in serial_prof.items()}}
        if not args.no_secondary and world == 1 and not args.host_text and not args.ablate:
            # outside the timed region: what the headline value leaves out
            sec = {}
            try:
                sec["contiger"] = secondary_contiger(torch, shk, ctx, texts[0], offs, lens, K)
            except shk.ShkError as e:
                sec["contiger"] = {"error": str(e)}
            try:
                sec["host_text"] = secondary_host_text(torch, shk, new_ctx, texts, offs, lens)
            except shk.ShkError as e:
                sec["host_text"] = {"error": str(e)}
            out["secondary"] = sec
        if not args.no_cpu_baseline and world == 1:   # reported on rank 0 at N = 1 only
            text_cpu = texts[0].cpu().numpy().tobytes()
            out["cpu_baseline"] = cpu_baseline(torch, text_cpu, offs, lens, K, qb)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    ctx.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
