"""Quotient-range sharding over several GPUs (one process per GPU, torch.distributed).

`sharded_count` is the multi-GPU form of csrc/shk_api.hip `merge_stage`: every rank stages the
key words it OWNS (routing is the caller's all-to-all), and the ranks take the deNoise decision
together on all-reduced statistics, so a round fires after the same global chunk on every
shard -- the chunk at which the whole filter's distinct count reaches the trigger
(cqf/CQF_mt.h:837, 860-869). Works with RCCL ("nccl") on GPUs and with gloo on the CPU.

Collectives per batch: ONE all-to-all of key words (8 B per routed key over xGMI) and ONE small
all-reduce per decision (statistics, kernel flag bits and the ranks' local return codes travel in
the same vector; a batch without a deNoise point takes one decision). A rank-local failure is
therefore seen by every rank at the next decision and raised everywhere -- no rank is left
waiting in a collective its peers never enter."""
import os

import torch
import torch.distributed as dist

from . import HASH_FULL_BIT, HIST_BINS, LOOKBACK_BIT, SOFT_BITS, ShkError, Summary

NBITS = 10       # kernel flag bits (csrc/shk_device.h SHK_E_*)
NCODES = 8       # SHK_ERR_* codes -1 .. -8 (include/shk.h)


class ShardState:
    """filter-wide bookkeeping kept identically on every rank (runtime->ndistinct_elts etc.)"""

    def __init__(self, trigger, rounds_left, device):
        self.trigger, self.rounds_left, self.rounds_done = trigger, rounds_left, 0
        self.ndistinct = 0          # whole filter
        self.nelts = 0
        self.device = device
        self.pending_rc = 0         # return code of a rank-local call not yet shown to the peers
        self.collectives = 0        # small all-reduces issued (diagnostics)
        self.one_pass_points = 0    # deNoise points taken in one rebuild per shard (_one_pass_point)
        self.other_points = 0       # ... and on the three-pass path
        self.guesses = 0            # sampled predictions of the point's chunk, and how many the exact histogram confirmed
        self.guesses_right = 0
        self.inexact_rounds = 0     # rounds whose range walk restarted at every shard (only after a flagged one-pass try)
        self.keep = None


HASH_EXCHANGE_SECONDS = [0.0, 0.0]  # diagnostics: wall time in the library's hash + route call / in starting the exchange
COLLECTIVE_SECONDS = [0.0, 0]      # diagnostics: wall time spent in the small collectives (with their host round trip), and their number


def _allreduce(vals, st, op=None):
    import time
    t0 = time.perf_counter()
    t = torch.tensor(vals, dtype=torch.int64, device=st.device)
    dist.all_reduce(t, op=op or dist.ReduceOp.SUM)
    st.collectives += 1
    out = [int(x) for x in t.tolist()]
    COLLECTIVE_SECONDS[0] += time.perf_counter() - t0
    COLLECTIVE_SECONDS[1] += 1
    return out


def _local(st, call, default=None):
    """run a rank-local library call; a failure is remembered (not raised) until the next decision"""
    try:
        return call()
    except ShkError as e:
        if not st.pending_rc:
            st.pending_rc = e.code
        return default


class _Decision:
    """statistics of one try/summary on every rank, reduced: sums of the counters, OR of the flag bits"""
    __slots__ = ("newd", "before", "added", "removed", "hist", "chist", "bits", "local")

    @property
    def hard(self):
        return self.bits & ~(SOFT_BITS | HASH_FULL_BIT | LOOKBACK_BIT) & 0xFFFFFFFF

    @property
    def hfull(self):
        return self.bits & (HASH_FULL_BIT | LOOKBACK_BIT)

    @property
    def soft(self):
        return self.bits & SOFT_BITS


def _decide(ctx, st, call, chist_range=None, extra=None):
    """`call` -> Summary on this rank; ONE all-reduce carries its counters, its flag bits (one entry per bit), the
    exact first-chunk histogram of [lo, hi] when asked for (plus a "have it" count), and this rank's pending return
    code (one entry per code). Raises the same ShkError on every rank when any rank failed."""
    s = _local(st, call, None)
    if s is None:
        s = Summary()
    vec = [s.new_distinct, s.before, s.added, s.removed] + list(s.hist)
    vec += [(s.err_bits >> b) & 1 for b in range(NBITS)]
    vec += [1 if st.pending_rc == -c else 0 for c in range(1, NCODES + 1)]
    nch = 0
    if chist_range is not None:
        lo, hi = chist_range
        h = _local(st, lambda: ctx.stage_chunk_hist(hi + 1), None) if not st.pending_rc else None
        nch = hi - lo + 1
        vec += [0 if h is None else 1] + ([0] * nch if h is None else list(h[lo:hi + 1]))
    if extra:
        vec += list(extra)
    red = _allreduce(vec, st)
    codes = red[4 + HIST_BINS + NBITS: 4 + HIST_BINS + NBITS + NCODES]
    for c in range(1, NCODES + 1):
        if codes[c - 1]:
            st.pending_rc = 0
            raise ShkError(-c, "%s (on %d of the ranks)" % (ctx.L.shk_strerror(-c).decode(), codes[c - 1]))
    d = _Decision()
    d.newd, d.before, d.added, d.removed = red[0:4]
    d.hist = red[4:4 + HIST_BINS]
    d.bits = sum(1 << b for b in range(NBITS) if red[4 + HIST_BINS + b])
    d.local = s
    d.chist = None
    if chist_range is not None:
        base = 4 + HIST_BINS + NBITS + NCODES
        if red[base] == dist.get_world_size():
            d.chist = red[base + 1: base + 1 + nch]
    return d


def check(ctx, st):
    """make a rank-local failure of the last calls (accept/commit of the final batch) known everywhere; call once
    after the last batch"""
    red = _allreduce([1 if st.pending_rc == -c else 0 for c in range(1, NCODES + 1)], st)
    for c in range(1, NCODES + 1):
        if red[c - 1]:
            st.pending_rc = 0
            raise ShkError(-c, "%s (on %d of the ranks)" % (ctx.L.shk_strerror(-c).decode(), red[c - 1]))


def _point(st, d, lo, hi):
    """global chunk at which the filter-wide distinct count reaches the trigger (exact histogram of [lo, hi])"""
    run = st.ndistinct
    for i, v in enumerate(d.chist):
        run += v
        if run >= st.trigger:
            return lo + i
    return hi


def _codes(st):
    return [1 if st.pending_rc == -c else 0 for c in range(1, NCODES + 1)]


def _raise_codes(ctx, st, codes):
    for c in range(1, NCODES + 1):
        if codes[c - 1]:
            st.pending_rc = 0
            raise ShkError(-c, "%s (on %d of the ranks)" % (ctx.L.shk_strerror(-c).decode(), codes[c - 1]))


def _sample_stride(ctx):
    """the library's rule (csrc/shk_api.hip shk_create): every 8th region once a shard has 2^14 of them"""
    e = os.environ.get("SHK_SAMPLE_STRIDE")
    if e is not None:
        return int(e)
    return 8 if ctx.totals().nslots >= (1 << 22) else 0


def _sample_point(ctx, st, lo, hi):
    """-> (verdict, chunk): 0 no deNoise point expected in [lo, hi], 1 expected at `chunk`, 2 cannot tell. One statistics
    pass over a sample of every shard's regions, one all-reduce (the library's sample_locate, filter-wide)."""
    n = hi - lo + 1
    h, nr, ns, eb = _local(st, lambda: ctx.stage_sample(lo, hi), ([0] * (hi + 1), 0, 0, 1))
    red = _allreduce(list(h[lo:hi + 1]) + [nr, ns, 1 if eb else 0] + _codes(st), st)
    _raise_codes(ctx, st, red[n + 3:])
    if red[n + 2] or not red[n + 1] or st.ndistinct >= st.trigger:
        return 2, 0
    scale = red[n] / red[n + 1]
    need = st.trigger - st.ndistinct
    cum, at = 0.0, None
    for i in range(n):
        cum += scale * red[i]
        if at is None and cum >= need:
            at = lo + i
    margin = 6.0 * (scale * cum + 1.0) ** 0.5
    if cum + margin < need:
        return 0, 0
    if cum - margin >= need and at is not None:
        return 1, at
    return 2, 0


def _dbg(msg):
    if os.environ.get("SHK_DEBUG_FUSED") and dist.get_rank() == 0:
        print("SHK_DEBUG_FUSED (sharded)", msg, flush=True)


def _one_pass_point(ctx, st, lo, split, hi, out, known=False, words=True):
    """The deNoise point after global chunk `split`, the round and the chunks behind it in ONE rebuild per shard
    (include/shk.h shk_stage_point_*). `split` may be a guess: the pass records the exact first-chunk histogram and the
    ranks check it. Returns "done", or the exact chunk of the point when `split` was wrong, or None (not this way:
    nothing has been written; the caller takes another path).
    known: the round is due whatever the histogram says and has been booked by the caller (split = lo - 1: the round, then
    the chunks [lo, hi]); words=False: the round alone (shk_stage_round_try)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    n = hi - lo + 1 if words else 0
    p = _local(st, (lambda: ctx.stage_point_try(lo, split, hi)) if words else ctx.stage_round_try, None)
    ok = p is not None and not p.err_bits
    h = (_local(st, lambda: ctx.stage_chunk_hist(hi + 1), None) if ok and words and not st.pending_rc else None)
    if ok and not words:
        h = []
    bits = 0xFFFF if p is None else p.err_bits
    mine = [0, 0, 0] if p is None else [p.islots, p.ifin, p.first_used]
    slots = [0] * (3 * world)
    slots[3 * rank:3 * rank + 3] = mine
    vec = [(bits >> b) & 1 for b in range(NBITS)] + _codes(st) + [0 if h is None else 1] + ([0] * n if h is None else list(h[lo:hi + 1] if words else [])) + slots
    red = _allreduce(vec, st)
    _raise_codes(ctx, st, red[NBITS:NBITS + NCODES])
    base = NBITS + NCODES
    if any(red[:NBITS]) or red[base] != world:
        _dbg("point at %d of [%d, %d]: flags %s" % (split, lo, hi, red[:NBITS]))
        return None
    chist = red[base + 1:base + 1 + n]
    if known:
        newd_before = sum(chist[:max(0, split - lo + 1)])
    else:
        run, ch = st.ndistinct, None
        for i, v in enumerate(chist):
            run += v
            if run >= st.trigger:
                ch = lo + i
                break
        if ch is None:
            _dbg("guess %d of [%d, %d]: no point in the range" % (split, lo, hi))
            return None                       # (the trigger is not reached in [lo, hi] at all)
        if ch != split:
            _dbg("guess %d of [%d, %d]: the point is at %d" % (split, lo, hi, ch))
            return ch
        newd_before = run - st.ndistinct
    # every shard's table at the split as it lies in the single table: what the shards in front of it carry over its border
    g = red[base + 1 + n:]
    size = ctx.totals().nslots           # own quotients per shard (the same on every rank)
    carry, prev_fp = [0] * world, [-1] * world
    for r in range(1, world):
        fend = (r - 1) * size + max(carry[r - 1] + g[3 * (r - 1)], g[3 * (r - 1) + 1])
        prev_fp[r] = fend - r * size
        carry[r] = max(0, prev_fp[r])
    # the round's range walk runs through the shards one after the other (each range starts where the last one ended)
    state = torch.zeros(2, dtype=torch.int64, device=st.device)
    if rank > 0:
        dist.recv(state, src=rank - 1)
    nxt_used = g[3 * (rank + 1) + 2] if rank + 1 < world else 0
    so, nprot, web = _local(st, lambda: ctx.stage_point_walk(carry[rank], prev_fp[rank], rank == world - 1, nxt_used,
                                                             [int(x) for x in state.tolist()]), ((0, 0), 0, 0xFFFF))
    if rank + 1 < world:
        st.keep = torch.tensor(list(so), dtype=torch.int64, device=st.device)   # (alive until the next point: the send is asynchronous on a GPU)
        dist.send(st.keep, dst=rank + 1)
    acc = None
    if not web and not st.pending_rc:
        acc = _local(st, lambda: ctx.stage_point_finish(p), None)
    bad = 1 if (acc is None or acc.err_bits or web) else 0
    vec = ([0, 0, 0, 0] if bad else [p.new_after, p.added_after, p.removed, p.added_before]) + [bad] + _codes(st)
    red = _allreduce(vec, st)
    _raise_codes(ctx, st, red[5:])
    if red[4]:
        _dbg("point at %d of [%d, %d]: walk / second go flagged on %d rank(s)" % (split, lo, hi, red[4]))
        return None
    new_after, added_after, removed, added_before = red[0:4]
    if st.rounds_left > (0 if known else 1) and st.ndistinct + newd_before - removed + new_after >= st.trigger:
        _dbg("point at %d of [%d, %d]: a second point inside the rest" % (split, lo, hi))
        return None                       # a second point inside the rest: the rounds are taken one by one
    _local(st, lambda: ctx.stage_accept(acc))
    st.ndistinct += newd_before - removed + new_after
    st.nelts += added_before - removed + added_after
    if not known:
        st.rounds_left -= 1
        st.rounds_done += 1
        st.one_pass_points += 1
        out["denoise_rounds"] += 1
    out["removed"] += removed
    out["kmers"] += added_before + added_after
    out["new_distinct"] += newd_before + new_after
    _dbg("one-pass point at %d of [%d, %d]: removed %d" % (split, lo, hi, removed))
    return "done"


def sharded_denoise(ctx, st):
    """one deNoise round now on the whole sharded filter (the reference's --endDeNoise round; does not use up the
    rounds); returns the number of singletons removed"""
    out = {"kmers": 0, "new_distinct": 0, "removed": 0, "denoise_rounds": 0}
    if not os.environ.get("SHK_NO_FUSED_POINT") and _one_pass_point(ctx, st, 0, 0, 0, out, known=True, words=False) == "done":
        return out["removed"]
    st.inexact_rounds += 1
    r = _local(st, ctx.denoise, 0)
    red = _allreduce([r] + _codes(st), st)
    _raise_codes(ctx, st, red[1:])
    st.ndistinct -= red[0]
    st.nelts -= red[0]
    return red[0]


def sharded_count(ctx, st, nchunks):
    """insert the staged words of global chunks [0, nchunks); returns dict(kmers, new_distinct, removed, rounds)"""
    out = {"kmers": 0, "new_distinct": 0, "removed": 0, "denoise_rounds": 0}
    stride = _sample_stride(ctx)

    def book(d):
        st.ndistinct += d.newd
        st.nelts += d.added
        out["kmers"] += d.added
        out["new_distinct"] += d.newd

    def fail(d):
        bits = d.hard | d.hfull | d.soft
        _local(st, lambda: ctx.error_for_bits(bits))
        check(ctx, st)          # every rank raises here (the bits are the same everywhere)

    lo = 0
    while lo < nchunks:
        hi = nchunks - 1
        watch = st.rounds_left > 0
        verdict = 2
        if watch and stride > 1 and not os.environ.get("SHK_NO_FUSED_POINT"):
            # where the deNoise point falls, guessed from a sample of the regions; the one-pass point checks the guess
            verdict, guess = _sample_point(ctx, st, lo, hi)
            _dbg("sample of [%d, %d]: verdict %d at %d" % (lo, hi, verdict, guess))
            if verdict == 1 and guess + 1 < nchunks:
                st.guesses += 1
                r = _one_pass_point(ctx, st, lo, guess, nchunks - 1, out)
                if r == "done":
                    st.guesses_right += 1
                    lo = nchunks
                    continue
                if r is not None and r + 1 < nchunks:
                    if _one_pass_point(ctx, st, lo, r, nchunks - 1, out) == "done":
                        lo = nchunks
                        continue
        # common case: one try per rank does statistics (and, in the single-launch scheme, the table); accepted when
        # no rank saw an error and the whole filter stays below the trigger. While rounds are left the try also
        # records first chunks and their histogram rides along, so a deNoise point is located without more passes.
        wh = watch and verdict != 0      # (verdict 0: the sample rules a point out; should it be wrong the summary below is redone)
        d = _decide(ctx, st, lambda: ctx.stage_try(lo, hi, lo, 0, 2 if wh else 0), (lo, hi) if wh else None)
        if d.hard:
            fail(d)
        crosses = watch and st.ndistinct + d.newd >= st.trigger
        if not (d.hfull or d.soft) and not crosses:
            _local(st, lambda: ctx.stage_accept(d.local))
            book(d)
            lo = hi + 1
            continue
        point = _point(st, d, lo, hi) if (crosses and not d.hfull and d.chist is not None) else None
        shift = 0
        if point is None:
            while True:
                span = hi - lo + 1
                shift = 0
                while ((span + (1 << shift) - 1) >> shift) > HIST_BINS:
                    shift += 1
                d = _decide(ctx, st, lambda: ctx.stage_summary(lo, hi, lo, shift, 2 if watch else 0), (lo, hi) if watch else None)
                if d.hard:
                    fail(d)
                if d.hfull:
                    if hi == lo:
                        fail(d)
                    hi = lo + (hi - lo) // 2
                    continue
                break
            crosses = watch and st.ndistinct + d.newd >= st.trigger
            if crosses and d.chist is not None:
                point = _point(st, d, lo, hi)
        if crosses and point is not None and point + 1 < nchunks and hi == nchunks - 1 and not os.environ.get("SHK_NO_FUSED_POINT"):
            # the exact chunk is known and the point lies inside the batch: everything in one rebuild per shard
            if _one_pass_point(ctx, st, lo, point, nchunks - 1, out) == "done":
                lo = nchunks
                continue
        fire = False
        accepted = False
        if crosses:
            if point is not None:
                hi = point
            else:
                # 32-bin refinement (contexts without the exact histogram)
                base = lo
                d = _decide(ctx, st, lambda: ctx.stage_summary(lo, hi, lo, shift, True))
                if d.hard:
                    fail(d)
                while True:
                    run = st.ndistinct + d.before
                    b = 0
                    while b < HIST_BINS:
                        if run + d.hist[b] >= st.trigger:
                            break
                        run += d.hist[b]
                        b += 1
                    b = min(b, HIST_BINS - 1)
                    b_lo = base + (b << shift)
                    b_hi = min(b_lo + (1 << shift) - 1, hi)
                    if shift == 0:
                        hi = b_lo
                        break
                    span2 = b_hi - b_lo + 1
                    shift = 0
                    while ((span2 + (1 << shift) - 1) >> shift) > HIST_BINS:
                        shift += 1
                    base = b_lo
                    d = _decide(ctx, st, lambda: ctx.stage_summary(lo, b_hi, base, shift, True))
                    if d.hard:
                        fail(d)
            fire = True
            d = _decide(ctx, st, lambda: ctx.stage_try(lo, hi, lo, 0, 0))
            if not (d.hard or d.hfull or d.soft):
                _local(st, lambda: ctx.stage_accept(d.local))
                accepted = True
            else:
                d = _decide(ctx, st, lambda: ctx.stage_summary(lo, hi, lo, 0, 0))
        if d.hard or d.hfull or d.soft:
            fail(d)
        if not accepted:
            _local(st, lambda: ctx.stage_commit(lo, hi, d.local))
        book(d)
        if fire:
            st.rounds_left -= 1
            st.rounds_done += 1
            st.other_points += 1
            out["denoise_rounds"] += 1
            fused = False
            exact = not os.environ.get("SHK_NO_FUSED_POINT")
            if exact and hi + 1 < nchunks and not os.environ.get("SHK_NO_FUSED_DENOISE"):
                # the round and the chunks behind the point in one rebuild, the range walk over the single table's layout
                if _one_pass_point(ctx, st, hi + 1, hi, nchunks - 1, out, known=True) == "done":
                    fused = True
                    hi = nchunks - 1
            if not fused and exact:
                # the round alone, the same way
                fused = _one_pass_point(ctx, st, 0, 0, 0, out, known=True, words=False) == "done"
            if not fused and hi + 1 < nchunks and not os.environ.get("SHK_NO_FUSED_DENOISE"):
                # (last resort, and SHK_NO_FUSED_POINT: each shard walks its own part from its own first slot -- range-end
                # singletons next to shard borders may differ from the single table's, st.inexact_rounds counts these)
                # one pass: drop the singletons and insert the chunks behind the deNoise point; taken when no rank
                # objects and the trigger is not reached again inside the rest
                d = _decide(ctx, st, lambda: ctx.stage_try_denoise(hi + 1, nchunks - 1))
                again = st.rounds_left > 0 and st.ndistinct - d.removed + d.newd >= st.trigger
                if not d.bits and not again:
                    _local(st, lambda: ctx.stage_accept(d.local))
                    st.ndistinct += d.newd - d.removed
                    st.nelts += d.added - d.removed
                    out["removed"] += d.removed
                    out["kmers"] += d.added
                    out["new_distinct"] += d.newd
                    fused = True
                    hi = nchunks - 1
                    st.inexact_rounds += 1
            if not fused:
                st.inexact_rounds += 1
                r = _local(st, ctx.denoise, 0)
                red = _allreduce([r] + [1 if st.pending_rc == -c else 0 for c in range(1, NCODES + 1)], st)
                for c in range(1, NCODES + 1):
                    if red[c]:
                        st.pending_rc = 0
                        raise ShkError(-c, "%s (on %d of the ranks)" % (ctx.L.shk_strerror(-c).decode(), red[c]))
                st.ndistinct -= red[0]
                st.nelts -= red[0]
                out["removed"] += red[0]
        lo = hi + 1
    return out


# Elements per peer per collective call. One RCCL send/receive of 2 GiB or more arrives damaged on this stack: with
# plain torch tensors and one rank, all_to_all / all_to_all_single of 2^28 int64 (2 GiB) deliver exactly half of the
# words wrong, 2^27 words (1 GiB) are intact (tools/a2a_probe.py, profiles/r02_a2a_probe.json) -- a 32-bit byte count
# somewhere below torch.distributed. Messages are therefore cut into pieces of at most 2^27 words (1 GiB) per peer.
ROUTE_PIECE = 1 << 27


class _CAI:
    """expose a raw device pointer to torch through __cuda_array_interface__"""

    def __init__(self, ptr, n, typestr="<i8"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def wrap_words(ptr, n, device):
    """int64 tensor view of n key words at a library-owned pointer (device memory on a GPU,
    host memory in the CPU emulator build)"""
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=device)
    if device.type == "cuda":
        return torch.as_tensor(_CAI(ptr, n), device=device)
    import ctypes
    return torch.frombuffer((ctypes.c_int64 * n).from_address(ptr), dtype=torch.int64)


def wrap_bytes(ptr, n, device):
    if device.type == "cuda":
        return torch.as_tensor(_CAI(ptr, n, "|u1"), device=device)
    import ctypes
    return torch.frombuffer((ctypes.c_uint8 * n).from_address(ptr), dtype=torch.uint8)


def _recv_buffer(ctx, n, device):
    """Receive side of the all-to-all: two buffers per context, used alternately (batch s is staged from one while batch
    s + 1 arrives in the other), each as large as a batch may be (max_batch_keys: more could not be staged anyway). A fresh
    multi-GB tensor per batch would go through the caching allocator with a different size every time -- on an MI355X
    that more than doubled the time of a step (bench.py --force-dist: 123 ms against 54)."""
    if device.type != "cuda":
        return torch.empty((n,), dtype=torch.int64, device=device)
    cap = int(ctx.cfg.max_batch_keys)
    if n > cap:
        raise ShkError(-7, "more key words for this shard (%d) than max_batch_keys (%d)" % (n, cap))
    pool = getattr(ctx, "_recv_pool", None)
    if pool is None:
        pool = ctx._recv_pool = {"bufs": [None, None], "next": 0}
    i = pool["next"]
    pool["next"] ^= 1
    if pool["bufs"][i] is None:
        pool["bufs"][i] = torch.empty((cap,), dtype=torch.int64, device=device)
    return pool["bufs"][i][:n]


def reserve_exchange(ctx, device):
    """allocate the exchange's buffers (the library's two send buffers, the two receive tensors) at set-up time. Left to
    their first use they are allocated inside the first steps of a build -- 10 GB each at the bench's batch size -- and on
    an MI355X that made about every second run of `bench.py --force-dist` three to four times slower per step, for the
    whole build, with unchanged kernel times (measured; with the reservation: 8 of 8 runs at 38 ms per step)"""
    ctx.route_reserve()
    for _ in range(2):
        _recv_buffer(ctx, 0, device)


class Exchange:
    """one all-to-all of key words in flight: start() after shk_route_words, wait() before shk_stage_words. Between the
    two the caller may hash and route the NEXT batch (the library keeps two send buffers), which hides the exchange
    behind compute: xGMI moves 8 B per routed key while the CUs hash."""

    def __init__(self, ctx, nwords, hb, world, rank, device, async_op=True, local_rc=0, routed=None, keep_own=False):
        # local_rc: the return code of the rank-local work in front of this exchange (shk_hash_chunks), 0 = fine.
        # A rank-local failure -- there, or in shk_route_words here -- must not keep this rank out of the all-gather its
        # peers enter: the code travels in the gathered vector (one extra column) and EVERY rank raises after the gather.
        # routed = (device pointer, counts per shard): the words are binned already (shk_hash_route_chunks)
        rc, dp, sc = int(local_rc), None, [0] * world
        if not rc and routed is not None:
            dp, sc = routed
        elif not rc:
            try:
                dp, sc = ctx.route_words(nwords, world)
            except ShkError as e:
                rc = e.code
        self.world, self.device = world, device
        # every rank learns every bin size in one all-gather: its own receive counts and the number of pieces
        mine = torch.tensor(list(sc) + [rc], dtype=torch.int64, device=device)
        allc = torch.empty((world * (world + 1),), dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(allc, mine)
        allc = allc.view(world, world + 1).tolist()
        codes = [int(row[world]) for row in allc]
        if any(codes):
            bad = next(p for p in range(world) if codes[p])
            raise ShkError(codes[bad], "rank %d failed in front of the exchange (libshk error %d); raised on every rank" % (bad, codes[bad]))
        allc = [row[:world] for row in allc]
        self.send = wrap_words(dp, nwords, device)
        rc = [allc[p][rank] for p in range(world)]
        # every rank sees every bin size: a shard that would receive more than it can stage fails on ALL ranks here
        # (contexts of one job are created alike), not on its own rank inside a collective the others then wait in
        cap = int(ctx.cfg.max_batch_keys)
        worst = max(sum(allc[p][r] for p in range(world)) for r in range(world))
        if worst > cap:
            raise ShkError(-7, "a shard would receive %d key words, more than max_batch_keys (%d)" % (worst, cap))
        # keep_own (a decision of the whole job: every rank passes the same flag): the words a rank owns itself stay where
        # the routing left them -- `own` = (device pointer, count) inside its send buffer, valid until the call after
        # next of shk_hash_route_chunks / shk_route_words -- and shk_stage_words_pair reads them from there next to the
        # received ones: 1/world of the words never goes through the collective (with one rank: none does)
        self.own = (0, 0)
        soff = [sum(sc[:p]) for p in range(world)]
        if keep_own:
            if dp and sc[rank]:
                self.own = (int(dp) + 8 * soff[rank], int(sc[rank]))
            for p in range(world):
                allc[p][p] = 0
            rc[rank] = 0
            sc = [0 if p == rank else sc[p] for p in range(world)]      # (soff stays: the bins' places in the send buffer)
        mx = max(max(row) for row in allc)
        self.recv = _recv_buffer(ctx, sum(rc), device)
        roff = [sum(rc[:p]) for p in range(world)]
        self.work = []
        if keep_own and world == 1:
            return
        if world == 1 and device.type != "cuda":
            self.recv.copy_(self.send)
            return
        # (one rank on a GPU still goes through RCCL: its self-send copies with a few workgroups next to the running
        # kernels -- a plain tensor copy on torch's stream takes the whole chip and cost 16 ms per 832 M-key step)
        piece_words = int(os.environ.get("SHK_ROUTE_PIECE", ROUTE_PIECE))
        grouped = device.type == "cuda" and not os.environ.get("SHK_A2A_SINGLE")
        for r0 in range(0, max(mx, 1), piece_words):
            ins = [self.send[soff[p] + min(r0, sc[p]): soff[p] + min(r0 + piece_words, sc[p])] for p in range(world)]
            outs = [self.recv[roff[p] + min(r0, rc[p]): roff[p] + min(r0 + piece_words, rc[p])] for p in range(world)]
            if grouped:
                # RCCL: grouped sends/receives straight between the bins and their places in `recv` (views, no staging)
                w = dist.all_to_all(outs, ins, async_op=async_op)
                if async_op:
                    self.work.append(w)
                continue
            isz, osz = [int(x.numel()) for x in ins], [int(x.numel()) for x in outs]
            piece = torch.empty((sum(osz),), dtype=torch.int64, device=device)
            dist.all_to_all_single(piece, torch.cat(ins), output_split_sizes=osz, input_split_sizes=isz)
            o = 0
            for p in range(world):
                outs[p].copy_(piece[o:o + osz[p]])
                o += osz[p]

    def wait(self):
        """the received words, complete and visible to the library's stream"""
        for w in self.work:
            w.wait()
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()
        return self.recv


def hash_and_exchange(ctx, text, offs, lens, hb, world, rank, device, on_device=False, text_bytes=None, async_op=True, keep_own=False):
    """shk_hash_chunks + Exchange with the rank-local failure of either carried to every rank (see Exchange).
    keep_own: the rank's own words are not copied (Exchange.own; stage with stage_received(..., own=ex.own))"""
    import time
    t0 = time.perf_counter()
    rc, nw, routed = 0, 0, None
    try:
        if offs and not os.environ.get("SHK_NO_ROLL"):
            # one pass over the text: every k-mer hashed and sent straight to its owner's bin of the send buffer
            dp, sc, nw = ctx.hash_route_chunks(text, offs, lens, world, on_device=on_device, text_bytes=text_bytes)
            routed = (dp, sc)
        elif offs:
            _, nw = ctx.hash_chunks(text, offs, lens, on_device=on_device, text_bytes=text_bytes)
    except ShkError as e:
        rc = e.code
    t1 = time.perf_counter()
    ex = Exchange(ctx, nw, hb, world, rank, device, async_op=async_op, local_rc=rc, routed=routed, keep_own=keep_own)
    HASH_EXCHANGE_SECONDS[0] += t1 - t0
    HASH_EXCHANGE_SECONDS[1] += time.perf_counter() - t1
    return ex


def stage_received(ctx, st, recv, own=None):
    """shk_stage_words in front of a collective decision: a rank-local failure is parked in the shard state and shown
    to the peers by the next all-reduce (sharded_count raises it on every rank). own = Exchange.own: the rank's own
    words are read where the routing left them (shk_stage_words_pair)"""
    if own and own[1]:
        _local(st, lambda: ctx.stage_words_pair(own[0], own[1], recv.data_ptr() if recv.numel() else 0, recv.numel()))
    else:
        _local(st, lambda: ctx.stage_words(recv.data_ptr(), recv.numel()))


def route_words(ctx, nwords, hb, world, rank, device):
    """Exchange the key words shk_hash_chunks left in the context (their chunk indices are already
    global: a sharded context labels chunk i as i * world + rank, so the ranks' parts interleave like
    the reference's file queue, cqf/CQF_mt.h:828-830): the library bins them by owner
    (shk_route_words), one all-to-all moves the bins. Blocking form of `Exchange`."""
    return Exchange(ctx, nwords, hb, world, rank, device, async_op=False).wait()


_HOSTLIB = None


def _hostlib():
    """libshkhost.so: the host-only pieces (chunker, per-rank stitch)"""
    global _HOSTLIB
    if _HOSTLIB is None:
        import ctypes as C
        L = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libshkhost.so"))
        L.shkh_shard_summary.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
        L.shkh_shard_layout.restype = C.c_longlong
        L.shkh_shard_layout.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_char_p, C.c_uint64,
                                        C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.shkh_shard_spill.argtypes = [C.c_char_p, C.c_char_p]
        L.shkh_shard_apply_spill.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_char_p, C.c_char_p, C.c_uint64]
        _HOSTLIB = L
    return _HOSTLIB


def cqf_header(qb, seed, nelts, ndistinct):
    """the 128-byte qfmetadata image qf_serialize writes (cqf/gqf.h:62-77, cqf/CQF_mt.h:986-987) for the WHOLE filter"""
    import math
    import struct
    nslots = 1 << qb
    xnslots = nslots + int(10 * math.sqrt(float(nslots)))
    nblocks = (xnslots + 63) // 64
    h = bytearray(128)
    struct.pack_into("<Q", h, 0, nblocks * 89)
    struct.pack_into("<I", h, 8, seed)
    struct.pack_into("<QQQQQQ", h, 16, nslots, xnslots, qb + 8, 0, 8, 8)
    h[64:80] = (nslots << 8).to_bytes(16, "little")
    struct.pack_into("<QQQQQ", h, 80, nblocks, nelts, ndistinct, 0, xnslots // (1 << 16) + 2)
    return bytes(h)


def export_cqf(ctx, st, path, world, rank, device, qb, k=21):
    """One .cqf of the whole filter written BY ALL RANKS, each placing its own blocks (qf_serialize's bytes,
    cqf/gqf.c:2379-2394): nobody gathers anybody's table. A shard is a function on the free pointer, f -> max(f + a, b)
    (a = slots its runs take, b = where they end with nothing carried in): the ranks all-gather their (a, b), fold the
    pairs of the ranks in front of them into the free pointer they start from, lay their runs out into their own block
    range of the single table (host/stitch.cpp: occupieds, runends, slots, block offsets) and exchange what spills
    behind it -- normally a few hundred slots. Rank 0 writes the header; every rank pwrites its blocks at their
    offset. Returns the path on every rank."""
    import ctypes as C
    import math
    L = _hostlib()
    nslots = 1 << qb
    xnslots = nslots + int(10 * math.sqrt(float(nslots)))
    nblocks = (xnslots + 63) // 64
    per = nslots // world
    shard = ctx.blocks()                                   # this shard's table, device -> host
    ab = (C.c_uint64 * 2)()
    rc = L.shkh_shard_summary(shard, len(shard) // 89, rank, world, qb, ab)
    # the pair, the local return code and (later) the spill's start and length travel in all-gathers: a rank whose
    # shard is damaged still takes part, and every rank raises
    mine = torch.tensor([int(ab[0]), int(ab[1]), rc], dtype=torch.int64, device=device)
    allv = torch.empty((world * 3,), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(allv, mine)
    allv = allv.view(world, 3).tolist()
    if any(v[2] for v in allv):
        raise ShkError(-5, "a shard's table is inconsistent (rank %d)" % next(p for p in range(world) if allv[p][2]))
    free_in = 0
    for p in range(rank):
        free_in = max(free_in + allv[p][0], allv[p][1])
    b_lo = per * rank // 64
    b_hi = nblocks if rank == world - 1 else per * (rank + 1) // 64
    own = C.create_string_buffer((b_hi - b_lo) * 89)
    ss, fo = C.c_uint64(), C.c_uint64()
    n = L.shkh_shard_layout(shard, len(shard) // 89, rank, world, qb, free_in, own, len(own), C.byref(ss), C.byref(fo))
    del shard
    sl, re_ = C.create_string_buffer(max(int(n), 1)), C.create_string_buffer(max(int(n), 0) // 8 + 1)
    if n > 0:
        L.shkh_shard_spill(sl, re_)
    mine = torch.tensor([int(ss.value), int(n)], dtype=torch.int64, device=device)
    alls = torch.empty((world * 2,), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(alls, mine)
    alls = alls.view(world, 2).tolist()
    if any(v[1] < 0 for v in alls):
        raise ShkError(-3, "the stitched runs pass the end of the table (rank %d)" % next(p for p in range(world) if alls[p][1] < 0))
    mx = max(v[1] for v in alls)
    if mx > 0:
        # spills: slot bytes, then the run-end bits, padded to the longest
        width = mx + mx // 8 + 1
        blob = torch.zeros((width,), dtype=torch.uint8)
        if n > 0:
            blob[:n] = torch.frombuffer(bytearray(sl.raw[:n]), dtype=torch.uint8)
            blob[mx:mx + n // 8 + 1] = torch.frombuffer(bytearray(re_.raw[:n // 8 + 1]), dtype=torch.uint8)
        blob = blob.to(device)
        allb = torch.empty((world * width,), dtype=torch.uint8, device=device)
        dist.all_gather_into_tensor(allb, blob)
        allb = allb.view(world, width).cpu()
        for p in range(rank):                              # whatever earlier ranks spill may land in my blocks
            sp, np_ = alls[p]
            if np_ > 0 and sp + np_ > b_lo * 64:
                row = allb[p].numpy().tobytes()
                L.shkh_shard_apply_spill(own, rank, world, qb, sp, row[:np_], row[mx:mx + np_ // 8 + 1], np_)
    if rank == 0:
        with open(path, "wb") as f:
            f.write(cqf_header(qb, ctx.cfg.seed, st.nelts, st.ndistinct))
            f.truncate(128 + nblocks * 89)
    dist.barrier()
    fd = os.open(path, os.O_WRONLY)
    try:
        os.pwrite(fd, own.raw, 128 + b_lo * 89)
    finally:
        os.close(fd)
    dist.barrier()
    return path
