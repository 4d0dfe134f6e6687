"""Quotient-range sharding over several GPUs (one process per GPU, torch.distributed).

`sharded_count` is the multi-GPU form of csrc/shk_api.hip `merge_stage`: every rank stages the
key words it OWNS (routing is the caller's all-to-all), and the ranks take the deNoise decision
together on all-reduced statistics, so a round fires after the same global chunk on every
shard -- the chunk at which the whole filter's distinct count reaches the trigger
(cqf/CQF_mt.h:837, 860-869). Works with RCCL ("nccl") on GPUs and with gloo on the CPU."""
import os

import torch
import torch.distributed as dist

from . import HASH_FULL_BIT, HIST_BINS, LOOKBACK_BIT, SOFT_BITS, ShkError


class ShardState:
    """filter-wide bookkeeping kept identically on every rank (runtime->ndistinct_elts etc.)"""

    def __init__(self, trigger, rounds_left, device):
        self.trigger, self.rounds_left, self.rounds_done = trigger, rounds_left, 0
        self.ndistinct = 0          # whole filter
        self.nelts = 0
        self.device = device


def _allreduce(vals, device, op=None):
    t = torch.tensor(vals, dtype=torch.int64, device=device)
    dist.all_reduce(t, op=op or dist.ReduceOp.SUM)
    return [int(x) for x in t.tolist()]


def _summary(ctx, st, lo, hi, base, shift, want_hist, single=False):
    """local summary (or single-launch try) + reduction; returns
    (newd, before, hist[], hard_bits_any, hash_full_any, soft_any, local)"""
    s = ctx.stage_try(lo, hi, base, shift, want_hist) if single else ctx.stage_summary(lo, hi, base, shift, want_hist)
    red = _allreduce([s.new_distinct, s.before] + list(s.hist), st.device)
    flags = _allreduce([s.err_bits & ~(SOFT_BITS | HASH_FULL_BIT | LOOKBACK_BIT) & 0xFFFFFFFF,
                        s.err_bits & (HASH_FULL_BIT | LOOKBACK_BIT), s.err_bits & SOFT_BITS], st.device, dist.ReduceOp.MAX)
    return red[0], red[1], red[2:], flags[0], flags[1], flags[2], s


def _exact_point(ctx, st, lo, hi):
    """global chunk at which the filter-wide distinct count reaches the trigger, from the exact
    per-chunk histograms of the pass just run with want_hist=2 (None when a rank has none)"""
    h = ctx.stage_chunk_hist(hi + 1)
    ok = _allreduce([0 if h is None else 1], st.device, dist.ReduceOp.MIN)[0]
    if not ok:
        return None
    t = torch.tensor(h[lo:hi + 1], dtype=torch.int64, device=st.device)
    dist.all_reduce(t)
    run = st.ndistinct
    for i, v in enumerate(t.tolist()):
        run += v
        if run >= st.trigger:
            return lo + i
    return hi


def sharded_count(ctx, st, nchunks):
    """insert the staged words of global chunks [0, nchunks); returns dict(kmers, new_distinct, removed, rounds)"""
    out = {"kmers": 0, "new_distinct": 0, "removed": 0, "denoise_rounds": 0}
    lo = 0
    while lo < nchunks:
        hi = nchunks - 1
        watch = st.rounds_left > 0
        # common case: one try per rank does statistics (and, in the single-launch scheme, the table);
        # accepted when no rank saw an error and the whole filter stays below the trigger. While rounds
        # are left the try also records first chunks, so a deNoise point is located without more passes.
        newd, before, hist, hard, hfull, soft, loc = _summary(ctx, st, lo, hi, lo, 0, 2 if watch else 0, single=True)
        if hard:
            ctx.error_for_bits(hard)
        crosses = watch and st.ndistinct + newd >= st.trigger
        if not (hfull or soft) and not crosses:
            ctx.stage_accept(loc)
            added = _allreduce([loc.added], st.device)[0]
            st.ndistinct += newd
            st.nelts += added
            out["kmers"] += added
            out["new_distinct"] += newd
            lo = hi + 1
            continue
        point = _exact_point(ctx, st, lo, hi) if (crosses and not hfull) else None
        if point is None:
            while True:
                span = hi - lo + 1
                shift = 0
                while ((span + (1 << shift) - 1) >> shift) > HIST_BINS:
                    shift += 1
                newd, before, hist, hard, hfull, soft, loc = _summary(ctx, st, lo, hi, lo, shift, 2 if watch else 0)
                if hard:
                    ctx.error_for_bits(hard)
                if hfull:
                    if hi == lo:
                        ctx.error_for_bits(HASH_FULL_BIT)
                    hi = lo + (hi - lo) // 2
                    continue
                break
            crosses = watch and st.ndistinct + newd >= st.trigger
            if crosses:
                point = _exact_point(ctx, st, lo, hi)
        fire = False
        accepted = False
        if crosses:
            if point is not None:
                hi = point
            else:
                # 32-bin refinement (contexts without the exact histogram)
                base = lo
                newd, before, hist, hard, hfull, soft, loc = _summary(ctx, st, lo, hi, lo, shift, True)
                if hard:
                    ctx.error_for_bits(hard)
                while True:
                    run = st.ndistinct + before
                    b = 0
                    while b < HIST_BINS:
                        if run + hist[b] >= st.trigger:
                            break
                        run += hist[b]
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
                    newd, before, hist, hard, hfull, soft, loc = _summary(ctx, st, lo, b_hi, base, shift, True)
                    if hard:
                        ctx.error_for_bits(hard)
            fire = True
            newd, before, hist, hard, hfull, soft, loc = _summary(ctx, st, lo, hi, lo, 0, False, single=True)
            if not (hard or hfull or soft):
                ctx.stage_accept(loc)
                accepted = True
            else:
                newd, before, hist, hard, hfull, soft, loc = _summary(ctx, st, lo, hi, lo, 0, False)
        if hard or hfull or soft:
            ctx.error_for_bits(hard | hfull | soft)
        if not accepted:
            ctx.stage_commit(lo, hi, loc)
        added = _allreduce([loc.added], st.device)[0]
        st.ndistinct += newd
        st.nelts += added
        out["kmers"] += added
        out["new_distinct"] += newd
        if fire:
            st.rounds_left -= 1
            st.rounds_done += 1
            out["denoise_rounds"] += 1
            fused = False
            if hi + 1 < nchunks and not os.environ.get("SHK_NO_FUSED_DENOISE"):
                # one pass: drop the singletons and insert the chunks behind the deNoise point; taken when no rank
                # objects and the trigger is not reached again inside the rest
                loc = ctx.stage_try_denoise(hi + 1, nchunks - 1)
                red = _allreduce([loc.new_distinct, loc.removed, loc.added], st.device)
                bad = _allreduce([1 if loc.err_bits else 0], st.device, dist.ReduceOp.MAX)[0]
                again = st.rounds_left > 0 and st.ndistinct - red[1] + red[0] >= st.trigger
                if not bad and not again:
                    ctx.stage_accept(loc)
                    st.ndistinct += red[0] - red[1]
                    st.nelts += red[2] - red[1]
                    out["removed"] += red[1]
                    out["kmers"] += red[2]
                    out["new_distinct"] += red[0]
                    fused = True
                    hi = nchunks - 1
            if not fused:
                removed = _allreduce([ctx.denoise()], st.device)[0]
                st.ndistinct -= removed
                st.nelts -= removed
                out["removed"] += removed
        lo = hi + 1
    return out


ROUTE_PIECE = 1 << 25   # elements per peer per exchange (256 MB of key words)


class _CAI:
    """expose a raw device pointer to torch through __cuda_array_interface__"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (ptr, False), "version": 2}


def wrap_words(ptr, n, device):
    """int64 tensor view of n key words at a library-owned pointer (device memory on a GPU,
    host memory in the CPU emulator build)"""
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=device)
    if device.type == "cuda":
        return torch.as_tensor(_CAI(ptr, n), device=device)
    import ctypes
    return torch.frombuffer((ctypes.c_int64 * n).from_address(ptr), dtype=torch.int64)


def route_words(ctx, nwords, hb, world, rank, device):
    """Exchange the key words shk_hash_chunks left in the context (their chunk indices are already
    global: a sharded context labels chunk i as i * world + rank, so the ranks' parts interleave like
    the reference's file queue, cqf/CQF_mt.h:828-830): the library bins them by owner
    (shk_route_words), and the bins travel in all-to-alls of at most ROUTE_PIECE words per peer (one
    RCCL all_to_all_single of ~1.6 GB per peer was observed to deliver only its first 832 MB on this
    stack; bounded pieces also bound the staging memory)."""
    dp, sc = ctx.route_words(nwords, world)
    send = wrap_words(dp, nwords, device)
    send_counts = torch.tensor(sc, dtype=torch.int64, device=device)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts)
    rc = recv_counts.tolist()
    recv = torch.empty((sum(rc),), dtype=torch.int64, device=device)
    mx = _allreduce([max(sc + rc)], device, dist.ReduceOp.MAX)[0]
    soff = [sum(sc[:p]) for p in range(world)]
    roff = [sum(rc[:p]) for p in range(world)]
    for r0 in range(0, max(mx, 1), ROUTE_PIECE):
        ins = [send[soff[p] + min(r0, sc[p]): soff[p] + min(r0 + ROUTE_PIECE, sc[p])] for p in range(world)]
        outs = [recv[roff[p] + min(r0, rc[p]): roff[p] + min(r0 + ROUTE_PIECE, rc[p])] for p in range(world)]
        if world == 1 and not os.environ.get("SHK_A2A_NO_BYPASS"):     # (the variable lets a one-rank test drive the collective)
            outs[0].copy_(ins[0])
            continue
        if device.type == "cuda" and not os.environ.get("SHK_A2A_SINGLE"):
            # RCCL: grouped sends/receives straight between the bins and their places in `recv` (views, no staging copies)
            dist.all_to_all(outs, ins)
            continue
        isz, osz = [int(x.numel()) for x in ins], [int(x.numel()) for x in outs]
        piece = torch.empty((sum(osz),), dtype=torch.int64, device=device)
        dist.all_to_all_single(piece, torch.cat(ins), output_split_sizes=osz, input_split_sizes=isz)
        o = 0
        for p in range(world):
            outs[p].copy_(piece[o:o + osz[p]])
            o += osz[p]
    return recv
