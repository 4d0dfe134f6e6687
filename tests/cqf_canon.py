"""Pure-python canonical CQF layout builder (test infrastructure).

Given a multiset {key: count} it produces the exact block bytes the reference
filter holds for that multiset (bits_per_slot = 8), following the layout rules
read from /root/reference/cqf/gqf.c:
  * qfblock layout            gqf.c:63-86   (1 B offset, occupieds, runends, traveled, 64 slots)
  * counter encoding          gqf.c:1225-1255 (encode_counter)
  * run placement             gqf.c:1614-1915 (insert1_advance keeps runs at max(q, prev_end+1),
                                               remainders ascending)
  * block offset              gqf.c:599-601, 2868-2875 (block_offset_strict, clamped to 255)
Small inputs only (python loops)."""
import math
import numpy as np

def encode_counter(rem, count):
    """slots for (remainder, count>=1), gqf.c:1225-1255 with bits_per_slot=8"""
    if count == 1:
        return [rem]
    c = count - 1
    digits = [c % 128]
    c //= 128
    while c:
        digits.append((c % 128) | 0x80)
        c //= 128
    top = digits[-1]
    out = [rem]
    if top > rem:
        out.append(0)
    out.extend(reversed(digits))
    return out

def geometry(qb, hb):
    nslots = 1 << qb
    xnslots = nslots + int(10 * math.sqrt(float(nslots)))
    nblocks = (xnslots + 63) // 64
    return nslots, xnslots, nblocks

def build_blocks(qb, hb, counts):
    """counts: dict key->count (key < 2^hb, hb == qb+8). returns bytes of nblocks*89"""
    assert hb == qb + 8
    nslots, xnslots, nblocks = geometry(qb, hb)
    return build_blocks_geom(xnslots, counts)


def build_shard_blocks(qb, g, nshards, counts):
    """the table of quotient-range shard g of `nshards` (keys with quotient in [g, g+1) * 2^qb / nshards, stored with
    quotients relative to the shard's first one; its own overflow tail of 10 * sqrt(2^qb) slots, as libshk lays shards out)"""
    nslots = 1 << qb
    per = nslots // nshards
    q_lo = per * g
    local = {k - (q_lo << 8): c for k, c in counts.items() if q_lo <= (k >> 8) < q_lo + per}
    return build_blocks_geom(per + int(10 * math.sqrt(float(nslots))), local)


def build_blocks_geom(xnslots, counts):
    nblocks = (xnslots + 63) // 64
    slots = np.zeros(nblocks * 64, dtype=np.uint8)
    occ = np.zeros(nblocks * 64, dtype=np.uint8)
    rend = np.zeros(nblocks * 64, dtype=np.uint8)
    free = 0
    byq = {}
    for key, c in counts.items():
        byq.setdefault(key >> 8, []).append((key & 0xff, c))
    # end_before[b] = free pointer after all runs with quotient < 64*b
    run_ends = []  # (q, end)
    for q in sorted(byq):
        start = max(q, free)
        pos = start
        for r, c in sorted(byq[q]):
            for s in encode_counter(r, c):
                if pos >= xnslots:
                    raise OverflowError("table full")
                slots[pos] = s
                pos += 1
        occ[q] = 1
        rend[pos - 1] = 1
        free = pos
        run_ends.append((q, pos - 1))
    out = bytearray(nblocks * 89)
    # offsets: block b: max(0, (end of last run with q < 64b) - 64b + 1), clamp 255
    ends_q = [q for q, _ in run_ends]
    import bisect
    for b in range(nblocks):
        off = 0
        if b > 0:
            i = bisect.bisect_left(ends_q, 64 * b) - 1
            if i >= 0:
                off = max(0, run_ends[i][1] - 64 * b + 1)
        out[b * 89] = min(off, 255)
        o = int.from_bytes(np.packbits(occ[64 * b:64 * b + 64], bitorder='little').tobytes(), 'little')
        r = int.from_bytes(np.packbits(rend[64 * b:64 * b + 64], bitorder='little').tobytes(), 'little')
        out[b * 89 + 1:b * 89 + 9] = o.to_bytes(8, 'little')
        out[b * 89 + 9:b * 89 + 17] = r.to_bytes(8, 'little')
        out[b * 89 + 25:b * 89 + 89] = slots[64 * b:64 * b + 64].tobytes()
    return bytes(out)


def layout_used(qb, counts):
    """set of used slot positions + list of (q, start, end) runs for the canonical layout"""
    byq = {}
    for key, c in counts.items():
        byq.setdefault(key >> 8, []).append((key & 0xff, c))
    free = 0
    runs = []
    for q in sorted(byq):
        start = max(q, free)
        ln = sum(len(encode_counter(r, c)) for r, c in byq[q])
        runs.append((q, start, start + ln - 1))
        free = start + ln
    return runs

def denoise_survivors(qb, counts, min_len=1 << 20):
    """The multiset one reference deNoise round (t=1 schedule, cqf/CQF_mt.h:884-901,
    999-1039) leaves behind: every count>=2 entry, plus the singletons the reference's
    range walk skips: a one-slot cluster sitting exactly on the last slot of a work
    range (`while(start < end_bucket_id)` at CQF_mt.h:1024 / gqf.c:2881 never visits
    a cluster that starts on end_bucket_id)."""
    nslots = 1 << qb
    runs = layout_used(qb, counts)
    used = set()
    for q, s, e in runs:
        used.update(range(s, e + 1))
    occq = sorted(q for q, _, _ in runs)
    import bisect
    def first_empty(x):
        while x in used:
            x += 1
        return x
    def first_occ(x):
        i = bisect.bisect_left(occq, x)
        return occq[i] if i < len(occq) else None
    keep = {k: c for k, c in counts.items() if c >= 2}
    cur = first_occ(0)
    byq1 = {}
    for k, c in counts.items():
        byq1.setdefault(k >> 8, []).append((k, c))
    while cur is not None and cur < nslots:
        end = min(cur + min_len, nslots)
        end = first_empty(end) - 1
        # one-slot cluster starting at `end`
        if end in used and (end - 1) not in used and end in byq1 and len(byq1[end]) == 1 \
                and byq1[end][0][1] == 1:
            keep[byq1[end][0][0]] = 1
        cur = first_occ(end + 1)
    return keep
