"""Order-free comparison of two unitigs.fa texts (the device's and the sequential whole-pipeline oracle's,
oracle/contiger_pipeline.cpp): ids and orientation depend on the schedule in the reference itself
(concurrent_vector::push_back order), so what is compared is
  * the set of canonical sequences min(seq, RC(seq));
  * per unitig: LN, km (the int-truncated median) and KC = km * (LN - k + 1);
  * the link set, canonicalised: every `L:` tag as ((canonical sequence of the source, which end), (canonical sequence of
    the target, which of its ends is entered)), where an end is 'S' (the canonical sequence's first k-mer side) or 'E'.
"""
_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def rc(s):
    return s[::-1].translate(_COMP)


def canon_seq(s, k, circle=False):
    """min(seq, RC(seq)); a pure circle (the walk stopped where the next k-mer would have been the first again,
    contig_assembly.cpp:3176-3183) starts wherever its seed lay, so it is rotated to its smallest rotation over both
    strands first"""
    if circle:
        best = None
        for t in (s, rc(s)):
            body = t[:len(t) - (k - 1)]
            dbl = body + body
            for i in range(len(body)):
                r = dbl[i:i + len(body)]
                if best is None or r < best:
                    best = r
        return best + best[:k - 1]
    return min(s, rc(s))


def is_pure_circle(j, s, links, k):
    """its last k-1 bases repeat its first k-1 and it is linked to nothing but itself (a loop that hangs on a junction
    has the same shape but links to its neighbours, and its ends are real ends)"""
    return len(s) > k - 1 and s[-(k - 1):] == s[:k - 1] and all(t == j for _, t, _ in links)


def parse(fa: bytes, k):
    """[(seq, km, kc, [(side, j, sign), ...])] in file order; checks the record grammar"""
    lines = fa.split(b"\n")
    recs = []
    for h, s in zip(lines[0::2], lines[1::2]):
        if not h:
            continue
        parts = h.split()
        assert parts[0] == b">%d" % len(recs), parts[0]
        f = dict(x.split(b":", 2)[0::2] for x in parts[1:4])
        assert int(f[b"LN"]) == len(s)
        km, kc = int(f[b"km"]), int(f[b"KC"])
        assert kc == km * (len(s) - k + 1)
        links = []
        for x in parts[4:]:
            t = x.split(b":")
            assert t[0] == b"L" and t[1] in (b"+", b"-") and t[3] in (b"+", b"-")
            links.append((t[1], int(t[2]), t[3]))
        recs.append((s, km, kc, links))
    return recs


def canonical(recs, k, drop_invalid=False):
    """({canonical seq: (km, kc)}, set of canonical links, number of links whose target does not overlap the source by
    k-1 bases). The reference leaves map entries of cleared contigs behind (contig_assembly.cpp:3018-3025 vs :935-954), so its
    graph pass can emit links to ids that were never renumbered; such a link fails the overlap test and is counted (and
    left out with drop_invalid) instead of compared."""
    circ = [is_pure_circle(j, s, ls, k) for j, (s, km, kc, ls) in enumerate(recs)]
    cs = [canon_seq(r[0], k, circ[j]) for j, r in enumerate(recs)]
    units = {}
    for j, (s, km, kc, _) in enumerate(recs):
        assert cs[j] not in units, "a sequence is reported twice"
        units[cs[j]] = (km, kc)
    links, invalid = set(), 0
    for j, (s, km, kc, ls) in enumerate(recs):
        c = cs[j]
        fwd = c == s
        for side, t_id, sign in ls:
            if t_id >= len(recs):
                invalid += 1
                continue
            t = recs[t_id][0]
            tt = t if sign == b"+" else rc(t)            # the target as it is entered
            me = s if side == b"+" else rc(s)             # leaving through my last k-1 bases
            if me[-(k - 1):] != tt[:k - 1]:
                invalid += 1
                if drop_invalid:
                    continue
            tc = cs[t_id]
            if circ[j]:                                   # a pure circle's self links: no ends to name
                links.add((c, b"O", tc, b"O"))
                continue
            # which end of the canonical source is left, which end of the canonical target is entered
            src_end = b"E" if (side == b"+") == fwd else b"S"
            t_fwd = tc == t
            tgt_end = b"S" if (sign == b"+") == t_fwd else b"E"
            if tc == rc(tc):
                tgt_end = b"S"                            # a palindromic target has one end
            links.add((c, src_end, tc, tgt_end))
    return units, links, invalid


def _median_int(v):
    """int(median(v)), base/Utility.cpp:27-40 stored into Contig::median_abundance (an int)"""
    v = sorted(v)
    n = len(v)
    if n == 0:
        return 0
    if n == 1:
        return v[0]
    return int((v[n // 2 - 1] + v[n // 2]) / 2.0) if n % 2 == 0 else v[n // 2]


def admissible_km(seq, k, count, seed_kmers):
    """every km value SOME schedule of the reference gives this unitig. Its k-mer counts c[0..n) are fixed by the filter;
    what the schedule decides is who finds it:
      * a queued contig (one forward walk from either end, contig_assembly.cpp:2254-2269): the median of all counts;
      * a read's seed at position p, walking to one end first (processDataChunk :1886-1904): m1 = median of the counts
        from p to that end, then the median of (that many copies of m1) + the counts behind p.
    seed_kmers: the set of k-mers the reads offer as seeds (both seed rules), in read orientation."""
    n = len(seq) - k + 1
    c = [count(seq[i:i + k]) for i in range(n)]
    out = {_median_int(c)}
    for p in range(n):
        km = seq[p:p + k]
        if km in seed_kmers:                      # read orientation = this orientation: to the end first
            m1 = _median_int(c[p:])
            out.add(_median_int([m1] * (n - p) + c[:p]))
        if rc(km) in seed_kmers:                  # the read lies on the other strand: to the start first
            m1 = _median_int(c[:p + 1])
            out.add(_median_int([m1] * (p + 1) + c[p + 1:]))
    return out


def explain_one_sided(extras, common, k, hb, seed_kmers, seq_keys):
    """Unitigs only one side reports must be what the reference itself calls "possible because of hash collisions"
    (contig_assembly.cpp:3082): the traveled bit belongs to the filter ENTRY, so a read's seed k-mer whose key another
    k-mer shares is skipped as "already traveled" when that other k-mer was looked at first, and is walked when it was
    not -- a matter of the schedule (read by read there, batch by batch here). What such a seed alone reaches is then
    found by one schedule only. Accepted: every group of one-sided unitigs (joined by k-1 overlaps) holds a seed k-mer
    whose key is also the key of a k-mer outside the group. seq_keys(seqs, k, hb) -> uint64 keys of all k-mers in order."""
    import numpy as np
    extras = list(extras)
    if not extras:
        return 0
    # groups by k-1 overlap of ends (either strand)
    ends = {}
    for i, s in enumerate(extras):
        for e in (s[:k - 1], s[-(k - 1):], rc(s)[:k - 1], rc(s)[-(k - 1):]):
            ends.setdefault(e, set()).add(i)
    parent = list(range(len(extras)))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    for grp in ends.values():
        g = list(grp)
        for j in g[1:]:
            parent[find(j)] = find(g[0])
    groups = {}
    for i in range(len(extras)):
        groups.setdefault(find(i), []).append(i)
    common_keys = np.sort(seq_keys(list(common), k, hb))
    extra_keys = [seq_keys([s], k, hb) for s in extras]
    for members in groups.values():
        inside = np.concatenate([extra_keys[i] for i in members])
        others = np.concatenate([extra_keys[i] for i in range(len(extras)) if i not in members] + [common_keys[:0]])
        ok = False
        for i in members:
            s = extras[i]
            for p in range(len(s) - k + 1):
                km = s[p:p + k]
                if km in seed_kmers or rc(km) in seed_kmers:
                    key = extra_keys[i][p]
                    j = np.searchsorted(common_keys, key)
                    if (j < len(common_keys) and common_keys[j] == key) or (others == key).any() or (inside == key).sum() > 1:
                        ok = True
        assert ok, "a unitig is reported by one side only and no colliding seed explains it (len %d)" % len(extras[members[0]])
    return len(groups)
