"""Definition-level checks of a unitig set against the filter it was built from -- independent of anyone's reading of
get_unitig_forward. With "solid" = filter count >= abundance_min (the reference's only notion of a graph node,
src/contig_assembly.cpp:3074, 3109) and edges = k-1 overlaps, a correct Contiger output is the compacted de Bruijn
graph of the solid k-mers reachable from the seeds:

  1. every k-mer of every unitig is solid;
  2. the unitigs' k-mers (canonical) are exactly the solid k-mers reachable from the seeds, each exactly once.
     "Reachable" follows what the reference explores: solid successors and predecessors, and the solid k-mers that share
     the (k-1)-mer at either end -- get_unitig_forward queues every solid k-mer "with RC(current_kmer_fix) as prefix"
     (:3090-3120, :3147-3160) whether or not the two have a solid successor in common;
  3. inside a unitig every junction is one-in / one-out; both ends of every unitig are a branch (the end has several
     solid successors, or its one successor has another solid predecessor), a dead end, or the junction that closes a
     pure circle -- i.e. no unitig could be extended.

`count(kmer bytes) -> int` is the checker's filter lookup (the oracle's table, or the reference's)."""

_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def rc(s):
    return s[::-1].translate(_COMP)


def canon(s):
    r = rc(s)
    return s if s <= r else r


class Graph:
    def __init__(self, count, k, amin):
        self.count, self.k, self.amin = count, k, amin
        self._solid = {}

    def solid(self, km):
        c = canon(km)
        v = self._solid.get(c)
        if v is None:
            v = self._solid[c] = self.count(c) >= self.amin
        return v

    def succ(self, km):
        return [km[1:] + bytes([x]) for x in b"ACGT" if self.solid(km[1:] + bytes([x]))]

    def pred(self, km):
        return [bytes([x]) + km[:-1] for x in b"ACGT" if self.solid(bytes([x]) + km[:-1])]

    def reachable(self, seeds):
        seen, todo = set(), [s for s in seeds if self.solid(s)]
        while todo:
            km = todo.pop()
            c = canon(km)
            if c in seen:
                continue
            seen.add(c)
            todo.extend(self.succ(km))
            todo.extend(self.pred(km))
            todo.extend(z for z in (bytes([x]) + km[1:] for x in b"ACGT") if self.solid(z))     # same k-1 suffix
            todo.extend(z for z in (km[:-1] + bytes([x]) for x in b"ACGT") if self.solid(z))    # same k-1 prefix
        return seen


def check(unitigs, graph, seeds=None, sample=None, key=None, window=None):
    """unitigs: list of sequences (bytes). seeds: k-mers the build started from (None: skip the coverage half of 2).
    sample: check invariants 1 and 3 on this many unitigs only (2 needs all of them and is skipped then).
    key: the filter key of a k-mer (canonical hash mod range). Give it when the build ran the traveled-bit protocol on
    window: on long unitigs look at the inner junctions of the first and last `window` k-mers only.
    seeds taken from reads: the traveled bit belongs to the filter ENTRY, so a seed whose key another reachable k-mer
    shares may have been skipped as "already traveled" (the reference notes "possible because of hash collisions",
    :3082) -- the unitigs must then cover everything reachable from the seeds with unshared keys, and nothing that is
    not reachable from all of them."""
    k = graph.k
    todo = unitigs
    if sample is not None and sample < len(unitigs):
        import random
        todo = random.Random(1).sample(unitigs, sample)
    for s in todo:
        assert len(s) >= k
        kms = [s[i:i + k] for i in range(len(s) - k + 1)]
        inner = list(zip(kms, kms[1:]))
        if window is not None and len(kms) > 2 * window:
            inner = inner[:window] + inner[-window:]
            probe = kms[:window] + kms[-window:]
        else:
            probe = kms
        for km in probe:
            assert graph.solid(km), "a unitig holds a k-mer below the abundance threshold"
        for a, b in inner:                      # inner junctions: the only way on, the only way back
            assert graph.succ(a) == [b] and graph.pred(b) == [a], "a unitig runs through a branch"
        circle = len(kms) > 1 and kms[-1][1:] == kms[0][:-1] and graph.succ(kms[-1]) == [kms[0]] and graph.pred(kms[0]) == [kms[-1]]
        if circle:
            continue
        for end, nxt, back in ((kms[-1], graph.succ, graph.pred), (rc(kms[0]), graph.succ, graph.pred)):
            out = nxt(end)
            if len(out) == 1:
                # a single way on: then that successor must have another solid predecessor (else the unitig is not maximal)
                assert len(back(out[0])) > 1 or canon(out[0]) == canon(end), "a unitig stops although its extension is unambiguous"
    if sample is not None and sample < len(unitigs):
        return
    seen = {}
    for s in unitigs:
        for i in range(len(s) - k + 1):
            c = canon(s[i:i + k])
            assert c not in seen, "a solid k-mer lies in two unitigs (or twice in one)"
            seen[c] = 1
    if seeds is not None:
        want = graph.reachable(seeds)
        if key is None:
            assert set(seen) == want, (len(seen), len(want))
        else:
            owners = {}
            for c in want:
                owners.setdefault(key(c), set()).add(c)
            sure = [s for s in seeds if len(owners.get(key(canon(s)), ())) <= 1]
            must = graph.reachable(sure)
            assert must <= set(seen) <= want, (len(must), len(seen), len(want))
