"""Build planning for bench.py and the tests: the reference's sizing arithmetic and a model of how
full the filter gets under that sizing.

sizing()            src/CQF-deNoise.cpp:96-161 (qb, deNoise rounds, trigger) for given -k -n -N -e
predict_build()     expected occupied slots chunk by chunk for the synthetic read model bench.py
                    generates (uniform genome, uniform substitution errors), following the reference's
                    t = 1 deNoise schedule (CQF_mt.h:837-869)
plan_build()        the -N the bench passes: the exact number of k-mers it will present, with head room
                    added in 2 % steps until the predicted peak load is at most `max_load`

Why head room is needed at all: the reference's formula budgets `n * (enc + 1.5)` slots for the true
k-mers and ONE slot per false k-mer between two rounds, and then lowers the number of rounds until the
sum just fits 2^qb. False k-mers that occur twice inside one interval survive every later round (two
slots each, and they keep counting towards the trigger), so on data with recurring errors the rounds
come a little early, the last interval is longer than planned and the table overflows -- silently in
the reference (gqf.c has no capacity check), as SHK_ERR_TABLE_FULL here.
"""
import math


def poisson_cdf(x, mean):
    from scipy.stats import poisson
    return float(poisson.cdf(x, mean))


def mean_cdf2denoise(mean, fr):
    """cqf/CQF_mt.h:94-133: rounds such that a true k-mer (zero-truncated Poisson(mean) occurrences)
    is lost with probability <= fr. Same search, same tie rules."""
    cdf0 = poisson_cdf(0, mean)

    def cdfp(x):
        return (poisson_cdf(x, mean) - cdf0) / (1 - cdf0)
    start, end = 0, int(mean + 1)
    while cdfp(end) < fr:
        end *= 2
    while start <= end:
        if start == end:
            return start
        if start + 1 == end:
            t1, t2 = cdfp(start), cdfp(end)
            return end if t2 <= fr else (start if t1 <= fr else max(start - 1, 0))
        mid = (start + end) // 2
        c = cdfp(mid)
        if c < fr:
            start = mid + 1
        elif c > fr:
            end = mid - 1
        else:
            return start
    return start


def sizing(K, n_true, N_total, alpha, fr=0.0, ratio=None):
    """(qb, rounds, trigger) as src/CQF-deNoise.cpp:96-161 computes them. `ratio` = true:false k-mer
    ratio from an error profile (true2falseKmer_DP) replaces alpha when given."""
    if ratio is not None:
        num_true = int(N_total * ratio / (1 + ratio))
    else:
        num_true = int(N_total * (1 - alpha) ** K)
    num_false = N_total - num_true
    if not fr:
        fr = 1.0 / n_true
    nd = mean_cdf2denoise(float(num_true // n_true), fr)
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


def xnslots(qb, num_shards=1):
    """qf_init geometry, gqf.c:2197-2198 (every shard keeps a full-size overflow tail)"""
    g = 1 << qb
    return g // num_shards + int(10 * math.sqrt(float(g)))


def _entry_slots(count):
    """expected slots of one entry with this (mean) count under encode_counter (gqf.c:1225-1255) for a
    uniformly distributed 8-bit remainder: remainder + base-128 digits of count-1 + an escape zero when
    the top digit (|0x80 when there are several) exceeds the remainder"""
    if count < 1.5:
        return 1.0
    c = max(1, int(round(count)) - 1)
    nd, t = 1, c >> 7
    while t:
        nd += 1
        t >>= 7
    top = (c >> (7 * (nd - 1))) & 0x7F
    if nd > 1:
        top |= 0x80
    return 1.0 + nd + top / 256.0


_SLOT_GRID = {}


def _mean_entry_slots(mean):
    """_mean_entry_slots_exact on a grid of integer means, linearly interpolated (the simulation asks thousands of times)"""
    a = int(mean)
    for m in (a, a + 1):
        if m not in _SLOT_GRID:
            _SLOT_GRID[m] = _mean_entry_slots_exact(float(m))
    return _SLOT_GRID[a] + (mean - a) * (_SLOT_GRID[a + 1] - _SLOT_GRID[a])


def _mean_entry_slots_exact(mean):
    """E[_entry_slots(X)] for X ~ Poisson(mean) given X >= 2: the counts of a class spread around their mean, and
    the encoding grows by a slot at 129, 16385, ... (a class whose mean count is 125 has 40 % of its entries beyond 128)"""
    if mean <= 2.0:
        return _entry_slots(2)
    sd = math.sqrt(mean)
    lo, hi = max(2, int(mean - 6 * sd)), int(mean + 6 * sd) + 2
    logm = math.log(mean)
    num = den = 0.0
    for x in range(lo, hi + 1):
        w = math.exp(x * logm - mean - math.lgamma(x + 1.0))
        num += w * _entry_slots(x)
        den += w
    return num / den if den > 0 else _entry_slots(mean)


class _Class:
    """a population of `size` possible k-mers that each arrive at the same Poisson rate:
    p1 = fraction sitting in the table with count 1, p2 = with count >= 2 (survives deNoise)"""

    def __init__(self, size):
        self.size, self.p1, self.p2, self.mass2 = float(size), 0.0, 0.0, 0.0

    def arrive(self, rate):
        if self.size <= 0 or rate <= 0:
            return
        e = math.exp(-rate)
        p0 = 1.0 - self.p1 - self.p2
        once = p0 * rate * e
        twice = p0 * (1.0 - e - rate * e)
        up = self.p1 * (1.0 - e)
        # occurrences carried by the count >= 2 entries (for their mean count): arrivals to those already
        # there; 1 + X for a singleton that is hit (E[X; X >= 1] = rate); E[X; X >= 2] = rate (1 - e^-rate)
        self.mass2 += self.p2 * rate + self.p1 * ((1.0 - e) + rate) + p0 * rate * (1.0 - e)
        self.p1 += once - up
        self.p2 += twice + up

    def denoise(self):
        self.p1 = 0.0

    def distinct(self):
        return self.size * (self.p1 + self.p2)

    def slots(self):
        mean2 = self.mass2 / self.p2 if self.p2 > 0 else 2.0
        return self.size * (self.p1 + self.p2 * _mean_entry_slots(mean2))


def predict_build(K, G, L, err, kmers_per_chunk, nchunks, trigger, rounds, trace_every=0):
    """Expected course of a build over `nchunks` chunks of `kmers_per_chunk` k-mers read from a
    uniform-random genome of G bases (reads of length L at uniform positions, either strand,
    independent substitution errors of rate `err`): returns a dict with the peak of occupied slots,
    the chunk it is reached in, the rounds fired and the slots / distinct count at the end.
    Classes: the G true k-mers; their 3K single-error neighbours (these recur at high coverage); all
    k-mers with two or more errors (practically never seen twice: always singletons)."""
    ok = (1.0 - err) ** K
    true = _Class(G)
    one = _Class(3.0 * K * G)
    r_true = kmers_per_chunk * ok / G
    r_one = kmers_per_chunk * (err / 3.0) * (1.0 - err) ** (K - 1) / G
    multi_per_chunk = kmers_per_chunk * (1.0 - ok - K * err * (1.0 - err) ** (K - 1))
    multi = 0.0
    left, fired = rounds, 0
    peak, peak_chunk = 0.0, 0
    trace = []
    for c in range(nchunks):
        true.arrive(r_true)
        one.arrive(r_one)
        multi += multi_per_chunk
        used = true.slots() + one.slots() + multi
        if used > peak:
            peak, peak_chunk = used, c
        if left and true.distinct() + one.distinct() + multi >= trigger:
            left -= 1
            fired += 1
            true.denoise()
            one.denoise()
            multi = 0.0
        if trace_every and (c + 1) % trace_every == 0:
            trace.append((true.slots() + one.slots() + multi, true.distinct() + one.distinct() + multi, fired))
    return {"trace": trace, "peak_slots": peak, "peak_chunk": peak_chunk, "rounds_fired": fired,
            "end_slots": true.slots() + one.slots() + multi,
            "end_distinct": true.distinct() + one.distinct() + multi}


def plan_build(K, G, L, err, kmers_per_chunk, nchunks, n_true=None, world=1, max_load=0.95, max_rounds=4096):
    """The build bench.py times. The README recipe gives the command line: -N = the k-mers the run
    presents (exact), -n = n_true (default: the genome's G - K + 1 k-mers, "the size of the genome as
    the approximation", README.md:84), -e = err; src/CQF-deNoise.cpp:96-161 turns them into (qb, rounds,
    trigger). That formula fills 2^qb to the edge by construction and does not budget k-mers with recurring
    errors (module docstring), so the course of the build is predicted first; when the predicted peak
    exceeds `max_load` of the 2^qb slots, rounds are ADDED (trigger = n + false / (rounds + 1), the
    formula's own expression) until it does not -- the table keeps the reference's size, the build does
    more deNoise work than the formula asked for, never less."""
    n_true = n_true or (G - K + 1)
    N = int(kmers_per_chunk * nchunks)
    qb, nd0, trig0 = sizing(K, n_true, N, err)
    num_false = N - int(N * (1 - err) ** K)
    nd, trigger = nd0, trig0
    while True:
        p = predict_build(K, G, L, err, kmers_per_chunk, nchunks, trigger, nd)
        load = p["peak_slots"] / float(1 << qb)
        if load <= max_load or nd >= max_rounds:
            break
        nd += 1
        trigger = n_true + num_false // (nd + 1)
    return {"N": N, "n": n_true, "e": err, "qb": qb, "formula_rounds": nd0, "formula_trigger": trig0, "rounds": nd,
            "trigger": trigger, "predicted_peak_slots": int(p["peak_slots"]), "predicted_peak_load": load,
            "predicted_rounds_fired": p["rounds_fired"], "xnslots": xnslots(qb, world)}
