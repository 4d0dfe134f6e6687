// Filter sizing for the CQF-deNoise command line: how many quotient bits, how many deNoise rounds and at which
// distinct count a round fires, from -k -n -N and either -e or --errorProfile. The numbers must be the reference's
// (src/CQF-deNoise.cpp:96-161, cqf/CQF_mt.h:84-133, cqf/true2falseKmer_DP.cpp:12-50) because qb and the trigger
// shape the .cqf; every floating-point expression below therefore keeps the reference's operand order, while the
// code around them is organised by what is computed: ZeroTruncatedPoisson (occurrences of a true k-mer),
// rounds_for_loss_rate (its search), ErrorProfile (error-free windows of a read), FilterSizer (slots -> qb, rounds).
// boost::math's Poisson CDF is replaced by a log-space sum of the probability mass function in long double.
#pragma once
#include <math.h>
#include <stdint.h>
#include <fstream>
#include <string>
#include <vector>

namespace shk {

// Occurrences of one true k-mer in the data: Poisson(mean), conditioned on being seen at all.
class ZeroTruncatedPoisson {
 public:
  explicit ZeroTruncatedPoisson(double mean) : mean_(mean), at_zero_(below_or_at(0)) {}
  // P[X <= x] of the untruncated distribution
  double below_or_at(double x) const {
    if (x < 0) return 0;
    const long double m = (long double)mean_, log_m = logl(m);
    long double sum = 0;
    for (long i = 0, last = (long)floor(x); i <= last; i++) sum += expl(i * log_m - m - lgammal((long double)i + 1));
    return (double)(sum > 1 ? 1 : sum);
  }
  // P[X <= x | X >= 1], the expression of cqf/CQF_mt.h:101-103
  double seen_at_most(double x) const { return (below_or_at(x) - at_zero_) / (1 - at_zero_); }

 private:
  double mean_, at_zero_;
};

// A k-mer seen c times in total is lost when a round catches it at count 1; with d rounds spread over the data the
// reference bounds that by P[X <= d | X >= 1] <= fr and looks for the largest such d with the bisection of
// cqf/CQF_mt.h:94-133. The bisection's corner rules (which end of a two-element window wins, what an exact hit
// returns, a window that closes on an unchecked value) decide the result in edge cases, so they are kept as a table.
inline int rounds_for_loss_rate(double mean, double fr) {
  const ZeroTruncatedPoisson occ(mean);
  struct Window { int lo, hi; } w = {0, (int)(mean + 1)};
  while (occ.seen_at_most(w.hi) < fr) w.hi *= 2;        // grow until the upper end is too many rounds
  for (;;) {
    if (w.lo > w.hi) return w.lo;                        // closed on an empty window
    if (w.lo == w.hi) return w.lo;                       // one candidate left: taken unchecked
    if (w.lo + 1 == w.hi) {
      const double at_lo = occ.seen_at_most(w.lo), at_hi = occ.seen_at_most(w.hi);
      if (at_hi <= fr) return w.hi;
      if (at_lo <= fr) return w.lo;
      return w.lo > 0 ? w.lo - 1 : 0;
    }
    const int mid = (w.lo + w.hi) / 2;
    const double at_mid = occ.seen_at_most(mid);
    if (at_mid < fr) w.lo = mid + 1;
    else if (at_mid > fr) w.hi = mid - 1;
    else return w.lo;                                    // exact hit: the reference answers with the window's lower end
  }
}

// Per-base error rates of a read (one number per line of --errorProfile).
class ErrorProfile {
 public:
  static ErrorProfile from_file(const std::string &path) {
    ErrorProfile p;
    std::ifstream in(path);
    for (double v; in >> v;) p.rate_.push_back(v);
    return p;
  }
  explicit ErrorProfile(std::vector<double> rates = {}) : rate_(std::move(rates)) {}
  size_t read_length() const { return rate_.size(); }

  // Expected error-free K-windows of a read divided by the expected erroneous ones. The reference tracks, window
  // by window, where the most recent error sits ("no error in the last K bases", or at offset 0..K-1 of the
  // window); sliding by one base ages every state by one and the oldest merges into "no error". Here the states
  // live in a ring whose origin moves instead of being copied; products and sums are formed in the reference's
  // order (true2falseKmer_DP.cpp:25-47), so the doubles are identical.
  double true_to_false_ratio(size_t K) const {
    const size_t len = rate_.size();
    std::vector<double> clean_after(K + 1, 0.0);   // clean_after[a]: error at base a-1 of the first window, none behind it (a >= 1)
    double clean = 1;
    for (size_t x = 0; x < K; x++) clean *= (1 - rate_[x]);
    for (size_t a = 1; a <= K; a++) {
      double p = rate_[a - 1];
      for (size_t y = a; y < K; y++) p *= (1 - rate_[y]);
      clean_after[a] = p;
    }
    // ring[(origin + a) % K] holds state a = 1..K; state 0 ("clean") is kept apart
    std::vector<double> ring(K, 0.0);
    for (size_t a = 1; a <= K; a++) ring[a % K] = clean_after[a];
    size_t origin = 0;
    double expected_clean = clean;
    for (size_t x = K; x < len; x++) {
      const double ok = 1 - rate_[x];
      const size_t oldest = (origin + 1) % K;        // state 1: its error leaves the window now
      double next_clean = clean * ok;
      next_clean += ring[oldest] * ok;
      for (size_t a = 2; a <= K; a++) {              // states 2..K age by one (0 + v*ok == v*ok exactly)
        double &v = ring[(origin + a) % K];
        v = 0.0 + v * ok;
      }
      ring[oldest] = rate_[x];                       // the new base's own error becomes state K of the next window
      origin = (origin + 1) % K;
      clean = next_clean;
      expected_clean += clean;
    }
    return expected_clean / (len - K + 1 - expected_clean);
  }

 private:
  std::vector<double> rate_;
};

inline double true2falseKmer_DP(const std::string &errorFile, size_t K) {   // the reference's name for it
  return ErrorProfile::from_file(errorFile).true_to_false_ratio(K);
}

struct Sizing {
  uint64_t num_true_kmers, num_false_kmers, qb, hb, n_distinct_elts_for_DeNoise;
  int num_deNoise, lower_bound, upper_bound;
  double fr;
};

// Slots the build needs as a function of the number of rounds, and what follows from it.
class FilterSizer {
 public:
  FilterSizer(uint64_t n_true, uint64_t total, uint64_t num_true) : n_(n_true), true_(num_true), false_(total - num_true) {
    // slots of a true k-mer's counter: 7-bit groups of (mean occurrences + 1), src/CQF-deNoise.cpp:118-123
    for (uint64_t v = true_ / n_ + 1; v; v >>= 7) counter_slots_++;
  }
  uint64_t num_false() const { return false_; }
  double mean_occurrences() const { return (double)(true_ / n_); }
  // true k-mers: remainder + counter + half a slot; false k-mers: those of one interval, with a ninth on top.
  // (double product + integer quotient, truncated together: src/CQF-deNoise.cpp:124)
  uint64_t slots(uint64_t rounds) const { return (uint64_t)(n_ * (counter_slots_ + (double)3 / 2) + false_ * 10 / ((rounds + 1) * 9)); }
  uint64_t true_slots() const { return (uint64_t)(n_ * (counter_slots_ + (double)3 / 2)); }
  static uint64_t quotient_bits(uint64_t slots) {
    uint64_t qb = 1;
    for (uint64_t cap = 2; cap < slots; cap <<= 1) qb++;
    return qb;
  }
  // fewest rounds (not more than `rounds`) whose slots still fit 2^qb: the filter is filled to the edge on purpose
  int fewest_rounds_fitting(int rounds, uint64_t qb) const {
    uint64_t need = slots(rounds);
    while (rounds && need < (1ULL << qb)) need = slots(--rounds);
    return need >= (1ULL << qb) ? rounds + 1 : rounds;
  }
  // most rounds before the build would fit a filter of half the size (0 = never: the true k-mers alone need more)
  int most_rounds_same_size(int rounds, uint64_t qb) const {
    if (true_slots() > (1ULL << (qb - 1))) return 0;
    uint64_t need = slots(rounds);
    while (need >= (1ULL << (qb - 1))) need = slots(++rounds);
    return rounds - 1;
  }
  uint64_t trigger(int rounds) const { return n_ + false_ / (rounds + 1); }

 private:
  uint64_t n_, true_, false_;
  int counter_slots_ = 0;
};

inline Sizing size_filter(int K, uint64_t n_true_kmers, uint64_t total_kmers, double alpha, const std::string &errorProfile,
                          int num_deNoise, double fr) {
  uint64_t num_true;
  if (alpha == -1) {
    const double ratio = true2falseKmer_DP(errorProfile, K);
    num_true = (uint64_t)(total_kmers * ratio / (1 + ratio));
  } else {
    num_true = (uint64_t)(total_kmers * pow(1 - alpha, K));
  }
  const FilterSizer fs(n_true_kmers, total_kmers, num_true);
  Sizing s;
  s.fr = fr;
  if (num_deNoise < 0) {
    if (!s.fr) s.fr = 1.0 / n_true_kmers;
    num_deNoise = rounds_for_loss_rate(fs.mean_occurrences(), s.fr);
  }
  s.num_true_kmers = num_true;
  s.num_false_kmers = fs.num_false();
  s.qb = FilterSizer::quotient_bits(fs.slots(num_deNoise));
  s.hb = s.qb + 8;
  s.upper_bound = fs.most_rounds_same_size(num_deNoise, s.qb);
  s.num_deNoise = s.lower_bound = fs.fewest_rounds_fitting(num_deNoise, s.qb);
  s.n_distinct_elts_for_DeNoise = fs.trigger(s.num_deNoise);
  return s;
}

}  // namespace shk
