// Filter sizing: the arithmetic of src/CQF-deNoise.cpp:96-161 with mean_CDF2deNoise
// (cqf/CQF_mt.h:94-133) and true2falseKmer_DP (cqf/true2falseKmer_DP.cpp:12-50).
// boost::math's Poisson CDF is replaced by a summed pmf (long double).
#pragma once
#include <math.h>
#include <stdint.h>
#include <fstream>
#include <string>
#include <vector>

namespace shk {

inline double poisson_cdf(double mean, double x) {
  if (x < 0) return 0;
  long kmax = (long)floor(x);
  long double s = 0;
  for (long i = 0; i <= kmax; i++) s += expl(-(long double)mean + i * logl((long double)mean) - lgammal((long double)i + 1));
  return (double)(s > 1 ? 1 : s);
}

inline int mean_CDF2deNoise(double mean, double cdf_desired) {
  int start = 0, end = (int)(mean + 1), mid;
  const double cdf0 = poisson_cdf(mean, 0);
  auto cdf_positive = [&](double x) { return (poisson_cdf(mean, x) - cdf0) / (1 - cdf0); };
  while (cdf_positive(end) < cdf_desired) end *= 2;
  while (start <= end) {
    if (start == end) return start;
    else if (start + 1 == end) {
      double t1 = cdf_positive(start), t2 = cdf_positive(end);
      if (t2 <= cdf_desired) return end;
      else if (t1 <= cdf_desired) return start;
      else return start - 1 > 0 ? start - 1 : 0;
    }
    mid = (start + end) / 2;
    double cdf = cdf_positive(mid);
    if (cdf < cdf_desired) start = mid + 1;
    else if (cdf > cdf_desired) end = mid - 1;
    else return start;
  }
  return start;
}

inline double true2falseKmer_DP(const std::string &errorFile, size_t K) {
  std::vector<double> e;
  std::ifstream fin(errorFile);
  double tmp;
  while (fin >> tmp) e.push_back(tmp);
  const size_t seq_len = e.size();
  std::vector<double> DP(K + 1, 0), nDP(K + 1, 0);
  tmp = 1;
  for (size_t x = 0; x < K; x++) tmp *= (1 - e[x]);
  DP[0] = tmp;
  for (size_t x = 1; x <= K; x++) {
    tmp = e[x - 1];
    for (size_t y = x; y < K; y++) tmp *= (1 - e[y]);
    DP[x] = tmp;
  }
  double trueP = DP[0];
  for (size_t x = K; x < seq_len; x++) {
    nDP[0] = DP[0] * (1 - e[x]);
    for (size_t y = 1; y <= K; y++) nDP[y - 1] += DP[y] * (1 - e[x]);
    nDP[K] = e[x];
    trueP += nDP[0];
    DP = nDP;
    nDP.assign(K + 1, 0);
  }
  return trueP / ((double)(seq_len - K + 1) - trueP);
}

struct Sizing {
  uint64_t num_true_kmers, num_false_kmers, qb, hb, n_distinct_elts_for_DeNoise;
  int num_deNoise, lower_bound, upper_bound;
  double fr;
};

inline Sizing size_filter(int K, uint64_t n_true_kmers, uint64_t total_kmers, double alpha, const std::string &errorProfile,
                          int num_deNoise, double fr) {
  Sizing s;
  uint64_t num_true_kmers, num_false_kmers, num_slots;
  if (alpha == -1) {
    double t = true2falseKmer_DP(errorProfile, K);
    num_true_kmers = (uint64_t)(total_kmers * t / (1 + t));
  } else {
    num_true_kmers = (uint64_t)(total_kmers * pow(1 - alpha, K));
  }
  num_false_kmers = total_kmers - num_true_kmers;
  if (num_deNoise < 0) {
    if (!fr) fr = 1.0 / n_true_kmers;
    num_deNoise = mean_CDF2deNoise((double)(num_true_kmers / n_true_kmers), fr);
  }
  int enc = 0;
  uint64_t tmp = num_true_kmers / n_true_kmers + 1;
  while (tmp) { tmp >>= 7; enc++; }
  auto nslots = [&](uint64_t d) { return (uint64_t)(n_true_kmers * (enc + (double)3 / 2) + num_false_kmers * 10 / ((d + 1) * 9)); };
  num_slots = nslots(num_deNoise);
  uint64_t qb = 1, base = 2;
  while (base < num_slots) { qb++; base <<= 1; }
  uint64_t ub = num_deNoise, lb, st = num_slots;
  while (num_deNoise && st < (1ULL << qb)) { num_deNoise--; st = nslots(num_deNoise); }
  if (st >= (1ULL << qb)) num_deNoise++;
  s.n_distinct_elts_for_DeNoise = n_true_kmers + num_false_kmers / (num_deNoise + 1);
  lb = num_deNoise;
  st = (uint64_t)(n_true_kmers * (enc + (double)3 / 2));
  if (st > (1ULL << (qb - 1))) ub = 0;
  else {
    st = num_slots;
    while (st >= (1ULL << (qb - 1))) { ub++; st = nslots(ub); }
    if (st < (1ULL << (qb - 1))) ub--;
  }
  s.num_true_kmers = num_true_kmers; s.num_false_kmers = num_false_kmers; s.qb = qb; s.hb = qb + 8;
  s.num_deNoise = num_deNoise; s.lower_bound = (int)lb; s.upper_bound = (int)ub; s.fr = fr;
  return s;
}

}  // namespace shk
