// CQF-deNoise command line: the reference's flags and defaults (src/CQF-deNoise.cpp:18-51,
// README.md:46-66) on the GPU path. boost::program_options is replaced by a small parser;
// the settings banner and the counters printed follow src/CQF-deNoise.cpp:185-221.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "cqf_mt.hpp"
#include "sizing.hpp"

using namespace std;
using namespace shk;

static void usage(const char *argv0) {
  cerr << endl << argv0 << "  <options>\nOptions:\n"
       << "  -h [ --help ]            print help messages\n"
       << "  -k arg                   k-mer size\n"
       << "  -n [ --trueKmer ] arg    number of unique true k-mers\n"
       << "  -N arg                   total number of k-mers\n"
       << "  -e [ --alpha ] arg (=-1) average base error rate, when specified, the <errorProfile> is ignored\n"
       << "  --errorProfile arg       error profile in a file, each line with error rate for the corresponding base\n"
       << "  --fr arg (=0)            tolerable rate of true k-mers being wrongly removed, default: 1/<trueKmer>\n"
       << "  --deNoise arg (=-1)      number of rounds of deNoise, when specified, the <fr> is ignored\n"
       << "  --endDeNoise             call deNoise after processing all the k-mers (not counted into the <deNoise>)\n"
       << "  -t arg (=16)             number of threads (kept for compatibility; the GPU does the work)\n"
       << "  -f [ --format ] arg      format of the input: g(gzip); b(bzip2); f(plain fastq)\n"
       << "  -i [ --input ] arg       a file containing a list of read file name(s), should be in the same directory as the fastq file(s)\n"
       << "  -o [ --output ] arg      output file name\n"
       << "  (test hooks, not in the reference: --qb N --trigger N --rounds N --part-size N --overhead N --min-denoise-len N --device N)\n\n";
}

int main(int argc, char *argv[]) {
  int K = -1, num_deNoise = -1, thread_num = 16, device = 0;
  uint64_t n_true_kmers = 0, total_kmers = 0;
  double alpha = -1, fr = 0;
  string errorProfile, flist, output_file;
  bool end_deNoise = false, have_n = false, have_N = false;
  char fmt = 0;
  long long qb_override = -1, trigger_override = -1, rounds_override = -1, part_size = 1LL << 23, overhead = 65535, min_len = 0;
  if (argc == 1) { usage(argv[0]); return 0; }
  for (int i = 1; i < argc; i++) {
    string a = argv[i];
    auto val = [&]() -> string {
      size_t eq = a.find('=');
      if (a.rfind("--", 0) == 0 && eq != string::npos) return a.substr(eq + 1);
      if (i + 1 >= argc) { cerr << "missing value for " << a << endl; exit(0); }
      return argv[++i];
    };
    string name = a.substr(0, a.find('='));
    if (name == "-h" || name == "--help") { usage(argv[0]); return 0; }
    else if (name == "-k") K = atoi(val().c_str());
    else if (name == "-n" || name == "--trueKmer") { n_true_kmers = strtoull(val().c_str(), 0, 10); have_n = true; }
    else if (name == "-N") { total_kmers = strtoull(val().c_str(), 0, 10); have_N = true; }
    else if (name == "-e" || name == "--alpha") alpha = atof(val().c_str());
    else if (name == "--errorProfile") errorProfile = val();
    else if (name == "--fr") fr = atof(val().c_str());
    else if (name == "--deNoise") num_deNoise = atoi(val().c_str());
    else if (name == "--endDeNoise") end_deNoise = true;
    else if (name == "-t") thread_num = atoi(val().c_str());
    else if (name == "-f" || name == "--format") fmt = val()[0];
    else if (name == "-i" || name == "--input") flist = val();
    else if (name == "-o" || name == "--output") output_file = val();
    else if (name == "--qb") qb_override = atoll(val().c_str());
    else if (name == "--trigger") trigger_override = atoll(val().c_str());
    else if (name == "--rounds") rounds_override = atoll(val().c_str());
    else if (name == "--part-size") part_size = atoll(val().c_str());
    else if (name == "--overhead") overhead = atoll(val().c_str());
    else if (name == "--min-denoise-len") min_len = atoll(val().c_str());
    else if (name == "--device") device = atoi(val().c_str());
    else { cerr << "unrecognised option " << a << endl; usage(argv[0]); return 0; }
  }
  if (K < 0 || !have_n || !have_N || !fmt || flist.empty()) { usage(argv[0]); return 0; }
  if (alpha == -1 && errorProfile == "") {
    cerr << endl << "Please specify either <alpha> or <errorProfile>" << endl << endl;
    usage(argv[0]);
    return 0;
  }

  // list of read files, relative to the list's directory (src/CQF-deNoise.cpp:59-81)
  vector<string> files;
  string file_prefix = "";
  auto pos = flist.find_last_of("/\\");
  if (pos != string::npos) file_prefix = flist.substr(0, pos + 1);
  ifstream fin(flist);
  if (!fin.is_open()) { cerr << "Failed to open file: " << flist << endl; return 0; }
  string line;
  while (getline(fin, line)) { if (line.empty()) continue; files.push_back(file_prefix + line); }

  Sizing sz = size_filter(K, n_true_kmers, total_kmers, alpha, errorProfile, num_deNoise, fr);
  uint64_t qb = qb_override > 0 ? (uint64_t)qb_override : sz.qb;
  uint64_t hb = qb + 8;
  uint64_t trigger = trigger_override >= 0 ? (uint64_t)trigger_override : sz.n_distinct_elts_for_DeNoise;
  num_deNoise = rounds_override >= 0 ? (int)rounds_override : sz.num_deNoise;
  if (output_file.empty()) output_file = "k" + to_string(K) + ".t" + to_string(thread_num) + ".s" + to_string(qb) + ".ser";

  FILE_MODE ftype;
  if (fmt == 'g') ftype = GZIP; else if (fmt == 'b') ftype = BZIP2; else if (fmt == 'f') ftype = TEXT;
  else { cerr << "Unrecognized file type " << fmt << endl << "run following to get help" << endl << "\t" << argv[0] << " --help" << endl; return 0; }

  cerr << "CQF-deNoise settings:" << endl << "qb: " << qb << endl << "hb: " << hb << endl << "thread_num: " << thread_num << endl
       << "K: " << K << endl << "number of true k-mers: " << n_true_kmers << endl << "tolerable wrong removal rate: " << sz.fr << endl
       << "number of deNoise rounds: " << num_deNoise << endl
       << "deNoise after processing all k-mers: " << (end_deNoise ? "true" : "false") << endl
       << "number of unique k-mers triggering deNoise: " << trigger << endl;
  cerr << "#deNoise rounds leading to the same size of CQF: [" << sz.lower_bound << ", "
       << (sz.upper_bound == 0 ? string("+oo") : to_string(sz.upper_bound)) << "]" << endl << endl;

  try {
    CQF_mt cqf_mt(qb, hb, (uint16_t)thread_num, 2038074761);   // seed: src/CQF-deNoise.cpp:83
    cqf_mt.part_size = (uint64_t)part_size; cqf_mt.overhead = (uint32_t)overhead; cqf_mt.min_denoise_len = (uint64_t)min_len;
    cqf_mt.device = device;
    time_t start_time = time(NULL);
    cerr << "Start to build K-mer spectrum..." << endl;
    cqf_mt.build_KmerSpectrum(files, FASTQ, ftype, K, n_true_kmers, trigger, (uint32_t)num_deNoise, end_deNoise, sz.fr);
    cqf_mt.save(output_file);
    cerr << "Finished building K-mer spectrum!" << endl;
    cerr << "nelts: " << cqf_mt.nelts() << " ndistinct_elts: " << cqf_mt.ndistinct_elts() << " deNoise rounds: "
         << cqf_mt.denoise_rounds_done << " removed: " << cqf_mt.removed_total << endl;
    cerr << "Time for building K-mer spectrum: " << difftime(time(NULL), start_time) << " seconds." << endl;
  } catch (const std::exception &e) {
    cerr << "CQF-deNoise: " << e.what() << endl;
    return 1;
  }
  return 0;
}
