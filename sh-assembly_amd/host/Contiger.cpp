// Contiger command line, first slice: the reference's flags and defaults (src/contig_assembly.cpp:26-37) on the
// GPU path. Loads the .cqf built by CQF-deNoise with the same k, walks the read files chunk by chunk with the
// reference's chunker, takes every read's seed k-mer (shk_select_seeds), extends the seeds on the device and
// follows branches (shk_unitigs_add_seeds), and writes unitigs.fa. Seeds are taken one chunk at a time by default: the
// traveled bits set while a chunk's unitigs are walked prune the next chunk's seeds (16 chunks per batch: 6.0 s on the
// 4 Mb demo, 1 chunk: 0.9 s). Records carry the L: links of the graph pass
// (contig_assembly.cpp:1012-1084); ids and order are this program's, the reference's depend on its thread schedule.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/shk.h"
#include "fastq_chunker.hpp"

using namespace std;
using namespace shk;

static void usage(const char *argv0) {
  cerr << endl << argv0 << "  <options>\nOptions:\n"
       << "  -h [ --help ]                          print help messages\n"
       << "  -k arg                                 k-mer size\n"
       << "  -i [ --input ] arg                     a file containing a list of read file name(s)\n"
       << "  -f [ --format ] arg (=f)               format of the input: g(gzip); b(bzip2); f(plain fastq)\n"
       << "  -c [ --cqf ] arg                       the counting quotient filter built with the same 'k'\n"
       << "  -s [ --abundance_min ] arg (=2)        minimum coverage of k-mers used to extend the assembly\n"
       << "  -x [ --solid_abundance_min ] arg (=2)  minimum coverage of a solid k-mer to start the assembly\n"
       << "  -X [ --solid_abundance_max ] arg (=1000000) maximum coverage of a solid k-mer to start the assembly\n"
       << "  -t arg (=16)                           number of threads (kept for compatibility; the GPU does the work)\n"
       << "  -o [ --output ] arg (=unitigs.fa)      output contig file name (fasta)\n"
       << "  (hooks, not in the reference: --max-len N --part-size N --overhead N --batch-chunks N --device N)\n\n";
}

int main(int argc, char *argv[]) {
  int K = -1, device = 0;
  long long amin = 2, xmin = 2, xmax = 1000000, max_len = 1 << 26, part_size = 1LL << 23, overhead = 65535, batch_chunks = 1;
  string flist, cqf, output = "unitigs.fa";
  char fmt = 'f';
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
    else if (name == "-i" || name == "--input") flist = val();
    else if (name == "-f" || name == "--format") fmt = val()[0];
    else if (name == "-c" || name == "--cqf") cqf = val();
    else if (name == "-s" || name == "--abundance_min") amin = atoll(val().c_str());
    else if (name == "-x" || name == "--solid_abundance_min") xmin = atoll(val().c_str());
    else if (name == "-X" || name == "--solid_abundance_max") xmax = atoll(val().c_str());
    else if (name == "-t") (void)val();
    else if (name == "-o" || name == "--output") output = val();
    else if (name == "--max-len") max_len = atoll(val().c_str());
    else if (name == "--part-size") part_size = atoll(val().c_str());
    else if (name == "--overhead") overhead = atoll(val().c_str());
    else if (name == "--batch-chunks") batch_chunks = atoll(val().c_str());
    else if (name == "--device") device = atoi(val().c_str());
    else { cerr << "unrecognised option " << a << endl; usage(argv[0]); return 0; }
  }
  if (K < 0 || flist.empty() || cqf.empty()) { usage(argv[0]); return 0; }

  // list of read files, opened AS WRITTEN (relative names are relative to the working directory): the reference's
  // Contiger hands the lines to seqFile_batch unchanged (src/contig_assembly.cpp:248-258; its help text asks for
  // absolute names when the reads are elsewhere) -- unlike CQF-deNoise, which prefixes the list's directory
  // (src/CQF-deNoise.cpp:59-81)
  vector<string> files;
  ifstream fin(flist);
  if (!fin.is_open()) { cerr << "Failed to open file: " << flist << endl; return 0; }
  string line;
  while (getline(fin, line)) { if (line.empty()) continue; files.push_back(line); }
  FILE_MODE ftype;
  if (fmt == 'g') ftype = GZIP; else if (fmt == 'b') ftype = BZIP2; else if (fmt == 'f') ftype = TEXT;
  else { cerr << "Unrecognized file type " << fmt << endl; return 0; }

  // geometry from the .cqf header (qfmetadata: nslots at offset 16, gqf.h:62-77)
  uint64_t nslots = 0;
  { FILE *f = fopen(cqf.c_str(), "rb");
    unsigned char h[128];
    if (!f || fread(h, 1, 128, f) != 128) { cerr << "Failed to read " << cqf << endl; return 1; }
    fclose(f);
    memcpy(&nslots, h + 16, 8); }
  uint32_t qb = 0;
  while ((1ULL << qb) < nslots) qb++;

  shk_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.qb = qb; cfg.hb = qb + 8; cfg.seed = 2038074761; cfg.k = (uint32_t)K;
  cfg.max_batch_bytes = (uint64_t)batch_chunks * ((uint64_t)part_size + (uint64_t)overhead + 64);
  cfg.max_batch_keys = 1024;    // no counting in this program
  cfg.device = device;
  shk_ctx *ctx = nullptr;
  int rc = shk_create(&cfg, &ctx);
  if (rc) { cerr << "Contiger: " << shk_strerror(rc) << endl; return 1; }
  rc = shk_import_cqf(ctx, cqf.c_str());
  if (rc) { cerr << "Contiger: cannot load " << cqf << ": " << shk_strerror(rc) << endl; return 1; }

  time_t start_time = time(NULL);
  cerr << "Contiger settings:" << endl << "K: " << K << endl << "abundance_min: " << amin << endl << "solid_abundance_min: " << xmin << endl
       << "solid_abundance_max: " << xmax << endl << "cqf: " << cqf << " (qb " << qb << ")" << endl << endl;
  seqFile_batch seqs(files, FASTQ, ftype, (uint64_t)part_size, (uint32_t)overhead);
  if (seqs.bad()) { cerr << "Error: Wrong input file!" << endl; return 1; }
  shk_unitig_set *set = shk_unitig_set_new();
  vector<char> text;
  vector<uint64_t> off, len;
  uint64_t nseeds_total = 0;
  auto flush = [&]() -> int {
    if (off.empty()) return 0;
    uint64_t n = 0;
    // seeds of this batch of parts (processDataChunk's rule) and their walks, all on the device
    int r = shk_unitigs_add_reads(ctx, set, text.data(), 0, text.size(), off.data(), len.data(), (uint32_t)off.size(), (uint32_t)K,
                                  (uint64_t)amin, (uint64_t)xmin, (uint64_t)xmax, (uint32_t)max_len, &n);
    nseeds_total += n;
    text.clear(); off.clear(); len.clear();
    return r;
  };
  chunk ch;
  while (seqs.getDataChunk(ch)) {
    off.push_back(text.size()); len.push_back(ch.get_size());
    text.insert(text.end(), ch.get_reads(), ch.get_reads() + ch.get_size());
    free(ch.get_reads());
    if ((long long)off.size() == batch_chunks && (rc = flush())) break;
  }
  if (!rc) rc = flush();
  shk_unitig_stats st;
  memset(&st, 0, sizeof(st));
  if (!rc) rc = shk_unitig_set_write(set, (uint32_t)K, output.c_str(), &st);
  shk_unitig_set_free(set);
  shk_destroy(ctx);
  if (rc) { cerr << "Contiger: " << shk_strerror(rc) << endl; return 1; }
  cerr << "seeds: " << nseeds_total << " unitigs: " << st.unitigs << " total length: " << st.total_len << " rounds: " << st.rounds
       << " extensions: " << st.extensions << " duplicates: " << st.duplicates << " truncated: " << st.truncated << endl;
  cerr << "Time for finding unitigs: " << difftime(time(NULL), start_time) << " seconds." << endl;
  return 0;
}
