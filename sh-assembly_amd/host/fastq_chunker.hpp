// Host-side FASTQ chunker: same interface and behaviour as the reference's seqFile_batch /
// fastq_read_parts (cqf/CQF_mt.h:334-412, 561-585, 735-816, 933-957), written against
// stdio + zlib only (no boost lock-free queue: one host thread feeds the GPU).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <zlib.h>
#include <deque>
#include <memory>
#include <string>
#include <vector>

namespace shk {

enum FILE_TYPE { FASTA, FASTQ };            // base/global.h:106
enum FILE_MODE { TEXT, GZIP, BZIP2 };       // base/global.h:107

struct chunk {                              // cqf/chunk.h:23-41 (the part of it the path uses)
  char *reads = nullptr;
  uint64_t size = 0;
  char *get_reads() const { return reads; }
  uint64_t get_size() const { return size; }
  void set(char *r, uint64_t s) { reads = r; size = s; }
};

struct file_pointer {                       // cqf/CQF_mt.h:324-331
  FILE *in = nullptr;
  gzFile in_gzip = nullptr;
  void *in_bzip2 = nullptr;                 // BZFILE* (libbz2 is bound at run time, see fastq_chunker.cpp)
  int bzerror = 0;
  std::vector<char> part_buffer;            // carry-over between parts
  FILE_MODE fmode = TEXT;
  uint64_t part_filled = 0;
};

class seqFile_batch {
 public:
  // part_size / overhead default to the reference's constants (CQF_mt.h:742-743);
  // tests shrink them to exercise many part boundaries on small files.
  seqFile_batch(const std::vector<std::string> &file_names, FILE_TYPE ft, FILE_MODE fm,
                uint64_t part_size = 1ULL << 23, uint32_t overhead = 65535);
  ~seqFile_batch();
  // one part of the file at the head of the queue; the file goes back to the tail
  // (cqf/CQF_mt.h:364-390). The chunk is malloc'ed; the caller frees it.
  bool getDataChunk(chunk &data);
  int num_files() const { return (int)files_.size(); }
  bool bad() const { return bad_; }         // "Error: Wrong input file!" (CQF_mt.h:764-768)

 private:
  bool read_part(file_pointer *fp, chunk &out);
  bool is_eof(file_pointer *fp) const;
  std::deque<std::unique_ptr<file_pointer>> files_;
  uint64_t part_size_;
  uint32_t overhead_;
  bool bad_ = false;
};

}  // namespace shk
