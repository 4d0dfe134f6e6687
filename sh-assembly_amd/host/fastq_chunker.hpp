// Host-side FASTQ chunker. Hands out the same parts, in the same order, as the reference's
// seqFile_batch / fastq_read_parts (cqf/CQF_mt.h:334-412, 561-585, 735-816, 933-957) -- the parts are
// the points at which the deNoise trigger is tested, so their sizes are part of the result -- but is
// organised differently: a ByteSource per compression format, a growing carry buffer per file and a
// record-boundary search over a read-only text view. One host thread feeds the GPU (no lock-free queue).
#pragma once
#include <stdint.h>
#include <stdio.h>
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

// sequential byte stream over a plain, gzip or bzip2 file
class ByteSource {
 public:
  virtual ~ByteSource() {}
  virtual uint64_t read(char *dst, uint64_t n) = 0;   // bytes delivered (short only at the end of the stream)
  virtual bool at_end() const = 0;                    // the format's own end-of-stream flag (raised by a short read)
  virtual bool failed() const = 0;                    // the stream is damaged (decompressor / read error): no further data
  static std::unique_ptr<ByteSource> open(const std::string &path, FILE_MODE mode);   // null when unreadable
};

// Where to cut a buffer of FASTQ text so that the cut falls on a record start: the first of the four line starts
// behind position `n - overhead/2` that begins with '@' while the line two further on begins with '+' and is either
// bare or repeats the header (cqf/CQF_mt.h:781-808). 0 when the tail holds no such line (the whole buffer is carried).
uint64_t fastq_record_cut(const char *text, uint64_t n, uint32_t overhead);

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
  struct OpenFile {
    std::unique_ptr<ByteSource> src;
    std::vector<char> carry;                // text behind the last cut, prepended to the next part
  };
  bool next_part(OpenFile &f, chunk &out);
  std::deque<std::unique_ptr<OpenFile>> files_;
  uint64_t part_size_;
  uint32_t overhead_;
  bool bad_ = false;
};

}  // namespace shk
