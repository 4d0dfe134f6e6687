#include "fastq_chunker.hpp"

#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

namespace shk {

// ------------------------------------------------------------------ byte sources
namespace {

class PlainSource : public ByteSource {
 public:
  explicit PlainSource(FILE *f) : f_(f) {}
  ~PlainSource() override { fclose(f_); }
  uint64_t read(char *dst, uint64_t n) override { return fread(dst, 1, n, f_); }
  bool at_end() const override { return feof(f_) != 0; }
  bool failed() const override { return ferror(f_) != 0; }

 private:
  FILE *f_;
};

class GzipSource : public ByteSource {
 public:
  explicit GzipSource(gzFile g) : g_(g) { gzbuffer(g_, 1u << 26); }
  ~GzipSource() override { gzclose(g_); }
  uint64_t read(char *dst, uint64_t n) override {
    const int r = gzread(g_, dst, (unsigned)n);
    if (r < 0) { failed_ = true; return 0; }          // corrupt or truncated stream (Z_DATA_ERROR, Z_BUF_ERROR)
    if ((uint64_t)r < n) {                            // a short read: the clean end of the stream, or its premature end
      int e = Z_OK;
      gzerror(g_, &e);
      if (e != Z_OK && e != Z_STREAM_END) failed_ = true;
    }
    return (uint64_t)r;
  }
  bool at_end() const override { return gzeof(g_) != 0; }
  bool failed() const override { return failed_; }

 private:
  gzFile g_;
  bool failed_ = false;
};

// libbz2's stream-reading interface (bzlib.h: BZ2_bzReadOpen / BZ2_bzRead / BZ2_bzReadClose), bound with dlopen: the
// image ships the library but not its header. The reference calls the same three functions (cqf/CQF_mt.h:756, 948).
struct Bz2Api {
  void *(*read_open)(int *bzerror, FILE *f, int verbosity, int small, void *unused, int nunused);
  int (*read)(int *bzerror, void *b, void *buf, int len);
  void (*read_close)(int *bzerror, void *b);
  static const Bz2Api *get() {
    static Bz2Api api;
    static int state = 0;   // 0 untried, 1 ok, -1 missing
    if (state == 0) {
      void *h = dlopen("libbz2.so.1.0", RTLD_NOW);
      if (!h) h = dlopen("libbz2.so.1", RTLD_NOW);
      if (h) {
        api.read_open = (decltype(api.read_open))dlsym(h, "BZ2_bzReadOpen");
        api.read = (decltype(api.read))dlsym(h, "BZ2_bzRead");
        api.read_close = (decltype(api.read_close))dlsym(h, "BZ2_bzReadClose");
      }
      state = (h && api.read_open && api.read && api.read_close) ? 1 : -1;
    }
    return state == 1 ? &api : nullptr;
  }
};

class Bzip2Source : public ByteSource {
 public:
  Bzip2Source(FILE *f, void *bz) : f_(f), bz_(bz) {}
  ~Bzip2Source() override {
    int e;
    Bz2Api::get()->read_close(&e, bz_);
    fclose(f_);
  }
  uint64_t read(char *dst, uint64_t n) override {
    const int r = Bz2Api::get()->read(&err_, bz_, dst, (int)n);
    return r > 0 ? (uint64_t)r : 0;
  }
  bool at_end() const override { return err_ == 4; }   // BZ_STREAM_END (cqf/CQF_mt.h:567)
  bool failed() const override { return err_ < 0; }    // BZ_DATA_ERROR, BZ_UNEXPECTED_EOF, ...

 private:
  FILE *f_;
  void *bz_;
  int err_ = 0;
};

}  // namespace

std::unique_ptr<ByteSource> ByteSource::open(const std::string &path, FILE_MODE mode) {
  if (mode == GZIP) {
    gzFile g = gzopen(path.c_str(), "rb");
    return g ? std::unique_ptr<ByteSource>(new GzipSource(g)) : nullptr;
  }
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return nullptr;
  if (mode == TEXT) return std::unique_ptr<ByteSource>(new PlainSource(f));
  const Bz2Api *api = Bz2Api::get();
  if (!api) {
    fprintf(stderr, "bzip2 input needs libbz2.so.1.0 at run time (not found)\n");
    fclose(f);
    return nullptr;
  }
  setvbuf(f, NULL, _IOFBF, 1u << 26);
  int e = 0;
  void *bz = api->read_open(&e, f, 0, 0, NULL, 0);
  if (!bz) { fclose(f); return nullptr; }
  return std::unique_ptr<ByteSource>(new Bzip2Source(f, bz));
}

// ------------------------------------------------------------------ record boundary
namespace {

struct TextView {
  const char *p;
  int64_t n;
  bool eol(int64_t i) const { return p[i] == '\n' || p[i] == '\r'; }
  // Start of the first line that begins behind `from`: the byte after an end-of-line byte that is not itself
  // one. Only positions up to n - 3 are inspected, as the reference's scan does (CQF_mt.h:573-585); -1 if none.
  int64_t line_start_after(int64_t from) const {
    for (int64_t i = from; i < n - 2; i++)
      if (eol(i) && !eol(i + 1)) return i + 1;
    return -1;
  }
};

}  // namespace

uint64_t fastq_record_cut(const char *text, uint64_t n, uint32_t overhead) {
  const TextView v = {text, (int64_t)n};
  // nine consecutive line starts behind the probe position: four candidate record starts and the lines that
  // decide about them (candidate k needs the starts of lines k+1, k+2 and k+3)
  int64_t ls[9];
  int64_t from = v.n - overhead / 2;
  if (from < 0) from = 0;                  // a buffer shorter than the probe distance is scanned from its start
  for (int j = 0; j < 9; j++) {
    ls[j] = v.line_start_after(from);
    if (ls[j] < 0) return 0;
    from = ls[j];
  }
  for (int k = 0; k < 4; k++) {
    const int64_t head = ls[k], plus = ls[k + 2];
    if (text[head] != '@' || text[plus] != '+') continue;
    if (v.eol(plus + 1)) return (uint64_t)head;                       // bare '+' line
    const int64_t head_len = ls[k + 1] - head, plus_len = ls[k + 3] - plus;
    if (head_len == plus_len && memcmp(text + head + 1, text + plus + 1, (size_t)(plus_len - 1)) == 0)
      return (uint64_t)head;                                          // '+' line repeats the header
  }
  return 0;
}

// ------------------------------------------------------------------ parts, file by file
seqFile_batch::seqFile_batch(const std::vector<std::string> &file_names, FILE_TYPE, FILE_MODE fm, uint64_t part_size,
                             uint32_t overhead)
    : part_size_(part_size), overhead_(overhead) {
  if (part_size_ < overhead_) {            // the cut is searched in the last overhead/2 bytes OF A PART (CQF_mt.h:742-743, 781)
    fprintf(stderr, "Error: part size %llu below the chunker's overhead %u\n", (unsigned long long)part_size_, overhead_);
    bad_ = true;
    return;
  }
  for (const auto &name : file_names) {   // getFileReader, cqf/CQF_mt.h:933-957: unreadable files are skipped
    std::unique_ptr<OpenFile> f(new OpenFile());
    f->src = ByteSource::open(name, fm);
    if (f->src) files_.push_back(std::move(f));
  }
}

seqFile_batch::~seqFile_batch() {}

// One part = what is carried over from the previous part + the next part_size_ bytes of the stream, cut at a
// record start (the rest is carried). A stream that has raised its end flag gives no further part; the read
// that raises it hands out everything that is left, uncut (cqf/CQF_mt.h:747-776).
bool seqFile_batch::next_part(OpenFile &f, chunk &out) {
  if (f.src->at_end()) return false;
  const uint64_t carried = f.carry.size();
  if (carried >= overhead_) {
    // no record start in a whole part: not FASTQ (wrong -f, or lines longer than the chunker's overhead).
    // Checked BEFORE reading, so that a carry of any size never meets a fixed-size buffer.
    fprintf(stderr, "Error: Wrong input file!\n");
    bad_ = true;
    return false;
  }
  char *buf = (char *)malloc(carried + part_size_ + 1);
  if (!buf) { bad_ = true; return false; }
  if (carried) memcpy(buf, f.carry.data(), carried);
  const uint64_t got = f.src->read(buf + carried, part_size_);
  if (f.src->failed() || (got == 0 && !f.src->at_end())) {
    // a decompressor error (corrupt or truncated .gz / .bz2) or a read that makes no progress: an I/O error, not an
    // empty part -- the reference would loop on it or run over its buffer (CQF_mt.h:749-760 take gzread's -1 as a size)
    fprintf(stderr, "Error: cannot read on in an input file (corrupt or truncated?)\n");
    free(buf);
    bad_ = true;
    return false;
  }
  const uint64_t total = carried + got;
  if (f.src->at_end()) {
    out.set(buf, total);
    return true;
  }
  const uint64_t cut = fastq_record_cut(buf, total, overhead_);
  f.carry.assign(buf + cut, buf + total);
  out.set(buf, cut);
  return true;
}

bool seqFile_batch::getDataChunk(chunk &data) {
  while (!files_.empty() && !bad_) {
    std::unique_ptr<OpenFile> f = std::move(files_.front());
    files_.pop_front();
    if (next_part(*f, data)) {
      files_.push_back(std::move(f));   // round robin over the files (cqf/CQF_mt.h:828-830)
      return true;
    }
  }
  return false;
}

}  // namespace shk
