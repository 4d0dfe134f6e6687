#include "fastq_chunker.hpp"

#include <dlfcn.h>

#include <stdlib.h>
#include <string.h>

namespace shk {

// libbz2's stream-reading interface (bzlib.h: BZ2_bzReadOpen / BZ2_bzRead / BZ2_bzReadClose), bound with dlopen: the
// image ships the library but not its header. The reference calls the same three functions (cqf/CQF_mt.h:756, 948).
struct Bz2Api {
  void *(*read_open)(int *bzerror, FILE *f, int verbosity, int small, void *unused, int nunused);
  int (*read)(int *bzerror, void *b, void *buf, int len);
  void (*read_close)(int *bzerror, void *b);
};
static const Bz2Api *bz2_api() {
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

seqFile_batch::seqFile_batch(const std::vector<std::string> &file_names, FILE_TYPE, FILE_MODE fm, uint64_t part_size,
                             uint32_t overhead)
    : part_size_(part_size), overhead_(overhead) {
  for (const auto &fname : file_names) {   // getFileReader, cqf/CQF_mt.h:933-957: unreadable files are skipped
    std::unique_ptr<file_pointer> fp(new file_pointer());
    fp->fmode = fm;
    if (fm == TEXT) {
      fp->in = fopen(fname.c_str(), "rb");
      if (!fp->in) continue;
    } else if (fm == GZIP) {
      fp->in_gzip = gzopen(fname.c_str(), "rb");
      if (!fp->in_gzip) continue;
      gzbuffer(fp->in_gzip, 1u << 26);
    } else {
      // cqf/CQF_mt.h:944-954: FILE* + BZ2_bzReadOpen
      if (!bz2_api()) { fprintf(stderr, "bzip2 input needs libbz2.so.1.0 at run time (not found)\n"); continue; }
      fp->in = fopen(fname.c_str(), "rb");
      if (!fp->in) continue;
      setvbuf(fp->in, NULL, _IOFBF, 1u << 26);
      fp->in_bzip2 = bz2_api()->read_open(&fp->bzerror, fp->in, 0, 0, NULL, 0);
      if (!fp->in_bzip2) { fclose(fp->in); fp->in = nullptr; continue; }
    }
    fp->part_buffer.resize(overhead_ + part_size_);
    files_.push_back(std::move(fp));
  }
}

seqFile_batch::~seqFile_batch() {
  for (auto &fp : files_) {
    if (fp->in_bzip2) { int e; bz2_api()->read_close(&e, fp->in_bzip2); }
    if (fp->in) fclose(fp->in);
    if (fp->in_gzip) gzclose(fp->in_gzip);
  }
}

bool seqFile_batch::is_eof(file_pointer *fp) const {   // cqf/CQF_mt.h:561-570
  if (fp->fmode == TEXT) return feof(fp->in) != 0;
  if (fp->fmode == GZIP) return gzeof(fp->in_gzip) != 0;
  if (fp->fmode == BZIP2) return fp->bzerror == 4;   // BZ_STREAM_END
  return true;
}

// cqf/CQF_mt.h:573-585
static bool skip_next_eol(const char *part, int64_t &pos, int64_t max_pos) {
  int64_t i;
  for (i = pos; i < max_pos - 2; ++i)
    if ((part[i] == '\n' || part[i] == '\r') && !(part[i + 1] == '\n' || part[i + 1] == '\r')) break;
  if (i >= max_pos - 2) return false;
  pos = i + 1;
  return true;
}

// fastq_read_parts, cqf/CQF_mt.h:735-816
bool seqFile_batch::read_part(file_pointer *fp, chunk &out) {
  char *part = (char *)malloc(part_size_ + overhead_);
  memcpy(part, fp->part_buffer.data(), fp->part_filled);
  if (is_eof(fp)) { free(part); return false; }
  uint64_t readed = 0;
  if (fp->fmode == TEXT) readed = fread(part + fp->part_filled, 1, part_size_, fp->in);
  else if (fp->fmode == GZIP) { int r = gzread(fp->in_gzip, part + fp->part_filled, (unsigned)part_size_); readed = r > 0 ? (uint64_t)r : 0; }
  else if (fp->fmode == BZIP2) { int r = bz2_api()->read(&fp->bzerror, fp->in_bzip2, part + fp->part_filled, (int)part_size_); readed = r > 0 ? (uint64_t)r : 0; }
  const int64_t total_filled = (int64_t)(fp->part_filled + readed);
  if (fp->part_filled >= overhead_) {
    fprintf(stderr, "Error: Wrong input file!\n");
    bad_ = true;
    free(part);
    return false;
  }
  if (is_eof(fp)) { out.set(part, (uint64_t)total_filled); return true; }
  uint64_t size;
  int64_t line_start[9];
  int j;
  int64_t i = total_filled - overhead_ / 2;
  for (j = 0; j < 9; ++j) {
    if (!skip_next_eol(part, i, total_filled)) break;
    line_start[j] = i;
  }
  if (j < 9) size = 0;
  else {
    int k;
    for (k = 0; k < 4; ++k) {
      if (part[line_start[k]] == '@' && part[line_start[k + 2]] == '+') {
        if (part[line_start[k + 2] + 1] == '\n' || part[line_start[k + 2] + 1] == '\r') break;
        if (line_start[k + 1] - line_start[k] == line_start[k + 3] - line_start[k + 2] &&
            memcmp(part + line_start[k] + 1, part + line_start[k + 2] + 1, (size_t)(line_start[k + 3] - line_start[k + 2] - 1)) == 0)
          break;
      }
    }
    size = (k == 4) ? 0 : (uint64_t)line_start[k];
  }
  memcpy(fp->part_buffer.data(), part + size, (size_t)(total_filled - (int64_t)size));
  fp->part_filled = (uint64_t)total_filled - size;
  out.set(part, size);
  return true;
}

bool seqFile_batch::getDataChunk(chunk &data) {
  while (!files_.empty()) {
    std::unique_ptr<file_pointer> fp = std::move(files_.front());
    files_.pop_front();
    if (read_part(fp.get(), data)) {
      files_.push_back(std::move(fp));
      return true;
    }
    if (fp->in_bzip2) { int e; bz2_api()->read_close(&e, fp->in_bzip2); fp->in_bzip2 = nullptr; }
    if (fp->in) { fclose(fp->in); fp->in = nullptr; }
    if (fp->in_gzip) { gzclose(fp->in_gzip); fp->in_gzip = nullptr; }
    if (bad_) return false;
  }
  return false;
}

}  // namespace shk
