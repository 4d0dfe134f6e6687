#include "fastq_chunker.hpp"

#include <stdlib.h>
#include <string.h>

namespace shk {

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
      fprintf(stderr, "bzip2 input is not available in this build (no bzlib.h in the image)\n");
      continue;
    }
    fp->part_buffer.resize(overhead_ + part_size_);
    files_.push_back(std::move(fp));
  }
}

seqFile_batch::~seqFile_batch() {
  for (auto &fp : files_) {
    if (fp->in) fclose(fp->in);
    if (fp->in_gzip) gzclose(fp->in_gzip);
  }
}

bool seqFile_batch::is_eof(file_pointer *fp) const {   // cqf/CQF_mt.h:561-570
  if (fp->fmode == TEXT) return feof(fp->in) != 0;
  if (fp->fmode == GZIP) return gzeof(fp->in_gzip) != 0;
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
    if (fp->in) { fclose(fp->in); fp->in = nullptr; }
    if (fp->in_gzip) { gzclose(fp->in_gzip); fp->in_gzip = nullptr; }
    if (bad_) return false;
  }
  return false;
}

}  // namespace shk
