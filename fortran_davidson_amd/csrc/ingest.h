// On-disk matrix ingest (host side): the reference's text format and a raw float64 fast path, streamed
// by blocks of complete rows into a sink - the engine's pinned staging buffers (engine.hip) - so that no
// host N x N copy ever exists.  Format reference: read_matrix / write_matrix / write_vector in
// src/tests/test_utils.f90:118-166 (list-directed, one value per line, ROW-major).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

struct IngestSink {
  virtual ~IngestSink() {}
  // next staging buffer: row-major, leading dimension n, room for *cap_rows complete rows
  virtual int acquire(double** buf, int64_t* cap_rows) = 0;
  // rows [row0, row0 + nrows) are now in the buffer handed out by the last acquire()
  virtual int commit(int64_t row0, int64_t nrows) = 0;
  // rows this process stores: [*first, *first + *count) (a rank of a row-slab partition skips the rest)
  virtual void wanted(int64_t* first, int64_t* count) = 0;
};

// Whitespace/comma separated decimal numbers as Fortran list-directed output writes them ("E", "D" or no
// exponent letter, optional sign, r*c repeat form).  Parses complete tokens only: when `final` is false
// a token that touches the end of the buffer is left for the next call.  Returns the number of bytes
// consumed, or (size_t)-1 with *err set.  Appends to `out`.
size_t ingest_parse_text(const char* buf, size_t len, bool final, std::vector<double>* out, std::string* err);
// Same, with the buffer cut into `threads` pieces at token boundaries and parsed concurrently.
size_t ingest_parse_text_parallel(const char* buf, size_t len, bool final, std::vector<double>* out, int threads,
                                  std::string* err);

// n x n matrix, row-major text.  Every process parses the whole file and commits only its wanted rows.
int ingest_text_file(const char* path, int64_t n, IngestSink& sink, std::string* err);
// n x n matrix, raw little-endian float64, row-major, no header (size must be 8 n^2): pread of the wanted rows only.
int ingest_f64_file(const char* path, int64_t n, IngestSink& sink, std::string* err);
