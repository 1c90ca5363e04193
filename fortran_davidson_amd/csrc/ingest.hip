// Host-only: parser and file streamers of ingest.h.  No device code in this file.
#include "ingest.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <charconv>
#include <cstdlib>
#include <cstring>
#include <thread>

static inline bool is_sep(char ch) { return ch == ' ' || ch == '\n' || ch == '\r' || ch == '\t' || ch == ',' || ch == '\f' || ch == '\v'; }

// one token [p, q): optional "r*" repeat prefix, then a real number
static bool convert_token(const char* p, const char* q, std::vector<double>* out) {
  long repeat = 1;
  const char* star = (const char*)memchr(p, '*', (size_t)(q - p));
  if (star) {
    auto r = std::from_chars(p, star, repeat);
    if (r.ec != std::errc() || r.ptr != star || repeat < 1) return false;
    p = star + 1;
  }
  if (p < q && *p == '+') ++p;
  double v;
  auto r = std::from_chars(p, q, v);
  if (r.ec != std::errc() || r.ptr != q) {
    // Fortran "D" exponent letter, or an exponent written without a letter ("1.0+05"): rewrite and retry
    char tmp[80];
    size_t len = (size_t)(q - p);
    if (len == 0 || len >= sizeof(tmp) - 2) return false;
    size_t o = 0;
    bool seen_exp = false;
    for (size_t i = 0; i < len; ++i) {
      char ch = p[i];
      if (ch == 'D' || ch == 'd' || ch == 'Q' || ch == 'q' || ch == 'E' || ch == 'e') { ch = 'e'; seen_exp = true; }
      else if ((ch == '+' || ch == '-') && i > 0 && !seen_exp) { tmp[o++] = 'e'; seen_exp = true; }   // "1.0+05"
      tmp[o++] = ch;
    }
    auto r2 = std::from_chars(tmp, tmp + o, v);
    if (r2.ec != std::errc() || r2.ptr != tmp + o) return false;
  }
  if (repeat == 1) out->push_back(v);
  else out->insert(out->end(), (size_t)repeat, v);
  return true;
}

size_t ingest_parse_text(const char* buf, size_t len, bool final, std::vector<double>* out, std::string* err) {
  size_t i = 0;
  for (;;) {
    while (i < len && is_sep(buf[i])) ++i;
    if (i >= len) return len;
    size_t j = i;
    while (j < len && !is_sep(buf[j])) ++j;
    if (j == len && !final) return i;                 // possibly cut token: leave it for the next call
    if (!convert_token(buf + i, buf + j, out)) {
      if (err) *err = "not a number: '" + std::string(buf + i, std::min<size_t>(j - i, 40)) + "'";
      return (size_t)-1;
    }
    i = j;
  }
}

size_t ingest_parse_text_parallel(const char* buf, size_t len, bool final, std::vector<double>* out, int threads,
                                  std::string* err) {
  if (threads < 2 || len < (size_t)1 << 20) return ingest_parse_text(buf, len, final, out, err);
  // cut points at token starts
  std::vector<size_t> cut(threads + 1);
  cut[0] = 0; cut[threads] = len;
  for (int t = 1; t < threads; ++t) {
    size_t p = std::max(cut[t - 1], len / threads * t);
    while (p < len && p > 0 && !is_sep(buf[p - 1])) ++p;
    cut[t] = p;
  }
  std::vector<std::vector<double>> part(threads);
  std::vector<size_t> used(threads, 0);
  std::vector<std::string> errs(threads);
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([&, t] {
      part[t].reserve((cut[t + 1] - cut[t]) / 8 + 16);
      // inner pieces end at a separator, so their last token is complete
      used[t] = ingest_parse_text(buf + cut[t], cut[t + 1] - cut[t], t + 1 < threads ? true : final, &part[t], &errs[t]);
    });
  for (auto& th : pool) th.join();
  for (int t = 0; t < threads; ++t)
    if (used[t] == (size_t)-1) { if (err) *err = errs[t]; return (size_t)-1; }
  size_t total = 0;
  for (auto& p : part) total += p.size();
  out->reserve(out->size() + total);
  for (auto& p : part) out->insert(out->end(), p.begin(), p.end());
  return cut[threads - 1] + used[threads - 1];
}

namespace {
struct Fd {
  int fd = -1;
  ~Fd() { if (fd >= 0) close(fd); }
};
int parse_threads() {
  if (const char* ev = getenv("DAV_INGEST_THREADS")) return std::max(1, atoi(ev));
  unsigned hc = std::thread::hardware_concurrency();
  return (int)std::min<unsigned>(std::max<unsigned>(hc, 1), 16);
}
}  // namespace

int ingest_text_file(const char* path, int64_t n, IngestSink& sink, std::string* err) {
  Fd f;
  f.fd = open(path, O_RDONLY);
  if (f.fd < 0) { *err = std::string("cannot open ") + path + ": " + strerror(errno); return 1; }
  int64_t want0, wantn;
  sink.wanted(&want0, &wantn);
  size_t CHUNK = (size_t)64 << 20;                      // bytes of text per read; DAV_INGEST_CHUNK shrinks it (tests)
  if (const char* ev = getenv("DAV_INGEST_CHUNK")) CHUNK = std::max<size_t>(64, (size_t)atoll(ev));
  std::vector<char> text(CHUNK + 4096);
  std::vector<double> vals;         // parsed, not yet committed; vals[head..] belongs to row `row`, column `col0`
  size_t head = 0, carry = 0;
  int64_t row = 0;                  // next row to complete
  const int threads = parse_threads();
  bool eof = false;
  while (!eof) {
    ssize_t got = read(f.fd, text.data() + carry, CHUNK);
    if (got < 0) { *err = std::string("read error on ") + path + ": " + strerror(errno); return 1; }
    if (got == 0) eof = true;
    size_t len = carry + (size_t)got;
    std::string perr;
    size_t used = ingest_parse_text_parallel(text.data(), len, eof, &vals, threads, &perr);
    if (used == (size_t)-1) { *err = std::string(path) + ": " + perr; return 1; }
    carry = len - used;
    if (carry > 4096) { *err = std::string(path) + ": token longer than 4096 bytes"; return 1; }
    memmove(text.data(), text.data() + used, carry);
    // hand complete rows to the sink
    while ((int64_t)(vals.size() - head) >= n && row < n) {
      double* buf; int64_t cap;
      int64_t avail = (int64_t)((vals.size() - head) / (size_t)n);
      avail = std::min(avail, n - row);
      // rows outside the wanted range are dropped without staging
      if (row + avail <= want0 || row >= want0 + wantn) { head += (size_t)(avail * n); row += avail; continue; }
      if (row < want0) { int64_t skip = want0 - row; head += (size_t)(skip * n); row += skip; continue; }
      if (int rc = sink.acquire(&buf, &cap)) return rc;
      int64_t take = std::min({avail, cap, want0 + wantn - row});
      memcpy(buf, vals.data() + head, sizeof(double) * (size_t)(take * n));
      if (int rc = sink.commit(row, take)) return rc;
      head += (size_t)(take * n); row += take;
    }
    if (head > 0) { vals.erase(vals.begin(), vals.begin() + (ptrdiff_t)head); head = 0; }
    if (row >= n && !vals.empty()) break;
  }
  if (row < n) { *err = std::string(path) + ": expected " + std::to_string(n) + " x " + std::to_string(n) + " values, file ends in row " + std::to_string(row + 1); return 1; }
  if (!vals.empty()) { *err = std::string(path) + ": more than " + std::to_string(n) + " x " + std::to_string(n) + " values"; return 1; }
  return 0;
}

int ingest_f64_file(const char* path, int64_t n, IngestSink& sink, std::string* err) {
  Fd f;
  f.fd = open(path, O_RDONLY);
  if (f.fd < 0) { *err = std::string("cannot open ") + path + ": " + strerror(errno); return 1; }
  struct stat sb;
  if (fstat(f.fd, &sb) != 0 || (int64_t)sb.st_size != n * n * 8) {
    *err = std::string(path) + ": size is not 8 n^2 = " + std::to_string(n * n * 8) + " bytes";
    return 1;
  }
  int64_t want0, wantn;
  sink.wanted(&want0, &wantn);
  for (int64_t row = want0; row < want0 + wantn;) {
    double* buf; int64_t cap;
    if (int rc = sink.acquire(&buf, &cap)) return rc;
    int64_t take = std::min(cap, want0 + wantn - row);
    // the block is read by several threads: one pread stream from the page cache runs at ~10 GB/s, well
    // below what the host-to-device copy that follows can take
    const size_t bytes = sizeof(double) * (size_t)(take * n);
    const int T = (int)std::min<size_t>((size_t)std::min(parse_threads(), 8), bytes / ((size_t)4 << 20) + 1);
    std::vector<int> bad(T, 0);
    std::vector<std::thread> pool;
    for (int t = 0; t < T; ++t)
      pool.emplace_back([&, t] {
        size_t lo = bytes / T * t, hi = t + 1 == T ? bytes : bytes / T * (t + 1);
        while (lo < hi) {
          ssize_t got = pread(f.fd, (char*)buf + lo, hi - lo, (off_t)(row * n * 8) + (off_t)lo);
          if (got <= 0) { bad[t] = 1; return; }
          lo += (size_t)got;
        }
      });
    for (auto& th : pool) th.join();
    for (int t = 0; t < T; ++t)
      if (bad[t]) { *err = std::string("read error on ") + path; return 1; }
    if (int rc = sink.commit(row, take)) return rc;
    row += take;
  }
  return 0;
}
