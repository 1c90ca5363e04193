// Host-only pieces of the engine under AddressSanitizer + UBSan (the reference's only "sanitizer" is its
// Debug build's run-time checking, src/CMakeLists.txt:13-17).  fortran_davidson_amd/csrc/ingest.hip holds no device code: it is
// compiled here as plain C++ with g++ -fsanitize=address,undefined and driven through the text parser (every spelling of
// Fortran list-directed output, repeat counts, tokens cut by a buffer boundary, garbage) and the two file readers with a
// sink that checks what it is handed.  Built and run by tests/test_host_sanitizer.py; GPU sanitizers are not available.
#include "ingest.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); std::exit(1); } } while (0)

struct CheckSink : IngestSink {
  int64_t n, first, count, cap;
  std::vector<double> buf, got;
  std::vector<char> seen;
  CheckSink(int64_t n_, int64_t first_, int64_t count_, int64_t cap_) : n(n_), first(first_), count(count_), cap(cap_), buf((size_t)cap_ * n_),
                                                                          got((size_t)n_ * n_, -1.0), seen((size_t)n_, 0) {}
  int acquire(double** b, int64_t* cap_rows) override { *b = buf.data(); *cap_rows = cap; return 0; }
  int commit(int64_t row0, int64_t nrows) override {
    REQUIRE(nrows >= 0 && nrows <= cap && row0 >= 0 && row0 + nrows <= n);
    for (int64_t r = 0; r < nrows; ++r) {
      std::memcpy(&got[(size_t)(row0 + r) * n], &buf[(size_t)r * n], sizeof(double) * n);
      seen[(size_t)(row0 + r)] = 1;
    }
    return 0;
  }
  void wanted(int64_t* f, int64_t* c) override { *f = first; *c = count; }
};

static double value(int64_t i, int64_t j) { return (i == j ? 1.0 + (double)i : 1e-3 * std::sin((double)(i * 131 + j))); }

int main(int argc, char** argv) {
  REQUIRE(argc == 2);
  const std::string dir = argv[1];
  std::string err;
  // ---- parser: spellings ----------------------------------------------------------------------------------------------
  {
    const char* text = " 1.5  -2.25E+01, 3.0D-2\n4*0.5 +7 1.0E0\t.5 -.25e1 2*-1.0d0\r\n";
    std::vector<double> out;
    const size_t used = ingest_parse_text(text, std::strlen(text), true, &out, &err);
    REQUIRE(used == std::strlen(text));
    const double want[] = {1.5, -22.5, 0.03, 0.5, 0.5, 0.5, 0.5, 7.0, 1.0, 0.5, -2.5, -1.0, -1.0};
    REQUIRE(out.size() == sizeof(want) / sizeof(want[0]));
    for (size_t i = 0; i < out.size(); ++i) REQUIRE(out[i] == want[i]);
  }
  // a token that touches the end of a non-final buffer is left for the next call; every split point of a text
  {
    const std::string text = "12.5 -3.75E+00 6*1.25 9.0D+1 0.001\n";
    std::vector<double> whole;
    REQUIRE(ingest_parse_text(text.data(), text.size(), true, &whole, &err) == text.size());
    for (size_t cut = 0; cut <= text.size(); ++cut) {
      std::vector<double> out;
      const size_t used = ingest_parse_text(text.data(), cut, false, &out, &err);
      REQUIRE(used != (size_t)-1 && used <= cut);
      const size_t rest = ingest_parse_text(text.data() + used, text.size() - used, true, &out, &err);
      REQUIRE(rest == text.size() - used);
      REQUIRE(out == whole);
    }
  }
  // garbage is an error, not a crash; empty input is fine
  {
    std::vector<double> out;
    REQUIRE(ingest_parse_text("1.0 abc 2.0", 11, true, &out, &err) == (size_t)-1 && !err.empty());
    out.clear();
    REQUIRE(ingest_parse_text("3*", 2, true, &out, &err) == (size_t)-1);
    out.clear();
    REQUIRE(ingest_parse_text("", 0, true, &out, &err) == 0 && out.empty());
    REQUIRE(ingest_parse_text("   \n\t ", 6, true, &out, &err) == 6 && out.empty());
  }
  // parallel parser == serial parser, for several thread counts, on a text larger than its per-thread minimum
  {
    std::string text;
    char tmp[64];
    for (int i = 0; i < 200000; ++i) {
      std::snprintf(tmp, sizeof tmp, i % 7 == 0 ? "%.17E\n" : "%.17g ", value(i % 977, i % 131));
      text += tmp;
    }
    std::vector<double> serial;
    REQUIRE(ingest_parse_text(text.data(), text.size(), true, &serial, &err) == text.size());
    for (int threads : {1, 2, 5, 16}) {
      std::vector<double> par;
      REQUIRE(ingest_parse_text_parallel(text.data(), text.size(), true, &par, threads, &err) == text.size());
      REQUIRE(par == serial);
    }
  }
  // ---- file readers ---------------------------------------------------------------------------------------------------
  const int64_t n = 137;
  {
    const std::string tpath = dir + "/m.txt", bpath = dir + "/m.f64";
    FILE* ft = std::fopen(tpath.c_str(), "w");
    FILE* fb = std::fopen(bpath.c_str(), "wb");
    REQUIRE(ft && fb);
    for (int64_t i = 0; i < n; ++i)
      for (int64_t j = 0; j < n; ++j) {
        const double v = value(i, j);
        std::fprintf(ft, "  %.17E\n", v);
        std::fwrite(&v, sizeof v, 1, fb);
      }
    std::fclose(ft);
    std::fclose(fb);
    const int64_t wanted[4][3] = {{0, n, 9}, {40, 50, 7}, {130, 7, 64}, {0, 0, 5}};
    for (int pass = 0; pass < 2; ++pass)
      for (const auto& w : wanted) {
        const int64_t first = w[0], count = w[1], cap = w[2];
        CheckSink sink(n, first, count, cap);
        const int rc = pass == 0 ? ingest_text_file(tpath.c_str(), n, sink, &err) : ingest_f64_file(bpath.c_str(), n, sink, &err);
        REQUIRE(rc == 0);
        for (int64_t i = first; i < first + count; ++i) {
          REQUIRE(sink.seen[(size_t)i]);
          for (int64_t j = 0; j < n; ++j) REQUIRE(sink.got[(size_t)i * n + j] == value(i, j));
        }
      }
    // wrong sizes: one value short, one value long, wrong order for the binary file; a missing file
    CheckSink big(n + 1, 0, n + 1, 9), small(n - 1, 0, n - 1, 9), sink(n, 0, n, 9);      // a sink's rows are as long as the order asked for
    REQUIRE(ingest_f64_file(bpath.c_str(), n + 1, big, &err) != 0);
    REQUIRE(ingest_text_file(tpath.c_str(), n + 1, big, &err) != 0);
    REQUIRE(ingest_text_file(tpath.c_str(), n - 1, small, &err) != 0);
    REQUIRE(ingest_f64_file(bpath.c_str(), n - 1, small, &err) != 0);
    REQUIRE(ingest_text_file((dir + "/nothing_here.txt").c_str(), n, sink, &err) != 0);
    REQUIRE(ingest_f64_file((dir + "/nothing_here.f64").c_str(), n, sink, &err) != 0);
  }
  std::puts("host sanitizer driver: ok");
  return 0;
}
