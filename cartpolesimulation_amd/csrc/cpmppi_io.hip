// cpmppi_io.hip — host code only: the recording writer of SURVEY.md §8f N2 (cpmppi_write_recordings).
//
// The reference writes one CSV file per experiment with Python's csv module (CartPole/csv_logger.py:10-33,125-159: a comment
// block, the column names of CartPole/__init__.py:221-259, one row per saved time step), numbers in Python's float repr.  A
// batched run of E experiments on the device ends with E such files; formatting them through Python's csv writer took 1.0 s
// for 256 experiments of 10 s whose device loop took 0.11 s (round 3).  This unit formats and writes them natively, one
// thread per file, byte for byte what csv.writer produces for the same values (tests/test_recording.py compares the files):
//   * a number is written as Python's repr(float) of the DOUBLE it converts to: the shortest digit string that reads back
//     to the same double (std::to_chars), laid out by CPython's rule (Python/pystrtod.c format_float_short, mode 'r': fixed
//     notation while -4 < decimal point <= 16, else d.ddde+XX with at least two exponent digits; ".0" after an integer);
//   * rows end with "\r\n" (csv.writer's default line terminator), fields are joined with ',' and never need quoting
//     (numbers only); the comment block and the column-name row come from the caller as ready-made bytes.
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "cpmppi.h"
#include "cpmppi_internal.hpp"

namespace {

// repr(float) of CPython >= 3.1 into `out` (at most 32 bytes); returns the length.
int py_repr(double x, char* out) {
  if (std::isnan(x)) { memcpy(out, "nan", 3); return 3; }
  if (std::isinf(x)) { const char* s = x < 0 ? "-inf" : "inf"; const int n = (int)strlen(s); memcpy(out, s, n); return n; }
  char* p = out;
  if (std::signbit(x)) { *p++ = '-'; x = -x; }
  if (x == 0.0) { memcpy(p, "0.0", 3); return (int)(p - out) + 3; }
  // shortest round-trip digits in scientific form: d[.ddd]e[+-]XX
  char sci[40];
  const auto r = std::to_chars(sci, sci + sizeof(sci), x, std::chars_format::scientific);
  const char* e = sci;
  while (e < r.ptr && *e != 'e') ++e;
  char digits[24];
  int nd = 0;
  for (const char* c = sci; c < e; ++c)
    if (*c != '.') digits[nd++] = *c;
  int exp10 = 0;
  {
    const char* c = e + 1;
    bool neg = false;
    if (*c == '+') ++c; else if (*c == '-') { neg = true; ++c; }
    for (; c < r.ptr; ++c) exp10 = exp10 * 10 + (*c - '0');
    if (neg) exp10 = -exp10;
  }
  const int decpt = exp10 + 1;                 // position of the decimal point relative to the digit string
  if (decpt > -4 && decpt <= 16) {             // fixed notation
    if (decpt <= 0) {
      *p++ = '0'; *p++ = '.';
      for (int i = 0; i < -decpt; ++i) *p++ = '0';
      memcpy(p, digits, nd); p += nd;
    } else if (decpt >= nd) {
      memcpy(p, digits, nd); p += nd;
      for (int i = nd; i < decpt; ++i) *p++ = '0';
      *p++ = '.'; *p++ = '0';
    } else {
      memcpy(p, digits, decpt); p += decpt;
      *p++ = '.';
      memcpy(p, digits + decpt, nd - decpt); p += nd - decpt;
    }
  } else {                                     // exponent notation: d[.ddd]e+XX
    *p++ = digits[0];
    if (nd > 1) { *p++ = '.'; memcpy(p, digits + 1, nd - 1); p += nd - 1; }
    *p++ = 'e';
    int ex = decpt - 1;
    if (ex < 0) { *p++ = '-'; ex = -ex; } else { *p++ = '+'; }
    char tmp[8];
    int k = 0;
    do { tmp[k++] = (char)('0' + ex % 10); ex /= 10; } while (ex);
    if (k < 2) tmp[k++] = '0';
    while (k) *p++ = tmp[--k];
  }
  return (int)(p - out);
}

struct Job {
  const char* const* paths;
  uint32_t E, T;
  const char* preamble;
  size_t preamble_len;
  const float *states, *Q, *aDD, *xDD, *u, *tp, *te, *L;
  double m_pole, dt;
};

// One recording: columns of CartPole/__init__.py:221-259 in order - time, angle, angleD, angleDD, angle_cos, angle_sin,
// position, positionD, positionDD, Q_calculated, Q_applied, Q_ccrc, u, target_position, target_equilibrium, L,
// L_for_controller, m_pole, m_pole_for_controller, vertical_angle_offset (0), its cos (1) and sin (0), Q_update_time (0).
bool write_one(const Job& j, uint32_t e, std::string& buf) {
  buf.clear();
  buf.append(j.preamble, j.preamble_len);
  char tail[256], num[40];
  int tl = 0;
  auto add = [&](double v) { tl += py_repr(v, tail + tl); tail[tl++] = ','; };
  tail[tl++] = ',';
  add((double)j.tp[e]); add((double)j.te[e]); add((double)j.L[e]); add((double)j.L[e]); add(j.m_pole); add(j.m_pole);
  add(0.0); add(1.0); add(0.0);
  tl += py_repr(0.0, tail + tl);
  tail[tl++] = '\r'; tail[tl++] = '\n';
  const uint32_t E = j.E;
  for (uint32_t t = 0; t < j.T; ++t) {
    const float* s = j.states + ((size_t)t * E + e) * 6;
    const size_t i = (size_t)t * E + e;
    const double q = (double)j.Q[i];
    const double cols[13] = {(double)t * j.dt, (double)s[0], (double)s[1], (double)j.aDD[i], (double)s[2], (double)s[3], (double)s[4],
                             (double)s[5], (double)j.xDD[i], q, q, t == 0 ? 0.0 : (double)j.Q[i - E], (double)j.u[i]};
    for (int c = 0; c < 13; ++c) {
      const int n = py_repr(cols[c], num);
      if (c) buf.push_back(',');
      buf.append(num, n);
    }
    buf.append(tail, tl);
  }
  FILE* f = fopen(j.paths[e], "ab");            // ("a" as csv_logger.py opens it: the caller has made the name unique)
  if (!f) return false;
  const bool ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size();
  return (fclose(f) == 0) && ok;
}

}  // namespace

extern "C" {

int cpmppi_write_recordings(const char* const* paths, uint32_t E, uint32_t T, const char* preamble, size_t preamble_len,
                            const float* states, const float* Q, const float* angleDD, const float* positionDD, const float* u,
                            const float* target_position, const float* target_equilibrium, const float* L, double m_pole,
                            double dt_control, int n_threads) {
  if (!paths || E == 0 || !preamble || !states || !Q || !angleDD || !positionDD || !u || !target_position || !target_equilibrium || !L)
    return cpmppi_internal_fail(nullptr, CPMPPI_ERR_BAD_ARG, "cpmppi_write_recordings: null argument");
  for (uint32_t e = 0; e < E; ++e)
    if (!paths[e]) return cpmppi_internal_fail(nullptr, CPMPPI_ERR_BAD_ARG, "cpmppi_write_recordings: null path");
  const Job j{paths, E, T, preamble, preamble_len, states, Q, angleDD, positionDD, u, target_position, target_equilibrium, L, m_pole, dt_control};
  unsigned nt = n_threads > 0 ? (unsigned)n_threads : std::thread::hardware_concurrency();
  if (nt == 0) nt = 1;
  if (nt > 32) nt = 32;
  if (nt > E) nt = E;
  std::vector<int> failed(nt, -1);
  auto work = [&](unsigned w) {
    std::string buf;
    buf.reserve(preamble_len + (size_t)T * 400);
    for (uint32_t e = w; e < E; e += nt)
      if (!write_one(j, e, buf) && failed[w] < 0) failed[w] = (int)e;
  };
  if (nt == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (unsigned w = 0; w < nt; ++w) th.emplace_back(work, w);
    for (auto& t : th) t.join();
  }
  for (unsigned w = 0; w < nt; ++w)
    if (failed[w] >= 0)
      return cpmppi_internal_fail(nullptr, CPMPPI_ERR_BAD_ARG, std::string("cpmppi_write_recordings: cannot write ") + paths[failed[w]]);
  return CPMPPI_OK;
}

// tests: repr(float) as this unit formats it (compared with Python's own on random doubles)
int cpmppi_debug_py_repr(double x, char* out32) { return py_repr(x, out32); }

}  // extern "C"
