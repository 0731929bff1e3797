// cpmppi_io.hip — host code only: the recording writer of SURVEY.md §8f N2 (cpmppi_write_recordings).
//
// The reference writes one CSV file per experiment with Python's csv module (CartPole/csv_logger.py:10-58,125-159: a comment
// block, the column names of CartPole/__init__.py:221-259, one row per saved time step).  csv.writer renders a Python float with
// repr() and anything else with str(): what a column looks like depends on the TYPE the simulator holds it in, and the
// recording the reference itself wrote (tests/golden/schedule.npz "csv_rows", made by running its CartPole class) shows which:
//   * Python floats (time, Q_calculated, target_position, L, m_pole, the vertical-angle-offset columns): repr(float) of the
//     DOUBLE - the shortest digit string that reads back to the same double (std::to_chars), laid out by CPython's rule
//     (Python/pystrtod.c format_float_short, mode 'r': fixed notation while -4 < decimal point <= 16, else d.ddde+XX with at
//     least two exponent digits; ".0" after an integer);
//   * numpy float32 scalars (the state, angleDD, positionDD, Q_applied, Q_ccrc, u): str(numpy.float32) - the shortest digit
//     string that reads back to the same FLOAT (numpy's Dragon4 in unique mode = std::to_chars(float)), same layout rule;
//   * target_equilibrium: an int; L_for_controller / m_pole_for_controller: the controller informer's 'true' / 'default';
//     Q_update_time: None -> empty
//     before the first controller update inside the loop.
// Rows end with "\r\n" (csv.writer's default line terminator), fields are joined with ',' and never need quoting; the comment
// block and the column-name row come from the caller as ready-made bytes.  A batched run of E experiments ends with E files:
// formatting them through Python took 1.0 s for 256 experiments of 10 s whose device loop took 0.11 s (round 3); this unit
// writes them natively, one thread per file (6 ms).  Nothing is appended to or overwritten: a file is created exclusively
// under a temporary name and renamed when complete, and a failing call removes what it had written (advisor, round 4).
#include <unistd.h>

#include <cerrno>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <exception>
#include <string>
#include <thread>
#include <vector>

#include "cpmppi.h"
#include "cpmppi_internal.hpp"

namespace {

// Lay out the shortest round-trip digits of a finite positive number (`sci` = its std::to_chars scientific form).  notation:
// -1 = CPython's repr rule (by the DIGITS: fixed while -4 < decimal point <= 16), 0 = fixed, 1 = exponent (numpy decides by the
// VALUE: fixed while 1e-4 <= |x| < 1e16 - float32(1e-4) = 9.9999997e-05 prints as 1e-04, Python would print 0.0001).
int layout_short(const char* sci, const char* sci_end, char* p0, int notation) {
  char* p = p0;
  const char* e = sci;
  while (e < sci_end && *e != 'e') ++e;
  char digits[24];
  int nd = 0;
  for (const char* c = sci; c < e; ++c)
    if (*c != '.') digits[nd++] = *c;
  int exp10 = 0;
  {
    const char* c = e + 1;
    bool neg = false;
    if (*c == '+') ++c; else if (*c == '-') { neg = true; ++c; }
    for (; c < sci_end; ++c) exp10 = exp10 * 10 + (*c - '0');
    if (neg) exp10 = -exp10;
  }
  const int decpt = exp10 + 1;                 // position of the decimal point relative to the digit string
  if (notation < 0 ? (decpt > -4 && decpt <= 16) : notation == 0) {   // fixed notation
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
  return (int)(p - p0);
}

template <typename T>
int short_repr(T x, char* out, bool numpy_rule) {
  if (std::isnan(x)) { memcpy(out, "nan", 3); return 3; }
  if (std::isinf(x)) { const char* s = x < 0 ? "-inf" : "inf"; const int n = (int)strlen(s); memcpy(out, s, n); return n; }
  char* p = out;
  if (std::signbit(x)) { *p++ = '-'; x = -x; }
  if (x == 0) { memcpy(p, "0.0", 3); return (int)(p - out) + 3; }
  char sci[48];
  const auto r = std::to_chars(sci, sci + sizeof(sci), x, std::chars_format::scientific);
  const int notation = numpy_rule ? (((long double)x >= 1.e-4L && (long double)x < 1.e16L) ? 0 : 1) : -1;   // (numpy: scalartypes.c.src, *type_str_either)
  return (int)(p - out) + layout_short(sci, r.ptr, p, notation);
}

// repr(float) of CPython >= 3.1 / str(numpy.float32) into `out` (at most 32 bytes); returns the length.
int py_repr(double x, char* out) { return short_repr<double>(x, out, false); }
int np_str_f32(float x, char* out) { return short_repr<float>(x, out, true); }

struct Job {
  const char* const* paths;
  const char* preamble;
  size_t preamble_len;
  const cpmppi_recording* r;
};

// One recording: columns of CartPole/__init__.py:221-259 in order - time, angle, angleD, angleDD, angle_cos, angle_sin,
// position, positionD, positionDD, Q_calculated, Q_applied, Q_ccrc, u, target_position, target_equilibrium, L,
// L_for_controller, m_pole, m_pole_for_controller, vertical_angle_offset (0.0), its cos (1.0) and sin (0.0), Q_update_time.
// -> 0, or errno of the failing call; `tmp` receives the temporary name (removed here on failure).
int write_one(const Job& j, uint32_t e, std::string& buf, std::string& tmp) {
  const cpmppi_recording& r = *j.r;
  buf.clear();
  buf.append(j.preamble, j.preamble_len);
  char num[40], mp[40], qt[40];
  const int mpl = py_repr(r.m_pole, mp), qtl = py_repr(r.q_update_time, qt);
  const uint32_t E = r.E;
  auto f32col = [&](float v) { buf.push_back(','); buf.append(num, np_str_f32(v, num)); };
  auto f64col = [&](double v) { buf.push_back(','); buf.append(num, py_repr(v, num)); };
  for (uint32_t t = 0; t < r.rows; ++t) {
    const size_t i = (size_t)t * E + e;
    const float* s = r.states + i * 6;
    const float* dd = r.dd + i * 2;
    const float q = r.Q[i];
    buf.append(num, py_repr(r.time[t], num));
    f32col(s[0]); f32col(s[1]); f32col(dd[0]); f32col(s[2]); f32col(s[3]); f32col(s[4]); f32col(s[5]); f32col(dd[1]);
    const float qa = r.Q_applied ? r.Q_applied[i] : q;
    f64col((double)q); f32col(qa); f32col(r.Q_ccrc[i]); f32col(r.u_max * qa);
    f64col(r.target_position[i]);
    buf.push_back(',');
    buf.append(num, snprintf(num, sizeof(num), "%d", (int)r.target_equilibrium[i]));
    f64col((double)r.L[i]);
    const bool told = !r.informed || r.informed[i];
    const char* inf = told ? ",true" : ",default";
    const size_t infl = told ? 5 : 8;
    buf.append(inf, infl);
    if (r.m_pole_rows) f64col((double)r.m_pole_rows[i]); else { buf.push_back(','); buf.append(mp, mpl); }
    buf.append(inf, infl);
    if (r.angle_offset) {
      const double* ao = r.angle_offset + i * 3;
      f64col(ao[0]); f64col(ao[1]); f64col(ao[2]);
      buf.push_back(',');
    } else {
      buf.append(",0.0,1.0,0.0,", 13);
    }
    if (t >= r.first_update_row) buf.append(qt, qtl);
    buf.append("\r\n", 2);
  }
  // never append to or overwrite a recording: exclusive create under a temporary name, rename when complete
  if (access(j.paths[e], F_OK) == 0) return EEXIST;
  tmp = std::string(j.paths[e]) + ".part";
  remove(tmp.c_str());                         // (a leftover of a killed run; the name is this writer's own)
  FILE* f = fopen(tmp.c_str(), "wbx");
  if (!f) return errno ? errno : EIO;
  int err = 0;
  if (fwrite(buf.data(), 1, buf.size(), f) != buf.size()) err = errno ? errno : EIO;
  if (fclose(f) != 0 && !err) err = errno ? errno : EIO;
  if (!err && link(tmp.c_str(), j.paths[e]) != 0) {                               // (link fails if the name appeared meanwhile; rename would replace it)
    const int le = errno;
    // a file system without hard links (some network / FUSE mounts): rename, after one more look that the name is still free
    if ((le == EPERM || le == EOPNOTSUPP || le == ENOSYS || le == EMLINK) && access(j.paths[e], F_OK) != 0) {
      if (rename(tmp.c_str(), j.paths[e]) == 0) return 0;
      err = errno ? errno : EIO;
      remove(tmp.c_str());
      return err;
    }
    err = le ? le : EIO;
  }
  remove(tmp.c_str());
  return err;
}

}  // namespace

extern "C" {

int cpmppi_write_recordings(const char* const* paths, const char* preamble, size_t preamble_len, const cpmppi_recording* rec,
                            int n_threads) {
  if (!paths || !preamble || !rec || rec->E == 0 || !rec->time || !rec->states || !rec->dd || !rec->Q || !rec->Q_ccrc ||
      !rec->target_position || !rec->target_equilibrium || !rec->L)
    return cpmppi_internal_fail(nullptr, CPMPPI_ERR_BAD_ARG, "cpmppi_write_recordings: null argument");
  const uint32_t E = rec->E;
  for (uint32_t e = 0; e < E; ++e)
    if (!paths[e]) return cpmppi_internal_fail(nullptr, CPMPPI_ERR_BAD_ARG, "cpmppi_write_recordings: null path");
  std::vector<char> written;
  try {                                          // (threads and row buffers: no C++ exception leaves the C ABI)
  {
    std::vector<std::string> names(paths, paths + E);                    // two files of one call under one name would share a temporary
    std::sort(names.begin(), names.end());
    const auto dup = std::adjacent_find(names.begin(), names.end());
    if (dup != names.end())
      return cpmppi_internal_fail(nullptr, CPMPPI_ERR_BAD_ARG, "cpmppi_write_recordings: the same path twice: " + *dup);
  }
  const Job j{paths, preamble, preamble_len, rec};
  unsigned nt = n_threads > 0 ? (unsigned)n_threads : std::thread::hardware_concurrency();
  if (nt == 0) nt = 1;
  if (nt > 32) nt = 32;
  if (nt > E) nt = E;
  std::vector<int> failed(nt, -1), failed_errno(nt, 0);
  written.assign(E, 0);
  auto work = [&](unsigned w) {
    uint32_t e = w;
    try {
      std::string buf, tmp;
      buf.reserve(preamble_len + (size_t)rec->rows * 400);
      for (; e < E; e += nt) {
        const int err = write_one(j, e, buf, tmp);
        if (err == 0) written[e] = 1;
        else if (failed[w] < 0) { failed[w] = (int)e; failed_errno[w] = err; }
      }
    } catch (const std::exception&) {            // (a row buffer that could not grow)
      if (failed[w] < 0) { failed[w] = (int)e; failed_errno[w] = ENOMEM; }
    }
  };
  {
    std::vector<std::thread> th;
    th.reserve(nt);
    unsigned started = 0;
    try {
      for (; started + 1 < nt; ++started) th.emplace_back(work, started);
    } catch (const std::exception&) {}           // no more threads to be had: this thread takes the rest
    for (unsigned w = started; w < nt; ++w) work(w);
    for (auto& t : th) t.join();
  }
  std::string msg;
  int n_failed = 0;
  for (unsigned w = 0; w < nt; ++w)
    if (failed[w] >= 0) {
      if (!n_failed) msg = std::string("cpmppi_write_recordings: cannot write ") + paths[failed[w]] + ": " + strerror(failed_errno[w]);
      ++n_failed;
    }
  if (n_failed) {
    for (uint32_t e = 0; e < E; ++e)            // a failing call leaves nothing behind: a retry starts from a clean directory
      if (written[e]) remove(paths[e]);
    return cpmppi_internal_fail(nullptr, CPMPPI_ERR_IO, msg + " (the recordings this call had written were removed)");
  }
  return CPMPPI_OK;
  } catch (const std::exception&) {              // out of host memory (every worker has been joined or was never started)
    for (uint32_t e = 0; e < E && e < written.size(); ++e)
      if (written[e]) remove(paths[e]);
    return CPMPPI_ERR_NOMEM;
  }
}

// tests: repr(float) / str(numpy.float32) as this unit formats them (compared with Python's and numpy's own on random values)
int cpmppi_debug_py_repr(double x, char* out32) { return py_repr(x, out32); }
int cpmppi_debug_np_str_f32(float x, char* out32) { return np_str_f32(x, out32); }

}  // extern "C"
