// host_profile.cpp -- HMMER3/f ASCII reader and search-profile configuration (host side).
//
// Replaces what hmmsearch does with <hmmfile> before it scores anything (reference call site
// checkm/hmmer.py:70; the header subset CheckM itself reads: checkm/hmmerModelParser.py:54-83).
// Produces, per model, the tables the HIP kernels consume:
//   SSV/MSV   biased unsigned byte costs, re-expressed as signed "bias - cost" words in the
//             16-lane striped LDS image the SSV kernel reads with ds_read_b128;
//   Viterbi   signed 16-bit match scores + transition words + D->D prefix sums;
//   Fwd/Bwd   float odds ratios in the canonical 64-lane blocked layout.
// All transcendental functions are evaluated here, on the host, once per model.
#include <thread>
#include <memory>
#include "ckm_internal.h"
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <algorithm>

namespace ckm {

static const char kAlphabet[] = "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~";
static const double kLn2 = 0.69314718055994529;

// HMMER's default amino background (Swiss-Prot 50.8)
static const float kBg[K] = {0.0787945f, 0.0151600f, 0.0535222f, 0.0668298f, 0.0397062f, 0.0695071f, 0.0229198f,
                             0.0590092f, 0.0594422f, 0.0963728f, 0.0237718f, 0.0414386f, 0.0482904f, 0.0395639f,
                             0.0540978f, 0.0683364f, 0.0540687f, 0.0673417f, 0.0114135f, 0.0304133f};

static bool in_degeneracy(int sym, int res) {
  const char r = kAlphabet[res];
  switch (kAlphabet[sym]) {
    case 'B': return r == 'D' || r == 'N';
    case 'J': return r == 'I' || r == 'L';
    case 'Z': return r == 'E' || r == 'Q';
    case 'O': return r == 'K';
    case 'U': return r == 'C';
    case 'X': return true;
    default:  return sym == res;
  }
}

void digitize(const char *text, uint64_t n, uint8_t *dsq) {
  // (called from the FASTA reader's threads: the table is built by a thread-safe static initialiser)
  struct Lut {
    uint8_t v[256];
    Lut() {
      for (int c = 0; c < 256; ++c) v[c] = 26;  // anything unknown scores as X
      for (int i = 0; i < KP; ++i) { v[(unsigned char)kAlphabet[i]] = (uint8_t)i; v[(unsigned char)std::tolower((unsigned char)kAlphabet[i])] = (uint8_t)i; }
    }
  };
  static const Lut lut;
  for (uint64_t i = 0; i < n; ++i) dsq[i] = lut.v[(unsigned char)text[i]];
}

// ------------------------------------------------------------------------------------------------
// file reader
// ------------------------------------------------------------------------------------------------
// a probability field: "*" (zero) or -ln p >= 0.  hmmsearch converts with atof and takes whatever comes out; a token that is
// not a number, or a "probability" above 1, can only come from a damaged file, so it is refused here instead of scored.
static void parse_numbers(const std::string &line, size_t skip, float *dst, int n, int lineno, const std::string &path) {
  // whitespace-separated tokens, scanned in place (a 2000-model database is 20 million of these fields)
  const char *p = line.c_str();
  auto next_tok = [&](const char *&b, const char *&e) -> bool {
    while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n' || *p == '\v' || *p == '\f') ++p;
    if (!*p) return false;
    b = p;
    while (*p && !(*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n' || *p == '\v' || *p == '\f')) ++p;
    e = p;
    return true;
  };
  const char *b = nullptr, *e = nullptr;
  for (size_t i = 0; i < skip; ++i) (void)next_tok(b, e);
  for (int i = 0; i < n; ++i) {
    if (!next_tok(b, e)) throw Error(CKM_EFORMAT, path + ":" + std::to_string(lineno) + ": expected " + std::to_string(n) + " numeric fields");
    bool ok;
    if (e - b == 1 && *b == '*') { dst[i] = 0.0f; ok = true; }
    else {
      char *end = nullptr;
      const double v = strtod(b, &end);            // (the line is NUL-terminated and a token ends at whitespace, where strtod stops too)
      ok = !(end == b || end != e || !(v >= 0.0) || std::isinf(v));
      if (ok) dst[i] = expf((float)(-1.0 * v));
    }
    if (!ok) throw Error(CKM_EFORMAT, path + ":" + std::to_string(lineno) + ": '" + std::string(b, e) + "' is not a probability field (-ln p >= 0 or *)");
  }
}

static void match_occupancy(const HostHMM &h, std::vector<float> &mocc, std::vector<float> *iocc) {
  const int M = h.M;
  mocc.assign(M + 1, 0.f);
  mocc[1] = h.t[0 * 7 + 1] + h.t[0 * 7 + 0];
  for (int k = 2; k <= M; ++k)
    mocc[k] = mocc[k - 1] * (h.t[(k - 1) * 7 + 0] + h.t[(k - 1) * 7 + 1]) + (1.0f - mocc[k - 1]) * h.t[(k - 1) * 7 + 5];
  if (iocc) {
    iocc->assign(M + 1, 0.f);
    (*iocc)[0] = h.t[1] / h.t[3];
    for (int k = 1; k < M; ++k) (*iocc)[k] = mocc[k] * h.t[k * 7 + 1] / h.t[k * 7 + 3];
  }
}

// every record of a stream (a whole file, or the slice of it one reader thread got); `lineno` = lines before the stream's first one
static std::vector<HostHMM> read_hmm_stream(std::istream &in, const std::string &path, int lineno) {
  std::vector<HostHMM> out;
  std::string line;
  auto fail = [&](const std::string &m) -> Error { return Error(CKM_EFORMAT, path + ":" + std::to_string(lineno) + ": " + m); };
  auto next = [&]() { if (!std::getline(in, line)) throw fail("truncated record"); ++lineno; };
  while (std::getline(in, line)) {
    ++lineno;
    if (line.compare(0, 7, "HMMER3/") != 0) {
      if (line.find_first_not_of(" \t\r\n") != std::string::npos) throw fail("expected a HMMER3/ record");
      continue;
    }
    HostHMM h;
    bool in_body = false;
    while (!in_body) {
      next();
      if (line.compare(0, 4, "HMM ") == 0 || line.compare(0, 4, "HMM\t") == 0) { in_body = true; break; }
      std::istringstream is(line);
      std::string tag;
      if (!(is >> tag)) continue;
      std::string rest;
      std::getline(is, rest);
      size_t b = rest.find_first_not_of(" \t"), e = rest.find_last_not_of(" \t\r\n");
      rest = (b == std::string::npos) ? std::string() : rest.substr(b, e - b + 1);
      if (tag == "NAME") h.name = rest;
      else if (tag == "ACC") { h.acc = rest; h.has_acc = true; }
      else if (tag == "DESC") { h.desc = rest; h.has_desc = true; }
      else if (tag == "LENG") h.M = atoi(rest.c_str());
      else if (tag == "ALPH") { std::string a = rest; std::transform(a.begin(), a.end(), a.begin(), ::tolower); if (a != "amino") throw fail("only amino-acid profiles are supported"); }
      else if (tag == "GA" || tag == "TC" || tag == "NC") {
        double a, c;
        std::string r2 = rest; std::replace(r2.begin(), r2.end(), ';', ' ');
        if (sscanf(r2.c_str(), "%lf %lf", &a, &c) != 2) throw fail("bad " + tag + " line");
        if (tag == "GA") { h.ga[0] = a; h.ga[1] = c; h.has_ga = true; }
        if (tag == "TC") { h.tc[0] = a; h.tc[1] = c; h.has_tc = true; }
        if (tag == "NC") { h.nc[0] = a; h.nc[1] = c; h.has_nc = true; }
      } else if (tag == "STATS") {
        char loc[32], kind[32]; float v1, v2;
        if (sscanf(rest.c_str(), "%31s %31s %f %f", loc, kind, &v1, &v2) != 4 || strcmp(loc, "LOCAL") != 0) throw fail("bad STATS line");
        if (!strcmp(kind, "MSV")) { h.evparam[0] = v1; h.evparam[1] = v2; h.stats_mask |= 1; }
        else if (!strcmp(kind, "VITERBI")) { h.evparam[2] = v1; h.evparam[3] = v2; h.stats_mask |= 2; }
        else if (!strcmp(kind, "FORWARD")) { h.evparam[4] = v1; h.evparam[5] = v2; h.stats_mask |= 4; }
        else throw fail("unknown STATS kind");
      }
    }
    if (h.M <= 0) throw fail("LENG missing or not positive");
    if (h.name.empty()) throw fail("NAME missing");
    if (h.stats_mask != 7) throw fail("model " + h.name + " is not calibrated (STATS LOCAL MSV/VITERBI/FORWARD required)");
    const int M = h.M;
    h.t.assign((size_t)(M + 1) * 7, 0.f);
    h.mat.assign((size_t)(M + 1) * K, 0.f);
    h.ins.assign((size_t)(M + 1) * K, 0.f);
    next();  // transition legend
    next();
    {
      std::istringstream is(line); std::string first; is >> first;
      if (first == "COMPO") { parse_numbers(line, 1, h.compo, K, lineno, path); h.has_compo = true; next(); }
    }
    parse_numbers(line, 0, &h.ins[0], K, lineno, path);
    next();
    parse_numbers(line, 0, &h.t[0], 7, lineno, path);
    h.mat[0] = 1.0f;
    for (int k = 1; k <= M; ++k) {
      next();
      { char *endp = nullptr; const long idx = strtol(line.c_str(), &endp, 10); if (endp == line.c_str() || idx != k) throw fail("node index mismatch"); }
      parse_numbers(line, 1, &h.mat[(size_t)k * K], K, lineno, path);
      next(); parse_numbers(line, 0, &h.ins[(size_t)k * K], K, lineno, path);
      next(); parse_numbers(line, 0, &h.t[(size_t)k * 7], 7, lineno, path);
    }
    next();
    if (line.compare(0, 2, "//") != 0) throw fail("expected // at end of record");
    if (!h.has_compo) {
      std::vector<float> mocc, iocc;
      match_occupancy(h, mocc, &iocc);
      for (int x = 0; x < K; ++x) h.compo[x] = 0.f;
      for (int x = 0; x < K; ++x) h.compo[x] += h.ins[x] * iocc[0];
      for (int k = 1; k <= M; ++k)
        for (int x = 0; x < K; ++x) { h.compo[x] += h.mat[(size_t)k * K + x] * mocc[k]; h.compo[x] += h.ins[(size_t)k * K + x] * iocc[k]; }
      float s = 0.f; for (int x = 0; x < K; ++x) s += h.compo[x];
      for (int x = 0; x < K; ++x) h.compo[x] /= s;
      h.has_compo = true;
    }
    out.push_back(std::move(h));
  }
  return out;
}

// The text of a marker database is hundreds of MB (checkm.hmm: ~230 MB for 2000 models) and records are independent: the file is
// read once, cut at lines that start a record ("HMMER3/"), and the slices are parsed by up to 8 threads; records keep file order and
// an error reports the line of the FILE.  (The reference's hmmsearch re-reads and re-parses the file for every bin.)
std::vector<HostHMM> read_hmm_file(const std::string &path) {
  std::string text;
  {
    std::ifstream in(path, std::ios::binary);
    if (!in) throw Error(CKM_EIO, "cannot open HMM file " + path);
    in.seekg(0, std::ios::end);
    const std::streamoff sz = in.tellg();
    in.seekg(0, std::ios::beg);
    if (sz > 0) { text.resize((size_t)sz); in.read(&text[0], sz); text.resize((size_t)in.gcount()); }
  }
  std::vector<size_t> starts;            // offsets of lines beginning a record
  for (size_t pos = 0; pos < text.size();) {
    if (text.compare(pos, 7, "HMMER3/") == 0) starts.push_back(pos);
    const size_t nl = text.find('\n', pos);
    if (nl == std::string::npos) break;
    pos = nl + 1;
  }
  const size_t nrec = starts.size();
  const size_t nthreads = std::max<size_t>(1, std::min<size_t>({(size_t)16, (size_t)std::thread::hardware_concurrency(), text.size() / (4u << 20) + 1, std::max<size_t>(nrec, 1)}));
  std::vector<size_t> cut(nthreads + 1, text.size());
  cut[0] = 0;                                                     // (whatever precedes the first record belongs to the first slice, which rejects it)
  for (size_t t = 1; t < nthreads; ++t) cut[t] = starts[t * nrec / nthreads];
  std::vector<std::vector<HostHMM>> parts(nthreads);
  std::vector<std::unique_ptr<Error>> errs(nthreads);
  auto work = [&](size_t t) {
    try {
      const int lineno = (int)std::count(text.begin(), text.begin() + (std::ptrdiff_t)cut[t], '\n');
      std::istringstream in(text.substr(cut[t], cut[t + 1] - cut[t]));
      parts[t] = read_hmm_stream(in, path, lineno);
    } catch (const Error &e) { errs[t].reset(new Error(e)); }
  };
  if (nthreads == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (size_t t = 0; t < nthreads; ++t) th.emplace_back(work, t);
    for (auto &x : th) x.join();
  }
  std::vector<HostHMM> out;
  for (size_t t = 0; t < nthreads; ++t) {
    if (errs[t]) throw Error(*errs[t]);                           // the first error in file order
    for (auto &h : parts[t]) out.push_back(std::move(h));
  }
  if (out.empty()) throw Error(CKM_EFORMAT, path + ": no HMMER3 records");
  return out;
}

// ------------------------------------------------------------------------------------------------
// statistics
// ------------------------------------------------------------------------------------------------
double gumbel_surv(double x, double mu, double lambda) {
  const double y = lambda * (x - mu);
  const double ey = -exp(-y);
  if (fabs(ey) < 5e-9) return -ey;
  return 1 - exp(ey);
}
double exp_surv(double x, double mu, double lambda) { return (x < mu) ? 1.0 : exp(-lambda * (x - mu)); }
double exp_logsurv(double x, double mu, double lambda) { return (x < mu) ? 0.0 : -lambda * (x - mu); }

float flogsum(float a, float b) {
  // (the workers call this concurrently: the table is built by a thread-safe static initialiser)
  struct Table { float v[16000]; Table() { for (int i = 0; i < 16000; ++i) v[i] = (float)log(1. + exp((double)-i / 1000.)); } };
  static const Table table;
  const float *tbl = table.v;
  const float mx = std::max(a, b), mn = std::min(a, b);
  return (mn == -INFINITY || (mx - mn) >= 15.7f) ? mx : mx + tbl[(int)((mx - mn) * 1000.f)];
}

// smallest float x for which  !(surv(x) > F) : the filter "P > F -> reject" becomes "score < thr -> reject".
template <class Fn>
static float passing_threshold(Fn surv, double F) {
  auto ord = [](float f) { int32_t i; memcpy(&i, &f, 4); return (i < 0) ? (int32_t)0x80000000 - i : i; };
  auto unord = [](int32_t o) { int32_t i = (o < 0) ? (int32_t)0x80000000 - o : o; float f; memcpy(&f, &i, 4); return f; };
  int64_t lo = ord(-1.0e6f), hi = ord(1.0e6f);   // surv(lo) > F (reject), surv(hi) <= F (pass)
  if (!(surv((double)unord((int32_t)lo)) > F)) return -INFINITY;
  if (surv((double)unord((int32_t)hi)) > F) return INFINITY;
  while (hi - lo > 1) {
    const int64_t mid = lo + (hi - lo) / 2;
    if (surv((double)unord((int32_t)mid)) > F) lo = mid; else hi = mid;
  }
  return unord((int32_t)hi);
}

// ------------------------------------------------------------------------------------------------
// profile configuration
// ------------------------------------------------------------------------------------------------
static const int kCanonQ[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64};
int canon_Q(int M) {
  for (int q : kCanonQ) if (q * NL >= M) return q;
  return -1;
}
static const int kVitQH[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 24, 32};
int vit_QH_for(int M) {
  for (int q : kVitQH) if (q * 128 >= M) return q;
  return -1;
}
static const int kSsvQ[] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 26, 28, 30, 32, 36, 40, 48, 56, 64};
int ssv_Q_for(int M) {
  for (int q : kSsvQ) if (q * 32 >= M) return q;
  return M <= 4096 ? 65 : -1;          // 65 (kSsvNone): no SSV instance holds the model; the exact MSV kernel takes all of its pairs
}

// IEEE binary16 bits of k/256 for an integer |k| <= 2047 (exact: k has at most 11 significant bits)
static uint16_t half_bits_of_256th(int k) {
  if (k == 0) return 0;
  const uint16_t sign = k < 0 ? 0x8000u : 0u;
  unsigned a = (unsigned)(k < 0 ? -k : k);
  int e = 0;
  while ((a >> (e + 1)) != 0) ++e;                 // a = 1.xxx * 2^e
  const unsigned frac = (a << (10 - e)) & 0x3ffu;  // e <= 10
  return (uint16_t)(sign | (unsigned)((e - 8 + 15) << 10) | frac);
}

static uint8_t byte_cost(float scale_b, float sc) {
  sc = -1.0f * roundf(scale_b * sc);
  return (sc > 255.f) ? 255 : (uint8_t)(int)sc;
}
static uint8_t byte_cost_biased(float scale_b, int bias, float sc) {
  sc = -1.0f * roundf(scale_b * sc);
  return (sc > (float)(255 - bias)) ? 255 : (uint8_t)((int)sc + bias);
}
static int16_t word_score(float scale_w, float sc) {
  sc = roundf(scale_w * sc);
  if (sc >= 32767.0f) return 32767;
  if (sc <= -32768.0f) return -32768;
  return (int16_t)sc;
}

LenCfg len_config(const HostProfile &p, int L, bool multihit) {
  LenCfg c;
  const float nj = multihit ? 1.0f : 0.0f;
  c.move = (2.0f + nj) / ((float)L + 2.0f + nj);
  c.loop = 1.0f - c.move;
  c.w_move = word_score(p.scale_w, logf(c.move));
  c.tjb_b = byte_cost(p.scale_b, logf(3.0f / (float)(L + 3)));
  c.p1 = (float)L / (float)(L + 1);
  c.nullsc = (float)((float)L * log((double)c.p1) + log(1. - (double)c.p1));
  c.bias_tail = (float)L * logf(c.p1) + logf(1.0f - c.p1);
  return c;
}

HostProfile configure_profile(const HostHMM &h) {
  const int M = h.M;
  HostProfile p;
  p.M = M;
  p.fbQ = canon_Q(M);
  p.vitQH = vit_QH_for(M);
  p.ssvQ = ssv_Q_for(M);
  if (p.fbQ < 0 || p.ssvQ < 0 || p.vitQH < 0) throw Error(CKM_ERANGE, "model " + h.name + ": LENG " + std::to_string(M) + " exceeds the supported maximum of 4096");
  const float NEG = -INFINITY;
  // generic log-odds: transitions indexed by source node 0..M-1; entry B->M_k stored at k-1
  std::vector<float> gBM(M + 1, NEG), gMM(M + 1, NEG), gIM(M + 1, NEG), gDM(M + 1, NEG), gMD(M + 1, NEG), gDD(M + 1, NEG), gMI(M + 1, NEG), gII(M + 1, NEG);
  {
    std::vector<float> occ;
    match_occupancy(h, occ, nullptr);
    float Z = 0.f;
    for (int k = 1; k <= M; ++k) Z += occ[k] * (float)(M - k + 1);
    for (int k = 1; k <= M; ++k) gBM[k - 1] = (float)log(occ[k] / Z);
  }
  for (int k = 1; k < M; ++k) {
    const float *t = &h.t[(size_t)k * 7];
    gMM[k] = (float)log(t[0]); gMI[k] = (float)log(t[1]); gMD[k] = (float)log(t[2]);
    gIM[k] = (float)log(t[3]); gII[k] = (float)log(t[4]); gDM[k] = (float)log(t[5]); gDD[k] = (float)log(t[6]);
  }
  // match scores for all 29 symbols
  std::vector<float> msc((size_t)KP * (M + 1), NEG);
  for (int k = 1; k <= M; ++k) {
    float sc[KP];
    for (int x = 0; x < K; ++x) sc[x] = (float)log((double)h.mat[(size_t)k * K + x] / kBg[x]);
    sc[20] = NEG; sc[27] = NEG; sc[28] = NEG;
    for (int x = 21; x <= 26; ++x) {
      float num = 0.f, den = 0.f;
      for (int y = 0; y < K; ++y) if (in_degeneracy(x, y)) { num += sc[y] * kBg[y]; den += kBg[y]; }
      sc[x] = num / den;
    }
    for (int x = 0; x < KP; ++x) msc[(size_t)x * (M + 1) + k] = sc[x];
  }
  // ---- MSV / SSV ----
  {
    float mx = 0.0f;
    for (int x = 0; x < K; ++x) for (int k = 1; k <= M; ++k) mx = std::max(mx, msc[(size_t)x * (M + 1) + k]);
    p.scale_b = (float)(3.0 / kLn2);
    p.base_b = 190;
    p.bias_b = byte_cost(p.scale_b, -1.0f * mx);
    p.tbm_b = byte_cost(p.scale_b, logf(2.0f / ((float)M * (float)(M + 1))));
    p.tec_b = byte_cost(p.scale_b, logf(0.5f));
    p.rbv.assign((size_t)KP * (M + 1), 255);
    for (int x = 0; x < KP; ++x) for (int k = 1; k <= M; ++k) p.rbv[(size_t)x * (M + 1) + k] = byte_cost_biased(p.scale_b, p.bias_b, msc[(size_t)x * (M + 1) + k]);
    // LDS image for the SSV kernel.  16 lanes per sequence, Q packed registers per lane, position
    // p = q + Q*(2*z + h) (HMMER-style striping so the diagonal move is a register rename);
    // lane z fetches registers 4g..4g+3 of symbol x with one ds_read_b128 at (g*30 + x)*256 + z*16 (group-major, so
    // that symbol*256 + lane offset is a byte permute and g rides in the instruction's offset field).
    const int Q = p.ssvQ > 64 ? 0 : p.ssvQ, Qg = (Q + 3) / 4;         // (no LDS image for a model the SSV kernel cannot hold)
    p.ssv_tbl.assign((size_t)NROWS * Qg * 16 * 8, (int16_t)(p.bias_b - 255));
    for (int x = 0; x < KP; ++x)
      for (int q = 0; q < Q; ++q)
        for (int z = 0; z < 16; ++z)
          for (int hh = 0; hh < 2; ++hh) {
            const int pos = q + Q * (2 * z + hh);
            const int k = pos + 1;
            const int cost = (k <= M) ? p.rbv[(size_t)x * (M + 1) + k] : 255;
            const size_t idx = (((size_t)(q / 4) * NROWS + x) * 16 + z) * 8 + (size_t)(q % 4) * 2 + hh;
            p.ssv_tbl[idx] = (int16_t)(p.bias_b - cost);
          }
    p.ssv_tbl_h.resize(p.ssv_tbl.size());
    for (size_t i = 0; i < p.ssv_tbl.size(); ++i) p.ssv_tbl_h[i] = half_bits_of_256th(p.ssv_tbl[i]);
    // the 8-lane image (ssv_kernel_h8): position p = q + Q8 * (2 z + half), z = 0..7; both copies of a (group, symbol) row are the same
    static const int kSsv8Q[] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 26, 28, 30, 32};
    p.ssv8Q = 0;
    for (int q : kSsv8Q) if (q * 16 >= M) { p.ssv8Q = q; break; }
    if (p.ssv8Q) {
      const int Q8 = p.ssv8Q, Qg8 = (Q8 + 3) / 4;
      p.ssv8_tbl_h.assign((size_t)NROWS * Qg8 * 128, half_bits_of_256th(p.bias_b - 255));
      for (int x = 0; x < KP; ++x) for (int q = 0; q < Q8; ++q) for (int z = 0; z < 8; ++z) for (int hh = 0; hh < 2; ++hh) {
        const int k = q + Q8 * (2 * z + hh) + 1;
        const int cost = (k <= M) ? p.rbv[(size_t)x * (M + 1) + k] : 255;
        for (int c = 0; c < 2; ++c)
          p.ssv8_tbl_h[(((((size_t)(q / 4) * NROWS + x) * 2 + c) * 8 + z) * 8) + (size_t)(q % 4) * 2 + hh] = half_bits_of_256th(p.bias_b - cost);
      }
    }
  }
  // ---- Viterbi filter ----
  {
    const int QH = p.vitQH, CELLS = 2 * QH;              // cells per lane
    p.scale_w = (float)(500.0 / kLn2);
    p.base_w = 12000;
    auto cap = [](int16_t v, int16_t mx) { return v <= mx ? v : mx; };
    // word of cell c (node c+1) in table `which`; out-of-model cells are -32768 (= -inf)
    auto emis = [&](int x, int c) -> int16_t { return (c < M) ? word_score(p.scale_w, msc[(size_t)x * (M + 1) + c + 1]) : (int16_t)-32768; };
    auto trans = [&](int which, int c) -> int16_t {
      const int k = c + 1;
      if (k > M) return -32768;
      switch (which) {
        case 0: return cap(word_score(p.scale_w, gBM[k - 1]), 0);
        case 1: return cap(word_score(p.scale_w, gMM[k - 1]), 0);
        case 2: return cap(word_score(p.scale_w, gIM[k - 1]), 0);
        case 3: return cap(word_score(p.scale_w, gDM[k - 1]), 0);
        case 4: return (k < M) ? cap(word_score(p.scale_w, gMD[k]), 0) : (int16_t)-32768;
        case 5: return (k < M) ? cap(word_score(p.scale_w, gMI[k]), 0) : (int16_t)-32768;
        case 6: return (k < M) ? cap(word_score(p.scale_w, gII[k]), -1) : (int16_t)-32768;
        default: return (k < M) ? word_score(p.scale_w, gDD[k]) : (int16_t)-32768;
      }
    };
    auto pack = [](int16_t lo, int16_t hi) { return (uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16); };
    p.vit_e.assign((size_t)NROWS * QH * NL, pack(-32768, -32768));
    for (int x = 0; x < KP; ++x)
      for (int j = 0; j < QH; ++j)
        for (int z = 0; z < NL; ++z)
          p.vit_e[((size_t)x * QH + j) * NL + z] = pack(emis(x, z * CELLS + j), emis(x, z * CELLS + j + QH));
    p.vit_t.assign((size_t)8 * QH * NL, pack(-32768, -32768));
    for (int w = 0; w < 8; ++w)
      for (int j = 0; j < QH; ++j)
        for (int z = 0; z < NL; ++z)
          p.vit_t[((size_t)w * QH + j) * NL + z] = pack(trans(w, z * CELLS + j), trans(w, z * CELLS + j + QH));
    // the same words striped over 16 lanes (lane z owns cells z*2Q .. z*2Q+2Q-1, register j = (cell j, cell j+Q) of the lane)
    p.vit16Q = 0;
    static const int kVit16Q[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16};
    for (int q : kVit16Q) if (q * 32 >= M) { p.vit16Q = q; break; }
    if (p.vit16Q) {
      const int Q = p.vit16Q, C16 = 2 * Q;
      p.vit16_e.assign((size_t)NROWS * Q * 16, pack(-32768, -32768));
      for (int x = 0; x < KP; ++x) for (int j = 0; j < Q; ++j) for (int z = 0; z < 16; ++z)
        p.vit16_e[((size_t)x * Q + j) * 16 + z] = pack(emis(x, z * C16 + j), emis(x, z * C16 + j + Q));
      p.vit16_t.assign((size_t)8 * Q * 16, pack(-32768, -32768));
      for (int w = 0; w < 8; ++w) for (int j = 0; j < Q; ++j) for (int z = 0; z < 16; ++z)
        p.vit16_t[((size_t)w * Q + j) * 16 + z] = pack(trans(w, z * C16 + j), trans(w, z * C16 + j + Q));
    }
    p.wE_loop = word_score(p.scale_w, (float)(-kLn2));
    p.wE_move = word_score(p.scale_w, (float)(-kLn2));
  }
  // ---- Forward / Backward ----
  {
    const int Mp = p.fbQ * NL;
    p.rf.assign((size_t)NROWS * Mp, 0.f);
    const int FQ = p.fbQ;   // q-major, as above
    for (int x = 0; x < KP; ++x) for (int k = 1; k <= M; ++k) { const int c = k - 1; p.rf[(size_t)x * Mp + (c % FQ) * NL + c / FQ] = expf(msc[(size_t)x * (M + 1) + k]); }
    p.ftr.assign((size_t)8 * Mp, 0.f);
    for (int k = 1; k <= M; ++k) {
      const int i = k - 1;
      p.ftr[0 * Mp + i] = expf(gBM[k - 1]); p.ftr[1 * Mp + i] = expf(gMM[k - 1]);
      p.ftr[2 * Mp + i] = expf(gIM[k - 1]); p.ftr[3 * Mp + i] = expf(gDM[k - 1]);
      if (k < M) {
        p.ftr[4 * Mp + i] = expf(gMI[k]); p.ftr[5 * Mp + i] = expf(gII[k]);
        p.ftr[6 * Mp + i] = expf(gMD[k]); p.ftr[7 * Mp + i] = expf(gDD[k]);
      }
    }
    p.fE_loop = expf((float)(-kLn2));
    p.fE_move = expf((float)(-kLn2));
  }
  // ---- bias filter (two-state composition HMM) ----
  {
    const float L0 = 400.0f, L1 = (float)M / 8.0f;
    p.bt00 = L0 / (L0 + 1.0f); p.bt01 = 1.0f / (L0 + 1.0f);
    p.bt10 = 1.0f / (L1 + 1.0f); p.bt11 = L1 / (L1 + 1.0f);
    p.bpi0 = 0.999f; p.bpi1 = 0.001f;
    for (int x = 0; x < NROWS; ++x) p.beo1[x] = 1.0f;
    for (int x = 0; x < K; ++x) p.beo1[x] = h.compo[x] / kBg[x];
    for (int x = 21; x <= 26; ++x) {
      float num = 0.f, den = 0.f;
      for (int y = 0; y < K; ++y) if (in_degeneracy(x, y)) { num += h.compo[y]; den += kBg[y]; }
      p.beo1[x] = (den > 0.f) ? num / den : 0.f;
    }
  }
  // ---- filter thresholds in score space ----
  {
    const double mmu = h.evparam[0], mlam = h.evparam[1], vmu = h.evparam[2], vlam = h.evparam[3], ftau = h.evparam[4], flam = h.evparam[5];
    p.thr_msv_f1 = passing_threshold([&](double x) { return gumbel_surv(x, mmu, mlam); }, 0.02);
    p.thr_msv_f2 = passing_threshold([&](double x) { return gumbel_surv(x, mmu, mlam); }, 1e-3);
    p.thr_vit_f2 = passing_threshold([&](double x) { return gumbel_surv(x, vmu, vlam); }, 1e-3);
    p.thr_fwd_f3 = passing_threshold([&](double x) { return exp_surv(x, ftau, flam); }, 1e-5);
    // the first F1 test in NATS: bits(usc, nullsc) = (float)((double)(usc - nullsc) / ln 2) is monotone in the float usc - nullsc, so the
    // smallest such float whose bits reach thr_msv_f1 decides exactly as the bit-space test does, without a double divide per pair
    const float thr = p.thr_msv_f1;
    p.thr_msv_f1_nat = passing_threshold([&](double v) { return ((float)(v / kLn2) >= thr) ? 0.0 : 1.0; }, 0.5);
  }
  return p;
}

}  // namespace ckm
