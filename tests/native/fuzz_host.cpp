// fuzz_host.cpp -- sanitizer harness for the host-only parsers of libcheckm_hip (tests/test_native_sanitize.py builds it with
// -fsanitize=address,undefined from checkm_amd/csrc/host_profile.cpp + ckm_tables.cpp; no device, no HIP).
//   fuzz_host hmm  <valid.hmm>        <work_dir> <rounds> <seed>
//   fuzz_host dom  <valid.domtblout>  <work_dir> <rounds> <seed>
//   fuzz_host fasta <valid.faa>       <work_dir> <rounds> <seed>     (prints a digest of the undamaged file first)
// Every round damages a copy of the valid file (truncation, byte flips, line drops, token damage) and feeds it to the reader;
// the reader must either succeed or fail with an Error / a non-zero status -- never crash or trip a sanitizer.  Prints counts.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "ckm_internal.h"

namespace ckm { static std::string g_last; void set_last_error(const std::string &m) { g_last = m; } }

static uint64_t g_state = 1;
static uint32_t rnd() { g_state = g_state * 6364136223846793005ULL + 1442695040888963407ULL; return (uint32_t)(g_state >> 33); }

static std::string slurp(const char *path) { std::ifstream in(path, std::ios::binary); std::stringstream ss; ss << in.rdbuf(); return ss.str(); }

static std::string damage(const std::string &src) {
  std::string s = src;
  const int kind = rnd() % 6;
  if (s.empty()) return s;
  if (kind == 0) s.resize(rnd() % s.size());                                                  // truncation
  else if (kind == 1) { for (int k = 0, n = 1 + rnd() % 8; k < n; ++k) s[rnd() % s.size()] = (char)(rnd() & 0xff); }     // byte flips
  else if (kind == 2) { for (int k = 0, n = 1 + rnd() % 4; k < n; ++k) { size_t a = rnd() % s.size(), b = s.find('\n', a); if (b == std::string::npos) b = s.size(); s.erase(a, b - a); } }   // line tails dropped
  else if (kind == 3) { for (int k = 0, n = 1 + rnd() % 6; k < n; ++k) { size_t a = rnd() % s.size(); s[a] = " \t\n-.e*9"[rnd() % 8]; } }    // separators and number bits
  else if (kind == 4) { size_t a = rnd() % s.size(), len = rnd() % 200; s.insert(a, s.substr(rnd() % s.size(), len)); }                  // duplicated fragment
  else { size_t a = rnd() % s.size(); s.insert(a, std::string(1 + rnd() % 40, "0123456789"[rnd() % 10])); }                              // huge number
  return s;
}

int main(int argc, char **argv) {
  if (argc < 6) { fprintf(stderr, "usage: fuzz_host hmm|dom <valid file> <work dir> <rounds> <seed>\n"); return 2; }
  const std::string mode = argv[1], valid = slurp(argv[2]), work = argv[3];
  const int rounds = atoi(argv[4]); g_state = strtoull(argv[5], nullptr, 10) * 2654435761ULL + 1;
  const std::string path = work + "/fuzz_input.txt";
  int ok = 0, rejected = 0;
  for (int r = -1; r < rounds; ++r) {
    const std::string text = r < 0 ? valid : damage(valid);     // round -1: the undamaged file must be accepted
    { std::ofstream out(path, std::ios::binary); out << text; }
    if (mode == "hmm") {
      try {
        std::vector<ckm::HostHMM> hs = ckm::read_hmm_file(path);
        for (auto &h : hs) { ckm::HostProfile p = ckm::configure_profile(h); (void)ckm::len_config(p, 300, true); }
        ++ok;
      } catch (const ckm::Error &e) { ++rejected; if (r < 0) { fprintf(stderr, "valid HMM file rejected: %s\n", e.what()); return 1; } }
    } else if (mode == "fasta") {
      // the same file three times over three threads: the per-file results must agree with each other
      const char *paths[3] = {path.c_str(), path.c_str(), path.c_str()};
      std::vector<ckm::FastaBin> bins = ckm::read_fasta_bins(paths, 3, 3);
      for (auto &b : bins) {
        if (b.err_code) { fprintf(stderr, "FASTA file refused: %s\n", b.err.c_str()); if (r < 0) return 1; }
        if (b.names != bins[0].names || b.len != bins[0].len || b.dsq != bins[0].dsq || b.off != bins[0].off) { fprintf(stderr, "threads disagree\n"); return 1; }
        if (b.names.size() != b.len.size() || b.names.size() != b.off.size() || b.names.size() != b.descs.size()) { fprintf(stderr, "ragged record columns\n"); return 1; }
        for (size_t k = 0; k < b.len.size(); ++k) if (b.off[k] % 16 || b.off[k] + (uint64_t)b.len[k] > b.dsq.size()) { fprintf(stderr, "record outside its buffer\n"); return 1; }
      }
      if (bins[0].err_code) ++rejected; else ++ok;
      if (r < 0) {
        uint64_t h = 1469598103934665603ULL;
        auto mix = [&](const void *p, size_t n) { const unsigned char *c = (const unsigned char *)p; for (size_t k = 0; k < n; ++k) { h ^= c[k]; h *= 1099511628211ULL; } };
        const ckm::FastaBin &b = bins[0];
        for (size_t k = 0; k < b.names.size(); ++k) { mix(b.names[k].data(), b.names[k].size()); mix("|", 1); mix(b.descs[k].data(), b.descs[k].size()); mix("|", 1); mix(b.dsq.data() + b.off[k], (size_t)b.len[k]); mix("\n", 1); }
        printf("{\"nseq\": %zu, \"total_res\": %llu, \"maxL\": %d, \"bytes\": %zu, \"fnv\": \"%016llx\"}\n", b.names.size(), (unsigned long long)b.total_res, b.maxL, b.dsq.size(), (unsigned long long)h);
      }
    } else {
      const char *paths[2] = {path.c_str(), "/nonexistent/ckm_fuzz_missing.txt"};
      ckm_tables *t = nullptr;
      const int rc = ckm_tables_read(paths, 2, &t);
      if (rc == 0) {
        ckm_table_columns c; memset(&c, 0, sizeof(c));
        if (ckm_tables_get(t, &c) != 0) { fprintf(stderr, "ckm_tables_get failed on an accepted table\n"); return 1; }
        const char *keys[2] = {"PF00001.1", "TIGR00001"}; uint64_t unknown = 0;
        (void)ckm_tables_assign_models(t, keys, 2, &unknown);
        ckm_tables_free(t); ++ok;
      } else { ++rejected; if (r < 0) { fprintf(stderr, "valid table rejected: %s\n", ckm::g_last.c_str()); return 1; } }
    }
  }
  printf("{\"mode\": \"%s\", \"accepted\": %d, \"rejected\": %d}\n", mode.c_str(), ok, rejected);
  return 0;
}
