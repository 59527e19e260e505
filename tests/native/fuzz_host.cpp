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

// merge mode: <valid.faa> <work_dir> <nfiles> <seed>.  nfiles variants of the valid file (cut at different lengths, one empty, one with empty
// records) are read, then laid end to end and ordered on 1 thread and on 7: both must equal the serial loop written out below (what
// ckm_seqs_from_fasta did before the merge went parallel).  Prints the three digests and the times.
#include <algorithm>
#include <chrono>
struct Cols {
  std::vector<uint32_t> bin_off, seq_bin, order, order_off; std::vector<int32_t> len; std::vector<uint64_t> off, bin_res; std::vector<uint8_t> dsq;
  std::vector<std::string> names, descs; uint64_t total_res = 0; int maxL = 0;
  ckm::SeqColumns view() { return ckm::SeqColumns{&bin_off, &seq_bin, &order, &order_off, &len, &off, &bin_res, &dsq, &names, &descs, &total_res, &maxL}; }
  bool operator==(const Cols &o) const {
    return bin_off == o.bin_off && seq_bin == o.seq_bin && order == o.order && order_off == o.order_off && len == o.len && off == o.off && bin_res == o.bin_res &&
           dsq == o.dsq && names == o.names && descs == o.descs && total_res == o.total_res && maxL == o.maxL;
  }
};
static void serial_reference(std::vector<ckm::FastaBin> bins, Cols &s) {
  const uint32_t nbins = (uint32_t)bins.size();
  s.bin_off.assign(nbins + 1, 0);
  uint64_t pos = 0;
  for (uint32_t b = 0; b < nbins; ++b) {
    ckm::FastaBin &fb = bins[b];
    for (size_t r = 0; r < fb.names.size(); ++r) { s.names.push_back(fb.names[r]); s.descs.push_back(fb.descs[r]); s.len.push_back(fb.len[r]); s.off.push_back(pos + fb.off[r]); }
    s.dsq.insert(s.dsq.end(), fb.dsq.begin(), fb.dsq.end());
    pos += fb.dsq.size(); s.total_res += fb.total_res; s.maxL = std::max(s.maxL, fb.maxL);
    s.bin_off[b + 1] = (uint32_t)s.names.size();
  }
  s.dsq.resize(pos + 16, (uint8_t)ckm::PADCODE);
  const uint32_t nseq = (uint32_t)s.names.size();
  s.seq_bin.resize(nseq);
  for (uint32_t b = 0; b < nbins; ++b) for (uint32_t i = s.bin_off[b]; i < s.bin_off[b + 1]; ++i) s.seq_bin[i] = b;
  s.order_off.assign(nbins + 1, 0); s.bin_res.assign(nbins, 0);
  for (uint32_t b = 0; b < nbins; ++b) {
    s.order_off[b] = (uint32_t)s.order.size();
    const size_t first = s.order.size();
    for (uint32_t i = s.bin_off[b]; i < s.bin_off[b + 1]; ++i) if (s.len[i] > 0) { s.order.push_back(i); s.bin_res[b] += (uint64_t)s.len[i]; }
    std::stable_sort(s.order.begin() + first, s.order.end(), [&](uint32_t x, uint32_t y) { return s.len[x] > s.len[y]; });
  }
  s.order_off[nbins] = (uint32_t)s.order.size();
}
static int merge_mode(const std::string &valid, const std::string &work, int nfiles) {
  std::vector<std::string> files;
  for (int k = 0; k < nfiles; ++k) {
    std::string text = valid;
    if (k == 1) text.clear();                                                            // an empty file
    else if (k == 2) text = ">only_a_header\n>another one with a description\n" + valid;    // records without residues
    else if (k > 2) { size_t cut = text.size() / 4 + rnd() % (3 * text.size() / 4); cut = text.rfind("\n>", cut); if (cut != std::string::npos && cut > 0) text.resize(cut + 1); }
    files.push_back(work + "/merge_" + std::to_string(k) + ".faa");
    std::ofstream out(files.back(), std::ios::binary); out << text;
  }
  std::vector<const char *> paths; for (auto &f : files) paths.push_back(f.c_str());
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  Cols ref; serial_reference(ckm::read_fasta_bins(paths.data(), (uint32_t)paths.size(), 1), ref);
  double t_read[2], t_merge[2], t_order[2];
  const int threads[2] = {1, 7};
  for (int v = 0; v < 2; ++v) {
    Cols c;
    auto t0 = now();
    std::vector<ckm::FastaBin> bins = ckm::read_fasta_bins(paths.data(), (uint32_t)paths.size(), threads[v]);
    auto t1 = now();
    ckm::merge_fasta_bins(bins, threads[v], c.view());
    auto t2 = now();
    ckm::build_seq_order(threads[v], c.view());
    auto t3 = now();
    t_read[v] = ms(t0, t1); t_merge[v] = ms(t1, t2); t_order[v] = ms(t2, t3);
    if (!(c == ref)) { fprintf(stderr, "merge on %d thread(s) differs from the serial loop\n", threads[v]); return 1; }
    for (auto &fb : bins) if (!fb.dsq.empty() || !fb.names.empty()) { fprintf(stderr, "per-file buffers not released\n"); return 1; }
  }
  // an unreadable file among the others: the error of the first such bin, nothing else
  paths[nfiles / 2] = "/nonexistent/ckm_merge_missing.faa";
  {
    Cols c; bool threw = false;
    std::vector<ckm::FastaBin> bins = ckm::read_fasta_bins(paths.data(), (uint32_t)paths.size(), 7);
    try { ckm::merge_fasta_bins(bins, 7, c.view()); } catch (const ckm::Error &e) { threw = std::string(e.what()).find("ckm_merge_missing") != std::string::npos; }
    if (!threw) { fprintf(stderr, "an unreadable file was not reported\n"); return 1; }
  }
  printf("{\"mode\": \"merge\", \"files\": %d, \"nseq\": %zu, \"bytes\": %zu, \"empty_records\": %zu, \"read_ms\": [%.2f, %.2f], \"merge_ms\": [%.2f, %.2f], \"order_ms\": [%.2f, %.2f]}\n",
         nfiles, ref.names.size(), ref.dsq.size(), (size_t)std::count(ref.len.begin(), ref.len.end(), 0), t_read[0], t_read[1], t_merge[0], t_merge[1], t_order[0], t_order[1]);
  return 0;
}

int main(int argc, char **argv) {
  if (argc >= 6 && std::string(argv[1]) == "merge") { g_state = strtoull(argv[5], nullptr, 10) * 2654435761ULL + 1; return merge_mode(slurp(argv[2]), argv[3], atoi(argv[4])); }
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
