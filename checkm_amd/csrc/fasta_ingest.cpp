// fasta_ingest.cpp -- protein FASTA files (bins/<binId>/genes.faa) to digitised, 16-byte padded records, one thread per
// file.  Host code only.  What it replaces: the sequence-file reading hmmsearch does for every bin (process launched at
// checkm/hmmer.py:70 on the file prodigal or `-g` left at checkm/markerGeneFinder.py:113-127); the record rules are those
// CheckM itself applies to the same files (checkm/util/seqUtils.py:180-211): '>' starts a record, the name is the first
// blank-delimited word of the header, the rest of the header is the description, sequence lines are joined with blanks
// stripped, blank lines are skipped, text before the first header is ignored.
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <thread>
#include "ckm_internal.h"

namespace ckm {

static void parse_fasta_file(const char *path, FastaBin &o) {
  FILE *f = fopen(path, "rb");
  if (!f) { o.err_code = CKM_EIO; o.err = std::string("cannot open FASTA file ") + path; return; }
  fseek(f, 0, SEEK_END); const long sz = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<char> buf((size_t)std::max<long>(sz, 0));
  const size_t got = sz > 0 ? fread(buf.data(), 1, (size_t)sz, f) : 0;
  fclose(f);
  if ((long)got != sz) { o.err_code = CKM_EIO; o.err = std::string("short read on ") + path; return; }
  o.dsq.reserve(got + got / 16 + 64);
  uint64_t pos = 0, start = 0; bool open = false;
  auto close_seq = [&]() -> bool {                  // pad the record that just ended to a 16-byte boundary
    const uint64_t L = pos - start;
    if (L > 100000) { o.err_code = CKM_ERANGE; o.err = std::string("sequence longer than 100000 residues in ") + path; return false; }
    o.len.push_back((int32_t)L); o.total_res += L; o.maxL = std::max(o.maxL, (int)L);
    const uint64_t padded = (L + 15) & ~(uint64_t)15;
    o.dsq.resize(start + padded, (uint8_t)PADCODE);
    pos = start + padded;
    return true;
  };
  size_t i = 0;
  while (i < got) {
    size_t e = i; while (e < got && buf[e] != '\n') ++e;
    size_t le = e; if (le > i && buf[le - 1] == '\r') --le;
    if (le > i && buf[i] == '>') {
      if (open && !close_seq()) return;
      size_t n0 = i + 1, n1 = n0; while (n1 < le && !isspace((unsigned char)buf[n1])) ++n1;
      size_t d0 = n1; while (d0 < le && isspace((unsigned char)buf[d0])) ++d0;
      o.names.emplace_back(buf.data() + n0, n1 - n0);
      o.descs.emplace_back(buf.data() + d0, le - d0);
      start = pos; o.off.push_back(start); open = true;
    } else if (open && le > i) {
      size_t a = i, z = le;                           // strip blanks at both ends of the line
      while (a < z && isspace((unsigned char)buf[a])) ++a;
      while (z > a && isspace((unsigned char)buf[z - 1])) --z;
      if (z > a) { o.dsq.resize(pos + (z - a)); digitize(buf.data() + a, z - a, o.dsq.data() + pos); pos += z - a; }
    }
    i = e + 1;
  }
  if (open) close_seq();
}

std::vector<FastaBin> read_fasta_bins(const char *const *paths, uint32_t nbins, int nthreads) {
  std::vector<FastaBin> bins(nbins);
  const unsigned nth = (unsigned)std::max(1, std::min<int>(nthreads, (int)nbins));
  if (nth <= 1) { for (uint32_t b = 0; b < nbins; ++b) parse_fasta_file(paths[b], bins[b]); return bins; }
  std::vector<std::thread> th;
  for (unsigned k = 0; k < nth; ++k) th.emplace_back([&, k] { for (uint32_t b = k; b < nbins; b += nth) parse_fasta_file(paths[b], bins[b]); });
  for (auto &t : th) t.join();
  return bins;
}

}  // namespace ckm
