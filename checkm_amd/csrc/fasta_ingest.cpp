// fasta_ingest.cpp -- protein FASTA files (bins/<binId>/genes.faa) to digitised, 16-byte padded records, a file per thread,
// then laid end to end and ordered by length, a bin per thread.  Host code only.  What it replaces: the sequence-file reading hmmsearch does for every bin (process launched at
// checkm/hmmer.py:70 on the file prodigal or `-g` left at checkm/markerGeneFinder.py:113-127); the record rules are those
// CheckM itself applies to the same files (checkm/util/seqUtils.py:180-211): '>' starts a record, the name is the first
// blank-delimited word of the header, the rest of the header is the description, sequence lines are joined with blanks
// stripped, blank lines are skipped, text before the first header is ignored.
#include <algorithm>
#include <cctype>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <mutex>
#include <system_error>
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

// Host threads one ingest may use: a quarter of the machine's, at most 16 (two scan lanes ingest side by side, the gene caller and the table
// writers have threads of their own); CKM_INGEST_THREADS overrides.
int ingest_threads() {
  static const int n = [] {
    if (const char *e = getenv("CKM_INGEST_THREADS")) return std::max(1, std::min(64, atoi(e)));
    const int hw = (int)std::thread::hardware_concurrency();
    return std::max(1, std::min(16, hw / 4));
  }();
  return n;
}

// f(b) for every b in [0, n), on up to nthreads threads; bins are handed out one at a time (files differ in size).  The first exception
// is rethrown on the caller's thread after every thread has ended.
template <class F>
static void for_each_bin(uint32_t n, int nthreads, F f) {
  const unsigned nth = (unsigned)std::max(1, std::min<int>(nthreads, (int)n));
  if (nth <= 1) { for (uint32_t b = 0; b < n; ++b) f(b); return; }
  std::atomic<uint32_t> next{0};
  std::mutex m; std::exception_ptr err;
  auto body = [&] {
    for (;;) {
      const uint32_t b = next.fetch_add(1);
      if (b >= n) return;
      try { f(b); } catch (...) { std::lock_guard<std::mutex> g(m); if (!err) err = std::current_exception(); }
    }
  };
  std::vector<std::thread> th;
  for (unsigned k = 1; k < nth; ++k) {
    try { th.emplace_back(body); } catch (const std::system_error &) { break; }       // (no more threads to be had: the ones that exist do the work)
  }
  body();
  for (auto &t : th) t.join();
  if (err) std::rethrow_exception(err);
}

std::vector<FastaBin> read_fasta_bins(const char *const *paths, uint32_t nbins, int nthreads) {
  std::vector<FastaBin> bins(nbins);
  for_each_bin(nbins, nthreads, [&](uint32_t b) { parse_fasta_file(paths[b], bins[b]); });
  return bins;
}

// The files of a batch laid end to end: records, names and residues of bin b follow those of bin b - 1; 16 PADCODE bytes close the residue
// buffer.  Offsets come from prefix sums over the bins, so every bin is moved into place by its own thread (the serial loop this replaces
// cost as much as reading and digitising the files: 19 of 38 ms for 38 bins, profiles/r05F_emulated_rank_trace.txt).  Throws the first
// unreadable file's error, in bin order.  The per-file buffers are released as they are merged.
void merge_fasta_bins(std::vector<FastaBin> &bins, int nthreads, const SeqColumns &o) {
  const uint32_t nbins = (uint32_t)bins.size();
  for (auto &fb : bins) if (fb.err_code) throw Error(fb.err_code, fb.err);
  std::vector<uint64_t> dsq_at((size_t)nbins + 1, 0), rec_at((size_t)nbins + 1, 0);
  for (uint32_t b = 0; b < nbins; ++b) { dsq_at[b + 1] = dsq_at[b] + bins[b].dsq.size(); rec_at[b + 1] = rec_at[b] + bins[b].names.size(); }
  const uint64_t nrec = rec_at[nbins];
  if (nrec > 0xfffffff0ull) throw Error(CKM_ERANGE, "more than 2^32 sequences in one batch");
  o.bin_off->assign((size_t)nbins + 1, 0);
  for (uint32_t b = 0; b <= nbins; ++b) (*o.bin_off)[b] = (uint32_t)rec_at[b];
  o.names->assign(nrec, std::string()); o.descs->assign(nrec, std::string());
  o.len->assign(nrec, 0); o.off->assign(nrec, 0);
  o.dsq->assign(dsq_at[nbins] + 16, (uint8_t)PADCODE);
  for_each_bin(nbins, nthreads, [&](uint32_t b) {
    FastaBin &fb = bins[b];
    const uint64_t r0 = rec_at[b], d0 = dsq_at[b];
    for (size_t r = 0; r < fb.names.size(); ++r) {
      (*o.names)[r0 + r] = std::move(fb.names[r]); (*o.descs)[r0 + r] = std::move(fb.descs[r]);
      (*o.len)[r0 + r] = fb.len[r]; (*o.off)[r0 + r] = d0 + fb.off[r];
    }
    if (!fb.dsq.empty()) memcpy(o.dsq->data() + d0, fb.dsq.data(), fb.dsq.size());
    std::vector<uint8_t>().swap(fb.dsq);
    std::vector<std::string>().swap(fb.names); std::vector<std::string>().swap(fb.descs);
  });
  uint64_t total = 0; int maxL = 0;
  for (auto &fb : bins) { total += fb.total_res; maxL = std::max(maxL, fb.maxL); }
  *o.total_res += total; *o.maxL = std::max(*o.maxL, maxL);
}

// ONE order of all non-empty sequences: grouped by bin, longest first inside a bin, ties in file order (ckm_host.h: ckm_seqs::order);
// seq_bin, order_off and the residues of every bin beside it.  A bin per thread.
void build_seq_order(int nthreads, const SeqColumns &o) {
  const std::vector<uint32_t> &bin_off = *o.bin_off; const std::vector<int32_t> &len = *o.len;
  const uint32_t nbins = (uint32_t)bin_off.size() - 1, nseq = bin_off[nbins];
  o.seq_bin->assign(nseq, 0); o.order_off->assign((size_t)nbins + 1, 0); o.bin_res->assign(nbins, 0);
  std::vector<uint32_t> count(nbins, 0);
  for_each_bin(nbins, nthreads, [&](uint32_t b) {
    uint32_t c = 0; uint64_t res = 0;
    for (uint32_t i = bin_off[b]; i < bin_off[b + 1]; ++i) { (*o.seq_bin)[i] = b; if (len[i] > 0) { ++c; res += (uint64_t)len[i]; } }
    count[b] = c; (*o.bin_res)[b] = res;
  });
  for (uint32_t b = 0; b < nbins; ++b) (*o.order_off)[b + 1] = (*o.order_off)[b] + count[b];
  o.order->assign((*o.order_off)[nbins], 0);
  for_each_bin(nbins, nthreads, [&](uint32_t b) {
    uint32_t *first = o.order->data() + (*o.order_off)[b], *w = first;
    for (uint32_t i = bin_off[b]; i < bin_off[b + 1]; ++i) if (len[i] > 0) *w++ = i;
    std::stable_sort(first, w, [&](uint32_t x, uint32_t y) { return len[x] > len[y]; });
  });
}

}  // namespace ckm

// ---- nucleotide bins for ckm_genes_call: the files CheckM hands to prodigal by path (checkm/prodigal.py:86-93, `-i <bin>`) ----
// Record rules as checkm_amd/geneFinder.py: read_contigs_bytes states them (CheckM's own readFasta, checkm/util/seqUtils.py:180-211): a
// record begins with '>' at the start of a line, what precedes the first one is skipped, the id is the header's first blank-delimited word,
// the sequence is everything up to the next record with '\n', '\r', ' ' and '\t' removed.  A file per thread, then every bin moved into
// the batch's one text by its own thread.
namespace ckm {
namespace {
struct NucBin { std::unique_ptr<char[]> text; size_t len = 0; std::vector<uint64_t> off; std::vector<std::string> ids; std::string err; };

inline bool py_space(unsigned char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }          // bytes.split(None): space, \t \n \v \f \r

void parse_nuc_file(const char *path, NucBin &o) {
  FILE *f = fopen(path, "rb");
  if (!f) { o.err = std::string("cannot open FASTA file ") + path; return; }
  fseek(f, 0, SEEK_END); const long sz = ftell(f); fseek(f, 0, SEEK_SET);
  std::unique_ptr<char[]> buf(new char[(size_t)std::max<long>(sz, 1)]);             // (not zero-filled: 2 MB a bin)
  const size_t n = sz > 0 ? fread(buf.get(), 1, (size_t)sz, f) : 0;
  fclose(f);
  if ((long)n != sz) { o.err = std::string("short read on ") + path; return; }
  const char *d = buf.get();
  static const struct Keep { uint8_t t[256]; Keep() { for (int c = 0; c < 256; ++c) t[c] = !(c == '\n' || c == '\r' || c == ' ' || c == '\t'); } } keep;
  auto next_record = [&](size_t from) -> size_t {                 // index of the '\n' of the next "\n>" at or after `from`, or n
    for (size_t p = from; p + 1 < n;) {
      const void *q = memchr(d + p, '\n', n - 1 - p);
      if (!q) return n;
      p = (size_t)((const char *)q - d);
      if (d[p + 1] == '>') return p;
      ++p;
    }
    return n;
  };
  size_t pos;                                                      // first byte after the '>' of the current record
  if (n && d[0] == '>') pos = 1;
  else { const size_t p = next_record(0); if (p >= n) return; pos = p + 2; }
  o.text.reset(new char[n + 1]);
  char *w = o.text.get();
  for (;;) {
    const size_t end = next_record(pos);                           // the record is d[pos .. end)
    const void *nl = pos < end ? memchr(d + pos, '\n', end - pos) : nullptr;
    const size_t he = nl ? (size_t)((const char *)nl - d) : end;   // header d[pos .. he), body after the '\n'
    size_t a = pos; while (a < he && py_space((unsigned char)d[a])) ++a;
    size_t z = a; while (z < he && !py_space((unsigned char)d[z])) ++z;
    o.ids.emplace_back(d + a, z - a);
    o.off.push_back((uint64_t)(w - o.text.get()));
    if (nl) for (size_t i = he + 1; i < end; ++i) { const unsigned char c = (unsigned char)d[i]; *w = (char)c; w += keep.t[c]; }      // (branch-free: a byte per cycle)
    if (end >= n) break;
    pos = end + 2;
  }
  o.len = (size_t)(w - o.text.get());
}
}  // namespace
}  // namespace ckm

struct ckm_nuc_batch {
  std::string text; std::vector<uint64_t> contig_off, bin_bases; std::vector<uint32_t> bin_first;
  std::vector<std::string> ids; std::vector<const char *> id_ptr;
};

extern "C" int ckm_nuc_batch_read(const char *const *paths, uint32_t nbins, ckm_nuc_batch **out) {
  using namespace ckm;
  if (!out || (nbins && !paths)) { set_last_error("NULL argument"); return CKM_EINVAL; }
  *out = nullptr;
  try {
    std::vector<NucBin> bins(nbins);
    for_each_bin(nbins, ingest_threads(), [&](uint32_t b) { parse_nuc_file(paths[b], bins[b]); });
    for (auto &nb : bins) if (!nb.err.empty()) { set_last_error(nb.err); return CKM_EIO; }
    std::unique_ptr<ckm_nuc_batch> B(new ckm_nuc_batch);
    std::vector<uint64_t> text_at((size_t)nbins + 1, 0), rec_at((size_t)nbins + 1, 0);
    for (uint32_t b = 0; b < nbins; ++b) { text_at[b + 1] = text_at[b] + bins[b].len; rec_at[b + 1] = rec_at[b] + bins[b].ids.size(); }
    if (rec_at[nbins] > 0xfffffff0ull) { set_last_error("more than 2^32 contigs in one batch"); return CKM_ERANGE; }
    const size_t nrec = (size_t)rec_at[nbins];
    B->text.resize((size_t)text_at[nbins]); B->contig_off.assign(nrec + 1, 0); B->ids.assign(nrec, std::string());
    B->bin_first.assign((size_t)nbins + 1, 0); B->bin_bases.assign(nbins, 0);
    for (uint32_t b = 0; b <= nbins; ++b) B->bin_first[b] = (uint32_t)rec_at[b];
    B->contig_off[nrec] = text_at[nbins];
    for_each_bin(nbins, ingest_threads(), [&](uint32_t b) {
      NucBin &nb = bins[b];
      if (nb.len) memcpy(&B->text[0] + text_at[b], nb.text.get(), nb.len);
      for (size_t r = 0; r < nb.ids.size(); ++r) { B->contig_off[rec_at[b] + r] = text_at[b] + nb.off[r]; B->ids[rec_at[b] + r] = std::move(nb.ids[r]); }
      B->bin_bases[b] = nb.len;
      nb.text.reset();
    });
    B->id_ptr.resize(std::max<size_t>(nrec, 1), nullptr);
    for (size_t r = 0; r < nrec; ++r) B->id_ptr[r] = B->ids[r].c_str();
    *out = B.release();
    return CKM_OK;
  } catch (const std::bad_alloc &) { set_last_error("out of host memory"); return CKM_ENOMEM; }
  catch (const std::exception &e) { set_last_error(e.what()); return CKM_EINVAL; }
}
extern "C" int ckm_nuc_batch_view_get(const ckm_nuc_batch *b, ckm_nuc_batch_view *o) {
  if (!b || !o) { ckm::set_last_error("NULL argument"); return CKM_EINVAL; }
  o->text = b->text.data(); o->contig_off = b->contig_off.data(); o->bin_first = b->bin_first.data(); o->contig_ids = b->id_ptr.data();
  o->bin_bases = b->bin_bases.data(); o->ncontigs = (uint32_t)b->ids.size(); o->nbins = (uint32_t)b->bin_bases.size();
  return CKM_OK;
}
extern "C" void ckm_nuc_batch_free(ckm_nuc_batch *b) { delete b; }
