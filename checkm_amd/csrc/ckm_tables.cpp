// ckm_tables.cpp -- reader of existing domtblout text (bins/<binId>/hmmer.analyze.txt written by an earlier command):
// what HMMERParser.readHitsDOM / HmmerHitDOM do line by line in Python (checkm/hmmer.py:184-200, 255-285), done once for all
// bins of a run, on a few threads, into the column form ckm_reduce takes.  Host code only (no device is needed to parse).
//   - lines are right-stripped; the first empty line ends the table (the reference's IndexError path); '#' lines are skipped;
//   - a row is split on runs of whitespace (a leading blank yields an empty first token, as re.split does) and needs >= 23
//     tokens; tokens 22.. are re-joined with single blanks as the description;
//   - query accession '-' is replaced by the query name (hmmer.py:264-266);
//   - numbers are converted as Python's int()/float() convert them (strtol / strtod on the whole token).
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>
#include "ckm_internal.h"

using namespace ckm;

namespace {

struct BinTable {
  bool missing = false;
  std::string err;
  std::vector<std::string> tname, tacc, qname, qacc, desc;
  std::vector<int32_t> tlen, qlen, dom, ndom, hf, ht, af, at, ef, et;
  std::vector<double> fe, fs, fb, ce, ie, ds, db, acc;
};

bool to_int(const std::string &s, int32_t &v) {
  if (s.empty()) return false;
  errno = 0; char *end = nullptr;
  const long x = strtol(s.c_str(), &end, 10);
  if (errno || *end) return false;
  v = (int32_t)x; return true;
}
bool to_dbl(const std::string &s, double &v) {
  if (s.empty()) return false;
  errno = 0; char *end = nullptr;
  v = strtod(s.c_str(), &end);
  return *end == 0;           // overflow to inf / underflow to 0 are what float() returns too
}

void parse_file(const std::string &path, BinTable &t) {
  std::ifstream in(path, std::ios::binary);
  if (!in) { t.missing = true; return; }
  std::string line; int lineno = 0;
  while (std::getline(in, line)) {
    ++lineno;
    size_t n = line.size();
    while (n && (line[n - 1] == ' ' || line[n - 1] == '\t' || line[n - 1] == '\r' || line[n - 1] == '\n' || line[n - 1] == '\f' || line[n - 1] == '\v')) --n;
    line.resize(n);
    if (line.empty()) break;                  // an empty line ends the table
    if (line[0] == '#') continue;
    std::vector<std::string> tok;
    size_t i = 0;
    auto is_ws = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\f' || c == '\v'; };
    if (is_ws(line[0])) tok.emplace_back();   // re.split(r'\s+', ' x') -> ['', 'x']
    while (i < n) {
      while (i < n && is_ws(line[i])) ++i;
      if (i >= n) break;
      size_t j = i;
      while (j < n && !is_ws(line[j])) ++j;
      tok.emplace_back(line, i, j - i);
      i = j;
    }
    if (tok.size() < 23) { t.err = path + ":" + std::to_string(lineno) + ": fewer than 23 columns"; return; }
    int32_t iv[10]; double dv[8];
    const int icol[10] = {2, 5, 9, 10, 15, 16, 17, 18, 19, 20};
    const int dcol[8] = {6, 7, 8, 11, 12, 13, 14, 21};
    for (int k = 0; k < 10; ++k) if (!to_int(tok[icol[k]], iv[k])) { t.err = path + ":" + std::to_string(lineno) + ": column " + std::to_string(icol[k] + 1) + " is not an integer"; return; }
    for (int k = 0; k < 8; ++k) if (!to_dbl(tok[dcol[k]], dv[k])) { t.err = path + ":" + std::to_string(lineno) + ": column " + std::to_string(dcol[k] + 1) + " is not a number"; return; }
    t.tname.push_back(tok[0]); t.tacc.push_back(tok[1]); t.qname.push_back(tok[3]); t.qacc.push_back(tok[4] == "-" ? tok[3] : tok[4]);
    std::string d = tok[22];
    for (size_t k = 23; k < tok.size(); ++k) { d += ' '; d += tok[k]; }
    t.desc.push_back(std::move(d));
    t.tlen.push_back(iv[0]); t.qlen.push_back(iv[1]); t.dom.push_back(iv[2]); t.ndom.push_back(iv[3]);
    t.hf.push_back(iv[4]); t.ht.push_back(iv[5]); t.af.push_back(iv[6]); t.at.push_back(iv[7]); t.ef.push_back(iv[8]); t.et.push_back(iv[9]);
    t.fe.push_back(dv[0]); t.fs.push_back(dv[1]); t.fb.push_back(dv[2]); t.ce.push_back(dv[3]); t.ie.push_back(dv[4]); t.ds.push_back(dv[5]); t.db.push_back(dv[6]); t.acc.push_back(dv[7]);
  }
}

}  // namespace

struct ckm_tables {
  uint32_t nbins = 0;
  std::vector<uint64_t> bin_row_off;
  std::vector<uint8_t> missing;
  std::vector<std::string> tname, tacc, qname, qacc, desc;
  std::vector<const char *> p_tname, p_tacc, p_qname, p_qacc, p_desc;
  std::vector<uint32_t> seq, model;
  std::vector<int32_t> tlen, qlen, dom, ndom, hf, ht, af, at, ef, et;
  std::vector<double> fe, ce, ie, fs_d, ds_d, fb_d, db_d, acc_d;
  std::vector<float> fs, fb, ds, db, acc;
};

extern "C" int ckm_tables_read(const char *const *paths, uint32_t nbins, ckm_tables **out) {
  if (!out || (nbins && !paths)) { set_last_error("NULL argument"); return CKM_EINVAL; }
  *out = nullptr;
  try {
    std::vector<BinTable> bt(nbins);
    const unsigned nth = std::max(1u, std::min(8u, std::min(nbins, std::thread::hardware_concurrency())));
    std::vector<std::thread> th;
    for (unsigned k = 0; k < nth; ++k) th.emplace_back([&, k] { for (uint32_t b = k; b < nbins; b += nth) parse_file(paths[b], bt[b]); });
    for (auto &t : th) t.join();
    for (auto &b : bt) if (!b.err.empty()) { set_last_error(b.err); return CKM_EFORMAT; }
    std::unique_ptr<ckm_tables> t(new ckm_tables());
    t->nbins = nbins; t->bin_row_off.assign(nbins + 1, 0); t->missing.assign(std::max<uint32_t>(nbins, 1), 0);
    for (uint32_t b = 0; b < nbins; ++b) {
      BinTable &x = bt[b];
      t->bin_row_off[b] = t->tname.size(); t->missing[b] = x.missing ? 1 : 0;
      for (size_t r = 0; r < x.tname.size(); ++r) {
        t->tname.push_back(std::move(x.tname[r])); t->tacc.push_back(std::move(x.tacc[r])); t->qname.push_back(std::move(x.qname[r]));
        t->qacc.push_back(std::move(x.qacc[r])); t->desc.push_back(std::move(x.desc[r]));
        t->tlen.push_back(x.tlen[r]); t->qlen.push_back(x.qlen[r]); t->dom.push_back(x.dom[r]); t->ndom.push_back(x.ndom[r]);
        t->hf.push_back(x.hf[r]); t->ht.push_back(x.ht[r]); t->af.push_back(x.af[r]); t->at.push_back(x.at[r]); t->ef.push_back(x.ef[r]); t->et.push_back(x.et[r]);
        t->fe.push_back(x.fe[r]); t->ce.push_back(x.ce[r]); t->ie.push_back(x.ie[r]); t->fs_d.push_back(x.fs[r]); t->ds_d.push_back(x.ds[r]); t->fb_d.push_back(x.fb[r]); t->db_d.push_back(x.db[r]); t->acc_d.push_back(x.acc[r]);
        t->fs.push_back((float)x.fs[r]); t->fb.push_back((float)x.fb[r]); t->ds.push_back((float)x.ds[r]); t->db.push_back((float)x.db[r]); t->acc.push_back((float)x.acc[r]);
      }
    }
    const size_t n = t->tname.size();
    t->bin_row_off[nbins] = n;
    t->seq.resize(n); t->model.assign(n, UINT32_MAX);
    for (size_t r = 0; r < n; ++r) t->seq[r] = (uint32_t)r;
    auto ptrs = [](const std::vector<std::string> &v, std::vector<const char *> &p) { p.resize(std::max<size_t>(v.size(), 1), nullptr); for (size_t i = 0; i < v.size(); ++i) p[i] = v[i].c_str(); };
    ptrs(t->tname, t->p_tname); ptrs(t->tacc, t->p_tacc); ptrs(t->qname, t->p_qname); ptrs(t->qacc, t->p_qacc); ptrs(t->desc, t->p_desc);
    *out = t.release();
    return CKM_OK;
  } catch (const std::exception &e) { set_last_error(e.what()); return CKM_ENOMEM; }
}

extern "C" int ckm_tables_assign_models(ckm_tables *t, const char *const *keys, uint32_t nkeys, uint64_t *unknown) {
  if (!t || (nkeys && !keys)) { set_last_error("NULL argument"); return CKM_EINVAL; }
  std::unordered_map<std::string, uint32_t> idx;
  for (uint32_t k = 0; k < nkeys; ++k) if (keys[k]) idx.emplace(keys[k], k);       // first slot wins, as a dict lookup by accession would
  uint64_t miss = 0;
  for (size_t r = 0; r < t->qacc.size(); ++r) { auto it = idx.find(t->qacc[r]); if (it == idx.end()) { t->model[r] = UINT32_MAX; ++miss; } else t->model[r] = it->second; }
  if (unknown) *unknown = miss;
  return CKM_OK;
}

extern "C" int ckm_tables_get(const ckm_tables *t, ckm_table_columns *o) {
  if (!t || !o) { set_last_error("NULL argument"); return CKM_EINVAL; }
  memset(o, 0, sizeof(*o));
  ckm_hit_columns &c = o->cols;
  c.n = t->tname.size(); c.nbins = t->nbins; c.bin_row_off = t->bin_row_off.data();
  c.seq = t->seq.data(); c.model = t->model.data(); c.tlen = t->tlen.data(); c.qlen = t->qlen.data();
  c.full_evalue = t->fe.data(); c.full_score = t->fs.data(); c.full_bias = t->fb.data(); c.dom_idx = t->dom.data(); c.ndom = t->ndom.data();
  c.c_evalue = t->ce.data(); c.i_evalue = t->ie.data(); c.dom_score = t->ds.data(); c.dom_bias = t->db.data();
  c.hmm_from = t->hf.data(); c.hmm_to = t->ht.data(); c.ali_from = t->af.data(); c.ali_to = t->at.data(); c.env_from = t->ef.data(); c.env_to = t->et.data();
  c.acc = t->acc.data(); c.target_name = t->p_tname.data(); c.full_score_d = t->fs_d.data(); c.dom_score_d = t->ds_d.data();
  o->target_accession = t->p_tacc.data(); o->query_name = t->p_qname.data(); o->query_accession = t->p_qacc.data(); o->description = t->p_desc.data();
  o->full_bias_d = t->fb_d.data(); o->dom_bias_d = t->db_d.data(); o->acc_d = t->acc_d.data();
  o->bin_missing = t->missing.data();
  return CKM_OK;
}

extern "C" void ckm_tables_free(ckm_tables *t) { delete t; }
