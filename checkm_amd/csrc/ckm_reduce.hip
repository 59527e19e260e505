// ckm_reduce.hip -- the reduce half: hit vetting/filtering in the reference's exact (bug-compatible)
// order on the host, and the collocated-marker-set counting kernel on gfx950.
//
// Reference being replaced (paths in the CheckM tree):
//   ResultsManager.vetHit                       checkm/resultsParser.py:340-377
//   ResultsManager.addHit                       checkm/resultsParser.py:379-399
//   PFAM.filterHitsFromSameClan                 checkm/util/pfam.py:86-147
//   ResultsManager.identifyAdjacentMarkerGenes  checkm/resultsParser.py:401-479
//   ResultsManager.geneCounts                   checkm/resultsParser.py:513-537
//   MarkerSet.genomeCheck                       checkm/markerSets.py:206-238
// The filters are a few hundred list operations per bin whose OUTCOME depends on Python dict/list
// order; they stay on the host as ordered containers.  The counting is the data-parallel part and
// runs on the device: one thread per collocated set, one wave-level histogram per bin.
#include <hip/hip_runtime.h>
#include <functional>
#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>
#include "ckm_internal.h"

namespace ckm {

#define HIPCHK(expr)                                                                                \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) throw Error(CKM_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// ---- kernel -------------------------------------------------------------------------------------
// set s of bin b: present = #markers with count >= 1, multi = sum(count-1 | count > 1)
__global__ void count_sets_kernel(uint32_t nsets, const uint32_t *__restrict__ marker_off, const int32_t *__restrict__ marker_count,
                                  int32_t *__restrict__ set_present, int32_t *__restrict__ set_multi) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nsets) return;
  int present = 0, multi = 0;
  for (uint32_t i = marker_off[s]; i < marker_off[s + 1]; ++i) {
    const int c = marker_count[i];
    present += (c >= 1);
    multi += (c > 1) ? c - 1 : 0;
  }
  set_present[s] = present;
  set_multi[s] = multi;
}

// one wavefront per bin: histogram of copy numbers over the bin's UNIQUE markers (0,1,2,3,4,5+),
// plus the --individual_markers totals.
__global__ void bin_hist_kernel(uint32_t nbins, const uint32_t *__restrict__ set_off, const uint32_t *__restrict__ marker_off,
                                const int32_t *__restrict__ marker_count, const uint8_t *__restrict__ marker_first,
                                int32_t *__restrict__ hist, int32_t *__restrict__ present_total, int32_t *__restrict__ multi_total) {
  const uint32_t b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= nbins) return;
  const uint32_t m0 = marker_off[set_off[b]], m1 = marker_off[set_off[b + 1]];
  int h[6] = {0, 0, 0, 0, 0, 0}, pres = 0, mult = 0;
  for (uint32_t i = m0 + lane; i < m1; i += 64) {
    if (!marker_first[i]) continue;
    const int c = marker_count[i];
    const int k = c > 5 ? 5 : c;
#pragma unroll
    for (int j = 0; j < 6; ++j) h[j] += (k == j);
    if (c >= 1) { ++pres; mult += c - 1; }
  }
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
#pragma unroll
    for (int j = 0; j < 6; ++j) h[j] += __shfl_xor(h[j], s);
    pres += __shfl_xor(pres, s); mult += __shfl_xor(mult, s);
  }
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < 6; ++j) hist[(size_t)b * 6 + j] = h[j];
    present_total[b] = pres; multi_total[b] = mult;
  }
}


// all device arrays of one call live in ONE grow-only scratch buffer owned by the ctx (no per-call hipMalloc)
}  // namespace ckm
struct ckm_ctx;
void *ckm_ctx_reduce_scratch(ckm_ctx *ctx, size_t bytes);
namespace ckm {
static void run_count_sets(ckm_ctx *ctx, const ckm_marker_sets *ms, const int32_t *marker_count, const uint8_t *marker_first, int32_t *set_present,
                           int32_t *set_multi, int32_t *hist, int32_t *present_total, int32_t *multi_total) {
  const uint32_t nbins = ms->nbins, nsets = ms->set_off[nbins], nmark = ms->marker_off[nsets];
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_setoff = 0, o_moff = o_setoff + al((size_t)(nbins + 1) * 4), o_cnt = o_moff + al((size_t)(nsets + 1) * 4), o_first = o_cnt + al((size_t)nmark * 4),
               o_pres = o_first + al(nmark), o_multi = o_pres + al((size_t)nsets * 4), o_hist = o_multi + al((size_t)nsets * 4), o_pt = o_hist + al((size_t)nbins * 24),
               o_mt = o_pt + al((size_t)nbins * 4), total = o_mt + al((size_t)nbins * 4);
  char *base = static_cast<char *>(::ckm_ctx_reduce_scratch(ctx, total + 256));
  HIPCHK(hipMemcpyAsync(base + o_setoff, ms->set_off, (size_t)(nbins + 1) * 4, hipMemcpyHostToDevice, 0));
  HIPCHK(hipMemcpyAsync(base + o_moff, ms->marker_off, (size_t)(nsets + 1) * 4, hipMemcpyHostToDevice, 0));
  if (nmark) {
    HIPCHK(hipMemcpyAsync(base + o_cnt, marker_count, (size_t)nmark * 4, hipMemcpyHostToDevice, 0));
    HIPCHK(hipMemcpyAsync(base + o_first, marker_first, nmark, hipMemcpyHostToDevice, 0));
  }
  if (nsets) hipLaunchKernelGGL(count_sets_kernel, dim3((nsets + 255) / 256), dim3(256), 0, 0, nsets, (const uint32_t *)(base + o_moff), (const int32_t *)(base + o_cnt),
                                (int32_t *)(base + o_pres), (int32_t *)(base + o_multi));
  if (nbins) hipLaunchKernelGGL(bin_hist_kernel, dim3((nbins + 3) / 4), dim3(256), 0, 0, nbins, (const uint32_t *)(base + o_setoff), (const uint32_t *)(base + o_moff),
                                (const int32_t *)(base + o_cnt), (const uint8_t *)(base + o_first), (int32_t *)(base + o_hist), (int32_t *)(base + o_pt), (int32_t *)(base + o_mt));
  HIPCHK(hipGetLastError());
  if (nsets) { HIPCHK(hipMemcpyAsync(set_present, base + o_pres, (size_t)nsets * 4, hipMemcpyDeviceToHost, 0)); HIPCHK(hipMemcpyAsync(set_multi, base + o_multi, (size_t)nsets * 4, hipMemcpyDeviceToHost, 0)); }
  if (nbins) {
    HIPCHK(hipMemcpyAsync(hist, base + o_hist, (size_t)nbins * 24, hipMemcpyDeviceToHost, 0));
    HIPCHK(hipMemcpyAsync(present_total, base + o_pt, (size_t)nbins * 4, hipMemcpyDeviceToHost, 0));
    HIPCHK(hipMemcpyAsync(multi_total, base + o_mt, (size_t)nbins * 4, hipMemcpyDeviceToHost, 0));
  }
  HIPCHK(hipStreamSynchronize(0));
}

// ---- host filters -----------------------------------------------------------------------------------
struct RHit {
  uint64_t row, row2;
  uint32_t key, model;
  std::string name;
  int tlen, qlen, hmm_from, hmm_to, ali_from, ali_to, env_from, env_to;
  double full_e, i_e, full_sc, dom_sc;
};

// Python's  name[0:name.rfind('_')]  and  int(name[name.rfind('_')+1:])
static std::string scaffold_of(const std::string &n) {
  const size_t p = n.rfind('_');
  if (p == std::string::npos) return n.empty() ? std::string() : n.substr(0, n.size() - 1);   // s[0:-1]
  return n.substr(0, p);
}
static bool orf_number(const std::string &n, long long &out) {
  const size_t p = n.rfind('_');
  std::string t = (p == std::string::npos) ? n : n.substr(p + 1);
  size_t b = 0, e = t.size();
  while (b < e && isspace((unsigned char)t[b])) ++b;
  while (e > b && isspace((unsigned char)t[e - 1])) --e;
  if (b >= e) return false;
  bool neg = false;
  if (t[b] == '+' || t[b] == '-') { neg = t[b] == '-'; ++b; }
  if (b >= e || !isdigit((unsigned char)t[b])) return false;
  long long v = 0; bool prev_us = false;
  for (size_t i = b; i < e; ++i) {
    if (t[i] == '_') { if (prev_us || i + 1 >= e) return false; prev_us = true; continue; }
    if (!isdigit((unsigned char)t[i])) return false;
    prev_us = false; v = v * 10 + (t[i] - '0');
    if (v > (1LL << 60)) return false;
  }
  out = neg ? -v : v;
  return true;
}

// The value a reader of the domtblout TEXT gets back: %6.1f for scores, %9.2g for E-values (checkm/hmmer.py:270-277).
// std::to_chars/from_chars are correctly rounded like printf/strtod and several times faster.
static double text_round_f1(double v) {            // strtod(sprintf("%.1f", v))
  char buf[64];
  auto r = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::fixed, 1);
  double out = 0.0;
  std::from_chars(buf, r.ptr, out);
  return out;
}
static double text_round_g2(double v) {            // strtod(sprintf("%.2g", v)): two significant digits in either notation
  if (!(v == v) || v == 0.0 || std::isinf(v)) return v;
  char buf[64];
  auto r = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::scientific, 1);
  double out = 0.0;
  std::from_chars(buf, r.ptr, out);
  return out;
}

struct OrderedHits {   // dict key -> list, Python insertion order
  std::vector<uint32_t> keys;
  std::unordered_map<uint32_t, std::vector<RHit>> lists;
  std::vector<RHit> &get(uint32_t k) { auto it = lists.find(k); if (it == lists.end()) { keys.push_back(k); return lists[k]; } return it->second; }
  bool has(uint32_t k) const { return lists.count(k) != 0; }
};

static bool vet_hit(const RHit &h, const ckm_model_info *mi, const ckm_reduce_flags *fl, size_t voff) {
  if (!fl->skip_pseudogene_correction) {
    const double alen = (double)(h.ali_to - h.ali_from);
    if (alen / (double)h.qlen < 0.3) return false;
  }
  const int kind = fl->ignore_thresholds ? 0 : mi->thr_kind[voff + h.model];
  if (kind != 0) return mi->thr_full[voff + h.model] <= h.full_sc && mi->thr_dom[voff + h.model] <= h.dom_sc;
  if (h.full_e > fl->evalue_threshold) return false;
  const double alen = (double)(h.ali_to - h.ali_from);
  return alen / (double)h.qlen >= fl->length_threshold;
}

static bool nested_with(const ckm_model_info *mi, uint32_t a, uint32_t b) {
  if (!mi->nest_off) return false;
  for (uint32_t i = mi->nest_off[a]; i < mi->nest_off[a + 1]; ++i) if (mi->nest_idx[i] == b) return true;
  return false;
}

static OrderedHits clan_filter(OrderedHits &in, const ckm_model_info *mi, const std::vector<uint8_t> &key_is_pf) {
  OrderedHits out;
  std::vector<std::string> orf_order; std::unordered_map<std::string, std::vector<RHit>> by_orf;
  for (uint32_t k : in.keys) {
    std::vector<RHit> &hs = in.lists[k];
    if (key_is_pf[k]) {
      for (auto &h : hs) { auto it = by_orf.find(h.name); if (it == by_orf.end()) { orf_order.push_back(h.name); by_orf[h.name].push_back(h); } else it->second.push_back(h); }
    } else out.get(k) = hs;
  }
  for (const std::string &orf : orf_order) {
    std::vector<RHit> &hs = by_orf[orf];
    std::stable_sort(hs.begin(), hs.end(), [](const RHit &a, const RHit &b) { return a.full_e != b.full_e ? a.full_e < b.full_e : a.i_e < b.i_e; });
    std::vector<uint8_t> filtered(hs.size(), 0);
    for (size_t i = 0; i < hs.size(); ++i) {
      if (filtered[i]) continue;
      for (size_t j = i + 1; j < hs.size(); ++j) {
        if (filtered[j]) continue;
        if (mi->clan[hs[i].model] != mi->clan[hs[j].model]) continue;
        const int sI = hs[i].ali_from, eI = hs[i].ali_to, sJ = hs[j].ali_from, eJ = hs[j].ali_to;
        if (!((sI <= sJ && eI > sJ) || (sJ <= sI && eJ > sI))) continue;
        if (nested_with(mi, hs[i].model, hs[j].model)) continue;
        filtered[j] = 1;
      }
    }
    for (size_t i = 0; i < hs.size(); ++i) if (!filtered[i]) out.get(hs[i].key).push_back(hs[i]);
  }
  return out;
}

static void merge_adjacent(std::vector<RHit> &hits) {
  bool combined = true;
  while (combined) {
    if (hits.empty()) break;
    for (size_t i = 0; i < hits.size(); ++i) {
      const std::string orfI = hits[i].name, scafI = scaffold_of(orfI);
      combined = false;
      size_t jm = 0;
      for (size_t j = i + 1; j < hits.size(); ++j) {
        const std::string &orfJ = hits[j].name;
        if (scafI == scaffold_of(orfJ)) {
          long long nI, nJ;
          if (!orf_number(orfI, nI) || !orf_number(orfJ, nJ)) break;
          if (std::llabs(nI - nJ) == 1) { combined = true; jm = j; break; }
        }
      }
      if (combined) {
        RHit nh = hits[i]; const RHit &hj = hits[jm];
        nh.name = (orfI <= hj.name) ? orfI + "&&" + hj.name : hj.name + "&&" + orfI;
        nh.tlen = hits[i].tlen + hj.tlen;
        nh.hmm_from = std::min(hits[i].hmm_from, hj.hmm_from); nh.hmm_to = std::min(hits[i].hmm_to, hj.hmm_to);
        nh.ali_from = std::min(hits[i].ali_from, hj.ali_from); nh.ali_to = std::min(hits[i].ali_to, hj.ali_to);
        nh.env_from = std::min(hits[i].env_from, hj.env_from); nh.env_to = std::min(hits[i].env_to, hj.env_to);
        nh.row2 = hj.row;
        hits.erase(hits.begin() + jm);
        hits.erase(hits.begin() + i);
        hits.push_back(nh);
        break;
      }
    }
  }
}

}  // namespace ckm

using namespace ckm;

struct ckm_qa {
  uint32_t nbins = 0;
  std::vector<int32_t> hist, set_present, set_multi;
  std::vector<double> comp, cont;
  std::vector<uint32_t> set_off;
  std::vector<uint64_t> kept_bin_off, kept_row, kept_row2;
  std::vector<uint32_t> kept_key;
  std::vector<int32_t> kept_tlen, kept_hmm_from, kept_hmm_to, kept_ali_from, kept_ali_to, kept_env_from, kept_env_to;
};

// defined in ckm_api.hip
extern "C" int ckm_hits_columns(const ckm_hits *h, ckm_hit_columns *out);
const std::string &ckm_seq_name(const ckm_seqs *s, uint32_t i);
int ckm_ctx_device(const ckm_ctx *ctx);
void ckm_ctx_parallel_for(ckm_ctx *ctx, size_t n, size_t chunk, const std::function<void(size_t, size_t)> &f);

template <class F>
static int guarded_r(F &&f) {
  try { f(); return CKM_OK; }
  catch (const Error &e) { set_last_error(e.what()); return e.code; }
  catch (const std::bad_alloc &) { set_last_error("out of host memory"); return CKM_ENOMEM; }
  catch (const std::exception &e) { set_last_error(e.what()); return CKM_EINVAL; }
}

extern "C" int ckm_count_sets(ckm_ctx *ctx, const ckm_marker_sets *ms, const int32_t *marker_count, const uint8_t *marker_first,
                              int32_t *set_present, int32_t *set_multi, int32_t *hist, int32_t *present_total, int32_t *multi_total) {
  return guarded_r([&] {
    if (!ctx || !ms || !marker_count || !marker_first || !set_present || !set_multi || !hist || !present_total || !multi_total) throw Error(CKM_EINVAL, "NULL argument");
    HIPCHK(hipSetDevice(ckm_ctx_device(ctx)));
    run_count_sets(ctx, ms, marker_count, marker_first, set_present, set_multi, hist, present_total, multi_total);
  });
}

extern "C" int ckm_reduce(ckm_ctx *ctx, const ckm_hits *h, const ckm_hit_columns *ext, const ckm_seqs *s, const ckm_model_info *mi,
                          const ckm_reduce_flags *fl, const ckm_marker_sets *ms, ckm_qa **out) {
  return guarded_r([&] {
    if (!ctx || !mi || !fl || !ms || !out || (!h && !ext)) throw Error(CKM_EINVAL, "NULL argument");
    *out = nullptr;
    HIPCHK(hipSetDevice(ckm_ctx_device(ctx)));
    ckm_hit_columns cols;
    const bool from_search = (h != nullptr);
    if (from_search) { if (!s) throw Error(CKM_EINVAL, "ckm_seqs required with ckm_hits"); ckm_hits_columns(h, &cols); cols.target_name = nullptr; }
    else { cols = *ext; if (!cols.target_name) throw Error(CKM_EINVAL, "ext.target_name required"); }
    const uint32_t nbins = ms->nbins;
    if (cols.nbins != nbins) throw Error(CKM_EINVAL, "marker sets and hits disagree on the number of bins");
    uint32_t nkeys = 0;
    for (uint32_t m = 0; m < mi->nmodels; ++m) nkeys = std::max(nkeys, mi->key[m] + 1);
    const uint32_t nsets = ms->set_off[nbins], nmark = ms->marker_off[nsets];
    for (uint32_t i = 0; i < nmark; ++i) nkeys = std::max(nkeys, ms->marker_key[i] + 1);
    std::vector<uint8_t> key_is_pf(nkeys, 0);
    for (uint32_t m = 0; m < mi->nmodels; ++m) if (mi->is_pf[m]) key_is_pf[mi->key[m]] = 1;
    std::unique_ptr<ckm_qa> qa(new ckm_qa());
    qa->nbins = nbins;
    qa->kept_bin_off.assign(nbins + 1, 0);
    std::vector<int32_t> marker_count(nmark, 0);
    std::vector<uint8_t> marker_first(nmark, 0);
    // the ordered-container filters of one bin touch nothing of another bin: bins run on the context's host threads,
    // each into its own list of kept hits, concatenated in bin order afterwards
    struct Kept { std::vector<uint32_t> key; std::vector<RHit> hit; };
    std::vector<Kept> kept(nbins);
    ckm_ctx_parallel_for(ctx, nbins, 4, [&](size_t blo, size_t bhi) {
    for (uint32_t b = (uint32_t)blo; b < (uint32_t)bhi; ++b) {
      // unique-marker flags of this bin (getMarkerGenes() is a set)
      {
        std::unordered_map<uint32_t, bool> seen;
        for (uint32_t i = ms->marker_off[ms->set_off[b]]; i < ms->marker_off[ms->set_off[b + 1]]; ++i) { auto r = seen.emplace(ms->marker_key[i], true); marker_first[i] = r.second ? 1 : 0; }
      }
      if (fl->bin_select && !fl->bin_select[b]) continue;
      size_t voff = 0;                    // this bin's slice of the threshold tables (ckm_reduce_flags.bin_variant)
      if (fl->nvariants > 1 && fl->bin_variant) {
        if (fl->bin_variant[b] >= fl->nvariants) throw Error(CKM_EINVAL, "bin_variant out of range");
        voff = (size_t)fl->bin_variant[b] * mi->nmodels;
      }
      OrderedHits mh;
      for (uint64_t r = cols.bin_row_off[b]; r < cols.bin_row_off[b + 1]; ++r) {
        RHit x;
        x.row = r; x.row2 = UINT64_MAX; x.model = cols.model[r];
        if (x.model >= mi->nmodels) throw Error(CKM_EINVAL, "hit refers to a model outside ckm_model_info");
        x.key = mi->key[x.model];
        x.name = from_search ? ckm_seq_name(s, cols.seq[r]) : std::string(cols.target_name[r]);
        x.tlen = cols.tlen[r]; x.qlen = cols.qlen[r];
        x.hmm_from = cols.hmm_from[r]; x.hmm_to = cols.hmm_to[r]; x.ali_from = cols.ali_from[r]; x.ali_to = cols.ali_to[r]; x.env_from = cols.env_from[r]; x.env_to = cols.env_to[r];
        if (from_search) {   // the reference sees these through the domtblout text: %9.2g and %6.1f (checkm/hmmer.py:270-277)
          x.full_e = text_round_g2(cols.full_evalue[r]); x.i_e = text_round_g2(cols.i_evalue[r]);
          x.full_sc = text_round_f1((double)cols.full_score[r]); x.dom_sc = text_round_f1((double)cols.dom_score[r]);
        } else {
          x.full_e = cols.full_evalue[r]; x.i_e = cols.i_evalue[r];
          x.full_sc = cols.full_score_d ? cols.full_score_d[r] : (double)cols.full_score[r];
          x.dom_sc = cols.dom_score_d ? cols.dom_score_d[r] : (double)cols.dom_score[r];
        }
        if (!vet_hit(x, mi, fl, voff)) continue;
        // addHit: one domain per (marker, ORF); a strictly better one replaces and moves to the tail
        if (mh.has(x.key)) {
          std::vector<RHit> &lst = mh.lists[x.key];
          int prev = -1;
          for (size_t i = 0; i < lst.size(); ++i) if (lst[i].name == x.name) { prev = (int)i; break; }
          if (prev < 0) lst.push_back(x);
          else if (lst[prev].dom_sc < x.dom_sc) { lst.push_back(x); lst.erase(lst.begin() + prev); }
        } else mh.get(x.key).push_back(x);
      }
      OrderedHits filt = clan_filter(mh, mi, key_is_pf);
      if (!fl->skip_adj_correction) for (uint32_t k : filt.keys) merge_adjacent(filt.lists[k]);
      for (uint32_t k : filt.keys) for (const RHit &x : filt.lists[k]) { kept[b].key.push_back(k); kept[b].hit.push_back(x); }
      for (uint32_t i = ms->marker_off[ms->set_off[b]]; i < ms->marker_off[ms->set_off[b + 1]]; ++i) {
        auto it = filt.lists.find(ms->marker_key[i]);
        marker_count[i] = (it == filt.lists.end()) ? 0 : (int32_t)it->second.size();
      }
    }
    });
    for (uint32_t b = 0; b < nbins; ++b) {
      qa->kept_bin_off[b] = qa->kept_row.size();
      for (size_t j = 0; j < kept[b].hit.size(); ++j) {
        const RHit &x = kept[b].hit[j];
        qa->kept_key.push_back(kept[b].key[j]); qa->kept_row.push_back(x.row); qa->kept_row2.push_back(x.row2);
        qa->kept_tlen.push_back(x.tlen); qa->kept_hmm_from.push_back(x.hmm_from); qa->kept_hmm_to.push_back(x.hmm_to);
        qa->kept_ali_from.push_back(x.ali_from); qa->kept_ali_to.push_back(x.ali_to); qa->kept_env_from.push_back(x.env_from); qa->kept_env_to.push_back(x.env_to);
      }
    }
    qa->kept_bin_off[nbins] = qa->kept_row.size();
    // ---- the data-parallel part, on the device ----
    qa->set_off.assign(ms->set_off, ms->set_off + nbins + 1);
    qa->set_present.assign(nsets, 0); qa->set_multi.assign(nsets, 0); qa->hist.assign((size_t)nbins * 6, 0);
    std::vector<int32_t> ptot(nbins, 0), mtot(nbins, 0);
    run_count_sets(ctx, ms, marker_count.data(), marker_first.data(), qa->set_present.data(), qa->set_multi.data(), qa->hist.data(), ptot.data(), mtot.data());
    // ---- float64 division in the reference's accumulation order (markerSets.py:219-236) ----
    qa->comp.assign(nbins, 0.0); qa->cont.assign(nbins, 0.0);
    for (uint32_t b = 0; b < nbins; ++b) {
      const uint32_t s0 = ms->set_off[b], s1 = ms->set_off[b + 1];
      if (s0 == s1) continue;
      if (fl->individual_markers) {
        // numMarkers() counts set entries (with multiplicity), getMarkerGenes() is unique: markerSets.py:171-187,212-217
        const int nmarkers = (int)(ms->marker_off[s1] - ms->marker_off[s0]);
        qa->comp[b] = 100 * (double)ptot[b] / nmarkers; qa->cont[b] = 100 * (double)mtot[b] / nmarkers;
      } else {
        double comp = 0.0, cont = 0.0;
        for (uint32_t st = s0; st < s1; ++st) {
          const int len = (int)(ms->marker_off[st + 1] - ms->marker_off[st]);
          comp += (double)qa->set_present[st] / len; cont += (double)qa->set_multi[st] / len;
        }
        qa->comp[b] = 100 * comp / (double)(s1 - s0); qa->cont[b] = 100 * cont / (double)(s1 - s0);
      }
    }
    *out = qa.release();
  });
}

extern "C" int ckm_qa_columns_get(const ckm_qa *q, ckm_qa_columns *o) {
  if (!q || !o) { set_last_error("NULL argument"); return CKM_EINVAL; }
  o->nbins = q->nbins; o->hist = q->hist.data(); o->completeness = q->comp.data(); o->contamination = q->cont.data();
  o->set_off = q->set_off.data(); o->set_present = q->set_present.data(); o->set_multi = q->set_multi.data();
  o->nkept = q->kept_row.size(); o->kept_bin_off = q->kept_bin_off.data(); o->kept_key = q->kept_key.data(); o->kept_row = q->kept_row.data();
  o->kept_row2 = q->kept_row2.data(); o->kept_tlen = q->kept_tlen.data(); o->kept_hmm_from = q->kept_hmm_from.data(); o->kept_hmm_to = q->kept_hmm_to.data();
  o->kept_ali_from = q->kept_ali_from.data(); o->kept_ali_to = q->kept_ali_to.data(); o->kept_env_from = q->kept_env_from.data(); o->kept_env_to = q->kept_env_to.data();
  return CKM_OK;
}

extern "C" void ckm_qa_free(ckm_qa *q) { delete q; }
