// gene_dev.h -- per-thread bodies of the gene-calling kernels, written once and compiled twice: by hipcc into the kernels
// of libcheckm_hip.so (kernels_genes.hip, ckm_genes.hip), and by g++ into tests/emu/libgene_emu.so, a HOST EMULATION of the same pipeline
// that the CPU test suite (-m "not gpu") diffs against the gene oracle -- test infrastructure, never loaded by checkm_amd.
// The arithmetic follows oracle/gene_full.c (the restatement of Prodigal 2.6.3's single-genome mode as CheckM invokes it,
// checkm/prodigal.py:80-93; parity unpinned) operation by operation: doubles, the same order of additions, no contraction.
#pragma once
#include <cstdint>
#include <cstdlib>

#ifdef CKM_GENE_EMU
#define GFN inline
#define GLAM
#else
#include <hip/hip_runtime.h>
#define GFN __host__ __device__ inline
#define GLAM __host__ __device__
#endif

namespace ckm {
namespace gene {

constexpr int G_STOP = 3, G_PAD = 255, MAX_SAM_OVLP = 60, MAX_NODE_DIST = 500, OPER_DIST = 60, MASK_SIZE = 50, GC_WINDOW = 120;
constexpr double EDGE_BONUS = 0.74, EDGE_UPS = -1.00;
constexpr uint8_t CODE_PAD = 8;          // between sequences: reads as a known A, counts as neither G/C nor unknown

GFN double dmaxd(double a, double b) { return a > b ? a : b; }
GFN double dmind(double a, double b) { return a < b ? a : b; }
GFN int max_fr(int n1, int n2, int n3) { if (n1 > n2) return n1 > n3 ? 0 : 2; return n2 > n3 ? 1 : 2; }
GFN int popc64(unsigned long long x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __popcll(x);
#else
  return __builtin_popcountll(x);
#endif
}
GFN unsigned long long below64(int b) { return b <= 0 ? 0ull : (b >= 64 ? ~0ull : (~0ull >> (64 - b))); }      // bits [0, b)

// a sequence in the one-byte code (bits 0-1 base, an unknown base reads C; bit 2 unknown); strand-relative accessors
struct GSeq {
  const uint8_t *c; int slen;
  GFN int fwd(int p) const { return c[p] & 3; }
  GFN int at(int strand, int p) const { return strand == 1 ? (c[p] & 3) : 3 - (c[slen - 1 - p] & 3); }
  GFN int unk(int strand, int p) const { return ((strand == 1 ? c[p] : c[slen - 1 - p]) >> 2) & 1; }
  GFN bool is_stop(int strand, int i, int tt) const {
    if (i < 0 || i + 2 >= slen) return false;
    if (unk(strand, i) || unk(strand, i + 1) || unk(strand, i + 2)) return false;
    if (at(strand, i) != 3) return false;
    const int b1 = at(strand, i + 1), b2 = at(strand, i + 2);
    if (b1 == 0 && (b2 == 0 || b2 == 2)) return true;
    if (b1 == 2 && b2 == 0) return tt != 4;
    return false;
  }
  GFN int mer(int strand, int len, int pos) const { int ndx = 0; for (int i = 0; i < len; ++i) ndx |= at(strand, pos + i) << (2 * i); return ndx; }
};

// ---- bit planes over the padded text buffer (one bit per base, 64 per word) and their prefix counts ----
// rank(P, W, pos) = number of set bits of plane W at positions < pos, with P the exclusive prefix of popcounts per word
GFN uint32_t plane_rank(const uint32_t *P, const unsigned long long *W, uint64_t pos) {
  return P[pos >> 6] + (uint32_t)popc64(W[pos >> 6] & below64((int)(pos & 63)));
}

// ---- nodes: structure of arrays, in the gene finder's working order (position, forward strand first) sequence by sequence; every
// bin's range starts at a multiple of 256 (padding entries have type 255) ----
struct Nodes {
  uint32_t *bin, *seq; int32_t *ndx, *sv; int8_t *strand; uint8_t *type, *edge;
  uint32_t *chx;                     // position in the chain array
  int32_t *ctr;                      // [n][3] GC-frame codon counts of the node's ORF (training)
  uint8_t *gcb_cls; double *gcb_term;
  double *cscore; uint8_t *rbs0, *rbs1;
  uint32_t *dp_min; int32_t *star;   // star[n][3]: relative node index or -1
  double *gcb, *csc, *rscore, *uscore, *tscore, *sscore, *gc_cont, *mot_score;
  uint32_t *mot;                     // packed: ndx (12 bits) | len << 12 (3 bits) | spacendx << 15 (2 bits) | spacer << 17 (5 bits)
  unsigned long long *upw;           // the 18 bases upstream of a start node (positions start-21 .. start-4), first base lowest, 2 bits each
  double *score; int32_t *traceb, *tracef, *ov_mark; uint8_t *elim;
};
GFN uint32_t mot_pack(int ndx, int len, int spacendx, int spacer) { return (uint32_t)ndx | ((uint32_t)len << 12) | ((uint32_t)spacendx << 15) | ((uint32_t)spacer << 17); }
GFN int mot_ndx(uint32_t m) { return (int)(m & 4095u); }
GFN int mot_len(uint32_t m) { return (int)((m >> 12) & 7u); }
GFN int mot_spacendx(uint32_t m) { return (int)((m >> 15) & 3u); }
GFN int mot_spacer(uint32_t m) { return (int)((m >> 17) & 31u); }

// per-bin training tables on the device
struct TrainDev {
  const double *st_wt;               // [nbins]
  const double *bias;                // [nbins][3]
  const double *type_wt;             // [nbins][3]
  const double *rbs_wt;              // [nbins][28]
  const double *ups_comp;            // [nbins][32][4]
  const double *mot_wt;              // [nbins][4][4][4096]  (only bins without Shine-Dalgarno usage hold values)
  const double *no_mot;              // [nbins]
  const uint8_t *uses_sd;            // [nbins]
  const double *lfac_raw;            // [nbins][1001]: log((1 - p^g) / p^g), g = 0..1000 codons (entry 0 unused), p = P(no stop)
  const double *l80;                 // [nbins]
  const double *gene_dc;             // [nbins][4096]
};

// view of ONE sequence's nodes (relative indices) for the ordered walks
struct NView {
  Nodes n; uint32_t lo; int nn;
  GFN int ndx(int i) const { return n.ndx[lo + i]; }
  GFN int sv(int i) const { return n.sv[lo + i]; }
  GFN int strand(int i) const { return n.strand[lo + i]; }
  GFN int type(int i) const { return n.type[lo + i]; }
  GFN bool stop(int i) const { return n.type[lo + i] == G_STOP; }
  GFN int edge(int i) const { return n.edge[lo + i]; }
};

GFN double igm_core(int s1, int x1, double r1, double u1, int s2, int x2, double r2, double u2, double st_wt) {
  double rval = 0.0; int ovlp = 0;
  if ((s1 == 1 && s2 == 1 && (x1 + 2 == x2 || x1 - 1 == x2)) || (s1 == -1 && s2 == -1 && (x1 + 2 == x2 || x1 - 1 == x2))) {
    if (s1 == 1 && r2 < 0) rval -= r2;
    if (s1 == -1 && r1 < 0) rval -= r1;
    if (s1 == 1 && u2 < 0) rval -= u2;
    if (s1 == -1 && u1 < 0) rval -= u1;
  }
  const int dist = abs(x1 - x2);
  if (s1 == 1 && s2 == 1 && x1 + 2 >= x2) ovlp = 1;
  else if (s1 == -1 && s2 == -1 && x1 >= x2 + 2) ovlp = 1;
  if (dist > 3 * OPER_DIST || s1 != s2) rval -= 0.15 * st_wt;
  else if ((dist <= OPER_DIST && ovlp == 0) || dist < 0.25 * OPER_DIST) rval += (2.0 - (double)dist / OPER_DIST) * 0.15 * st_wt;
  return rval;
}
// intergenic_mod of two nodes of a view (relative indices)
GFN double igm_nodes(const NView &V, int a, int b, double st_wt) {
  return igm_core(V.strand(a), V.ndx(a), V.n.rscore[V.lo + a], V.n.uscore[V.lo + a], V.strand(b), V.ndx(b), V.n.rscore[V.lo + b], V.n.uscore[V.lo + b], st_wt);
}

// ---- the dynamic program's connection score (node.c: score_connection), over a source S of node data ----
struct DpNode { int ndx, sv, strand, stop; };

template <class S>
GFN double dp_igm(const S &src, double st_wt, int k1, const DpNode &n1, int k2, const DpNode &n2) {
  double rval = 0.0; int ovlp = 0;
  if ((n1.strand == 1 && n2.strand == 1 && (n1.ndx + 2 == n2.ndx || n1.ndx - 1 == n2.ndx)) ||
      (n1.strand == -1 && n2.strand == -1 && (n1.ndx + 2 == n2.ndx || n1.ndx - 1 == n2.ndx))) {
    if (n1.strand == 1 && src.rscore(k2) < 0) rval -= src.rscore(k2);
    if (n1.strand == -1 && src.rscore(k1) < 0) rval -= src.rscore(k1);
    if (n1.strand == 1 && src.uscore(k2) < 0) rval -= src.uscore(k2);
    if (n1.strand == -1 && src.uscore(k1) < 0) rval -= src.uscore(k1);
  }
  const int dist = abs(n1.ndx - n2.ndx);
  if (n1.strand == 1 && n2.strand == 1 && n1.ndx + 2 >= n2.ndx) ovlp = 1;
  else if (n1.strand == -1 && n2.strand == -1 && n1.ndx >= n2.ndx + 2) ovlp = 1;
  if (dist > 3 * 60 || n1.strand != n2.strand) rval -= 0.15 * st_wt;
  else if ((dist <= 60 && ovlp == 0) || dist < 0.25 * 60) rval += (2.0 - (double)dist / 60) * 0.15 * st_wt;
  return rval;
}
// score of the connection p1 -> p2 (node indices relative to the sequence's first node); false: no such connection.
// The source serves p1 and p2 through node / val / score / tb (a device source may assume both are in its LDS ring) and everything reached
// THROUGH them -- an overlapping start p3, the node a trace-back points to -- through node3 / val3 / ndx_any (anywhere in the sequence).
// KNOWN: the caller enumerates p1 by class (strand KS1, stop KST1), so the twelve cases fold to the ones that class can take -- the
// same statements either way.
// dp_connection_s is the connection WITHOUT the predecessor's own score (score_connection's `score`, the term added to n1->score): of
// the dynamic program's state it reads only tb(p1) -- as "has p1 a predecessor at all" for a forward stop or a reverse start, and as the
// position of that predecessor in the two cases dp_pair_dynamic names.  Everything else is fixed before the sweep starts, which is what
// lets the sweep take its nodes 64 at a time (kernels_genes.hip: gene_dp_kernel).
template <class S, bool KNOWN, int KS1, bool KST1>
GFN bool dp_connection_s(const S &src, double st_wt, int p1, int p2, const DpNode &n2, double &score_out, int &mark) {
  const int flag = src.flag;
  DpNode n1 = src.node(p1);
  if (KNOWN) { n1.strand = KS1; n1.stop = KST1; }
  int left = n1.ndx, right = n2.ndx, ovlp = 0, maxfr = -1;
  double score = 0.0, scr_mod = 0.0;
  const int s1 = n1.strand, s2 = n2.strand; const bool st1 = n1.stop, st2 = n2.stop;
  if (!st1 && !st2 && s1 == s2) return false;
  else if (s1 == 1 && !st1 && s2 == -1) return false;
  else if (s1 == -1 && st1 && s2 == 1) return false;
  else if (s1 == -1 && !st1 && s2 == 1 && st2) return false;
  const int tb1 = src.tb(p1);
  if (tb1 == -1 && s1 == 1 && st1) return false;
  if (tb1 == -1 && s1 == -1 && !st1) return false;
  if (s1 == s2 && s1 == 1 && !st1 && st2) {
    if (n2.sv >= n1.ndx) return false;
    if (n1.ndx % 3 != n2.ndx % 3) return false;
    right += 2;
    if (flag == 0) scr_mod = src.val(p1); else score = src.val(p1);
  } else if (s1 == s2 && s1 == -1 && st1 && !st2) {
    if (n1.sv <= n2.ndx) return false;
    if (n1.ndx % 3 != n2.ndx % 3) return false;
    left -= 2;
    if (flag == 0) scr_mod = src.val(p2); else score = src.val(p2);
  } else if (s1 == 1 && st1 && s2 == 1 && !st2) {
    left += 2;
    if (left >= right) return false;
    if (flag == 1) score = dp_igm(src, st_wt, p1, n1, p2, n2);
  } else if (s1 == 1 && st1 && s2 == -1 && st2) {
    left += 2; right -= 2;
    if (left >= right) return false;
    double maxval = 0.0; int best_ov = 0;
    for (int i = 0; i < 3; ++i) {
      const int p3 = src.star(p2, i);
      if (p3 == -1) continue;
      const DpNode n3 = src.node3(p3);
      const int ov = left - n3.sv + 1;
      if (ov <= 0 || ov >= 200) continue;
      if (ov >= n3.ndx - left) continue;
      if (tb1 == -1) continue;
      if (ov >= n3.sv - src.ndx_any(tb1) - 2) continue;
      const double v = flag == 1 ? src.val3(p3) + dp_igm(src, st_wt, p3, n3, p2, n2) : src.val3(p3);
      if (v > maxval) { maxfr = i; maxval = v; best_ov = ov; }
    }
    if (maxfr != -1) { ovlp = best_ov; if (flag == 0) scr_mod = maxval; else score = maxval; }
    else if (flag == 1) score = dp_igm(src, st_wt, p1, n1, p2, n2);
  } else if (s1 == -1 && !st1 && s2 == -1 && st2) {
    right -= 2;
    if (left >= right) return false;
    if (flag == 1) score = dp_igm(src, st_wt, p1, n1, p2, n2);
  } else if (s1 == -1 && !st1 && s2 == 1 && !st2) {
    if (left >= right) return false;
    if (flag == 1) score = dp_igm(src, st_wt, p1, n1, p2, n2);
  } else if (s1 == 1 && st1 && s2 == -1 && !st2) {
    if (n2.sv - 2 >= n1.ndx + 2) return false;
    ovlp = (n1.ndx + 2) - (n2.sv - 2) + 1;
    if (ovlp >= 200) return false;
    if ((n1.ndx + 2 - n2.sv - 2 + 1) >= (n2.ndx - n1.ndx + 3 + 1)) return false;
    const int bnd = tb1 == -1 ? 0 : src.ndx_any(tb1);
    if ((n1.ndx + 2 - n2.sv - 2 + 1) >= (n2.sv - 3 - bnd + 1)) return false;
    left = n2.sv - 2;
    if (flag == 0) scr_mod = src.val(p2); else score = src.val(p2) - 0.15 * st_wt;
  } else if (s1 == s2 && s1 == 1 && st1 && st2) {
    if (n2.sv >= n1.ndx) return false;
    const int p3 = src.star(p1, n2.ndx % 3);
    if (p3 == -1) return false;
    const DpNode n3 = src.node3(p3);
    left = n3.ndx; right += 2;
    if (flag == 0) scr_mod = src.val3(p3); else score = src.val3(p3) + dp_igm(src, st_wt, p1, n1, p3, n3);
  } else if (s1 == s2 && s1 == -1 && st1 && st2) {
    if (n1.sv <= n2.ndx) return false;
    const int p3 = src.star(p2, n1.ndx % 3);
    if (p3 == -1) return false;
    const DpNode n3 = src.node3(p3);
    left -= 2; right = n3.ndx;
    if (flag == 0) scr_mod = src.val3(p3); else score = src.val3(p3) + dp_igm(src, st_wt, p3, n3, p2, n2);
  }
  if (flag == 0) score = ((double)(right - left + 1 - (ovlp * 2))) * scr_mod;
  score_out = score;
  mark = maxfr;
  return true;
}
template <class S, bool KNOWN, int KS1, bool KST1>
GFN bool dp_connection_x(const S &src, double st_wt, int p1, int p2, const DpNode &n2, double &total, int &mark) {
  double score;
  if (!dp_connection_s<S, KNOWN, KS1, KST1>(src, st_wt, p1, p2, n2, score, mark)) return false;
  total = src.score(p1) + score;
  return true;
}
template <class S>
GFN bool dp_connection(const S &src, double st_wt, int p1, int p2, const DpNode &n2, double &total, int &mark) {
  return dp_connection_x<S, false, 0, false>(src, st_wt, p1, p2, n2, total, mark);
}
// node classes of the dynamic program: 0 forward start, 1 forward stop, 2 reverse start, 3 reverse stop
GFN int dp_class(int strand, bool stop) { return (strand == -1 ? 2 : 0) + (stop ? 1 : 0); }
// can a node of class c1 precede one of class c2 at all (the four early exits of score_connection)?
GFN bool dp_pair_possible(int c1, int c2) {
  const unsigned ok = (1u << (1 * 4 + 0)) | (1u << (2 * 4 + 0)) |                    // -> forward start: forward stop, reverse start
                      (1u << (0 * 4 + 1)) | (1u << (1 * 4 + 1)) |                    // -> forward stop: forward start, forward stop
                      (1u << (1 * 4 + 2)) | (1u << (3 * 4 + 2)) |                    // -> reverse start: forward stop, reverse stop
                      (1u << (1 * 4 + 3)) | (1u << (2 * 4 + 3)) | (1u << (3 * 4 + 3));   // -> reverse stop: forward stop, reverse start, reverse stop
  return (ok >> (c1 * 4 + c2)) & 1u;
}
// the class pairs whose connection reads the POSITION of p1's predecessor (forward stop -> reverse start: the bound on the overlap;
// forward stop -> reverse stop: which overlapping starts qualify)
GFN bool dp_pair_dynamic(int c1, int c2) { return c1 == 1 && c2 >= 2; }
// does a class need a predecessor of its own to be one (score_connection's "edge artifacts")?
GFN bool dp_class_needs_tb(int c1) { return c1 == 1 || c1 == 2; }
// A forward stop or a reverse start i connects only to nodes at positions > sv(i) - 4 (its own open reading frame's starts and the stops
// inside it; a reverse gene's own stop and the forward stops that overlap it): the tests of score_connection that say so are
// `n2.sv >= n1.ndx`, `n2.sv - 2 >= n1.ndx + 2` and, for reverse stop -> reverse start in one frame, `n1.sv <= n2.ndx` (then sv(i) is
// that stop).  Nodes are in position order, so the candidates of such a node begin at the first node with ndx >= dp_pos_floor.
GFN bool dp_class_pos_bounded(int c2) { return c2 == 1 || c2 == 2; }
GFN int dp_pos_floor(int sv) { return sv - 3; }
template <class S>
GFN bool dp_connection_s_class(int c1, const S &src, double st_wt, int p1, int p2, const DpNode &n2, double &score, int &mark) {
  switch (c1) {
    case 0: return dp_connection_s<S, true, 1, false>(src, st_wt, p1, p2, n2, score, mark);
    case 1: return dp_connection_s<S, true, 1, true>(src, st_wt, p1, p2, n2, score, mark);
    case 2: return dp_connection_s<S, true, -1, false>(src, st_wt, p1, p2, n2, score, mark);
    default: return dp_connection_s<S, true, -1, true>(src, st_wt, p1, p2, n2, score, mark);
  }
}
template <class S>
GFN bool dp_connection_class(int c1, const S &src, double st_wt, int p1, int p2, const DpNode &n2, double &total, int &mark) {
  switch (c1) {
    case 0: return dp_connection_x<S, true, 1, false>(src, st_wt, p1, p2, n2, total, mark);
    case 1: return dp_connection_x<S, true, 1, true>(src, st_wt, p1, p2, n2, total, mark);
    case 2: return dp_connection_x<S, true, -1, false>(src, st_wt, p1, p2, n2, total, mark);
    default: return dp_connection_x<S, true, -1, true>(src, st_wt, p1, p2, n2, total, mark);
  }
}
// the reference keeps the LAST candidate (largest index) that reaches the running maximum; candidates may arrive in any order here
GFN void dp_take(double tot, int j, int mark, double &best, int &bj, int &bmark) {
  if (tot >= 0.0 && (bj < 0 || tot > best || (tot == best && j > bj))) { best = tot; bj = j; bmark = mark; }
}

// ---- per-node bodies ----
// hexamer log-odds of an ORF summed codon by codon FROM THE STOP towards the start (node.c: raw_coding_score, first pass)
GFN double node_cscore(const uint8_t *txt, int slen, int strand, int ps, int pe, const double *dc) {
  const GSeq s{txt, slen};
  double score = 0.0;
  int hi = 0;
  if (pe + 2 < slen && pe >= 0) hi = s.at(strand, pe) | (s.at(strand, pe + 1) << 2) | (s.at(strand, pe + 2) << 4);
  for (int j = pe - 3; j >= ps; j -= 3) {
    const int lo = s.at(strand, j) | (s.at(strand, j + 1) << 2) | (s.at(strand, j + 2) << 4);
    score += dc[lo | (hi << 6)];
    hi = lo;
  }
  return score;
}

GFN int sd_bin_exact(double c, int f) {
  if (c < 6.0) return 0;
  if (c == 6.0) return f == 2 ? 1 : f == 3 ? 2 : f == 1 ? 6 : 13;
  if (c == 8.0) return f == 3 ? 3 : f == 2 ? 11 : f == 1 ? 12 : 15;
  if (c == 9.0) return f == 3 ? 3 : f == 2 ? 11 : f == 1 ? 12 : 16;
  if (c == 11.0) return f == 3 ? 10 : f == 2 ? 20 : f == 1 ? 21 : 22;
  if (c == 12.0) return f == 3 ? 10 : f == 2 ? 20 : f == 1 ? 23 : 24;
  if (c == 14.0) return f == 3 ? 10 : f == 2 ? 25 : f == 1 ? 26 : 27;
  return 0;
}
GFN int sd_bin_mm(double c, int f) {
  if (c < 6.0) return 0;
  if (c == 6.0) return f == 3 ? 2 : f == 2 ? 4 : f == 1 ? 5 : 9;
  if (c == 7.0) return f == 3 ? 2 : f == 2 ? 7 : f == 1 ? 8 : 14;
  if (c == 9.0) return f == 3 ? 3 : f == 2 ? 17 : f == 1 ? 18 : 19;
  return 0;
}
GFN int shine_dalgarno(const GSeq &s, int strand, int pos, int start, const double *rwt, int mm) {
  double match[6];
  int max_val = 0;
  const int lim0 = start - 4 - pos, limit = lim0 < 6 ? lim0 : 6;
  for (int i = 0; i < 6; ++i) match[i] = -10.0;
  for (int i = 0; i < limit; ++i) {
    if (pos + i < 0) continue;
    const int u = s.unk(strand, pos + i), q = s.at(strand, pos + i);
    const bool a = !u && q == 0, g = !u && q == 2;
    if (i % 3 == 0) match[i] = a ? 2.0 : (mm ? -3.0 : -10.0);
    else match[i] = g ? 3.0 : (mm ? -2.0 : -10.0);
  }
  for (int i = limit; i >= (mm ? 5 : 3); --i) {
    for (int j = 0; j <= limit - i; ++j) {
      double cur = -2.0; int mism = 0;
      for (int k = j; k < j + i; ++k) {
        cur += match[k];
        if (match[k] < 0.0) mism++;
        if (mm && match[k] < 0.0 && (k <= j + 1 || k >= j + i - 2)) cur -= 10.0;
      }
      if (mm ? mism != 1 : mism > 0) continue;
      const int rdis = start - (pos + j + i);
      int f;
      if (!mm) {
        if (rdis < 5 && i < 5) f = 2;
        else if (rdis < 5 && i >= 5) f = 1;
        else if (rdis > 10 && rdis <= 12 && i < 5) f = 1;
        else if (rdis > 10 && rdis <= 12 && i >= 5) f = 2;
        else if (rdis >= 13) f = 3;
        else f = 0;
      } else {
        if (rdis < 5) f = 1;
        else if (rdis > 10 && rdis <= 12) f = 2;
        else if (rdis >= 13) f = 3;
        else f = 0;
      }
      if (rdis > 15 || cur < 6.0) continue;
      const int cv = mm ? sd_bin_mm(cur, f) : sd_bin_exact(cur, f);
      if (rwt[cv] < rwt[max_val]) continue;
      if (rwt[cv] == rwt[max_val] && cv < max_val) continue;
      max_val = cv;
    }
  }
  return max_val;
}
GFN void node_rbs(const GSeq &s, int strand, int start, const double *rwt, int &r0, int &r1) {
  r0 = 0; r1 = 0;
  for (int j = start - 20; j <= start - 6; ++j) {
    if (j < 0) continue;
    const int c0 = shine_dalgarno(s, strand, j, start, rwt, 0), c1 = shine_dalgarno(s, strand, j, start, rwt, 1);
    if (c0 > r0) r0 = c0;
    if (c1 > r1) r1 = c1;
  }
}

GFN int best_rbs(int rb0, int rb1, const double *rbs_wt) {
  if (rbs_wt[rb0] > rbs_wt[rb1] + 1.0 || rb1 == 0) return rb0;
  if (rbs_wt[rb0] < rbs_wt[rb1] - 1.0 || rb0 == 0) return rb1;
  return rb0 > rb1 ? rb0 : rb1;
}

// the 18 bases upstream of a start (strand position start): positions start-21 .. start-4, first base lowest; positions before the
// sequence start hold zeros (the words that would read them are skipped by their j < 0 test)
GFN unsigned long long upstream_window(const GSeq &s, int strand, int start) {
  unsigned long long w = 0;
  for (int k = 0; k < 18; ++k) { const int p = start - 21 + k; if (p >= 0) w |= (unsigned long long)s.at(strand, p) << (2 * k); }
  return w;
}
GFN int upw_mer(unsigned long long w, int start, int len, int j) { return (int)((w >> (2 * (j - (start - 21)))) & ((1ull << (2 * len)) - 1)); }
GFN int spacer_ndx(int j, int start, int i) { if (j <= start - 16 - i) return 3; if (j <= start - 14 - i) return 2; if (j >= start - 7 - i) return 1; return 0; }

// the words the first training round counts for a start (update_motif_counts, stage 0): f(length - 3, word) for lengths 6 .. 3 at the
// positions of the upstream window
template <class F> GFN void motif_words_stage0(unsigned long long upw, int start, F f) {
  for (int i = 3; i >= 0; --i) for (int j = start - 18 - i; j <= start - 6 - i; ++j) { if (j < 0) continue; f(i, upw_mer(upw, start, i + 3, j)); }
}

// the words the later rounds count for a start whose current motif is m (update_motif_counts, stages 1 and 2): f(length - 3, spacer class, word)
template <class F> GFN void motif_words_stage12(uint32_t m, unsigned long long upw, int start, int stage, F f) {
  const int ml = mot_len(m);
  if (ml == 0) return;
  f(ml - 3, mot_spacendx(m), mot_ndx(m));
  if (stage != 1) return;
  const int sp = mot_spacer(m);
  for (int i = 0; i < ml - 3; ++i) for (int j = start - sp - ml; j <= start - sp - (i + 3); ++j) { if (j < 0) continue; f(i, spacer_ndx(j, start, i), upw_mer(upw, start, i + 3, j)); }
}

// find_best_upstream_motif: mot_wt = the bin's [4][4][4096] table.  Returns the packed motif and its score.
GFN uint32_t best_upstream_motif(const double *mot_wt, double no_mot, unsigned long long upw, int start, int stage, double &mot_score) {
  int max_spacer = 0, max_spacendx = 0, max_len = 0, max_ndx = 0; double max_sc = -100.0;
  // The thirteen words of a motif length are read TOGETHER, then compared in the reference's order: thirteen loads in flight instead of
  // thirteen round trips to the 512 KB table, one behind the other (round 5's loop: 4.6 us per read with other calls' kernels loading the
  // memory system -- this lambda was 28 % of a call's wavefront-cycles, profiles/r06e).
  for (int i = 3; i >= 0; --i) {
    const int j0 = start - 18 - i;
    double sc[13]; int idx[13];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int u = 0; u < 13; ++u) {
      const int j = j0 + u;
      idx[u] = j < 0 ? 0 : upw_mer(upw, start, i + 3, j);
      sc[u] = j < 0 ? 0.0 : mot_wt[((size_t)i * 4 + spacer_ndx(j, start, i)) * 4096 + idx[u]];
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int u = 0; u < 13; ++u) {
      const int j = j0 + u;
      if (j < 0) continue;
      if (sc[u] > max_sc) { max_sc = sc[u]; max_spacendx = spacer_ndx(j, start, i); max_spacer = start - j - i - 3; max_ndx = idx[u]; max_len = i + 3; }
    }
  }
  if (stage == 2 && (max_sc == -4.0 || max_sc < no_mot + 0.69)) { mot_score = no_mot; return mot_pack(0, 0, 0, 0); }
  mot_score = max_sc;
  return mot_pack(max_ndx, max_len, max_spacendx, max_spacer);
}

GFN char amino(const GSeq &s, int strand, int i, int tt) {
  const char *code = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF";
  if (s.unk(strand, i) || s.unk(strand, i + 1) || s.unk(strand, i + 2)) return 'X';
  const int c = s.at(strand, i) * 16 + s.at(strand, i + 1) * 4 + s.at(strand, i + 2);
  if (tt == 4 && c == 3 * 16 + 2 * 4 + 0) return 'W';
  return code[c];
}

}  // namespace gene
}  // namespace ckm
