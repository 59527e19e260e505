// ckm_genes.hip -- C ABI of gene calling on the device: nucleotide contigs of MANY bins in, genes and their proteins out.
// What it replaces: the two `prodigal -p single -m -f gff -g <11|4> -a genes.faa` runs per bin of checkm/prodigal.py:80-133 (`-p meta`,
// which CheckM uses below 100 kb, is NOT built: such a bin comes back untrained and the caller decides).
//
// Division of the work (one call = one translation table for a batch of bins):
//   device   start / stop nodes of the bins' training sequences and of their contigs (kernels_orf.hip: one flag pass serves both);
//            hexamer coding sums and Shine-Dalgarno bins of every start node (kernels_genes.hip); both dynamic programs -- the training
//            pass over a whole bin and the final pass per contig -- one workgroup per sequence, all sequences of the batch side by side.
//   host     (the context's thread pool, a bin or a contig per task) everything that is a single ordered sweep over a bin's nodes or
//            bases: node order and -m masks, GC-frame plot and bias, overlapping-start tables, path untangling, hexamer statistics, the
//            start-site training iterations, node scores, gene records, start tweaks, translation.
// The arithmetic follows oracle/gene_full.c (the restatement of Prodigal 2.6.3's single-genome mode; parity unpinned) operation by
// operation: doubles, the same order of additions; tests/test_gpu_genes.py compares gene for gene and score for score.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>
#include "ckm_host.h"
#include "gene_types.h"

namespace ckm {
void launch_orf_flags(hipStream_t stream, const uint8_t *text, uint8_t *flags, uint64_t n);
void launch_orf_chain(hipStream_t stream, const uint8_t *planes, uint64_t nwin, const uint64_t *contig_off, const int32_t *contig_len, uint32_t ncontigs, int tt4, int closed,
                      void *nodes, unsigned long long *nnodes, unsigned long long cap);
void launch_gene_cscore(hipStream_t st, const GeneSeqDev &seqs, const GeneNodesDev &nd, const double *gene_dc, uint32_t nnodes);
void launch_gene_rbs(hipStream_t st, const GeneSeqDev &seqs, const GeneNodesDev &nd, const double *rbs_wt, uint32_t nnodes);
void launch_gene_dp(hipStream_t st, const GeneNodesDev &nd, const uint32_t *seq_first, const double *st_wt_of_seq, uint32_t nseq, int flag);
struct OrfNodeG { uint32_t contig; int32_t ndx, stop_val; uint8_t type, strand_rev, edge, pad; };

namespace {
constexpr int G_STOP = 3, MAX_SAM_OVLP = 60, MAX_NODE_DIST = 500, OPER_DIST = 60, MASK_SIZE = 50, GC_WINDOW = 120;
constexpr double EDGE_BONUS = 0.74, EDGE_UPS = -1.00;

// (field order: what the training iterations sweep over 30-50 times per bin -- kind, position, coding score, SD bins, then the motif --
//  sits in the first 88 bytes; with the declaration order of the reference's struct a sweep touched two to three cache lines per node and
//  the start-site training of a 192-bin batch was memory-bound: more threads made it slower)
struct GNode {
  int32_t type = 0, edge = 0, ndx = 0, strand = 1, stop_val = 0;
  int32_t rbs[2] = {0, 0};
  int32_t gc_bias = 0;
  double cscore = 0;
  int32_t star_ptr[3] = {-1, -1, -1}; int32_t traceb = -1;
  double score = 0;
  int32_t mot_ndx = 0, mot_len = 0, mot_spacer = 0, mot_spacendx = 0; double mot_score = 0;
  double gc_score[3] = {0, 0, 0}, gc_cont = 0;
  double uscore = 0, tscore = 0, rscore = 0, sscore = 0;
  int32_t tracef = -1, ov_mark = -1, elim = 0;
};
struct GTrain {
  double gc = 0; int trans_table = 11; double st_wt = 4.35; double bias[3] = {0, 0, 0}; double type_wt[3] = {0, 0, 0}; int uses_sd = 0;
  double rbs_wt[28] = {0}; double ups_comp[32][4] = {{0}}; std::vector<double> mot_wt; double no_mot = 0; double gene_dc[4096] = {0};
  GTrain() : mot_wt((size_t)4 * 4 * 4096, 0.0) {}
  double &mw(int a, int b, int c) { return mot_wt[((size_t)a * 4 + b) * 4096 + c]; }
  double mw(int a, int b, int c) const { return mot_wt[((size_t)a * 4 + b) * 4096 + c]; }
};
// a sequence in the one-byte code (bits 0-1 base, unknown reads C; bit 2 unknown); strand-relative accessors
struct GSeq {
  const uint8_t *c = nullptr; int slen = 0;
  int fwd(int p) const { return c[p] & 3; }
  int rev(int p) const { return 3 - (c[slen - 1 - p] & 3); }
  int at(int strand, int p) const { return strand == 1 ? fwd(p) : rev(p); }
  int unk(int strand, int p) const { return (strand == 1 ? c[p] : c[slen - 1 - p]) >> 2; }
  bool gc(int p) const { const int b = c[p] & 3; return b == 1 || b == 2; }
  bool is_stop(int strand, int i, int tt) const {
    if (i < 0 || i + 2 >= slen) return false;
    if (unk(strand, i) || unk(strand, i + 1) || unk(strand, i + 2)) return false;
    if (at(strand, i) != 3) return false;
    const int b1 = at(strand, i + 1), b2 = at(strand, i + 2);
    if (b1 == 0 && (b2 == 0 || b2 == 2)) return true;
    if (b1 == 2 && b2 == 0) return tt != 4;
    return false;
  }
  int mer(int strand, int len, int pos) const { int ndx = 0; for (int i = 0; i < len; ++i) ndx |= at(strand, pos + i) << (2 * i); return ndx; }
};
struct Mask { int begin, end; };

inline double dmaxd(double a, double b) { return a > b ? a : b; }
inline double dmind(double a, double b) { return a < b ? a : b; }
int max_fr(int n1, int n2, int n3) { if (n1 > n2) return n1 > n3 ? 0 : 2; return n2 > n3 ? 1 : 2; }

std::vector<Mask> find_masks(const GSeq &s) {
  std::vector<Mask> m; int run = -1;
  for (int i = 0; i <= s.slen; ++i) {
    const bool u = i < s.slen && (s.c[i] >> 2);
    if (u && run < 0) run = i;
    if (!u && run >= 0) { if (i - run >= MASK_SIZE) m.push_back({run, i - 1}); run = -1; }
  }
  return m;
}
bool cross_mask(int x, int y, const std::vector<Mask> &m) { for (const Mask &k : m) if (y >= k.begin && x <= k.end) return true; return false; }

// device nodes of one sequence -> working order, -m masks applied (a start whose ORF crosses a mask is dropped, and with its last start a
// stop node goes too: add_nodes only records a stop behind a recorded start)
void finish_nodes(std::vector<GNode> &nodes, const std::vector<Mask> &masks) {
  if (!masks.empty()) {
    std::vector<GNode> keep; keep.reserve(nodes.size());
    for (const GNode &n : nodes) {
      if (n.type == G_STOP) { keep.push_back(n); continue; }
      const int x = std::min(n.ndx, n.stop_val), y = std::max(n.ndx, n.stop_val);
      if (!cross_mask(x, y, masks)) keep.push_back(n);
    }
    // stop nodes without a surviving start of their ORF (same strand, stop_val == the stop's position)
    std::vector<std::pair<int, int>> have;                       // (strand, stop position) of surviving starts
    for (const GNode &n : keep) if (n.type != G_STOP) have.push_back({n.strand, n.stop_val});
    std::sort(have.begin(), have.end());
    nodes.clear();
    for (const GNode &n : keep) {
      if (n.type == G_STOP && !std::binary_search(have.begin(), have.end(), std::make_pair(n.strand, n.ndx))) continue;
      nodes.push_back(n);
    }
  }
  auto before = [](const GNode &x, const GNode &y) {
    if (x.ndx != y.ndx) return x.ndx < y.ndx;
    return x.strand > y.strand;                                  // forward strand first
  };
  if (!std::is_sorted(nodes.begin(), nodes.end(), before)) std::sort(nodes.begin(), nodes.end(), before);      // (the caller hands them over in order; the filter above keeps it)
}

std::vector<int> calc_most_gc_frame(const GSeq &s) {
  const int slen = s.slen;
  std::vector<int> gp(slen + 3, -1), fwd(slen + 3, 0), bwd(slen + 3, 0), tot(slen + 3, 0);
  for (int j = 0; j < slen; ++j) {
    fwd[j] = (j < 3 ? 0 : fwd[j - 3]) + (s.gc(j) ? 1 : 0);
    const int r = slen - j - 1;
    bwd[r] = (j < 3 ? 0 : bwd[r + 3]) + (s.gc(r) ? 1 : 0);
  }
  for (int i = 0; i < slen; ++i) {
    tot[i] = fwd[i] + bwd[i] - (s.gc(i) ? 1 : 0);
    if (i - GC_WINDOW / 2 >= 0) tot[i] -= fwd[i - GC_WINDOW / 2];
    if (i + GC_WINDOW / 2 < slen) tot[i] -= bwd[i + GC_WINDOW / 2];
  }
  for (int i = 0; i < slen - 2; i += 3) { const int w = max_fr(tot[i], tot[i + 1], tot[i + 2]); gp[i] = gp[i + 1] = gp[i + 2] = w; }
  return gp;
}
void record_gc_bias(const std::vector<int> &gc, std::vector<GNode> &nod, GTrain &t) {
  const int nn = (int)nod.size();
  int ctr[3][3] = {{0}}, last[3] = {0, 0, 0};
  if (nn == 0) return;
  for (int i = nn - 1; i >= 0; --i) {
    const int fr = nod[i].ndx % 3, frmod = 3 - fr;
    if (nod[i].strand == 1 && nod[i].type == G_STOP) {
      for (int j = 0; j < 3; ++j) ctr[fr][j] = 0;
      last[fr] = nod[i].ndx; ctr[fr][(gc[nod[i].ndx] + frmod) % 3] = 1;
    } else if (nod[i].strand == 1) {
      for (int j = last[fr] - 3; j >= nod[i].ndx; j -= 3) ctr[fr][(gc[j] + frmod) % 3]++;
      nod[i].gc_bias = max_fr(ctr[fr][0], ctr[fr][1], ctr[fr][2]);
      for (int j = 0; j < 3; ++j) { nod[i].gc_score[j] = 3.0 * ctr[fr][j]; nod[i].gc_score[j] /= 1.0 * (nod[i].stop_val - nod[i].ndx + 3); }
      last[fr] = nod[i].ndx;
    }
  }
  for (int i = 0; i < nn; ++i) {
    const int fr = nod[i].ndx % 3, frmod = fr;
    if (nod[i].strand == -1 && nod[i].type == G_STOP) {
      for (int j = 0; j < 3; ++j) ctr[fr][j] = 0;
      last[fr] = nod[i].ndx; ctr[fr][((3 - gc[nod[i].ndx]) + frmod) % 3] = 1;
    } else if (nod[i].strand == -1) {
      for (int j = last[fr] + 3; j <= nod[i].ndx; j += 3) ctr[fr][((3 - gc[j]) + frmod) % 3]++;
      nod[i].gc_bias = max_fr(ctr[fr][0], ctr[fr][1], ctr[fr][2]);
      for (int j = 0; j < 3; ++j) { nod[i].gc_score[j] = 3.0 * ctr[fr][j]; nod[i].gc_score[j] /= 1.0 * (nod[i].ndx - nod[i].stop_val + 3); }
      last[fr] = nod[i].ndx;
    }
  }
  for (int i = 0; i < 3; ++i) t.bias[i] = 0.0;
  for (int i = 0; i < nn; ++i) if (nod[i].type != G_STOP) {
    const int len = std::abs(nod[i].stop_val - nod[i].ndx) + 1;
    t.bias[nod[i].gc_bias] += (nod[i].gc_score[nod[i].gc_bias] * len) / 1000.0;
  }
  const double tot = t.bias[0] + t.bias[1] + t.bias[2];
  for (int i = 0; i < 3; ++i) t.bias[i] *= (3.0 / tot);
}

double intergenic_mod(const GNode &n1, const GNode &n2, const GTrain &t) {
  double rval = 0.0; int ovlp = 0;
  if ((n1.strand == 1 && n2.strand == 1 && (n1.ndx + 2 == n2.ndx || n1.ndx - 1 == n2.ndx)) ||
      (n1.strand == -1 && n2.strand == -1 && (n1.ndx + 2 == n2.ndx || n1.ndx - 1 == n2.ndx))) {
    if (n1.strand == 1 && n2.rscore < 0) rval -= n2.rscore;
    if (n1.strand == -1 && n1.rscore < 0) rval -= n1.rscore;
    if (n1.strand == 1 && n2.uscore < 0) rval -= n2.uscore;
    if (n1.strand == -1 && n1.uscore < 0) rval -= n1.uscore;
  }
  const int dist = std::abs(n1.ndx - n2.ndx);
  if (n1.strand == 1 && n2.strand == 1 && n1.ndx + 2 >= n2.ndx) ovlp = 1;
  else if (n1.strand == -1 && n2.strand == -1 && n1.ndx >= n2.ndx + 2) ovlp = 1;
  if (dist > 3 * OPER_DIST || n1.strand != n2.strand) rval -= 0.15 * t.st_wt;
  else if ((dist <= OPER_DIST && ovlp == 0) || dist < 0.25 * OPER_DIST) rval += (2.0 - (double)dist / OPER_DIST) * 0.15 * t.st_wt;
  return rval;
}

void record_overlapping_starts(std::vector<GNode> &nod, const GTrain &t, int flag) {
  const int nn = (int)nod.size();
  for (int i = 0; i < nn; ++i) {
    for (int j = 0; j < 3; ++j) nod[i].star_ptr[j] = -1;
    if (nod[i].type != G_STOP || nod[i].edge == 1) continue;
    double max_sc = -100.0;
    if (nod[i].strand == 1) {
      for (int j = i + 3; j >= 0; --j) {
        if (j >= nn || nod[j].ndx > nod[i].ndx + 2) continue;
        if (nod[j].ndx + MAX_SAM_OVLP < nod[i].ndx) break;
        if (nod[j].strand == 1 && nod[j].type != G_STOP) {
          if (nod[j].stop_val <= nod[i].ndx) continue;
          const int f = nod[j].ndx % 3;
          if (flag == 0 && nod[i].star_ptr[f] == -1) nod[i].star_ptr[f] = j;
          else if (flag == 1) { const double sc = nod[j].cscore + nod[j].sscore + intergenic_mod(nod[i], nod[j], t); if (sc > max_sc) { nod[i].star_ptr[f] = j; max_sc = sc; } }
        }
      }
    } else {
      for (int j = i - 3; j < nn; ++j) {
        if (j < 0 || nod[j].ndx < nod[i].ndx - 2) continue;
        if (nod[j].ndx - MAX_SAM_OVLP > nod[i].ndx) break;
        if (nod[j].strand == -1 && nod[j].type != G_STOP) {
          if (nod[j].stop_val >= nod[i].ndx) continue;
          const int f = nod[j].ndx % 3;
          if (flag == 0 && nod[i].star_ptr[f] == -1) nod[i].star_ptr[f] = j;
          else if (flag == 1) { const double sc = nod[j].cscore + nod[j].sscore + intergenic_mod(nod[j], nod[i], t); if (sc > max_sc) { nod[i].star_ptr[f] = j; max_sc = sc; } }
        }
      }
    }
  }
}

// first candidate predecessor of every node (dprog.c: the 500-node horizon, moved back behind the stop of a giant ORF)
void dp_window(const std::vector<GNode> &nod, std::vector<uint32_t> &dp_min) {
  const int nn = (int)nod.size();
  dp_min.assign(nn, 0);
  for (int i = 0; i < nn; ++i) {
    int mn = i < MAX_NODE_DIST ? 0 : i - MAX_NODE_DIST;
    if (nod[i].strand == -1 && nod[i].type != G_STOP && nod[mn].ndx >= nod[i].stop_val) while (mn >= 0 && nod[mn].ndx != nod[i].stop_val) mn--;
    if (nod[i].strand == 1 && nod[i].type == G_STOP && nod[mn].ndx >= nod[i].stop_val) while (mn >= 0 && nod[mn].ndx != nod[i].stop_val) mn--;
    mn = mn < MAX_NODE_DIST ? 0 : mn - MAX_NODE_DIST;
    dp_min[i] = (uint32_t)mn;
  }
}
// what follows the forward sweep of the dynamic program (the sweep itself ran on the device: score / traceb / ov_mark are in place)
int dprog_finish(std::vector<GNode> &nod) {
  const int nn = (int)nod.size();
  int max_ndx = -1; double max_sc = -1.0;
  if (nn == 0) return -1;
  for (int i = nn - 1; i >= 0; --i) {
    if (nod[i].strand == 1 && nod[i].type != G_STOP) continue;
    if (nod[i].strand == -1 && nod[i].type == G_STOP) continue;
    if (nod[i].score > max_sc) { max_sc = nod[i].score; max_ndx = i; }
  }
  if (max_ndx < 0) return -1;
  int path = max_ndx;
  while (nod[path].traceb != -1) {
    const int nxt = nod[path].traceb;
    if (nod[path].strand == -1 && nod[path].type == G_STOP && nod[nxt].strand == 1 && nod[nxt].type == G_STOP && nod[path].ov_mark != -1 && nod[path].ndx > nod[nxt].ndx) {
      const int tmp = nod[path].star_ptr[nod[path].ov_mark];
      int i;
      for (i = tmp; nod[i].ndx != nod[tmp].stop_val; --i);
      nod[path].traceb = tmp; nod[tmp].traceb = i; nod[i].ov_mark = -1; nod[i].traceb = nxt;
    }
    path = nod[path].traceb;
  }
  path = max_ndx;
  while (nod[path].traceb != -1) {
    const int nxt = nod[path].traceb;
    if (nod[path].strand == -1 && nod[path].type != G_STOP && nod[nxt].strand == 1 && nod[nxt].type == G_STOP) {
      int i;
      for (i = path; nod[i].ndx != nod[path].stop_val; --i);
      nod[path].traceb = i; nod[i].traceb = nxt;
    }
    if (nod[path].strand == 1 && nod[path].type == G_STOP && nod[nxt].strand == 1 && nod[nxt].type == G_STOP) {
      nod[path].traceb = nod[nxt].star_ptr[nod[path].ndx % 3];
      nod[nod[path].traceb].traceb = nxt;
    }
    if (nod[path].strand == -1 && nod[path].type == G_STOP && nod[nxt].strand == -1 && nod[nxt].type == G_STOP) {
      nod[path].traceb = nod[path].star_ptr[nod[nxt].ndx % 3];
      nod[nod[path].traceb].traceb = nxt;
    }
    path = nod[path].traceb;
  }
  path = max_ndx;
  while (nod[path].traceb != -1) { nod[nod[path].traceb].tracef = path; path = nod[path].traceb; }
  return nod[max_ndx].traceb == -1 ? -1 : max_ndx;
}

void calc_dicodon_gene(GTrain &t, const GSeq &s, const std::vector<GNode> &nod, int dbeg) {
  std::vector<double> bg(4096, 0.0); std::vector<int> counts(4096, 0), bc(4096, 0);
  const int slen = s.slen;
  long long g = 0;
  {
    // both strands' hexamers at every position: the reverse hexamer at position r is the reverse complement of the forward one at slen-6-r
    int f = 0;
    for (int i = 0; i < 5 && i < slen; ++i) f |= s.fwd(i) << (2 * i);
    for (int i = 0; i < slen - 5; ++i) {
      f = (i == 0 ? f : (f >> 2)) | (s.fwd(i + 5) << 10);
      bc[f & 4095]++;
      int r = 0;                                               // reverse complement of f, first base lowest
      for (int k = 0; k < 6; ++k) r |= (3 - ((f >> (2 * k)) & 3)) << (2 * (5 - k));
      bc[r]++;
      g += 2;
    }
    for (int i = 0; i < 4096; ++i) bg[i] = g ? (double)bc[i] / (double)g : 0.0;
  }
  int glob = 0, left = -1, right = -1, in_gene = 0;
  for (int path = dbeg; path != -1; path = nod[path].traceb) {
    if (nod[path].strand == -1 && nod[path].type != G_STOP) { in_gene = -1; left = slen - nod[path].ndx - 1; }
    if (nod[path].strand == 1 && nod[path].type == G_STOP) { in_gene = 1; right = nod[path].ndx + 2; }
    if (in_gene == -1 && nod[path].strand == -1 && nod[path].type == G_STOP) {
      right = slen - nod[path].ndx + 1;
      for (int i = left; i < right - 5; i += 3) { counts[s.mer(-1, 6, i)]++; glob++; }
      in_gene = 0;
    }
    if (in_gene == 1 && nod[path].strand == 1 && nod[path].type != G_STOP) {
      left = nod[path].ndx;
      for (int i = left; i < right - 5; i += 3) { counts[s.mer(1, 6, i)]++; glob++; }
      in_gene = 0;
    }
  }
  for (int i = 0; i < 4096; ++i) {
    const double prob = glob ? (counts[i] * 1.0) / (glob * 1.0) : 0.0;
    if (prob == 0 && bg[i] != 0) t.gene_dc[i] = -5.0;
    else if (bg[i] == 0) t.gene_dc[i] = 0.0;
    else t.gene_dc[i] = log(prob / bg[i]);
    if (t.gene_dc[i] > 5.0) t.gene_dc[i] = 5.0;
    if (t.gene_dc[i] < -5.0) t.gene_dc[i] = -5.0;
  }
}

// raw_coding_score behind its first pass (the per-ORF hexamer sums came from the device)
void coding_score_passes(std::vector<GNode> &nod, const GTrain &t) {
  const int nn = (int)nod.size();
  double score[3], lfac, no_stop, gsize; const double gc = t.gc;
  if (t.trans_table != 11) { no_stop = ((1 - gc) * (1 - gc) * gc) / 8.0; no_stop += ((1 - gc) * (1 - gc) * (1 - gc)) / 8.0; no_stop = 1 - no_stop; }
  else { no_stop = ((1 - gc) * (1 - gc) * gc) / 4.0; no_stop += ((1 - gc) * (1 - gc) * (1 - gc)) / 8.0; no_stop = 1 - no_stop; }
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = 0; i < 3; ++i) score[i] = -10000;
    for (int k = 0; k < nn; ++k) {
      const int i = pass == 0 ? k : nn - 1 - k;
      if (nod[i].strand != (pass == 0 ? 1 : -1)) continue;
      const int fr = nod[i].ndx % 3;
      if (nod[i].type == G_STOP) score[fr] = -10000;
      else if (nod[i].cscore > score[fr]) score[fr] = nod[i].cscore;
      else nod[i].cscore -= (score[fr] - nod[i].cscore);
    }
  }
  const double l80 = log((1 - pow(no_stop, 80)) / pow(no_stop, 80));
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = 0; i < 3; ++i) score[i] = -10000;
    for (int k = 0; k < nn; ++k) {
      const int i = pass == 0 ? k : nn - 1 - k;
      if (nod[i].strand != (pass == 0 ? 1 : -1)) continue;
      const int fr = nod[i].ndx % 3;
      if (nod[i].type == G_STOP) { score[fr] = -10000; continue; }
      gsize = ((double)(std::abs(nod[i].stop_val - nod[i].ndx) + 3.0)) / 3.0;
      if (gsize > 1000.0) { lfac = log((1 - pow(no_stop, 1000.0)) / pow(no_stop, 1000.0)); lfac -= l80; lfac *= (gsize - 80) / 920.0; }
      else { lfac = log((1 - pow(no_stop, gsize)) / pow(no_stop, gsize)); lfac -= l80; }
      if (lfac > score[fr]) score[fr] = lfac;
      else lfac -= dmaxd(dmind(score[fr] - lfac, lfac), 0);
      if (lfac > 3.0 && nod[i].cscore < 0.5 * lfac) nod[i].cscore = 0.5 * lfac;
      nod[i].cscore += lfac;
    }
  }
}

void count_upstream_composition(const GSeq &s, int strand, int start, GTrain &t) {
  int count = 0;
  for (int i = 1; i < 45; ++i) { if (i > 2 && i < 15) continue; if (start - i >= 0) t.ups_comp[count][s.at(strand, start - i)] += 1.0; count++; }
}
void score_upstream_composition(const GSeq &s, int strand, int start, GNode &n, const GTrain &t) {
  int count = 0; n.uscore = 0.0;
  for (int i = 1; i < 45; ++i) { if (i > 2 && i < 15) continue; if (start - i < 0) continue; n.uscore += 0.4 * t.st_wt * t.ups_comp[count][s.at(strand, start - i)]; count++; }
}
void ups_to_log(GTrain &t) {
  for (int i = 0; i < 32; ++i) {
    double sum = 0.0;
    for (int j = 0; j < 4; ++j) sum += t.ups_comp[i][j];
    if (sum == 0.0) { for (int j = 0; j < 4; ++j) t.ups_comp[i][j] = 0.0; continue; }
    for (int j = 0; j < 4; ++j) {
      double x = t.ups_comp[i][j] / sum; const bool at = (j == 0 || j == 3);
      if (t.gc > 0.1 && t.gc < 0.9) x = at ? log(x * 2.0 / (1.0 - t.gc)) : log(x * 2.0 / t.gc);
      else if (t.gc <= 0.1) x = at ? log(x * 2.0 / 0.90) : log(x * 2.0 / 0.10);
      else x = at ? log(x * 2.0 / 0.10) : log(x * 2.0 / 0.90);
      if (x > 4.0) x = 4.0;
      if (x < -4.0) x = -4.0;
      t.ups_comp[i][j] = x;
    }
  }
}
int best_rbs(const GNode &n, const GTrain &t) {
  if (t.rbs_wt[n.rbs[0]] > t.rbs_wt[n.rbs[1]] + 1.0 || n.rbs[1] == 0) return n.rbs[0];
  if (t.rbs_wt[n.rbs[0]] < t.rbs_wt[n.rbs[1]] - 1.0 || n.rbs[0] == 0) return n.rbs[1];
  return n.rbs[0] > n.rbs[1] ? n.rbs[0] : n.rbs[1];
}
void type_bg(const std::vector<GNode> &nod, double *tbg) {
  double sum = 0.0;
  for (int i = 0; i < 3; ++i) tbg[i] = 0.0;
  for (const GNode &n : nod) if (n.type != G_STOP) tbg[n.type] += 1.0;
  for (int i = 0; i < 3; ++i) sum += tbg[i];
  for (int i = 0; i < 3; ++i) tbg[i] = sum ? tbg[i] / sum : 0.0;
}
double update_type_wt(GTrain &t, double *treal, const double *tbg) {
  double sum = 0.0;
  for (int j = 0; j < 3; ++j) sum += treal[j];
  if (sum == 0.0) for (int j = 0; j < 3; ++j) t.type_wt[j] = 0.0;
  else for (int j = 0; j < 3; ++j) {
    treal[j] /= sum;
    t.type_wt[j] = tbg[j] != 0 ? log(treal[j] / tbg[j]) : -4.0;
    if (t.type_wt[j] > 4.0) t.type_wt[j] = 4.0;
    if (t.type_wt[j] < -4.0) t.type_wt[j] = -4.0;
  }
  return sum;
}
void train_starts_sd(const GSeq &s, std::vector<GNode> &nod, GTrain &t) {
  const int nn = (int)nod.size(), slen = s.slen;
  int rbs[3], type[3], bndx[3]; double sum, rbg[28], rreal[28], best[3], sthresh = 35.0, tbg[3], treal[3];
  const double wt = t.st_wt;
  for (int j = 0; j < 3; ++j) t.type_wt[j] = 0.0;
  for (int j = 0; j < 28; ++j) t.rbs_wt[j] = 0.0;
  memset(t.ups_comp, 0, sizeof(t.ups_comp));
  type_bg(nod, tbg);
  for (int it = 0; it < 10; ++it) {
    for (int j = 0; j < 28; ++j) rbg[j] = 0.0;
    for (int j = 0; j < nn; ++j) { if (nod[j].type == G_STOP || nod[j].edge == 1) continue; rbg[best_rbs(nod[j], t)] += 1.0; }
    sum = 0.0; for (int j = 0; j < 28; ++j) sum += rbg[j];
    for (int j = 0; j < 28; ++j) rbg[j] = sum ? rbg[j] / sum : 0.0;
    for (int j = 0; j < 28; ++j) rreal[j] = 0.0;
    for (int j = 0; j < 3; ++j) treal[j] = 0.0;
    for (int pass = 0; pass < 2; ++pass) {
      const int str = pass == 0 ? 1 : -1;
      for (int j = 0; j < 3; ++j) { best[j] = 0.0; bndx[j] = -1; rbs[j] = 0; type[j] = 0; }
      for (int k = 0; k < nn; ++k) {
        const int j = pass == 0 ? k : nn - 1 - k;
        if (nod[j].type != G_STOP && nod[j].edge == 1) continue;
        const int fr = nod[j].ndx % 3;
        if (nod[j].type == G_STOP && nod[j].strand == str) {
          if (best[fr] >= sthresh && nod[bndx[fr]].ndx % 3 == fr) {
            rreal[rbs[fr]] += 1.0; treal[type[fr]] += 1.0;
            if (it == 9) count_upstream_composition(s, str, str == 1 ? nod[bndx[fr]].ndx : slen - 1 - nod[bndx[fr]].ndx, t);
          }
          best[fr] = 0.0; bndx[fr] = -1; rbs[fr] = 0; type[fr] = 0;
        } else if (nod[j].strand == str && nod[j].type != G_STOP) {
          const int mr = best_rbs(nod[j], t);
          const double v = nod[j].cscore + wt * t.rbs_wt[mr] + wt * t.type_wt[nod[j].type];
          if (v >= best[fr]) { best[fr] = nod[j].cscore + wt * t.rbs_wt[mr]; best[fr] += wt * t.type_wt[nod[j].type]; bndx[fr] = j; type[fr] = nod[j].type; rbs[fr] = mr; }
        }
      }
    }
    sum = 0.0; for (int j = 0; j < 28; ++j) sum += rreal[j];
    if (sum == 0.0) for (int j = 0; j < 28; ++j) t.rbs_wt[j] = 0.0;
    else for (int j = 0; j < 28; ++j) {
      rreal[j] /= sum;
      t.rbs_wt[j] = rbg[j] != 0 ? log(rreal[j] / rbg[j]) : -4.0;
      if (t.rbs_wt[j] > 4.0) t.rbs_wt[j] = 4.0;
      if (t.rbs_wt[j] < -4.0) t.rbs_wt[j] = -4.0;
    }
    sum = update_type_wt(t, treal, tbg);
    if (sum <= (double)nn / 2000.0) sthresh /= 2.0;
  }
  ups_to_log(t);
}
void determine_sd_usage(GTrain &t) {
  t.uses_sd = 1;
  if (t.rbs_wt[0] >= 0.0) t.uses_sd = 0;
  if (t.rbs_wt[16] < 1.0 && t.rbs_wt[13] < 1.0 && t.rbs_wt[15] < 1.0 && (t.rbs_wt[0] >= -0.5 || (t.rbs_wt[22] < 2.0 && t.rbs_wt[24] < 2.0 && t.rbs_wt[27] < 2.0))) t.uses_sd = 0;
}
int spacer_ndx(int j, int start, int i) { if (j <= start - 16 - i) return 3; if (j <= start - 14 - i) return 2; if (j >= start - 7 - i) return 1; return 0; }
void find_best_upstream_motif(const GTrain &t, const GSeq &s, GNode &n, int stage) {
  if (n.type == G_STOP || n.edge == 1) return;
  const int start = n.strand == 1 ? n.ndx : s.slen - 1 - n.ndx;
  int max_spacer = 0, max_spacendx = 0, max_len = 0, max_ndx = 0; double max_sc = -100.0;
  for (int i = 3; i >= 0; --i) for (int j = start - 18 - i; j <= start - 6 - i; ++j) {
    if (j < 0) continue;
    const int spacer = start - j - i - 3, sp = spacer_ndx(j, start, i), index = s.mer(n.strand, i + 3, j);
    const double score = t.mw(i, sp, index);
    if (score > max_sc) { max_sc = score; max_spacendx = sp; max_spacer = spacer; max_ndx = index; max_len = i + 3; }
  }
  if (stage == 2 && (max_sc == -4.0 || max_sc < t.no_mot + 0.69)) { n.mot_ndx = 0; n.mot_len = 0; n.mot_spacendx = 0; n.mot_spacer = 0; n.mot_score = t.no_mot; }
  else { n.mot_ndx = max_ndx; n.mot_len = max_len; n.mot_spacendx = max_spacendx; n.mot_spacer = max_spacer; n.mot_score = max_sc; }
}
struct MotTab { std::vector<double> v; MotTab() : v((size_t)4 * 4 * 4096, 0.0) {} double &at(int a, int b, int c) { return v[((size_t)a * 4 + b) * 4096 + c]; } void zero() { std::fill(v.begin(), v.end(), 0.0); } };
void update_motif_counts(MotTab &mcnt, double &zero, const GSeq &s, const GNode &n, int stage) {
  if (n.type == G_STOP || n.edge == 1) return;
  if (n.mot_len == 0) { zero += 1.0; return; }
  const int start = n.strand == 1 ? n.ndx : s.slen - 1 - n.ndx;
  if (stage == 0) {
    for (int i = 3; i >= 0; --i) for (int j = start - 18 - i; j <= start - 6 - i; ++j) { if (j < 0) continue; const int m = s.mer(n.strand, i + 3, j); for (int k = 0; k < 4; ++k) mcnt.at(i, k, m) += 1.0; }
  } else if (stage == 1) {
    mcnt.at(n.mot_len - 3, n.mot_spacendx, n.mot_ndx) += 1.0;
    for (int i = 0; i < n.mot_len - 3; ++i) for (int j = start - n.mot_spacer - n.mot_len; j <= start - n.mot_spacer - (i + 3); ++j) {
      if (j < 0) continue;
      mcnt.at(i, spacer_ndx(j, start, i), s.mer(n.strand, i + 3, j)) += 1.0;
    }
  } else mcnt.at(n.mot_len - 3, n.mot_spacendx, n.mot_ndx) += 1.0;
}
// The 52 upstream words of a start node (lengths 6..3, thirteen positions each) do not change over the 20 training iterations: they are
// read off the sequence once per node (0xffff: before the sequence start) and every iteration only looks the weights up, in the order of
// find_best_upstream_motif (so the first maximum wins as there).  Entry k = (3 - i) * 13 + d: word length i + 3 at start - 18 - i + d.
constexpr int MOT_WORDS = 52;
inline int mot_sp_of_d(int d) { return d <= 2 ? 3 : d <= 4 ? 2 : d >= 11 ? 1 : 0; }       // spacer_ndx(start - 18 - i + d, start, i)
void upstream_words(const GSeq &s, const GNode &n, uint16_t *w) {
  const int start = n.strand == 1 ? n.ndx : s.slen - 1 - n.ndx;
  int k = 0;
  for (int i = 3; i >= 0; --i) for (int d = 0; d <= 12; ++d) { const int j = start - 18 - i + d; w[k++] = j < 0 ? (uint16_t)0xffff : (uint16_t)s.mer(n.strand, i + 3, j); }
}
void find_best_upstream_motif_w(const GTrain &t, const uint16_t *w, GNode &n, int stage) {
  int max_spacer = 0, max_spacendx = 0, max_len = 0, max_ndx = 0; double max_sc = -100.0;
  int k = 0;
  for (int i = 3; i >= 0; --i) for (int d = 0; d <= 12; ++d, ++k) {
    if (w[k] == 0xffff) continue;
    const int sp = mot_sp_of_d(d), index = w[k];
    const double score = t.mw(i, sp, index);
    if (score > max_sc) { max_sc = score; max_spacendx = sp; max_spacer = 15 - d; max_ndx = index; max_len = i + 3; }
  }
  if (stage == 2 && (max_sc == -4.0 || max_sc < t.no_mot + 0.69)) { n.mot_ndx = 0; n.mot_len = 0; n.mot_spacendx = 0; n.mot_spacer = 0; n.mot_score = t.no_mot; }
  else { n.mot_ndx = max_ndx; n.mot_len = max_len; n.mot_spacendx = max_spacendx; n.mot_spacer = max_spacer; n.mot_score = max_sc; }
}
void update_motif_counts_stage0_w(MotTab &mcnt, double &zero, const uint16_t *w, const GNode &n) {
  if (n.mot_len == 0) { zero += 1.0; return; }
  int k = 0;
  for (int i = 3; i >= 0; --i) for (int d = 0; d <= 12; ++d, ++k) { if (w[k] == 0xffff) continue; for (int q = 0; q < 4; ++q) mcnt.at(i, q, w[k]) += 1.0; }
}
void build_coverage_map(MotTab &real, std::vector<int> &good, double ng) {
  auto G = [&](int a, int b, int c) -> int & { return good[((size_t)a * 4 + b) * 4096 + c]; };
  const double thresh = 0.2; int decomp[3];
  std::fill(good.begin(), good.end(), 0);
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 64; ++j) if (real.at(0, i, j) / ng >= thresh) for (int k = 0; k < 4; ++k) G(0, k, j) = 1;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 256; ++j) {
    decomp[0] = (j & 252) >> 2; decomp[1] = j & 63;
    if (G(0, i, decomp[0]) == 0 || G(0, i, decomp[1]) == 0) continue;
    G(1, i, j) = 1;
  }
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 1024; ++j) {
    decomp[0] = (j & 1008) >> 4; decomp[1] = (j & 252) >> 2; decomp[2] = j & 63;
    if (G(0, i, decomp[0]) == 0 || G(0, i, decomp[1]) == 0 || G(0, i, decomp[2]) == 0) continue;
    G(2, i, j) = 1;
    int tmp = j;
    for (int k = 0; k <= 16; k += 16) { tmp = tmp ^ k; for (int l = 0; l <= 32; l += 32) { tmp = tmp ^ l; if (G(2, i, tmp) == 0) G(2, i, tmp) = 2; } }
  }
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4096; ++j) {
    decomp[0] = (j & 4092) >> 2; decomp[1] = j & 1023;
    if (G(2, i, decomp[0]) == 0 || G(2, i, decomp[1]) == 0) continue;
    G(3, i, j) = (G(2, i, decomp[0]) == 1 && G(2, i, decomp[1]) == 1) ? 1 : 2;
  }
}
void train_starts_nonsd(const GSeq &s, std::vector<GNode> &nod, GTrain &t) {
  const int nn = (int)nod.size(), slen = s.slen;
  MotTab mbg, mreal; std::vector<int> mgood((size_t)4 * 4 * 4096, 0);
  int bndx[3], stage; double sum, ngenes, best[3], sthresh = 35.0, tbg[3], treal[3], zbg, zreal;
  const double wt = t.st_wt;
  for (int j = 0; j < 3; ++j) t.type_wt[j] = 0.0;
  std::fill(t.mot_wt.begin(), t.mot_wt.end(), 0.0); t.no_mot = 0.0;
  memset(t.ups_comp, 0, sizeof(t.ups_comp));
  type_bg(nod, tbg);
  std::vector<int32_t> wof(nn, -1); std::vector<uint16_t> words;
  { size_t ns = 0; for (int j = 0; j < nn; ++j) if (!(nod[j].type == G_STOP || nod[j].edge == 1)) wof[j] = (int32_t)(ns++);
    words.resize(ns * MOT_WORDS);
    for (int j = 0; j < nn; ++j) if (wof[j] >= 0) upstream_words(s, nod[j], words.data() + (size_t)wof[j] * MOT_WORDS); }
  for (int it = 0; it < 20; ++it) {
    stage = it < 4 ? 0 : it < 12 ? 1 : 2;
    mbg.zero(); zbg = 0.0;
    for (int j = 0; j < nn; ++j) {
      if (nod[j].type == G_STOP || nod[j].edge == 1) continue;
      const uint16_t *w = words.data() + (size_t)wof[j] * MOT_WORDS;
      find_best_upstream_motif_w(t, w, nod[j], stage);
      if (stage == 0) update_motif_counts_stage0_w(mbg, zbg, w, nod[j]); else update_motif_counts(mbg, zbg, s, nod[j], stage);
    }
    sum = zbg;
    for (double x : mbg.v) sum += x;
    if (sum != 0.0) { for (double &x : mbg.v) x /= sum; zbg /= sum; }
    mreal.zero(); zreal = 0.0; ngenes = 0.0;
    for (int j = 0; j < 3; ++j) treal[j] = 0.0;
    for (int pass = 0; pass < 2; ++pass) {
      const int str = pass == 0 ? 1 : -1;
      for (int j = 0; j < 3; ++j) { best[j] = 0.0; bndx[j] = -1; }
      for (int k = 0; k < nn; ++k) {
        const int j = pass == 0 ? k : nn - 1 - k;
        if (nod[j].type != G_STOP && nod[j].edge == 1) continue;
        const int fr = nod[j].ndx % 3;
        if (nod[j].type == G_STOP && nod[j].strand == str) {
          if (best[fr] >= sthresh) {
            ngenes += 1.0; treal[nod[bndx[fr]].type] += 1.0;
            update_motif_counts(mreal, zreal, s, nod[bndx[fr]], stage);
            if (it == 19) count_upstream_composition(s, str, str == 1 ? nod[bndx[fr]].ndx : slen - 1 - nod[bndx[fr]].ndx, t);
          }
          best[fr] = 0.0; bndx[fr] = -1;
        } else if (nod[j].strand == str && nod[j].type != G_STOP) {
          const double v = nod[j].cscore + wt * nod[j].mot_score + wt * t.type_wt[nod[j].type];
          if (v >= best[fr]) { best[fr] = nod[j].cscore + wt * nod[j].mot_score; best[fr] += wt * t.type_wt[nod[j].type]; bndx[fr] = j; }
        }
      }
    }
    if (stage < 2) build_coverage_map(mreal, mgood, ngenes);
    sum = zreal;
    for (double x : mreal.v) sum += x;
    if (sum == 0.0) { std::fill(t.mot_wt.begin(), t.mot_wt.end(), 0.0); t.no_mot = 0.0; }
    else {
      for (size_t c = 0; c < mreal.v.size(); ++c) {
        if (mgood[c] == 0) { zreal += mreal.v[c]; zbg += mreal.v[c]; mreal.v[c] = 0.0; mbg.v[c] = 0.0; }
        mreal.v[c] /= sum;
        double v = mbg.v[c] != 0 ? log(mreal.v[c] / mbg.v[c]) : -4.0;
        if (v > 4.0) v = 4.0;
        if (v < -4.0) v = -4.0;
        t.mot_wt[c] = v;
      }
      zreal /= sum;
      t.no_mot = zbg != 0 ? log(zreal / zbg) : -4.0;
      if (t.no_mot > 4.0) t.no_mot = 4.0;
      if (t.no_mot < -4.0) t.no_mot = -4.0;
    }
    sum = update_type_wt(t, treal, tbg);
    if (sum <= (double)nn / 2000.0) sthresh /= 2.0;
  }
  ups_to_log(t);
}

void calc_orf_gc(const GSeq &s, std::vector<GNode> &nod) {
  for (GNode &n : nod) {
    if (n.type == G_STOP) continue;
    const int a = n.strand == 1 ? n.ndx : n.stop_val, b = n.strand == 1 ? n.stop_val + 2 : n.ndx;
    int g = 0;
    for (int j = (a < 0 ? 0 : a); j <= b && j < s.slen; ++j) g += s.gc(j) ? 1 : 0;
    n.gc_cont = (double)g / (double)(std::abs(n.stop_val - n.ndx) + 3);
  }
}
// score_nodes behind the device parts (raw hexamer sums and SD bins are in place)
void score_nodes_rest(const GSeq &s, std::vector<GNode> &nod, const GTrain &t, int closed) {
  const int nn = (int)nod.size(), slen = s.slen, tt = t.trans_table;
  calc_orf_gc(s, nod);
  coding_score_passes(nod, t);
  if (t.uses_sd != 1) for (GNode &n : nod) { if (n.type == G_STOP || n.edge == 1) continue; find_best_upstream_motif(t, s, n, 2); }
  for (int i = 0; i < nn; ++i) {
    GNode &n = nod[i];
    if (n.type == G_STOP) continue;
    int edge_gene = 0;
    if (n.edge == 1) edge_gene++;
    if ((n.strand == 1 && !s.is_stop(1, n.stop_val, tt)) || (n.strand == -1 && !s.is_stop(-1, slen - 1 - n.stop_val, tt))) edge_gene++;
    if (n.edge == 1) { n.tscore = EDGE_BONUS * t.st_wt / edge_gene; n.uscore = 0.0; n.rscore = 0.0; }
    else {
      n.tscore = t.type_wt[n.type] * t.st_wt;
      const double rbs1 = t.rbs_wt[n.rbs[0]], rbs2 = t.rbs_wt[n.rbs[1]], sd_score = dmaxd(rbs1, rbs2) * t.st_wt;
      if (t.uses_sd == 1) n.rscore = sd_score;
      else { n.rscore = t.st_wt * n.mot_score; if (n.rscore < sd_score && t.no_mot > -0.5) n.rscore = sd_score; }
      score_upstream_composition(s, n.strand, n.strand == 1 ? n.ndx : slen - 1 - n.ndx, n, t);
      if (closed == 0 && n.ndx <= 2 && n.strand == 1) n.uscore += EDGE_UPS * t.st_wt;
      else if (closed == 0 && n.ndx >= slen - 3 && n.strand == -1) n.uscore += EDGE_UPS * t.st_wt;
      else if (i < 500 && n.strand == 1) { for (int j = i - 1; j >= 0; --j) if (nod[j].edge == 1 && n.stop_val == nod[j].stop_val) { n.uscore += EDGE_UPS * t.st_wt; break; } }
      else if (i >= nn - 500 && n.strand == -1) { for (int j = i + 1; j < nn; ++j) if (nod[j].edge == 1 && n.stop_val == nod[j].stop_val) { n.uscore += EDGE_UPS * t.st_wt; break; } }
    }
    if (((n.ndx <= 2 && n.strand == 1) || (n.ndx >= slen - 3 && n.strand == -1)) && n.edge == 0 && closed == 0) {
      edge_gene++; n.edge = 1; n.tscore = 0.0; n.uscore = EDGE_BONUS * t.st_wt / edge_gene; n.rscore = 0.0;
    }
    if (n.edge == 0 && edge_gene == 1) n.uscore -= 0.5 * EDGE_BONUS * t.st_wt;
    if (edge_gene == 0 && std::abs(n.ndx - n.stop_val) < 250) {
      const double negf = 250.0 / (float)std::abs(n.ndx - n.stop_val), posf = (float)std::abs(n.ndx - n.stop_val) / 250.0;
      if (n.rscore < 0) n.rscore *= negf;
      if (n.uscore < 0) n.uscore *= negf;
      if (n.tscore < 0) n.tscore *= negf;
      if (n.rscore > 0) n.rscore *= posf;
      if (n.uscore > 0) n.uscore *= posf;
      if (n.tscore > 0) n.tscore *= posf;
    }
    n.sscore = n.tscore + n.rscore + n.uscore;
    if (n.cscore < 0.0) { if (edge_gene > 0 && n.edge == 0) n.sscore -= t.st_wt; else n.sscore -= 0.5; }
  }
}

void eliminate_bad_genes(std::vector<GNode> &nod, int dbeg, const GTrain &t) {
  if (dbeg == -1) return;
  int path = dbeg;
  while (nod[path].traceb != -1) path = nod[path].traceb;
  while (nod[path].tracef != -1) {
    if (nod[path].strand == 1 && nod[path].type == G_STOP) nod[nod[path].tracef].sscore += intergenic_mod(nod[path], nod[nod[path].tracef], t);
    if (nod[path].strand == -1 && nod[path].type != G_STOP) nod[path].sscore += intergenic_mod(nod[path], nod[nod[path].tracef], t);
    path = nod[path].tracef;
  }
  path = dbeg;
  while (nod[path].traceb != -1) path = nod[path].traceb;
  while (nod[path].tracef != -1) {
    if (nod[path].strand == 1 && nod[path].type != G_STOP && nod[path].cscore + nod[path].sscore < 0) { nod[path].elim = 1; nod[nod[path].tracef].elim = 1; }
    if (nod[path].strand == -1 && nod[path].type == G_STOP && nod[nod[path].tracef].cscore + nod[nod[path].tracef].sscore < 0) { nod[path].elim = 1; nod[nod[path].tracef].elim = 1; }
    path = nod[path].tracef;
  }
}
struct GeneRec { int begin = 0, end = 0, start_ndx = 0, stop_ndx = 0; };
std::vector<GeneRec> add_genes(const std::vector<GNode> &nod, int dbeg) {
  std::vector<GeneRec> gl;
  if (dbeg == -1) return gl;
  int path = dbeg;
  while (nod[path].traceb != -1) path = nod[path].traceb;
  GeneRec cur;
  while (path != -1) {
    if (nod[path].elim == 1) { path = nod[path].tracef; continue; }
    if (nod[path].strand == 1 && nod[path].type != G_STOP) { cur.begin = nod[path].ndx + 1; cur.start_ndx = path; }
    if (nod[path].strand == -1 && nod[path].type == G_STOP) { cur.begin = nod[path].ndx - 1; cur.stop_ndx = path; }
    if (nod[path].strand == 1 && nod[path].type == G_STOP) { cur.end = nod[path].ndx + 3; cur.stop_ndx = path; gl.push_back(cur); }
    if (nod[path].strand == -1 && nod[path].type != G_STOP) { cur.end = nod[path].ndx + 1; cur.start_ndx = path; gl.push_back(cur); }
    path = nod[path].tracef;
  }
  return gl;
}
void tweak_final_starts(std::vector<GeneRec> &genes, const std::vector<GNode> &nod, const GTrain &t) {
  const int ng = (int)genes.size(), nn = (int)nod.size();
  for (int i = 0; i < ng; ++i) {
    const int ndx = genes[i].start_ndx;
    const double sc = nod[ndx].sscore + nod[ndx].cscore;
    double igm = 0.0;
    if (i > 0 && nod[ndx].strand == 1 && nod[genes[i - 1].start_ndx].strand == 1) igm = intergenic_mod(nod[genes[i - 1].stop_ndx], nod[ndx], t);
    if (i > 0 && nod[ndx].strand == 1 && nod[genes[i - 1].start_ndx].strand == -1) igm = intergenic_mod(nod[genes[i - 1].start_ndx], nod[ndx], t);
    if (i < ng - 1 && nod[ndx].strand == -1 && nod[genes[i + 1].start_ndx].strand == 1) igm = intergenic_mod(nod[ndx], nod[genes[i + 1].start_ndx], t);
    if (i < ng - 1 && nod[ndx].strand == -1 && nod[genes[i + 1].start_ndx].strand == -1) igm = intergenic_mod(nod[ndx], nod[genes[i + 1].stop_ndx], t);
    int maxndx[2] = {-1, -1}; double maxsc[2] = {0, 0}, maxigm[2] = {0, 0};
    for (int j = ndx - 100; j < ndx + 100; ++j) {
      if (j < 0 || j >= nn || j == ndx) continue;
      if (nod[j].type == G_STOP || nod[j].stop_val != nod[ndx].stop_val) continue;
      double tigm = 0.0;
      if (i > 0 && nod[j].strand == 1 && nod[genes[i - 1].start_ndx].strand == 1) {
        if (nod[genes[i - 1].stop_ndx].ndx - nod[j].ndx > MAX_SAM_OVLP) continue;
        tigm = intergenic_mod(nod[genes[i - 1].stop_ndx], nod[j], t);
      }
      if (i > 0 && nod[j].strand == 1 && nod[genes[i - 1].start_ndx].strand == -1) {
        if (nod[genes[i - 1].start_ndx].ndx - nod[j].ndx >= 0) continue;
        tigm = intergenic_mod(nod[genes[i - 1].start_ndx], nod[j], t);
      }
      if (i < ng - 1 && nod[j].strand == -1 && nod[genes[i + 1].start_ndx].strand == 1) {
        if (nod[j].ndx - nod[genes[i + 1].start_ndx].ndx >= 0) continue;
        tigm = intergenic_mod(nod[j], nod[genes[i + 1].start_ndx], t);
      }
      if (i < ng - 1 && nod[j].strand == -1 && nod[genes[i + 1].start_ndx].strand == -1) {
        if (nod[j].ndx - nod[genes[i + 1].stop_ndx].ndx > MAX_SAM_OVLP) continue;
        tigm = intergenic_mod(nod[j], nod[genes[i + 1].stop_ndx], t);
      }
      const double v = nod[j].cscore + nod[j].sscore;
      if (maxndx[0] == -1) { maxndx[0] = j; maxsc[0] = v; maxigm[0] = tigm; }
      else if (v + tigm > maxsc[0] + maxigm[0]) { maxndx[1] = maxndx[0]; maxsc[1] = maxsc[0]; maxigm[1] = maxigm[0]; maxndx[0] = j; maxsc[0] = v; maxigm[0] = tigm; }
      else if (maxndx[1] == -1 || v + tigm > maxsc[1] + maxigm[1]) { maxndx[1] = j; maxsc[1] = v; maxigm[1] = tigm; }
    }
    for (int j = 0; j < 2; ++j) {
      const int m = maxndx[j];
      if (m == -1) continue;
      if (nod[m].tscore < nod[ndx].tscore && maxsc[j] - nod[m].tscore >= sc - nod[ndx].tscore + t.st_wt && nod[m].rscore > nod[ndx].rscore &&
          nod[m].uscore > nod[ndx].uscore && nod[m].cscore > nod[ndx].cscore && std::abs(nod[m].ndx - nod[ndx].ndx) > 15) {
        maxsc[j] += nod[ndx].tscore - nod[m].tscore;
      } else if (std::abs(nod[m].ndx - nod[ndx].ndx) <= 15 && nod[m].rscore + nod[m].tscore > nod[ndx].rscore + nod[ndx].tscore && nod[ndx].edge == 0 && nod[m].edge == 0) {
        if (nod[ndx].cscore > nod[m].cscore) maxsc[j] += nod[ndx].cscore - nod[m].cscore;
        if (nod[ndx].uscore > nod[m].uscore) maxsc[j] += nod[ndx].uscore - nod[m].uscore;
        if (igm > maxigm[j]) maxsc[j] += igm - maxigm[j];
      } else maxsc[j] = -1000.0;
    }
    int m = -1;
    for (int j = 0; j < 2; ++j) {
      if (maxndx[j] == -1) continue;
      if (m == -1 && maxsc[j] + maxigm[j] > sc + igm) m = j;
      else if (m >= 0 && maxsc[j] + maxigm[j] > maxsc[m] + maxigm[m]) m = j;
    }
    if (m != -1 && nod[maxndx[m]].strand == 1) { genes[i].start_ndx = maxndx[m]; genes[i].begin = nod[maxndx[m]].ndx + 1; }
    else if (m != -1 && nod[maxndx[m]].strand == -1) { genes[i].start_ndx = maxndx[m]; genes[i].end = nod[maxndx[m]].ndx + 1; }
  }
}
double confidence(double score, double st_wt) {
  double conf;
  if (score / st_wt < 41) { conf = exp(score / st_wt); conf = (conf / (conf + 1)) * 100.0; } else conf = 99.99;
  if (conf <= 50.00) conf = 50.00;
  return conf;
}
char amino(const GSeq &s, int strand, int i, int tt) {
  static const char *code = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF";
  if (s.unk(strand, i) || s.unk(strand, i + 1) || s.unk(strand, i + 2)) return 'X';
  const int c = s.at(strand, i) * 16 + s.at(strand, i + 1) * 4 + s.at(strand, i + 2);
  if (tt == 4 && c == 3 * 16 + 2 * 4 + 0) return 'W';
  return code[c];
}

// ---- node tables on the device ------------------------------------------------------------------------------------------------------
struct NodeTable {
  // host columns (padded per bin to a multiple of 256 with type 255)
  std::vector<uint32_t> bin, seq, dp_min; std::vector<int32_t> ndx, stop_val, star_ptr, traceb, ov_mark; std::vector<int8_t> strand; std::vector<uint8_t> type, edge, rbs0, rbs1;
  std::vector<double> cscore, gcb, csc, rscore, uscore, score;
  std::vector<uint32_t> seq_first;            // [nseq_used + 1] node ranges of the sequences that take part in the dynamic program
  std::vector<double> st_wt_of_seq;
  DevBuf d_bin, d_seq, d_dpmin, d_ndx, d_sv, d_star, d_tb, d_ov, d_strand, d_type, d_edge, d_rbs0, d_rbs1, d_cscore, d_gcb, d_csc, d_rs, d_us, d_score, d_first, d_stwt;
  size_t n() const { return ndx.size(); }
  template <class T> static void up(DevBuf &d, const std::vector<T> &v, hipStream_t st) { d.ensure(std::max<size_t>(64, v.size() * sizeof(T))); if (!v.empty()) HIPCHK(hipMemcpyAsync(d.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st)); }
  void upload_static(hipStream_t st) {
    up(d_bin, bin, st); up(d_seq, seq, st); up(d_ndx, ndx, st); up(d_sv, stop_val, st); up(d_strand, strand, st); up(d_type, type, st); up(d_edge, edge, st);
    d_cscore.ensure(std::max<size_t>(64, n() * 8)); d_rbs0.ensure(std::max<size_t>(64, n())); d_rbs1.ensure(std::max<size_t>(64, n()));
    HIPCHK(hipMemsetAsync(d_cscore.p, 0, std::max<size_t>(64, n() * 8), st)); HIPCHK(hipMemsetAsync(d_rbs0.p, 0, std::max<size_t>(64, n()), st)); HIPCHK(hipMemsetAsync(d_rbs1.p, 0, std::max<size_t>(64, n()), st));
  }
  void upload_dp(hipStream_t st) {
    score.assign(n(), 0.0); traceb.assign(n(), -1); ov_mark.assign(n(), -1);
    up(d_dpmin, dp_min, st); up(d_star, star_ptr, st); up(d_gcb, gcb, st); up(d_csc, csc, st); up(d_rs, rscore, st); up(d_us, uscore, st);
    up(d_score, score, st); up(d_tb, traceb, st); up(d_ov, ov_mark, st); up(d_first, seq_first, st); up(d_stwt, st_wt_of_seq, st);
  }
  GeneNodesDev dev() {
    GeneNodesDev d;
    d.bin = d_bin.as<uint32_t>(); d.seq = d_seq.as<uint32_t>(); d.ndx = d_ndx.as<int32_t>(); d.stop_val = d_sv.as<int32_t>(); d.strand = d_strand.as<int8_t>();
    d.type = d_type.as<uint8_t>(); d.edge = d_edge.as<uint8_t>(); d.cscore = d_cscore.as<double>(); d.rbs0 = d_rbs0.as<uint8_t>(); d.rbs1 = d_rbs1.as<uint8_t>();
    d.dp_min = d_dpmin.as<uint32_t>(); d.star_ptr = d_star.as<int32_t>(); d.gcb = d_gcb.as<double>(); d.csc = d_csc.as<double>(); d.rscore = d_rs.as<double>(); d.uscore = d_us.as<double>();
    d.score = d_score.as<double>(); d.traceb = d_tb.as<int32_t>(); d.ov_mark = d_ov.as<int32_t>();
    return d;
  }
};
}  // namespace
}  // namespace ckm
using namespace ckm;

struct ckm_genes {
  std::vector<uint32_t> bin, contig; std::vector<int32_t> begin, end, rbs_bin, mot_len, mot_ndx, mot_spacer; std::vector<int8_t> strand; std::vector<uint8_t> start_type, partial_left, partial_right;
  std::vector<double> gc_cont, conf, score, cscore, sscore, rscore, uscore, tscore;
  std::vector<uint64_t> prot_off; std::string prot;
  std::vector<uint8_t> bin_trained, bin_uses_sd; std::vector<double> bin_gc; std::vector<uint64_t> bin_bases, bin_coding, bin_nodes_train, bin_nodes_find;
  double ms_nodes = 0, ms_dp_train = 0, ms_score = 0, ms_dp_find = 0, ms_host = 0;
};

namespace {
struct SeqNodes { std::vector<GNode> nodes; };
}

extern "C" int ckm_genes_call(ckm_ctx *ctx, const char *text, const uint64_t *contig_off, uint32_t ncontigs, const uint32_t *bin_first, uint32_t nbins,
                              int trans_table, int closed, int mask_runs, ckm_genes **out) {
  return guarded([&] {
    if (!ctx || !text || !contig_off || !bin_first || !out) throw Error(CKM_EINVAL, "NULL argument");
    if (trans_table != 11 && trans_table != 4) throw Error(CKM_EINVAL, "translation table must be 11 or 4 (checkm/prodigal.py:86-93)");
    if (bin_first[nbins] != ncontigs) throw Error(CKM_EINVAL, "bin_first[nbins] must equal the number of contigs");
    *out = nullptr;
    ctx->settle();
    HIPCHK(hipSetDevice(ctx->device));
    Worker *w = &ctx->w[0];
    hipStream_t st = w->stream;
    const double t_begin = now_ms();
    // most of a call is ordered sweeps on host threads (a bin or a contig per task): a pool of its own for the call.  Measured on a
    // 256-thread host with two tables side by side (192 bins of 2 Mb, seconds for both): 12 threads 9.4, 16 8.9, 24 9.8, 32 10.1, 48 10.6 --
    // the sweeps are bound by memory, not by cores.  CKM_GENE_THREADS overrides.
    int gthreads = std::max(8, std::min(16, (int)std::thread::hardware_concurrency() / 8));
    if (const char *e = getenv("CKM_GENE_THREADS")) gthreads = std::max(1, std::min(128, atoi(e)));
    HostPool gpool(gthreads);
    auto prun = [&](size_t n, size_t chunk, const std::function<void(size_t, size_t)> &f) { gpool.run(n, chunk, f); };
    const bool tr_on = getenv("CKM_TRACE") != nullptr;
    auto tp = [&](const char *label) { if (tr_on) fprintf(stderr, "ckm-trace genes table %d %9.1f ms  %s\n", trans_table, now_ms() - t_begin, label); };
    if (tr_on) fprintf(stderr, "ckm-trace genes table %d: %d host threads (hardware_concurrency %u), %u bins, %u contigs\n", trans_table, gthreads, std::thread::hardware_concurrency(), nbins, ncontigs);
    static const char sep[13] = "TTAATTAATTAA";
    // ---- layout: one training sequence per bin (its contigs joined and, beyond one contig, closed by the separator); contigs are sub-ranges ----
    std::vector<uint64_t> seq_off; std::vector<int32_t> seq_len;          // sequence table: [0, nbins) training sequences, then the contigs
    std::vector<uint64_t> bin_total(nbins, 0);
    seq_off.resize((size_t)nbins + ncontigs); seq_len.resize((size_t)nbins + ncontigs);
    uint64_t pos = 0;
    for (uint32_t b = 0; b < nbins; ++b) {
      const uint32_t c0 = bin_first[b], c1 = bin_first[b + 1];
      const bool multi = c1 - c0 > 1;
      seq_off[b] = pos;
      uint64_t p = pos;
      for (uint32_t c = c0; c < c1; ++c) {
        const uint64_t n = contig_off[c + 1] - contig_off[c];
        if (n > 0x7ffffff0ull) throw Error(CKM_ERANGE, "contig longer than 2^31 bases");
        seq_off[nbins + c] = p; seq_len[nbins + c] = (int32_t)n; bin_total[b] += n;
        p += n + (multi ? 12 : 0);
      }
      if (p - pos > 0x7ffffff0ull) throw Error(CKM_ERANGE, "bin longer than 2^31 bases");
      seq_len[b] = (int32_t)(p - pos);
      pos = (p + 2 + 15) & ~(uint64_t)15;
    }
    const uint64_t body = (pos + 63) & ~(uint64_t)63;
    std::vector<uint8_t> ascii(64 + body + 128, (uint8_t)'N'), code(body + 64, (uint8_t)5);
    std::vector<uint64_t> bin_gc_count(nbins, 0);
    prun(nbins, 1, [&](size_t lo, size_t hi) {
      for (size_t b = lo; b < hi; ++b) {
        const uint32_t c0 = bin_first[b], c1 = bin_first[b + 1]; const bool multi = c1 - c0 > 1;
        uint64_t gcc = 0;
        for (uint32_t c = c0; c < c1; ++c) {
          const uint64_t o = seq_off[nbins + c]; const int n = seq_len[nbins + c];
          memcpy(ascii.data() + 64 + o, text + contig_off[c], (size_t)n);
          if (multi) memcpy(ascii.data() + 64 + o + n, sep, 12);
        }
        const uint64_t o = seq_off[b]; const int n = seq_len[b];
        for (int i = 0; i < n; ++i) {
          uint8_t v;
          switch (ascii[64 + o + i]) { case 'A': case 'a': v = 0; break; case 'C': case 'c': v = 1; gcc++; break; case 'G': case 'g': v = 2; gcc++; break;
                                        case 'T': case 't': case 'U': case 'u': v = 3; break; default: v = 5; }
          code[o + i] = v;
        }
        bin_gc_count[b] = gcc;
      }
    });
    tp("text packed");
    // ---- device: text, flags, nodes of the training sequences and of the contigs ----
    DevBuf d_ascii, d_flags, d_code, d_off, d_len, d_nodes, d_cnt;
    const uint32_t nseq = nbins + ncontigs;
    d_ascii.ensure(ascii.size()); d_flags.ensure(body + 256); d_code.ensure(code.size());
    d_off.ensure(std::max<size_t>(8, (size_t)nseq * 8)); d_len.ensure(std::max<size_t>(4, (size_t)nseq * 4)); d_cnt.ensure(16);
    HIPCHK(hipMemcpyAsync(d_ascii.p, ascii.data(), ascii.size(), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_code.p, code.data(), code.size(), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_off.p, seq_off.data(), (size_t)nseq * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_len.p, seq_len.data(), (size_t)nseq * 4, hipMemcpyHostToDevice, st));
    uint64_t bases = 0; for (uint32_t b = 0; b < nbins; ++b) bases += (uint64_t)seq_len[b];
    unsigned long long cap = std::max<unsigned long long>(1 << 16, bases / 2), n_all = 0;
    launch_orf_flags(st, d_ascii.as<uint8_t>() + 64, d_flags.as<uint8_t>(), body);
    std::vector<OrfNodeG> raw;
    for (int attempt = 0; attempt < 2; ++attempt) {
      d_nodes.ensure((size_t)cap * sizeof(OrfNodeG));
      HIPCHK(hipMemsetAsync(d_cnt.p, 0, 16, st));
      launch_orf_chain(st, d_flags.as<uint8_t>(), body / 64, d_off.as<uint64_t>(), d_len.as<int32_t>(), nseq, trans_table == 4 ? 1 : 0, closed ? 1 : 0, d_nodes.p, d_cnt.as<unsigned long long>(), cap);
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpyAsync(&n_all, d_cnt.p, 8, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      if (n_all <= cap) break;
      cap = n_all + 1024;
    }
    raw.resize((size_t)n_all);
    if (n_all) HIPCHK(hipMemcpy(raw.data(), d_nodes.p, (size_t)n_all * sizeof(OrfNodeG), hipMemcpyDeviceToHost));
    const double t_nodes = now_ms();
    tp("nodes on the host");
    // per sequence: nodes in working order (position, forward strand first), masks applied.  The device appends nodes in no particular order:
    // they are dealt to their sequences and sorted as the 16-byte records they arrive as (sorting the working structs, ~200 bytes each,
    // took 2 s of a 192-bin call).
    std::vector<SeqNodes> sn(nseq);
    auto gseq = [&](uint32_t s) { GSeq q; q.c = code.data() + seq_off[s]; q.slen = seq_len[s]; return q; };
    {
      std::vector<size_t> at(nseq + 1, 0);
      for (const OrfNodeG &x : raw) at[x.contig + 1]++;
      for (uint32_t s = 0; s < nseq; ++s) at[s + 1] += at[s];
      std::vector<OrfNodeG> flat(raw.size());
      { std::vector<size_t> cur(at.begin(), at.end() - 1); for (const OrfNodeG &x : raw) flat[cur[x.contig]++] = x; }
      raw.clear(); raw.shrink_to_fit();
      prun(nseq, 1, [&](size_t lo, size_t hi) {
        for (size_t s = lo; s < hi; ++s) {
          OrfNodeG *f0 = flat.data() + at[s], *f1 = flat.data() + at[s + 1];
          std::sort(f0, f1, [](const OrfNodeG &x, const OrfNodeG &y) { return x.ndx != y.ndx ? x.ndx < y.ndx : x.strand_rev < y.strand_rev; });
          std::vector<GNode> &nod = sn[s].nodes; nod.resize((size_t)(f1 - f0));
          for (size_t k = 0; k < nod.size(); ++k) { const OrfNodeG &x = f0[k]; GNode &g = nod[k]; g.type = x.type; g.edge = x.edge; g.ndx = x.ndx; g.strand = x.strand_rev ? -1 : 1; g.stop_val = x.stop_val; }
          std::vector<Mask> m; if (mask_runs) m = find_masks(gseq((uint32_t)s));
          finish_nodes(nod, m);
        }
      });
    }
    tp("nodes sorted, masks applied");
    // ---- training ----
    std::vector<GTrain> tr(nbins);
    std::vector<uint8_t> trained(nbins, 0);
    for (uint32_t b = 0; b < nbins; ++b) {
      tr[b].trans_table = trans_table; tr[b].gc = seq_len[b] ? (double)bin_gc_count[b] / (double)seq_len[b] : 0.0;
      trained[b] = bin_total[b] >= 20000 ? 1 : 0;                         // (prodigal refuses to train on less; CheckM switches to -p meta below 100 kb, which is not built)
    }
    prun(nbins, 1, [&](size_t lo, size_t hi) {
      for (size_t b = lo; b < hi; ++b) {
        if (!trained[b]) continue;
        const GSeq q = gseq((uint32_t)b);
        const std::vector<int> gcf = calc_most_gc_frame(q);
        record_gc_bias(gcf, sn[b].nodes, tr[b]);
        record_overlapping_starts(sn[b].nodes, tr[b], 0);
      }
    });
    tp("gc frames, overlapping starts");
    GeneSeqDev sd; sd.txt = d_code.as<uint8_t>(); sd.off = d_off.as<uint64_t>(); sd.len = d_len.as<int32_t>();
    auto build_table = [&](NodeTable &T, bool training, int flag) {
      // training: sequences [0, nbins) of trained bins; else: contigs of trained bins.  Static columns + the dynamic program's inputs.
      // The layout first (a bin's sequences one behind the other, the bin padded to a multiple of 256 nodes), then the columns, a sequence per task.
      struct Job { uint32_t b, s; size_t first; };
      struct Pad { size_t from, to; uint32_t b, s; };
      std::vector<Job> jobs; std::vector<Pad> pads;
      size_t k = 0;
      for (uint32_t b = 0; b < nbins; ++b) {
        if (!trained[b]) continue;
        const uint32_t s0 = training ? b : nbins + bin_first[b], s1 = training ? b + 1 : nbins + bin_first[b + 1];
        for (uint32_t s = s0; s < s1; ++s) {
          T.seq_first.push_back((uint32_t)k); T.st_wt_of_seq.push_back(tr[b].st_wt);
          jobs.push_back({b, s, k}); k += sn[s].nodes.size();
        }
        T.seq_first.push_back((uint32_t)k);              // (closes the bin's last sequence; the next bin opens a new entry: ranges with gaps for the padding)
        T.st_wt_of_seq.push_back(tr[b].st_wt);
        const size_t k2 = (k + 255) & ~(size_t)255;
        pads.push_back({k, k2, b, s0}); k = k2;
      }
      if (k > 0xfffffff0ull) throw Error(CKM_ERANGE, "more than 2^32 nodes in one gene-calling batch");
      T.bin.resize(k); T.seq.resize(k); T.ndx.resize(k); T.stop_val.resize(k); T.strand.resize(k); T.type.resize(k); T.edge.resize(k);
      prun(jobs.size(), 1, [&](size_t lo, size_t hi) {
        for (size_t j = lo; j < hi; ++j) {
          const Job &jb = jobs[j]; const std::vector<GNode> &nod = sn[jb.s].nodes;
          for (size_t i = 0; i < nod.size(); ++i) {
            const GNode &g = nod[i]; const size_t x = jb.first + i;
            T.bin[x] = jb.b; T.seq[x] = jb.s; T.ndx[x] = g.ndx; T.stop_val[x] = g.stop_val; T.strand[x] = (int8_t)g.strand; T.type[x] = (uint8_t)g.type; T.edge[x] = (uint8_t)g.edge;
          }
        }
      });
      for (const Pad &pd : pads)
        for (size_t x = pd.from; x < pd.to; ++x) { T.bin[x] = pd.b; T.seq[x] = pd.s; T.ndx[x] = 0; T.stop_val[x] = 0; T.strand[x] = 1; T.type[x] = 255; T.edge[x] = 0; }
      (void)flag;
    };
    // NOTE on seq_first: entries come in (first, ..., first, END) groups per bin; the kernel treats [seq_first[k], seq_first[k+1]) as a sequence, so the END -> next bin's first
    // range holds only padding nodes (type 255), which the kernel skips.
    auto fill_dp = [&](NodeTable &T, bool training, int flag) {
      T.dp_min.assign(T.n(), 0); T.star_ptr.assign(T.n() * 3, -1); T.gcb.assign(T.n(), 0.0); T.csc.assign(T.n(), 0.0); T.rscore.assign(T.n(), 0.0); T.uscore.assign(T.n(), 0.0);
      std::vector<std::pair<uint32_t, size_t>> jobs;        // (sequence, first node)
      size_t k = 0;
      for (uint32_t b = 0; b < nbins; ++b) {
        if (!trained[b]) continue;
        const uint32_t s0 = training ? b : nbins + bin_first[b], s1 = training ? b + 1 : nbins + bin_first[b + 1];
        for (uint32_t s = s0; s < s1; ++s) { jobs.push_back({s, k}); k += sn[s].nodes.size(); }
        k = (k + 255) & ~(size_t)255;
      }
      prun(jobs.size(), 1, [&](size_t lo, size_t hi) {
        for (size_t j = lo; j < hi; ++j) {
          const uint32_t s = jobs[j].first; const size_t f = jobs[j].second;
          const std::vector<GNode> &nod = sn[s].nodes;
          const uint32_t b = T.bin[f < T.n() ? f : 0];
          std::vector<uint32_t> mn; dp_window(nod, mn);
          for (size_t i = 0; i < nod.size(); ++i) {
            T.dp_min[f + i] = mn[i];
            for (int q = 0; q < 3; ++q) T.star_ptr[(f + i) * 3 + q] = nod[i].star_ptr[q];
            if (flag == 0) T.gcb[f + i] = tr[b].bias[0] * nod[i].gc_score[0] + tr[b].bias[1] * nod[i].gc_score[1] + tr[b].bias[2] * nod[i].gc_score[2];
            else { T.csc[f + i] = nod[i].cscore + nod[i].sscore; T.rscore[f + i] = nod[i].rscore; T.uscore[f + i] = nod[i].uscore; }
          }
        }
      });
      return jobs;
    };
    auto read_dp = [&](NodeTable &T, const std::vector<std::pair<uint32_t, size_t>> &jobs) {
      HIPCHK(hipMemcpyAsync(T.score.data(), T.d_score.p, T.n() * 8, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(T.traceb.data(), T.d_tb.p, T.n() * 4, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(T.ov_mark.data(), T.d_ov.p, T.n() * 4, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      prun(jobs.size(), 1, [&](size_t lo, size_t hi) {
        for (size_t j = lo; j < hi; ++j) {
          std::vector<GNode> &nod = sn[jobs[j].first].nodes; const size_t f = jobs[j].second;
          for (size_t i = 0; i < nod.size(); ++i) { nod[i].score = T.score[f + i]; nod[i].traceb = T.traceb[f + i] < 0 ? -1 : T.traceb[f + i] - (int32_t)f; nod[i].ov_mark = T.ov_mark[f + i]; nod[i].tracef = -1; }
        }
      });
    };
    auto run_dp = [&](NodeTable &T, int flag, double &ms) {
      hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
      HIPCHK(hipEventRecord(e0, st));
      launch_gene_dp(st, T.dev(), T.d_first.as<uint32_t>(), T.d_stwt.as<double>(), (uint32_t)T.seq_first.size() - 1, flag);
      HIPCHK(hipEventRecord(e1, st)); HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(st));
      float t = 0.f; HIPCHK(hipEventElapsedTime(&t, e0, e1)); ms += t;
      (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    };
    std::unique_ptr<ckm_genes> o(new ckm_genes());
    NodeTable TT;
    build_table(TT, true, 0);
    std::vector<int> ipath(nbins, -1);
    if (TT.n()) {
      const auto jobs = fill_dp(TT, true, 0);
      tp("training table built");
      TT.upload_static(st); TT.upload_dp(st);
      run_dp(TT, 0, o->ms_dp_train);
      tp("training dp done");
      read_dp(TT, jobs);
      tp("training dp read back");
      // hexamer statistics of the first gene set; their sums and the SD bins per start node
      std::vector<double> dc_all((size_t)nbins * 4096, 0.0), rw_all((size_t)nbins * 28, 0.0);
      prun(nbins, 1, [&](size_t lo, size_t hi) {
        for (size_t b = lo; b < hi; ++b) {
          if (!trained[b]) continue;
          ipath[b] = dprog_finish(sn[b].nodes);
          calc_dicodon_gene(tr[b], gseq((uint32_t)b), sn[b].nodes, ipath[b]);
          memcpy(dc_all.data() + b * 4096, tr[b].gene_dc, sizeof(double) * 4096);
        }
      });
      tp("hexamer statistics");
      DevBuf d_dc, d_rw;
      d_dc.ensure(dc_all.size() * 8); d_rw.ensure(rw_all.size() * 8);
      HIPCHK(hipMemcpyAsync(d_dc.p, dc_all.data(), dc_all.size() * 8, hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(d_rw.p, rw_all.data(), rw_all.size() * 8, hipMemcpyHostToDevice, st));
      hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
      HIPCHK(hipEventRecord(e0, st));
      launch_gene_cscore(st, sd, TT.dev(), d_dc.as<double>(), (uint32_t)TT.n());
      launch_gene_rbs(st, sd, TT.dev(), d_rw.as<double>(), (uint32_t)TT.n());
      HIPCHK(hipEventRecord(e1, st));
      TT.cscore.resize(TT.n()); TT.rbs0.resize(TT.n()); TT.rbs1.resize(TT.n());
      HIPCHK(hipMemcpyAsync(TT.cscore.data(), TT.d_cscore.p, TT.n() * 8, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(TT.rbs0.data(), TT.d_rbs0.p, TT.n(), hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(TT.rbs1.data(), TT.d_rbs1.p, TT.n(), hipMemcpyDeviceToHost, st));
      HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(st));
      { float t = 0.f; HIPCHK(hipEventElapsedTime(&t, e0, e1)); o->ms_score += t; (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); }
      prun(jobs.size(), 1, [&](size_t lo, size_t hi) {
        for (size_t j = lo; j < hi; ++j) {
          const uint32_t b = jobs[j].first; const size_t f = jobs[j].second;
          std::vector<GNode> &nod = sn[b].nodes;
          for (size_t i = 0; i < nod.size(); ++i) { nod[i].cscore = TT.cscore[f + i]; nod[i].rbs[0] = TT.rbs0[f + i]; nod[i].rbs[1] = TT.rbs1[f + i]; }
          const GSeq q = gseq(b);
          coding_score_passes(nod, tr[b]);
          train_starts_sd(q, nod, tr[b]);
          determine_sd_usage(tr[b]);
          if (tr[b].uses_sd == 0) train_starts_nonsd(q, nod, tr[b]);
          std::vector<GNode>().swap(nod);            // the training nodes are done
        }
      });
    }
    tp("start-site training done");
    // ---- gene finding, contig by contig ----
    NodeTable TF;
    build_table(TF, false, 1);
    tp("contig table built");
    std::vector<std::vector<GeneRec>> genes_of(nseq);
    if (TF.n()) {
      std::vector<double> dc_all((size_t)nbins * 4096, 0.0), rw_all((size_t)nbins * 28, 0.0);
      for (uint32_t b = 0; b < nbins; ++b) if (trained[b]) { memcpy(dc_all.data() + (size_t)b * 4096, tr[b].gene_dc, sizeof(double) * 4096); memcpy(rw_all.data() + (size_t)b * 28, tr[b].rbs_wt, sizeof(double) * 28); }
      DevBuf d_dc, d_rw;
      d_dc.ensure(dc_all.size() * 8); d_rw.ensure(rw_all.size() * 8);
      HIPCHK(hipMemcpyAsync(d_dc.p, dc_all.data(), dc_all.size() * 8, hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(d_rw.p, rw_all.data(), rw_all.size() * 8, hipMemcpyHostToDevice, st));
      TF.upload_static(st);
      hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
      HIPCHK(hipEventRecord(e0, st));
      launch_gene_cscore(st, sd, TF.dev(), d_dc.as<double>(), (uint32_t)TF.n());
      launch_gene_rbs(st, sd, TF.dev(), d_rw.as<double>(), (uint32_t)TF.n());
      HIPCHK(hipEventRecord(e1, st));
      TF.cscore.resize(TF.n()); TF.rbs0.resize(TF.n()); TF.rbs1.resize(TF.n());
      HIPCHK(hipMemcpyAsync(TF.cscore.data(), TF.d_cscore.p, TF.n() * 8, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(TF.rbs0.data(), TF.d_rbs0.p, TF.n(), hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(TF.rbs1.data(), TF.d_rbs1.p, TF.n(), hipMemcpyDeviceToHost, st));
      HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(st));
      { float t = 0.f; HIPCHK(hipEventElapsedTime(&t, e0, e1)); o->ms_score += t; (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); }
      tp("contig coding sums and SD bins from the device");
      // node scores, overlapping starts; then the dynamic program's inputs
      {
        std::vector<std::pair<uint32_t, size_t>> jobs0; size_t k = 0;
        for (uint32_t b = 0; b < nbins; ++b) {
          if (!trained[b]) continue;
          for (uint32_t s = nbins + bin_first[b]; s < nbins + bin_first[b + 1]; ++s) { jobs0.push_back({s, k}); k += sn[s].nodes.size(); }
          k = (k + 255) & ~(size_t)255;
        }
        prun(jobs0.size(), 1, [&](size_t lo, size_t hi) {
          for (size_t j = lo; j < hi; ++j) {
            const uint32_t s = jobs0[j].first; const size_t f = jobs0[j].second;
            std::vector<GNode> &nod = sn[s].nodes;
            const uint32_t b = TF.bin[f < TF.n() ? f : 0];
            // (score_nodes looks for Shine-Dalgarno sites only in organisms that use them: elsewhere every node keeps bin 0)
            const bool sdm = tr[b].uses_sd == 1;
            for (size_t i = 0; i < nod.size(); ++i) { nod[i].cscore = TF.cscore[f + i]; nod[i].rbs[0] = sdm ? TF.rbs0[f + i] : 0; nod[i].rbs[1] = sdm ? TF.rbs1[f + i] : 0; }
            score_nodes_rest(gseq(s), nod, tr[b], closed);
            record_overlapping_starts(nod, tr[b], 1);
            for (size_t i = 0; i < nod.size(); ++i) TF.edge[f + i] = (uint8_t)nod[i].edge;
          }
        });
        NodeTable::up(TF.d_edge, TF.edge, st);
      }
      // (score_nodes turns starts at the sequence edges into edge nodes: that static column changes)
      tp("node scores");
      const auto jobs = fill_dp(TF, false, 1);
      TF.upload_dp(st);
      tp("final dp inputs up");
      run_dp(TF, 1, o->ms_dp_find);
      tp("final dp done");
      read_dp(TF, jobs);
      prun(jobs.size(), 1, [&](size_t lo, size_t hi) {
        for (size_t j = lo; j < hi; ++j) {
          const uint32_t s = jobs[j].first; const size_t f = jobs[j].second;
          std::vector<GNode> &nod = sn[s].nodes;
          const uint32_t b = TF.bin[f < TF.n() ? f : 0];
          const int ip = dprog_finish(nod);
          eliminate_bad_genes(nod, ip, tr[b]);
          genes_of[s] = add_genes(nod, ip);
          tweak_final_starts(genes_of[s], nod, tr[b]);
        }
      });
    }
    tp("genes picked, starts tweaked");
    // ---- records ----
    o->bin_trained.assign(trained.begin(), trained.end()); o->bin_uses_sd.resize(nbins); o->bin_gc.resize(nbins); o->bin_bases.assign(bin_total.begin(), bin_total.end());
    o->bin_coding.assign(nbins, 0); o->bin_nodes_train.assign(nbins, 0); o->bin_nodes_find.assign(nbins, 0);
    // (a bin per task into a record set of its own -- the translations are most of the work -- then appended in bin order)
    std::vector<ckm_genes> part(nbins);
    prun(nbins, 1, [&](size_t blo, size_t bhi) {
    for (uint32_t b = (uint32_t)blo; b < (uint32_t)bhi; ++b) {
      ckm_genes *o = &part[b];
      o->bin_coding.assign(1, 0);                                // (of this bin)
      for (uint32_t c = bin_first[b]; c < bin_first[b + 1]; ++c) {
        const uint32_t s = nbins + c;
        const std::vector<GNode> &nod = sn[s].nodes; const GTrain &t = tr[b]; const GSeq q = gseq(s);
        for (const GeneRec &g : genes_of[s]) {
          const GNode &sn_ = nod[g.start_ndx], &sp = nod[g.stop_ndx];
          o->bin.push_back(b); o->contig.push_back(c); o->begin.push_back(g.begin); o->end.push_back(g.end); o->strand.push_back((int8_t)sn_.strand);
          o->start_type.push_back((uint8_t)(sn_.edge ? 3 : sn_.type));
          o->partial_left.push_back((uint8_t)(sn_.strand == 1 ? sn_.edge : sp.edge)); o->partial_right.push_back((uint8_t)(sn_.strand == 1 ? sp.edge : sn_.edge));
          const double rbs1 = t.rbs_wt[sn_.rbs[0]] * t.st_wt, rbs2 = t.rbs_wt[sn_.rbs[1]] * t.st_wt;
          int rb = -1, ml = 0, mx = 0, ms = 0;
          if (t.uses_sd == 1) rb = rbs1 > rbs2 ? sn_.rbs[0] : sn_.rbs[1];
          else if (t.no_mot > -0.5 && rbs1 > rbs2 && rbs1 > sn_.mot_score * t.st_wt) rb = sn_.rbs[0];
          else if (t.no_mot > -0.5 && rbs2 >= rbs1 && rbs2 > sn_.mot_score * t.st_wt) rb = sn_.rbs[1];
          else { ml = sn_.mot_len; mx = sn_.mot_ndx; ms = sn_.mot_spacer; }
          o->rbs_bin.push_back(rb); o->mot_len.push_back(ml); o->mot_ndx.push_back(mx); o->mot_spacer.push_back(ms);
          o->gc_cont.push_back(sn_.gc_cont); o->cscore.push_back(sn_.cscore); o->sscore.push_back(sn_.sscore); o->rscore.push_back(sn_.rscore); o->uscore.push_back(sn_.uscore); o->tscore.push_back(sn_.tscore);
          o->score.push_back(sn_.cscore + sn_.sscore); o->conf.push_back(confidence(sn_.cscore + sn_.sscore, t.st_wt));
          // protein
          const int slen = q.slen, strand = sn_.strand;
          const int pb = strand == 1 ? g.begin - 1 : slen - g.end, pe = strand == 1 ? g.end - 1 : slen - g.begin;
          const bool partial5 = strand == 1 ? o->partial_left.back() : o->partial_right.back();
          o->prot_off.push_back(o->prot.size());
          for (int i = pb; i + 2 <= pe; i += 3) { char a = amino(q, strand, i, trans_table); if (i == pb && !partial5) a = 'M'; o->prot.push_back(a); }
          o->bin_coding[0] += (uint64_t)(g.end - g.begin + 1);
        }
      }
    }
    });
    auto app = [](auto &dst, const auto &src) { dst.insert(dst.end(), src.begin(), src.end()); };
    for (uint32_t b = 0; b < nbins; ++b) {
      o->bin_uses_sd[b] = (uint8_t)tr[b].uses_sd; o->bin_gc[b] = tr[b].gc;
      for (uint32_t c = bin_first[b]; c < bin_first[b + 1]; ++c) o->bin_nodes_find[b] += sn[nbins + c].nodes.size();
      const ckm_genes &q = part[b];
      o->bin_coding[b] = q.bin_coding[0];
      const uint64_t p0 = o->prot.size();
      app(o->bin, q.bin); app(o->contig, q.contig); app(o->begin, q.begin); app(o->end, q.end); app(o->strand, q.strand); app(o->start_type, q.start_type);
      app(o->partial_left, q.partial_left); app(o->partial_right, q.partial_right); app(o->rbs_bin, q.rbs_bin); app(o->mot_len, q.mot_len); app(o->mot_ndx, q.mot_ndx);
      app(o->mot_spacer, q.mot_spacer); app(o->gc_cont, q.gc_cont); app(o->cscore, q.cscore); app(o->sscore, q.sscore); app(o->rscore, q.rscore); app(o->uscore, q.uscore);
      app(o->tscore, q.tscore); app(o->score, q.score); app(o->conf, q.conf);
      for (uint64_t x : q.prot_off) o->prot_off.push_back(p0 + x);
      o->prot.append(q.prot);
      part[b] = ckm_genes();
    }
    o->prot_off.push_back(o->prot.size());
    tp("records and proteins");
    o->ms_nodes = t_nodes - t_begin; o->ms_host = now_ms() - t_begin;
    *out = o.release();
  });
}

extern "C" int ckm_genes_columns_get(const ckm_genes *g, ckm_genes_columns *c) {
  if (!g || !c) { set_last_error("NULL argument"); return CKM_EINVAL; }
  c->n = g->begin.size(); c->bin = g->bin.data(); c->contig = g->contig.data(); c->begin = g->begin.data(); c->end = g->end.data(); c->strand = g->strand.data();
  c->start_type = g->start_type.data(); c->partial_left = g->partial_left.data(); c->partial_right = g->partial_right.data();
  c->rbs_bin = g->rbs_bin.data(); c->mot_len = g->mot_len.data(); c->mot_ndx = g->mot_ndx.data(); c->mot_spacer = g->mot_spacer.data();
  c->gc_cont = g->gc_cont.data(); c->conf = g->conf.data(); c->score = g->score.data(); c->cscore = g->cscore.data(); c->sscore = g->sscore.data();
  c->rscore = g->rscore.data(); c->uscore = g->uscore.data(); c->tscore = g->tscore.data();
  c->prot_off = g->prot_off.data(); c->prot = g->prot.data();
  c->nbins = g->bin_trained.size(); c->bin_trained = g->bin_trained.data(); c->bin_uses_sd = g->bin_uses_sd.data(); c->bin_gc = g->bin_gc.data();
  c->bin_bases = g->bin_bases.data(); c->bin_coding = g->bin_coding.data(); c->bin_nodes = g->bin_nodes_find.data();
  c->ms_nodes = g->ms_nodes; c->ms_dp_train = g->ms_dp_train; c->ms_score = g->ms_score; c->ms_dp_find = g->ms_dp_find; c->ms_total = g->ms_host;
  return CKM_OK;
}
extern "C" void ckm_genes_free(ckm_genes *g) { delete g; }
