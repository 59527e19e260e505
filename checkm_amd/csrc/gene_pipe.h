// gene_pipe.h -- the gene finder's pipeline, nodes resident on the device from the codon flags to the gene records.
// What it replaces: the two `prodigal -p single -m -f gff -g <11|4> -a genes.faa` runs per bin of checkm/prodigal.py:80-133.
//
// One call = one translation table for a batch of bins.  Stages (every "map" is one thread per index; x_* are the cooperating kernels):
//   text      contigs laid out as one training sequence per bin (joined by TTAATTAATTAA), one-byte base codes, bit planes of G/C and of
//             unknown bases, starts of 50-runs of unknown bases (the -m masks), codon flags (kernels_orf.hip)
//   nodes     x_chain finds the start / stop nodes of all six frames of the training sequences AND of the contigs; a node's place in the
//             working order (position, forward strand first) is its RANK in two bit planes (node on the forward / reverse strand at this
//             base), so the records are scattered into sorted structure-of-arrays columns without a sort; each (sequence, strand, frame)
//             chain also keeps its nodes in chain order, an open reading frame being a run [starts, inner to outer][its stop]
//   training  GC-frame plot as range counts over "winning frame of a codon triple" planes, ordered bias sums, overlapping starts, the
//             dynamic program, trace-back walks (a thread per bin), hexamer statistics by atomics; the logarithms of every table are taken
//             on the HOST (libm, as the oracle takes them): counts come down, weights go up -- 10 round trips for the Shine-Dalgarno
//             model, 20 more for bins that do not use it
//   genes     hexamer sums, SD bins, GC content, coding-score passes per reading frame, upstream motifs, start scores, overlapping starts, the
//             dynamic program, trace-back, bad-gene elimination, start tweaks (a thread per contig), records and translations
// The arithmetic follows oracle/gene_full.c operation by operation; integer counts replace the oracle's `+= 1.0` sums (exact either way).
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <functional>
#include <string>
#include "gene_exec.h"

namespace ckm {
namespace gene {

struct PipeInput {
  const char *text; const uint64_t *contig_off; uint32_t ncontigs; const uint32_t *bin_first; uint32_t nbins; int trans_table; int mask_runs;
  std::function<void(size_t, const std::function<void(size_t)> &)> pfor;      // host-parallel loop over bins (table arithmetic)
  std::function<void(const char *)> trace;
};
struct GeneResult {
  std::vector<uint32_t> bin, contig; std::vector<int32_t> begin, end, rbs_bin, mot_len, mot_ndx, mot_spacer; std::vector<int8_t> strand; std::vector<uint8_t> start_type, partial_left, partial_right;
  std::vector<double> gc_cont, conf, score, cscore, sscore, rscore, uscore, tscore;
  std::vector<uint64_t> prot_off; std::string prot;
  std::vector<uint8_t> bin_trained, bin_uses_sd; std::vector<double> bin_gc; std::vector<uint64_t> bin_bases, bin_coding, bin_nodes_train, bin_nodes_find;
  double ms_dp_train = 0, ms_score = 0, ms_dp_find = 0;
};

struct OrfRec { uint32_t seq; int32_t ndx, sv; uint8_t type, strand_rev, edge, pad; };

// host copy of a bin's training results
struct GTrainH {
  double gc = 0; int trans_table = 11; double st_wt = 4.35; double bias[3] = {0, 0, 0}; double type_wt[3] = {0, 0, 0}; int uses_sd = 0;
  double rbs_wt[28] = {0}; double ups_comp[32][4] = {{0}}; std::vector<double> mot_wt; double no_mot = 0; double gene_dc[4096] = {0};
};

// device arrays of one node set
struct NodeSet {
  size_t n = 0;
  GBuf b_bin, b_seq, b_ndx, b_sv, b_strand, b_type, b_edge, b_edge2, b_chx, b_ctr, b_cls, b_term, b_cscore, b_rbs0, b_rbs1, b_dpmin, b_star, b_gcb, b_csc, b_rs, b_us, b_ts, b_ss, b_gcc, b_ms, b_mot,
      b_upw, b_score, b_tb, b_tf, b_ov, b_elim;
  void alloc(GExec &e, size_t nodes, bool training) {
    n = nodes;
    const size_t m = std::max<size_t>(nodes, 64);
    b_bin.ensure(m * 4); b_seq.ensure(m * 4); b_ndx.ensure(m * 4); b_sv.ensure(m * 4); b_strand.ensure(m); b_type.ensure(m); b_edge.ensure(m); b_edge2.ensure(m); b_chx.ensure(m * 4);
    b_cscore.ensure(m * 8); b_rbs0.ensure(m); b_rbs1.ensure(m); b_dpmin.ensure(m * 4); b_star.ensure(m * 12); b_score.ensure(m * 8); b_tb.ensure(m * 4); b_tf.ensure(m * 4); b_ov.ensure(m * 4);
    b_mot.ensure(m * 4); b_ms.ensure(m * 8); b_upw.ensure(m * 8);
    if (training) { b_ctr.ensure(m * 12); b_cls.ensure(m); b_term.ensure(m * 8); b_gcb.ensure(m * 8); }
    else { b_csc.ensure(m * 8); b_rs.ensure(m * 8); b_us.ensure(m * 8); b_ts.ensure(m * 8); b_ss.ensure(m * 8); b_gcc.ensure(m * 8); b_elim.ensure(m); }
    g_zero(e, b_type.p, 0xff, m); g_zero(e, b_edge.p, 0, m); g_zero(e, b_cscore.p, 0, m * 8); g_zero(e, b_rbs0.p, 0, m); g_zero(e, b_rbs1.p, 0, m);
    g_zero(e, b_score.p, 0, m * 8); g_zero(e, b_tb.p, 0xff, m * 4); g_zero(e, b_tf.p, 0xff, m * 4); g_zero(e, b_ov.p, 0xff, m * 4); g_zero(e, b_star.p, 0xff, m * 12);
    g_zero(e, b_mot.p, 0, m * 4); g_zero(e, b_ms.p, 0, m * 8); g_zero(e, b_upw.p, 0, m * 8);
    if (!training) { g_zero(e, b_rs.p, 0, m * 8); g_zero(e, b_us.p, 0, m * 8); g_zero(e, b_ts.p, 0, m * 8); g_zero(e, b_ss.p, 0, m * 8); g_zero(e, b_gcc.p, 0, m * 8); g_zero(e, b_elim.p, 0, m); g_zero(e, b_csc.p, 0, m * 8); }
    else { g_zero(e, b_gcb.p, 0, m * 8); g_zero(e, b_ctr.p, 0, m * 12); g_zero(e, b_cls.p, 0, m); g_zero(e, b_term.p, 0, m * 8); }
  }
  Nodes dev() const {
    Nodes d;
    d.bin = b_bin.as<uint32_t>(); d.seq = b_seq.as<uint32_t>(); d.ndx = b_ndx.as<int32_t>(); d.sv = b_sv.as<int32_t>(); d.strand = b_strand.as<int8_t>(); d.type = b_type.as<uint8_t>(); d.edge = b_edge.as<uint8_t>();
    d.chx = b_chx.as<uint32_t>(); d.ctr = b_ctr.as<int32_t>(); d.gcb_cls = b_cls.as<uint8_t>(); d.gcb_term = b_term.as<double>(); d.cscore = b_cscore.as<double>(); d.rbs0 = b_rbs0.as<uint8_t>(); d.rbs1 = b_rbs1.as<uint8_t>();
    d.dp_min = b_dpmin.as<uint32_t>(); d.star = b_star.as<int32_t>(); d.gcb = b_gcb.as<double>(); d.csc = b_csc.as<double>(); d.rscore = b_rs.as<double>(); d.uscore = b_us.as<double>(); d.tscore = b_ts.as<double>();
    d.sscore = b_ss.as<double>(); d.gc_cont = b_gcc.as<double>(); d.mot_score = b_ms.as<double>(); d.mot = b_mot.as<uint32_t>(); d.upw = b_upw.as<unsigned long long>(); d.score = b_score.as<double>();
    d.traceb = b_tb.as<int32_t>(); d.tracef = b_tf.as<int32_t>(); d.ov_mark = b_ov.as<int32_t>(); d.elim = b_elim.as<uint8_t>();
    return d;
  }
};

// ---- ordered walks over one sequence's nodes (relative indices; dprog.c / gene.c), a thread per sequence ----
// max_ndx: where the path ends (x_path_ends: the scan over all of the sequence's nodes is a kernel of its own)
// Every hop of these walks is a load that depends on the last one (0.7 - 2 us on the device): a step therefore asks for everything it
// needs of the next node -- its link included -- at once, carries it into the next step in registers, and reads memory again only after the
// rare step that rewires the path.  (Written as dprog.c has it, `while (traceb[path] != -1) { nxt = traceb[path]; ...; path = traceb[path]; }`,
// a step was three round trips: the stores of the rare branch keep the compiler from holding a link across them.)
struct WalkNode { int tb, strand, ndx, sv, mark; bool stop; };
GFN WalkNode walk_node(const NView &V, int i) {
  const Nodes &n = V.n; const uint32_t g = V.lo + (uint32_t)i;
  WalkNode w; w.tb = n.traceb[g]; w.strand = n.strand[g]; w.stop = n.type[g] == G_STOP; w.ndx = n.ndx[g]; w.sv = n.sv[g]; w.mark = n.ov_mark[g];
  return w;
}
// visit(i, node): every node of the finished path, from its end to its beginning (what a walk along traceb from the returned index meets)
template <class F>
GFN int walk_dprog_finish(const NView &V, int max_ndx, F visit) {
  const Nodes &n = V.n; const uint32_t lo = V.lo; const int nn = V.nn;
  if (nn == 0) return -1;
  if (max_ndx < 0) return -1;
  if (n.traceb[lo + max_ndx] == -1) return -1;                     // (a path of one node: nothing to untangle, no gene)
  // first pass: the triple overlaps
  int path = max_ndx;
  WalkNode a = walk_node(V, path);
  while (a.tb != -1) {
    const int nxt = a.tb;
    const WalkNode b = walk_node(V, nxt);
    if (a.strand == -1 && a.stop && b.strand == 1 && b.stop && a.mark != -1 && a.ndx > b.ndx) {
      const int tmp = n.star[(size_t)(lo + path) * 3 + a.mark];
      int i;
      for (i = tmp; V.ndx(i) != V.sv(tmp); --i);
      n.traceb[lo + path] = tmp; n.traceb[lo + tmp] = i; n.ov_mark[lo + i] = -1; n.traceb[lo + i] = nxt;
      path = tmp; a = walk_node(V, path);
      continue;
    }
    path = nxt; a = b;
  }
  // second pass: the double overlaps; a node's link is final when the pass leaves it, so the forward pointers (dprog.c's third pass) and
  // the caller's visit ride along
  path = max_ndx;
  a = walk_node(V, path);
  while (a.tb != -1) {
    const int nxt = a.tb;
    const WalkNode b = walk_node(V, nxt);
    bool rewired = false;
    if (a.strand == -1 && !a.stop && b.strand == 1 && b.stop) {
      int i;
      for (i = path; V.ndx(i) != a.sv; --i);
      n.traceb[lo + path] = i; n.traceb[lo + i] = nxt;
      rewired = true;
    }
    if (a.strand == 1 && a.stop && b.strand == 1 && b.stop) {
      n.traceb[lo + path] = n.star[(size_t)(lo + nxt) * 3 + a.ndx % 3];
      n.traceb[lo + n.traceb[lo + path]] = nxt;
      rewired = true;
    }
    if (a.strand == -1 && a.stop && b.strand == -1 && b.stop) {
      n.traceb[lo + path] = n.star[(size_t)(lo + path) * 3 + b.ndx % 3];
      n.traceb[lo + n.traceb[lo + path]] = nxt;
      rewired = true;
    }
    visit(path, a);
    const int from = path;
    if (rewired) { path = n.traceb[lo + path]; a = walk_node(V, path); }
    else { path = nxt; a = b; }
    n.tracef[lo + path] = from;
  }
  visit(path, a);
  return max_ndx;
}
GFN int walk_dprog_finish(const NView &V, int max_ndx) { return walk_dprog_finish(V, max_ndx, [](int, const WalkNode &) {}); }

struct GeneSlot { int32_t begin, end, start_ndx, stop_ndx; };

GFN void walk_eliminate_bad_genes(const NView &V, int dbeg, double st_wt) {
  const Nodes &n = V.n; const uint32_t lo = V.lo;
  if (dbeg == -1) return;
  int path = dbeg;
  while (n.traceb[lo + path] != -1) path = n.traceb[lo + path];
  while (n.tracef[lo + path] != -1) {
    const int f = n.tracef[lo + path];
    if (V.strand(path) == 1 && V.stop(path)) n.sscore[lo + f] += igm_nodes(V, path, f, st_wt);
    if (V.strand(path) == -1 && !V.stop(path)) n.sscore[lo + path] += igm_nodes(V, path, f, st_wt);
    path = f;
  }
  path = dbeg;
  while (n.traceb[lo + path] != -1) path = n.traceb[lo + path];
  while (n.tracef[lo + path] != -1) {
    const int f = n.tracef[lo + path];
    if (V.strand(path) == 1 && !V.stop(path) && n.cscore[lo + path] + n.sscore[lo + path] < 0) { n.elim[lo + path] = 1; n.elim[lo + f] = 1; }
    if (V.strand(path) == -1 && V.stop(path) && n.cscore[lo + f] + n.sscore[lo + f] < 0) { n.elim[lo + path] = 1; n.elim[lo + f] = 1; }
    path = f;
  }
}
GFN int walk_add_genes(const NView &V, int dbeg, GeneSlot *gl) {
  const Nodes &n = V.n; const uint32_t lo = V.lo;
  if (dbeg == -1) return 0;
  int path = dbeg, ng = 0;
  while (n.traceb[lo + path] != -1) path = n.traceb[lo + path];
  GeneSlot cur{0, 0, 0, 0};
  while (path != -1) {
    if (n.elim[lo + path] == 1) { path = n.tracef[lo + path]; continue; }
    if (V.strand(path) == 1 && !V.stop(path)) { cur.begin = V.ndx(path) + 1; cur.start_ndx = path; }
    if (V.strand(path) == -1 && V.stop(path)) { cur.begin = V.ndx(path) - 1; cur.stop_ndx = path; }
    if (V.strand(path) == 1 && V.stop(path)) { cur.end = V.ndx(path) + 3; cur.stop_ndx = path; gl[ng++] = cur; }
    if (V.strand(path) == -1 && !V.stop(path)) { cur.end = V.ndx(path) + 1; cur.start_ndx = path; gl[ng++] = cur; }
    path = n.tracef[lo + path];
  }
  return ng;
}
GFN void walk_tweak_final_starts(const NView &V, GeneSlot *genes, int ng, double st_wt) {
  const Nodes &n = V.n; const uint32_t lo = V.lo; const int nn = V.nn;
  for (int i = 0; i < ng; ++i) {
    const int ndx = genes[i].start_ndx;
    const double sc = n.sscore[lo + ndx] + n.cscore[lo + ndx];
    double igm = 0.0;
    if (i > 0 && V.strand(ndx) == 1 && V.strand(genes[i - 1].start_ndx) == 1) igm = igm_nodes(V, genes[i - 1].stop_ndx, ndx, st_wt);
    if (i > 0 && V.strand(ndx) == 1 && V.strand(genes[i - 1].start_ndx) == -1) igm = igm_nodes(V, genes[i - 1].start_ndx, ndx, st_wt);
    if (i < ng - 1 && V.strand(ndx) == -1 && V.strand(genes[i + 1].start_ndx) == 1) igm = igm_nodes(V, ndx, genes[i + 1].start_ndx, st_wt);
    if (i < ng - 1 && V.strand(ndx) == -1 && V.strand(genes[i + 1].start_ndx) == -1) igm = igm_nodes(V, ndx, genes[i + 1].stop_ndx, st_wt);
    int maxndx[2] = {-1, -1}; double maxsc[2] = {0, 0}, maxigm[2] = {0, 0};
    for (int j = ndx - 100; j < ndx + 100; ++j) {
      if (j < 0 || j >= nn || j == ndx) continue;
      if (V.stop(j) || V.sv(j) != V.sv(ndx)) continue;
      double tigm = 0.0;
      if (i > 0 && V.strand(j) == 1 && V.strand(genes[i - 1].start_ndx) == 1) {
        if (V.ndx(genes[i - 1].stop_ndx) - V.ndx(j) > MAX_SAM_OVLP) continue;
        tigm = igm_nodes(V, genes[i - 1].stop_ndx, j, st_wt);
      }
      if (i > 0 && V.strand(j) == 1 && V.strand(genes[i - 1].start_ndx) == -1) {
        if (V.ndx(genes[i - 1].start_ndx) - V.ndx(j) >= 0) continue;
        tigm = igm_nodes(V, genes[i - 1].start_ndx, j, st_wt);
      }
      if (i < ng - 1 && V.strand(j) == -1 && V.strand(genes[i + 1].start_ndx) == 1) {
        if (V.ndx(j) - V.ndx(genes[i + 1].start_ndx) >= 0) continue;
        tigm = igm_nodes(V, j, genes[i + 1].start_ndx, st_wt);
      }
      if (i < ng - 1 && V.strand(j) == -1 && V.strand(genes[i + 1].start_ndx) == -1) {
        if (V.ndx(j) - V.ndx(genes[i + 1].stop_ndx) > MAX_SAM_OVLP) continue;
        tigm = igm_nodes(V, j, genes[i + 1].stop_ndx, st_wt);
      }
      const double v = n.cscore[lo + j] + n.sscore[lo + j];
      if (maxndx[0] == -1) { maxndx[0] = j; maxsc[0] = v; maxigm[0] = tigm; }
      else if (v + tigm > maxsc[0] + maxigm[0]) { maxndx[1] = maxndx[0]; maxsc[1] = maxsc[0]; maxigm[1] = maxigm[0]; maxndx[0] = j; maxsc[0] = v; maxigm[0] = tigm; }
      else if (maxndx[1] == -1 || v + tigm > maxsc[1] + maxigm[1]) { maxndx[1] = j; maxsc[1] = v; maxigm[1] = tigm; }
    }
    for (int j = 0; j < 2; ++j) {
      const int m = maxndx[j];
      if (m == -1) continue;
      if (n.tscore[lo + m] < n.tscore[lo + ndx] && maxsc[j] - n.tscore[lo + m] >= sc - n.tscore[lo + ndx] + st_wt && n.rscore[lo + m] > n.rscore[lo + ndx] &&
          n.uscore[lo + m] > n.uscore[lo + ndx] && n.cscore[lo + m] > n.cscore[lo + ndx] && abs(V.ndx(m) - V.ndx(ndx)) > 15) {
        maxsc[j] += n.tscore[lo + ndx] - n.tscore[lo + m];
      } else if (abs(V.ndx(m) - V.ndx(ndx)) <= 15 && n.rscore[lo + m] + n.tscore[lo + m] > n.rscore[lo + ndx] + n.tscore[lo + ndx] && V.edge(ndx) == 0 && V.edge(m) == 0) {
        if (n.cscore[lo + ndx] > n.cscore[lo + m]) maxsc[j] += n.cscore[lo + ndx] - n.cscore[lo + m];
        if (n.uscore[lo + ndx] > n.uscore[lo + m]) maxsc[j] += n.uscore[lo + ndx] - n.uscore[lo + m];
        if (igm > maxigm[j]) maxsc[j] += igm - maxigm[j];
      } else maxsc[j] = -1000.0;
    }
    int m = -1;
    for (int j = 0; j < 2; ++j) {
      if (maxndx[j] == -1) continue;
      if (m == -1 && maxsc[j] + maxigm[j] > sc + igm) m = j;
      else if (m >= 0 && maxsc[j] + maxigm[j] > maxsc[m] + maxigm[m]) m = j;
    }
    if (m != -1 && V.strand(maxndx[m]) == 1) { genes[i].start_ndx = maxndx[m]; genes[i].begin = V.ndx(maxndx[m]) + 1; }
    else if (m != -1 && V.strand(maxndx[m]) == -1) { genes[i].start_ndx = maxndx[m]; genes[i].end = V.ndx(maxndx[m]) + 1; }
  }
}

// ---- host arithmetic of the training tables (libm) ----
inline void host_ups_to_log(GTrainH &t) {
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
inline double host_update_type_wt(GTrainH &t, double *treal, const double *tbg) {
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
inline void host_determine_sd_usage(GTrainH &t) {
  t.uses_sd = 1;
  if (t.rbs_wt[0] >= 0.0) t.uses_sd = 0;
  if (t.rbs_wt[16] < 1.0 && t.rbs_wt[13] < 1.0 && t.rbs_wt[15] < 1.0 && (t.rbs_wt[0] >= -0.5 || (t.rbs_wt[22] < 2.0 && t.rbs_wt[24] < 2.0 && t.rbs_wt[27] < 2.0))) t.uses_sd = 0;
}
inline void host_build_coverage_map(const std::vector<double> &real, std::vector<int> &good, double ng) {
  auto R = [&](int a, int b, int c) { return real[((size_t)a * 4 + b) * 4096 + c]; };
  auto G = [&](int a, int b, int c) -> int & { return good[((size_t)a * 4 + b) * 4096 + c]; };
  const double thresh = 0.2; int decomp[3];
  std::fill(good.begin(), good.end(), 0);
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 64; ++j) if (R(0, i, j) / ng >= thresh) for (int k = 0; k < 4; ++k) G(0, k, j) = 1;
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
inline double host_confidence(double score, double st_wt) {
  double conf;
  if (score / st_wt < 41) { conf = exp(score / st_wt); conf = (conf / (conf + 1)) * 100.0; } else conf = 99.99;
  if (conf <= 50.00) conf = 50.00;
  return conf;
}

// counters of one bin during the start-site training (uint32 each)
constexpr int TC_RBG = 0, TC_RREAL = 28, TC_TREAL = 56, TC_TBG = 59, TC_UPS = 62, TC_NGENES = 190, TC_ZBG = 191, TC_ZREAL = 192, TC_SIZE = 200;
constexpr size_t MOT_N = (size_t)4 * 4 * 4096;
constexpr int TEXT_PIECE = 1024;          // bases of a contig one wavefront lays out (16 steps of 64)

// walk over the starts of the open reading frame whose STOP node is x, outermost first (the order of the reference's frame sweeps):
// the chain array holds [sentinel][starts of ORF 1, inner to outer][its stop][starts of ORF 2 ...]
template <class F> GFN void for_orf_starts(const Nodes &nd, const uint32_t *CH, uint32_t x, F f) {
  for (uint32_t q = nd.chx[x]; q > 0;) {
    --q;
    const uint32_t y = CH[q];
    if (y == 0xffffffffu || nd.type[y] == G_STOP) break;
    f(y);
  }
}

inline void gene_pipeline(GExec &e, const PipeInput &in, GeneResult &out) {
  const uint32_t nbins = in.nbins, ncontigs = in.ncontigs, nseq = nbins + ncontigs;
  const int tt = in.trans_table;
  auto tp = [&](const char *s) { if (in.trace) in.trace(s); };
  // ---- layout: one training sequence per bin (its contigs joined and, beyond one contig, closed by the separator); contigs are sub-ranges ----
  std::vector<uint64_t> seq_off(nseq); std::vector<int32_t> seq_len(nseq); std::vector<uint32_t> seq_bin(nseq), piece_first(ncontigs + 1, 0);
  std::vector<uint64_t> bin_total(nbins, 0);
  uint64_t pos = 0;
  for (uint32_t b = 0; b < nbins; ++b) {
    const uint32_t c0 = in.bin_first[b], c1 = in.bin_first[b + 1];
    const bool multi = c1 - c0 > 1;
    seq_off[b] = pos; seq_bin[b] = b;
    uint64_t p = pos;
    for (uint32_t c = c0; c < c1; ++c) {
      const uint64_t n = in.contig_off[c + 1] - in.contig_off[c];
      if (n > 0x7ffffff0ull) g_fail("contig longer than 2^31 bases");
      seq_off[nbins + c] = p; seq_len[nbins + c] = (int32_t)n; seq_bin[nbins + c] = b; bin_total[b] += n;
      piece_first[c + 1] = piece_first[c] + (uint32_t)std::max<uint64_t>(1, (n + TEXT_PIECE - 1) / TEXT_PIECE);
      p += n + (multi ? 12 : 0);
    }
    if (p - pos > 0x7ffffff0ull) g_fail("bin longer than 2^31 bases");
    seq_len[b] = (int32_t)(p - pos);
    pos = (p + 2 + 15) & ~(uint64_t)15;
  }
  const uint64_t body = (pos + 63) & ~(uint64_t)63, nwin = body / 64;
  if (body > 0xfffffff00ull) g_fail("gene-calling batch beyond 64 Gbase");
  const uint64_t raw_bytes = in.contig_off[ncontigs];
  std::vector<uint8_t> trained(nbins, 0);
  for (uint32_t b = 0; b < nbins; ++b) trained[b] = bin_total[b] >= 20000 ? 1 : 0;          // (prodigal refuses to train on less; CheckM switches to -p meta below 100 kb, which is not built)
  out.bin_trained.assign(trained.begin(), trained.end()); out.bin_bases.assign(bin_total.begin(), bin_total.end());
  out.bin_uses_sd.assign(nbins, 0); out.bin_gc.assign(nbins, 0.0); out.bin_coding.assign(nbins, 0); out.bin_nodes_train.assign(nbins, 0); out.bin_nodes_find.assign(nbins, 0);
  out.prot_off.assign(1, 0);
  if (!nbins) return;

  // ---- device: text ----
  GBuf d_raw, d_coff, d_pf, d_pc, d_bfirst, d_ascii, d_code, d_off, d_len, d_sbin, d_gcc, d_flags, d_gcw, d_uw, d_r50, d_pg, d_pr, d_scan;
  d_raw.ensure(raw_bytes + 64); d_coff.ensure((size_t)(ncontigs + 1) * 8); d_pf.ensure((size_t)(ncontigs + 1) * 4); d_bfirst.ensure((size_t)(nbins + 1) * 4);
  d_ascii.ensure(64 + body + 128); d_code.ensure(body + 64); d_off.ensure((size_t)nseq * 8); d_len.ensure((size_t)nseq * 4); d_sbin.ensure((size_t)nseq * 4); d_gcc.ensure((size_t)nbins * 8);
  d_flags.ensure(body + 256); d_gcw.ensure((nwin + 8) * 8); d_uw.ensure((nwin + 2) * 8); d_r50.ensure((nwin + 2) * 8); d_pg.ensure((nwin + 2) * 4); d_pr.ensure((nwin + 2) * 4);
  g_h2d(e, d_raw.p, in.text, raw_bytes); g_h2d(e, d_coff.p, in.contig_off, (size_t)(ncontigs + 1) * 8); g_h2d(e, d_pf.p, piece_first.data(), (size_t)(ncontigs + 1) * 4);
  g_h2d(e, d_bfirst.p, in.bin_first, (size_t)(nbins + 1) * 4); g_h2d(e, d_off.p, seq_off.data(), (size_t)nseq * 8); g_h2d(e, d_len.p, seq_len.data(), (size_t)nseq * 4);
  g_h2d(e, d_sbin.p, seq_bin.data(), (size_t)nseq * 4);
  g_zero(e, d_ascii.p, 'N', 64 + body + 128); g_zero(e, d_code.p, CODE_PAD, body + 64); g_zero(e, d_gcc.p, 0, (size_t)nbins * 8);
  g_zero(e, d_gcw.p, 0, (nwin + 8) * 8); g_zero(e, d_uw.p, 0, (nwin + 2) * 8); g_zero(e, d_r50.p, 0, (nwin + 2) * 8);
  const uint64_t *soff = d_off.as<uint64_t>(); const int32_t *slen_d = d_len.as<int32_t>(); const uint32_t *sbin = d_sbin.as<uint32_t>();
  uint8_t *code = d_code.as<uint8_t>();
  {
    const uint8_t *raw = d_raw.as<uint8_t>(); const uint64_t *coff = d_coff.as<uint64_t>(); const uint32_t *pf = d_pf.as<uint32_t>(), *bfirst = d_bfirst.as<uint32_t>();
    uint8_t *ascii = d_ascii.as<uint8_t>() + 64; unsigned long long *gcc = d_gcc.as<unsigned long long>();
    const uint32_t nb = nbins, nc = ncontigs;
    // a wavefront per 1024-base piece of a contig, 64 bases at a time (round 5: a thread per 64-base piece, whose 64 byte reads and
    // 128 byte writes touched a cache line each -- 14 % of a call's wavefront-cycles, profiles/r06e; a wavefront per 64 bases paid the
    // piece's six dependent look-ups for one line of text, profiles/r06n): the bytes of a step are one line, the G + C count one atomic
    // per wavefront.  The contig of every piece comes from the host.
    uint32_t *gcc32 = reinterpret_cast<uint32_t *>(gcc);          // (low words of the 64-bit sums: a bin holds fewer than 2^31 bases)
    std::vector<uint32_t> h_pc(piece_first[ncontigs]);
    for (uint32_t c = 0; c < ncontigs; ++c) std::fill(h_pc.begin() + piece_first[c], h_pc.begin() + piece_first[c + 1], c);
    d_pc.ensure(std::max<size_t>(1, h_pc.size()) * 4);
    g_h2d(e, d_pc.p, h_pc.data(), h_pc.size() * 4);
    const uint32_t *pc = d_pc.as<uint32_t>();
    (void)nc;
    g_map(e, (size_t)piece_first[ncontigs] * 64, [=] GLAM(size_t t) {
      const size_t p = t >> 6; const int lane = (int)(t & 63);
      const uint32_t c = pc[p]; const uint64_t k = p - pf[c], len = coff[c + 1] - coff[c];
      const uint64_t done = (uint64_t)TEXT_PIECE * k; const int n = (int)(len - done < (uint64_t)TEXT_PIECE ? len - done : (uint64_t)TEXT_PIECE);
      const uint8_t *src = raw + coff[c] + done; const uint64_t o = soff[nb + c] + done;
      uint32_t gc = 0;
      for (int i = lane; i < n; i += 64) {
        const uint8_t ch = src[i]; uint8_t v;
        switch (ch) { case 'A': case 'a': v = 0; break; case 'C': case 'c': v = 1; gc++; break; case 'G': case 'g': v = 2; gc++; break;
                      case 'T': case 't': case 'U': case 'u': v = 3; break; default: v = 5; }
        ascii[o + i] = ch; code[o + i] = v;
      }
      g_count_n(&gcc32[2 * (size_t)sbin[nb + c]], gc);
    });
    {
      // the separator behind every contig of a bin of several
      g_map(e, (size_t)ncontigs, [=] GLAM(size_t c) {
        const uint64_t len = coff[c + 1] - coff[c]; const uint32_t b = sbin[nb + c];
        if (bfirst[b + 1] - bfirst[b] <= 1) return;
        const uint64_t o = soff[nb + c] + len; const char sep[13] = "TTAATTAATTAA";
        for (int i = 0; i < 12; ++i) { ascii[o + i] = (uint8_t)sep[i]; code[o + i] = sep[i] == 'T' ? 3 : 0; }
      });
    }
  }
  x_orf_flags(e, d_ascii.as<uint8_t>() + 64, d_flags.as<unsigned long long>(), body);
  unsigned long long *gcw = d_gcw.as<unsigned long long>() + 2, *uw = d_uw.as<unsigned long long>(), *r50 = d_r50.as<unsigned long long>();
  uint32_t *pg = d_pg.as<uint32_t>(), *pr = d_pr.as<uint32_t>();
  g_map(e, nwin, [=] GLAM(size_t w) {
    const uint8_t *c = code + 64 * w; unsigned long long g = 0, u = 0;
    for (int i = 0; i < 64; ++i) { const uint8_t v = c[i]; const int b = v & 3; if (b == 1 || b == 2) g |= 1ull << i; if ((v >> 2) & 1) u |= 1ull << i; }
    gcw[w] = g; uw[w] = u; pg[w] = (uint32_t)popc64(g);
  });
  g_map(e, nwin, [=] GLAM(size_t w) {
    // bit p of r50[w]: the 50 bases from position 64 w + p on are all unknown.  a_k[p] = k unknown bases from p on; 128-bit pairs (lo, hi).
    unsigned long long lo = uw[w], hi = uw[w + 1];
#define CKM_SHR(L, H, k, OL, OH) { OL = ((L) >> (k)) | ((H) << (64 - (k))); OH = (H) >> (k); }
    unsigned long long tl, th, a2l, a2h, a4l, a4h, a8l, a8h, a16l, a16h, a32l, a32h;
    CKM_SHR(lo, hi, 1, tl, th); a2l = lo & tl; a2h = hi & th;
    CKM_SHR(a2l, a2h, 2, tl, th); a4l = a2l & tl; a4h = a2h & th;
    CKM_SHR(a4l, a4h, 4, tl, th); a8l = a4l & tl; a8h = a4h & th;
    CKM_SHR(a8l, a8h, 8, tl, th); a16l = a8l & tl; a16h = a8h & th;
    CKM_SHR(a16l, a16h, 16, tl, th); a32l = a16l & tl; a32h = a16h & th;
    (void)a32h;
    unsigned long long s16l, s16h, s2l, s2h;
    CKM_SHR(a16l, a16h, 32, s16l, s16h); CKM_SHR(a2l, a2h, 48, s2l, s2h);
    (void)s16h; (void)s2h;
#undef CKM_SHR
    const unsigned long long r = a32l & s16l & s2l;
    r50[w] = r; pr[w] = (uint32_t)popc64(r);
  });
  x_scan_u32(e, pg, nwin, d_scan); x_scan_u32(e, pr, nwin, d_scan);
  tp("text, planes, flags");

  // ---- winning GC frame of every codon triple of the training sequences (sequence.c: calc_most_gc_frame) as two bit planes ----
  std::vector<uint64_t> tri_base(nbins);
  for (uint32_t b = 0; b < nbins; ++b) tri_base[b] = (seq_off[b] + 2) / 3;
  const uint64_t ntri = tri_base[nbins - 1] + ((uint64_t)seq_len[nbins - 1] + 2) / 3 + 1, ntw = (ntri + 63) / 64 + 1;
  GBuf d_tb, d_w0, d_w1, d_p0, d_p1;
  d_tb.ensure((size_t)nbins * 8); d_w0.ensure((ntw + 2) * 8); d_w1.ensure((ntw + 2) * 8); d_p0.ensure((ntw + 2) * 4); d_p1.ensure((ntw + 2) * 4);
  g_h2d(e, d_tb.p, tri_base.data(), (size_t)nbins * 8);
  const uint64_t *tbase = d_tb.as<uint64_t>(); unsigned long long *w0 = d_w0.as<unsigned long long>(), *w1 = d_w1.as<unsigned long long>(); uint32_t *p0 = d_p0.as<uint32_t>(), *p1 = d_p1.as<uint32_t>();
  {
    const uint32_t nb = nbins;
    g_map(e, ntw, [=] GLAM(size_t tw) {
      unsigned long long o0 = 0, o1 = 0;
      const unsigned long long M0 = 0x9249249249249249ull, M1 = 0x2492492492492492ull, M2 = 0x4924924924924924ull;
      for (int k = 0; k < 64; ++k) {
        const uint64_t T = 64 * (uint64_t)tw + k;
        if (T < tbase[0]) continue;
        uint32_t lo = 0, hi = nb;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (tbase[mid] <= T) lo = mid; else hi = mid; }
        const uint32_t b = lo; const long long m = (long long)(T - tbase[b]); const int sl = slen_d[b];
        const long long i = 3 * m;
        if (i >= sl - 2) continue;                               // (beyond the last whole triple: reads as frame 2, the value -1 of the reference's table behaves as)
        const long long q = (long long)soff[b] + i - 57;          // first base of the 117-base window; bit t of X = G/C at q + t
        int t_lo = (int)(57 - i > 0 ? 57 - i : 0), t_hi = (int)(sl - i + 57 < 117 ? sl - i + 57 : 117);
        const long long wq = q >> 6; const int bit = (int)(q & 63);
        const unsigned long long x0 = gcw[wq], x1 = gcw[wq + 1], x2 = gcw[wq + 2];
        unsigned long long L = bit ? ((x0 >> bit) | (x1 << (64 - bit))) : x0, H = bit ? ((x1 >> bit) | (x2 << (64 - bit))) : x1;
        L &= below64(t_hi < 64 ? t_hi : 64) & ~below64(t_lo < 64 ? t_lo : 64);
        H &= below64(t_hi - 64) & ~below64(t_lo - 64);
        const int n0 = popc64(L & M0) + popc64(H & M2), n1 = popc64(L & M1) + popc64(H & M0), n2 = popc64(L & M2) + popc64(H & M1);
        const int wfr = max_fr(n0, n1, n2);
        if (wfr == 0) o0 |= 1ull << k; else if (wfr == 1) o1 |= 1ull << k;
      }
      w0[tw] = o0; w1[tw] = o1; p0[tw] = (uint32_t)popc64(o0); p1[tw] = (uint32_t)popc64(o1);
    });
  }
  x_scan_u32(e, p0, ntw, d_scan); x_scan_u32(e, p1, ntw, d_scan);

  // ---- nodes ----
  // the scans: one per (contig, strand, frame); a training sequence of several contigs is cut at its separators (one piece per contig)
  std::vector<SubChain> subs;
  const uint32_t nchains = (nseq + ncontigs) * 6;
  subs.reserve((size_t)ncontigs * 12 + (size_t)nbins * 6);
  auto whole = [&](uint32_t s, uint32_t chain0) {
    const int sl = seq_len[s];
    if (sl < 3) return;
    for (int sub = 0; sub < 6; ++sub) {
      const int frame = sub % 3; int jtop = sl - 3; jtop -= ((jtop % 3) - frame + 3) % 3;
      SubChain q; q.seq = s; q.chain = chain0 + (uint32_t)sub; q.top = jtop; q.bottom = 0; q.rev = sub >= 3; q.frame = (uint8_t)frame; q.after_stop = 0; q.pad = 0;
      subs.push_back(q);
    }
  };
  for (uint32_t b = 0; b < nbins; ++b) {
    const uint32_t c0 = in.bin_first[b], c1 = in.bin_first[b + 1];
    for (uint32_t c = c0; c < c1; ++c) whole(nbins + c, (nbins + c) * 6);
    if (c1 - c0 <= 1) { whole(b, b * 6); continue; }
    // the stop of frame f inside the separator that begins at strand position p: TAA at p + 1, p + 5, p + 9
    auto sep_stop = [](long long p, int f) { for (int k = 1; k <= 9; k += 4) if ((p + k) % 3 == f) return (int)(p + k); return -1; };
    const int sl = seq_len[b];
    for (uint32_t c = c0; c < c1; ++c) {
      const long long cbeg = (long long)(seq_off[nbins + c] - seq_off[b]), cend = cbeg + seq_len[nbins + c];      // the contig in the training sequence; its separator is [cend, cend + 12)
      for (int sub = 0; sub < 6; ++sub) {
        const int frame = sub % 3; const bool rev = sub >= 3;
        SubChain q; q.seq = b; q.chain = (nseq + c) * 6 + (uint32_t)sub; q.rev = rev; q.frame = (uint8_t)frame; q.pad = 0;
        if (!rev) {
          q.after_stop = 1; q.top = sep_stop(cend, frame);                                   // below the separator behind the contig ...
          q.bottom = c == c0 ? 0 : sep_stop(cbeg - 12, frame);                               // ... down to the stop of the separator before it (the first contig: to the sequence's start)
        } else {
          // strand positions: the separator [p, p + 12) of the forward strand is [sl - 12 - p, sl - p) here and reads the same
          if (c == c0) { int jtop = sl - 3; jtop -= ((jtop % 3) - frame + 3) % 3; q.after_stop = 0; q.top = jtop; }
          else { q.after_stop = 1; q.top = sep_stop((long long)sl - 12 - (cbeg - 12), frame); }
          q.bottom = sep_stop((long long)sl - 12 - cend, frame);
        }
        subs.push_back(q);
      }
    }
  }
  GBuf d_np, d_ccnt, d_rec, d_rect, d_recc, d_nrec, d_pn, d_rk, d_subs;
  d_np.ensure(4 * nwin * 8 + 64); d_ccnt.ensure(((size_t)nchains + 2) * 4); d_nrec.ensure(16); d_pn.ensure(2 * (nwin + 2) * 4); d_rk.ensure((size_t)nseq * 8 + 64);
  d_subs.ensure(std::max<size_t>(1, subs.size()) * sizeof(SubChain));
  g_h2d(e, d_subs.p, subs.data(), subs.size() * sizeof(SubChain));
  uint64_t bases = 0; for (uint32_t b = 0; b < nbins; ++b) bases += (uint64_t)seq_len[b];
  unsigned long long cap = std::max<unsigned long long>(1 << 16, bases / 2) + (unsigned long long)subs.size() * 256, n_all = 0;      // (a chain takes its record slots 256 at a time)
  ChainArgs ca;
  ca.planes = d_flags.as<unsigned long long>(); ca.nwin = nwin; ca.seq_off = soff; ca.seq_len = slen_d; ca.nbins = nbins; ca.tt4 = tt == 4 ? 1 : 0;
  ca.sc = reinterpret_cast<const SubChain *>(d_subs.p); ca.nsc = (uint32_t)subs.size();
  ca.r50 = in.mask_runs ? r50 : nullptr; ca.pr50 = in.mask_runs ? pr : nullptr;
  for (int attempt = 0; attempt < 2; ++attempt) {
    d_rec.ensure((size_t)cap * sizeof(OrfRec)); d_rect.ensure((size_t)cap * 4); d_recc.ensure((size_t)cap * 4);
    g_zero(e, d_np.p, 0, 4 * nwin * 8); g_zero(e, d_ccnt.p, 0, ((size_t)nchains + 2) * 4); g_zero(e, d_nrec.p, 0, 16);
    ca.node_planes = d_np.as<unsigned long long>(); ca.chain_cnt = d_ccnt.as<uint32_t>(); ca.rec = d_rec.p; ca.rec_t = d_rect.as<uint32_t>(); ca.rec_c = d_recc.as<uint32_t>();
    ca.nrec = d_nrec.as<unsigned long long>(); ca.cap = cap;
    x_chain(e, ca);
    g_d2h(e, &n_all, d_nrec.p, 8); g_sync(e);
    if (n_all <= cap) break;
    cap = n_all + 1024;
  }
  if (n_all > 0xfffffff0ull) g_fail("more than 2^32 nodes in one gene-calling batch");
  // ranks: exclusive prefix of nodes per word, per set; a chain's entries begin one slot behind its offset (slot 0: the sentinel)
  unsigned long long *np = d_np.as<unsigned long long>(); uint32_t *pn = d_pn.as<uint32_t>(); uint32_t *ccnt = d_ccnt.as<uint32_t>();
  {
    const uint64_t nw = nwin;
    g_map(e, 2 * nwin, [=] GLAM(size_t k) { const size_t set = k / nw, w = k % nw; pn[set * (nw + 2) + w] = (uint32_t)(popc64(np[(set * 2) * nw + w]) + popc64(np[(set * 2 + 1) * nw + w])); });
    g_map(e, (size_t)nchains, [=] GLAM(size_t c) { ccnt[c] += 1; });
  }
  x_scan_u32(e, pn, nwin, d_scan); x_scan_u32(e, pn + (nwin + 2), nwin, d_scan); x_scan_u32(e, ccnt, (size_t)nchains, d_scan);
  uint32_t *rk = d_rk.as<uint32_t>();
  {
    const uint32_t nb = nbins; const uint64_t nw = nwin;
    g_map(e, nseq, [=] GLAM(size_t s) {
      const size_t set = s < nb ? 0 : 1;
      const unsigned long long *F = np + (set * 2) * nw, *R = np + (set * 2 + 1) * nw; const uint32_t *P = pn + set * (nw + 2);
      const uint64_t a = soff[s], z = soff[s] + (uint64_t)slen_d[s];
      rk[2 * s] = P[a >> 6] + (uint32_t)popc64(F[a >> 6] & below64((int)(a & 63))) + (uint32_t)popc64(R[a >> 6] & below64((int)(a & 63)));
      rk[2 * s + 1] = P[z >> 6] + (uint32_t)((z >> 6) < nw ? popc64(F[z >> 6] & below64((int)(z & 63))) + popc64(R[z >> 6] & below64((int)(z & 63))) : 0);
    });
  }
  std::vector<uint32_t> h_rk((size_t)nseq * 2); uint32_t n_chain_slots = 0;
  g_d2h(e, h_rk.data(), rk, (size_t)nseq * 8); g_d2h(e, &n_chain_slots, ccnt + (size_t)nchains, 4); g_sync(e);
  // node ranges: a bin's nodes start at a multiple of 256; untrained bins keep none
  std::vector<uint32_t> seq_lo(nseq, 0), seq_n(nseq, 0); std::vector<long long> xbase(nseq, -1);
  size_t NT[2] = {0, 0};
  for (int set = 0; set < 2; ++set) {
    size_t k = 0;
    for (uint32_t b = 0; b < nbins; ++b) {
      if (!trained[b]) continue;
      const uint32_t s0 = set == 0 ? b : nbins + in.bin_first[b], s1 = set == 0 ? b + 1 : nbins + in.bin_first[b + 1];
      for (uint32_t s = s0; s < s1; ++s) {
        seq_lo[s] = (uint32_t)k; seq_n[s] = h_rk[2 * s + 1] - h_rk[2 * s]; xbase[s] = (long long)k - (long long)h_rk[2 * s]; k += seq_n[s];
        if (set == 0) out.bin_nodes_train[b] += seq_n[s]; else out.bin_nodes_find[b] += seq_n[s];
      }
      k = (k + 255) & ~(size_t)255;
    }
    if (k > 0xfffffff0ull) g_fail("more than 2^32 nodes in one gene-calling batch");
    NT[set] = k;
  }
  GBuf d_slo, d_sn, d_xb, d_ch, d_stwt;
  d_slo.ensure((size_t)nseq * 4); d_sn.ensure((size_t)nseq * 4); d_xb.ensure((size_t)nseq * 8); d_ch.ensure(((size_t)n_chain_slots + 64) * 4); d_stwt.ensure((size_t)nbins * 8);
  g_h2d(e, d_slo.p, seq_lo.data(), (size_t)nseq * 4); g_h2d(e, d_sn.p, seq_n.data(), (size_t)nseq * 4); g_h2d(e, d_xb.p, xbase.data(), (size_t)nseq * 8);
  g_zero(e, d_ch.p, 0xff, ((size_t)n_chain_slots + 64) * 4);
  const uint32_t *slo = d_slo.as<uint32_t>(), *sn = d_sn.as<uint32_t>(); const long long *xb = d_xb.as<long long>(); uint32_t *CH = d_ch.as<uint32_t>();
  std::vector<GTrainH> tr(nbins);
  std::vector<double> h_stwt(nbins);
  for (uint32_t b = 0; b < nbins; ++b) { tr[b].trans_table = tt; h_stwt[b] = tr[b].st_wt; }
  g_h2d(e, d_stwt.p, h_stwt.data(), (size_t)nbins * 8);
  const double *stwt = d_stwt.as<double>();
  NodeSet TS, FS;
  TS.alloc(e, NT[0], true); FS.alloc(e, NT[1], false);
  {
    const Nodes t0 = TS.dev(), t1 = FS.dev(); const OrfRec *rec = reinterpret_cast<const OrfRec *>(d_rec.p); const uint32_t *rect = d_rect.as<uint32_t>(), *recc = d_recc.as<uint32_t>();
    const uint32_t nb = nbins; const uint64_t nw = nwin;
    g_map(e, (size_t)n_all, [=] GLAM(size_t r) {
      const OrfRec q = rec[r]; const uint32_t s = q.seq;
      if (s == 0xffffffffu) return;                                      // (a slot its chain reserved and did not use)
      if (xb[s] == -1 && sn[s] == 0) return;                            // (sequence of an untrained bin, or one without nodes)
      const size_t set = s < nb ? 0 : 1;
      const Nodes &nd = set == 0 ? t0 : t1;
      const unsigned long long *F = np + (set * 2) * nw, *R = np + (set * 2 + 1) * nw; const uint32_t *P = pn + set * (nw + 2);
      const uint64_t g = soff[s] + (uint64_t)q.ndx; const int bit = (int)(g & 63);
      const uint32_t rank = P[g >> 6] + (uint32_t)popc64(F[g >> 6] & below64(bit)) + (uint32_t)popc64(R[g >> 6] & below64(bit)) + (q.strand_rev ? (uint32_t)((F[g >> 6] >> bit) & 1ull) : 0u);
      const uint32_t x = (uint32_t)(xb[s] + (long long)rank);
      nd.bin[x] = sbin[s]; nd.seq[x] = s; nd.ndx[x] = q.ndx; nd.sv[x] = q.sv; nd.strand[x] = q.strand_rev ? -1 : 1; nd.type[x] = q.type; nd.edge[x] = q.edge;
      const uint32_t slot = ccnt[recc[r]] + 1 + rect[r];
      CH[slot] = x; nd.chx[x] = slot;
    });
  }
  for (uint32_t b = 0; b < nbins; ++b) tr[b].gc = 0.0;
  std::vector<unsigned long long> h_gcc(nbins);
  g_d2h(e, h_gcc.data(), d_gcc.p, (size_t)nbins * 8); g_sync(e);
  for (uint32_t b = 0; b < nbins; ++b) tr[b].gc = seq_len[b] ? (double)h_gcc[b] / (double)seq_len[b] : 0.0;
  tp("nodes in working order");

  // per-bin constants of the coding-score passes (libm on the host)
  std::vector<double> h_lfac((size_t)nbins * 1001, 0.0), h_l80(nbins, 0.0);
  in.pfor(nbins, [&](size_t b) {
    if (!trained[b]) return;
    const double gc = tr[b].gc; double no_stop;
    if (tt != 11) { no_stop = ((1 - gc) * (1 - gc) * gc) / 8.0; no_stop += ((1 - gc) * (1 - gc) * (1 - gc)) / 8.0; no_stop = 1 - no_stop; }
    else { no_stop = ((1 - gc) * (1 - gc) * gc) / 4.0; no_stop += ((1 - gc) * (1 - gc) * (1 - gc)) / 8.0; no_stop = 1 - no_stop; }
    h_l80[b] = log((1 - pow(no_stop, 80)) / pow(no_stop, 80));
    for (int g = 1; g <= 1000; ++g) { const double gs = (double)g; h_lfac[b * 1001 + g] = log((1 - pow(no_stop, gs)) / pow(no_stop, gs)); }
  });
  GBuf d_lfac, d_l80;
  d_lfac.ensure(h_lfac.size() * 8); d_l80.ensure((size_t)nbins * 8);
  g_h2d(e, d_lfac.p, h_lfac.data(), h_lfac.size() * 8); g_h2d(e, d_l80.p, h_l80.data(), (size_t)nbins * 8);
  const double *lfac_raw = d_lfac.as<double>(), *l80 = d_l80.as<double>();

  // kernels shared by the two node sets -------------------------------------------------------------------------------------------------
  auto overlapping_starts = [&](const Nodes nd, size_t n, int flag) {
    g_map(e, n, [=] GLAM(size_t x) {
      if (nd.type[x] != G_STOP || nd.edge[x] == 1) return;
      const uint32_t s = nd.seq[x]; const NView V{nd, slo[s], (int)sn[s]};
      const int i = (int)(x - V.lo), nn = V.nn; const double st_wt = stwt[nd.bin[x]];
      int sp[3] = {-1, -1, -1};
      double max_sc = -100.0;
      if (V.strand(i) == 1) {
        for (int j = i + 3; j >= 0; --j) {
          if (j >= nn || V.ndx(j) > V.ndx(i) + 2) continue;
          if (V.ndx(j) + MAX_SAM_OVLP < V.ndx(i)) break;
          if (V.strand(j) == 1 && !V.stop(j)) {
            if (V.sv(j) <= V.ndx(i)) continue;
            const int f = V.ndx(j) % 3;
            if (flag == 0 && sp[f] == -1) sp[f] = j;
            else if (flag == 1) { const double sc = nd.cscore[V.lo + j] + nd.sscore[V.lo + j] + igm_nodes(V, i, j, st_wt); if (sc > max_sc) { sp[f] = j; max_sc = sc; } }
          }
        }
      } else {
        for (int j = i - 3; j < nn; ++j) {
          if (j < 0 || V.ndx(j) < V.ndx(i) - 2) continue;
          if (V.ndx(j) - MAX_SAM_OVLP > V.ndx(i)) break;
          if (V.strand(j) == -1 && !V.stop(j)) {
            if (V.sv(j) >= V.ndx(i)) continue;
            const int f = V.ndx(j) % 3;
            if (flag == 0 && sp[f] == -1) sp[f] = j;
            else if (flag == 1) { const double sc = nd.cscore[V.lo + j] + nd.sscore[V.lo + j] + igm_nodes(V, j, i, st_wt); if (sc > max_sc) { sp[f] = j; max_sc = sc; } }
          }
        }
      }
      for (int f = 0; f < 3; ++f) nd.star[x * 3 + f] = sp[f];
    });
  };
  auto dp_windows = [&](const Nodes nd, size_t n) {
    g_map(e, n, [=] GLAM(size_t x) {
      if (nd.type[x] == G_PAD) return;
      const uint32_t s = nd.seq[x]; const NView V{nd, slo[s], (int)sn[s]};
      const int i = (int)(x - V.lo);
      int mn = i < MAX_NODE_DIST ? 0 : i - MAX_NODE_DIST;
      if (V.strand(i) == -1 && !V.stop(i) && V.ndx(mn) >= V.sv(i)) while (mn >= 0 && V.ndx(mn) != V.sv(i)) mn--;
      if (V.strand(i) == 1 && V.stop(i) && V.ndx(mn) >= V.sv(i)) while (mn >= 0 && V.ndx(mn) != V.sv(i)) mn--;
      mn = mn < MAX_NODE_DIST ? 0 : mn - MAX_NODE_DIST;
      nd.dp_min[x] = (uint32_t)mn;
    });
  };
  // raw_coding_score behind its first pass: both frame sweeps of an open reading frame by the thread of its stop node
  auto coding_passes = [&](const Nodes nd, size_t n) {
    const uint32_t *ch = CH;
    g_map(e, n, [=] GLAM(size_t x) {
      if (nd.type[x] != G_STOP) return;
      const uint32_t b = nd.bin[x]; const double *lf = lfac_raw + (size_t)b * 1001; const double l80b = l80[b];
      double sa = -10000, sb = -10000;
      for_orf_starts(nd, ch, (uint32_t)x, [&](uint32_t y) {
        double cs = nd.cscore[y];
        if (cs > sa) sa = cs; else cs -= (sa - cs);
        const double gsize = ((double)(abs(nd.sv[y] - nd.ndx[y]) + 3.0)) / 3.0;
        double lfac;
        if (gsize > 1000.0) { lfac = lf[1000]; lfac -= l80b; lfac *= (gsize - 80) / 920.0; }
        else { lfac = lf[(int)gsize]; lfac -= l80b; }
        if (lfac > sb) sb = lfac; else lfac -= dmaxd(dmind(sb - lfac, lfac), 0);
        if (lfac > 3.0 && cs < 0.5 * lfac) cs = 0.5 * lfac;
        cs += lfac;
        nd.cscore[y] = cs;
      });
    });
  };
  auto upstream_windows = [&](const Nodes nd, size_t n) {
    g_map(e, n, [=] GLAM(size_t x) {
      if (nd.type[x] >= G_STOP || nd.edge[x] == 1) return;
      const uint32_t s = nd.seq[x]; const GSeq q{code + soff[s], slen_d[s]};
      const int strand = nd.strand[x], start = strand == 1 ? nd.ndx[x] : q.slen - 1 - nd.ndx[x];
      nd.upw[x] = upstream_window(q, strand, start);
    });
  };
  auto run_dp = [&](const Nodes nd, uint32_t s0, uint32_t ns, int flag, double &ms) {
    GTimer t; t.begin(e);
    x_dp(e, nd, slo + s0, sn + s0, sbin + s0, stwt, ns, flag);
    ms += t.end(e);
  };

  // =====================================================  training  =====================================================
  const Nodes tn = TS.dev();
  GBuf d_bias, d_dc, d_par, d_cnt, d_gi, d_gcnt, d_hex, d_hcnt, d_ipath;
  d_bias.ensure((size_t)nbins * 24); d_dc.ensure((size_t)nbins * 4096 * 8); d_cnt.ensure((size_t)nbins * TC_SIZE * 4);
  d_gcnt.ensure((size_t)nbins * 4); d_hex.ensure((size_t)nbins * 4096 * 4); d_hcnt.ensure((size_t)nbins * 4096 * 4); d_ipath.ensure((size_t)nbins * 4);
  // the start-site weights of every bin in ONE block -- [28 Shine-Dalgarno bins][3 start types][threshold] x nbins -- so that a training
  // round uploads them with one copy
  const size_t npar = (size_t)nbins * 32;
  d_par.ensure(npar * 8);
  double *const rw_dev = d_par.as<double>(), *const tw_dev = rw_dev + (size_t)nbins * 28, *const sth_dev = tw_dev + (size_t)nbins * 3;
  double *bias = d_bias.as<double>();
  if (NT[0]) {
    // GC-frame codon counts of every start node's reading frame, its bias class and its term of the ordered sum (node.c: record_gc_bias)
    g_map(e, NT[0], [=] GLAM(size_t x) {
      const int t = tn.type[x];
      if (t >= G_STOP) return;
      const uint32_t b = tn.bin[x]; const int ndx = tn.ndx[x], sv = tn.sv[x], fr = ndx % 3; const bool fwd = tn.strand[x] == 1;
      const long long m0 = ((fwd ? ndx : sv) - fr) / 3, m1 = ((fwd ? sv : ndx) - fr) / 3;
      const uint64_t T0 = tbase[b] + (uint64_t)m0, T1 = tbase[b] + (uint64_t)m1 + 1;
      int cw[3];
      cw[0] = (int)(plane_rank(p0, w0, T1) - plane_rank(p0, w0, T0)); cw[1] = (int)(plane_rank(p1, w1, T1) - plane_rank(p1, w1, T0)); cw[2] = (int)(T1 - T0) - cw[0] - cw[1];
      int ctr[3];
      for (int k = 0; k < 3; ++k) ctr[k] = fwd ? cw[(k + fr) % 3] : cw[(fr - k + 3) % 3];
      const int gb = max_fr(ctr[0], ctr[1], ctr[2]);
      double gs = 3.0 * ctr[gb]; gs /= 1.0 * (fwd ? sv - ndx + 3 : ndx - sv + 3);
      const int len = abs(sv - ndx) + 1;
      tn.ctr[x * 3] = ctr[0]; tn.ctr[x * 3 + 1] = ctr[1]; tn.ctr[x * 3 + 2] = ctr[2]; tn.gcb_cls[x] = (uint8_t)gb; tn.gcb_term[x] = (gs * len) / 1000.0;
    });
    x_gc_bias(e, tn, slo, sn, nbins, bias);
    g_map(e, NT[0], [=] GLAM(size_t x) {
      if (tn.type[x] == G_PAD) return;
      const uint32_t b = tn.bin[x]; double g0 = 0, g1 = 0, g2 = 0;
      if (tn.type[x] != G_STOP) {
        const int ndx = tn.ndx[x], sv = tn.sv[x]; const double L = 1.0 * (tn.strand[x] == 1 ? sv - ndx + 3 : ndx - sv + 3);
        g0 = 3.0 * tn.ctr[x * 3]; g0 /= L; g1 = 3.0 * tn.ctr[x * 3 + 1]; g1 /= L; g2 = 3.0 * tn.ctr[x * 3 + 2]; g2 /= L;
      }
      const double a = bias[b * 3] * g0, c = bias[b * 3 + 1] * g1, d = bias[b * 3 + 2] * g2;
      tn.gcb[x] = (a + c) + d;
    });
    overlapping_starts(tn, NT[0], 0);
    dp_windows(tn, NT[0]);
    tp("training dp inputs");
    run_dp(tn, 0, nbins, 0, out.ms_dp_train);
    tp("training dp done");
    // trace-back of every bin and the genes of the first pass, as (strand, left, right) intervals in the bin's own node slots
    d_gi.ensure(std::max<size_t>(NT[0], 64) * 12);
    int32_t *gi = reinterpret_cast<int32_t *>(d_gi.p); uint32_t *gcnt = d_gcnt.as<uint32_t>(); int32_t *ipath = d_ipath.as<int32_t>();
    x_path_ends(e, tn, slo, sn, nbins, ipath);
    g_map_waves(e, nbins, [=] GLAM(size_t b) {
      gcnt[b] = 0;
      if (sn[b] == 0) { ipath[b] = -1; return; }
      const NView V{tn, slo[b], (int)sn[b]}; const int sl = slen_d[b];
      int left = -1, right = -1, in_gene = 0; uint32_t ng = 0;
      const int dbeg = walk_dprog_finish(V, ipath[b], [&](int, const WalkNode &w) {          // (the first gene set is read off the path while it is finished)
        if (w.strand == -1 && !w.stop) { in_gene = -1; left = sl - w.ndx - 1; }
        if (w.strand == 1 && w.stop) { in_gene = 1; right = w.ndx + 2; }
        if (in_gene == -1 && w.strand == -1 && w.stop) {
          right = sl - w.ndx + 1;
          int32_t *g = gi + (size_t)(V.lo + ng) * 3; g[0] = -1; g[1] = left; g[2] = right; ng++;
          in_gene = 0;
        }
        if (in_gene == 1 && w.strand == 1 && !w.stop) {
          left = w.ndx;
          int32_t *g = gi + (size_t)(V.lo + ng) * 3; g[0] = 1; g[1] = left; g[2] = right; ng++;
          in_gene = 0;
        }
      });
      ipath[b] = dbeg;
      gcnt[b] = ng;
    });
    uint32_t *hcnt = d_hcnt.as<uint32_t>();
    g_zero(e, d_hcnt.p, 0, (size_t)nbins * 4096 * 4);
    g_map(e, NT[0], [=] GLAM(size_t x) {
      if (tn.type[x] == G_PAD) return;
      const uint32_t b = tn.bin[x]; const uint32_t g = (uint32_t)(x - slo[b]);
      if (g >= gcnt[b]) return;
      const int32_t *iv = gi + x * 3; const GSeq q{code + soff[b], slen_d[b]};
      for (int i = iv[1]; i < iv[2] - 5; i += 3) g_atomic_add(&hcnt[(size_t)b * 4096 + q.mer(iv[0], 6, i)], 1u);
    });
    x_hexamer_background(e, code, soff, slen_d, nbins, *std::max_element(seq_len.begin(), seq_len.begin() + nbins), d_hex.as<uint32_t>());
    std::vector<uint32_t> h_hex((size_t)nbins * 4096), h_hcnt((size_t)nbins * 4096);
    g_d2h(e, h_hex.data(), d_hex.p, h_hex.size() * 4); g_d2h(e, h_hcnt.data(), d_hcnt.p, h_hcnt.size() * 4); g_sync(e);
    std::vector<double> h_dc((size_t)nbins * 4096, 0.0);
    in.pfor(nbins, [&](size_t b) {
      if (!trained[b]) return;
      const uint32_t *H = h_hex.data() + b * 4096, *cn = h_hcnt.data() + b * 4096;
      const long long g = seq_len[b] > 5 ? 2ll * (seq_len[b] - 5) : 0;
      long long glob = 0; for (int i = 0; i < 4096; ++i) glob += cn[i];
      for (int i = 0; i < 4096; ++i) {
        int r = 0;
        for (int k = 0; k < 6; ++k) r |= (3 - ((i >> (2 * k)) & 3)) << (2 * (5 - k));
        const double bg = g ? (double)((long long)H[i] + (long long)H[r]) / (double)g : 0.0;
        const double prob = glob ? (cn[i] * 1.0) / (glob * 1.0) : 0.0;
        double v;
        if (prob == 0 && bg != 0) v = -5.0;
        else if (bg == 0) v = 0.0;
        else v = log(prob / bg);
        if (v > 5.0) v = 5.0;
        if (v < -5.0) v = -5.0;
        tr[b].gene_dc[i] = v; h_dc[b * 4096 + i] = v;
      }
    });
    tp("hexamer statistics");
    g_h2d(e, d_dc.p, h_dc.data(), h_dc.size() * 8);
    g_zero(e, rw_dev, 0, (size_t)nbins * 28 * 8);
    {
      GTimer t; t.begin(e);
      x_cscore(e, code, soff, slen_d, tn, d_dc.as<double>(), (uint32_t)NT[0]);
      x_rbs(e, code, soff, slen_d, tn, rw_dev, (uint32_t)NT[0]);
      out.ms_score += t.end(e);
    }
    coding_passes(tn, NT[0]);
    // ---- start-site model: Shine-Dalgarno bins (node.c: train_starts_sd); counts on the device, logarithms on the host ----
    uint32_t *cnt = d_cnt.as<uint32_t>(); double *rw = rw_dev, *tw = tw_dev, *sth = sth_dev;
    g_host_reserve(e, 0, npar * 8 + (size_t)nbins * TC_SIZE * 4 + 1024);
    uint32_t *h_cnt = g_host<uint32_t>(e, 0, (size_t)nbins * TC_SIZE);
    double *h_par = g_host<double>(e, 0, npar), *h_rw = h_par, *h_tw = h_par + (size_t)nbins * 28, *h_sth = h_tw + (size_t)nbins * 3;
    for (size_t k = 0; k < (size_t)nbins * 31; ++k) h_par[k] = 0.0;
    for (uint32_t b = 0; b < nbins; ++b) h_sth[b] = 35.0;
    const size_t cnt_bytes = (size_t)nbins * TC_SIZE * 4;
    std::vector<std::array<double, 3>> tbg(nbins);
    g_zero(e, d_cnt.p, 0, cnt_bytes);
    g_map(e, NT[0], [=] GLAM(size_t x) { if (tn.type[x] < G_STOP) g_count(&cnt[(size_t)tn.bin[x] * TC_SIZE + TC_TBG + tn.type[x]]); });
    g_down(e, h_cnt, cnt, cnt_bytes); g_sync(e);
    for (uint32_t b = 0; b < nbins; ++b) {
      double sum = 0.0;
      for (int i = 0; i < 3; ++i) { tbg[b][i] = (double)h_cnt[(size_t)b * TC_SIZE + TC_TBG + i]; sum += tbg[b][i]; }
      for (int i = 0; i < 3; ++i) tbg[b][i] = sum ? tbg[b][i] / sum : 0.0;
    }
    const uint32_t *ch = CH;
    for (int it = 0; it < 10; ++it) {
      g_up(e, rw_dev, h_par, npar * 8);
      g_zero(e, d_cnt.p, 0, cnt_bytes);
      const int last = it == 9;
      g_map(e, NT[0], [=] GLAM(size_t x) {
        const int t = tn.type[x];
        if (t == G_PAD) return;
        const uint32_t b = tn.bin[x]; const double *rwb = rw + (size_t)b * 28; uint32_t *cb = cnt + (size_t)b * TC_SIZE;
        if (t != G_STOP) { if (tn.edge[x] != 1) g_count(&cb[TC_RBG + best_rbs(tn.rbs0[x], tn.rbs1[x], rwb)]); return; }
        const double wt = stwt[b], *twb = tw + (size_t)b * 3;
        double best = 0.0; int bx = -1, brbs = 0, btype = 0; uint32_t by = 0;
        for_orf_starts(tn, ch, (uint32_t)x, [&](uint32_t y) {
          if (tn.edge[y] == 1) return;
          const int mr = best_rbs(tn.rbs0[y], tn.rbs1[y], rwb);
          const double v = tn.cscore[y] + wt * rwb[mr] + wt * twb[tn.type[y]];
          if (v >= best) { best = tn.cscore[y] + wt * rwb[mr]; best += wt * twb[tn.type[y]]; bx = 1; by = y; btype = tn.type[y]; brbs = mr; }
        });
        if (bx == 1 && best >= sth[b]) {
          g_count(&cb[TC_RREAL + brbs]); g_count(&cb[TC_TREAL + btype]);
          if (last) {
            const uint32_t s = tn.seq[by]; const GSeq q{code + soff[s], slen_d[s]};
            const int str = tn.strand[by], start = str == 1 ? tn.ndx[by] : q.slen - 1 - tn.ndx[by];
            int count = 0;
            for (int i = 1; i < 45; ++i) { if (i > 2 && i < 15) continue; if (start - i >= 0) g_count(&cb[TC_UPS + count * 4 + q.at(str, start - i)]); count++; }
          }
        }
      });
      g_down(e, h_cnt, cnt, cnt_bytes); g_sync(e);
      for (uint32_t b = 0; b < nbins; ++b) {
        if (!trained[b]) continue;
        const uint32_t *c = h_cnt + (size_t)b * TC_SIZE; GTrainH &t = tr[b];
        double rbg[28], rreal[28], treal[3], sum = 0.0;
        for (int j = 0; j < 28; ++j) { rbg[j] = (double)c[TC_RBG + j]; sum += rbg[j]; }
        for (int j = 0; j < 28; ++j) rbg[j] = sum ? rbg[j] / sum : 0.0;
        for (int j = 0; j < 28; ++j) rreal[j] = (double)c[TC_RREAL + j];
        for (int j = 0; j < 3; ++j) treal[j] = (double)c[TC_TREAL + j];
        sum = 0.0; for (int j = 0; j < 28; ++j) sum += rreal[j];
        if (sum == 0.0) for (int j = 0; j < 28; ++j) t.rbs_wt[j] = 0.0;
        else for (int j = 0; j < 28; ++j) {
          rreal[j] /= sum;
          t.rbs_wt[j] = rbg[j] != 0 ? log(rreal[j] / rbg[j]) : -4.0;
          if (t.rbs_wt[j] > 4.0) t.rbs_wt[j] = 4.0;
          if (t.rbs_wt[j] < -4.0) t.rbs_wt[j] = -4.0;
        }
        sum = host_update_type_wt(t, treal, tbg[b].data());
        if (sum <= (double)seq_n[b] / 2000.0) h_sth[b] /= 2.0;
        for (int j = 0; j < 28; ++j) h_rw[(size_t)b * 28 + j] = t.rbs_wt[j];
        for (int j = 0; j < 3; ++j) h_tw[(size_t)b * 3 + j] = t.type_wt[j];
        if (last) { for (int i = 0; i < 32; ++i) for (int j = 0; j < 4; ++j) t.ups_comp[i][j] = (double)c[TC_UPS + i * 4 + j]; host_ups_to_log(t); }
      }
    }
    for (uint32_t b = 0; b < nbins; ++b) if (trained[b]) host_determine_sd_usage(tr[b]);
    tp("Shine-Dalgarno training");
    // ---- bins that do not use Shine-Dalgarno sites: upstream motifs (node.c: train_starts_nonsd) ----
    std::vector<uint32_t> ns_bins; std::vector<int32_t> slot_of(nbins, -1);
    for (uint32_t b = 0; b < nbins; ++b) if (trained[b] && tr[b].uses_sd == 0) { slot_of[b] = (int32_t)ns_bins.size(); ns_bins.push_back(b); }
    if (!ns_bins.empty()) {
      const size_t nsl = ns_bins.size();
      // One block of weights going up -- [motif tables of the bins without Shine-Dalgarno sites][no-motif weight of every bin] -- and one
      // block of counters coming down -- [stage-0 background][per-bin counters][stage-0 real][stage 1-2 background][stage 1-2 real] -- so
      // that a round is two uploads, one clear, two kernels and one download.
      GBuf d_slot, d_mpar, d_mall;
      const size_t n_mpar = nsl * MOT_N + nbins, n_s0 = nsl * 4 * 4096, n_cnt = (size_t)nbins * TC_SIZE, n_mall = 2 * n_s0 + n_cnt + 2 * nsl * MOT_N;
      d_slot.ensure((size_t)nbins * 4); d_mpar.ensure(n_mpar * 8); d_mall.ensure(n_mall * 4);
      g_h2d(e, d_slot.p, slot_of.data(), (size_t)nbins * 4);
      const int32_t *slot = d_slot.as<int32_t>(); double *mw = d_mpar.as<double>(), *nm = mw + nsl * MOT_N;
      uint32_t *bg0 = d_mall.as<uint32_t>(), *cnt = bg0 + n_s0, *real0 = cnt + n_cnt, *mbg = real0 + n_s0, *mreal = mbg + nsl * MOT_N;       // (`cnt`: the rounds' own counters from here on)
      g_host_reserve(e, 1, n_mpar * 8 + n_mall * 4 + 4096);
      double *h_mpar = g_host<double>(e, 1, n_mpar), *h_nm = h_mpar + nsl * MOT_N;
      uint32_t *h_mall = g_host<uint32_t>(e, 1, n_mall), *h_bg0 = h_mall, *h_cnt = h_bg0 + n_s0, *h_real0 = h_cnt + n_cnt, *h_mbg = h_real0 + n_s0, *h_mreal = h_mbg + nsl * MOT_N;
      for (size_t k = 0; k < n_mpar; ++k) h_mpar[k] = 0.0;
      upstream_windows(tn, NT[0]);
      // the twenty rounds visit the nodes of THESE bins only: their 256-node blocks, one after the other (a bin's range starts at a multiple of 256)
      std::vector<uint32_t> h_blk;
      for (uint32_t b : ns_bins) for (uint32_t k = 0; k < (seq_n[b] + 255) / 256; ++k) h_blk.push_back(seq_lo[b] + k * 256);
      GBuf d_blk; d_blk.ensure(std::max<size_t>(1, h_blk.size()) * 4);
      g_h2d(e, d_blk.p, h_blk.data(), h_blk.size() * 4);
      const uint32_t *blk = d_blk.as<uint32_t>(); const size_t n_ns = h_blk.size() * 256;
      // the rounds' background words: every bin's nodes in eight parts, a workgroup each (x_motif_bg)
      std::vector<MotifPart> h_bgp;
      for (size_t k = 0; k < nsl; ++k) {
        // (a part holds fewer than 65536 nodes: the later rounds count in 16-bit halves)
        const uint32_t b = ns_bins[k], lo = seq_lo[b], n = seq_n[b], per = std::min<uint32_t>(65280u, std::max<uint32_t>(256u, ((n + 7) / 8 + 255) & ~255u));
        for (uint32_t a = 0; a < n; a += per) h_bgp.push_back(MotifPart{lo + a, std::min(n, a + per) + lo, (uint32_t)k, 0});
      }
      GBuf d_bgp; d_bgp.ensure(std::max<size_t>(1, h_bgp.size()) * sizeof(MotifPart));
      g_h2d(e, d_bgp.p, h_bgp.data(), h_bgp.size() * sizeof(MotifPart));
      std::vector<double> zbg0(nsl, 0.0);
      std::vector<std::vector<int>> h_good(nsl, std::vector<int>(MOT_N, 0));
      for (uint32_t b : ns_bins) { GTrainH &t = tr[b]; for (int j = 0; j < 3; ++j) t.type_wt[j] = 0.0; t.no_mot = 0.0; memset(t.ups_comp, 0, sizeof(t.ups_comp)); h_sth[b] = 35.0; for (int j = 0; j < 3; ++j) h_tw[(size_t)b * 3 + j] = 0.0; }
      for (int it = 0; it < 20; ++it) {
        const int stage = it < 4 ? 0 : it < 12 ? 1 : 2, last = it == 19;
        g_up(e, mw, h_mpar, n_mpar * 8);
        g_up(e, rw_dev, h_par, npar * 8);
        // counters: stage 0 keeps its background of the first round; the later stages clear (and fetch) everything behind it
        uint32_t *const c_lo = it == 0 ? bg0 : cnt; uint32_t *const c_hi = stage == 0 ? real0 + n_s0 : mreal + nsl * MOT_N;
        g_zero(e, c_lo, 0, (size_t)(c_hi - c_lo) * 4);
        const int count_bg0 = it == 0;
        // the best motif of every start node under the current weights, and the background counts
        g_map(e, n_ns, [=] GLAM(size_t xi) {
          const size_t x = (size_t)blk[xi >> 8] + (xi & 255);
          if (tn.type[x] >= G_STOP || tn.edge[x] == 1) return;
          const uint32_t b = tn.bin[x]; const int k = slot[b];
          if (k < 0) return;
          const uint32_t s = tn.seq[x]; const int sl = slen_d[s], strand = tn.strand[x], start = strand == 1 ? tn.ndx[x] : sl - 1 - tn.ndx[x];
          const unsigned long long upw = tn.upw[x];
          double ms; const uint32_t m = best_upstream_motif(mw + (size_t)k * MOT_N, nm[b], upw, start, stage, ms);
          tn.mot[x] = m; tn.mot_score[x] = ms;
          uint32_t *cb = cnt + (size_t)b * TC_SIZE;
          if (stage == 0 && !count_bg0) return;
          if (mot_len(m) == 0) { g_count(&cb[TC_ZBG]); return; }
          // (the background WORDS of every stage are counted by x_motif_bg below, workgroup by workgroup in LDS: the first round's depend on
          //  the windows only, the later rounds' on the motif just stored)
        });
        if (count_bg0 || stage > 0) x_motif_bg(e, stage, tn, slen_d, d_bgp.as<MotifPart>(), (uint32_t)h_bgp.size(), stage == 0 ? bg0 : mbg);
        // the best start of every open reading frame, and the counts of the ones above the threshold
        g_map(e, n_ns, [=] GLAM(size_t xi) {
          const size_t x = (size_t)blk[xi >> 8] + (xi & 255);
          if (tn.type[x] != G_STOP) return;
          const uint32_t b = tn.bin[x]; const int k = slot[b];
          if (k < 0) return;
          const double wt = stwt[b], *twb = tw + (size_t)b * 3; uint32_t *cb = cnt + (size_t)b * TC_SIZE;
          double best = 0.0; int bx = -1; uint32_t by = 0;
          for_orf_starts(tn, ch, (uint32_t)x, [&](uint32_t y) {
            if (tn.edge[y] == 1) return;
            const double v = tn.cscore[y] + wt * tn.mot_score[y] + wt * twb[tn.type[y]];
            if (v >= best) { best = tn.cscore[y] + wt * tn.mot_score[y]; best += wt * twb[tn.type[y]]; bx = 1; by = y; }
          });
          if (bx != 1 || !(best >= sth[b])) return;
          g_count(&cb[TC_NGENES]); g_count(&cb[TC_TREAL + tn.type[by]]);
          const uint32_t s = tn.seq[by]; const GSeq q{code + soff[s], slen_d[s]};
          const int str = tn.strand[by], start = str == 1 ? tn.ndx[by] : q.slen - 1 - tn.ndx[by];
          const uint32_t m = tn.mot[by]; const unsigned long long upw = tn.upw[by];
          if (mot_len(m) == 0) g_count(&cb[TC_ZREAL]);
          else if (stage == 0) {
            for (int i = 3; i >= 0; --i) for (int j = start - 18 - i; j <= start - 6 - i; ++j) { if (j < 0) continue; g_atomic_add(&real0[((size_t)k * 4 + i) * 4096 + upw_mer(upw, start, i + 3, j)], 1u); }
          } else if (stage == 1) {
            uint32_t *tab = mreal + (size_t)k * MOT_N; const int ml = mot_len(m), sp = mot_spacer(m);
            g_atomic_add(&tab[((size_t)(ml - 3) * 4 + mot_spacendx(m)) * 4096 + mot_ndx(m)], 1u);
            for (int i = 0; i < ml - 3; ++i) for (int j = start - sp - ml; j <= start - sp - (i + 3); ++j) {
              if (j < 0) continue;
              g_atomic_add(&tab[((size_t)i * 4 + spacer_ndx(j, start, i)) * 4096 + upw_mer(upw, start, i + 3, j)], 1u);
            }
          } else g_atomic_add(&mreal[(size_t)k * MOT_N + ((size_t)(mot_len(m) - 3) * 4 + mot_spacendx(m)) * 4096 + mot_ndx(m)], 1u);
          if (last) {
            int count = 0;
            for (int i = 1; i < 45; ++i) { if (i > 2 && i < 15) continue; if (start - i >= 0) g_count(&cb[TC_UPS + count * 4 + q.at(str, start - i)]); count++; }
          }
        });
        g_down(e, h_mall + (c_lo - bg0), c_lo, (size_t)(c_hi - c_lo) * 4);
        g_sync(e);
        in.pfor(nsl, [&](size_t k) {
          const uint32_t b = ns_bins[k]; GTrainH &t = tr[b]; const uint32_t *c = h_cnt + (size_t)b * TC_SIZE;
          std::vector<double> vbg(MOT_N), vreal(MOT_N); std::vector<int> &good = h_good[k];
          if (it == 0) zbg0[k] = (double)c[TC_ZBG];
          double zbg = stage == 0 ? zbg0[k] : (double)c[TC_ZBG], zreal = (double)c[TC_ZREAL];
          const double ngenes = (double)c[TC_NGENES];
          if (stage == 0) {
            // (stage 0 counts every word of a start once per spacer class: the four classes hold the same numbers)
            for (int i = 0; i < 4; ++i) for (int q = 0; q < 4; ++q) for (int w = 0; w < 4096; ++w) {
              vbg[((size_t)i * 4 + q) * 4096 + w] = (double)h_bg0[((size_t)k * 4 + i) * 4096 + w]; vreal[((size_t)i * 4 + q) * 4096 + w] = (double)h_real0[((size_t)k * 4 + i) * 4096 + w];
            }
          } else for (size_t i = 0; i < MOT_N; ++i) { vbg[i] = (double)h_mbg[k * MOT_N + i]; vreal[i] = (double)h_mreal[k * MOT_N + i]; }
          double sum = zbg;
          for (double x : vbg) sum += x;
          if (sum != 0.0) { for (double &x : vbg) x /= sum; zbg /= sum; }
          double treal[3];
          for (int j = 0; j < 3; ++j) treal[j] = (double)c[TC_TREAL + j];
          if (stage < 2) host_build_coverage_map(vreal, good, ngenes);
          sum = zreal;
          for (double x : vreal) sum += x;
          double *mwt = h_mpar + k * MOT_N;
          if (sum == 0.0) { std::fill(mwt, mwt + MOT_N, 0.0); t.no_mot = 0.0; }
          else {
            for (size_t q = 0; q < MOT_N; ++q) {
              if (good[q] == 0) { zreal += vreal[q]; zbg += vreal[q]; vreal[q] = 0.0; vbg[q] = 0.0; }
              vreal[q] /= sum;
              double v = vbg[q] != 0 ? log(vreal[q] / vbg[q]) : -4.0;
              if (v > 4.0) v = 4.0;
              if (v < -4.0) v = -4.0;
              mwt[q] = v;
            }
            zreal /= sum;
            t.no_mot = zbg != 0 ? log(zreal / zbg) : -4.0;
            if (t.no_mot > 4.0) t.no_mot = 4.0;
            if (t.no_mot < -4.0) t.no_mot = -4.0;
          }
          sum = host_update_type_wt(t, treal, tbg[b].data());
          if (sum <= (double)seq_n[b] / 2000.0) h_sth[b] /= 2.0;
          h_nm[b] = t.no_mot;
          for (int j = 0; j < 3; ++j) h_tw[(size_t)b * 3 + j] = t.type_wt[j];
          if (last) { for (int i = 0; i < 32; ++i) for (int j = 0; j < 4; ++j) t.ups_comp[i][j] = (double)c[TC_UPS + i * 4 + j]; host_ups_to_log(t); t.mot_wt.assign(mwt, mwt + MOT_N); }
        });
      }
      tp("upstream-motif training");
    }
  }
  tp("training done");

  // =====================================================  gene finding, contig by contig  =====================================================
  if (NT[1]) {
    const Nodes fn = FS.dev();
    // tables of the trained bins
    std::vector<double> h_dc((size_t)nbins * 4096, 0.0), h_rw((size_t)nbins * 28, 0.0), h_tw((size_t)nbins * 3, 0.0), h_ups((size_t)nbins * 128, 0.0), h_nm(nbins, 0.0);
    std::vector<uint8_t> h_sd(nbins, 0); std::vector<int32_t> slot_of(nbins, -1); std::vector<uint32_t> ns_bins;
    for (uint32_t b = 0; b < nbins; ++b) {
      if (!trained[b]) continue;
      const GTrainH &t = tr[b];
      memcpy(h_dc.data() + (size_t)b * 4096, t.gene_dc, sizeof(double) * 4096); memcpy(h_rw.data() + (size_t)b * 28, t.rbs_wt, sizeof(double) * 28);
      for (int j = 0; j < 3; ++j) h_tw[(size_t)b * 3 + j] = t.type_wt[j];
      for (int i = 0; i < 32; ++i) for (int j = 0; j < 4; ++j) h_ups[(size_t)b * 128 + i * 4 + j] = t.ups_comp[i][j];
      h_nm[b] = t.no_mot; h_sd[b] = (uint8_t)t.uses_sd;
      if (t.uses_sd != 1) { slot_of[b] = (int32_t)ns_bins.size(); ns_bins.push_back(b); }
      out.bin_uses_sd[b] = (uint8_t)t.uses_sd; out.bin_gc[b] = t.gc;
    }
    for (uint32_t b = 0; b < nbins; ++b) if (!trained[b]) { out.bin_uses_sd[b] = 0; out.bin_gc[b] = tr[b].gc; }
    GBuf d_ups, d_nm, d_sd, d_slot, d_mw;
    d_ups.ensure((size_t)nbins * 128 * 8); d_nm.ensure((size_t)nbins * 8); d_sd.ensure(nbins); d_slot.ensure((size_t)nbins * 4); d_mw.ensure(std::max<size_t>(1, ns_bins.size()) * MOT_N * 8);
    g_h2d(e, d_dc.p, h_dc.data(), h_dc.size() * 8); g_h2d(e, rw_dev, h_rw.data(), h_rw.size() * 8); g_h2d(e, tw_dev, h_tw.data(), h_tw.size() * 8); g_h2d(e, d_ups.p, h_ups.data(), h_ups.size() * 8);
    g_h2d(e, d_nm.p, h_nm.data(), (size_t)nbins * 8); g_h2d(e, d_sd.p, h_sd.data(), nbins); g_h2d(e, d_slot.p, slot_of.data(), (size_t)nbins * 4);
    for (size_t k = 0; k < ns_bins.size(); ++k) {
      const std::vector<double> &m = tr[ns_bins[k]].mot_wt;
      if (m.size() == MOT_N) g_h2d(e, d_mw.as<double>() + k * MOT_N, m.data(), MOT_N * 8); else g_zero(e, d_mw.as<double>() + k * MOT_N, 0, MOT_N * 8);
    }
    const double *rw = rw_dev, *tw = tw_dev, *ups = d_ups.as<double>(), *nm = d_nm.as<double>(), *mw = d_mw.as<double>();
    const uint8_t *uses_sd = d_sd.as<uint8_t>(); const int32_t *slot = d_slot.as<int32_t>();
    {
      GTimer t; t.begin(e);
      x_cscore(e, code, soff, slen_d, fn, d_dc.as<double>(), (uint32_t)NT[1]);
      x_rbs(e, code, soff, slen_d, fn, rw, (uint32_t)NT[1]);
      out.ms_score += t.end(e);
    }
    upstream_windows(fn, NT[1]);
    coding_passes(fn, NT[1]);
    // score_nodes behind the coding score: GC content, upstream motif, start scores (node.c: score_nodes); the new edge flags go to a second column
    uint8_t *edge2 = FS.b_edge2.as<uint8_t>();
    g_map(e, NT[1], [=] GLAM(size_t x) {
      const int ty = fn.type[x];
      if (ty == G_PAD) return;
      edge2[x] = fn.edge[x];
      if (ty == G_STOP) return;
      const uint32_t b = fn.bin[x], s = fn.seq[x]; const GSeq q{code + soff[s], slen_d[s]}; const int sl = q.slen;
      const NView V{fn, slo[s], (int)sn[s]}; const int i = (int)(x - V.lo), nn = V.nn;
      const int ndx = fn.ndx[x], sv = fn.sv[x], strand = fn.strand[x]; const double st_wt = stwt[b];
      // (score_nodes looks for Shine-Dalgarno sites only in organisms that use them: elsewhere every node keeps bin 0)
      const bool sdm = uses_sd[b] == 1;
      const int rb0 = sdm ? fn.rbs0[x] : 0, rb1 = sdm ? fn.rbs1[x] : 0;
      fn.rbs0[x] = (uint8_t)rb0; fn.rbs1[x] = (uint8_t)rb1;
      { // calc_orf_gc
        const int a = strand == 1 ? ndx : sv, z = strand == 1 ? sv + 2 : ndx;
        const int a0 = a < 0 ? 0 : a, z0 = z < sl - 1 ? z : sl - 1;
        int g = 0;
        if (z0 >= a0) g = (int)(plane_rank(pg, gcw, soff[s] + (uint64_t)z0 + 1) - plane_rank(pg, gcw, soff[s] + (uint64_t)a0));
        fn.gc_cont[x] = (double)g / (double)(abs(sv - ndx) + 3);
      }
      const int start = strand == 1 ? ndx : sl - 1 - ndx;
      double mot_score = 0.0; uint32_t mot = 0;
      if (!sdm && fn.edge[x] != 1) { mot = best_upstream_motif(mw + (size_t)slot[b] * MOT_N, nm[b], fn.upw[x], start, 2, mot_score); fn.mot[x] = mot; fn.mot_score[x] = mot_score; }
      int edge_gene = 0, edge = fn.edge[x];
      double tscore, rscore, uscore;
      if (edge == 1) edge_gene++;
      if ((strand == 1 && !q.is_stop(1, sv, tt)) || (strand == -1 && !q.is_stop(-1, sl - 1 - sv, tt))) edge_gene++;
      if (edge == 1) { tscore = EDGE_BONUS * st_wt / edge_gene; uscore = 0.0; rscore = 0.0; }
      else {
        tscore = tw[(size_t)b * 3 + ty] * st_wt;
        const double rbs1 = rw[(size_t)b * 28 + rb0], rbs2 = rw[(size_t)b * 28 + rb1], sd_score = dmaxd(rbs1, rbs2) * st_wt;
        if (sdm) rscore = sd_score;
        else { rscore = st_wt * mot_score; if (rscore < sd_score && nm[b] > -0.5) rscore = sd_score; }
        { // score_upstream_composition
          int count = 0; uscore = 0.0; const double *uc = ups + (size_t)b * 128;
          for (int k = 1; k < 45; ++k) { if (k > 2 && k < 15) continue; if (start - k < 0) continue; uscore += 0.4 * st_wt * uc[count * 4 + q.at(strand, start - k)]; count++; }
        }
        if (ndx <= 2 && strand == 1) uscore += EDGE_UPS * st_wt;
        else if (ndx >= sl - 3 && strand == -1) uscore += EDGE_UPS * st_wt;
        else if (i < 500 && strand == 1) {
          for (int j = i - 1; j >= 0; --j) {
            // (the nodes before this one have been through this loop in the reference: a start at the sequence's edge is an edge node by then)
            const bool ej = V.edge(j) == 1 || (!V.stop(j) && ((V.ndx(j) <= 2 && V.strand(j) == 1) || (V.ndx(j) >= sl - 3 && V.strand(j) == -1)));
            if (ej && sv == V.sv(j)) { uscore += EDGE_UPS * st_wt; break; }
          }
        } else if (i >= nn - 500 && strand == -1) {
          for (int j = i + 1; j < nn; ++j) if (V.edge(j) == 1 && sv == V.sv(j)) { uscore += EDGE_UPS * st_wt; break; }
        }
      }
      if (((ndx <= 2 && strand == 1) || (ndx >= sl - 3 && strand == -1)) && edge == 0) {
        edge_gene++; edge = 1; tscore = 0.0; uscore = EDGE_BONUS * st_wt / edge_gene; rscore = 0.0;
      }
      if (edge == 0 && edge_gene == 1) uscore -= 0.5 * EDGE_BONUS * st_wt;
      if (edge_gene == 0 && abs(ndx - sv) < 250) {
        const double negf = 250.0 / (float)abs(ndx - sv), posf = (float)abs(ndx - sv) / 250.0;
        if (rscore < 0) rscore *= negf;
        if (uscore < 0) uscore *= negf;
        if (tscore < 0) tscore *= negf;
        if (rscore > 0) rscore *= posf;
        if (uscore > 0) uscore *= posf;
        if (tscore > 0) tscore *= posf;
      }
      double sscore = tscore + rscore + uscore;
      const double cs = fn.cscore[x];
      if (cs < 0.0) { if (edge_gene > 0 && edge == 0) sscore -= st_wt; else sscore -= 0.5; }
      fn.tscore[x] = tscore; fn.rscore[x] = rscore; fn.uscore[x] = uscore; fn.sscore[x] = sscore; edge2[x] = (uint8_t)edge;
    });
    Nodes fn2 = fn; fn2.edge = edge2;                                  // (score_nodes turns starts at the sequence edges into edge nodes)
    overlapping_starts(fn2, NT[1], 1);
    dp_windows(fn2, NT[1]);
    g_map(e, NT[1], [=] GLAM(size_t x) { if (fn2.type[x] != G_PAD) fn2.csc[x] = fn2.cscore[x] + fn2.sscore[x]; });
    tp("node scores");
    run_dp(fn2, nbins, ncontigs, 1, out.ms_dp_find);
    tp("final dp done");
    // trace-back, bad genes, gene list, start tweaks: a thread per contig; the genes of a contig sit in its own node slots
    GBuf d_gl, d_gc2, d_pend;
    d_gl.ensure(std::max<size_t>(NT[1], 64) * sizeof(GeneSlot)); d_gc2.ensure((size_t)(ncontigs + 1) * 4); d_pend.ensure((size_t)(ncontigs + 1) * 4);
    GeneSlot *gl = reinterpret_cast<GeneSlot *>(d_gl.p); uint32_t *gcn = d_gc2.as<uint32_t>(); int32_t *pend = d_pend.as<int32_t>();
    {
      const uint32_t nb = nbins;
      x_path_ends(e, fn2, slo + nb, sn + nb, ncontigs, pend);
      g_map_waves(e, ncontigs, [=] GLAM(size_t c) {
        const uint32_t s = nb + (uint32_t)c;
        gcn[c] = 0;
        if (sn[s] == 0) return;
        const NView V{fn2, slo[s], (int)sn[s]}; const double st_wt = stwt[sbin[s]];
        const int ip = walk_dprog_finish(V, pend[c]);
        walk_eliminate_bad_genes(V, ip, st_wt);
        const int ng = walk_add_genes(V, ip, gl + V.lo);
        walk_tweak_final_starts(V, gl + V.lo, ng, st_wt);
        gcn[c] = (uint32_t)ng;
      });
    }
    x_scan_u32(e, gcn, ncontigs, d_scan);
    std::vector<uint32_t> h_goff(ncontigs + 1);
    g_d2h(e, h_goff.data(), gcn, (size_t)(ncontigs + 1) * 4); g_sync(e);
    const size_t ngenes = h_goff[ncontigs];
    tp("genes picked, starts tweaked");
    // ---- records ----
    GBuf d_ri, d_rd, d_pl, d_prot;
    const size_t ng64 = std::max<size_t>(ngenes, 64);
    d_ri.ensure(ng64 * 12 * 4); d_rd.ensure(ng64 * 7 * 8); d_pl.ensure((ng64 + 2) * 4);
    int32_t *ri = reinterpret_cast<int32_t *>(d_ri.p); double *rd = d_rd.as<double>(); uint32_t *pl = d_pl.as<uint32_t>();
    {
      const uint32_t nb = nbins, nc = ncontigs;
      g_map(e, NT[1], [=] GLAM(size_t x) {
        if (fn2.type[x] == G_PAD) return;
        const uint32_t s = fn2.seq[x], c = s - nb; const uint32_t g = (uint32_t)(x - slo[s]);
        const uint32_t g0 = gcn[c], g1 = c + 1 <= nc ? gcn[c + 1] : g0;
        if (g >= g1 - g0) return;
        const size_t r = (size_t)g0 + g; const GeneSlot G = gl[x]; const uint32_t lo = slo[s], b = fn2.bin[x];
        const uint32_t a = lo + (uint32_t)G.start_ndx, z = lo + (uint32_t)G.stop_ndx; const double st_wt = stwt[b];
        const int strand = fn2.strand[a], sedge = fn2.edge[a], pedge = fn2.edge[z];
        const int pleft = strand == 1 ? sedge : pedge, pright = strand == 1 ? pedge : sedge;
        const double rbs1 = rw[(size_t)b * 28 + fn2.rbs0[a]] * st_wt, rbs2 = rw[(size_t)b * 28 + fn2.rbs1[a]] * st_wt;
        const uint32_t m = fn2.mot[a]; const double msc = fn2.mot_score[a];
        int rb = -1, ml = 0, mx = 0, ms = 0;
        if (uses_sd[b] == 1) rb = rbs1 > rbs2 ? fn2.rbs0[a] : fn2.rbs1[a];
        else if (nm[b] > -0.5 && rbs1 > rbs2 && rbs1 > msc * st_wt) rb = fn2.rbs0[a];
        else if (nm[b] > -0.5 && rbs2 >= rbs1 && rbs2 > msc * st_wt) rb = fn2.rbs1[a];
        else { ml = mot_len(m); mx = mot_ndx(m); ms = mot_spacer(m); }
        int32_t *o = ri + r * 12;
        o[0] = (int32_t)b; o[1] = (int32_t)c; o[2] = G.begin; o[3] = G.end; o[4] = strand; o[5] = sedge ? 3 : fn2.type[a]; o[6] = pleft; o[7] = pright; o[8] = rb; o[9] = ml; o[10] = mx; o[11] = ms;
        double *d = rd + r * 7;
        d[0] = fn2.gc_cont[a]; d[1] = fn2.cscore[a]; d[2] = fn2.sscore[a]; d[3] = fn2.rscore[a]; d[4] = fn2.uscore[a]; d[5] = fn2.tscore[a]; d[6] = fn2.cscore[a] + fn2.sscore[a];
        const int sl = slen_d[s];
        const int pb = strand == 1 ? G.begin - 1 : sl - G.end, pe = strand == 1 ? G.end - 1 : sl - G.begin;
        pl[r] = pe - pb >= 2 ? (uint32_t)((pe - pb - 2) / 3 + 1) : 0u;
      });
    }
    if (2 * bases / 3 + ngenes >= 0xfffffff0ull) g_fail("more than 2^32 amino acids in one gene-calling batch");      // (32-bit protein offsets)
    x_scan_u32(e, pl, ngenes, d_scan);
    std::vector<uint32_t> h_pl(ngenes + 1);
    g_d2h(e, h_pl.data(), pl, (ngenes + 1) * 4); g_sync(e);
    const size_t nprot = h_pl[ngenes];
    d_prot.ensure(nprot + 64);
    char *prot = reinterpret_cast<char *>(d_prot.p);
    {
      const uint32_t nb = nbins; const int ttab = tt;
      g_map(e, ngenes, [=] GLAM(size_t r) {
        const int32_t *o = ri + r * 12; const uint32_t s = nb + (uint32_t)o[1]; const GSeq q{code + soff[s], slen_d[s]};
        const int strand = o[4], sl = q.slen, begin = o[2], end = o[3];
        const int pb = strand == 1 ? begin - 1 : sl - end, pe = strand == 1 ? end - 1 : sl - begin;
        const bool partial5 = strand == 1 ? o[6] : o[7];
        char *dst = prot + pl[r];
        for (int i = pb; i + 2 <= pe; i += 3) { char a = amino(q, strand, i, ttab); if (i == pb && !partial5) a = 'M'; *dst++ = a; }
      });
    }
    std::vector<int32_t> h_ri(ngenes * 12); std::vector<double> h_rd(ngenes * 7);
    out.prot.resize(nprot);
    g_d2h(e, h_ri.data(), ri, ngenes * 12 * 4); g_d2h(e, h_rd.data(), rd, ngenes * 7 * 8); g_d2h(e, &out.prot[0], prot, nprot); g_sync(e);
    out.bin.resize(ngenes); out.contig.resize(ngenes); out.begin.resize(ngenes); out.end.resize(ngenes); out.strand.resize(ngenes); out.start_type.resize(ngenes); out.partial_left.resize(ngenes);
    out.partial_right.resize(ngenes); out.rbs_bin.resize(ngenes); out.mot_len.resize(ngenes); out.mot_ndx.resize(ngenes); out.mot_spacer.resize(ngenes); out.gc_cont.resize(ngenes); out.conf.resize(ngenes);
    out.score.resize(ngenes); out.cscore.resize(ngenes); out.sscore.resize(ngenes); out.rscore.resize(ngenes); out.uscore.resize(ngenes); out.tscore.resize(ngenes); out.prot_off.resize(ngenes + 1);
    for (size_t r = 0; r < ngenes; ++r) {
      const int32_t *o = h_ri.data() + r * 12; const double *d = h_rd.data() + r * 7;
      out.bin[r] = (uint32_t)o[0]; out.contig[r] = (uint32_t)o[1]; out.begin[r] = o[2]; out.end[r] = o[3]; out.strand[r] = (int8_t)o[4]; out.start_type[r] = (uint8_t)o[5];
      out.partial_left[r] = (uint8_t)o[6]; out.partial_right[r] = (uint8_t)o[7]; out.rbs_bin[r] = o[8]; out.mot_len[r] = o[9]; out.mot_ndx[r] = o[10]; out.mot_spacer[r] = o[11];
      out.gc_cont[r] = d[0]; out.cscore[r] = d[1]; out.sscore[r] = d[2]; out.rscore[r] = d[3]; out.uscore[r] = d[4]; out.tscore[r] = d[5]; out.score[r] = d[6];
      out.conf[r] = host_confidence(d[6], tr[o[0]].st_wt);
      out.prot_off[r] = h_pl[r];
      out.bin_coding[o[0]] += (uint64_t)(o[3] - o[2] + 1);
    }
    out.prot_off[ngenes] = nprot;
    tp("records and proteins");
  } else {
    for (uint32_t b = 0; b < nbins; ++b) { out.bin_uses_sd[b] = (uint8_t)(trained[b] ? tr[b].uses_sd : 0); out.bin_gc[b] = tr[b].gc; }
    g_sync(e);          // (the buffers of this function go back to the block cache when it returns: nothing of this call may still be queued)
  }
}

}  // namespace gene
}  // namespace ckm
