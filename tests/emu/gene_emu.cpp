// gene_emu.cpp -- TEST INFRASTRUCTURE: the gene-calling pipeline of libcheckm_hip.so (checkm_amd/csrc/gene_pipe.h, gene_dev.h) compiled by
// g++ against a HOST executor, so that the CPU test suite (-m "not gpu") can diff the pipeline's logic -- rank-based node order, chain
// arrays, range-count GC frames, per-frame sweeps as per-ORF walks, the count / logarithm split of the training loops, record building --
// against the gene oracle without a GPU.  Nothing in checkm_amd loads this library; the product has no CPU path (tests/test_boundary.py).
// The cooperating kernels (wave ballots, LDS, ordered sums) are restated here as scalar loops with the semantics kernels_genes.hip gives them.
#define CKM_GENE_EMU 1
#include "../../checkm_amd/csrc/gene_pipe.h"

namespace ckm {
namespace gene {

void x_orf_flags(GExec &, const uint8_t *ascii, unsigned long long *planes, uint64_t body) {
  const uint64_t nwin = body / 64;
  for (uint64_t k = 0; k < 8 * nwin; ++k) planes[k] = 0;
  auto B = [&](long long p) -> int {           // 0 A 1 C 2 G 3 T, -1 other (the buffer has 64 readable bytes of 'N' on either side)
    switch (ascii[p]) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': case 'U': case 'u': return 3; default: return -1; }
  };
  for (long long i = 0; i < (long long)body; ++i) {
    const int a = B(i), b = B(i + 1), c = B(i + 2);
    const unsigned long long bit = 1ull << (i & 63); const uint64_t w = (uint64_t)i >> 6;
    const bool taag = a == 3 && b == 0 && (c == 0 || c == 2), tga = a == 3 && b == 2 && c == 0;
    if (taag || tga) planes[0 * nwin + w] |= bit;
    if (taag) planes[1 * nwin + w] |= bit;
    const bool tg = b == 3 && c == 2;
    const int st = tg ? (a == 0 ? 1 : a == 2 ? 2 : a == 3 ? 3 : 0) : 0;          // 1 ATG 2 GTG 3 TTG
    if (st & 1) planes[2 * nwin + w] |= bit;
    if (st & 2) planes[3 * nwin + w] |= bit;
    // reverse strand: the codon whose first base is the complement of base i (its bases are i, i-1, i-2)
    const int r0 = B(i) < 0 ? -1 : 3 - B(i), r1 = B(i - 1) < 0 ? -1 : 3 - B(i - 1), r2 = B(i - 2) < 0 ? -1 : 3 - B(i - 2);
    const bool rtaag = r0 == 3 && r1 == 0 && (r2 == 0 || r2 == 2), rtga = r0 == 3 && r1 == 2 && r2 == 0;
    if (rtaag || rtga) planes[4 * nwin + w] |= bit;
    if (rtaag) planes[5 * nwin + w] |= bit;
    const bool rtg = r1 == 3 && r2 == 2;
    const int rst = rtg ? (r0 == 0 ? 1 : r0 == 2 ? 2 : r0 == 3 ? 3 : 0) : 0;
    if (rst & 1) planes[6 * nwin + w] |= bit;
    if (rst & 2) planes[7 * nwin + w] |= bit;
  }
}

void x_scan_u32(GExec &, uint32_t *a, size_t n, GBuf &) {
  uint32_t run = 0;
  for (size_t i = 0; i < n; ++i) { const uint32_t x = a[i]; a[i] = run; run += x; }
  a[n] = run;
}

void x_chain(GExec &, const ChainArgs &a) {
  OrfRec *rec = reinterpret_cast<OrfRec *>(a.rec);
  for (uint32_t q = 0; q < a.nsc; ++q) {
    const SubChain sc = a.sc[q];
    const uint32_t si = sc.seq; const int rev = sc.rev, frame = sc.frame, slen = a.seq_len[si];
    if (slen < 3) continue;
    const uint64_t base = a.seq_off[si], nwin = a.nwin;
    const unsigned long long *p_stop = a.planes + (uint64_t)((rev ? 4 : 0) + (a.tt4 ? 1 : 0)) * nwin;
    const unsigned long long *p_lo = a.planes + (uint64_t)((rev ? 4 : 0) + 2) * nwin, *p_hi = a.planes + (uint64_t)((rev ? 4 : 0) + 3) * nwin;
    unsigned long long *node_plane = a.node_planes + (uint64_t)((si < a.nbins ? 0 : 2) + (rev ? 1 : 0)) * nwin;
    uint32_t t = 0;
    auto emit = [&](int ndx_s, int type, int sv_s, int edge) {
      const int ndx = rev ? slen - 1 - ndx_s : ndx_s; const uint64_t g = base + (uint64_t)ndx;
      node_plane[g >> 6] |= 1ull << (g & 63);
      const unsigned long long k = (*a.nrec)++;
      if (k < a.cap) { OrfRec nd; nd.seq = si; nd.type = (uint8_t)type; nd.strand_rev = (uint8_t)rev; nd.edge = (uint8_t)edge; nd.pad = 0; nd.ndx = ndx; nd.sv = rev ? slen - 1 - sv_s : sv_s; rec[k] = nd; a.rec_t[k] = t; a.rec_c[k] = sc.chain; }
      t++;
    };
    int last = sc.top; bool last_real = sc.after_stop, saw = false, any_stop = sc.after_stop;
    for (int j = sc.after_stop ? sc.top - 3 : sc.top; j >= sc.bottom; j -= 3) {
      const uint64_t pos = base + (uint64_t)(rev ? slen - 1 - j : j); const uint64_t wi = pos >> 6; const int bit = (int)(pos & 63);
      const bool is_stop = (p_stop[wi] >> bit) & 1ull;
      const int st = (int)(((p_lo[wi] >> bit) & 1ull) | (((p_hi[wi] >> bit) & 1ull) << 1)) - 1;
      const int mind = any_stop ? 90 : 60;
      bool start_node = false, edge_node = false;
      if (!is_stop && last < slen) {
        if (st >= 0 && last - j + 3 >= mind) start_node = true;
        else if (j <= 2 && (last - j) > 60) edge_node = true;
      }
      if ((start_node || edge_node) && a.r50) {
        const int x = rev ? slen - 1 - last : j, y = rev ? slen - 1 - j : last; const int x0 = x - 49 > 0 ? x - 49 : 0;
        if (plane_rank(a.pr50, a.r50, base + (uint64_t)y + 1) != plane_rank(a.pr50, a.r50, base + (uint64_t)x0)) start_node = edge_node = false;
      }
      if (start_node) { emit(j, st, last, 0); saw = true; }
      if (edge_node) { emit(j, 0, last, 1); saw = true; }
      if (is_stop) { if (saw) emit(last, 3, j, last_real ? 0 : 1); last = j; last_real = true; any_stop = true; saw = false; }
    }
    if (saw) emit(last, 3, frame - 6, last_real ? 0 : 1);
    a.chain_cnt[sc.chain] = t;
  }
}

void x_gc_bias(GExec &, const Nodes &nd, const uint32_t *seq_lo, const uint32_t *seq_n, uint32_t nbins, double *bias) {
  for (uint32_t b = 0; b < nbins; ++b) {
    double acc[3] = {0, 0, 0};
    if (seq_n[b] == 0) { bias[b * 3] = bias[b * 3 + 1] = bias[b * 3 + 2] = 0.0; continue; }
    for (uint32_t i = 0; i < seq_n[b]; ++i) if (nd.type[seq_lo[b] + i] < G_STOP) acc[nd.gcb_cls[seq_lo[b] + i]] += nd.gcb_term[seq_lo[b] + i];
    const double tot = acc[0] + acc[1] + acc[2];
    for (int k = 0; k < 3; ++k) { acc[k] *= (3.0 / tot); bias[b * 3 + k] = acc[k]; }
  }
}

struct DpFlat {
  const Nodes &nd; uint32_t first; int flag;
  DpNode node(int rel) const { const uint32_t g = first + (uint32_t)rel; DpNode n; n.ndx = nd.ndx[g]; n.sv = nd.sv[g]; n.strand = nd.strand[g]; n.stop = nd.type[g] == 3; return n; }
  int ndx(int rel) const { return nd.ndx[first + (uint32_t)rel]; }
  int ndx_any(int rel) const { return ndx(rel); }
  DpNode node3(int rel) const { return node(rel); }
  double val3(int rel) const { return val(rel); }
  int star(int rel, int f) const { return nd.star[(size_t)(first + (uint32_t)rel) * 3 + f]; }
  double val(int rel) const { return flag == 0 ? nd.gcb[first + (uint32_t)rel] : nd.csc[first + (uint32_t)rel]; }
  double score(int rel) const { return nd.score[first + (uint32_t)rel]; }
  int tb(int rel) const { return nd.traceb[first + (uint32_t)rel]; }
  double rscore(int rel) const { return nd.rscore[first + (uint32_t)rel]; }
  double uscore(int rel) const { return nd.uscore[first + (uint32_t)rel]; }
};
// The sweep as the device kernel takes it (kernels_genes.hip: gene_dp_kernel): 64 nodes at a time.  For a block, (A) every node's candidates
// BEFORE the block are final and scored in any order; the connection of a candidate INSIDE the block to it is scored WITHOUT the
// candidate's own score (dp_connection_s) into a 64 x 64 table -- except the pairs whose connection reads the candidate's predecessor
// (dp_pair_dynamic); a forward stop's / reverse start's candidates begin at dp_pos_floor.  (B) the block's nodes are then resolved one
// after the other: node t is final once t - 1 is, and is offered to the nodes behind it with its table entry or, for the dynamic pairs,
// the whole connection function.
struct DpFlatBlk : DpFlat {
  int blk0;
  DpFlatBlk(const Nodes &n, uint32_t f, int fl, int b0) : DpFlat{n, f, fl}, blk0(b0) {}
  int tb(int rel) const { return rel >= blk0 ? 0 : DpFlat::tb(rel); }      // (a node of the block: "has a predecessor" is decided in (B))
};
void x_dp(GExec &, const Nodes &nd, const uint32_t *seq_lo, const uint32_t *seq_n, const uint32_t *seq_bin, const double *st_wt, uint32_t nseq, int flag) {
  constexpr int B = 64;
  std::vector<double> table(B * B), best(B); std::vector<int> bj(B), bm(B), cls(B);
  for (uint32_t s = 0; s < nseq; ++s) {
    const DpFlat S{nd, seq_lo[s], flag}; const int nn = (int)seq_n[s]; const double w = st_wt[seq_bin[s]];
    for (int i0 = 0; i0 < nn; i0 += B) {
      const int cnt = nn - i0 < B ? nn - i0 : B;
      const DpFlatBlk SB(nd, seq_lo[s], flag, i0);
      for (int k = 0; k < B * B; ++k) table[k] = -__builtin_inf();
      for (int l = 0; l < cnt; ++l) {
        const int i = i0 + l; const DpNode n2 = S.node(i); const int c2 = dp_class(n2.strand, n2.stop);
        best[l] = -1.0; bj[l] = -1; bm[l] = -1; cls[l] = c2;
        int lo = (int)nd.dp_min[S.first + i];
        if (dp_class_pos_bounded(c2)) { while (lo < i && S.ndx(lo) < dp_pos_floor(n2.sv)) ++lo; }
        for (int c1 = 0; c1 < 4; ++c1) {
          if (!dp_pair_possible(c1, c2)) continue;
          for (int j = i - 1; j >= lo; --j) {          // (descending on purpose: the tie rule must not lean on the order)
            const DpNode n1 = S.node(j);
            if (dp_class(n1.strand, n1.stop) != c1) continue;
            double v; int mark;
            if (j >= i0) {
              if (dp_pair_dynamic(c1, c2)) continue;
              if (dp_connection_s_class(c1, SB, w, j, i, n2, v, mark)) { if (mark != -1) g_fail("a static connection with an overlap mark"); table[(j - i0) * B + l] = v; }
            } else if (dp_connection_class(c1, S, w, j, i, n2, v, mark)) dp_take(v, j, mark, best[l], bj[l], bm[l]);
          }
        }
      }
      for (int t = 0; t < cnt; ++t) {
        const uint32_t g = S.first + (uint32_t)(i0 + t);
        if (bj[t] >= 0) { nd.score[g] = best[t]; nd.traceb[g] = bj[t]; nd.ov_mark[g] = bm[t]; }
        const int c1 = cls[t];
        if (dp_class_needs_tb(c1) && bj[t] < 0) continue;
        const double sc_t = nd.score[g];
        for (int l = t + 1; l < cnt; ++l) {
          dp_take(sc_t + table[t * B + l], i0 + t, -1, best[l], bj[l], bm[l]);
          if (dp_pair_dynamic(c1, cls[l])) {
            double v; int mark;
            if (dp_connection_class(c1, S, w, i0 + t, i0 + l, S.node(i0 + l), v, mark)) dp_take(v, i0 + t, mark, best[l], bj[l], bm[l]);
          }
        }
      }
    }
  }
}

void x_path_ends(GExec &, const Nodes &nd, const uint32_t *seq_lo, const uint32_t *seq_n, uint32_t nseq, int32_t *end_rel) {
  for (uint32_t s = 0; s < nseq; ++s) {
    double best = -1.0; int bi = -1;
    for (int i = 0; i < (int)seq_n[s]; ++i) {           // (ascending on purpose: the tie rule -- the last of equals -- must not lean on the order)
      const uint32_t g = seq_lo[s] + (uint32_t)i; const int str = nd.strand[g]; const bool st = nd.type[g] == G_STOP;
      if ((str == 1 && !st) || (str == -1 && st)) continue;
      const double sc = nd.score[g];
      if (sc > -1.0 && (sc > best || (sc == best && i > bi))) { best = sc; bi = i; }
    }
    end_rel[s] = bi;
  }
}

void x_motif_bg(GExec &, int stage, const Nodes &nd, const int32_t *seq_len, const MotifPart *parts, uint32_t nparts, uint32_t *tab) {
  for (uint32_t p = 0; p < nparts; ++p) for (uint32_t x = parts[p].lo; x < parts[p].hi; ++x) {
    if (nd.type[x] >= G_STOP || nd.edge[x] == 1) continue;
    const int sl = seq_len[nd.seq[x]], strand = nd.strand[x], start = strand == 1 ? nd.ndx[x] : sl - 1 - nd.ndx[x];
    if (stage == 0) motif_words_stage0(nd.upw[x], start, [&](int i, int w) { tab[((size_t)parts[p].slot * 4 + i) * 4096 + w]++; });
    else motif_words_stage12(nd.mot[x], nd.upw[x], start, stage, [&](int i, int sp, int w) { tab[(size_t)parts[p].slot * 65536 + ((size_t)i * 4 + sp) * 4096 + w]++; });
  }
}

void x_hexamer_background(GExec &, const uint8_t *code, const uint64_t *seq_off, const int32_t *seq_len, uint32_t nbins, int, uint32_t *hist) {
  memset(hist, 0, (size_t)nbins * 4096 * 4);
  for (uint32_t b = 0; b < nbins; ++b) {
    const uint8_t *c = code + seq_off[b];
    for (int i = 0; i < seq_len[b] - 5; ++i) { int f = 0; for (int k = 0; k < 6; ++k) f |= (c[i + k] & 3) << (2 * k); hist[(size_t)b * 4096 + f]++; }
  }
}

void x_cscore(GExec &, const uint8_t *code, const uint64_t *seq_off, const int32_t *seq_len, const Nodes &nd, const double *gene_dc, uint32_t n) {
  for (uint32_t i = 0; i < n; ++i) {
    if (nd.type[i] >= G_STOP) continue;
    const uint32_t sq = nd.seq[i]; const int slen = seq_len[sq], strand = nd.strand[i];
    const int ps = strand == 1 ? nd.ndx[i] : slen - 1 - nd.ndx[i], pe = strand == 1 ? nd.sv[i] : slen - 1 - nd.sv[i];
    nd.cscore[i] = node_cscore(code + seq_off[sq], slen, strand, ps, pe, gene_dc + (size_t)nd.bin[i] * 4096);
  }
}
void x_rbs(GExec &, const uint8_t *code, const uint64_t *seq_off, const int32_t *seq_len, const Nodes &nd, const double *rbs_wt, uint32_t n) {
  for (uint32_t i = 0; i < n; ++i) {
    if (nd.type[i] >= G_STOP || nd.edge[i]) continue;
    const uint32_t sq = nd.seq[i]; const GSeq q{code + seq_off[sq], seq_len[sq]};
    const int strand = nd.strand[i], start = strand == 1 ? nd.ndx[i] : q.slen - 1 - nd.ndx[i];
    int r0, r1; node_rbs(q, strand, start, rbs_wt + (size_t)nd.bin[i] * 28, r0, r1);
    nd.rbs0[i] = (uint8_t)r0; nd.rbs1[i] = (uint8_t)r1;
  }
}

}  // namespace gene
}  // namespace ckm

using namespace ckm::gene;

struct emu_genes { GeneResult r; std::string err; };

// the columns of include/checkm_hip.h's ckm_genes_columns, in that order (tests/test_gene_emu.py reads them with the same ctypes struct)
struct emu_columns {
  uint64_t n; const uint32_t *bin, *contig; const int32_t *begin, *end; const int8_t *strand; const uint8_t *start_type, *partial_left, *partial_right;
  const int32_t *rbs_bin, *mot_len, *mot_ndx, *mot_spacer; const double *gc_cont, *conf, *score, *cscore, *sscore, *rscore, *uscore, *tscore;
  const uint64_t *prot_off; const char *prot;
  uint64_t nbins; const uint8_t *bin_trained, *bin_uses_sd; const double *bin_gc; const uint64_t *bin_bases, *bin_coding, *bin_nodes;
};

extern "C" int emu_genes_call(const char *text, const uint64_t *contig_off, uint32_t ncontigs, const uint32_t *bin_first, uint32_t nbins, int trans_table, int mask_runs, emu_genes **out) {
  emu_genes *o = new emu_genes();
  *out = o;
  try {
    PipeInput in;
    in.text = text; in.contig_off = contig_off; in.ncontigs = ncontigs; in.bin_first = bin_first; in.nbins = nbins; in.trans_table = trans_table; in.mask_runs = mask_runs;
    in.pfor = [](size_t n, const std::function<void(size_t)> &f) { for (size_t i = 0; i < n; ++i) f(i); };
    GExec e;
    gene_pipeline(e, in, o->r);
    return 0;
  } catch (const std::exception &ex) { o->err = ex.what(); return -1; }
}
extern "C" const char *emu_genes_error(const emu_genes *g) { return g->err.c_str(); }
extern "C" void emu_genes_columns(const emu_genes *gg, emu_columns *c) {
  const GeneResult *g = &gg->r;
  c->n = g->begin.size(); c->bin = g->bin.data(); c->contig = g->contig.data(); c->begin = g->begin.data(); c->end = g->end.data(); c->strand = g->strand.data();
  c->start_type = g->start_type.data(); c->partial_left = g->partial_left.data(); c->partial_right = g->partial_right.data();
  c->rbs_bin = g->rbs_bin.data(); c->mot_len = g->mot_len.data(); c->mot_ndx = g->mot_ndx.data(); c->mot_spacer = g->mot_spacer.data();
  c->gc_cont = g->gc_cont.data(); c->conf = g->conf.data(); c->score = g->score.data(); c->cscore = g->cscore.data(); c->sscore = g->sscore.data();
  c->rscore = g->rscore.data(); c->uscore = g->uscore.data(); c->tscore = g->tscore.data(); c->prot_off = g->prot_off.data(); c->prot = g->prot.data();
  c->nbins = g->bin_trained.size(); c->bin_trained = g->bin_trained.data(); c->bin_uses_sd = g->bin_uses_sd.data(); c->bin_gc = g->bin_gc.data();
  c->bin_bases = g->bin_bases.data(); c->bin_coding = g->bin_coding.data(); c->bin_nodes = g->bin_nodes_find.data();
}
extern "C" void emu_genes_free(emu_genes *g) { delete g; }
