/* gene_full.c -- TEST INFRASTRUCTURE ONLY (see oracle/README.md): a scalar restatement of the gene finder CheckM runs in front of the
 * marker-gene scan, single-genome mode, as CheckM invokes it.
 *
 * Reference call site: checkm/prodigal.py:80-93 -- `prodigal -p single -q -m -f gff -g <11|4> -a genes.faa -i <bin>` (`-p meta` below
 * 100 kb, checkm/prodigal.py:80-83: NOT restated, see below), twice per bin; the coding density of the two runs picks the table
 * (checkm/prodigal.py:117-133).  Prodigal (V2.6.3, Hyatt et al. 2010, BMC Bioinformatics 11:119) is a third-party C program that is
 * absent from /root/reference and from this image.  This file restates its published single-genome algorithm from the paper and from
 * memory of the source, function by function:
 *
 *   training on the whole bin (all contigs joined by TTAATTAATTAA, read_seq_training):
 *     add_nodes (+ the -m masks: runs of >= 50 N hide the starts whose ORF would cross them)   node.c
 *     calc_most_gc_frame / record_gc_bias       GC-frame plot, per-ORF frame bias              sequence.c, node.c
 *     record_overlapping_starts, dprog(flag 0)   first dynamic program on GC-frame bias alone   node.c, dprog.c
 *     calc_dicodon_gene, raw_coding_score        hexamer log-odds of the genes found; coding score of every start   node.c
 *     rbs_score, train_starts_sd, determine_sd_usage, train_starts_nonsd    start-site model (Shine-Dalgarno bins, or upstream motifs)
 *   gene finding, contig by contig:
 *     add_nodes, score_nodes, record_overlapping_starts(1), dprog(flag 1), eliminate_bad_genes, add_genes, tweak_final_starts,
 *     record_gene_data; translations (table 11 / 4).
 *
 * PARITY UNPINNED: no prodigal binary, source or output exists here; nothing ties this file to a real Prodigal run, and constants
 * recalled from memory may differ from the original's.  NOT restated: `-p meta` (its 50 pre-trained parameter files are data, not an
 * algorithm that can be restated), the GFF "conf"/score attribute formatting beyond what CheckM reads, closed-ends mode off by default as
 * in CheckM's call.  What the product must equal is THIS file, gene for gene (tests/test_gpu_genes.py); what this file is checked against
 * without Prodigal: a second formulation of its dynamic program and coding score (tests/test_gene_full_oracle.py) and hand-derived cases.
 *
 * Bases: 0 A, 1 C, 2 G, 3 T; anything else is unknown: it reads as C on the forward strand and G on the reverse strand for composition
 * (as the packed two-bit sequence of the original does) and translates to X; it is never part of a start or a stop codon.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define STOP 3
#define ATG 0
#define GTG 1
#define TTG 2
#define MIN_GENE 90
#define MIN_EDGE_GENE 60
#define MAX_SAM_OVLP 60
#define MAX_OPP_OVLP 200
#define MAX_NODE_DIST 500
#define OPER_DIST 60
#define EDGE_BONUS 0.74
#define EDGE_UPS -1.00
#define MASK_SIZE 50
#define GC_WINDOW 120

typedef struct {
  int32_t type, edge, ndx, strand, stop_val;
  int32_t star_ptr[3], gc_bias;
  double gc_score[3], cscore, gc_cont;
  int32_t rbs[2];
  int32_t mot_ndx, mot_len, mot_spacer, mot_spacendx; double mot_score;
  double uscore, tscore, rscore, sscore;
  int32_t traceb, tracef, ov_mark; double score; int32_t elim;
} pg_node;

typedef struct {
  double gc; int32_t trans_table; double st_wt; double bias[3]; double type_wt[3]; int32_t uses_sd;
  double rbs_wt[28]; double ups_comp[32][4]; double mot_wt[4][4][4096]; double no_mot; double gene_dc[4096];
} pg_training;

typedef struct {
  int32_t contig, begin, end, strand;           /* 1-based inclusive coordinates on the contig */
  int32_t start_type;                           /* 0 ATG 1 GTG 2 TTG 3 Edge */
  int32_t partial_left, partial_right;
  int32_t rbs_bin;                              /* SD bin chosen (uses_sd) or -1 */
  int32_t mot_len, mot_ndx, mot_spacer;         /* upstream motif (non-SD organisms), len 0 = none */
  double gc_cont, conf, score, cscore, sscore, rscore, uscore, tscore;
} pg_gene;

typedef struct { int32_t begin, end; } pg_mask;

typedef struct { const uint8_t *seq, *rseq; const uint8_t *unk, *runk; int slen; } pg_seq;

static double dmax(double a, double b) { return a > b ? a : b; }
static double dmin(double a, double b) { return a < b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }

/* ---- sequence predicates ---- */
static int is_stop(const pg_seq *s, const uint8_t *q, const uint8_t *u, int i, int tt) {
  if (i < 0 || i + 2 >= s->slen) return 0;
  if (u[i] || u[i + 1] || u[i + 2]) return 0;
  if (q[i] != 3) return 0;
  if (q[i + 1] == 0 && (q[i + 2] == 0 || q[i + 2] == 2)) return 1;       /* TAA TAG */
  if (q[i + 1] == 2 && q[i + 2] == 0) return tt != 4;                     /* TGA */
  return 0;
}
static int start_type(const pg_seq *s, const uint8_t *q, const uint8_t *u, int i) {
  if (i < 0 || i + 2 >= s->slen) return -1;
  if (u[i] || u[i + 1] || u[i + 2]) return -1;
  if (q[i + 1] != 3 || q[i + 2] != 2) return -1;
  if (q[i] == 0) return ATG;
  if (q[i] == 2) return GTG;
  if (q[i] == 3) return TTG;
  return -1;
}
static int is_gc(const uint8_t *q, int i) { return q[i] == 1 || q[i] == 2; }
static int mer_ndx(int len, const uint8_t *q, int pos) { int ndx = 0; for (int i = 0; i < len; ++i) ndx |= (int)q[pos + i] << (2 * i); return ndx; }

static int cross_mask(int x, int y, const pg_mask *m, int nm) {
  for (int i = 0; i < nm; ++i) if (y >= m[i].begin && x <= m[i].end) return 1;
  return 0;
}

/* ---- nodes (node.c: add_nodes) ---- */
static int add_nodes(const pg_seq *s, pg_node *nodes, int closed, const pg_mask *ml, int nm, int tt) {
  int nn = 0, last[3], saw[3], mind[3];
  const int slen = s->slen, slmod = slen % 3;
  for (int strand = 1; strand >= -1; strand -= 2) {
    const uint8_t *q = strand == 1 ? s->seq : s->rseq, *u = strand == 1 ? s->unk : s->runk;
    for (int i = 0; i < 3; ++i) {
      last[(i + slmod) % 3] = slen + i; saw[i] = 0; mind[i] = MIN_EDGE_GENE;
      if (!closed) while (last[(i + slmod) % 3] + 2 > slen - 1) last[(i + slmod) % 3] -= 3;
    }
#define NEWNODE(NDX, TYPE, SV, EDGE) do { pg_node *n_ = &nodes[nn++]; memset(n_, 0, sizeof(*n_)); n_->ndx = (strand == 1) ? (NDX) : slen - 1 - (NDX); n_->type = (TYPE); \
    n_->strand = strand; n_->stop_val = (strand == 1) ? (SV) : slen - 1 - (SV); n_->edge = (EDGE); } while (0)
    for (int i = slen - 3; i >= 0; --i) {
      const int f = i % 3;
      if (is_stop(s, q, u, i, tt)) {
        if (saw[f]) NEWNODE(last[f], STOP, i, is_stop(s, q, u, last[f], tt) ? 0 : 1);
        mind[f] = MIN_GENE; last[f] = i; saw[f] = 0;
        continue;
      }
      if (last[f] >= slen) continue;
      const int st = start_type(s, q, u, i);
      if (last[f] - i + 3 >= mind[f] && st >= 0) {
        const int x = strand == 1 ? i : slen - 1 - last[f], y = strand == 1 ? last[f] : slen - 1 - i;
        if (cross_mask(x, y, ml, nm)) continue;
        saw[f] = 1; NEWNODE(i, st, last[f], 0);
      } else if (i <= 2 && !closed && (last[f] - i) > MIN_EDGE_GENE) {
        const int x = strand == 1 ? i : slen - 1 - last[f], y = strand == 1 ? last[f] : slen - 1 - i;
        if (cross_mask(x, y, ml, nm)) continue;
        saw[f] = 1; NEWNODE(i, ATG, last[f], 1);
      }
    }
    for (int i = 0; i < 3; ++i)
      if (saw[i]) NEWNODE(last[i], STOP, i - 6, is_stop(s, q, u, last[i], tt) ? 0 : 1);
#undef NEWNODE
  }
  return nn;
}
static int cmp_nodes(const void *a, const void *b) {
  const pg_node *x = (const pg_node *)a, *y = (const pg_node *)b;
  if (x->ndx != y->ndx) return x->ndx < y->ndx ? -1 : 1;
  if (x->strand != y->strand) return x->strand > y->strand ? -1 : 1;      /* forward strand first */
  return 0;
}

/* ---- GC frame plot (sequence.c: calc_most_gc_frame) ---- */
static int max_fr(int n1, int n2, int n3) {
  if (n1 > n2) return n1 > n3 ? 0 : 2;
  return n2 > n3 ? 1 : 2;
}
static int *calc_most_gc_frame(const uint8_t *q, int slen) {
  int *gp = malloc(sizeof(int) * (slen + 3)), *fwd = malloc(sizeof(int) * (slen + 3)), *bwd = malloc(sizeof(int) * (slen + 3)), *tot = malloc(sizeof(int) * (slen + 3));
  for (int i = 0; i < slen; ++i) { fwd[i] = bwd[i] = tot[i] = 0; gp[i] = -1; }
  for (int j = 0; j < slen; ++j) {
    fwd[j] = (j < 3 ? 0 : fwd[j - 3]) + is_gc(q, j);
    const int r = slen - j - 1;
    bwd[r] = (j < 3 ? 0 : bwd[r + 3]) + is_gc(q, r);
  }
  for (int i = 0; i < slen; ++i) {
    tot[i] = fwd[i] + bwd[i] - is_gc(q, i);
    if (i - GC_WINDOW / 2 >= 0) tot[i] -= fwd[i - GC_WINDOW / 2];
    if (i + GC_WINDOW / 2 < slen) tot[i] -= bwd[i + GC_WINDOW / 2];
  }
  for (int i = 0; i < slen - 2; i += 3) {
    const int win = max_fr(tot[i], tot[i + 1], tot[i + 2]);
    gp[i] = gp[i + 1] = gp[i + 2] = win;
  }
  free(fwd); free(bwd); free(tot);
  return gp;
}

static void record_gc_bias(const int *gc, pg_node *nod, int nn, pg_training *t) {
  int ctr[3][3], last[3] = {0, 0, 0};
  if (nn == 0) return;
  memset(ctr, 0, sizeof(ctr));
  for (int i = nn - 1; i >= 0; --i) {
    const int fr = nod[i].ndx % 3, frmod = 3 - fr;
    if (nod[i].strand == 1 && nod[i].type == STOP) {
      for (int j = 0; j < 3; ++j) ctr[fr][j] = 0;
      last[fr] = nod[i].ndx;
      ctr[fr][(gc[nod[i].ndx] + frmod) % 3] = 1;
    } else if (nod[i].strand == 1) {
      for (int j = last[fr] - 3; j >= nod[i].ndx; j -= 3) ctr[fr][(gc[j] + frmod) % 3]++;
      nod[i].gc_bias = max_fr(ctr[fr][0], ctr[fr][1], ctr[fr][2]);
      for (int j = 0; j < 3; ++j) { nod[i].gc_score[j] = 3.0 * ctr[fr][j]; nod[i].gc_score[j] /= 1.0 * (nod[i].stop_val - nod[i].ndx + 3); }
      last[fr] = nod[i].ndx;
    }
  }
  for (int i = 0; i < nn; ++i) {
    const int fr = nod[i].ndx % 3, frmod = fr;
    if (nod[i].strand == -1 && nod[i].type == STOP) {
      for (int j = 0; j < 3; ++j) ctr[fr][j] = 0;
      last[fr] = nod[i].ndx;
      ctr[fr][((3 - gc[nod[i].ndx]) + frmod) % 3] = 1;
    } else if (nod[i].strand == -1) {
      for (int j = last[fr] + 3; j <= nod[i].ndx; j += 3) ctr[fr][((3 - gc[j]) + frmod) % 3]++;
      nod[i].gc_bias = max_fr(ctr[fr][0], ctr[fr][1], ctr[fr][2]);
      for (int j = 0; j < 3; ++j) { nod[i].gc_score[j] = 3.0 * ctr[fr][j]; nod[i].gc_score[j] /= 1.0 * (nod[i].ndx - nod[i].stop_val + 3); }
      last[fr] = nod[i].ndx;
    }
  }
  for (int i = 0; i < 3; ++i) t->bias[i] = 0.0;
  for (int i = 0; i < nn; ++i) if (nod[i].type != STOP) {
    const int len = abs(nod[i].stop_val - nod[i].ndx) + 1;
    t->bias[nod[i].gc_bias] += (nod[i].gc_score[nod[i].gc_bias] * len) / 1000.0;
  }
  const double tot = t->bias[0] + t->bias[1] + t->bias[2];
  for (int i = 0; i < 3; ++i) t->bias[i] *= (3.0 / tot);
}

/* ---- intergenic modifier, overlapping starts, connections, dynamic program (node.c, dprog.c) ---- */
static double intergenic_mod(const pg_node *n1, const pg_node *n2, const pg_training *t) {
  double rval = 0.0; int ovlp = 0;
  if ((n1->strand == 1 && n2->strand == 1 && (n1->ndx + 2 == n2->ndx || n1->ndx - 1 == n2->ndx)) ||
      (n1->strand == -1 && n2->strand == -1 && (n1->ndx + 2 == n2->ndx || n1->ndx - 1 == n2->ndx))) {
    if (n1->strand == 1 && n2->rscore < 0) rval -= n2->rscore;
    if (n1->strand == -1 && n1->rscore < 0) rval -= n1->rscore;
    if (n1->strand == 1 && n2->uscore < 0) rval -= n2->uscore;
    if (n1->strand == -1 && n1->uscore < 0) rval -= n1->uscore;
  }
  const int dist = abs(n1->ndx - n2->ndx);
  if (n1->strand == 1 && n2->strand == 1 && n1->ndx + 2 >= n2->ndx) ovlp = 1;
  else if (n1->strand == -1 && n2->strand == -1 && n1->ndx >= n2->ndx + 2) ovlp = 1;
  if (dist > 3 * OPER_DIST || n1->strand != n2->strand) rval -= 0.15 * t->st_wt;
  else if ((dist <= OPER_DIST && ovlp == 0) || dist < 0.25 * OPER_DIST) rval += (2.0 - (double)dist / OPER_DIST) * 0.15 * t->st_wt;
  return rval;
}

static void record_overlapping_starts(pg_node *nod, int nn, const pg_training *t, int flag) {
  for (int i = 0; i < nn; ++i) {
    for (int j = 0; j < 3; ++j) nod[i].star_ptr[j] = -1;
    if (nod[i].type != STOP || nod[i].edge == 1) continue;
    double max_sc = -100.0;
    if (nod[i].strand == 1) {
      for (int j = i + 3; j >= 0; --j) {
        if (j >= nn || nod[j].ndx > nod[i].ndx + 2) continue;
        if (nod[j].ndx + MAX_SAM_OVLP < nod[i].ndx) break;
        if (nod[j].strand == 1 && nod[j].type != STOP) {
          if (nod[j].stop_val <= nod[i].ndx) continue;
          const int f = nod[j].ndx % 3;
          if (flag == 0 && nod[i].star_ptr[f] == -1) nod[i].star_ptr[f] = j;
          else if (flag == 1) {
            const double sc = nod[j].cscore + nod[j].sscore + intergenic_mod(&nod[i], &nod[j], t);
            if (sc > max_sc) { nod[i].star_ptr[f] = j; max_sc = sc; }
          }
        }
      }
    } else {
      for (int j = i - 3; j < nn; ++j) {
        if (j < 0 || nod[j].ndx < nod[i].ndx - 2) continue;
        if (nod[j].ndx - MAX_SAM_OVLP > nod[i].ndx) break;
        if (nod[j].strand == -1 && nod[j].type != STOP) {
          if (nod[j].stop_val >= nod[i].ndx) continue;
          const int f = nod[j].ndx % 3;
          if (flag == 0 && nod[i].star_ptr[f] == -1) nod[i].star_ptr[f] = j;
          else if (flag == 1) {
            const double sc = nod[j].cscore + nod[j].sscore + intergenic_mod(&nod[j], &nod[i], t);
            if (sc > max_sc) { nod[i].star_ptr[f] = j; max_sc = sc; }
          }
        }
      }
    }
  }
}

static double gcb(const pg_training *t, const pg_node *n) { return t->bias[0] * n->gc_score[0] + t->bias[1] * n->gc_score[1] + t->bias[2] * n->gc_score[2]; }

static void score_connection(pg_node *nod, int p1, int p2, const pg_training *t, int flag) {
  pg_node *n1 = &nod[p1], *n2 = &nod[p2], *n3;
  int left = n1->ndx, right = n2->ndx, bnd, ovlp = 0, maxfr = -1;
  double score = 0.0, scr_mod = 0.0, maxval;
  const int s1 = n1->strand, s2 = n2->strand, st1 = n1->type == STOP, st2 = n2->type == STOP;
  /* invalid connections */
  if (!st1 && !st2 && s1 == s2) return;
  else if (s1 == 1 && !st1 && s2 == -1) return;
  else if (s1 == -1 && st1 && s2 == 1) return;
  else if (s1 == -1 && !st1 && s2 == 1 && st2) return;
  /* edge artifacts */
  if (n1->traceb == -1 && s1 == 1 && st1) return;
  if (n1->traceb == -1 && s1 == -1 && !st1) return;
  /* genes */
  if (s1 == s2 && s1 == 1 && !st1 && st2) {                        /* 5'fwd -> 3'fwd */
    if (n2->stop_val >= n1->ndx) return;
    if (n1->ndx % 3 != n2->ndx % 3) return;
    right += 2;
    if (flag == 0) scr_mod = gcb(t, n1); else score = n1->cscore + n1->sscore;
  } else if (s1 == s2 && s1 == -1 && st1 && !st2) {                /* 3'rev -> 5'rev */
    if (n1->stop_val <= n2->ndx) return;
    if (n1->ndx % 3 != n2->ndx % 3) return;
    left -= 2;
    if (flag == 0) scr_mod = gcb(t, n2); else score = n2->cscore + n2->sscore;
  }
  /* intergenic space */
  else if (s1 == 1 && st1 && s2 == 1 && !st2) {                    /* 3'fwd -> 5'fwd */
    left += 2;
    if (left >= right) return;
    if (flag == 1) score = intergenic_mod(n1, n2, t);
  } else if (s1 == 1 && st1 && s2 == -1 && st2) {                  /* 3'fwd -> 3'rev */
    left += 2; right -= 2;
    if (left >= right) return;
    /* three consecutive overlapping genes f r r */
    maxfr = -1; maxval = 0.0;
    for (int i = 0; i < 3; ++i) {
      if (n2->star_ptr[i] == -1) continue;
      n3 = &nod[n2->star_ptr[i]];
      ovlp = left - n3->stop_val + 1;
      if (ovlp <= 0 || ovlp >= MAX_OPP_OVLP) continue;
      if (ovlp >= n3->ndx - left) continue;
      if (n1->traceb == -1) continue;
      if (ovlp >= n3->stop_val - nod[n1->traceb].ndx - 2) continue;
      const double v = flag == 1 ? n3->cscore + n3->sscore + intergenic_mod(n3, n2, t) : gcb(t, n3);
      if (v > maxval) { maxfr = i; maxval = v; }
    }
    if (maxfr != -1) {
      n3 = &nod[n2->star_ptr[maxfr]];
      ovlp = left - n3->stop_val + 1;
      if (flag == 0) scr_mod = gcb(t, n3); else score = n3->cscore + n3->sscore + intergenic_mod(n3, n2, t);
    } else { ovlp = 0; if (flag == 1) score = intergenic_mod(n1, n2, t); }
  } else if (s1 == -1 && !st1 && s2 == -1 && st2) {                /* 5'rev -> 3'rev */
    right -= 2;
    if (left >= right) return;
    if (flag == 1) score = intergenic_mod(n1, n2, t);
  } else if (s1 == -1 && !st1 && s2 == 1 && !st2) {                /* 5'rev -> 5'fwd */
    if (left >= right) return;
    if (flag == 1) score = intergenic_mod(n1, n2, t);
  }
  /* overlapping opposite-strand 3' ends: 3'fwd -> 5'rev */
  else if (s1 == 1 && st1 && s2 == -1 && !st2) {
    if (n2->stop_val - 2 >= n1->ndx + 2) return;
    ovlp = (n1->ndx + 2) - (n2->stop_val - 2) + 1;
    if (ovlp >= MAX_OPP_OVLP) return;
    if ((n1->ndx + 2 - n2->stop_val - 2 + 1) >= (n2->ndx - n1->ndx + 3 + 1)) return;
    bnd = n1->traceb == -1 ? 0 : nod[n1->traceb].ndx;
    if ((n1->ndx + 2 - n2->stop_val - 2 + 1) >= (n2->stop_val - 3 - bnd + 1)) return;
    left = n2->stop_val - 2;
    if (flag == 0) scr_mod = gcb(t, n2); else score = n2->cscore + n2->sscore - 0.15 * t->st_wt;
  }
  /* overlapping same strand */
  else if (s1 == s2 && s1 == 1 && st1 && st2) {                    /* 3'fwd -> 3'fwd */
    if (n2->stop_val >= n1->ndx) return;
    if (n1->star_ptr[n2->ndx % 3] == -1) return;
    n3 = &nod[n1->star_ptr[n2->ndx % 3]];
    left = n3->ndx; right += 2;
    if (flag == 0) scr_mod = gcb(t, n3); else score = n3->cscore + n3->sscore + intergenic_mod(n1, n3, t);
  } else if (s1 == s2 && s1 == -1 && st1 && st2) {                 /* 3'rev -> 3'rev */
    if (n1->stop_val <= n2->ndx) return;
    if (n2->star_ptr[n1->ndx % 3] == -1) return;
    n3 = &nod[n2->star_ptr[n1->ndx % 3]];
    left -= 2; right = n3->ndx;
    if (flag == 0) scr_mod = gcb(t, n3); else score = n3->cscore + n3->sscore + intergenic_mod(n3, n2, t);
  }
  if (flag == 0) score = ((double)(right - left + 1 - (ovlp * 2))) * scr_mod;
  if (n1->score + score >= n2->score) { n2->score = n1->score + score; n2->traceb = p1; n2->ov_mark = maxfr; }
}

static int dprog(pg_node *nod, int nn, const pg_training *t, int flag) {
  int max_ndx = -1; double max_sc = -1.0;
  if (nn == 0) return -1;
  for (int i = 0; i < nn; ++i) { nod[i].score = 0; nod[i].traceb = -1; nod[i].tracef = -1; }
  for (int i = 0; i < nn; ++i) {
    int min = i < MAX_NODE_DIST ? 0 : i - MAX_NODE_DIST;
    if (nod[i].strand == -1 && nod[i].type != STOP && nod[min].ndx >= nod[i].stop_val)
      while (min >= 0 && nod[min].ndx != nod[i].stop_val) min--;
    if (nod[i].strand == 1 && nod[i].type == STOP && nod[min].ndx >= nod[i].stop_val)
      while (min >= 0 && nod[min].ndx != nod[i].stop_val) min--;
    min = min < MAX_NODE_DIST ? 0 : min - MAX_NODE_DIST;
    for (int j = min; j < i; ++j) score_connection(nod, j, i, t, flag);
  }
  for (int i = nn - 1; i >= 0; --i) {
    if (nod[i].strand == 1 && nod[i].type != STOP) continue;
    if (nod[i].strand == -1 && nod[i].type == STOP) continue;
    if (nod[i].score > max_sc) { max_sc = nod[i].score; max_ndx = i; }
  }
  if (max_ndx < 0) return -1;
  /* first pass: untangle the triple overlaps */
  int path = max_ndx;
  while (nod[path].traceb != -1) {
    const int nxt = nod[path].traceb;
    if (nod[path].strand == -1 && nod[path].type == STOP && nod[nxt].strand == 1 && nod[nxt].type == STOP && nod[path].ov_mark != -1 && nod[path].ndx > nod[nxt].ndx) {
      const int tmp = nod[path].star_ptr[nod[path].ov_mark];
      int i;
      for (i = tmp; nod[i].ndx != nod[tmp].stop_val; --i);
      nod[path].traceb = tmp; nod[tmp].traceb = i; nod[i].ov_mark = -1; nod[i].traceb = nxt;
    }
    path = nod[path].traceb;
  }
  /* second pass: untangle the simple overlaps */
  path = max_ndx;
  while (nod[path].traceb != -1) {
    const int nxt = nod[path].traceb;
    if (nod[path].strand == -1 && nod[path].type != STOP && nod[nxt].strand == 1 && nod[nxt].type == STOP) {
      int i;
      for (i = path; nod[i].ndx != nod[path].stop_val; --i);
      nod[path].traceb = i; nod[i].traceb = nxt;
    }
    if (nod[path].strand == 1 && nod[path].type == STOP && nod[nxt].strand == 1 && nod[nxt].type == STOP) {
      nod[path].traceb = nod[nxt].star_ptr[nod[path].ndx % 3];
      nod[nod[path].traceb].traceb = nxt;
    }
    if (nod[path].strand == -1 && nod[path].type == STOP && nod[nxt].strand == -1 && nod[nxt].type == STOP) {
      nod[path].traceb = nod[path].star_ptr[nod[nxt].ndx % 3];
      nod[nod[path].traceb].traceb = nxt;
    }
    path = nod[path].traceb;
  }
  path = max_ndx;
  while (nod[path].traceb != -1) { nod[nod[path].traceb].tracef = path; path = nod[path].traceb; }
  return nod[max_ndx].traceb == -1 ? -1 : max_ndx;
}

/* ---- hexamer statistics and coding score (node.c: calc_dicodon_gene, raw_coding_score) ---- */
static void calc_dicodon_gene(pg_training *t, const pg_seq *s, const pg_node *nod, int dbeg) {
  static double prob[4096], bg[4096]; static int counts[4096];
  int glob = 0, left = -1, right = -1, in_gene = 0;
  const int slen = s->slen;
  memset(counts, 0, sizeof(counts));
  { /* calc_mer_bg(6): every position, both strands */
    static int bc[4096]; long long g = 0; memset(bc, 0, sizeof(bc));
    for (int i = 0; i < slen - 5; ++i) { bc[mer_ndx(6, s->seq, i)]++; bc[mer_ndx(6, s->rseq, i)]++; g += 2; }
    for (int i = 0; i < 4096; ++i) bg[i] = g ? (double)bc[i] / (double)g : 0.0;
  }
  for (int path = dbeg; path != -1; path = nod[path].traceb) {
    if (nod[path].strand == -1 && nod[path].type != STOP) { in_gene = -1; left = slen - nod[path].ndx - 1; }
    if (nod[path].strand == 1 && nod[path].type == STOP) { in_gene = 1; right = nod[path].ndx + 2; }
    if (in_gene == -1 && nod[path].strand == -1 && nod[path].type == STOP) {
      right = slen - nod[path].ndx + 1;
      for (int i = left; i < right - 5; i += 3) { counts[mer_ndx(6, s->rseq, i)]++; glob++; }
      in_gene = 0;
    }
    if (in_gene == 1 && nod[path].strand == 1 && nod[path].type != STOP) {
      left = nod[path].ndx;
      for (int i = left; i < right - 5; i += 3) { counts[mer_ndx(6, s->seq, i)]++; glob++; }
      in_gene = 0;
    }
  }
  for (int i = 0; i < 4096; ++i) {
    prob[i] = glob ? (counts[i] * 1.0) / (glob * 1.0) : 0.0;
    if (prob[i] == 0 && bg[i] != 0) t->gene_dc[i] = -5.0;
    else if (bg[i] == 0) t->gene_dc[i] = 0.0;
    else t->gene_dc[i] = log(prob[i] / bg[i]);
    if (t->gene_dc[i] > 5.0) t->gene_dc[i] = 5.0;
    if (t->gene_dc[i] < -5.0) t->gene_dc[i] = -5.0;
  }
}

static void raw_coding_score(const pg_seq *s, pg_node *nod, int nn, const pg_training *t) {
  int last[3] = {0, 0, 0}; double score[3], lfac, no_stop, gsize;
  const int slen = s->slen; const double gc = t->gc;
  if (t->trans_table != 11) { no_stop = ((1 - gc) * (1 - gc) * gc) / 8.0; no_stop += ((1 - gc) * (1 - gc) * (1 - gc)) / 8.0; no_stop = 1 - no_stop; }
  else { no_stop = ((1 - gc) * (1 - gc) * gc) / 4.0; no_stop += ((1 - gc) * (1 - gc) * (1 - gc)) / 8.0; no_stop = 1 - no_stop; }
  /* first pass: start -> stop */
  for (int i = 0; i < 3; ++i) score[i] = 0.0;
  for (int i = nn - 1; i >= 0; --i) if (nod[i].strand == 1) {
    const int fr = nod[i].ndx % 3;
    if (nod[i].type == STOP) { last[fr] = nod[i].ndx; score[fr] = 0.0; }
    else {
      for (int j = last[fr] - 3; j >= nod[i].ndx; j -= 3) score[fr] += t->gene_dc[mer_ndx(6, s->seq, j)];
      nod[i].cscore = score[fr]; last[fr] = nod[i].ndx;
    }
  }
  for (int i = 0; i < 3; ++i) score[i] = 0.0;
  for (int i = 0; i < nn; ++i) if (nod[i].strand == -1) {
    const int fr = nod[i].ndx % 3;
    if (nod[i].type == STOP) { last[fr] = nod[i].ndx; score[fr] = 0.0; }
    else {
      for (int j = last[fr] + 3; j <= nod[i].ndx; j += 3) score[fr] += t->gene_dc[mer_ndx(6, s->rseq, slen - j - 1)];
      nod[i].cscore = score[fr]; last[fr] = nod[i].ndx;
    }
  }
  /* second pass: penalise start nodes with ascending coding to their left */
  for (int i = 0; i < 3; ++i) score[i] = -10000;
  for (int i = 0; i < nn; ++i) if (nod[i].strand == 1) {
    const int fr = nod[i].ndx % 3;
    if (nod[i].type == STOP) score[fr] = -10000;
    else if (nod[i].cscore > score[fr]) score[fr] = nod[i].cscore;
    else nod[i].cscore -= (score[fr] - nod[i].cscore);
  }
  for (int i = 0; i < 3; ++i) score[i] = -10000;
  for (int i = nn - 1; i >= 0; --i) if (nod[i].strand == -1) {
    const int fr = nod[i].ndx % 3;
    if (nod[i].type == STOP) score[fr] = -10000;
    else if (nod[i].cscore > score[fr]) score[fr] = nod[i].cscore;
    else nod[i].cscore -= (score[fr] - nod[i].cscore);
  }
  /* third pass: length factor */
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = 0; i < 3; ++i) score[i] = -10000;
    for (int k = 0; k < nn; ++k) {
      const int i = pass == 0 ? k : nn - 1 - k;
      if (nod[i].strand != (pass == 0 ? 1 : -1)) continue;
      const int fr = nod[i].ndx % 3;
      if (nod[i].type == STOP) { score[fr] = -10000; continue; }
      gsize = ((double)(abs(nod[i].stop_val - nod[i].ndx) + 3.0)) / 3.0;
      if (gsize > 1000.0) {
        lfac = log((1 - pow(no_stop, 1000.0)) / pow(no_stop, 1000.0));
        lfac -= log((1 - pow(no_stop, 80)) / pow(no_stop, 80));
        lfac *= (gsize - 80) / 920.0;
      } else {
        lfac = log((1 - pow(no_stop, gsize)) / pow(no_stop, gsize));
        lfac -= log((1 - pow(no_stop, 80)) / pow(no_stop, 80));
      }
      if (lfac > score[fr]) score[fr] = lfac;
      else lfac -= dmax(dmin(score[fr] - lfac, lfac), 0);
      if (lfac > 3.0 && nod[i].cscore < 0.5 * lfac) nod[i].cscore = 0.5 * lfac;
      nod[i].cscore += lfac;
    }
  }
}

/* ---- Shine-Dalgarno bins (sequence.c: shine_dalgarno_exact / _mm) ---- */
static int sd_bin_exact(double cur_ctr, int dis_flag) {
  if (cur_ctr < 6.0) return 0;
  if (cur_ctr == 6.0 && dis_flag == 2) return 1;
  if (cur_ctr == 6.0 && dis_flag == 3) return 2;
  if (cur_ctr == 8.0 && dis_flag == 3) return 3;
  if (cur_ctr == 9.0 && dis_flag == 3) return 3;
  if (cur_ctr == 6.0 && dis_flag == 1) return 6;
  if (cur_ctr == 11.0 && dis_flag == 3) return 10;
  if (cur_ctr == 12.0 && dis_flag == 3) return 10;
  if (cur_ctr == 14.0 && dis_flag == 3) return 10;
  if (cur_ctr == 8.0 && dis_flag == 2) return 11;
  if (cur_ctr == 9.0 && dis_flag == 2) return 11;
  if (cur_ctr == 8.0 && dis_flag == 1) return 12;
  if (cur_ctr == 9.0 && dis_flag == 1) return 12;
  if (cur_ctr == 6.0 && dis_flag == 0) return 13;
  if (cur_ctr == 8.0 && dis_flag == 0) return 15;
  if (cur_ctr == 9.0 && dis_flag == 0) return 16;
  if (cur_ctr == 11.0 && dis_flag == 2) return 20;
  if (cur_ctr == 11.0 && dis_flag == 1) return 21;
  if (cur_ctr == 11.0 && dis_flag == 0) return 22;
  if (cur_ctr == 12.0 && dis_flag == 2) return 20;
  if (cur_ctr == 12.0 && dis_flag == 1) return 23;
  if (cur_ctr == 12.0 && dis_flag == 0) return 24;
  if (cur_ctr == 14.0 && dis_flag == 2) return 25;
  if (cur_ctr == 14.0 && dis_flag == 1) return 26;
  if (cur_ctr == 14.0 && dis_flag == 0) return 27;
  return 0;
}
static int sd_bin_mm(double cur_ctr, int dis_flag) {
  if (cur_ctr < 6.0) return 0;
  if (cur_ctr == 6.0 && dis_flag == 3) return 2;
  if (cur_ctr == 7.0 && dis_flag == 3) return 2;
  if (cur_ctr == 9.0 && dis_flag == 3) return 3;
  if (cur_ctr == 6.0 && dis_flag == 2) return 4;
  if (cur_ctr == 6.0 && dis_flag == 1) return 5;
  if (cur_ctr == 6.0 && dis_flag == 0) return 9;
  if (cur_ctr == 7.0 && dis_flag == 2) return 7;
  if (cur_ctr == 7.0 && dis_flag == 1) return 8;
  if (cur_ctr == 7.0 && dis_flag == 0) return 14;
  if (cur_ctr == 9.0 && dis_flag == 2) return 17;
  if (cur_ctr == 9.0 && dis_flag == 1) return 18;
  if (cur_ctr == 9.0 && dis_flag == 0) return 19;
  return 0;
}
static int shine_dalgarno(const uint8_t *q, const uint8_t *u, int pos, int start, const double *rwt, int mm) {
  double match[6], cur_ctr; int max_val = 0;
  const int limit = imin(6, start - 4 - pos);
  for (int i = 0; i < 6; ++i) match[i] = -10.0;
  for (int i = 0; i < limit; ++i) {
    if (pos + i < 0) continue;
    const int a = !u[pos + i] && q[pos + i] == 0, g = !u[pos + i] && q[pos + i] == 2;
    if (i % 3 == 0) match[i] = a ? 2.0 : (mm ? -3.0 : -10.0);
    else match[i] = g ? 3.0 : (mm ? -2.0 : -10.0);
  }
  for (int i = limit; i >= (mm ? 5 : 3); --i) {
    for (int j = 0; j <= limit - i; ++j) {
      cur_ctr = -2.0; int mism = 0;
      for (int k = j; k < j + i; ++k) {
        cur_ctr += match[k];
        if (match[k] < 0.0) mism++;
        if (mm && match[k] < 0.0 && (k <= j + 1 || k >= j + i - 2)) cur_ctr -= 10.0;
      }
      if (mm ? mism != 1 : mism > 0) continue;
      const int rdis = start - (pos + j + i);
      int dis_flag;
      if (!mm) {
        if (rdis < 5 && i < 5) dis_flag = 2;
        else if (rdis < 5 && i >= 5) dis_flag = 1;
        else if (rdis > 10 && rdis <= 12 && i < 5) dis_flag = 1;
        else if (rdis > 10 && rdis <= 12 && i >= 5) dis_flag = 2;
        else if (rdis >= 13) dis_flag = 3;
        else dis_flag = 0;
      } else {
        if (rdis < 5) dis_flag = 1;
        else if (rdis > 10 && rdis <= 12) dis_flag = 2;
        else if (rdis >= 13) dis_flag = 3;
        else dis_flag = 0;
      }
      if (rdis > 15 || cur_ctr < 6.0) continue;
      const int cur_val = mm ? sd_bin_mm(cur_ctr, dis_flag) : sd_bin_exact(cur_ctr, dis_flag);
      if (rwt[cur_val] < rwt[max_val]) continue;
      if (rwt[cur_val] == rwt[max_val] && cur_val < max_val) continue;
      max_val = cur_val;
    }
  }
  return max_val;
}
static void rbs_score(const pg_seq *s, pg_node *nod, int nn, const pg_training *t) {
  const int slen = s->slen;
  for (int i = 0; i < nn; ++i) {
    if (nod[i].type == STOP || nod[i].edge == 1) continue;
    nod[i].rbs[0] = nod[i].rbs[1] = 0;
    const uint8_t *q = nod[i].strand == 1 ? s->seq : s->rseq, *u = nod[i].strand == 1 ? s->unk : s->runk;
    const int start = nod[i].strand == 1 ? nod[i].ndx : slen - 1 - nod[i].ndx;
    for (int j = start - 20; j <= start - 6; ++j) {
      if (j < 0) continue;
      const int c0 = shine_dalgarno(q, u, j, start, t->rbs_wt, 0), c1 = shine_dalgarno(q, u, j, start, t->rbs_wt, 1);
      if (c0 > nod[i].rbs[0]) nod[i].rbs[0] = c0;
      if (c1 > nod[i].rbs[1]) nod[i].rbs[1] = c1;
    }
  }
}

/* ---- upstream composition ---- */
static void count_upstream_composition(const uint8_t *q, int start, pg_training *t) {
  int count = 0;
  for (int i = 1; i < 45; ++i) {
    if (i > 2 && i < 15) continue;
    if (start - i >= 0) t->ups_comp[count][q[start - i]] += 1.0;
    count++;
  }
}
static void score_upstream_composition(const uint8_t *q, int start, pg_node *n, const pg_training *t) {
  int count = 0;
  n->uscore = 0.0;
  for (int i = 1; i < 45; ++i) {
    if (i > 2 && i < 15) continue;
    if (start - i < 0) continue;
    n->uscore += 0.4 * t->st_wt * t->ups_comp[count][q[start - i]];
    count++;
  }
}
static void ups_to_log(pg_training *t) {
  for (int i = 0; i < 32; ++i) {
    double sum = 0.0;
    for (int j = 0; j < 4; ++j) sum += t->ups_comp[i][j];
    if (sum == 0.0) { for (int j = 0; j < 4; ++j) t->ups_comp[i][j] = 0.0; continue; }
    for (int j = 0; j < 4; ++j) {
      double x = t->ups_comp[i][j] / sum;
      const int at = (j == 0 || j == 3);
      if (t->gc > 0.1 && t->gc < 0.9) x = at ? log(x * 2.0 / (1.0 - t->gc)) : log(x * 2.0 / t->gc);
      else if (t->gc <= 0.1) x = at ? log(x * 2.0 / 0.90) : log(x * 2.0 / 0.10);
      else x = at ? log(x * 2.0 / 0.10) : log(x * 2.0 / 0.90);
      if (x > 4.0) x = 4.0;
      if (x < -4.0) x = -4.0;
      t->ups_comp[i][j] = x;
    }
  }
}

static int best_rbs(const pg_node *n, const pg_training *t) {
  if (t->rbs_wt[n->rbs[0]] > t->rbs_wt[n->rbs[1]] + 1.0 || n->rbs[1] == 0) return n->rbs[0];
  if (t->rbs_wt[n->rbs[0]] < t->rbs_wt[n->rbs[1]] - 1.0 || n->rbs[0] == 0) return n->rbs[1];
  return n->rbs[0] > n->rbs[1] ? n->rbs[0] : n->rbs[1];
}
static void type_bg(const pg_node *nod, int nn, double *tbg) {
  double sum = 0.0;
  for (int i = 0; i < 3; ++i) tbg[i] = 0.0;
  for (int i = 0; i < nn; ++i) if (nod[i].type != STOP) tbg[nod[i].type] += 1.0;
  for (int i = 0; i < 3; ++i) sum += tbg[i];
  for (int i = 0; i < 3; ++i) tbg[i] = sum ? tbg[i] / sum : 0.0;
}
static void update_type_wt(pg_training *t, double *treal, const double *tbg, double *sum_out) {
  double sum = 0.0;
  for (int j = 0; j < 3; ++j) sum += treal[j];
  if (sum == 0.0) for (int j = 0; j < 3; ++j) t->type_wt[j] = 0.0;
  else for (int j = 0; j < 3; ++j) {
    treal[j] /= sum;
    t->type_wt[j] = tbg[j] != 0 ? log(treal[j] / tbg[j]) : -4.0;
    if (t->type_wt[j] > 4.0) t->type_wt[j] = 4.0;
    if (t->type_wt[j] < -4.0) t->type_wt[j] = -4.0;
  }
  *sum_out = sum;
}

/* node.c: train_starts_sd */
static void train_starts_sd(const pg_seq *s, pg_node *nod, int nn, pg_training *t) {
  int rbs[3], type[3], bndx[3]; double sum, rbg[28], rreal[28], best[3], sthresh = 35.0, tbg[3], treal[3];
  const double wt = t->st_wt; const int slen = s->slen;
  for (int j = 0; j < 3; ++j) t->type_wt[j] = 0.0;
  for (int j = 0; j < 28; ++j) t->rbs_wt[j] = 0.0;
  memset(t->ups_comp, 0, sizeof(t->ups_comp));
  type_bg(nod, nn, tbg);
  for (int it = 0; it < 10; ++it) {
    for (int j = 0; j < 28; ++j) rbg[j] = 0.0;
    for (int j = 0; j < nn; ++j) { if (nod[j].type == STOP || nod[j].edge == 1) continue; rbg[best_rbs(&nod[j], t)] += 1.0; }
    sum = 0.0; for (int j = 0; j < 28; ++j) sum += rbg[j];
    for (int j = 0; j < 28; ++j) rbg[j] = sum ? rbg[j] / sum : 0.0;
    for (int j = 0; j < 28; ++j) rreal[j] = 0.0;
    for (int j = 0; j < 3; ++j) treal[j] = 0.0;
    for (int pass = 0; pass < 2; ++pass) {
      const int str = pass == 0 ? 1 : -1;
      for (int j = 0; j < 3; ++j) { best[j] = 0.0; bndx[j] = -1; rbs[j] = 0; type[j] = 0; }
      for (int k = 0; k < nn; ++k) {
        const int j = pass == 0 ? k : nn - 1 - k;
        if (nod[j].type != STOP && nod[j].edge == 1) continue;
        const int fr = nod[j].ndx % 3;
        if (nod[j].type == STOP && nod[j].strand == str) {
          if (best[fr] >= sthresh && nod[bndx[fr]].ndx % 3 == fr) {
            rreal[rbs[fr]] += 1.0; treal[type[fr]] += 1.0;
            if (it == 9) count_upstream_composition(str == 1 ? s->seq : s->rseq, str == 1 ? nod[bndx[fr]].ndx : slen - 1 - nod[bndx[fr]].ndx, t);
          }
          best[fr] = 0.0; bndx[fr] = -1; rbs[fr] = 0; type[fr] = 0;
        } else if (nod[j].strand == str && nod[j].type != STOP) {
          const int mr = best_rbs(&nod[j], t);
          const double v = nod[j].cscore + wt * t->rbs_wt[mr] + wt * t->type_wt[nod[j].type];
          if (v >= best[fr]) { best[fr] = nod[j].cscore + wt * t->rbs_wt[mr]; best[fr] += wt * t->type_wt[nod[j].type]; bndx[fr] = j; type[fr] = nod[j].type; rbs[fr] = mr; }
        }
      }
    }
    sum = 0.0; for (int j = 0; j < 28; ++j) sum += rreal[j];
    if (sum == 0.0) for (int j = 0; j < 28; ++j) t->rbs_wt[j] = 0.0;
    else for (int j = 0; j < 28; ++j) {
      rreal[j] /= sum;
      t->rbs_wt[j] = rbg[j] != 0 ? log(rreal[j] / rbg[j]) : -4.0;
      if (t->rbs_wt[j] > 4.0) t->rbs_wt[j] = 4.0;
      if (t->rbs_wt[j] < -4.0) t->rbs_wt[j] = -4.0;
    }
    update_type_wt(t, treal, tbg, &sum);
    if (sum <= (double)nn / 2000.0) sthresh /= 2.0;
  }
  ups_to_log(t);
}
static void determine_sd_usage(pg_training *t) {
  t->uses_sd = 1;
  if (t->rbs_wt[0] >= 0.0) t->uses_sd = 0;
  if (t->rbs_wt[16] < 1.0 && t->rbs_wt[13] < 1.0 && t->rbs_wt[15] < 1.0 &&
      (t->rbs_wt[0] >= -0.5 || (t->rbs_wt[22] < 2.0 && t->rbs_wt[24] < 2.0 && t->rbs_wt[27] < 2.0))) t->uses_sd = 0;
}

/* ---- upstream motifs for organisms without SD (node.c: find_best_upstream_motif, update_motif_counts, build_coverage_map, train_starts_nonsd) ---- */
static int spacer_ndx(int j, int start, int i) {
  if (j <= start - 16 - i) return 3;
  if (j <= start - 14 - i) return 2;
  if (j >= start - 7 - i) return 1;
  return 0;
}
static void find_best_upstream_motif(const pg_training *t, const pg_seq *s, pg_node *n, int stage) {
  if (n->type == STOP || n->edge == 1) return;
  const uint8_t *q = n->strand == 1 ? s->seq : s->rseq;
  const int start = n->strand == 1 ? n->ndx : s->slen - 1 - n->ndx;
  int max_spacer = 0, max_spacendx = 0, max_len = 0, max_ndx = 0; double max_sc = -100.0;
  for (int i = 3; i >= 0; --i) for (int j = start - 18 - i; j <= start - 6 - i; ++j) {
    if (j < 0) continue;
    const int spacer = start - j - i - 3, sp = spacer_ndx(j, start, i), index = mer_ndx(i + 3, q, j);
    const double score = t->mot_wt[i][sp][index];
    if (score > max_sc) { max_sc = score; max_spacendx = sp; max_spacer = spacer; max_ndx = index; max_len = i + 3; }
  }
  if (stage == 2 && (max_sc == -4.0 || max_sc < t->no_mot + 0.69)) { n->mot_ndx = 0; n->mot_len = 0; n->mot_spacendx = 0; n->mot_spacer = 0; n->mot_score = t->no_mot; }
  else { n->mot_ndx = max_ndx; n->mot_len = max_len; n->mot_spacendx = max_spacendx; n->mot_spacer = max_spacer; n->mot_score = max_sc; }
}
typedef double mot_tab[4][4][4096];
static void update_motif_counts(mot_tab mcnt, double *zero, const pg_seq *s, const pg_node *n, int stage) {
  if (n->type == STOP || n->edge == 1) return;
  if (n->mot_len == 0) { *zero += 1.0; return; }
  const uint8_t *q = n->strand == 1 ? s->seq : s->rseq;
  const int start = n->strand == 1 ? n->ndx : s->slen - 1 - n->ndx;
  if (stage == 0) {
    for (int i = 3; i >= 0; --i) for (int j = start - 18 - i; j <= start - 6 - i; ++j) {
      if (j < 0) continue;
      for (int k = 0; k < 4; ++k) mcnt[i][k][mer_ndx(i + 3, q, j)] += 1.0;
    }
  } else if (stage == 1) {
    mcnt[n->mot_len - 3][n->mot_spacendx][n->mot_ndx] += 1.0;
    for (int i = 0; i < n->mot_len - 3; ++i) for (int j = start - n->mot_spacer - n->mot_len; j <= start - n->mot_spacer - (i + 3); ++j) {
      if (j < 0) continue;
      mcnt[i][spacer_ndx(j, start, i)][mer_ndx(i + 3, q, j)] += 1.0;
    }
  } else mcnt[n->mot_len - 3][n->mot_spacendx][n->mot_ndx] += 1.0;
}
typedef int good_tab[4][4][4096];
static void build_coverage_map(mot_tab real, good_tab good, double ng, int stage) {
  const double thresh = 0.2; int decomp[3];
  (void)stage;
  memset(good, 0, sizeof(good_tab));
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 64; ++j) if (real[0][i][j] / ng >= thresh) for (int k = 0; k < 4; ++k) good[0][k][j] = 1;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 256; ++j) {
    decomp[0] = (j & 252) >> 2; decomp[1] = j & 63;
    if (good[0][i][decomp[0]] == 0 || good[0][i][decomp[1]] == 0) continue;
    good[1][i][j] = 1;
  }
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 1024; ++j) {
    decomp[0] = (j & 1008) >> 4; decomp[1] = (j & 252) >> 2; decomp[2] = j & 63;
    if (good[0][i][decomp[0]] == 0 || good[0][i][decomp[1]] == 0 || good[0][i][decomp[2]] == 0) continue;
    good[2][i][j] = 1;
    int tmp = j;
    for (int k = 0; k <= 16; k += 16) { tmp = tmp ^ k; for (int l = 0; l <= 32; l += 32) { tmp = tmp ^ l; if (good[2][i][tmp] == 0) good[2][i][tmp] = 2; } }
  }
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4096; ++j) {
    decomp[0] = (j & 4092) >> 2; decomp[1] = j & 1023;
    if (good[2][i][decomp[0]] == 0 || good[2][i][decomp[1]] == 0) continue;
    good[3][i][j] = (good[2][i][decomp[0]] == 1 && good[2][i][decomp[1]] == 1) ? 1 : 2;
  }
}
static void train_starts_nonsd(const pg_seq *s, pg_node *nod, int nn, pg_training *t) {
  static mot_tab mbg, mreal; static good_tab mgood;
  int bndx[3], stage; double sum, ngenes, best[3], sthresh = 35.0, tbg[3], treal[3], zbg, zreal;
  const double wt = t->st_wt; const int slen = s->slen;
  for (int j = 0; j < 3; ++j) t->type_wt[j] = 0.0;
  memset(t->mot_wt, 0, sizeof(t->mot_wt)); t->no_mot = 0.0;
  memset(t->ups_comp, 0, sizeof(t->ups_comp));
  type_bg(nod, nn, tbg);
  for (int it = 0; it < 20; ++it) {
    stage = it < 4 ? 0 : it < 12 ? 1 : 2;
    memset(mbg, 0, sizeof(mbg)); zbg = 0.0;
    for (int j = 0; j < nn; ++j) {
      if (nod[j].type == STOP || nod[j].edge == 1) continue;
      find_best_upstream_motif(t, s, &nod[j], stage);
      update_motif_counts(mbg, &zbg, s, &nod[j], stage);
    }
    sum = zbg;
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int c = 0; c < 4096; ++c) sum += mbg[a][b][c];
    if (sum != 0.0) { for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int c = 0; c < 4096; ++c) mbg[a][b][c] /= sum; zbg /= sum; }
    memset(mreal, 0, sizeof(mreal)); zreal = 0.0; ngenes = 0.0;
    for (int j = 0; j < 3; ++j) treal[j] = 0.0;
    for (int pass = 0; pass < 2; ++pass) {
      const int str = pass == 0 ? 1 : -1;
      for (int j = 0; j < 3; ++j) { best[j] = 0.0; bndx[j] = -1; }
      for (int k = 0; k < nn; ++k) {
        const int j = pass == 0 ? k : nn - 1 - k;
        if (nod[j].type != STOP && nod[j].edge == 1) continue;
        const int fr = nod[j].ndx % 3;
        if (nod[j].type == STOP && nod[j].strand == str) {
          if (best[fr] >= sthresh) {
            ngenes += 1.0; treal[nod[bndx[fr]].type] += 1.0;
            update_motif_counts(mreal, &zreal, s, &nod[bndx[fr]], stage);
            if (it == 19) count_upstream_composition(str == 1 ? s->seq : s->rseq, str == 1 ? nod[bndx[fr]].ndx : slen - 1 - nod[bndx[fr]].ndx, t);
          }
          best[fr] = 0.0; bndx[fr] = -1;
        } else if (nod[j].strand == str && nod[j].type != STOP) {
          const double v = nod[j].cscore + wt * nod[j].mot_score + wt * t->type_wt[nod[j].type];
          if (v >= best[fr]) { best[fr] = nod[j].cscore + wt * nod[j].mot_score; best[fr] += wt * t->type_wt[nod[j].type]; bndx[fr] = j; }
        }
      }
    }
    if (stage < 2) build_coverage_map(mreal, mgood, ngenes, stage);
    sum = zreal;
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int c = 0; c < 4096; ++c) sum += mreal[a][b][c];
    if (sum == 0.0) { memset(t->mot_wt, 0, sizeof(t->mot_wt)); t->no_mot = 0.0; }
    else {
      for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int c = 0; c < 4096; ++c) {
        if (mgood[a][b][c] == 0) { zreal += mreal[a][b][c]; zbg += mreal[a][b][c]; mreal[a][b][c] = 0.0; mbg[a][b][c] = 0.0; }
        mreal[a][b][c] /= sum;
        double v = mbg[a][b][c] != 0 ? log(mreal[a][b][c] / mbg[a][b][c]) : -4.0;
        if (v > 4.0) v = 4.0;
        if (v < -4.0) v = -4.0;
        t->mot_wt[a][b][c] = v;
      }
      zreal /= sum;
      t->no_mot = zbg != 0 ? log(zreal / zbg) : -4.0;
      if (t->no_mot > 4.0) t->no_mot = 4.0;
      if (t->no_mot < -4.0) t->no_mot = -4.0;
    }
    update_type_wt(t, treal, tbg, &sum);
    if (sum <= (double)nn / 2000.0) sthresh /= 2.0;
  }
  ups_to_log(t);
}

/* ---- node scoring (node.c: score_nodes), single-genome mode ---- */
static void calc_orf_gc(const pg_seq *s, pg_node *nod, int nn) {
  /* GC content of every start -> stop span (reported per gene) */
  for (int i = 0; i < nn; ++i) {
    if (nod[i].type == STOP) continue;
    const int a = nod[i].strand == 1 ? nod[i].ndx : nod[i].stop_val, b = nod[i].strand == 1 ? nod[i].stop_val + 2 : nod[i].ndx;
    int g = 0;
    for (int j = (a < 0 ? 0 : a); j <= b && j < s->slen; ++j) g += is_gc(s->seq, j);
    nod[i].gc_cont = (double)g / (double)(abs(nod[i].stop_val - nod[i].ndx) + 3);
  }
}
static void score_nodes(const pg_seq *s, pg_node *nod, int nn, const pg_training *t, int closed) {
  const int slen = s->slen, tt = t->trans_table;
  calc_orf_gc(s, nod, nn);
  raw_coding_score(s, nod, nn, t);
  if (t->uses_sd == 1) rbs_score(s, nod, nn, t);
  else for (int i = 0; i < nn; ++i) { if (nod[i].type == STOP || nod[i].edge == 1) continue; find_best_upstream_motif(t, s, &nod[i], 2); }
  for (int i = 0; i < nn; ++i) {
    if (nod[i].type == STOP) continue;
    int edge_gene = 0;
    if (nod[i].edge == 1) edge_gene++;
    if ((nod[i].strand == 1 && !is_stop(s, s->seq, s->unk, nod[i].stop_val, tt)) || (nod[i].strand == -1 && !is_stop(s, s->rseq, s->runk, slen - 1 - nod[i].stop_val, tt))) edge_gene++;
    if (nod[i].edge == 1) { nod[i].tscore = EDGE_BONUS * t->st_wt / edge_gene; nod[i].uscore = 0.0; nod[i].rscore = 0.0; }
    else {
      nod[i].tscore = t->type_wt[nod[i].type] * t->st_wt;
      const double rbs1 = t->rbs_wt[nod[i].rbs[0]], rbs2 = t->rbs_wt[nod[i].rbs[1]], sd_score = dmax(rbs1, rbs2) * t->st_wt;
      if (t->uses_sd == 1) nod[i].rscore = sd_score;
      else { nod[i].rscore = t->st_wt * nod[i].mot_score; if (nod[i].rscore < sd_score && t->no_mot > -0.5) nod[i].rscore = sd_score; }
      if (nod[i].strand == 1) score_upstream_composition(s->seq, nod[i].ndx, &nod[i], t);
      else score_upstream_composition(s->rseq, slen - 1 - nod[i].ndx, &nod[i], t);
      if (closed == 0 && nod[i].ndx <= 2 && nod[i].strand == 1) nod[i].uscore += EDGE_UPS * t->st_wt;
      else if (closed == 0 && nod[i].ndx >= slen - 3 && nod[i].strand == -1) nod[i].uscore += EDGE_UPS * t->st_wt;
      else if (i < 500 && nod[i].strand == 1) {
        for (int j = i - 1; j >= 0; --j) if (nod[j].edge == 1 && nod[i].stop_val == nod[j].stop_val) { nod[i].uscore += EDGE_UPS * t->st_wt; break; }
      } else if (i >= nn - 500 && nod[i].strand == -1) {
        for (int j = i + 1; j < nn; ++j) if (nod[j].edge == 1 && nod[i].stop_val == nod[j].stop_val) { nod[i].uscore += EDGE_UPS * t->st_wt; break; }
      }
    }
    if (((nod[i].ndx <= 2 && nod[i].strand == 1) || (nod[i].ndx >= slen - 3 && nod[i].strand == -1)) && nod[i].edge == 0 && closed == 0) {
      edge_gene++; nod[i].edge = 1; nod[i].tscore = 0.0; nod[i].uscore = EDGE_BONUS * t->st_wt / edge_gene; nod[i].rscore = 0.0;
    }
    if (nod[i].edge == 0 && edge_gene == 1) nod[i].uscore -= 0.5 * EDGE_BONUS * t->st_wt;
    if (edge_gene == 0 && abs(nod[i].ndx - nod[i].stop_val) < 250) {
      const double negf = 250.0 / (float)abs(nod[i].ndx - nod[i].stop_val), posf = (float)abs(nod[i].ndx - nod[i].stop_val) / 250.0;
      if (nod[i].rscore < 0) nod[i].rscore *= negf;
      if (nod[i].uscore < 0) nod[i].uscore *= negf;
      if (nod[i].tscore < 0) nod[i].tscore *= negf;
      if (nod[i].rscore > 0) nod[i].rscore *= posf;
      if (nod[i].uscore > 0) nod[i].uscore *= posf;
      if (nod[i].tscore > 0) nod[i].tscore *= posf;
    }
    nod[i].sscore = nod[i].tscore + nod[i].rscore + nod[i].uscore;
    if (nod[i].cscore < 0.0) {
      if (edge_gene > 0 && nod[i].edge == 0) nod[i].sscore -= t->st_wt;
      else nod[i].sscore -= 0.5;
    }
  }
}

/* ---- after the dynamic program (dprog.c: eliminate_bad_genes; gene.c: add_genes, tweak_final_starts, record_gene_data) ---- */
static void eliminate_bad_genes(pg_node *nod, int dbeg, const pg_training *t) {
  if (dbeg == -1) return;
  int path = dbeg;
  while (nod[path].traceb != -1) path = nod[path].traceb;
  while (nod[path].tracef != -1) {
    if (nod[path].strand == 1 && nod[path].type == STOP) nod[nod[path].tracef].sscore += intergenic_mod(&nod[path], &nod[nod[path].tracef], t);
    if (nod[path].strand == -1 && nod[path].type != STOP) nod[path].sscore += intergenic_mod(&nod[path], &nod[nod[path].tracef], t);
    path = nod[path].tracef;
  }
  path = dbeg;
  while (nod[path].traceb != -1) path = nod[path].traceb;
  while (nod[path].tracef != -1) {
    if (nod[path].strand == 1 && nod[path].type != STOP && nod[path].cscore + nod[path].sscore < 0) { nod[path].elim = 1; nod[nod[path].tracef].elim = 1; }
    if (nod[path].strand == -1 && nod[path].type == STOP && nod[nod[path].tracef].cscore + nod[nod[path].tracef].sscore < 0) { nod[path].elim = 1; nod[nod[path].tracef].elim = 1; }
    path = nod[path].tracef;
  }
}
typedef struct { int begin, end, start_ndx, stop_ndx; } gene_rec;
static int add_genes(gene_rec *gl, int cap, const pg_node *nod, int dbeg) {
  if (dbeg == -1) return 0;
  int path = dbeg, ctr = 0;
  while (nod[path].traceb != -1) path = nod[path].traceb;
  while (path != -1 && ctr < cap) {
    if (nod[path].elim == 1) { path = nod[path].tracef; continue; }
    if (nod[path].strand == 1 && nod[path].type != STOP) { gl[ctr].begin = nod[path].ndx + 1; gl[ctr].start_ndx = path; }
    if (nod[path].strand == -1 && nod[path].type == STOP) { gl[ctr].begin = nod[path].ndx - 1; gl[ctr].stop_ndx = path; }
    if (nod[path].strand == 1 && nod[path].type == STOP) { gl[ctr].end = nod[path].ndx + 3; gl[ctr].stop_ndx = path; ctr++; }
    if (nod[path].strand == -1 && nod[path].type != STOP) { gl[ctr].end = nod[path].ndx + 1; gl[ctr].start_ndx = path; ctr++; }
    path = nod[path].tracef;
  }
  return ctr;
}
static void tweak_final_starts(gene_rec *genes, int ng, const pg_node *nod, int nn, const pg_training *t) {
  for (int i = 0; i < ng; ++i) {
    const int ndx = genes[i].start_ndx;
    const double sc = nod[ndx].sscore + nod[ndx].cscore;
    double igm = 0.0;
    if (i > 0 && nod[ndx].strand == 1 && nod[genes[i - 1].start_ndx].strand == 1) igm = intergenic_mod(&nod[genes[i - 1].stop_ndx], &nod[ndx], t);
    if (i > 0 && nod[ndx].strand == 1 && nod[genes[i - 1].start_ndx].strand == -1) igm = intergenic_mod(&nod[genes[i - 1].start_ndx], &nod[ndx], t);
    if (i < ng - 1 && nod[ndx].strand == -1 && nod[genes[i + 1].start_ndx].strand == 1) igm = intergenic_mod(&nod[ndx], &nod[genes[i + 1].start_ndx], t);
    if (i < ng - 1 && nod[ndx].strand == -1 && nod[genes[i + 1].start_ndx].strand == -1) igm = intergenic_mod(&nod[ndx], &nod[genes[i + 1].stop_ndx], t);
    int maxndx[2] = {-1, -1}; double maxsc[2] = {0, 0}, maxigm[2] = {0, 0};
    for (int j = ndx - 100; j < ndx + 100; ++j) {
      if (j < 0 || j >= nn || j == ndx) continue;
      if (nod[j].type == STOP || nod[j].stop_val != nod[ndx].stop_val) continue;
      double tigm = 0.0;
      if (i > 0 && nod[j].strand == 1 && nod[genes[i - 1].start_ndx].strand == 1) {
        if (nod[genes[i - 1].stop_ndx].ndx - nod[j].ndx > MAX_SAM_OVLP) continue;
        tigm = intergenic_mod(&nod[genes[i - 1].stop_ndx], &nod[j], t);
      }
      if (i > 0 && nod[j].strand == 1 && nod[genes[i - 1].start_ndx].strand == -1) {
        if (nod[genes[i - 1].start_ndx].ndx - nod[j].ndx >= 0) continue;
        tigm = intergenic_mod(&nod[genes[i - 1].start_ndx], &nod[j], t);
      }
      if (i < ng - 1 && nod[j].strand == -1 && nod[genes[i + 1].start_ndx].strand == 1) {
        if (nod[j].ndx - nod[genes[i + 1].start_ndx].ndx >= 0) continue;
        tigm = intergenic_mod(&nod[j], &nod[genes[i + 1].start_ndx], t);
      }
      if (i < ng - 1 && nod[j].strand == -1 && nod[genes[i + 1].start_ndx].strand == -1) {
        if (nod[j].ndx - nod[genes[i + 1].stop_ndx].ndx > MAX_SAM_OVLP) continue;
        tigm = intergenic_mod(&nod[j], &nod[genes[i + 1].stop_ndx], t);
      }
      const double v = nod[j].cscore + nod[j].sscore;
      if (maxndx[0] == -1) { maxndx[0] = j; maxsc[0] = v; maxigm[0] = tigm; }
      else if (v + tigm > maxsc[0] + maxigm[0]) { maxndx[1] = maxndx[0]; maxsc[1] = maxsc[0]; maxigm[1] = maxigm[0]; maxndx[0] = j; maxsc[0] = v; maxigm[0] = tigm; }
      else if (maxndx[1] == -1 || v + tigm > maxsc[1] + maxigm[1]) { maxndx[1] = j; maxsc[1] = v; maxigm[1] = tigm; }
    }
    for (int j = 0; j < 2; ++j) {
      const int m = maxndx[j];
      if (m == -1) continue;
      if (nod[m].tscore < nod[ndx].tscore && maxsc[j] - nod[m].tscore >= sc - nod[ndx].tscore + t->st_wt && nod[m].rscore > nod[ndx].rscore &&
          nod[m].uscore > nod[ndx].uscore && nod[m].cscore > nod[ndx].cscore && abs(nod[m].ndx - nod[ndx].ndx) > 15) {
        maxsc[j] += nod[ndx].tscore - nod[m].tscore;
      } else if (abs(nod[m].ndx - nod[ndx].ndx) <= 15 && nod[m].rscore + nod[m].tscore > nod[ndx].rscore + nod[ndx].tscore && nod[ndx].edge == 0 && nod[m].edge == 0) {
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
static double confidence(double score, double st_wt) {
  double conf;
  if (score / st_wt < 41) { conf = exp(score / st_wt); conf = (conf / (conf + 1)) * 100.0; } else conf = 99.99;
  if (conf <= 50.00) conf = 50.00;
  return conf;
}

/* ---- the whole pipeline ---- */
static void make_seq(const uint8_t *raw, int n, uint8_t *seq, uint8_t *unk, uint8_t *rseq, uint8_t *runk) {
  for (int i = 0; i < n; ++i) { const int u = raw[i] > 3; unk[i] = (uint8_t)u; seq[i] = u ? 1 : raw[i]; }
  for (int i = 0; i < n; ++i) { runk[i] = unk[n - 1 - i]; rseq[i] = (uint8_t)(3 - seq[n - 1 - i]); }
}
static int find_masks(const uint8_t *unk, int n, pg_mask *m, int cap) {
  int nm = 0, run = -1;
  for (int i = 0; i <= n; ++i) {
    const int u = i < n && unk[i];
    if (u && run < 0) run = i;
    if (!u && run >= 0) { if (i - run >= MASK_SIZE && nm < cap) { m[nm].begin = run; m[nm].end = i - 1; nm++; } run = -1; }
  }
  return nm;
}

/* Train on all contigs of a bin (digitized: 0..3, anything else unknown).  Returns 0, or -1 when the bin is shorter than 20000 bases
 * (prodigal refuses to train on less in single mode). */
int pg_train(const uint8_t *const *contigs, const int32_t *lens, int ncontig, int trans_table, int closed, int do_mask, pg_training *t) {
  static const uint8_t sep[12] = {3, 3, 0, 0, 3, 3, 0, 0, 3, 3, 0, 0};   /* TTAATTAATTAA */
  long long total = 0;
  for (int c = 0; c < ncontig; ++c) total += lens[c];
  const long long slen_ll = total + (ncontig > 1 ? 12LL * ncontig : 0);
  if (total < 20000 || slen_ll > 2000000000LL) return -1;
  const int slen = (int)slen_ll;
  uint8_t *raw = malloc((size_t)slen + 16), *seq = malloc((size_t)slen + 16), *unk = malloc((size_t)slen + 16), *rseq = malloc((size_t)slen + 16), *runk = malloc((size_t)slen + 16);
  int pos = 0; long long gcc = 0;
  for (int c = 0; c < ncontig; ++c) {
    memcpy(raw + pos, contigs[c], (size_t)lens[c]);
    for (int i = 0; i < lens[c]; ++i) gcc += (contigs[c][i] == 1 || contigs[c][i] == 2);
    pos += lens[c];
    if (ncontig > 1) { memcpy(raw + pos, sep, 12); pos += 12; }
  }
  make_seq(raw, slen, seq, unk, rseq, runk);
  memset(t, 0, sizeof(*t));
  t->st_wt = 4.35; t->trans_table = trans_table; t->gc = (double)gcc / (double)slen;
  pg_mask *ml = malloc(sizeof(pg_mask) * 100000); int nm = 0;
  if (do_mask) nm = find_masks(unk, slen, ml, 100000);
  pg_seq s = {seq, rseq, unk, runk, slen};
  pg_node *nodes = malloc(sizeof(pg_node) * ((size_t)slen / 4 + 1000));       /* at most one node per codon position and strand... bounded below */
  /* (a node needs a start or stop codon: at most 2 * slen / 3 + 12; slen/4 is exceeded only by degenerate sequences, so count first) */
  free(nodes);
  nodes = malloc(sizeof(pg_node) * ((size_t)slen * 2 / 3 + 64) + 64);
  int nn = add_nodes(&s, nodes, closed, ml, nm, trans_table);
  qsort(nodes, (size_t)nn, sizeof(pg_node), cmp_nodes);
  int *gcf = calc_most_gc_frame(seq, slen);
  record_gc_bias(gcf, nodes, nn, t);
  free(gcf);
  record_overlapping_starts(nodes, nn, t, 0);
  const int ipath = dprog(nodes, nn, t, 0);
  calc_dicodon_gene(t, &s, nodes, ipath);
  raw_coding_score(&s, nodes, nn, t);
  rbs_score(&s, nodes, nn, t);
  train_starts_sd(&s, nodes, nn, t);
  determine_sd_usage(t);
  if (t->uses_sd == 0) train_starts_nonsd(&s, nodes, nn, t);
  free(nodes); free(ml); free(raw); free(seq); free(unk); free(rseq); free(runk);
  return 0;
}

/* Genes of ONE contig with a trained model.  Returns the number of genes (which may exceed cap: call again). */
int pg_find(const uint8_t *contig, int slen, int contig_index, const pg_training *t, int closed, int do_mask, pg_gene *out, int cap) {
  if (slen < 3) return 0;
  uint8_t *seq = malloc((size_t)slen + 16), *unk = malloc((size_t)slen + 16), *rseq = malloc((size_t)slen + 16), *runk = malloc((size_t)slen + 16);
  make_seq(contig, slen, seq, unk, rseq, runk);
  pg_mask *ml = malloc(sizeof(pg_mask) * 10000); int nm = 0;
  if (do_mask) nm = find_masks(unk, slen, ml, 10000);
  pg_seq s = {seq, rseq, unk, runk, slen};
  pg_node *nodes = malloc(sizeof(pg_node) * ((size_t)slen * 2 / 3 + 64) + 64);
  int nn = add_nodes(&s, nodes, closed, ml, nm, t->trans_table);
  qsort(nodes, (size_t)nn, sizeof(pg_node), cmp_nodes);
  score_nodes(&s, nodes, nn, t, closed);
  record_overlapping_starts(nodes, nn, t, 1);
  const int ipath = dprog(nodes, nn, t, 1);
  eliminate_bad_genes(nodes, ipath, t);
  gene_rec *gl = malloc(sizeof(gene_rec) * ((size_t)nn / 2 + 16));
  int ng = add_genes(gl, nn / 2 + 16, nodes, ipath);
  tweak_final_starts(gl, ng, nodes, nn, t);
  for (int i = 0; i < ng && i < cap; ++i) {
    const pg_node *st = &nodes[gl[i].start_ndx], *sp = &nodes[gl[i].stop_ndx];
    pg_gene *g = &out[i];
    memset(g, 0, sizeof(*g));
    g->contig = contig_index; g->begin = gl[i].begin; g->end = gl[i].end; g->strand = st->strand;
    g->start_type = st->edge ? 3 : st->type;
    g->partial_left = (st->strand == 1) ? st->edge : sp->edge;
    g->partial_right = (st->strand == 1) ? sp->edge : st->edge;
    const double rbs1 = t->rbs_wt[st->rbs[0]] * t->st_wt, rbs2 = t->rbs_wt[st->rbs[1]] * t->st_wt;
    g->rbs_bin = -1; g->mot_len = 0; g->mot_ndx = 0; g->mot_spacer = 0;
    if (t->uses_sd == 1) g->rbs_bin = rbs1 > rbs2 ? st->rbs[0] : st->rbs[1];
    else if (t->no_mot > -0.5 && rbs1 > rbs2 && rbs1 > st->mot_score * t->st_wt) g->rbs_bin = st->rbs[0];
    else if (t->no_mot > -0.5 && rbs2 >= rbs1 && rbs2 > st->mot_score * t->st_wt) g->rbs_bin = st->rbs[1];
    else { g->mot_len = st->mot_len; g->mot_ndx = st->mot_ndx; g->mot_spacer = st->mot_spacer; }
    g->gc_cont = st->gc_cont; g->cscore = st->cscore; g->sscore = st->sscore; g->rscore = st->rscore; g->uscore = st->uscore; g->tscore = st->tscore;
    g->score = st->cscore + st->sscore; g->conf = confidence(st->cscore + st->sscore, t->st_wt);
  }
  free(gl); free(nodes); free(ml); free(seq); free(unk); free(rseq); free(runk);
  return ng;
}

/* Protein of a gene (gene.c: write_translations): codons from the gene's 5' end; the first codon of a complete gene reads M whatever
 * start codon it is; a codon with an unknown base reads X; the stop reads '*'.  out must hold (end - begin + 1) / 3 + 1 bytes. */
static char amino(const uint8_t *q, const uint8_t *u, int i, int tt) {
  static const char *code = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF";   /* index 16*b0 + 4*b1 + b2 over A C G T */
  if (u[i] || u[i + 1] || u[i + 2]) return 'X';
  const int c = q[i] * 16 + q[i + 1] * 4 + q[i + 2];
  if (tt == 4 && c == 3 * 16 + 2 * 4 + 0) return 'W';
  return code[c];
}
int pg_translate(const uint8_t *contig, int slen, const pg_gene *g, int tt, char *out) {
  uint8_t *seq = malloc((size_t)slen + 16), *unk = malloc((size_t)slen + 16), *rseq = malloc((size_t)slen + 16), *runk = malloc((size_t)slen + 16);
  make_seq(contig, slen, seq, unk, rseq, runk);
  const uint8_t *q = g->strand == 1 ? seq : rseq, *u = g->strand == 1 ? unk : runk;
  const int b = g->strand == 1 ? g->begin - 1 : slen - g->end, e = g->strand == 1 ? g->end - 1 : slen - g->begin;
  const int partial5 = g->strand == 1 ? g->partial_left : g->partial_right;
  int n = 0;
  for (int i = b; i + 2 <= e; i += 3) {
    char a = amino(q, u, i, tt);
    if (i == b && !partial5) a = 'M';
    out[n++] = a;
  }
  out[n] = 0;
  free(seq); free(unk); free(rseq); free(runk);
  return n;
}

int pg_sizeof_training(void) { return (int)sizeof(pg_training); }
int pg_sizeof_gene(void) { return (int)sizeof(pg_gene); }

/* ---- test hooks (tests/test_gene_full_oracle.py) ---- */
void pg_test_rbs(const uint8_t *raw, int slen, int start, int *exact, int *mm) {
  uint8_t *seq = malloc((size_t)slen + 16), *unk = malloc((size_t)slen + 16), *rseq = malloc((size_t)slen + 16), *runk = malloc((size_t)slen + 16);
  make_seq(raw, slen, seq, unk, rseq, runk);
  double rwt[28]; for (int i = 0; i < 28; ++i) rwt[i] = 0.0;
  int r0 = 0, r1 = 0;
  for (int j = start - 20; j <= start - 6; ++j) {
    if (j < 0) continue;
    const int c0 = shine_dalgarno(seq, unk, j, start, rwt, 0), c1 = shine_dalgarno(seq, unk, j, start, rwt, 1);
    if (c0 > r0) r0 = c0;
    if (c1 > r1) r1 = c1;
  }
  *exact = r0; *mm = r1;
  free(seq); free(unk); free(rseq); free(runk);
}
/* nodes of one contig in working order and the first candidate predecessor dprog gives each of them */
int pg_test_window(const uint8_t *raw, int slen, int tt, int32_t *ndx, int32_t *sv, int32_t *strand, int32_t *type, int32_t *mn, int cap) {
  uint8_t *seq = malloc((size_t)slen + 16), *unk = malloc((size_t)slen + 16), *rseq = malloc((size_t)slen + 16), *runk = malloc((size_t)slen + 16);
  make_seq(raw, slen, seq, unk, rseq, runk);
  pg_seq s = {seq, rseq, unk, runk, slen};
  pg_node *nod = malloc(sizeof(pg_node) * ((size_t)slen * 2 / 3 + 64) + 64);
  const int nn = add_nodes(&s, nod, 0, NULL, 0, tt);
  qsort(nod, (size_t)nn, sizeof(pg_node), cmp_nodes);
  for (int i = 0; i < nn && i < cap; ++i) {
    int min = i < MAX_NODE_DIST ? 0 : i - MAX_NODE_DIST;
    if (nod[i].strand == -1 && nod[i].type != STOP && nod[min].ndx >= nod[i].stop_val) while (min >= 0 && nod[min].ndx != nod[i].stop_val) min--;
    if (nod[i].strand == 1 && nod[i].type == STOP && nod[min].ndx >= nod[i].stop_val) while (min >= 0 && nod[min].ndx != nod[i].stop_val) min--;
    min = min < MAX_NODE_DIST ? 0 : min - MAX_NODE_DIST;
    ndx[i] = nod[i].ndx; sv[i] = nod[i].stop_val; strand[i] = nod[i].strand; type[i] = nod[i].type; mn[i] = min;
  }
  free(nod); free(seq); free(unk); free(rseq); free(runk);
  return nn;
}
