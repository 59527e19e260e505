/* gene_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle/README.md): a scalar restatement of the deterministic FRONT END of the gene
 * finder CheckM runs in front of the marker-gene scan.
 *
 * Reference call site: checkm/prodigal.py:74,86-93,131-133 -- `prodigal -p single|meta -q -m -f gff -g <11|4> -a ... -i ...`, twice
 * per bin (translation tables 11 and 4).  Prodigal itself (V2.6.3, Hyatt et al. 2010) is a third-party C program that is absent
 * from /root/reference and from this image, so this file restates its PUBLISHED node-extraction step (node.c: add_nodes, with
 * sequence.c: is_start / is_stop / is_atg / is_gtg / is_ttg) from the description of the algorithm and from memory of the source:
 *
 *   for each strand, scanning positions i = slen-3 .. 0 with three per-frame registers (last stop seen, "a start was recorded since",
 *   minimum ORF length: 60 nt while the frame still runs off the sequence edge, 90 nt once a stop has been seen):
 *     - a stop codon closes the frame's open stretch: if a start was recorded in it, a STOP node is emitted for the DOWNSTREAM stop
 *       (ndx = last, stop_val = this stop); last = i;
 *     - a start codon (ATG / GTG / TTG) with last - i + 3 >= minimum length emits a START node (ndx = i, stop_val = last);
 *     - the first codon of each frame (i <= 2), when it is no such start and last - i > 60, emits an EDGE start (the gene may run
 *       off the 5' end of an unclosed contig; prodigal types it ATG and flags edge = 1);
 *     - after the scan every frame with a recorded start emits its final STOP node (stop_val = frame - 6);
 *   reverse strand: the same over the reverse complement, coordinates mapped back (ndx = slen - 1 - i).
 *   Translation table 11 stops at TAA / TAG / TGA, table 4 at TAA / TAG (TGA codes for Trp); both start at ATG / GTG / TTG.
 *   A stop node whose `ndx` codon is not a stop codon (the stretch ran off the 3' edge) carries edge = 1.
 *
 * PARITY UNPINNED: no prodigal binary, source or output exists here; nothing ties this file to a real Prodigal run.  Known gaps
 * against Prodigal 2.6.3: `-m` (runs of >= 50 N mask gene starts that would cross them) is not restated -- N is simply "no codon";
 * the node ORDER prodigal works with (qsort by ndx, then strand) is the order p_nodes_sorted() returns.  Everything behind the
 * nodes (GC-frame training, hexamer / RBS scoring, the dynamic program over nodes, gene records) is NOT restated: this is the
 * first slice of SURVEY 8f N1, the part that is a pure byte scan.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MIN_GENE 90
#define MIN_EDGE_GENE 60
enum { T_ATG = 0, T_GTG = 1, T_TTG = 2, T_STOP = 3 };

typedef struct { int32_t ndx, type, strand, stop_val, edge; } go_node;

/* bases: 0 A, 1 C, 2 G, 3 T, 4 anything else */
static int codon(const uint8_t *s, int i) { return (s[i] > 3 || s[i + 1] > 3 || s[i + 2] > 3) ? -1 : s[i] * 16 + s[i + 1] * 4 + s[i + 2]; }
static int is_stop(const uint8_t *s, int i, int tt) {
  const int c = codon(s, i);
  if (c == 3 * 16 + 0 * 4 + 0 || c == 3 * 16 + 0 * 4 + 2) return 1;       /* TAA TAG */
  if (c == 3 * 16 + 2 * 4 + 0) return tt != 4;                             /* TGA: Trp in table 4 */
  return 0;
}
static int start_type(const uint8_t *s, int i) {
  const int c = codon(s, i);
  if (c == 0 * 16 + 3 * 4 + 2) return T_ATG;
  if (c == 2 * 16 + 3 * 4 + 2) return T_GTG;
  if (c == 3 * 16 + 3 * 4 + 2) return T_TTG;
  return -1;
}

static int scan_strand(const uint8_t *s, int slen, int tt, int closed, int strand, go_node *out, int cap) {
  int nn = 0, last[3], saw[3], mind[3];
  const int slmod = slen % 3;
#define EMIT(NDX, TYPE, SV, EDGE) do { if (nn < cap) { out[nn].ndx = (strand == 1) ? (NDX) : slen - 1 - (NDX); out[nn].type = (TYPE); out[nn].strand = strand; \
                                        out[nn].stop_val = (strand == 1) ? (SV) : slen - 1 - (SV); out[nn].edge = (EDGE); } ++nn; } while (0)
  for (int i = 0; i < 3; ++i) {
    last[(i + slmod) % 3] = slen + i; saw[i] = 0; mind[i] = MIN_EDGE_GENE;
    if (!closed) while (last[(i + slmod) % 3] + 2 > slen - 1) last[(i + slmod) % 3] -= 3;
  }
  for (int i = slen - 3; i >= 0; --i) {
    const int f = i % 3;
    if (is_stop(s, i, tt)) {
      if (saw[f]) EMIT(last[f], T_STOP, i, is_stop(s, last[f], tt) ? 0 : 1);
      mind[f] = MIN_GENE; last[f] = i; saw[f] = 0;
      continue;
    }
    if (last[f] >= slen) continue;
    const int st = start_type(s, i);
    if (last[f] - i + 3 >= mind[f] && st >= 0) { saw[f] = 1; EMIT(i, st, last[f], 0); }
    else if (i <= 2 && !closed && (last[f] - i) > MIN_EDGE_GENE) { saw[f] = 1; EMIT(i, T_ATG, last[f], 1); }
  }
  for (int i = 0; i < 3; ++i)
    if (saw[i]) EMIT(last[i], T_STOP, i - 6, (last[i] + 2 < slen && is_stop(s, last[i], tt)) ? 0 : 1);
#undef EMIT
  return nn;
}

/* Nodes of one contig, both strands, in emission order (forward strand first).  Returns the number of nodes the contig has (which may
 * exceed cap: call again with a larger buffer). */
int go_nodes(const uint8_t *seq, int slen, int trans_table, int closed, go_node *out, int cap) {
  if (slen < 3) return 0;
  uint8_t *rc = (uint8_t *)malloc((size_t)slen);
  for (int i = 0; i < slen; ++i) { const uint8_t b = seq[slen - 1 - i]; rc[i] = b > 3 ? 4 : (uint8_t)(3 - b); }
  int n = scan_strand(seq, slen, trans_table, closed, 1, out, cap);
  n += scan_strand(rc, slen, trans_table, closed, -1, (n < cap) ? out + n : out, (n < cap) ? cap - n : 0);
  free(rc);
  return n;
}

/* prodigal's working order: qsort by ndx, then strand with the forward strand first (node.c: compare_nodes) */
static int cmp(const void *a, const void *b) {
  const go_node *x = (const go_node *)a, *y = (const go_node *)b;
  if (x->ndx != y->ndx) return x->ndx < y->ndx ? -1 : 1;
  if (x->strand != y->strand) return x->strand > y->strand ? -1 : 1;      /* forward strand first */
  if (x->type != y->type) return x->type < y->type ? -1 : 1;
  if (x->stop_val != y->stop_val) return x->stop_val < y->stop_val ? -1 : 1;
  return x->edge - y->edge;
}
void go_sort(go_node *nodes, int n) { qsort(nodes, (size_t)n, sizeof(go_node), cmp); }

/* ASCII nucleotides -> 0..4 */
void go_digitize(const char *text, int64_t n, uint8_t *out) {
  for (int64_t i = 0; i < n; ++i) {
    switch (text[i]) { case 'A': case 'a': out[i] = 0; break; case 'C': case 'c': out[i] = 1; break; case 'G': case 'g': out[i] = 2; break;
                       case 'T': case 't': case 'U': case 'u': out[i] = 3; break; default: out[i] = 4; }
  }
}
