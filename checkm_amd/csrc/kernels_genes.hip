// kernels_genes.hip -- the body of gene calling on the device, gfx950 only: what the gene finder CheckM runs in front of the
// marker-gene scan (`prodigal -p single -m -g 11|4`, checkm/prodigal.py:80-93) spends its time on, once the start / stop nodes exist
// (kernels_orf.hip):
//   gene_cscore_kernel   thread per START node: the hexamer log-odds of its ORF summed codon by codon FROM THE STOP towards the start
//                        (node.c: raw_coding_score, first pass -- the order of the double additions is the reference order, so the sum
//                        is the oracle's bit for bit); the 4096-entry table of the bin sits in LDS.
//   gene_rbs_kernel      thread per START node: the best Shine-Dalgarno bin (exact and one-mismatch) in the 20 bases upstream
//                        (sequence.c: shine_dalgarno_exact / _mm), against the bin's 28 weights.
//   gene_dp_kernel       the dynamic program over the nodes (dprog.c: dprog, node.c: score_connection): ONE WORKGROUP (four wavefronts)
//                        per sequence (the whole bin in the training pass, a contig in the final pass); the nodes go in order, and the up
//                        to 500 (and, behind a giant ORF, more) predecessor candidates of a node are scored 256 at a time, one per thread,
//                        then reduced to the best connection with the reference's tie rule (the LAST candidate that reaches the maximum wins).
// The oracle is oracle/gene_full.c (a restatement of Prodigal 2.6.3's single-genome mode; parity unpinned: no prodigal exists here).
#include <hip/hip_runtime.h>
#include <cstdint>
#include "gene_types.h"

namespace ckm {

#define GLD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)       /* L2-served: sees this wave's earlier stores */
#define GST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// base code at strand-relative position p of sequence [base, base + slen): forward = code & 3, reverse = 3 - (code of slen-1-p) & 3
__device__ __forceinline__ int g_code(const uint8_t *__restrict__ txt, uint64_t base, int slen, int strand, int p) {
  return strand == 1 ? (txt[base + (uint64_t)p] & 3) : 3 - (txt[base + (uint64_t)(slen - 1 - p)] & 3);
}
__device__ __forceinline__ int g_unk(const uint8_t *__restrict__ txt, uint64_t base, int slen, int strand, int p) {
  return (strand == 1 ? txt[base + (uint64_t)p] : txt[base + (uint64_t)(slen - 1 - p)]) >> 2;
}

// ---- coding score: thread per node (start nodes only) ----
__global__ void __launch_bounds__(256) gene_cscore_kernel(GeneSeqDev seqs, GeneNodesDev nd, const double *__restrict__ gene_dc /* [nbins][4096] */, uint32_t nnodes) {
  __shared__ double dc[4096];
  // a block works on nodes of ONE bin (the host pads every bin's node range to a multiple of the block size)
  const uint32_t i0 = blockIdx.x * blockDim.x;
  if (i0 >= nnodes) return;
  const uint32_t bin = nd.bin[i0];
  for (int k = threadIdx.x; k < 4096; k += blockDim.x) dc[k] = gene_dc[(size_t)bin * 4096 + k];
  __syncthreads();
  const uint32_t i = i0 + threadIdx.x;
  if (i >= nnodes || nd.type[i] == 3 || nd.type[i] == 255) return;
  const uint32_t sq = nd.seq[i];
  const uint64_t base = seqs.off[sq]; const int slen = seqs.len[sq];
  const int strand = nd.strand[i];
  // strand-relative coordinates: the start at ps, its stop at pe (pe > ps); hexamers at pe-3, pe-6, ..., ps
  const int ps = strand == 1 ? nd.ndx[i] : slen - 1 - nd.ndx[i], pe = strand == 1 ? nd.stop_val[i] : slen - 1 - nd.stop_val[i];
  double score = 0.0;
  int hi = 0;                                   // codon at j+3 (6 bits, first base lowest)
  {
    const int j = pe;                           // codon at the stop position itself: the upper half of the first hexamer (pe may lie beyond the sequence: closed ends never get here)
    if (j + 2 < slen && j >= 0) hi = g_code(seqs.txt, base, slen, strand, j) | (g_code(seqs.txt, base, slen, strand, j + 1) << 2) | (g_code(seqs.txt, base, slen, strand, j + 2) << 4);
  }
  for (int j = pe - 3; j >= ps; j -= 3) {
    const int lo = g_code(seqs.txt, base, slen, strand, j) | (g_code(seqs.txt, base, slen, strand, j + 1) << 2) | (g_code(seqs.txt, base, slen, strand, j + 2) << 4);
    score += dc[lo | (hi << 6)];
    hi = lo;
  }
  nd.cscore[i] = score;
}

// ---- Shine-Dalgarno bins: thread per node (start nodes, not edge) ----
__device__ __forceinline__ int sd_bin_exact(double c, int f) {
  if (c < 6.0) return 0;
  if (c == 6.0) return f == 2 ? 1 : f == 3 ? 2 : f == 1 ? 6 : 13;
  if (c == 8.0) return f == 3 ? 3 : f == 2 ? 11 : f == 1 ? 12 : 15;
  if (c == 9.0) return f == 3 ? 3 : f == 2 ? 11 : f == 1 ? 12 : 16;
  if (c == 11.0) return f == 3 ? 10 : f == 2 ? 20 : f == 1 ? 21 : 22;
  if (c == 12.0) return f == 3 ? 10 : f == 2 ? 20 : f == 1 ? 23 : 24;
  if (c == 14.0) return f == 3 ? 10 : f == 2 ? 25 : f == 1 ? 26 : 27;
  return 0;
}
__device__ __forceinline__ int sd_bin_mm(double c, int f) {
  if (c < 6.0) return 0;
  if (c == 6.0) return f == 3 ? 2 : f == 2 ? 4 : f == 1 ? 5 : 9;
  if (c == 7.0) return f == 3 ? 2 : f == 2 ? 7 : f == 1 ? 8 : 14;
  if (c == 9.0) return f == 3 ? 3 : f == 2 ? 17 : f == 1 ? 18 : 19;
  return 0;
}
__device__ __forceinline__ int shine_dalgarno(const uint8_t *__restrict__ txt, uint64_t base, int slen, int strand, int pos, int start, const double *rwt, int mm) {
  double match[6];
  int max_val = 0;
  const int limit = min(6, start - 4 - pos);
#pragma unroll
  for (int i = 0; i < 6; ++i) match[i] = -10.0;
  for (int i = 0; i < limit; ++i) {
    if (pos + i < 0) continue;
    const int u = g_unk(txt, base, slen, strand, pos + i), q = g_code(txt, base, slen, strand, pos + i);
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
__global__ void __launch_bounds__(256) gene_rbs_kernel(GeneSeqDev seqs, GeneNodesDev nd, const double *__restrict__ rbs_wt /* [nbins][28] */, uint32_t nnodes) {
  __shared__ double rwt[28];
  const uint32_t i0 = blockIdx.x * blockDim.x;
  if (i0 >= nnodes) return;
  if (threadIdx.x < 28) rwt[threadIdx.x] = rbs_wt[(size_t)nd.bin[i0] * 28 + threadIdx.x];
  __syncthreads();
  const uint32_t i = i0 + threadIdx.x;
  if (i >= nnodes || nd.type[i] == 3 || nd.type[i] == 255 || nd.edge[i]) return;
  const uint32_t sq = nd.seq[i];
  const uint64_t base = seqs.off[sq]; const int slen = seqs.len[sq];
  const int strand = nd.strand[i];
  const int start = strand == 1 ? nd.ndx[i] : slen - 1 - nd.ndx[i];
  int r0 = 0, r1 = 0;
  for (int j = start - 20; j <= start - 6; ++j) {
    if (j < 0) continue;
    const int c0 = shine_dalgarno(seqs.txt, base, slen, strand, j, start, rwt, 0), c1 = shine_dalgarno(seqs.txt, base, slen, strand, j, start, rwt, 1);
    if (c0 > r0) r0 = c0;
    if (c1 > r1) r1 = c1;
  }
  nd.rbs0[i] = (uint8_t)r0; nd.rbs1[i] = (uint8_t)r1;
}

// ---- the dynamic program ----
// The nodes of a sequence go in order; node i looks back over about a thousand predecessors (dprog.c: 500 nodes, and 500 more behind the
// node that far back), and scoring one connection is a chain of dependent reads: the predecessor's position / strand / type, its score and
// trace-back, the start nodes it overlaps, their coding scores.  From global memory that chain is ~3 us per round of 64 candidates
// (measured: 30 us per node, 3.1 s for the training pass of 2 Mb bins).  So the last 2048 nodes live in an LDS RING -- 36 bytes per node:
// position, stop position, trace-back, {flags, three overlapping-start offsets} packed in a word, score, connection value (GC-frame bias
// x GC score in the training pass, coding + start score in the final pass) -- filled 64 nodes ahead of the sweep by all lanes; what lies
// further back (behind a giant ORF) and the two doubles only operon neighbours need (rscore, uscore) are read from global memory.
constexpr int DPW = 2048;
struct DpRing { int ndx[DPW], sv[DPW], tb[DPW], pk[DPW], lo[DPW]; double score[DPW], val[DPW]; };
struct DpNode { int ndx, sv, strand, stop; };

struct DpSrc {
  const GeneNodesDev &nd; DpRing &r; uint32_t first; int lo_rel, hi_rel; int flag;      // nodes with relative index in [lo_rel, hi_rel) are in the ring
  __device__ __forceinline__ bool ring(int rel) const { return rel >= lo_rel && rel < hi_rel; }
  __device__ __forceinline__ DpNode node(int rel) const {
    DpNode n;
    if (ring(rel)) { const int k = rel & (DPW - 1); const int pk = r.pk[k]; n.ndx = r.ndx[k]; n.sv = r.sv[k]; n.strand = (pk & 2) ? -1 : 1; n.stop = pk & 1; }
    else { const uint32_t g = first + (uint32_t)rel; n.ndx = nd.ndx[g]; n.sv = nd.stop_val[g]; n.strand = nd.strand[g]; n.stop = nd.type[g] == 3; }
    return n;
  }
  __device__ __forceinline__ int ndx(int rel) const { return ring(rel) ? r.ndx[rel & (DPW - 1)] : nd.ndx[first + (uint32_t)rel]; }
  __device__ __forceinline__ int star(int rel, int f) const {           // relative index of the overlapping start of frame f, or -1
    if (ring(rel)) { const int pk = r.pk[rel & (DPW - 1)]; if (!(pk & 8)) { const int o = (int)(int8_t)((pk >> (8 + 8 * f)) & 0xff); return o == -128 ? -1 : rel + o; } }
    return nd.star_ptr[(size_t)(first + (uint32_t)rel) * 3 + f];
  }
  __device__ __forceinline__ double val(int rel) const { return ring(rel) ? r.val[rel & (DPW - 1)] : (flag == 0 ? nd.gcb[first + (uint32_t)rel] : nd.csc[first + (uint32_t)rel]); }
  __device__ __forceinline__ double score(int rel) const {
    if (ring(rel)) return r.score[rel & (DPW - 1)];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    return GLD(&nd.score[first + (uint32_t)rel]);
  }
  __device__ __forceinline__ int tb(int rel) const {                     // relative, or -1
    if (ring(rel)) return r.tb[rel & (DPW - 1)];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    const int t = GLD(&nd.traceb[first + (uint32_t)rel]);
    return t < 0 ? -1 : t - (int)first;
  }
  __device__ __forceinline__ double rscore(int rel) const { return nd.rscore[first + (uint32_t)rel]; }
  __device__ __forceinline__ double uscore(int rel) const { return nd.uscore[first + (uint32_t)rel]; }
};

__device__ __forceinline__ double dp_igm(const DpSrc &S, double st_wt, int k1, const DpNode &n1, int k2, const DpNode &n2) {
  double rval = 0.0; int ovlp = 0;
  if ((n1.strand == 1 && n2.strand == 1 && (n1.ndx + 2 == n2.ndx || n1.ndx - 1 == n2.ndx)) ||
      (n1.strand == -1 && n2.strand == -1 && (n1.ndx + 2 == n2.ndx || n1.ndx - 1 == n2.ndx))) {
    if (n1.strand == 1 && S.rscore(k2) < 0) rval -= S.rscore(k2);
    if (n1.strand == -1 && S.rscore(k1) < 0) rval -= S.rscore(k1);
    if (n1.strand == 1 && S.uscore(k2) < 0) rval -= S.uscore(k2);
    if (n1.strand == -1 && S.uscore(k1) < 0) rval -= S.uscore(k1);
  }
  const int dist = abs(n1.ndx - n2.ndx);
  if (n1.strand == 1 && n2.strand == 1 && n1.ndx + 2 >= n2.ndx) ovlp = 1;
  else if (n1.strand == -1 && n2.strand == -1 && n1.ndx >= n2.ndx + 2) ovlp = 1;
  if (dist > 3 * 60 || n1.strand != n2.strand) rval -= 0.15 * st_wt;
  else if ((dist <= 60 && ovlp == 0) || dist < 0.25 * 60) rval += (2.0 - (double)dist / 60) * 0.15 * st_wt;
  return rval;
}
// score of the connection p1 -> p2 (node indices relative to the sequence's first node); false: no such connection
__device__ __forceinline__ bool dp_connection(const DpSrc &S, double st_wt, int p1, int p2, const DpNode &n2, double &total, int &mark) {
  const int flag = S.flag;
  const DpNode n1 = S.node(p1);
  int left = n1.ndx, right = n2.ndx, ovlp = 0, maxfr = -1;
  double score = 0.0, scr_mod = 0.0;
  const int s1 = n1.strand, s2 = n2.strand; const bool st1 = n1.stop, st2 = n2.stop;
  if (!st1 && !st2 && s1 == s2) return false;
  else if (s1 == 1 && !st1 && s2 == -1) return false;
  else if (s1 == -1 && st1 && s2 == 1) return false;
  else if (s1 == -1 && !st1 && s2 == 1 && st2) return false;
  const int tb1 = S.tb(p1);
  if (tb1 == -1 && s1 == 1 && st1) return false;
  if (tb1 == -1 && s1 == -1 && !st1) return false;
  if (s1 == s2 && s1 == 1 && !st1 && st2) {
    if (n2.sv >= n1.ndx) return false;
    if (n1.ndx % 3 != n2.ndx % 3) return false;
    right += 2;
    if (flag == 0) scr_mod = S.val(p1); else score = S.val(p1);
  } else if (s1 == s2 && s1 == -1 && st1 && !st2) {
    if (n1.sv <= n2.ndx) return false;
    if (n1.ndx % 3 != n2.ndx % 3) return false;
    left -= 2;
    if (flag == 0) scr_mod = S.val(p2); else score = S.val(p2);
  } else if (s1 == 1 && st1 && s2 == 1 && !st2) {
    left += 2;
    if (left >= right) return false;
    if (flag == 1) score = dp_igm(S, st_wt, p1, n1, p2, n2);
  } else if (s1 == 1 && st1 && s2 == -1 && st2) {
    left += 2; right -= 2;
    if (left >= right) return false;
    double maxval = 0.0; int best_ov = 0;
    for (int i = 0; i < 3; ++i) {
      const int p3 = S.star(p2, i);
      if (p3 == -1) continue;
      const DpNode n3 = S.node(p3);
      const int ov = left - n3.sv + 1;
      if (ov <= 0 || ov >= 200) continue;
      if (ov >= n3.ndx - left) continue;
      if (tb1 == -1) continue;
      if (ov >= n3.sv - S.ndx(tb1) - 2) continue;
      const double v = flag == 1 ? S.val(p3) + dp_igm(S, st_wt, p3, n3, p2, n2) : S.val(p3);
      if (v > maxval) { maxfr = i; maxval = v; best_ov = ov; }
    }
    if (maxfr != -1) { ovlp = best_ov; if (flag == 0) scr_mod = maxval; else score = maxval; }
    else if (flag == 1) score = dp_igm(S, st_wt, p1, n1, p2, n2);
  } else if (s1 == -1 && !st1 && s2 == -1 && st2) {
    right -= 2;
    if (left >= right) return false;
    if (flag == 1) score = dp_igm(S, st_wt, p1, n1, p2, n2);
  } else if (s1 == -1 && !st1 && s2 == 1 && !st2) {
    if (left >= right) return false;
    if (flag == 1) score = dp_igm(S, st_wt, p1, n1, p2, n2);
  } else if (s1 == 1 && st1 && s2 == -1 && !st2) {
    if (n2.sv - 2 >= n1.ndx + 2) return false;
    ovlp = (n1.ndx + 2) - (n2.sv - 2) + 1;
    if (ovlp >= 200) return false;
    if ((n1.ndx + 2 - n2.sv - 2 + 1) >= (n2.ndx - n1.ndx + 3 + 1)) return false;
    const int bnd = tb1 == -1 ? 0 : S.ndx(tb1);
    if ((n1.ndx + 2 - n2.sv - 2 + 1) >= (n2.sv - 3 - bnd + 1)) return false;
    left = n2.sv - 2;
    if (flag == 0) scr_mod = S.val(p2); else score = S.val(p2) - 0.15 * st_wt;
  } else if (s1 == s2 && s1 == 1 && st1 && st2) {
    if (n2.sv >= n1.ndx) return false;
    const int p3 = S.star(p1, n2.ndx % 3);
    if (p3 == -1) return false;
    const DpNode n3 = S.node(p3);
    left = n3.ndx; right += 2;
    if (flag == 0) scr_mod = S.val(p3); else score = S.val(p3) + dp_igm(S, st_wt, p1, n1, p3, n3);
  } else if (s1 == s2 && s1 == -1 && st1 && st2) {
    if (n1.sv <= n2.ndx) return false;
    const int p3 = S.star(p2, n1.ndx % 3);
    if (p3 == -1) return false;
    const DpNode n3 = S.node(p3);
    left -= 2; right = n3.ndx;
    if (flag == 0) scr_mod = S.val(p3); else score = S.val(p3) + dp_igm(S, st_wt, p3, n3, p2, n2);
  }
  if (flag == 0) score = ((double)(right - left + 1 - (ovlp * 2))) * scr_mod;
  total = S.score(p1) + score;
  mark = maxfr;
  return true;
}

// one workgroup of DP_NT threads per sequence; seq_first[s] .. seq_first[s+1] are its nodes (already in working order); traceb is written as
// an ABSOLUTE node index (-1: none); score / traceb / ov_mark must arrive zero / -1 / -1 from the host.
// A node's ~1000 candidates are spread over the workgroup's wavefronts (a round of 64 candidates costs ~1 us of dependent LDS reads and
// divergent cases, 16 rounds per node with one wavefront: 39 us per node measured); every wavefront reduces its own best candidate by
// shuffles, the leaders' results meet in LDS and thread 0 applies the reference's tie rule across them (maximum total, then the LARGEST j:
// the sequential loop keeps the last candidate that reaches the running maximum).
constexpr int DP_NT = 256, DP_NW = DP_NT / 64;
__global__ void __launch_bounds__(DP_NT) gene_dp_kernel(GeneNodesDev nd, const uint32_t *__restrict__ seq_first, const double *__restrict__ st_wt_of_seq, uint32_t nseq, int flag) {
  __shared__ DpRing ring;
  __shared__ double red_best[DP_NW]; __shared__ int red_j[DP_NW], red_mark[DP_NW];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (uint32_t s = blockIdx.x; s < nseq; s += gridDim.x) {
    const uint32_t first = seq_first[s], end = seq_first[s + 1];
    const int nn = (int)(end - first);
    const double st_wt = st_wt_of_seq[s];
    __syncthreads();
    for (int i0 = 0; i0 < nn; i0 += 64) {
      // the next 64 nodes enter the ring (their static fields; score 0, no trace-back yet): overwrites nodes i0 - DPW .. i0 + 63 - DPW
      {
        const int rel = i0 + tid;
        if (tid < 64 && rel < nn) {
          const uint32_t g = first + (uint32_t)rel; const int k = rel & (DPW - 1);
          ring.ndx[k] = nd.ndx[g]; ring.sv[k] = nd.stop_val[g]; ring.tb[k] = -1; ring.score[k] = 0.0;
          ring.val[k] = flag == 0 ? nd.gcb[g] : nd.csc[g];
          int pk = (nd.type[g] == 3 ? 1 : 0) | (nd.strand[g] == -1 ? 2 : 0) | (nd.type[g] == 255 ? 4 : 0);
          for (int f = 0; f < 3; ++f) {
            const int sp = nd.star_ptr[(size_t)g * 3 + f]; const int o = sp < 0 ? -128 : sp - rel;
            if (sp >= 0 && (o < -127 || o > 127)) pk |= 8;          // (does not fit the packed offset: this node's overlapping starts are read from global memory)
            pk |= (o & 0xff) << (8 + 8 * f);
          }
          ring.pk[k] = pk; ring.lo[k] = (int)nd.dp_min[g];
        }
      }
      __syncthreads();
      const int i1 = min(nn, i0 + 64);
      const DpSrc S{nd, ring, first, max(0, i0 + 64 - DPW), i1, flag};
      for (int i = i0; i < i1; ++i) {
        const int pki = ring.pk[i & (DPW - 1)];
        if (pki & 4) continue;                                   // (padding node; the same word for every thread)
        const DpNode n2 = S.node(i);
        const int lo = ring.lo[i & (DPW - 1)];
        double best = -1.0; int bj = -1, bmark = -1;             // best candidate of this thread: the LAST j of the thread's that reaches its maximum (j ascends)
        for (int j = lo + tid; j < i; j += DP_NT) {
          double tot; int mark;
          if (!dp_connection(S, st_wt, j, i, n2, tot, mark)) continue;
          if (tot >= 0.0 && tot >= best) { best = tot; bj = j; bmark = mark; }
        }
        // reduction: maximum total, ties to the larger j
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) {
          const double ob = __shfl_xor(best, sft); const int oj = __shfl_xor(bj, sft), om = __shfl_xor(bmark, sft);
          if (oj >= 0 && (bj < 0 || ob > best || (ob == best && oj > bj))) { best = ob; bj = oj; bmark = om; }
        }
        if (lane == 0) { red_best[wv] = best; red_j[wv] = bj; red_mark[wv] = bmark; }
        __syncthreads();
        if (tid == 0) {
          for (int k = 1; k < DP_NW; ++k) {
            const double ob = red_best[k]; const int oj = red_j[k], om = red_mark[k];
            if (oj >= 0 && (bj < 0 || ob > best || (ob == best && oj > bj))) { best = ob; bj = oj; bmark = om; }
          }
          if (bj >= 0) {
            ring.score[i & (DPW - 1)] = best; ring.tb[i & (DPW - 1)] = bj;
            GST(&nd.score[first + (uint32_t)i], best); GST(&nd.traceb[first + (uint32_t)i], (int)first + bj); GST(&nd.ov_mark[first + (uint32_t)i], bmark);
          }
        }
        __syncthreads();
      }
    }
  }
}

void launch_gene_cscore(hipStream_t st, const GeneSeqDev &seqs, const GeneNodesDev &nd, const double *gene_dc, uint32_t nnodes) {
  if (nnodes) hipLaunchKernelGGL(gene_cscore_kernel, dim3((nnodes + 255) / 256), dim3(256), 0, st, seqs, nd, gene_dc, nnodes);
}
void launch_gene_rbs(hipStream_t st, const GeneSeqDev &seqs, const GeneNodesDev &nd, const double *rbs_wt, uint32_t nnodes) {
  if (nnodes) hipLaunchKernelGGL(gene_rbs_kernel, dim3((nnodes + 255) / 256), dim3(256), 0, st, seqs, nd, rbs_wt, nnodes);
}
void launch_gene_dp(hipStream_t st, const GeneNodesDev &nd, const uint32_t *seq_first, const double *st_wt_of_seq, uint32_t nseq, int flag) {
  if (nseq) hipLaunchKernelGGL(gene_dp_kernel, dim3(nseq), dim3(DP_NT), 0, st, nd, seq_first, st_wt_of_seq, nseq, flag);
}

}  // namespace ckm
