// kernels_genes.hip -- the cooperating kernels of gene calling on the device, gfx950 only; the thread-per-index kernels
// are the lambdas of gene_pipe.h, the per-thread arithmetic is gene_dev.h.  What CheckM runs in front of the marker-gene scan is
// `prodigal -p single -m -g 11|4` (checkm/prodigal.py:80-93); the oracle is oracle/gene_full.c (parity unpinned: no prodigal exists here).
//   chain_kernel        one wavefront per (sequence, strand, frame): start / stop nodes (node.c: add_nodes) from the codon-flag planes, 64
//                       codons per step, the sequential registers of add_nodes as prefix operations on ballots; -m masks as range counts
//                       over the plane of 50-runs of unknown bases; every node sets its bit in the node plane of its strand (its rank
//                       there is its place in the working order) and takes the next event number of its chain
//   scan kernels        exclusive prefix sums of 32-bit counts (ranks of the bit planes, chain offsets, gene and protein offsets)
//   gc_bias_kernel      one wavefront per bin: the ORDERED sum of the start nodes' GC-frame terms (node.c: record_gc_bias), 64 nodes loaded
//                       at a time, added one after the other in node order
//   gene_dp_kernel      the dynamic program (dprog.c: dprog, node.c: score_connection): one workgroup per sequence, 64 nodes per step, the last 1216 nodes in LDS
//   hexbg_kernel        hexamer histogram of a bin's training sequence in LDS (4096 counters), flushed by atomics
//   cscore / rbs        per start node: hexamer log-odds sum with the bin's table in LDS; Shine-Dalgarno bins against the bin's 28 weights
#include <mutex>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include "gene_exec.h"

namespace ckm {
void launch_orf_flags(hipStream_t stream, const uint8_t *text, uint8_t *flags, uint64_t n);
namespace gene {

#define GLD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)       /* L2-served: sees this wave's earlier stores */
#define GST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

void x_orf_flags(GExec &e, const uint8_t *ascii, unsigned long long *planes, uint64_t body) { launch_orf_flags(e.st, ascii, reinterpret_cast<uint8_t *>(planes), body); }

// ---- scans ----
constexpr int SCAN_PER_BLOCK = 4096;      // 256 threads x 16
__global__ void __launch_bounds__(256) scan_local_kernel(uint32_t *a, size_t n, uint32_t *sums) {
  __shared__ uint32_t wsum[4];
  const size_t base = (size_t)blockIdx.x * SCAN_PER_BLOCK + (size_t)threadIdx.x * 16;
  uint32_t v[16], run = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { const size_t i = base + k; const uint32_t x = i < n ? a[i] : 0u; v[k] = run; run += x; }
  // exclusive scan of the threads' totals: inside the wave by shuffles, across the four waves through LDS
  uint32_t inc = run;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d); if (lane >= d) inc += o; }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  uint32_t off = inc - run;
  for (int k = 0; k < wv; ++k) off += wsum[k];
#pragma unroll
  for (int k = 0; k < 16; ++k) { const size_t i = base + k; if (i <= n) a[i] = v[k] + off; }
  if (threadIdx.x == 255) sums[blockIdx.x] = off + run;
}
__global__ void __launch_bounds__(256) scan_sums_kernel(uint32_t *sums, uint32_t nblocks) {
  __shared__ uint32_t wsum[4]; __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t b0 = 0; b0 < nblocks; b0 += 256) {
    const uint32_t i = b0 + threadIdx.x; const uint32_t x = i < nblocks ? sums[i] : 0u;
    uint32_t inc = x; const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d); if (lane >= d) inc += o; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    uint32_t off = inc - x + carry_s;
    for (int k = 0; k < wv; ++k) off += wsum[k];
    if (i < nblocks) sums[i] = off;
    __syncthreads();
    if (threadIdx.x == 255) carry_s = off + x;
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) scan_add_kernel(uint32_t *a, size_t n, const uint32_t *sums) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i <= n) a[i] += sums[i / SCAN_PER_BLOCK];
}
void x_scan_u32(GExec &e, uint32_t *a, size_t n, GBuf &scratch) {
  const size_t nblocks = (n + 1 + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK;
  scratch.ensure((nblocks + 64) * 4);
  uint32_t *sums = scratch.as<uint32_t>();
  hipLaunchKernelGGL(scan_local_kernel, dim3((unsigned)nblocks), dim3(256), 0, e.st, a, n, sums);
  if (nblocks > 1) {
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(256), 0, e.st, sums, (uint32_t)nblocks);
    hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, e.st, a, n, sums);
  }
}

// ---- nodes of every chain ----
constexpr int ORF_MIN_GENE = 90, ORF_MIN_EDGE_GENE = 60;
struct OrfRecK { uint32_t seq; int32_t ndx, sv; uint8_t type, strand_rev, edge, pad; };

// Scan coordinate j = position on the strand being read (forward: j = i; reverse: j is the index into the reverse complement, forward
// position slen-1-j), descending from the last complete codon of the frame.  Open ends only (CheckM never passes -c).
__global__ void __launch_bounds__(64) chain_kernel(ChainArgs a) {
  if (blockIdx.x >= a.nsc) return;
  const SubChain sc = a.sc[blockIdx.x];
  const uint32_t si = sc.seq;
  const int rev = sc.rev, frame = sc.frame;
  const int slen = a.seq_len[si];
  if (slen < 3) return;
  const uint64_t base = a.seq_off[si];
  const int lane = threadIdx.x;
  const uint64_t nwin = a.nwin;
  const unsigned long long *p_stop = a.planes + (uint64_t)((rev ? 4 : 0) + (a.tt4 ? 1 : 0)) * nwin;
  const unsigned long long *p_lo = a.planes + (uint64_t)((rev ? 4 : 0) + 2) * nwin, *p_hi = a.planes + (uint64_t)((rev ? 4 : 0) + 3) * nwin;
  unsigned long long *node_plane = a.node_planes + (uint64_t)((si < a.nbins ? 0 : 2) + (rev ? 1 : 0)) * nwin;
  OrfRecK *rec = reinterpret_cast<OrfRecK *>(a.rec);
  uint32_t tbase = 0;                                               // events of this chain so far
  // Record slots are taken from the shared counter CHUNK at a time (round 6): one atomic per ~20 steps.  Round 5 took them once per step
  // -- 6 M same-address atomics at ~7 ns WERE the kernel's 42 ms for a 48-bin call, with every wavefront of the call asleep on the
  // counter (31 % of the call's wavefront-cycles: profiles/r06d) -- round 4 once per node.  The slots a wavefront does not use are
  // marked (seq = ~0) and skipped by the scatter.
  constexpr unsigned long long CHUNK = 256;
  unsigned long long chunk_next = 0, chunk_end = 0;                 // uniform
  auto take = [&](uint32_t nev) -> unsigned long long {
    if (chunk_next + nev > chunk_end) {
      for (unsigned long long k = chunk_next + (unsigned)lane; k < chunk_end; k += 64) if (k < a.cap) rec[k].seq = 0xffffffffu;
      const unsigned long long need = nev > CHUNK ? nev : CHUNK;
      unsigned long long slot = 0;
      if (lane == 0) slot = atomicAdd(a.nrec, need);
      slot = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(slot >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(slot & 0xffffffffull));
      chunk_next = slot; chunk_end = slot + need;
    }
    const unsigned long long s0 = chunk_next; chunk_next += nev;
    return s0;
  };
  auto emit = [&](int ndx_s, int type, int sv_s, int edge, uint32_t t, unsigned long long k) {
    const int ndx = rev ? slen - 1 - ndx_s : ndx_s;
    const uint64_t g = base + (uint64_t)ndx;
    atomicOr(&node_plane[g >> 6], 1ull << (g & 63));
    if (k < a.cap) {
      OrfRecK nd; nd.seq = si; nd.type = (uint8_t)type; nd.strand_rev = (uint8_t)rev; nd.edge = (uint8_t)edge; nd.pad = 0;
      nd.ndx = ndx; nd.sv = rev ? slen - 1 - sv_s : sv_s;
      rec[k] = nd; a.rec_t[k] = t; a.rec_c[k] = sc.chain;
    }
  };
  int last = sc.top;
  bool last_real = sc.after_stop, saw = false, any_stop = sc.after_stop;
  const int bottom = sc.bottom;
  for (int jhi = sc.after_stop ? sc.top - 3 : sc.top; jhi >= bottom; jhi -= 192) {
    const int j = jhi - 3 * lane;
    const bool in = j >= bottom;
    bool is_stop = false; int st = -1;                           // st: -1 none, 0 ATG, 1 GTG, 2 TTG
    if (in) {
      const uint64_t pos = base + (uint64_t)(rev ? slen - 1 - j : j);
      const uint64_t wi = pos >> 6; const int bit = (int)(pos & 63);
      is_stop = (p_stop[wi] >> bit) & 1ull;
      st = (int)(((p_lo[wi] >> bit) & 1ull) | (((p_hi[wi] >> bit) & 1ull) << 1)) - 1;
    }
    const unsigned long long stops = __ballot(is_stop);
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));          // lanes scanned before me
    const unsigned long long sb = stops & below;
    const int qlane = sb ? 63 - __clzll((long long)sb) : -1;
    const int my_last = qlane >= 0 ? jhi - 3 * qlane : last;
    const bool my_last_real = qlane >= 0 ? true : last_real;
    const bool my_any_stop = any_stop || sb != 0ull;
    const int mind = my_any_stop ? ORF_MIN_GENE : ORF_MIN_EDGE_GENE;
    bool start_node = false, edge_node = false;
    if (in && !is_stop && my_last < slen) {
      if (st >= 0 && my_last - j + 3 >= mind) start_node = true;
      else if (j <= 2 && (my_last - j) > ORF_MIN_EDGE_GENE) edge_node = true;
    }
    if ((start_node || edge_node) && a.r50) {
      // -m: a start whose reading frame crosses a run of >= 50 unknown bases is not a node (forward coordinates x .. y of the frame)
      const int x = rev ? slen - 1 - my_last : j, y = rev ? slen - 1 - j : my_last;
      const int x0 = x - 49 > 0 ? x - 49 : 0;
      if (plane_rank(a.pr50, a.r50, base + (uint64_t)y + 1) != plane_rank(a.pr50, a.r50, base + (uint64_t)x0)) start_node = edge_node = false;
    }
    const unsigned long long starts = __ballot(start_node || edge_node);
    bool stop_node = false;
    if (is_stop) {
      // was a start recorded between the previous stop (or the chunk's beginning, with the carry) and me?
      const unsigned long long between = starts & below & (qlane >= 0 ? ~(~0ull >> (63 - qlane)) : ~0ull);
      stop_node = (between != 0ull) || (qlane < 0 && saw);
    }
    const unsigned long long events = starts | __ballot(stop_node);
    const uint32_t before = (uint32_t)__popcll(events & below), nev = (uint32_t)__popcll(events);
    const uint32_t t = tbase + before;
    unsigned long long slot0 = 0;
    if (nev) slot0 = take(nev);
    if (start_node) emit(j, st, my_last, 0, t, slot0 + before);
    if (edge_node) emit(j, 0, my_last, 1, t, slot0 + before);
    if (stop_node) emit(my_last, 3, j, my_last_real ? 0 : 1, t, slot0 + before);
    tbase += nev;
    if (stops) {
      const int ql = 63 - __clzll((long long)stops);
      last = jhi - 3 * ql; last_real = true; any_stop = true;
      saw = (starts & (ql == 63 ? 0ull : (~0ull << (ql + 1)))) != 0ull;
    } else saw = saw || starts != 0ull;
  }
  if (saw) { const unsigned long long k = take(1); if (lane == 0) emit(last, 3, frame - 6, last_real ? 0 : 1, tbase, k); tbase++; }
  for (unsigned long long k = chunk_next + (unsigned)lane; k < chunk_end; k += 64) if (k < a.cap) rec[k].seq = 0xffffffffu;
  if (lane == 0) a.chain_cnt[sc.chain] = tbase;
}
void x_chain(GExec &e, const ChainArgs &a) {
  if (a.nsc) hipLaunchKernelGGL(chain_kernel, dim3(a.nsc), dim3(64), 0, e.st, a);
}

// ---- ordered GC-bias sums ----
__global__ void __launch_bounds__(64) gc_bias_kernel(Nodes nd, const uint32_t *__restrict__ seq_lo, const uint32_t *__restrict__ seq_n, uint32_t nbins, double *__restrict__ bias) {
  const uint32_t b = blockIdx.x;
  if (b >= nbins) return;
  const uint32_t lo = seq_lo[b], nn = seq_n[b];
  const int lane = threadIdx.x;
  double acc = 0.0;                                                 // lane k: the sum of class k (k = 0, 1, 2)
  if (nn == 0) { if (lane < 3) bias[(size_t)b * 3 + lane] = 0.0; return; }
  // The next 64 nodes are asked for before the current 64 are added (the loads of a chunk were two round trips that nothing else hid: the type, then -- under
  // its test -- class and term), and the adds are branch-free: a term goes to its class's sum, +0.0 to the other two -- which leaves them
  // as they are (the sums start at +0.0 and every term is a non-negative product, so no sum is ever -0.0).
  auto fetch = [&](uint32_t i0, int &cls, double &term) {
    const uint32_t i = i0 + (uint32_t)lane;
    const uint32_t g = lo + (i < nn ? i : nn - 1);                  // (all three loads at once: class and term are zero-filled where no start node wrote them)
    const int t = nd.type[g]; const int c = nd.gcb_cls[g]; const double v = nd.gcb_term[g];
    const bool start = i < nn && t < G_STOP;
    cls = start ? c : 3; term = start ? v : 0.0;                    // class 3: not a start node
  };
  int cls, ncls = 3; double term, nterm = 0.0;
  fetch(0, cls, term);
  for (uint32_t i0 = 0; i0 < nn; i0 += 64) {
    if (i0 + 64 < nn) fetch(i0 + 64, ncls, nterm);
    // one after the other, in node order: lane l's class and term through v_readlane (uniform) -- a __shfl per element (ds_bpermute: an
    // LDS round trip) made this kernel 16 ms of a 48-bin call's critical path
    const int tlo = (int)(__builtin_bit_cast(unsigned long long, term) & 0xffffffffull), thi = (int)(__builtin_bit_cast(unsigned long long, term) >> 32);
#pragma unroll
    for (int l = 0; l < 64; ++l) {
      // lane k keeps class k's sum: every lane adds node l's term if its number is node l's class, +0.0 otherwise.  Vector instructions
      // only -- three v_readlane, a compare, two selects, one add: with the selects on the scalar unit (s_cmp / s_cselect between the
      // v_readlane and the v_add_f64) a node cost 200 cycles, the two units waiting on each other's registers
      const int c = __builtin_amdgcn_readlane(cls, l);
      const double v = __builtin_bit_cast(double, ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(thi, l) << 32) | (unsigned)__builtin_amdgcn_readlane(tlo, l));
      acc += lane == c ? v : 0.0;
    }
    cls = ncls; term = nterm;
  }
  const int alo = (int)(__builtin_bit_cast(unsigned long long, acc) & 0xffffffffull), ahi = (int)(__builtin_bit_cast(unsigned long long, acc) >> 32);
  auto sum_of = [&](int k) { return __builtin_bit_cast(double, ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(ahi, k) << 32) | (unsigned)__builtin_amdgcn_readlane(alo, k)); };
  double acc0 = sum_of(0), acc1 = sum_of(1), acc2 = sum_of(2);
  if (lane == 0) {
    const double tot = acc0 + acc1 + acc2;
    acc0 *= (3.0 / tot); acc1 *= (3.0 / tot); acc2 *= (3.0 / tot);
    bias[(size_t)b * 3] = acc0; bias[(size_t)b * 3 + 1] = acc1; bias[(size_t)b * 3 + 2] = acc2;
  }
}
void x_gc_bias(GExec &e, const Nodes &nd, const uint32_t *seq_lo, const uint32_t *seq_n, uint32_t nbins, double *bias) {
  if (nbins) hipLaunchKernelGGL(gc_bias_kernel, dim3(nbins), dim3(64), 0, e.st, nd, seq_lo, seq_n, nbins, bias);
}

// ---- where a sequence's trace-back begins: one wavefront per sequence ----
__global__ void __launch_bounds__(64) path_ends_kernel(Nodes nd, const uint32_t *__restrict__ seq_lo, const uint32_t *__restrict__ seq_n, uint32_t nseq, int32_t *__restrict__ end_rel) {
  const uint32_t s = blockIdx.x;
  if (s >= nseq) return;
  const uint32_t lo = seq_lo[s]; const int nn = (int)seq_n[s], lane = threadIdx.x;
  double best = -1.0; int bi = -1;                        // (the reference starts from max_sc = -1 and takes strictly greater scores, scanning from the last node down)
  for (int i = lane; i < nn; i += 64) {
    const int str = nd.strand[lo + i]; const bool st = nd.type[lo + i] == G_STOP;
    if ((str == 1 && !st) || (str == -1 && st)) continue;
    const double sc = nd.score[lo + i];
    if (sc > best || (sc == best && bi >= 0 && i > bi)) { if (sc > -1.0) { best = sc; bi = i; } }
  }
#pragma unroll
  for (int sft = 32; sft >= 1; sft >>= 1) {
    const double ob = __shfl_xor(best, sft); const int oi = __shfl_xor(bi, sft);
    if (oi >= 0 && (bi < 0 || ob > best || (ob == best && oi > bi))) { best = ob; bi = oi; }
  }
  if (lane == 0) end_rel[s] = bi;
}
void x_path_ends(GExec &e, const Nodes &nd, const uint32_t *seq_lo, const uint32_t *seq_n, uint32_t nseq, int32_t *end_rel) {
  if (nseq) hipLaunchKernelGGL(path_ends_kernel, dim3(nseq), dim3(64), 0, e.st, nd, seq_lo, seq_n, nseq, end_rel);
}

// ---- hexamer sums and Shine-Dalgarno bins ----
// (the bin's 32 KB table is read through the caches, not staged in LDS: many calls are in flight, the dynamic programs of the others hold
//  most of every compute unit's LDS, and a kernel that asks for 32 KB of it waits for them)
__global__ void __launch_bounds__(256) cscore_kernel(const uint8_t *__restrict__ code, const uint64_t *__restrict__ seq_off, const int32_t *__restrict__ seq_len, Nodes nd,
                                                     const double *__restrict__ gene_dc /* [nbins][4096] */, uint32_t nnodes) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nnodes || nd.type[i] >= G_STOP) return;
  const uint32_t sq = nd.seq[i];
  const int slen = seq_len[sq], strand = nd.strand[i];
  const int ps = strand == 1 ? nd.ndx[i] : slen - 1 - nd.ndx[i], pe = strand == 1 ? nd.sv[i] : slen - 1 - nd.sv[i];
  nd.cscore[i] = node_cscore(code + seq_off[sq], slen, strand, ps, pe, gene_dc + (size_t)nd.bin[i] * 4096);
}
__global__ void __launch_bounds__(256) rbs_kernel(const uint8_t *__restrict__ code, const uint64_t *__restrict__ seq_off, const int32_t *__restrict__ seq_len, Nodes nd,
                                                  const double *__restrict__ rbs_wt /* [nbins][28] */, uint32_t nnodes) {
  __shared__ double rwt[28];
  const uint32_t i0 = blockIdx.x * blockDim.x;
  if (i0 >= nnodes) return;
  if (nd.type[i0] == G_PAD) return;
  if (threadIdx.x < 28) rwt[threadIdx.x] = rbs_wt[(size_t)nd.bin[i0] * 28 + threadIdx.x];
  __syncthreads();
  const uint32_t i = i0 + threadIdx.x;
  if (i >= nnodes || nd.type[i] >= G_STOP || nd.edge[i]) return;
  const uint32_t sq = nd.seq[i];
  const GSeq q{code + seq_off[sq], seq_len[sq]};
  const int strand = nd.strand[i], start = strand == 1 ? nd.ndx[i] : q.slen - 1 - nd.ndx[i];
  int r0, r1;
  node_rbs(q, strand, start, rwt, r0, r1);
  nd.rbs0[i] = (uint8_t)r0; nd.rbs1[i] = (uint8_t)r1;
}
void x_cscore(GExec &e, const uint8_t *code, const uint64_t *seq_off, const int32_t *seq_len, const Nodes &nd, const double *gene_dc, uint32_t n) {
  if (n) hipLaunchKernelGGL(cscore_kernel, dim3((n + 255) / 256), dim3(256), 0, e.st, code, seq_off, seq_len, nd, gene_dc, n);
}
void x_rbs(GExec &e, const uint8_t *code, const uint64_t *seq_off, const int32_t *seq_len, const Nodes &nd, const double *rbs_wt, uint32_t n) {
  if (n) hipLaunchKernelGGL(rbs_kernel, dim3((n + 255) / 256), dim3(256), 0, e.st, code, seq_off, seq_len, nd, rbs_wt, n);
}

// ---- background words of the upstream-motif training's rounds ----
// A workgroup per eighth of a bin's nodes, the counters in LDS, one flush: the thread-per-node kernel sent 52 atomics per start (first
// round) or up to seven (stage 1) to tables in memory -- 42 M + 45 M per 48-bin call, and its wavefronts stayed resident until they had
// drained (27 % of a call's wavefront-cycles: profiles/r06o).  Stage 0: 5440 counters (64 + 256 + 1024 + 4096 words of 3 .. 6 bases).
// Stages 1 / 2: four spacer classes of those, 21760 counters held as 16-bit halves of 10880 words (a part has fewer than 65536 starts).
constexpr int MOT_WORDS = 5440;
__device__ __forceinline__ int mot_off(int i) { return ((64 << (2 * i)) - 64) / 3; }
__global__ void __launch_bounds__(256) motif_bg0_kernel(Nodes nd, const int32_t *__restrict__ seq_len, const MotifPart *__restrict__ parts, uint32_t *__restrict__ bg0) {
  __shared__ uint32_t h[MOT_WORDS];
  const MotifPart pt = parts[blockIdx.x];
  for (int k = threadIdx.x; k < MOT_WORDS; k += 256) h[k] = 0;
  __syncthreads();
  for (uint32_t x = pt.lo + threadIdx.x; x < pt.hi; x += 256) {
    if (nd.type[x] >= G_STOP || nd.edge[x] == 1) continue;
    const int sl = seq_len[nd.seq[x]], strand = nd.strand[x], start = strand == 1 ? nd.ndx[x] : sl - 1 - nd.ndx[x];
    motif_words_stage0(nd.upw[x], start, [&](int i, int w) { atomicAdd(&h[mot_off(i) + w], 1u); });
  }
  __syncthreads();
  for (int k = threadIdx.x; k < MOT_WORDS; k += 256) {
    const uint32_t v = h[k];
    if (!v) continue;
    const int i = k < 64 ? 0 : k < 320 ? 1 : k < 1344 ? 2 : 3;
    atomicAdd(&bg0[((size_t)pt.slot * 4 + i) * 4096 + (k - mot_off(i))], v);
  }
}
__global__ void __launch_bounds__(256) motif_bg12_kernel(int stage, Nodes nd, const int32_t *__restrict__ seq_len, const MotifPart *__restrict__ parts, uint32_t *__restrict__ tab) {
  __shared__ uint32_t h[2 * MOT_WORDS];                            // entry e = 4 * mot_off(i) + sp * 4^(i+3) + w lives in half (e & 1) of word e >> 1
  const MotifPart pt = parts[blockIdx.x];
  for (int k = threadIdx.x; k < 2 * MOT_WORDS; k += 256) h[k] = 0;
  __syncthreads();
  for (uint32_t x = pt.lo + threadIdx.x; x < pt.hi; x += 256) {
    if (nd.type[x] >= G_STOP || nd.edge[x] == 1) continue;
    const int sl = seq_len[nd.seq[x]], strand = nd.strand[x], start = strand == 1 ? nd.ndx[x] : sl - 1 - nd.ndx[x];
    motif_words_stage12(nd.mot[x], nd.upw[x], start, stage, [&](int i, int sp, int w) {
      const int e = 4 * mot_off(i) + (sp << (2 * (i + 3))) + w;
      atomicAdd(&h[e >> 1], (e & 1) ? 0x10000u : 1u);
    });
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 2 * MOT_WORDS; k += 256) {
    const uint32_t v = h[k];
    if (!v) continue;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const uint32_t c = half ? v >> 16 : v & 0xffffu;
      if (!c) continue;
      const int e = 2 * k + half;
      const int i = e < 4 * 64 ? 0 : e < 4 * 320 ? 1 : e < 4 * 1344 ? 2 : 3, r = e - 4 * mot_off(i), sp = r >> (2 * (i + 3)), w = r & ((1 << (2 * (i + 3))) - 1);
      atomicAdd(&tab[(size_t)pt.slot * 65536 + ((size_t)i * 4 + sp) * 4096 + w], c);
    }
  }
}
void x_motif_bg(GExec &e, int stage, const Nodes &nd, const int32_t *seq_len, const MotifPart *parts, uint32_t nparts, uint32_t *tab) {
  if (!nparts) return;
  if (stage == 0) hipLaunchKernelGGL(motif_bg0_kernel, dim3(nparts), dim3(256), 0, e.st, nd, seq_len, parts, tab);
  else hipLaunchKernelGGL(motif_bg12_kernel, dim3(nparts), dim3(256), 0, e.st, stage, nd, seq_len, parts, tab);
}

// ---- hexamer background ----
constexpr int HEX_SLICE = 65536;
__global__ void __launch_bounds__(256) hexbg_kernel(const uint8_t *__restrict__ code, const uint64_t *__restrict__ seq_off, const int32_t *__restrict__ seq_len, uint32_t *__restrict__ hist) {
  __shared__ uint32_t h[4096];
  const uint32_t b = blockIdx.y;
  const int slen = seq_len[b];
  const long long a = (long long)blockIdx.x * HEX_SLICE;
  if (a >= slen - 5) return;
  for (int k = threadIdx.x; k < 4096; k += 256) h[k] = 0;
  __syncthreads();
  const uint8_t *c = code + seq_off[b];
  const long long z = a + HEX_SLICE < slen - 5 ? a + HEX_SLICE : slen - 5;
  for (long long i = a + threadIdx.x; i < z; i += 256) {
    int f = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) f |= (c[i + k] & 3) << (2 * k);
    atomicAdd(&h[f], 1u);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 4096; k += 256) if (h[k]) atomicAdd(&hist[(size_t)b * 4096 + k], h[k]);
}
void x_hexamer_background(GExec &e, const uint8_t *code, const uint64_t *seq_off, const int32_t *seq_len, uint32_t nbins, int max_len, uint32_t *hist) {
  if (!nbins) return;
  HIPCHK(hipMemsetAsync(hist, 0, (size_t)nbins * 4096 * 4, e.st));
  const uint32_t slices = (uint32_t)((std::max(max_len, 6) - 5 + HEX_SLICE - 1) / HEX_SLICE);      // (of the longest bin; a shorter bin's surplus blocks return at once)
  hipLaunchKernelGGL(hexbg_kernel, dim3(slices, nbins), dim3(256), 0, e.st, code, seq_off, seq_len, hist);
}

// ---- the dynamic program ----
// The nodes of a sequence go in order; node i looks back over the 1000 nodes before it (dprog.c: 500 nodes, and 500 more behind the node
// that far back; further when a giant open reading frame sits there).  Of the sweep's own state a connection j -> i reads the score of j
// and -- for four of the ten class pairs -- whether j has a predecessor and where it lies (gene_dev.h: dp_connection_s); everything else
// is fixed before the sweep starts.  Rounds 4-5 took the nodes one at a time, a barrier per node (3.6-4.8 us per node, 58 % of the
// wavefronts' cycles waiting: profiles/r05t).  Round 6 takes them 64 at a time:
//   (A) a wavefront takes a node i of the block (heaviest classes first, dealt by an LDS counter) and scores its candidates class by class
//       out of four LDS rings of node indices -- a wavefront's 64 candidates share one class, so the connection function folds to that
//       class's cases, and the six class pairs that cannot connect are never touched.  A candidate BEFORE the block is final: connection +
//       score, into the wavefront's running best (one DPP reduction per node).  A candidate INSIDE the block is not: its connection WITHOUT
//       its score goes into a 64 x 64 table in LDS (the pairs that read the candidate's predecessor -- dp_pair_dynamic -- are left out).
//       A forward stop's / reverse start's candidates begin at the first node its open reading frame can reach (dp_pos_floor: a binary
//       search over the ring when the node enters), not 1000 nodes back;
//   (B) one wavefront, lane per node of the block: for t = 0 .. 63 node t is final (every candidate before it has been offered), its score
//       and predecessor are broadcast by v_readlane, and every lane behind t takes `score(t) + table[t][lane]` -- one LDS read, one add, one
//       compare; only a forward stop t meeting reverse nodes runs the connection function here.
// (B) is a chain of 64 dependent steps on one wavefront; with (A) and (B) in turn (the first form of this round, 83 ms on 48 bins) the other
// fifteen wait for it.  So (A) is cut in two and the blocks are pipelined -- per block k, two phases and two barriers:
//   P(k)  wavefront 0: (B) of block k | wavefront 1: block k + 2 enters the rings (its global-memory reads hide here), then joins | the
//         others: (A1) of block k + 1 -- its candidates BEFORE block k, nine tenths of them, all final already;
//   Q(k)  all wavefronts: (A2) of block k + 1 -- its candidates inside block k (final now) and its table; block k leaves for global memory.
// The last 1216 nodes live in an LDS ring of 40-byte records read with wide loads: {position, stop position, flags + packed
// overlapping-start offsets, trace-back and window start as 16-bit distances}, {score, connection value}, {class counts}.  What does not
// fit the rings (a window that starts behind a giant open reading frame, more than 896 nodes of one class in a window) goes through the
// generic loop with global-memory fall-backs.
// One rule the first pipelined form broke (98.6k genes of 99,984, and not the same ones twice): what ONE lane stores to LDS and the other
// lanes of the same wavefront load back later in program order -- the class totals of dp_enter -- needs a barrier (or the loads made
// volatile) between the two: to the compiler that is one thread storing under `lane == 0` and loading again, and it forwards the value it
// already holds.  Two dp_enter calls in a row by one wavefront (the prologue) are therefore separated by __syncthreads.
constexpr int DPB = 64;                                       // nodes per block
// (the class totals: written by the entering wavefront's lane 0, read by its other lanes and by the scoring wavefronts -- never plain accesses)
#define DP_LDS_LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define DP_LDS_ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
constexpr int DPW = 1216, DPC = 1024, DP_FAR = 0xffff;        // 1216 = the 1000-node window + the block being resolved + the two entered ahead (+ slack)
struct DpRec { int ndx, sv, pk; uint32_t tbl; };            // pk: bit 0 stop, 1 reverse, 3 stars-in-global, 4-5 (overlap mark + 1), 8.. three signed byte offsets; tbl: trace-back distance | window distance << 16
struct DpRing { DpRec rec[DPW]; double2 sv2[DPW]; ushort4 cnt[DPW]; unsigned short cls[4][DPC]; };
struct DpBlock {
  double table[DPB * DPB];                                   // [t][(l + t) & 63]: connection t -> l without t's score; -inf: none
  double best1[2][DPB]; int key1[2][DPB];                    // (A1)'s result per node of a block, by the block's parity
  double best2[DPB]; int key2[DPB];                          // (A2)'s
  int lo_eff[2][DPB];                                        // where a node's candidates begin (window start, or the position floor), by the block's parity
  unsigned char order[2][DPB];                               // a block's nodes, heaviest class first
  uint32_t next1, next2;                                     // the dealers of (A1) and (A2)
};
__device__ __forceinline__ int dp_slot(int rel) { return rel % DPW; }
__device__ __forceinline__ int dp_tab(int t, int l) { return t * DPB + ((l + t) & (DPB - 1)); }      // (the rotation spreads a column over the banks)

template <int FLAG>
struct DpSrc {
  static constexpr int flag = FLAG;
  const Nodes &nd; DpRing &r; uint32_t first; int lo_rel, hi_rel;                 // nodes with relative index in [lo_rel, hi_rel) are in the ring
  int blk0;                                                                       // nodes from here on are not final: (A) is told "has a predecessor", (B) decides
  __device__ __forceinline__ bool ring(int rel) const { return rel >= lo_rel && rel < hi_rel; }
  __device__ __forceinline__ DpNode node(int rel) const {
    DpNode n;
    if (ring(rel)) { const DpRec q = r.rec[dp_slot(rel)]; n.ndx = q.ndx; n.sv = q.sv; n.strand = (q.pk & 2) ? -1 : 1; n.stop = q.pk & 1; }
    else { const uint32_t g = first + (uint32_t)rel; n.ndx = nd.ndx[g]; n.sv = nd.sv[g]; n.strand = nd.strand[g]; n.stop = nd.type[g] == 3; }
    return n;
  }
  __device__ __forceinline__ int ndx(int rel) const { return ring(rel) ? r.rec[dp_slot(rel)].ndx : nd.ndx[first + (uint32_t)rel]; }
  __device__ __forceinline__ int star(int rel, int f) const {           // relative index of the overlapping start of frame f, or -1
    if (ring(rel)) { const int pk = r.rec[dp_slot(rel)].pk; if (!(pk & 8)) { const int o = (int)(int8_t)((pk >> (8 + 8 * f)) & 0xff); return o == -128 ? -1 : rel + o; } }
    return nd.star[(size_t)(first + (uint32_t)rel) * 3 + f];
  }
  __device__ __forceinline__ double val(int rel) const { return ring(rel) ? r.sv2[dp_slot(rel)].y : (FLAG == 0 ? nd.gcb[first + (uint32_t)rel] : nd.csc[first + (uint32_t)rel]); }
  __device__ __forceinline__ double score(int rel) const {
    if (ring(rel)) return r.sv2[dp_slot(rel)].x;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    return GLD(&nd.score[first + (uint32_t)rel]);
  }
  __device__ __forceinline__ int tb(int rel) const {                     // relative, or -1
    if (rel >= blk0) return 0;
    if (ring(rel)) { const uint32_t d = r.rec[dp_slot(rel)].tbl & 0xffffu; if (d != (uint32_t)DP_FAR) return d == 0 ? -1 : rel - (int)d; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    return GLD(&nd.traceb[first + (uint32_t)rel]);
  }
  __device__ __forceinline__ double rscore(int rel) const { return nd.rscore[first + (uint32_t)rel]; }
  __device__ __forceinline__ double uscore(int rel) const { return nd.uscore[first + (uint32_t)rel]; }
  __device__ __forceinline__ DpNode node3(int rel) const { return node(rel); }
  __device__ __forceinline__ double val3(int rel) const { return val(rel); }
  __device__ __forceinline__ int ndx_any(int rel) const { return ndx(rel); }
};
// The same source for candidates that are KNOWN to sit in the ring (the class path: every candidate comes out of a class ring, node i is
// in the ring): position, stop position, flags, trace-back distance in ONE 16-byte LDS read, score and value in another, no
// ring-or-global question on the way.  What is reached through a candidate (overlapping starts, the node its trace-back points to) may lie
// outside and goes through the general accessors.
template <int FLAG>
struct DpRingSrc {
  static constexpr int flag = FLAG;
  const DpSrc<FLAG> &g; mutable int last_rel; mutable DpRec last;
  __device__ __forceinline__ const DpRec &rec(int rel) const { if (rel != last_rel) { last = g.r.rec[dp_slot(rel)]; last_rel = rel; } return last; }
  __device__ __forceinline__ DpNode node(int rel) const { const DpRec &q = rec(rel); DpNode n; n.ndx = q.ndx; n.sv = q.sv; n.strand = (q.pk & 2) ? -1 : 1; n.stop = q.pk & 1; return n; }
  __device__ __forceinline__ int star(int rel, int f) const {
    const int pk = rec(rel).pk;
    if (!(pk & 8)) { const int o = (int)(int8_t)((pk >> (8 + 8 * f)) & 0xff); return o == -128 ? -1 : rel + o; }
    return g.nd.star[(size_t)(g.first + (uint32_t)rel) * 3 + f];
  }
  __device__ __forceinline__ double val(int rel) const { return g.r.sv2[dp_slot(rel)].y; }
  __device__ __forceinline__ double score(int rel) const { return g.r.sv2[dp_slot(rel)].x; }
  __device__ __forceinline__ int tb(int rel) const {
    if (rel >= g.blk0) return 0;
    const uint32_t d = rec(rel).tbl & 0xffffu;
    if (d != (uint32_t)DP_FAR) return d == 0 ? -1 : rel - (int)d;
    return g.tb(rel);
  }
  __device__ __forceinline__ double rscore(int rel) const { return g.rscore(rel); }
  __device__ __forceinline__ double uscore(int rel) const { return g.uscore(rel); }
  __device__ __forceinline__ DpNode node3(int rel) const { return g.node(rel); }
  __device__ __forceinline__ double val3(int rel) const { return g.val(rel); }
  __device__ __forceinline__ int ndx_any(int rel) const { return g.ndx(rel); }
};

// (best total, candidate key) of the wavefront: xor butterfly on data-parallel-primitive moves and lane swaps -- no LDS round trip (ds_bpermute, what
// __shfl_xor compiles to, is six dependent ~100-cycle trips per node)
template <int CTRL> __device__ __forceinline__ int dp_dpp(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ void dp_merge(double &best, int &key, double ob, int ok) {
  if (ok >= 0 && (key < 0 || ob > best || (ob == best && ok > key))) { best = ob; key = ok; }
}
__device__ __forceinline__ int dp_lo32(double d) { return (int)(__builtin_bit_cast(unsigned long long, d) & 0xffffffffull); }
__device__ __forceinline__ int dp_hi32(double d) { return (int)(__builtin_bit_cast(unsigned long long, d) >> 32); }
__device__ __forceinline__ double dp_mk64(int l, int h) { return __builtin_bit_cast(double, ((unsigned long long)(unsigned)h << 32) | (unsigned)l); }
__device__ __forceinline__ void dp_wave_best(double &best, int &key) {
#define CKM_DP_STEP(CTRL) { const double ob = dp_mk64(dp_dpp<CTRL>(dp_lo32(best)), dp_dpp<CTRL>(dp_hi32(best))); const int ok = dp_dpp<CTRL>(key); dp_merge(best, key, ob, ok); }
  CKM_DP_STEP(0xB1)      /* quad_perm [1,0,3,2]: partner xor 1 */
  CKM_DP_STEP(0x4E)      /* quad_perm [2,3,0,1]: partner xor 2 */
  CKM_DP_STEP(0x141)     /* row_half_mirror: pairs the two quads of a half row */
  CKM_DP_STEP(0x140)     /* row_mirror: pairs the two halves of a row */
#undef CKM_DP_STEP
  {   // across the rows of 16 lanes: the row leaders' values through readlane (uniform), merged by every lane
    double b1 = dp_mk64(__builtin_amdgcn_readlane(dp_lo32(best), 16), __builtin_amdgcn_readlane(dp_hi32(best), 16)); int k1 = __builtin_amdgcn_readlane(key, 16);
    double b2 = dp_mk64(__builtin_amdgcn_readlane(dp_lo32(best), 32), __builtin_amdgcn_readlane(dp_hi32(best), 32)); int k2 = __builtin_amdgcn_readlane(key, 32);
    double b3 = dp_mk64(__builtin_amdgcn_readlane(dp_lo32(best), 48), __builtin_amdgcn_readlane(dp_hi32(best), 48)); int k3 = __builtin_amdgcn_readlane(key, 48);
    double b0 = dp_mk64(__builtin_amdgcn_readlane(dp_lo32(best), 0), __builtin_amdgcn_readlane(dp_hi32(best), 0)); int k0 = __builtin_amdgcn_readlane(key, 0);
    best = b0; key = k0; dp_merge(best, key, b1, k1); dp_merge(best, key, b2, k2); dp_merge(best, key, b3, k3);
  }
}
// the oldest node no block entered while block i0 is resolved overwrites (that block and the two behind it are in the ring by then)
__device__ __forceinline__ int dp_ring_lo(int i0) { const int v = i0 + 3 * DPB - DPW; return v > 0 ? v : 0; }

// the 64 nodes from e0 on enter the rings (one wavefront; the slots they take over are 1216 nodes / 1024 class members back)
template <int FLAG>
__device__ __forceinline__ void dp_enter(const Nodes &nd, DpRing &ring, DpBlock &blk, uint32_t *ring_tot, uint32_t first, int nn, int e0, int lane) {
  const int rel = e0 + lane; const bool in = rel < nn;
  int cls = -1, pk = 0; const uint32_t g = first + (uint32_t)(in ? rel : 0);
  if (in) { const bool st = nd.type[g] == 3; const int str = nd.strand[g]; cls = dp_class(str, st); pk = (st ? 1 : 0) | (str == -1 ? 2 : 0); }
  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  uint32_t before[4], tot[4];
  unsigned long long m[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { m[c] = __ballot(cls == c); tot[c] = DP_LDS_LD(&ring_tot[c]); before[c] = tot[c] + (uint32_t)__popcll(m[c] & below); tot[c] += (uint32_t)__popcll(m[c]); }
  int lo = 0, sv = 0;
  if (in) {
    const int k = dp_slot(rel);
    for (int f = 0; f < 3; ++f) {
      const int sp = nd.star[(size_t)g * 3 + f]; const int o = sp < 0 ? -128 : sp - rel;
      if (sp >= 0 && (o < -127 || o > 127)) pk |= 8;          // (does not fit the packed offset: this node's overlapping starts are read from global memory)
      pk |= (o & 0xff) << (8 + 8 * f);
    }
    lo = (int)nd.dp_min[g]; const uint32_t lod = rel - lo < DP_FAR ? (uint32_t)(rel - lo) : (uint32_t)DP_FAR;
    sv = nd.sv[g];
    DpRec q; q.ndx = nd.ndx[g]; q.sv = sv; q.pk = pk; q.tbl = lod << 16;                 // trace-back distance 0: none yet
    ring.rec[k] = q;
    ring.sv2[k] = make_double2(0.0, FLAG == 0 ? nd.gcb[g] : nd.csc[g]);
    ring.cnt[k] = make_ushort4((unsigned short)before[0], (unsigned short)before[1], (unsigned short)before[2], (unsigned short)before[3]);
    ring.cls[cls][before[cls] & (DPC - 1)] = (unsigned short)rel;
  }
  if (lane == 0) { for (int c = 0; c < 4; ++c) DP_LDS_ST(&ring_tot[c], tot[c]); }
  // where the node's candidates begin: a forward stop / reverse start reaches back to dp_pos_floor only (the records of this block are in
  // the ring by now -- LDS operations of one wavefront complete in order)
  if (in) {
    int le = lo;
    if (dp_class_pos_bounded(cls) && lo >= dp_ring_lo(e0 - 2 * DPB)) {      // (the nodes are entered two blocks before they are scored: what is in the ring now is still there then)
      const int fl = dp_pos_floor(sv);
      int a = lo, b = rel;                                     // smallest j in [lo, rel] with j == rel or ndx(j) >= fl
      while (a < b) { const int mid = (a + b) >> 1; if (ring.rec[dp_slot(mid)].ndx >= fl) b = mid; else a = mid + 1; }
      le = a;
    }
    blk.lo_eff[(e0 / DPB) & 1][lane] = le;
  }
  // heaviest first: reverse stops (three classes of candidates, the overlapping-start loop), forward starts (two), the position-bounded rest
  const unsigned long long h0 = m[3], h1 = m[0], h2 = m[1] | m[2];
  if (in) {
    const int pos = cls == 3 ? __popcll(h0 & below) : cls == 0 ? __popcll(h0) + __popcll(h1 & below) : __popcll(h0) + __popcll(h1) + __popcll(h2 & below);
    blk.order[(e0 / DPB) & 1][pos] = (unsigned char)lane;
  }
}

// Candidates of the nodes of block i0 .. i1: PART 1 -- those before block i0 - 64 (final when block i0 - 64 is being resolved); PART 2 -- those
// in block i0 - 64 (final once it is resolved) and, without their score, those inside the block itself (into the table).  A wavefront takes a
// node at a time from the dealer; its best goes to (pb, pk).
template <int FLAG, int PART>
__device__ __forceinline__ void dp_candidates(const Nodes &nd, DpRing &ring, DpBlock &blk, const uint32_t *ring_tot, uint32_t first, double st_wt,
                                              int i0, int cnt, int lane) {
  const int par = (i0 / DPB) & 1, i1 = i0 + cnt;
  const int j_split = i0 - DPB;                                  // PART 1: j < j_split; PART 2: j >= j_split
  const int ring_lo = dp_ring_lo(PART == 1 ? i0 - DPB : i0);     // (PART 1 runs while block i0 - 64 is resolved and block i0 + 64 enters)
  const DpSrc<FLAG> S{nd, ring, first, ring_lo, i1, i0};
  for (;;) {
    uint32_t u = 0;
    if (lane == 0) u = atomicAdd(PART == 1 ? &blk.next1 : &blk.next2, 1u);        // (named LDS objects, not pointers handed in: the address space stays known)
    u = (uint32_t)__builtin_amdgcn_readfirstlane((int)u);
    if (u >= (uint32_t)cnt) break;
    const int l = blk.order[par][u], i = i0 + l;
    const DpRec qi = ring.rec[dp_slot(i)];
    DpNode n2; n2.ndx = qi.ndx; n2.sv = qi.sv; n2.strand = (qi.pk & 2) ? -1 : 1; n2.stop = qi.pk & 1;
    const uint32_t lod = qi.tbl >> 16;
    const int lo = lod == (uint32_t)DP_FAR ? (int)nd.dp_min[first + (uint32_t)i] : i - (int)lod;
    const int le = blk.lo_eff[par][l];
    const int c2 = dp_class(n2.strand, n2.stop);
    if (PART == 2 && lane < l) blk.table[dp_tab(lane, l)] = -__builtin_inf();
    double best = -1.0; int bj = -1, bmark = -1;
    // class ranges: the first candidate must be in the node ring, every class's members behind it in the class rings (which grow by up
    // to two blocks while this runs: 128 entries of slack)
    bool by_class = le >= ring_lo;
    int a[4], n_lo[4], n_hi[4];
    if (by_class) {
      const ushort4 ca = ring.cnt[dp_slot(le)], cb = ring.cnt[dp_slot(i)];
      const int sp = j_split > le ? j_split : le;               // the first candidate of PART 2
      const ushort4 cs = ring.cnt[dp_slot(sp < i ? sp : i)];
      const unsigned short ua[4] = {ca.x, ca.y, ca.z, ca.w}, ub[4] = {cb.x, cb.y, cb.z, cb.w}, us[4] = {cs.x, cs.y, cs.z, cs.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        a[c] = ua[c];
        const int n_all = (unsigned short)(ub[c] - ua[c]), n_one = (unsigned short)(us[c] - ua[c]);
        n_lo[c] = PART == 1 ? 0 : n_one; n_hi[c] = PART == 1 ? n_one : n_all;
        if ((unsigned short)((unsigned short)DP_LDS_LD(&ring_tot[c]) - ua[c]) > DPC - 2 * DPB) by_class = false;
      }
    }
    if (by_class) {
      const DpRingSrc<FLAG> R{S, -1, DpRec{0, 0, 0, 0}};
#pragma unroll
      for (int c1 = 0; c1 < 4; ++c1) {
        if (!dp_pair_possible(c1, c2)) continue;
        const bool dyn = dp_pair_dynamic(c1, c2);
        for (int t = n_lo[c1] + lane; t < n_hi[c1]; t += 64) {
          const int jl = ring.cls[c1][(a[c1] + t) & (DPC - 1)];                           // low 16 bits of the index
          const int j = i - ((i - jl) & 0xffff);
          const bool inblk = j >= i0;
          if (inblk && dyn) continue;
          double sc; int mark; bool ok;
          if (c1 == 0) ok = dp_connection_s<DpRingSrc<FLAG>, true, 1, false>(R, st_wt, j, i, n2, sc, mark);
          else if (c1 == 1) ok = dp_connection_s<DpRingSrc<FLAG>, true, 1, true>(R, st_wt, j, i, n2, sc, mark);
          else if (c1 == 2) ok = dp_connection_s<DpRingSrc<FLAG>, true, -1, false>(R, st_wt, j, i, n2, sc, mark);
          else ok = dp_connection_s<DpRingSrc<FLAG>, true, -1, true>(R, st_wt, j, i, n2, sc, mark);
          if (ok) {
            if (inblk) blk.table[dp_tab(j - i0, l)] = sc;
            else dp_take(R.score(j) + sc, j, mark, best, bj, bmark);
          }
        }
      }
    } else {
      const int ja = PART == 1 ? lo : (j_split > lo ? j_split : lo), jb = PART == 1 ? (j_split < i ? j_split : i) : i;
      for (int j = ja + lane; j < jb; j += 64) {
        const bool inblk = j >= i0;
        const DpNode n1 = S.node(j);
        if (inblk && dp_pair_dynamic(dp_class(n1.strand, n1.stop), c2)) continue;
        double sc; int mark;
        if (dp_connection_s<DpSrc<FLAG>, false, 0, false>(S, st_wt, j, i, n2, sc, mark)) {
          if (inblk) blk.table[dp_tab(j - i0, l)] = sc;
          else dp_take(S.score(j) + sc, j, mark, best, bj, bmark);
        }
      }
    }
    int key = bj < 0 ? -1 : bj * 4 + (bmark + 1);                 // candidate index and overlap mark travel together
    dp_wave_best(best, key);
    if (lane == 0) {
      if (PART == 1) { blk.best1[par][l] = best; blk.key1[par][l] = key; }
      else { blk.best2[l] = best; blk.key2[l] = key; }
    }
  }
}

// (B): the nodes of block i0 .. i0 + cnt are resolved one after the other by ONE wavefront, a lane per node
template <int FLAG>
__device__ __forceinline__ void dp_resolve(const Nodes &nd, DpRing &ring, DpBlock &blk, uint32_t first, double st_wt, int i0, int cnt, int lane) {
  const int par = (i0 / DPB) & 1;
  const DpSrc<FLAG> S{nd, ring, first, dp_ring_lo(i0), i0 + cnt, 0x7fffffff};
  const bool in = lane < cnt;
  const int i = i0 + (in ? lane : 0);
  const int k = dp_slot(i);
  const DpRec qi = ring.rec[k];
  DpNode n2; n2.ndx = qi.ndx; n2.sv = qi.sv; n2.strand = (qi.pk & 2) ? -1 : 1; n2.stop = qi.pk & 1;
  const int c2 = dp_class(n2.strand, n2.stop);
  // The pairs left to this loop are forward stop t -> reverse node behind it (dp_pair_dynamic: they read WHERE t's predecessor lies).
  // score_connection's two cases for them, with everything that does not depend on t read BEFORE the chain of 64 dependent steps begins
  // -- the node's own value; a reverse stop's three overlapping starts with their positions and their value (+ intergenic term in the
  // final sweep) -- so that a step is registers and integer compares: through the general connection function a forward stop's step was
  // a dozen dependent LDS reads.  Same integer expressions and the same floating-point operations in the same order.
  // Measured with the cycle counter around the phases (48 bins): this wavefront's 64 steps ARE phase P (99 % of it; the scoring
  // wavefronts are busy for 75 % of P), a thousand cycles per step -- and neither the reads taken out of the chain here (62.3 -> 60 ms) nor
  // a raised wave priority (no change) moves that much: a step is 50 - 290 instructions of masked double-precision compares and selects,
  // issued one dependent instruction at a time.
  const double my_val = ring.sv2[k].y;
  int s_ok[3] = {0, 0, 0}, s_ndx[3] = {0, 0, 0}, s_sv[3] = {0, 0, 0}; double s_w[3] = {0.0, 0.0, 0.0};
  if (in && c2 == 3) {
#pragma unroll
    for (int f = 0; f < 3; ++f) {
      const int p3 = S.star(i, f);
      if (p3 == -1) continue;
      const DpNode n3 = S.node(p3);
      s_ok[f] = 1; s_ndx[f] = n3.ndx; s_sv[f] = n3.sv;
      s_w[f] = FLAG == 1 ? S.val(p3) + dp_igm(S, st_wt, p3, n3, i, n2) : S.val(p3);
    }
  }
  double best = in ? blk.best1[par][lane] : -1.0; int key0 = in ? blk.key1[par][lane] : -1;
  if (in) dp_merge(best, key0, blk.best2[lane], blk.key2[lane]);
  int bj = key0 < 0 ? -1 : key0 >> 2, bmark = key0 < 0 ? -1 : (key0 & 3) - 1;
  // Nothing a step needs may be a memory round trip away (under fifteen scoring wavefronts an LDS read comes back after hundreds of
  // cycles, and the steps are a chain): the position of every lane's current predecessor rides in a register beside its index, and the
  // table column of this lane is read three rows ahead.
  int bjx = in && bj >= 0 ? S.ndx(bj) : 0;
  double tab0 = blk.table[dp_tab(0, lane)], tab1 = blk.table[dp_tab(1, lane)], tab2 = blk.table[dp_tab(2, lane)];       // (rows of this block's table: written in the phase before)
#pragma unroll 1
  for (int t = 0; t < cnt; ++t) {
   {
    const double tab_t = tab0;
    tab0 = tab1; tab1 = tab2; tab2 = blk.table[dp_tab((t + 3) & (DPB - 1), lane)];       // three rows ahead (a rolled loop: unrolled eight times the 64 steps were 14 KB of code)
    // node i0 + t is final: every lane learns it, its own lane publishes it to the ring
    const double bt = dp_mk64(__builtin_amdgcn_readlane(dp_lo32(best), t), __builtin_amdgcn_readlane(dp_hi32(best), t));
    const int jt = __builtin_amdgcn_readlane(bj, t), c1 = __builtin_amdgcn_readlane(c2, t);
    const int t_ndx = __builtin_amdgcn_readlane(n2.ndx, t);
    const int t_tbx = __builtin_amdgcn_readlane(bjx, t);                 // position of t's predecessor (meaningful when jt >= 0)
    if (lane == t && bj >= 0) {
      ring.sv2[k].x = best;
      const uint32_t d = i - bj < DP_FAR ? (uint32_t)(i - bj) : (uint32_t)DP_FAR;
      ring.rec[k].tbl = (qi.tbl & 0xffff0000u) | d;
      ring.rec[k].pk = qi.pk | ((bmark + 1) << 4);
      if (d == (uint32_t)DP_FAR) GST(&nd.traceb[first + (uint32_t)i], bj);                  // (a trace-back the 16-bit distance cannot hold is read from global memory)
    }
    if (dp_class_needs_tb(c1) && jt < 0) continue;
    const double sc_t = jt >= 0 ? bt : 0.0;
    // One offer per step and lane, without a divergent branch (a masked `if` is a compare, a save of the execution mask, a branch and a
    // restore, each waiting for the one before): the table's connection -- or, where t is a forward stop and this lane a reverse node
    // (dp_pair_dynamic; the table holds -inf there, which no offer takes), the connection worked out here.  jt >= 0 in that case (a forward
    // stop without a predecessor was skipped above).
    double offer = sc_t + tab_t; int omark = -1; bool ok = in & (lane > t);
    if (c1 == 1) {                                                   // (uniform)
      const int n1x = t_ndx, tbx = t_tbx;                            // position of t, position of t's predecessor
      // forward stop -> reverse start: the genes may overlap by less than 200 bases
      const int ovlp2 = (n1x + 2) - (n2.sv - 2) + 1;
      const bool ok2 = !(n2.sv - 2 >= n1x + 2) & !(ovlp2 >= 200) & !((n1x + 2 - n2.sv - 2 + 1) >= (n2.ndx - n1x + 3 + 1)) & !((n1x + 2 - n2.sv - 2 + 1) >= (n2.sv - 3 - tbx + 1));
      const int left2 = n2.sv - 2, right2 = n2.ndx;
      const double score2 = FLAG == 0 ? ((double)(right2 - left2 + 1 - (ovlp2 * 2))) * my_val : my_val - 0.15 * st_wt;
      // forward stop -> reverse stop: through the best overlapping start, if one fits
      const int left = n1x + 2, right = n2.ndx - 2;
      double maxval = 0.0; int best_ov = 0, maxfr = -1;
#pragma unroll
      for (int f = 0; f < 3; ++f) {
        const int ov = left - s_sv[f] + 1;
        const bool fits = (s_ok[f] != 0) & !(ov <= 0 || ov >= 200) & !(ov >= s_ndx[f] - left) & !(ov >= s_sv[f] - tbx - 2) & (s_w[f] > maxval);
        maxfr = fits ? f : maxfr; maxval = fits ? s_w[f] : maxval; best_ov = fits ? ov : best_ov;
      }
      double score3, scr_mod = 0.0; int ovlp3 = 0;
      { double rval = 0.0; rval -= 0.15 * st_wt; score3 = FLAG == 1 ? rval : 0.0; }        // (no start fits: dp_igm of two strands -- that alone -- in the final sweep)
      if (FLAG == 0) scr_mod = maxfr != -1 ? maxval : 0.0; else score3 = maxfr != -1 ? maxval : score3;
      ovlp3 = maxfr != -1 ? best_ov : 0;
      if (FLAG == 0) score3 = ((double)(right - left + 1 - (ovlp3 * 2))) * scr_mod;
      const bool dyn = c2 >= 2;
      offer = dyn ? sc_t + (c2 == 2 ? score2 : score3) : offer;
      omark = (dyn & (c2 == 3)) ? maxfr : -1;
      ok = ok & (!dyn | (c2 == 2 ? ok2 : left < right));
    }
    {
      const int j = i0 + t;
      const bool take = ok & (offer >= 0.0) & ((bj < 0) | (offer > best) | ((offer == best) & (j > bj)));        // dp_take, every term evaluated
      best = take ? offer : best; bj = take ? j : bj; bmark = take ? omark : bmark; bjx = take ? t_ndx : bjx;
    }
   }
  }
}

// a block's results leave the ring for global memory (one wavefront)
__device__ __forceinline__ void dp_flush(const Nodes &nd, DpRing &ring, uint32_t first, int nn, int i0, int lane) {
  const int rel = i0 + lane;
  if (rel >= nn) return;
  const int k = dp_slot(rel); const DpRec q = ring.rec[k]; const uint32_t d = q.tbl & 0xffffu;
  if (d != 0) {
    GST(&nd.score[first + (uint32_t)rel], ring.sv2[k].x); GST(&nd.ov_mark[first + (uint32_t)rel], ((q.pk >> 4) & 3) - 1);
    if (d != (uint32_t)DP_FAR) GST(&nd.traceb[first + (uint32_t)rel], rel - (int)d);
  }
}

// Per block k of 64 nodes, two phases and two barriers:
//   P(k)  wavefront 0 resolves block k (B) | wavefront 1 enters block k + 2 into the rings | the others score the candidates of block
//         k + 1 that lie before block k (A1: nine tenths of a node's candidates -- all final)
//   Q(k)  all wavefronts: block k leaves for global memory; the candidates of block k + 1 inside block k (final now) and the table of
//         block k + 1 (A2)
template <int FLAG, int DP_NT>
__global__ void __launch_bounds__(DP_NT) gene_dp_kernel(Nodes nd, const uint32_t *__restrict__ seq_lo, const uint32_t *__restrict__ seq_n, const uint32_t *__restrict__ seq_bin,
                                                        const double *__restrict__ st_wt_of_bin, uint32_t nseq) {
  constexpr int DP_NW = DP_NT / 64;
  static_assert(DP_NW >= 4, "gene_dp_kernel: a resolver, an enterer and two scorers at least");
  __shared__ DpRing ring;
  __shared__ DpBlock blk;
  __shared__ uint32_t ring_tot[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (uint32_t s = blockIdx.x; s < nseq; s += gridDim.x) {
    const uint32_t first = seq_lo[s];
    const int nn = (int)seq_n[s];
    const double st_wt = st_wt_of_bin[seq_bin[s]];
    __syncthreads();
    if (tid < 4) ring_tot[tid] = 0;
    if (tid == 0) { blk.next1 = 0; blk.next2 = 0; }
    __syncthreads();
    if (nn == 0) continue;
    // prologue: blocks 0 and 1 enter; block 0 has no candidates before it (A1 empty), its table is (A2)
    if (wv == 1) dp_enter<FLAG>(nd, ring, blk, ring_tot, first, nn, 0, lane);
    if (wv == 0) { blk.best1[0][lane] = -1.0; blk.key1[0][lane] = -1; }
    __syncthreads();                   // (the class totals lane 0 stored are read by every lane of the next entering: a barrier between the two, not program order alone)
    if (wv == 1 && nn > DPB) dp_enter<FLAG>(nd, ring, blk, ring_tot, first, nn, DPB, lane);
    __syncthreads();
    dp_candidates<FLAG, 2>(nd, ring, blk, ring_tot, first, st_wt, 0, min(DPB, nn), lane);
    __syncthreads();
    for (int i0 = 0; i0 < nn; i0 += DPB) {
      const int cnt = min(DPB, nn - i0), i1 = i0 + cnt;
      const int cnt1 = i1 < nn ? min(DPB, nn - i1) : 0;                           // nodes of the next block
      // ---- P ----
      if (wv == 0) {
        if (lane == 0) blk.next2 = 0;
        dp_resolve<FLAG>(nd, ring, blk, first, st_wt, i0, cnt, lane);
      } else if (wv == 1) {
        if (i1 + DPB < nn) dp_enter<FLAG>(nd, ring, blk, ring_tot, first, nn, i1 + DPB, lane);
      }
      {          // (the resolver and the enterer join the scorers when they are done)
        if (cnt1) dp_candidates<FLAG, 1>(nd, ring, blk, ring_tot, first, st_wt, i1, cnt1, lane);
      }
      __syncthreads();
      // ---- Q ----
      if (wv == 2) dp_flush(nd, ring, first, nn, i0, lane);
      if (tid == 0) blk.next1 = 0;
      if (cnt1) dp_candidates<FLAG, 2>(nd, ring, blk, ring_tot, first, st_wt, i1, cnt1, lane);
      __syncthreads();
    }
  }
}
void x_dp(GExec &e, const Nodes &nd, const uint32_t *seq_lo, const uint32_t *seq_n, const uint32_t *seq_bin, const double *st_wt, uint32_t nseq, int flag) {
  if (!nseq) return;
  // threads per sequence: CKM_GENE_DP_THREADS = 256 | 512 | 1024 for measurements
  static const int nt = [] { const char *v = getenv("CKM_GENE_DP_THREADS"); const int n = v ? atoi(v) : 1024; return n == 256 || n == 512 ? n : 1024; }();
#define CKM_DP_LAUNCH(F, T) hipLaunchKernelGGL((gene_dp_kernel<F, T>), dim3(nseq), dim3(T), 0, e.st, nd, seq_lo, seq_n, seq_bin, st_wt, nseq)
  if (flag == 0) { if (nt == 256) CKM_DP_LAUNCH(0, 256); else if (nt == 512) CKM_DP_LAUNCH(0, 512); else CKM_DP_LAUNCH(0, 1024); }
  else { if (nt == 256) CKM_DP_LAUNCH(1, 256); else if (nt == 512) CKM_DP_LAUNCH(1, 512); else CKM_DP_LAUNCH(1, 1024); }
#undef CKM_DP_LAUNCH
}

}  // namespace gene
}  // namespace ckm
