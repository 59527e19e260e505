// kernels_orf.hip -- first slice of gene calling on the device: the deterministic FRONT END of the gene finder CheckM
// runs before the marker-gene scan (`prodigal -p single -m -g 11|4`, checkm/prodigal.py:74,86-93,131-133): start / stop codon flags
// of all six frames, and the start / stop NODES prodigal's dynamic program works on (node.c: add_nodes).  gfx950 only.
//
//   orf_flags_kernel   one thread per 64 bases, bit-parallel: reads the nucleotide text once (1 B / base), writes eight bit planes
//                      (stop of table 11 / of table 4 / start type, both strands: 1 B / base in all) -- a streaming kernel priced
//                      against the HBM roofline: 2 algorithmic bytes per base.  Contigs are laid out with >= 2 separator bytes ('N')
//                      between them, so no codon spans two contigs.
//   orf_chain_kernel   one wavefront per (contig, strand, frame): the frame's codons from the 3' end to the 5' end, 64 per step; the
//                      sequential registers of add_nodes (last stop, start-seen, minimum length) become wave-wide prefix operations
//                      on ballots (nearest stop / any start among the lanes scanned before me) plus three carried scalars.
// The oracle (oracle/gene_oracle.c) states what is and is not restated of prodigal, and that none of it is pinned to a real prodigal.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "dev_types.h"

namespace ckm {

constexpr int ORF_MIN_GENE = 90, ORF_MIN_EDGE_GENE = 60;

// ---- the streaming kernel: 64 bases per thread, bit-parallel ------------------------------------------------------------------------
// The 64 bases of a window become four 64-bit masks (which positions hold A, C, G, T/U; anything else is in none of them), built four
// bytes at a time with exact byte-equality tests on 32-bit words; every codon test of the window is then a handful of 64-bit AND / OR
// operations on shifted masks (bit k = base k of the window, two halo bases on either side).  Output: eight bit planes of one 64-bit
// word per window -- 1 byte per base in all, as many bytes as were read.
//   plane 0  forward codon at i is a stop of table 11 (TAA TAG TGA)     plane 1  ... of table 4 (TAA TAG)
//   plane 2 / 3  low / high bit of the forward start type (1 ATG, 2 GTG, 3 TTG; 0 none)
//   plane 4-7  the same for the reverse-strand codon whose first base is the complement of base i (its bases are i, i-1, i-2)
// (8 planes of n / 64 words each)

__device__ __forceinline__ uint32_t bytes_equal(uint32_t w, uint32_t letter4) {     // 0x80 in every byte of w that equals the letter
  const uint32_t z = w ^ letter4;
  return ~(((z & 0x7f7f7f7fu) + 0x7f7f7f7fu) | z | 0x7f7f7f7fu);
}
__device__ __forceinline__ uint32_t gather4(uint32_t m) {          // bits 7, 15, 23, 31 -> bits 0..3
  m >>= 7;
  return (m | (m >> 7) | (m >> 14) | (m >> 21)) & 0xfu;
}
struct Masks4 { uint32_t a, c, g, t; };
__device__ __forceinline__ Masks4 word_masks(uint32_t w) {
  const uint32_t lc = w | 0x20202020u;                             // lower case
  Masks4 m;
  m.a = gather4(bytes_equal(lc, 0x61616161u)); m.c = gather4(bytes_equal(lc, 0x63636363u)); m.g = gather4(bytes_equal(lc, 0x67676767u));
  m.t = gather4(bytes_equal(lc & 0xfefefefeu, 0x74747474u));       // 't' and 'u' differ in bit 0 only
  return m;
}

__global__ void __launch_bounds__(256) orf_flags_kernel(const uint8_t *__restrict__ text, unsigned long long *__restrict__ planes, uint64_t nwin) {
  const uint64_t win = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (win >= nwin) return;
  const uint8_t *src = text + win * 64ull;                         // 64-byte aligned; the buffer has >= 64 bytes of 'N' before and after
  unsigned long long A = 0, C = 0, G = 0, T = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint4 v = reinterpret_cast<const uint4 *>(src)[q];
    const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const Masks4 m = word_masks(wd[k]);
      const int sh = (q * 4 + k) * 4;
      A |= (unsigned long long)m.a << sh; C |= (unsigned long long)m.c << sh; G |= (unsigned long long)m.g << sh; T |= (unsigned long long)m.t << sh;
    }
  }
  // halo: bases -4..-1 (bit 3 = base -1) and 64..67 (bit 0 = base 64)
  const Masks4 hp = word_masks(*reinterpret_cast<const uint32_t *>(src - 4)), hn = word_masks(*reinterpret_cast<const uint32_t *>(src + 64));
#define NEXT1(X, h) (((X) >> 1) | ((unsigned long long)((h) & 1u) << 63))
#define NEXT2(X, h) (((X) >> 2) | ((unsigned long long)((h) & 3u) << 62))
#define PREV1(X, h) (((X) << 1) | (unsigned long long)(((h) >> 3) & 1u))
#define PREV2(X, h) (((X) << 2) | (unsigned long long)(((h) >> 2) & 3u))
  const unsigned long long A1 = NEXT1(A, hn.a), A2 = NEXT2(A, hn.a), G1 = NEXT1(G, hn.g), G2 = NEXT2(G, hn.g), T1 = NEXT1(T, hn.t);
  const unsigned long long Tm1 = PREV1(T, hp.t), Tm2 = PREV2(T, hp.t), Cm1 = PREV1(C, hp.c), Cm2 = PREV2(C, hp.c), Am1 = PREV1(A, hp.a);
#undef NEXT1
#undef NEXT2
#undef PREV1
#undef PREV2
  const unsigned long long TA = T & A1, T1G2 = T1 & G2;
  const unsigned long long fTAAG = TA & (A2 | G2), fTGA = T & G1 & A2, fATG = A & T1G2, fGTG = G & T1G2, fTTG = T & T1G2;
  // reverse strand: the codon read from base i leftwards, complemented (A <-> T, C <-> G): TAA = A T T, TAG = A T C, TGA = A C T,
  // ATG = T A C, GTG = C A C, TTG = A A C
  const unsigned long long ATm = A & Tm1, Am1Cm2 = Am1 & Cm2;
  const unsigned long long rTAAG = ATm & (Tm2 | Cm2), rTGA = A & Cm1 & Tm2, rATG = T & Am1Cm2, rGTG = C & Am1Cm2, rTTG = A & Am1Cm2;
  planes[0 * nwin + win] = fTAAG | fTGA; planes[1 * nwin + win] = fTAAG; planes[2 * nwin + win] = fATG | fTTG; planes[3 * nwin + win] = fGTG | fTTG;
  planes[4 * nwin + win] = rTAAG | rTGA; planes[5 * nwin + win] = rTAAG; planes[6 * nwin + win] = rATG | rTTG; planes[7 * nwin + win] = rGTG | rTTG;
}

struct OrfNode { uint32_t contig; int32_t ndx, stop_val; uint8_t type, strand_rev, edge, pad; };     // type 0 ATG, 1 GTG, 2 TTG, 3 stop

// One wavefront per (contig, strand, frame).  Scan coordinate j = position on the strand being read (forward: j = i; reverse: j is the
// index into the reverse complement, forward position slen-1-j), descending from the last complete codon of the frame.
__global__ void __launch_bounds__(64) orf_chain_kernel(const unsigned long long *__restrict__ planes, uint64_t nwin, const uint64_t *__restrict__ contig_off /* start of each contig in the padded buffer */,
                                                        const int32_t *__restrict__ contig_len, uint32_t ncontigs, int tt4, int closed,
                                                        OrfNode *__restrict__ nodes, unsigned long long *__restrict__ nnodes, unsigned long long cap) {
  const uint32_t job = blockIdx.x;
  const uint32_t ci = job / 6u, sub = job % 6u;
  if (ci >= ncontigs) return;
  const int rev = sub >= 3u, frame = (int)(sub % 3u);
  const int slen = contig_len[ci];
  if (slen < 3) return;
  const uint64_t base = contig_off[ci];                          // position of the contig's first base in the padded buffer
  const int lane = threadIdx.x;
  const unsigned long long *p_stop = planes + (uint64_t)((rev ? 4 : 0) + (tt4 ? 1 : 0)) * nwin;
  const unsigned long long *p_lo = planes + (uint64_t)((rev ? 4 : 0) + 2) * nwin, *p_hi = planes + (uint64_t)((rev ? 4 : 0) + 3) * nwin;
  // the frame's first scanned position: the largest j <= slen-3 with j % 3 == frame
  int jtop = slen - 3; jtop -= ((jtop % 3) - frame + 3) % 3;
  auto emit = [&](int ndx_s, int type, int sv_s, int edge) {
    const unsigned long long k = atomicAdd(nnodes, 1ull);
    if (k < cap) {
      OrfNode nd; nd.contig = ci; nd.type = (uint8_t)type; nd.strand_rev = (uint8_t)rev; nd.edge = (uint8_t)edge; nd.pad = 0;
      nd.ndx = rev ? slen - 1 - ndx_s : ndx_s; nd.stop_val = rev ? slen - 1 - sv_s : sv_s;
      nodes[k] = nd;
    }
  };
  // carried registers of add_nodes for this frame (uniform across the wave)
  int last = closed ? slen + ((frame - slen % 3 + 3) % 3) : jtop;     // closed ends: the virtual stop sits beyond the sequence (last >= slen: nothing starts before a real stop)
  bool last_real = false, saw = false, any_stop = false;
  for (int jhi = jtop; jhi >= 0; jhi -= 192) {
    const int j = jhi - 3 * lane;
    bool is_stop = false; int st = -1;                           // st: -1 none, 0 ATG, 1 GTG, 2 TTG
    if (j >= 0) {
      const uint64_t pos = base + (uint64_t)(rev ? slen - 1 - j : j);
      const uint64_t wi = pos >> 6; const int bit = (int)(pos & 63);
      is_stop = (p_stop[wi] >> bit) & 1ull;
      st = (int)(((p_lo[wi] >> bit) & 1ull) | (((p_hi[wi] >> bit) & 1ull) << 1)) - 1;
    }
    const unsigned long long stops = __ballot(is_stop);
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));          // lanes scanned before me
    const unsigned long long sb = stops & below;
    // nearest stop scanned before me: in this chunk (highest lane below mine) or the carried one
    const int qlane = sb ? 63 - __clzll((long long)sb) : -1;
    const int my_last = qlane >= 0 ? jhi - 3 * qlane : last;
    const bool my_last_real = qlane >= 0 ? true : last_real;
    const bool my_any_stop = any_stop || sb != 0ull;
    const int mind = my_any_stop ? ORF_MIN_GENE : ORF_MIN_EDGE_GENE;
    bool start_node = false, edge_node = false;
    if (j >= 0 && !is_stop && my_last < slen) {
      if (st >= 0 && my_last - j + 3 >= mind) start_node = true;
      else if (j <= 2 && !closed && (my_last - j) > ORF_MIN_EDGE_GENE) edge_node = true;
    }
    const unsigned long long starts = __ballot(start_node || edge_node);
    if (start_node) emit(j, st, my_last, 0);
    if (edge_node) emit(j, 0, my_last, 1);
    if (is_stop) {
      // was a start recorded between the previous stop (or the chunk's beginning, with the carry) and me?
      const unsigned long long between = starts & below & (qlane >= 0 ? ~(~0ull >> (63 - qlane)) : ~0ull);
      const bool my_saw = (between != 0ull) || (qlane < 0 && saw);
      if (my_saw) emit(my_last, 3, j, my_last_real ? 0 : 1);
    }
    // carry to the next chunk: state after the chunk's last lane
    if (stops) {
      const int ql = 63 - __clzll((long long)stops);
      last = jhi - 3 * ql; last_real = true; any_stop = true;
      saw = (starts & (ql == 63 ? 0ull : (~0ull << (ql + 1)))) != 0ull;
    } else saw = saw || starts != 0ull;
  }
  if (saw && lane == 0) emit(last, 3, frame - 6, last_real ? 0 : 1);
}

// pseudo-random nucleotides for the streaming measurement of ckm_debug_orf_flags (no host buffer of that size has to exist)
__global__ void orf_fill_kernel(uint8_t *text, uint64_t n, uint32_t seed) {
  const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4ull;
  if (i >= n) return;
  uint32_t x = (uint32_t)(i >> 2) * 2654435761u + seed;
  x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
  const uint32_t lut = 0x54474341u;      // "ACGT"
  uint32_t w = 0;
  for (int b = 0; b < 4; ++b) w |= ((lut >> (8 * ((x >> (2 * b)) & 3u))) & 0xffu) << (8 * b);
  *reinterpret_cast<uint32_t *>(text + i) = w;
}
void launch_orf_fill(hipStream_t stream, uint8_t *text, uint64_t n, uint32_t seed) {
  const uint64_t threads = (n + 3) / 4;
  hipLaunchKernelGGL(orf_fill_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, text, n, seed);
}

// text: n bytes (a multiple of 64) with >= 64 readable bytes before and after; planes: 8 x (n / 64) 64-bit words
void launch_orf_flags(hipStream_t stream, const uint8_t *text, uint8_t *planes, uint64_t n) {
  const uint64_t nwin = n / 64;
  if (nwin) hipLaunchKernelGGL(orf_flags_kernel, dim3((unsigned)((nwin + 255) / 256)), dim3(256), 0, stream, text, reinterpret_cast<unsigned long long *>(planes), nwin);
}
void launch_orf_chain(hipStream_t stream, const uint8_t *planes, uint64_t nwin, const uint64_t *contig_off, const int32_t *contig_len, uint32_t ncontigs, int tt4, int closed,
                      void *nodes, unsigned long long *nnodes, unsigned long long cap) {
  if (ncontigs) hipLaunchKernelGGL(orf_chain_kernel, dim3(ncontigs * 6), dim3(64), 0, stream, reinterpret_cast<const unsigned long long *>(planes), nwin, contig_off, contig_len, ncontigs, tt4, closed,
                                   reinterpret_cast<OrfNode *>(nodes), nnodes, cap);
}

}  // namespace ckm
