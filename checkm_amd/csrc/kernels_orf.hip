// kernels_orf.hip -- first slice of gene calling on the device: the deterministic FRONT END of the gene finder CheckM
// runs before the marker-gene scan (`prodigal -p single -m -g 11|4`, checkm/prodigal.py:74,86-93,131-133): start / stop codon flags
// of all six frames, and the start / stop NODES prodigal's dynamic program works on (node.c: add_nodes).  gfx950 only.
//
//   orf_flags_kernel   one thread per 64 bases: reads the nucleotide text once (1 B / base), writes one flag byte per base
//                      (1 B / base) -- a pure streaming kernel, bound by HBM: 2 algorithmic bytes per base.
//                        bit 0     forward codon at i is a stop of table 11 (TAA TAG TGA)      bit 1  ... of table 4 (TAA TAG)
//                        bits 2-3  forward codon at i is a start: 1 ATG, 2 GTG, 3 TTG
//                        bits 4-7  the same for the reverse-strand codon whose first base is the complement of base i
//                      Contigs are laid out with >= 2 separator bytes ('N') between them, so no codon spans two contigs.
//   orf_chain_kernel   one wavefront per (contig, strand, frame): the frame's codons from the 3' end to the 5' end, 64 per step; the
//                      sequential registers of add_nodes (last stop, start-seen, minimum length) become wave-wide prefix operations
//                      on ballots (nearest stop / any start among the lanes scanned before me) plus three carried scalars.
// The oracle (oracle/gene_oracle.c) states what is and is not restated of prodigal, and that none of it is pinned to a real prodigal.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "dev_types.h"

namespace ckm {

constexpr int ORF_MIN_GENE = 90, ORF_MIN_EDGE_GENE = 60;

__device__ __forceinline__ uint32_t nt_code(uint32_t ch) {        // ASCII -> 0 A, 1 C, 2 G, 3 T/U, 4 other
  ch &= 0xDFu;                                                    // upper case
  return ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : (ch == 'T' || ch == 'U') ? 3u : 4u;
}

// flags of a forward codon (b0 b1 b2) -- low nibble layout described above
__device__ __forceinline__ uint32_t codon_flags(uint32_t b0, uint32_t b1, uint32_t b2) {
  if ((b0 | b1 | b2) > 3u) return 0u;
  const uint32_t c = b0 * 16u + b1 * 4u + b2;
  uint32_t f = 0u;
  if (c == 48u || c == 50u) f |= 3u;            // TAA TAG: stop in both tables
  if (c == 56u) f |= 1u;                        // TGA: stop in table 11 only
  if (c == 14u) f |= 1u << 2;                   // ATG
  if (c == 46u) f |= 2u << 2;                   // GTG
  if (c == 62u) f |= 3u << 2;                   // TTG
  return f;
}

__global__ void __launch_bounds__(256) orf_flags_kernel(const uint8_t *__restrict__ text, uint8_t *__restrict__ flags, uint64_t n) {
  const uint64_t w0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 64ull;
  if (w0 >= n) return;
  // 64 bases + 2 of halo on either side, as codes in registers (the buffer is padded by 16 bytes of 'N' at both ends)
  uint32_t code[68];
  const uint4 *src = reinterpret_cast<const uint4 *>(text + w0);        // w0 is a multiple of 64 and the buffer 16-byte aligned
  const uint32_t hl = *reinterpret_cast<const uint32_t *>(text + w0 - 4), hr = *reinterpret_cast<const uint32_t *>(text + w0 + 64);
  code[0] = nt_code((hl >> 16) & 0xff); code[1] = nt_code(hl >> 24);
  code[66] = nt_code(hr & 0xff); code[67] = nt_code((hr >> 8) & 0xff);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint4 v = src[q];
    const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int b = 0; b < 4; ++b) code[2 + q * 16 + k * 4 + b] = nt_code((wd[k] >> (8 * b)) & 0xff);
  }
  uint32_t outw[16];
#pragma unroll
  for (int j = 0; j < 64; ++j) {
    const uint32_t c0 = code[2 + j], f1 = code[3 + j], f2 = code[4 + j], r1 = code[1 + j], r2 = code[j];
    const uint32_t fw = codon_flags(c0, f1, f2);
    // reverse-strand codon starting at i: complements of bases i, i-1, i-2 (complement of code c <= 3 is 3 - c)
    const uint32_t rv = ((c0 | r1 | r2) > 3u) ? 0u : codon_flags(3u - c0, 3u - r1, 3u - r2);
    const uint32_t fb = fw | (rv << 4);
    if ((j & 3) == 0) outw[j >> 2] = fb; else outw[j >> 2] |= fb << (8 * (j & 3));
  }
  uint4 *dst = reinterpret_cast<uint4 *>(flags + w0);
#pragma unroll
  for (int q = 0; q < 4; ++q) dst[q] = make_uint4(outw[q * 4], outw[q * 4 + 1], outw[q * 4 + 2], outw[q * 4 + 3]);
}

struct OrfNode { uint32_t contig; int32_t ndx, stop_val; uint8_t type, strand_rev, edge, pad; };     // type 0 ATG, 1 GTG, 2 TTG, 3 stop

// One wavefront per (contig, strand, frame).  Scan coordinate j = position on the strand being read (forward: j = i; reverse: j is the
// index into the reverse complement, forward position slen-1-j), descending from the last complete codon of the frame.
__global__ void __launch_bounds__(64) orf_chain_kernel(const uint8_t *__restrict__ flags, const uint64_t *__restrict__ contig_off /* start of each contig in the padded buffer */,
                                                        const int32_t *__restrict__ contig_len, uint32_t ncontigs, int tt4, int closed,
                                                        OrfNode *__restrict__ nodes, unsigned long long *__restrict__ nnodes, unsigned long long cap) {
  const uint32_t job = blockIdx.x;
  const uint32_t ci = job / 6u, sub = job % 6u;
  if (ci >= ncontigs) return;
  const int rev = sub >= 3u, frame = (int)(sub % 3u);
  const int slen = contig_len[ci];
  if (slen < 3) return;
  const uint8_t *fl = flags + contig_off[ci];
  const int lane = threadIdx.x;
  const int shift = rev ? 4 : 0;
  const uint32_t stopbit = tt4 ? 2u : 1u;
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
    uint32_t f = 0;
    if (j >= 0) f = ((uint32_t)fl[rev ? slen - 1 - j : j] >> shift) & 0xfu;
    const bool is_stop = j >= 0 && (f & stopbit);
    const int st = (int)((f >> 2) & 3u) - 1;                      // -1 none, 0 ATG, 1 GTG, 2 TTG
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

void launch_orf_flags(hipStream_t stream, const uint8_t *text, uint8_t *flags, uint64_t n) {
  const uint64_t threads = (n + 63) / 64;
  if (threads) hipLaunchKernelGGL(orf_flags_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, text, flags, n);
}
void launch_orf_chain(hipStream_t stream, const uint8_t *flags, const uint64_t *contig_off, const int32_t *contig_len, uint32_t ncontigs, int tt4, int closed,
                      void *nodes, unsigned long long *nnodes, unsigned long long cap) {
  if (ncontigs) hipLaunchKernelGGL(orf_chain_kernel, dim3(ncontigs * 6), dim3(64), 0, stream, flags, contig_off, contig_len, ncontigs, tt4, closed,
                                   reinterpret_cast<OrfNode *>(nodes), nnodes, cap);
}

}  // namespace ckm
