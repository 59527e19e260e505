// gene_types.h -- device-side tables of the gene-calling kernels (kernels_genes.hip) and their host builder (ckm_genes.hip).
#pragma once
#include <cstdint>

namespace ckm {

// sequences: one byte per base -- bits 0-1 the base (0 A, 1 C, 2 G, 3 T; an unknown base reads as C), bit 2 set for an unknown base.
// A "sequence" is a whole bin's training sequence (its contigs joined by TTAATTAATTAA) or one contig (a sub-range of the former).
struct GeneSeqDev {
  const uint8_t *txt;
  const uint64_t *off;        // [nseq] first byte
  const int32_t *len;         // [nseq]
};

// nodes, structure of arrays, in the gene finder's working order (position, forward strand first) sequence by sequence; every BIN's range
// is padded to a multiple of 256 entries with type = 255 so that a workgroup of the per-node kernels sees one bin's tables
struct GeneNodesDev {
  const uint32_t *bin, *seq;  // [n] bin (tables), sequence (coordinates)
  const int32_t *ndx, *stop_val;
  const int8_t *strand;       // +1 / -1
  const uint8_t *type;        // 0 ATG 1 GTG 2 TTG 3 STOP 255 padding
  const uint8_t *edge;
  double *cscore;             // out: gene_cscore_kernel (raw sum; the host applies the two sequential passes behind it)
  uint8_t *rbs0, *rbs1;       // out: gene_rbs_kernel
  // dynamic program
  const uint32_t *dp_min;     // [n] first candidate predecessor, relative to the sequence's first node
  const int32_t *star_ptr;    // [n][3] relative node index or -1
  const double *gcb;          // flag 0: bias . gc_score of the node
  const double *csc;          // flag 1: cscore + sscore
  const double *rscore, *uscore;
  double *score; int32_t *traceb, *ov_mark;      // in: 0 / -1 / -1; out (traceb absolute)
};

}  // namespace ckm
