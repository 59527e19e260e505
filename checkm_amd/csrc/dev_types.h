// dev_types.h -- plain structs shared by host orchestration and the gfx950 kernels.
#pragma once
#include <cstdint>

namespace ckm {

struct DevModel {
  int32_t M, ssvQ, fbQ, vitQH;
  // MSV
  int32_t base_b, bias_b, tbm_b, tec_b;
  float   scale_b;
  // Viterbi filter
  float   scale_w; int32_t base_w, wE_loop, wE_move;
  // Forward
  float   fE_loop, fE_move;
  // bias filter
  float   bt00, bt01, bt10, bt11, bpi0, bpi1;
  float   beo1[30];
  // filter thresholds (score space)
  float   thr_msv_f1, thr_msv_f2, thr_vit_f2, thr_fwd_f3;
  // tables in HBM
  const int16_t *ssv_tbl;   // [Qg][30][16][8]
  const uint8_t *rbv;       // [29][M+1]
  const uint32_t *vit_e;    // [30][vitQH][64] packed emission words (cell j | cell j+QH of each lane)
  const uint32_t *vit_t;    // [8][vitQH][64]  packed transition words: BM MM IM DM (into) MD MI II DD (from)
  const float   *rf;        // [30][Mp]
  const float   *ftr;       // [8][Mp]
};

// length-dependent specials, indexed by sequence length L
struct LenEntry {
  float   loop_m, move_m;   // multihit N/C/J loop, move
  float   loop_u, move_u;   // unihit
  float   nullsc;           // null1
  float   bias_tail;        // L*logf(p1) + logf(1-p1)
  int32_t w_move;           // Viterbi-filter N/C/J move word (multihit)
  int32_t tjb_b;            // MSV tjb byte cost
};

struct SsvBlockWork {        // one workgroup of the SSV kernel
  uint32_t model;
  uint32_t list_start;       // first entry of lists[] (sequence ids, length-sorted)
  uint32_t count;            // sequences in this block
  uint32_t pair_start;       // first output slot
};

struct PairRec {             // a (model, sequence) pair travelling down the cascade
  uint32_t model, seq;
  float    usc;              // MSV score, nats (+inf on overflow)
  float    filtersc;         // bias-filter null score
};

struct ScaleEvent {          // one Forward rescale: pair slot, row, factor
  uint32_t slot; int32_t row; float scale; uint32_t pad;
};

struct FwdOut {              // per pair slot
  float xC; int32_t nscale;
};

struct FbWork {              // one Forward/Backward/OA work item (whole sequence, or one envelope)
  uint32_t model, seq;
  int32_t  i0, Ld, Lcfg, multihit;     // subsequence [i0, i0+Ld) of the target; length model configured for Lcfg
  uint64_t xs_off;                     // float offset: forward special rows (Ld+1)*6  [E N J B C scale]
  uint64_t aux_off;                    // float offset: parser mode -> decoding terms (Ld+1)*3 [bt et njcp];
                                       // full mode -> (Ld+1)*3 [ppN ppJ ppC], then 128B-aligned (Ld+1)*5 [oN oB oE oJ oC]
  uint64_t mxf_off, mxb_off;           // float offsets of the (Ld+1) x 3*Mp matrices (full mode only)
  uint64_t path_off;                   // int32 offset + 1 of Mp entries: residue (1-based, within the envelope) emitted by each match state of the
                                       // OA path, 0 = node not matched (alignment requests); 0 = no path wanted
  uint32_t slot, full;                 // full: 0 parser (specials only), 1 matrix rows M,I, 2 matrix rows M,I,D (trace ensemble)
};

constexpr int ENS_NSAMPLES = 200;     // stochastic tracebacks per multi-domain region (HMMER's default)

struct EnsWork {             // one multi-domain region handed to the trace-ensemble kernels (offsets in floats into the workspace)
  uint32_t model, seq;
  int32_t  i0, Ld, Lcfg, cap;          // residues [i0, i0+Ld) of the target; cap = segment slots per trace
  uint64_t xs_off, mx_off;             // multihit Forward of the region: special rows (Ld+1)*6, matrix (Ld+1) x 3*Mp (M I D)
  uint64_t code_off;                   // uint16 [200][Ld+1] state codes per residue
  uint64_t ratio_off;                  // float  [200][Ld+1] null2 odds ratio per residue and trace
  uint64_t seg_off, nseg_off;          // int32  [200][cap][4] sampled segments (last domain first), int32 [200] counts (-1 = overflow)
  uint64_t n2_off;                     // float  [Ld] sum of the ratios over traces
};

struct FinishArgs {
  const DevModel *models; const LenEntry *lentab; const int32_t *seq_len; const uint32_t *lists;
  const SsvBlockWork *work;            // the table the SSV launches used (one entry per SSV block)
  const uint16_t *maxv;                // Smax per pair (0 = degenerate: recompute exactly)
  PairRec *survivors; uint32_t *nsurv; uint32_t cap_surv;
  PairRec *noresult;  uint32_t *nnores; uint32_t cap_nores;
};

struct EnvOut {
  float   xC; int32_t nscale;
  float   oasc;
  int32_t hmm_from, hmm_to, ali_from, ali_to;
  int32_t range_err;
  float   null2[20];
};

}  // namespace ckm
