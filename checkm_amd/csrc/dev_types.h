// dev_types.h -- plain structs shared by host orchestration and the gfx950 kernels.
#pragma once
#include <cstdint>

namespace ckm {

struct DevModel {
  int32_t M, ssvQ, fbQ, vitQH;
  int32_t fb_cls, vit_cls;  // index of fbQ / of the FAST Viterbi kernel's class among the instantiated classes (queues of the device-driven cascade)
  int32_t vitx_cls, vit16Q; // class of the EXACT Viterbi kernel (always the wave-per-pair kernel); packed registers per lane of the 16-lane kernel (0: none)
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
  float   thr_msv_f1_nat;   // smallest float v with (float)((double)v / ln 2) >= thr_msv_f1: the F1 test of the SSV epilogue, in nats
  // tables in HBM
  const int16_t *ssv_tbl;   // [Qg][30][16][8]  bias - cost as int16 (msv16_kernel, the exact MSV stage)
  const uint16_t *ssv_tbl_h; // same layout, (bias - cost)/256 as IEEE half bits (ssv_kernel_h)
  const uint16_t *ssv8_tbl_h; // [Qg8][30][2 copies][8 lanes][8] the 8-lane image of models of <= 512 nodes (ssv_kernel_h8), or null
  const uint8_t *rbv;       // [29][M+1]
  const uint32_t *vit_e;    // [30][vitQH][64] packed emission words (cell j | cell j+QH of each lane)
  const uint32_t *vit_t;    // [8][vitQH][64]  packed transition words: BM MM IM DM (into) MD MI II DD (from)
  const uint32_t *vit16_e;  // [30][vit16Q][16] the same words striped over 16 lanes (models of <= 512 nodes: four pairs per wavefront)
  const uint32_t *vit16_t;  // [8][vit16Q][16]
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

constexpr uint64_t FB_PATH_WITH_PP = 1ull << 63;      // FbWork::path_off: the posterior floats behind the path were asked for
struct FbWork {              // one Forward/Backward/OA work item (whole sequence, or one envelope)
  uint32_t model, seq;
  int32_t  i0, Ld, Lcfg, multihit;     // subsequence [i0, i0+Ld) of the target; length model configured for Lcfg
  uint64_t xs_off;                     // float offset: forward special rows (Ld+1)*6  [E N J B C scale]
  uint64_t aux_off;                    // float offset: parser mode -> decoding terms (Ld+1)*3 [bt et njcp];
                                       // full mode -> (Ld+1)*3 [ppN ppJ ppC], then 128B-aligned (Ld+1)*5 [oN oB oE oJ oC]
  uint64_t mxf_off, mxb_off;           // float offsets of the (Ld+1) x 3*Mp matrices (full == 1: q-major planes; full == 2: (Ld+1) x 4*Mp, cell-major float4 {M, I, D, 0})
  uint64_t path_off;                   // int32 offset + 1 of Mp entries: residue (1-based, within the envelope) emitted by each match state of the
                                       // OA path, 0 = node not matched (alignment requests); 0 = no path wanted.  With FB_PATH_WITH_PP set (the host
                                       // reserved them; the item's posterior rows then have a matrix of their own, mxb_off != mxf_off) Ld + 1 floats
                                       // follow: the posterior probability of each residue on the path
  uint32_t slot, full;                 // full: 0 parser (specials only), 1 matrix rows M,I, 2 matrix rows M,I,D (trace ensemble)
  uint32_t cand, pass;                 // device-driven cascade: candidate id of a parser item; id of its record in the pass table
};

// A queue of work-item indices consumed by a persistent kernel: the wavefronts of a launch share its entries by striding (wavefront w
// of W takes entries w, w + W, ...) up to min(*count, cap).  Host-built queues (alignment requests, diagnostics, second envelope
// rounds) and device-built ones (the cascade: only the device knows *count when the kernel is launched) look alike.
struct WorkQueue { const uint32_t *list; const uint32_t *count; uint32_t cap; };

constexpr int ENS_NSAMPLES = 200;     // stochastic tracebacks per multi-domain region (HMMER's default)

struct EnsWork {             // one multi-domain region handed to the trace-ensemble kernels (offsets in floats into the workspace)
  uint32_t model, seq;
  int32_t  i0, Ld, Lcfg, cap;          // residues [i0, i0+Ld) of the target; cap = segment slots per trace
  uint64_t xs_off, mx_off;             // multihit Forward of the region: special rows (Ld+1)*6, matrix (Ld+1) x Mp nodes of float4 {M, I, D, 0} (cell-major)
  uint64_t code_off;                   // uint16 [200][Ld+1] state codes per residue
  uint64_t ratio_off;                  // float  [200][Ld+1] null2 odds ratio per residue and trace
  uint64_t seg_off, nseg_off;          // int32  [200][cap][4] sampled segments (last domain first), int32 [200] counts (-1 = overflow)
  uint64_t n2_off;                     // float  [Ld] sum of the ratios over traces
  uint64_t host_off;                   // device-driven cascade: float offset of this region's exported results (counts, segments, sums) in the pinned result buffer
};

struct SsvEpi {               // by-value argument of the SSV launches: where the fused finish of the MSV stage puts its output
  const LenEntry *lentab;
  PairRec *survivors; uint32_t *nsurv; uint32_t cap_surv;      // pairs that pass F1 (or overflowed the byte score)
  PairRec *noresult;  uint32_t *nnores; uint32_t cap_nores;    // pairs SSV cannot decide (J state usable, or Smax = 0): exact MSV kernel
  uint16_t *maxv;                      // diagnostics only (ckm_debug_stages): Smax per pair is stored and NOTHING else happens; null in the product
};

// ---- device-driven cascade (ckm_cascade.hip, kernels_*.hip epilogues) ---------------------------------------------------------
constexpr int NV16 = 12;              // Viterbi filter, 16 lanes per pair: packed registers per lane Q16 in {1,2,3,4,5,6,7,8,10,12,14,16} (models of <= 512 nodes)
constexpr int NVW = 14;               // Viterbi filter, wavefront per pair: QH in {1,2,3,4,5,6,7,8,10,12,14,16, 24,32}   (24, 32: models of 2049..4096 nodes)
constexpr int NVC = NV16 + NVW;       // queue classes: the 16-lane classes first, then the wave classes
constexpr int NFC = 12;               // Forward/Backward register classes Q in {1,2,3,4,6,8,12,16,24,32, 48,64}

enum CascadeCounter : int {           // uint32 counters in device memory (count and head arrays share this layout)
  CC_CAND = 0, CC_NORES, CC_FWORK, CC_EWORK, CC_RWORK, CC_PASS, CC_REG, CC_EVENTS, CC_STATUS, CC_EVENTS_E, CC_EVENTS_R,
  CC_ENSQ = 11,                       // 11..14: multi-domain regions of sequence part 0, 1, 2, >= 3 (one trace-ensemble launch per part)
  CC_VQ = 16, CC_VXQ = CC_VQ + NVC, CC_FQ = CC_VXQ + NVC, CC_BQ = CC_FQ + NFC, CC_EQ = CC_BQ + NFC, CC_RQ = CC_EQ + NFC, CC_END = CC_RQ + NFC
};
constexpr int CC_SIZE = 128;
static_assert(CC_END <= CC_SIZE, "counter block too small");

enum CascadeStatus : uint32_t {       // bits of counter CC_STATUS: a table was too small for this search (the host retries with larger ones)
  CS_CAND = 1, CS_VQ = 2, CS_FWORK = 4, CS_EWORK = 8, CS_RWORK = 16, CS_PASS = 32, CS_REG = 64, CS_EVENTS = 128,
  CS_WS = 256                         // the float workspace ran out: the host runs this part of the search in workspace-sized batches instead
};

struct PassRec {                      // one pair the device let through F3 (pinned host memory): everything the host needs to take the
  uint32_t cand, fwork, model, seq;   // filter decisions again EXACTLY (libm logarithms) and to assemble the rows
  float    usc, bias_d, bias_e;       // MSV score; bias filter: d0+d1 and its power-of-two exponent
  float    vit_fast, vit_exact;       // Viterbi filter scores (nats); vit_exact only when route == 2
  uint32_t vit_flag, route;           // route: 0 Viterbi skipped (MSV P <= F2), 1 fast kernel only, 2 exact kernel ran
  float    fwd_xC; int32_t nscale;
};

constexpr uint32_t REGION_DEFERRED = 0xfffffffeu;

struct RegionRec {                    // one region found by the posterior heuristics (pinned host memory)
  uint32_t pass; int32_t i, j;        // residues i..j of the target (1-based)
  int32_t  multi;                     // 0: one envelope (target = envelope item), 1: trace ensemble (target = region item)
  uint32_t target;                    // index into the envelope / region work tables; REGION_DEFERRED: no workspace left on the device, the host
                                      // rescores it in workspace-sized batches; 0xffffffff: no table entry left
  uint32_t pad;                       // multi: float offset of the region's exported ensemble results in the pinned buffer (0xffffffff: none)
};

struct CascadeDev {                   // by-value kernel argument: where the epilogues of the filter / Forward / Backward kernels put their output
  PairRec *cand; uint32_t cap_cand;
  float *bias_raw;                    // [cap_cand][2]
  float *vit_fast, *vit_exact; uint32_t *vit_flag; uint8_t *route;      // [cap_cand]
  uint32_t *vq, *vxq; uint32_t cap_vq;                                  // [NVC][cap_vq] candidate ids
  uint32_t *fq, *bq, *eq, *rq; uint32_t cap_fq, cap_eq, cap_rq;         // [NFC][cap] work-item indices (bq shares cap_fq)
  uint32_t *cnt;                                                        // [CC_SIZE] counters of THIS group's chain: CC_CAND, CC_NORES and the queue lengths
  uint32_t *gcnt;                                                       // [CC_SIZE] counters shared by all groups of the lane: work tables, pass / region records, events, status
  FbWork *fwork; uint32_t cap_fwork;
  FbWork *ework; uint32_t cap_ework;
  FbWork *rwork; EnsWork *ens; uint32_t cap_rwork;
  uint32_t *ensq; uint32_t *ensq_cnt;                                  // [cap_rwork] region ids of this group's sequence PART, and their count (shared by the part's groups)
  unsigned long long *ws_top; unsigned long long ws_cap;                // bump allocator over the float workspace (units: floats): rows of the parser items
  unsigned long long *ws2_top; unsigned long long ws2_base, ws2_cap;    // second zone [ws2_base, ws2_base + ws2_cap): matrices of envelopes and ensemble regions;
                                                                        // a region that finds no room here is left to the host (RegionRec.target = REGION_DEFERRED)
  PassRec *h_pass; uint32_t cap_pass;
  RegionRec *h_reg; uint32_t cap_reg;
  unsigned long long *hens_top; unsigned long long hens_cap;            // bump allocator over the pinned buffer the ensemble results are exported to (floats)
  const int32_t *seq_len;
  float margin_msv, margin_vit, margin_fwd;                             // bits: widths of the conservative bands around F1/F2/F3
  uint32_t env_inplace;                                                 // envelope posterior rows overwrite the Forward rows (3 arrays per row instead of 5)
};

struct EnvOut {
  float   xC; int32_t nscale;
  float   oasc;
  int32_t hmm_from, hmm_to, ali_from, ali_to;
  int32_t range_err;
  float   null2[20];
};

}  // namespace ckm
