// ckm_internal.h -- shared declarations of libcheckm_hip.so (not part of the ABI).
#pragma once
#include <cstdint>
#include <cstddef>
#include <string>
#include <vector>
#include <stdexcept>
#include "../../include/checkm_hip.h"

namespace ckm {

constexpr int K = 20;       // canonical residues
constexpr int KP = 29;      // HMMER amino alphabet "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~"
constexpr int NROWS = 30;   // emission rows held on the device: 29 symbols + 1 all-impossible pad row
constexpr int PADCODE = 29;
constexpr int NL = 64;      // lanes of the canonical float order (one wavefront)

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

void set_last_error(const std::string &m);

// One record of a HMMER3/f file, probabilities (not -ln p).
struct HostHMM {
  std::string name, acc, desc;
  bool has_acc = false, has_desc = false;
  int M = 0;
  std::vector<float> t, mat, ins;    // [(M+1)*7], [(M+1)*20], [(M+1)*20]
  float compo[K] = {0};
  bool has_compo = false;
  float evparam[6] = {0};
  int stats_mask = 0;
  double ga[2] = {0, 0}, tc[2] = {0, 0}, nc[2] = {0, 0};
  bool has_ga = false, has_tc = false, has_nc = false;
};

std::vector<HostHMM> read_hmm_file(const std::string &path);

// Search profile in the three score systems, host copy.
struct HostProfile {
  int M = 0;
  // MSV / SSV (unsigned byte costs, 1/3 bit)
  int base_b = 190, bias_b = 0, tbm_b = 0, tec_b = 0;
  float scale_b = 0;
  std::vector<uint8_t> rbv;       // [KP][M+1]
  int ssvQ = 0;                   // packed i16x2 registers per lane (16 lanes per sequence): ceil(M/32) rounded to an instantiated size
  std::vector<uint16_t> ssv_tbl_h; // the same image as IEEE half bits of (bias - cost)/256
  int ssv8Q = 0;                   // 8 lanes per sequence (models of <= 512 nodes): packed registers per lane, ceil(M/16) rounded up to an instance; 0 = none
  std::vector<uint16_t> ssv8_tbl_h; // LDS image [ssv8Qg][NROWS][2 copies][8 lanes][4 regs][2 halves]
  std::vector<int16_t> ssv_tbl;   // LDS image: [NROWS][ssvQg][16 lanes][4 regs][2 halves]
  // Viterbi filter (signed words, 1/500 bit); contiguous k, padded to vitQ*64
  int vitQH = 0;                  // packed registers per lane: lane z owns cells z*2QH.., register j = (cell j, cell j+QH)
  float scale_w = 0; int base_w = 12000; int wE_loop = 0, wE_move = 0;
  int vit16Q = 0;                 // 16-lane striping (four pairs per wavefront): packed registers per lane, 0 for models beyond 512 nodes
  std::vector<uint32_t> vit16_e, vit16_t;   // [NROWS][vit16Q][16], [8][vit16Q][16]
  std::vector<uint32_t> vit_e;    // [NROWS][vitQH][64]
  std::vector<uint32_t> vit_t;    // [8][vitQH][64]: BM MM IM DM (into k) MD MI II DD (from k)
  // Forward/Backward odds, canonical padded layout (fbQ*64)
  int fbQ = 0;
  std::vector<float> rf;          // [NROWS][Mp]
  std::vector<float> ftr;         // [8][Mp]: BM MM IM DM MI II MD DD
  float fE_loop = 0.5f, fE_move = 0.5f;
  // bias filter
  float bt00, bt01, bt10, bt11, bpi0, bpi1;
  float beo1[NROWS];
  // score-space thresholds equivalent to P<=F1 / P<=F2 / P<=F3 (smallest passing float)
  float thr_msv_f1, thr_msv_f2, thr_vit_f2, thr_fwd_f3;
  float thr_msv_f1_nat;     // smallest float v with (float)((double)v / ln 2) >= thr_msv_f1 (the SSV epilogue tests usc - nullsc against it)
};

HostProfile configure_profile(const HostHMM &h);
int  canon_Q(int M);         // canonical lanes-blocked Q for the float DP
int  ssv_Q_for(int M);       // instantiated SSV register count covering M
int  vit_QH_for(int M);      // instantiated Viterbi packed-register count covering M

// length-dependent specials (all host libm, shared by every stage)
struct LenCfg { float loop, move; int w_move; int tjb_b; float nullsc; float p1; float bias_tail; };
LenCfg len_config(const HostProfile &p, int L, bool multihit);

void digitize(const char *text, uint64_t n, uint8_t *dsq);

// One FASTA file, digitised: records padded to 16 bytes with PADCODE, offsets relative to the file's own buffer.
struct FastaBin {
  std::vector<std::string> names, descs;
  std::vector<int32_t> len;
  std::vector<uint64_t> off;
  std::vector<uint8_t> dsq;
  uint64_t total_res = 0;
  int maxL = 0;
  int err_code = 0; std::string err;      // err_code != 0: the file could not be read
};
std::vector<FastaBin> read_fasta_bins(const char *const *paths, uint32_t nbins, int nthreads);
// The host columns of a batch of sequences (ckm_host.h: ckm_seqs), by address, so that fasta_ingest.cpp stays free of device types
struct SeqColumns {
  std::vector<uint32_t> *bin_off, *seq_bin, *order, *order_off;
  std::vector<int32_t> *len;
  std::vector<uint64_t> *off, *bin_res;
  std::vector<uint8_t> *dsq;
  std::vector<std::string> *names, *descs;
  uint64_t *total_res; int *maxL;
};
void merge_fasta_bins(std::vector<FastaBin> &bins, int nthreads, const SeqColumns &o);    // fills bin_off, len, off, dsq, names, descs; adds to total_res / maxL
void build_seq_order(int nthreads, const SeqColumns &o);                                  // from bin_off + len: seq_bin, order, order_off, bin_res
int ingest_threads();                                                                     // host threads one ingest may use

// statistics
double gumbel_surv(double x, double mu, double lambda);
double exp_surv(double x, double mu, double lambda);
double exp_logsurv(double x, double mu, double lambda);
float  flogsum(float a, float b);

}  // namespace ckm
