/* p7oracle.h -- CPU oracle for the CheckM marker-gene scan half (hmmsearch pipeline).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under checkm_amd/ may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY UNPINNED (scan half): the arithmetic restated here lives in HMMER 3.x (hmmer.org,
 * Debian package `hmmer`, version unpinned by the reference: docker/Dockerfile:5,
 * checkm/hmmer.py:90 ">= 3.1b1").  Neither a HMMER binary, its sources nor any golden
 * domtblout exist in /root/reference, so this file restates the published algorithm
 * (Eddy 2011, HMMER 3.1-3.4 User Guide, p7 pipeline) and is anchored only on the reference's
 * call site (checkm/markerGeneFinder.py:140-142 -> checkm/hmmer.py:61-74) and on the
 * consumption contract of the output (checkm/hmmer.py:184-200, 255-285).
 */
#ifndef P7ORACLE_H
#define P7ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P7O_K   20
#define P7O_KP  29
#define P7O_NL  64          /* canonical lane count of the float DP evaluation order */

enum { P7O_MMU = 0, P7O_MLAMBDA, P7O_VMU, P7O_VLAMBDA, P7O_FTAU, P7O_FLAMBDA };

typedef struct {
  char  *name, *acc, *desc;
  int    M;
  float *t;        /* [(M+1)*7]  MM MI MD IM II DM DD, probabilities */
  float *mat;      /* [(M+1)*20] */
  float *ins;      /* [(M+1)*20] */
  float  compo[P7O_K];
  int    has_compo;
  float  evparam[6];
  int    has_stats;
  float  ga[2], tc[2], nc[2];
  int    has_ga, has_tc, has_nc;
} P7O_HMM;

typedef struct {
  int       n;
  P7O_HMM **hmm;
} P7O_HMMSET;

/* one row of domtblout, before text formatting */
typedef struct {
  int32_t model_idx, seq_idx;
  int32_t tlen, qlen;
  double  full_evalue;
  float   full_score, full_bias;
  int32_t dom_idx, ndom;
  double  c_evalue, i_evalue;
  float   dom_score, dom_bias;
  int32_t hmm_from, hmm_to, ali_from, ali_to, env_from, env_to;
  float   acc;
  double  full_lnP, dom_lnP;
} P7O_ROW;

/* per-(model,sequence) stage trace used for kernel parity */
typedef struct {
  int32_t msv_xJ;       /* final xJ byte (0..255); -1 on overflow */
  float   msv_sc;       /* usc in nats (+inf on overflow) */
  float   null_sc;      /* null1 score */
  float   bias_sc;      /* bias-filter null score (filtersc) */
  int32_t vit_xC;       /* final xC word; 32767 flag on overflow; -32768 = -inf */
  float   vit_sc;       /* nats */
  float   fwd_sc;       /* nats (ForwardParser, multihit) */
  float   fwd_xC;       /* scaled xC at row L */
  int32_t fwd_nscale;   /* number of rescale events */
  int32_t pass_msv, pass_bias, pass_vit, pass_fwd;
} P7O_STAGES;

P7O_HMMSET *p7o_hmmset_read(const char *path, char *err, int errlen);
void        p7o_hmmset_free(P7O_HMMSET *s);
int         p7o_hmmset_n(const P7O_HMMSET *s);
const P7O_HMM *p7o_hmmset_get(const P7O_HMMSET *s, int i);
int         p7o_hmm_M(const P7O_HMM *h);
const char *p7o_hmm_name(const P7O_HMM *h);
const char *p7o_hmm_acc(const P7O_HMM *h);

/* text residues -> digital codes (HMMER amino alphabet "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~") */
void p7o_digitize(const char *seq, int64_t n, uint8_t *dsq);

/* run every stage unconditionally on one pair (no early exit), for kernel parity */
int p7o_stages(const P7O_HMM *hmm, const uint8_t *dsq, int L, P7O_STAGES *out);

/* full pipeline for one target set (= one bin): models in the given order against nseq sequences.
 * names[] are the sequence names (used only as the sort tie-break, as hmmsearch does).
 * Returns rows in domtblout order.  Caller frees *rows with p7o_free(). */
int p7o_search(const P7O_HMMSET *set, const int32_t *model_idx, int nmodels,
               const uint8_t *dsq, const int64_t *offsets, int nseq, const char *const *names,
               double E, double domE, P7O_ROW **rows, int *nrows);

/* format rows as hmmsearch --domtblout text (header + rows + trailer). Returns malloc'd string. */
char *p7o_format_domtblout(const P7O_HMMSET *set, const P7O_ROW *rows, int nrows,
                           const char *const *names, const char *const *descs);

/* throughput probe for bench.py's cpu_baseline: run the filter cascade + domain stage over
 * all pairs, return number of rows reported. */
void p7o_free(void *p);

/* the integer filters (byte MSV, word Viterbi) in their striped AVX2 form (oracle/p7simd.c) -- same results; for bench.py's cpu_baseline
 * kind "port-simd".  p7o_set_simd(1) is a no-op on a CPU without AVX2. */
int  p7o_simd_available(void);
void p7o_set_simd(int on);
int  p7o_get_simd(void);
/* the MSV filter alone over every pair: cells scored (return) and the sum of the final bytes (*checksum) */
int64_t p7o_msv_probe(const P7O_HMMSET *set, const int32_t *model_idx, int nmodels, const uint8_t *dsq, const int64_t *offsets, int nseq, int64_t *checksum);

/* canonical-order float DP pieces exposed for kernel unit parity */
int p7o_envelope(const P7O_HMM *hmm, const uint8_t *dsq, int L_full, int ienv, int jenv,
                 float *envsc, float *oasc, float *null2 /*[20]*/, int32_t *coords /*[4] hmmfrom,hmmto,alifrom,alito*/,
                 float *fwd_xC, int32_t *nscale);

/* hmmalign restated: optimal-accuracy alignment of the whole sequence to the unihit local profile; path[M] = residue (1-based)
 * emitted by each match state, 0 = none */
int p7o_align(const P7O_HMM *hmm, const uint8_t *dsq, int L, int32_t *path);
int p7o_envelope_alignment(const P7O_HMM *hmm, const uint8_t *dsq, int L_full, int ienv, int jenv, int32_t *path, float *pp);

/* multi-domain regions: 200-trace ensemble of region ireg..jreg (1-based, inclusive) of a sequence of length L.
 * n2sum[Lr]: per position, sum over traces of the null2 odds ratio.  seg_all[200*cap] / nseg_all[200]: each trace's
 * sampled segments in region-local coordinates.  env/nenv: clustered envelopes (region-local), sorted by start. */
typedef struct { int32_t sqfrom, sqto, hmmfrom, hmmto; } P7O_SEG;
uint32_t p7o_ensemble_seed(int t);
/* 1 (default): one stream per region carried from trace to trace (HMMER's own use of its generator); 0: one generator sub-stream per trace (rounds 1-3) */
void p7o_set_ensemble_stream(int sequential);
int p7o_region_ensemble(const P7O_HMM *hmm, const uint8_t *dsq, int L, int ireg, int jreg,
                        float *n2sum, P7O_SEG *seg_all, int32_t *nseg_all, int cap, P7O_SEG *env, int envcap, int32_t *nenv);

#ifdef __cplusplus
}
#endif
#endif
