/* p7oracle.c -- scalar CPU restatement of the hmmsearch (HMMER 3.x) per-target pipeline that
 * CheckM shells out to (checkm/markerGeneFinder.py:140-142 -> checkm/hmmer.py:61-74:
 * `hmmsearch --domtblout T --cpu N --notextw -E 0.1 --domE 0.1 [--noali] hmm faa`).
 *
 * TEST INFRASTRUCTURE ONLY -- see p7oracle.h.  PARITY UNPINNED for the scan half (no HMMER
 * binary, source or golden output exists in /root/reference); the reduce half has its own,
 * pinned, oracle in oracle/reduce_oracle.py.  Independent of HMMER, tests/test_oracle_bruteforce.py checks
 * Forward, null1, decoding/null2 and the optimal-accuracy alignment of this file against an enumeration of
 * every state path of the published profile on tiny models.
 *
 * What is restated, stage by stage (HMMER 3 "p7_Pipeline", Eddy 2011 PLoS Comp Biol 7:e1002195):
 *   profile file      HMMER3/f ASCII (values are -ln p; '*' = 0)       consumer: checkm/hmmerModelParser.py:54-83
 *   config            multihit local profile, occupancy-weighted entry, insert emissions = background
 *   null1             L*log(L/(L+1)) + log(1/(L+1))
 *   MSV filter        unsigned 8-bit, 1/3-bit units, base 190, saturating; P<=0.02 (Gumbel)
 *   bias filter       2-state composition HMM Forward replaces null1; P<=0.02
 *   Viterbi filter    signed 16-bit, 1/500-bit units, base 12000, -32768 = -inf; P<=1e-3 (Gumbel)
 *   Forward           float32 probability space with sparse rescaling at xE>1e4; P<=1e-5 (exponential)
 *   Backward, posterior domain heuristics (rt1 .25, rt2 .10, rt3 .20), per-envelope unihit
 *   Forward/Backward/decoding, null2 by expectation (omega 1/256), optimal-accuracy alignment,
 *   bit scores, lnP, E = P*Z, c-E = P*domZ, reporting at E<=0.1 / domE<=0.1, domtblout columns.
 *
 * Floating point: HMMER's SSE build sums in a 4-lane striped order that is an artefact of its
 * host ISA (its NEON/VMX/AVX builds differ in the last bits too).  This oracle fixes ONE
 * evaluation order -- the "canonical 64-lane blocked order" documented in DESIGN.md section 4 --
 * for every order-dependent float reduction (Forward/Backward D->D chain, E-state sums, null2
 * sums).  All transcendental calls (log/exp) happen on per-row or per-model scalars, never
 * per cell.  Known deviations from HMMER, all flagged "DEV" below:
 *   DEV1 bias-filter Forward rescales by exact powers of two instead of dividing by the row max;
 *   DEV2 optimal-accuracy fill gates impossible transitions with -inf instead of *FLT_MIN;
 *   DEV3 (closed in round 4) multi-domain regions (rt3 test) are resolved as HMMER does -- 200 stochastic tracebacks of a
 *        multihit Forward matrix of the region, null2 by trace, single-linkage clustering of the sampled
 *        segments (overlap .8 of the smaller, diagonal 4, posterior .25, endpoint .02) -- with HMMER's fast
 *        generator (x*69069+1, seed 42 mixed as Easel does), re-seeded for every region and carried from the end of trace t
 *        into trace t+1, as hmmsearch carries its generator (every state of a trace but N takes one draw).  Rounds 1-3
 *        gave trace t its own substream (start + t*15485863 steps) by default; that mode survives only as an opt-in
 *        (p7o_set_ensemble_stream(0); product: CKM_ENS_STREAM=substream);
 *   DEV4 exp() for the probability-space tables uses libm expf, not HMMER's SSE polynomial.
 */
#define _GNU_SOURCE
#include "p7oracle.h"
#include "p7simd.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include <ctype.h>
#include <strings.h>
#include <sys/types.h>

#define LOG2C   0.69314718055994529
#define LOG2RC  1.44269504088896341
#define NEGINF  (-INFINITY)
#define F1 0.02
#define F2 1e-3
#define F3 1e-5
#define RT1 0.25f
#define RT2 0.10f
#define RT3 0.20f
#define OMEGA (1.0f/256.0f)

enum { tMM = 0, tMI, tMD, tIM, tII, tDM, tDD };

static const char AMINO[] = "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~";

/* Swiss-Prot 50.8 amino acid background (HMMER null1) */
static const float BGF[20] = {
  0.0787945f, 0.0151600f, 0.0535222f, 0.0668298f, 0.0397062f, 0.0695071f, 0.0229198f, 0.0590092f,
  0.0594422f, 0.0963728f, 0.0237718f, 0.0414386f, 0.0482904f, 0.0395639f, 0.0540978f, 0.0683364f,
  0.0540687f, 0.0673417f, 0.0114135f, 0.0304133f };

/* degen[x][y]: residue y is in degenerate symbol x (x = 21..26) */
static int degen(int x, int y)
{
  switch (x) {
  case 21: return (AMINO[y] == 'D' || AMINO[y] == 'N');   /* B */
  case 22: return (AMINO[y] == 'I' || AMINO[y] == 'L');   /* J */
  case 23: return (AMINO[y] == 'E' || AMINO[y] == 'Q');   /* Z */
  case 24: return (AMINO[y] == 'K');                      /* O pyrrolysine -> K */
  case 25: return (AMINO[y] == 'C');                      /* U selenocysteine -> C */
  case 26: return 1;                                       /* X */
  default: return (x == y);
  }
}

void p7o_free(void *p) { free(p); }

void p7o_digitize(const char *seq, int64_t n, uint8_t *dsq)
{
  static int8_t map[256]; static int init = 0;
  if (!init) {
    memset(map, -1, sizeof(map));
    for (int i = 0; i < P7O_KP; i++) { map[(unsigned char)AMINO[i]] = (int8_t)i; map[(unsigned char)tolower(AMINO[i])] = (int8_t)i; }
    init = 1;
  }
  for (int64_t i = 0; i < n; i++) { int8_t c = map[(unsigned char)seq[i]]; dsq[i] = (uint8_t)(c < 0 ? 26 : c); }
}

/* ------------------------------------------------------------------------------------------
 * HMMER3/f ASCII profile reader
 * ------------------------------------------------------------------------------------------ */
static char *xstrdup(const char *s) { char *r = malloc(strlen(s) + 1); strcpy(r, s); return r; }
static void rstrip(char *s) { size_t n = strlen(s); while (n && isspace((unsigned char)s[n-1])) s[--n] = 0; }
static float prob_tok(const char *tok) { return (*tok == '*') ? 0.0f : expf((float)(-1.0 * atof(tok))); }

static void hmm_free(P7O_HMM *h)
{ if (!h) return; free(h->name); free(h->acc); free(h->desc); free(h->t); free(h->mat); free(h->ins); free(h); }

void p7o_hmmset_free(P7O_HMMSET *s)
{ if (!s) return; for (int i = 0; i < s->n; i++) hmm_free(s->hmm[i]); free(s->hmm); free(s); }
int p7o_hmmset_n(const P7O_HMMSET *s) { return s->n; }
const P7O_HMM *p7o_hmmset_get(const P7O_HMMSET *s, int i) { return s->hmm[i]; }
int p7o_hmm_M(const P7O_HMM *h) { return h->M; }
const char *p7o_hmm_name(const P7O_HMM *h) { return h->name; }
const char *p7o_hmm_acc(const P7O_HMM *h) { return h->acc ? h->acc : ""; }

static int read_floats(char *line, int skip_first, float *dst, int n, int as_prob)
{
  char *save = NULL; char *tok = strtok_r(line, " \t\r\n", &save);
  if (skip_first) tok = strtok_r(NULL, " \t\r\n", &save);
  for (int i = 0; i < n; i++) {
    if (!tok) return -1;
    dst[i] = as_prob ? prob_tok(tok) : (float)atof(tok);
    tok = strtok_r(NULL, " \t\r\n", &save);
  }
  return 0;
}

static void calc_occupancy(const P7O_HMM *h, float *mocc, float *iocc)
{
  int M = h->M;
  mocc[0] = 0.f;
  mocc[1] = h->t[0*7+tMI] + h->t[0*7+tMM];
  for (int k = 2; k <= M; k++)
    mocc[k] = mocc[k-1] * (h->t[(k-1)*7+tMM] + h->t[(k-1)*7+tMI]) + (1.0f - mocc[k-1]) * h->t[(k-1)*7+tDM];
  if (iocc) {
    iocc[0] = h->t[0*7+tMI] / h->t[0*7+tIM];
    for (int k = 1; k < M; k++) iocc[k] = mocc[k] * h->t[k*7+tMI] / h->t[k*7+tIM];
    iocc[M] = 0.f;
  }
}

static void set_composition(P7O_HMM *h)
{
  int M = h->M; float *mocc = malloc(sizeof(float)*(M+1)), *iocc = malloc(sizeof(float)*(M+1));
  calc_occupancy(h, mocc, iocc);
  for (int x = 0; x < 20; x++) h->compo[x] = 0.f;
  for (int x = 0; x < 20; x++) h->compo[x] += h->ins[x] * iocc[0];
  for (int k = 1; k <= M; k++) for (int x = 0; x < 20; x++) {
    h->compo[x] += h->mat[k*20+x] * mocc[k];
    h->compo[x] += h->ins[k*20+x] * iocc[k];
  }
  float s = 0.f; for (int x = 0; x < 20; x++) s += h->compo[x];
  for (int x = 0; x < 20; x++) h->compo[x] /= s;
  h->has_compo = 1; free(mocc); free(iocc);
}

P7O_HMMSET *p7o_hmmset_read(const char *path, char *err, int errlen)
{
  FILE *fp = fopen(path, "r");
  if (!fp) { if (err) snprintf(err, errlen, "cannot open %s", path); return NULL; }
  P7O_HMMSET *set = calloc(1, sizeof(*set));
  size_t cap = 0; char *line = NULL; ssize_t len;
  P7O_HMM *h = NULL; int lineno = 0;
#define FAIL(msg) do { if (err) snprintf(err, errlen, "%s:%d: %s", path, lineno, msg); goto bad; } while (0)
  while ((len = getline(&line, &cap, fp)) >= 0) {
    lineno++;
    if (!h) {
      if (strncmp(line, "HMMER3/", 7) == 0) { h = calloc(1, sizeof(*h)); h->M = -1; }
      else { rstrip(line); if (*line) FAIL("expected HMMER3/ magic"); }
      continue;
    }
    if (strncmp(line, "HMM ", 4) == 0 || strncmp(line, "HMM\t", 4) == 0) {
      /* body */
      if (h->M <= 0) FAIL("LENG missing");
      if (!h->name) FAIL("NAME missing");
      int M = h->M;
      h->t = calloc((size_t)(M+1)*7, sizeof(float)); h->mat = calloc((size_t)(M+1)*20, sizeof(float)); h->ins = calloc((size_t)(M+1)*20, sizeof(float));
      if (getline(&line, &cap, fp) < 0) FAIL("truncated"); lineno++;    /* transition legend */
      if (getline(&line, &cap, fp) < 0) FAIL("truncated"); lineno++;
      { char *p = line; while (isspace((unsigned char)*p)) p++;
        if (strncmp(p, "COMPO", 5) == 0) {
          if (read_floats(line, 1, h->compo, 20, 1)) FAIL("bad COMPO"); h->has_compo = 1;
          if (getline(&line, &cap, fp) < 0) FAIL("truncated"); lineno++;
        } }
      if (read_floats(line, 0, h->ins, 20, 1)) FAIL("bad node-0 insert line");
      if (getline(&line, &cap, fp) < 0) FAIL("truncated"); lineno++;
      if (read_floats(line, 0, h->t, 7, 1)) FAIL("bad node-0 transitions");
      h->mat[0] = 1.0f;
      for (int k = 1; k <= M; k++) {
        if (getline(&line, &cap, fp) < 0) FAIL("truncated"); lineno++;
        { char *p = line; while (isspace((unsigned char)*p)) p++; if (atoi(p) != k) FAIL("node index mismatch"); }
        if (read_floats(line, 1, h->mat + k*20, 20, 1)) FAIL("bad match line");
        if (getline(&line, &cap, fp) < 0) FAIL("truncated"); lineno++;
        if (read_floats(line, 0, h->ins + k*20, 20, 1)) FAIL("bad insert line");
        if (getline(&line, &cap, fp) < 0) FAIL("truncated"); lineno++;
        if (read_floats(line, 0, h->t + k*7, 7, 1)) FAIL("bad transition line");
      }
      if (getline(&line, &cap, fp) < 0) FAIL("truncated"); lineno++;
      if (strncmp(line, "//", 2) != 0) FAIL("expected //");
      if (!h->has_compo) set_composition(h);
      set->hmm = realloc(set->hmm, sizeof(P7O_HMM*) * (set->n + 1)); set->hmm[set->n++] = h; h = NULL;
      continue;
    }
    /* header line */
    rstrip(line);
    char *p = line; while (*p && !isspace((unsigned char)*p)) p++;
    if (!*p) continue;               /* tag with no value: ignored here (CheckM's parser raises; see host mirror) */
    *p++ = 0; while (isspace((unsigned char)*p)) p++;
    if      (!strcmp(line, "NAME")) h->name = xstrdup(p);
    else if (!strcmp(line, "ACC"))  h->acc  = xstrdup(p);
    else if (!strcmp(line, "DESC")) h->desc = xstrdup(p);
    else if (!strcmp(line, "LENG")) h->M = atoi(p);
    else if (!strcmp(line, "ALPH")) { if (strcasecmp(p, "amino")) FAIL("only amino profiles supported"); }
    else if (!strcmp(line, "GA")) { if (sscanf(p, "%f %f", &h->ga[0], &h->ga[1]) == 2) h->has_ga = 1; }
    else if (!strcmp(line, "TC")) { if (sscanf(p, "%f %f", &h->tc[0], &h->tc[1]) == 2) h->has_tc = 1; }
    else if (!strcmp(line, "NC")) { if (sscanf(p, "%f %f", &h->nc[0], &h->nc[1]) == 2) h->has_nc = 1; }
    else if (!strcmp(line, "STATS")) {
      char a[32], b[32]; float v1, v2;
      if (sscanf(p, "%31s %31s %f %f", a, b, &v1, &v2) != 4 || strcmp(a, "LOCAL")) FAIL("bad STATS");
      if      (!strcmp(b, "MSV"))     { h->evparam[P7O_MMU] = v1;  h->evparam[P7O_MLAMBDA] = v2; h->has_stats |= 1; }
      else if (!strcmp(b, "VITERBI")) { h->evparam[P7O_VMU] = v1;  h->evparam[P7O_VLAMBDA] = v2; h->has_stats |= 2; }
      else if (!strcmp(b, "FORWARD")) { h->evparam[P7O_FTAU] = v1; h->evparam[P7O_FLAMBDA] = v2; h->has_stats |= 4; }
      else FAIL("bad STATS kind");
    }
  }
  free(line); fclose(fp);
  if (h) { hmm_free(h); if (err) snprintf(err, errlen, "%s: truncated record", path); p7o_hmmset_free(set); return NULL; }
  for (int i = 0; i < set->n; i++) if (set->hmm[i]->has_stats != 7) {
    if (err) snprintf(err, errlen, "%s: model %s lacks STATS LOCAL calibration", path, set->hmm[i]->name);
    p7o_hmmset_free(set); return NULL;
  }
  return set;
bad:
  free(line); fclose(fp); hmm_free(h); p7o_hmmset_free(set); return NULL;
#undef FAIL
}

/* ------------------------------------------------------------------------------------------
 * Profile configuration (generic scores + the three reduced-precision score systems)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const P7O_HMM *hmm;
  int M, Q, Mp;
  float *msc;                                   /* [KP][M+1] generic match scores (nats) */
  float *gBM, *gMM, *gIM, *gDM, *gMD, *gDD, *gMI, *gII;   /* generic log transitions, from-node index 0..M-1 (gBM[k-1] = entry at k) */
  /* MSV */
  uint8_t *rbv; int tbm_b, tec_b, base_b, bias_b; float scale_b;
  /* ViterbiFilter; "into k" arrays indexed by k=1..M, "from k" arrays indexed by k=1..M */
  int16_t *rwv; int16_t *wBM, *wMM, *wIM, *wDM, *wMD, *wMI, *wII, *wDD;
  float scale_w; int base_w, wE_loop, wE_move;
  /* Forward/Backward odds, canonical padded layout idx = k-1 in [0,Mp) */
  float *rf; float *fBM, *fMM, *fIM, *fDM, *fMI, *fII, *fMD, *fDD;
  float fE_loop, fE_move;                        /* multihit E->J, E->C */
  P7S_PROF *simd;                                /* the integer filters' striped tables (p7o_set_simd(1) on a CPU with AVX2), else NULL */
} PROF;

/* The integer filters in their striped AVX2 form (oracle/p7simd.c) instead of the scalar loops: same bytes, same words, so the same rows
 * -- what bench.py times as cpu_baseline kind "port-simd".  Off by default: the scalar loops are the oracle the GPU path is diffed with. */
static int g_simd = 0;
int p7o_simd_available(void) { __builtin_cpu_init(); return __builtin_cpu_supports("avx2") ? 1 : 0; }
void p7o_set_simd(int on) { g_simd = on && p7o_simd_available(); }
int p7o_get_simd(void) { return g_simd; }

static uint8_t unbiased_byteify(float scale_b, float sc)
{ sc = -1.0f * roundf(scale_b * sc); return (sc > 255.f) ? 255 : (uint8_t)(int)sc; }
static uint8_t biased_byteify(float scale_b, int bias_b, float sc)
{ sc = -1.0f * roundf(scale_b * sc); return (sc > (float)(255 - bias_b)) ? 255 : (uint8_t)((int)sc + bias_b); }
static int16_t wordify(float scale_w, float sc)
{ sc = roundf(scale_w * sc); if (sc >= 32767.0f) return 32767; if (sc <= -32768.0f) return -32768; return (int16_t)sc; }

static void prof_free(PROF *p)
{
  if (!p) return;
  p7s_free(p->simd);
  free(p->msc); free(p->gBM); free(p->rbv); free(p->rwv); free(p->wBM); free(p->rf); free(p->fBM); free(p);
}

/* canonical slots per lane: smallest member of the fixed set covering M */
static int canon_q(int M)
{
  static const int set[] = { 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64 };
  for (unsigned i = 0; i < sizeof(set)/sizeof(set[0]); i++) if (set[i] * P7O_NL >= M) return set[i];
  return (M + P7O_NL - 1) / P7O_NL;
}

static PROF *prof_create(const P7O_HMM *h)
{
  int M = h->M;
  PROF *p = calloc(1, sizeof(*p));
  p->hmm = h; p->M = M; p->Q = canon_q(M); p->Mp = p->Q * P7O_NL;
  /* --- generic transition scores --- */
  p->gBM = malloc(sizeof(float) * 8 * (size_t)(M+1));
  p->gMM = p->gBM + (M+1); p->gIM = p->gMM + (M+1); p->gDM = p->gIM + (M+1);
  p->gMD = p->gDM + (M+1); p->gDD = p->gMD + (M+1); p->gMI = p->gDD + (M+1); p->gII = p->gMI + (M+1);
  for (int k = 0; k <= M; k++) p->gBM[k] = p->gMM[k] = p->gIM[k] = p->gDM[k] = p->gMD[k] = p->gDD[k] = p->gMI[k] = p->gII[k] = NEGINF;
  { /* local entry: occ[k] / sum_i occ[i]*(M-i+1), stored off by one */
    float *occ = malloc(sizeof(float)*(M+1)); float Z = 0.f;
    calc_occupancy(h, occ, NULL);
    for (int k = 1; k <= M; k++) Z += occ[k] * (float)(M-k+1);
    for (int k = 1; k <= M; k++) p->gBM[k-1] = (float)log(occ[k] / Z);
    free(occ);
  }
  for (int k = 1; k < M; k++) {
    p->gMM[k] = (float)log(h->t[k*7+tMM]); p->gMI[k] = (float)log(h->t[k*7+tMI]); p->gMD[k] = (float)log(h->t[k*7+tMD]);
    p->gIM[k] = (float)log(h->t[k*7+tIM]); p->gII[k] = (float)log(h->t[k*7+tII]);
    p->gDM[k] = (float)log(h->t[k*7+tDM]); p->gDD[k] = (float)log(h->t[k*7+tDD]);
  }
  /* --- generic match scores, degenerate residues by background-weighted expectation --- */
  p->msc = malloc(sizeof(float) * P7O_KP * (size_t)(M+1));
  for (int x = 0; x < P7O_KP; x++) p->msc[x*(M+1)] = NEGINF;
  for (int k = 1; k <= M; k++) {
    float sc[P7O_KP];
    for (int x = 0; x < 20; x++) sc[x] = (float)log((double)h->mat[k*20+x] / BGF[x]);
    sc[20] = NEGINF; sc[27] = NEGINF; sc[28] = NEGINF;
    for (int x = 21; x <= 26; x++) {
      float result = 0.f, denom = 0.f;
      for (int y = 0; y < 20; y++) if (degen(x, y)) { result += sc[y] * BGF[y]; denom += BGF[y]; }
      sc[x] = result / denom;
    }
    for (int x = 0; x < P7O_KP; x++) p->msc[x*(M+1)+k] = sc[x];
  }
  /* --- MSV: unsigned bytes, 1/3 bit units --- */
  {
    float max = 0.0f;   /* insert scores (0) are part of the max in HMMER, so max >= 0 */
    for (int x = 0; x < 20; x++) for (int k = 1; k <= M; k++) if (p->msc[x*(M+1)+k] > max) max = p->msc[x*(M+1)+k];
    p->scale_b = (float)(3.0 / LOG2C);
    p->base_b  = 190;
    p->bias_b  = unbiased_byteify(p->scale_b, -1.0f * max);
    p->rbv = malloc((size_t)P7O_KP * (M+1));
    for (int x = 0; x < P7O_KP; x++) { p->rbv[x*(M+1)] = 255; for (int k = 1; k <= M; k++) p->rbv[x*(M+1)+k] = biased_byteify(p->scale_b, p->bias_b, p->msc[x*(M+1)+k]); }
    p->tbm_b = unbiased_byteify(p->scale_b, logf(2.0f / ((float)M * (float)(M+1))));
    p->tec_b = unbiased_byteify(p->scale_b, logf(0.5f));
  }
  /* --- ViterbiFilter: signed words, 1/500 bit units --- */
  {
    p->scale_w = (float)(500.0 / LOG2C); p->base_w = 12000;
    p->rwv = malloc(sizeof(int16_t) * P7O_KP * (size_t)(M+1));
    for (int x = 0; x < P7O_KP; x++) { p->rwv[x*(M+1)] = -32768; for (int k = 1; k <= M; k++) p->rwv[x*(M+1)+k] = wordify(p->scale_w, p->msc[x*(M+1)+k]); }
    p->wBM = malloc(sizeof(int16_t) * 8 * (size_t)(M+2));
    p->wMM = p->wBM + (M+2); p->wIM = p->wMM + (M+2); p->wDM = p->wIM + (M+2);
    p->wMD = p->wDM + (M+2); p->wMI = p->wMD + (M+2); p->wII = p->wMI + (M+2); p->wDD = p->wII + (M+2);
    for (int k = 0; k <= M+1; k++) p->wBM[k] = p->wMM[k] = p->wIM[k] = p->wDM[k] = p->wMD[k] = p->wMI[k] = p->wII[k] = p->wDD[k] = -32768;
#define CLAMPW(v, mx) ((v) <= (mx) ? (v) : (mx))
    for (int k = 1; k <= M; k++) {
      int16_t v;
      v = wordify(p->scale_w, p->gBM[k-1]); p->wBM[k] = CLAMPW(v, 0);
      v = wordify(p->scale_w, p->gMM[k-1]); p->wMM[k] = CLAMPW(v, 0);
      v = wordify(p->scale_w, p->gIM[k-1]); p->wIM[k] = CLAMPW(v, 0);
      v = wordify(p->scale_w, p->gDM[k-1]); p->wDM[k] = CLAMPW(v, 0);
      if (k < M) {
        v = wordify(p->scale_w, p->gMD[k]); p->wMD[k] = CLAMPW(v, 0);
        v = wordify(p->scale_w, p->gMI[k]); p->wMI[k] = CLAMPW(v, 0);
        v = wordify(p->scale_w, p->gII[k]); p->wII[k] = CLAMPW(v, -1);
        p->wDD[k] = wordify(p->scale_w, p->gDD[k]);
      }
    }
    p->wE_loop = wordify(p->scale_w, (float)(-LOG2C));
    p->wE_move = wordify(p->scale_w, (float)(-LOG2C));
  }
  /* --- Forward/Backward odds ratios, canonical padded layout --- */
  {
    int Mp = p->Mp;
    p->rf = calloc((size_t)P7O_KP * Mp, sizeof(float));
    for (int x = 0; x < P7O_KP; x++) for (int k = 1; k <= M; k++) p->rf[x*Mp + k-1] = expf(p->msc[x*(M+1)+k]);
    p->fBM = calloc(8 * (size_t)Mp, sizeof(float));
    p->fMM = p->fBM + Mp; p->fIM = p->fMM + Mp; p->fDM = p->fIM + Mp;
    p->fMI = p->fDM + Mp; p->fII = p->fMI + Mp; p->fMD = p->fII + Mp; p->fDD = p->fMD + Mp;
    for (int k = 1; k <= M; k++) {
      int i = k-1;
      p->fBM[i] = expf(p->gBM[k-1]); p->fMM[i] = expf(p->gMM[k-1]); p->fIM[i] = expf(p->gIM[k-1]); p->fDM[i] = expf(p->gDM[k-1]);
      if (k < M) { p->fMI[i] = expf(p->gMI[k]); p->fII[i] = expf(p->gII[k]); p->fMD[i] = expf(p->gMD[k]); p->fDD[i] = expf(p->gDD[k]); }
    }
    p->fE_loop = expf((float)(-LOG2C)); p->fE_move = expf((float)(-LOG2C));
  }
  if (g_simd) p->simd = p7s_create(M, p->rbv, p->bias_b, p->base_b, p->tbm_b, p->tec_b, p->rwv, p->wBM, p->base_w, p->wE_loop, p->wE_move);
  return p;
}

/* length-dependent specials */
typedef struct { float loop, move; int w_move; int tjb_b; float nullsc; float p1; } LENCFG;
static void lencfg(const PROF *p, int L, int multihit, LENCFG *c)
{
  float nj = multihit ? 1.0f : 0.0f;
  c->move = (2.0f + nj) / ((float)L + 2.0f + nj);
  c->loop = 1.0f - c->move;
  c->w_move = wordify(p->scale_w, logf(c->move));
  c->tjb_b  = unbiased_byteify(p->scale_b, logf(3.0f / (float)(L+3)));
  c->p1 = (float)L / (float)(L+1);
  c->nullsc = (float)((float)L * log((double)c->p1) + log(1. - (double)c->p1));
}

/* ------------------------------------------------------------------------------------------
 * Statistics (Easel)
 * ------------------------------------------------------------------------------------------ */
static double gumbel_surv(double x, double mu, double lambda)
{ double y = lambda * (x - mu); double ey = -exp(-y); if (fabs(ey) < 5e-9) return -ey; return 1 - exp(ey); }
static double exp_surv(double x, double mu, double lambda) { if (x < mu) return 1.0; return exp(-lambda * (x - mu)); }
static double exp_logsurv(double x, double mu, double lambda) { if (x < mu) return 0.0; return -lambda * (x - mu); }

static float flogsum_tbl[16000]; static int flogsum_init = 0;
static float flogsum(float a, float b)
{
  if (!flogsum_init) { for (int i = 0; i < 16000; i++) flogsum_tbl[i] = (float)log(1. + exp((double)-i / 1000.)); flogsum_init = 1; }
  const float max = (a > b) ? a : b, min = (a > b) ? b : a;
  return (min == NEGINF || (max - min) >= 15.7f) ? max : max + flogsum_tbl[(int)((max - min) * 1000.f)];
}

/* ------------------------------------------------------------------------------------------
 * MSV filter: u8 saturating arithmetic, exactly as the striped SSE filter computes it
 * (striping does not change max/saturating-add results).  Returns 0 ok, 1 overflow.
 * ------------------------------------------------------------------------------------------ */
static inline int sat_addu8(int a, int b) { int s = a + b; return s > 255 ? 255 : s; }
static inline int sat_subu8(int a, int b) { int s = a - b; return s < 0 ? 0 : s; }

static int msv_filter(const PROF *p, const LENCFG *lc, const uint8_t *dsq, int L, int *ret_xJ, float *ret_sc)
{
  int M = p->M;
  if (p->simd) {
    int xJ;
    if (p7s_msv(p->simd, dsq, L, lc->tjb_b, &xJ)) { *ret_xJ = -1; *ret_sc = INFINITY; return 1; }
    *ret_xJ = xJ;
    float sc = ((float)(xJ - lc->tjb_b) - (float)p->base_b);
    sc /= p->scale_b;
    sc -= 3.0f;
    *ret_sc = sc;
    return 0;
  }
  uint8_t *dp = calloc((size_t)M + 1, 1), *nw = calloc((size_t)M + 1, 1);
  int tjbm = (lc->tjb_b + p->tbm_b) & 0xff;     /* _mm_set1_epi8(tjb_b + tbm_b): low byte of the sum */
  int xJ = 0, xB = sat_subu8(p->base_b, tjbm);
  for (int i = 1; i <= L; i++) {
    const uint8_t *rsc = p->rbv + (size_t)dsq[i-1] * (M+1);
    int xE = 0;
    for (int k = 1; k <= M; k++) {
      int sv = dp[k-1] > xB ? dp[k-1] : xB;      /* dp[0] = 0 = -inf */
      sv = sat_addu8(sv, p->bias_b);
      sv = sat_subu8(sv, rsc[k]);
      if (sv > xE) xE = sv;
      nw[k] = (uint8_t)sv;
    }
    { uint8_t *t = dp; dp = nw; nw = t; dp[0] = 0; }
    if (sat_addu8(xE, p->bias_b) == 255) { free(dp); free(nw); *ret_xJ = -1; *ret_sc = INFINITY; return 1; }
    xE = sat_subu8(xE, p->tec_b);
    if (xE > xJ) xJ = xE;
    xB = (p->base_b > xJ) ? p->base_b : xJ;
    xB = sat_subu8(xB, tjbm);
  }
  free(dp); free(nw);
  *ret_xJ = xJ;
  float sc = ((float)(xJ - lc->tjb_b) - (float)p->base_b);
  sc /= p->scale_b;
  sc -= 3.0f;
  *ret_sc = sc;
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * Bias filter: Forward score of the 2-state composition HMM (p7_bg_SetFilter/FilterScore).
 * DEV1: rescaling by exact powers of two.
 * ------------------------------------------------------------------------------------------ */
typedef struct { float t00, t01, t10, t11, pi0, pi1; float eo1[P7O_KP]; } BIASHMM;

static void bias_setup(const P7O_HMM *h, BIASHMM *b)
{
  float L0 = 400.0f, L1 = (float)h->M / 8.0f;
  b->t00 = L0 / (L0 + 1.0f); b->t01 = 1.0f / (L0 + 1.0f);
  b->t10 = 1.0f / (L1 + 1.0f); b->t11 = L1 / (L1 + 1.0f);
  b->pi0 = 0.999f; b->pi1 = 0.001f;
  for (int x = 0; x < 20; x++) b->eo1[x] = h->compo[x] / BGF[x];
  b->eo1[20] = b->eo1[27] = b->eo1[28] = 1.0f;
  for (int x = 21; x <= 26; x++) {
    float num = 0.f, den = 0.f;
    for (int y = 0; y < 20; y++) if (degen(x, y)) { num += h->compo[y]; den += BGF[y]; }
    b->eo1[x] = (den > 0.f) ? num / den : 0.f;
  }
}

/* state-0 emission odds are f/f = 1 for every residue (degenerate ones: sum f / sum f = 1) */
static float bias_filter(const BIASHMM *b, const LENCFG *lc, const uint8_t *dsq, int L)
{
  float d0, d1; int nexp = 0;
  if (L == 0) return lc->nullsc;
  d0 = b->pi0; d1 = b->eo1[dsq[0]] * b->pi1;
  for (int i = 2; i <= L; i++) {
    float n0 = d0 * b->t00 + d1 * b->t10;
    float n1 = (d0 * b->t01 + d1 * b->t11) * b->eo1[dsq[i-1]];
    d0 = n0; d1 = n1;
    float mx = d0 > d1 ? d0 : d1;
    if (mx < 0x1p-40f)      { d0 *= 0x1p64f;  d1 *= 0x1p64f;  nexp -= 64; }
    else if (mx > 0x1p40f)  { d0 *= 0x1p-64f; d1 *= 0x1p-64f; nexp += 64; }
  }
  float tot = d0 + d1;               /* both states -> E with probability 1 */
  float nullsc = (float)(log((double)tot) + (double)nexp * LOG2C);
  return nullsc + (float)L * logf(lc->p1) + logf(1.0f - lc->p1);
}

/* ------------------------------------------------------------------------------------------
 * Viterbi filter: i16 saturating.  Returns 0 ok, 1 overflow.
 * ------------------------------------------------------------------------------------------ */
static inline int sat16(int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }

static int vit_filter(const PROF *p, const LENCFG *lc, const uint8_t *dsq, int L, int *ret_xC, float *ret_sc)
{
  int M = p->M;
  if (p->simd) {
    int xC;
    if (p7s_vit(p->simd, dsq, L, lc->w_move, &xC)) { *ret_xC = 32767; *ret_sc = INFINITY; return 1; }
    *ret_xC = xC;
    if (xC > -32768) {
      float sc = (float)xC + (float)lc->w_move - (float)p->base_w;
      sc /= p->scale_w;
      sc -= 3.0f;
      *ret_sc = sc;
    } else *ret_sc = NEGINF;
    return 0;
  }
  int16_t *vbase = malloc(sizeof(int16_t) * 6 * (size_t)(M+1));
  int16_t *mm = vbase;
  int16_t *im = mm + (M+1), *dm = im + (M+1), *mn = dm + (M+1), *in = mn + (M+1), *dn = in + (M+1);
  for (int k = 0; k <= M; k++) mm[k] = im[k] = dm[k] = mn[k] = in[k] = dn[k] = -32768;
  int xN = p->base_w, xB = xN + lc->w_move, xJ = -32768, xC = -32768, xE;
  for (int i = 1; i <= L; i++) {
    const int16_t *rsc = p->rwv + (size_t)dsq[i-1] * (M+1);
    xE = -32768;
    for (int k = 1; k <= M; k++) {
      int sv = sat16(xB + p->wBM[k]);
      int t;
      t = sat16(mm[k-1] + p->wMM[k]); if (t > sv) sv = t;
      t = sat16(im[k-1] + p->wIM[k]); if (t > sv) sv = t;
      t = sat16(dm[k-1] + p->wDM[k]); if (t > sv) sv = t;
      sv = sat16(sv + rsc[k]);
      if (sv > xE) xE = sv;
      mn[k] = (int16_t)sv;
      int a = sat16(mm[k] + p->wMI[k]), b2 = sat16(im[k] + p->wII[k]);
      in[k] = (int16_t)(a > b2 ? a : b2);
    }
    dn[1] = -32768;
    for (int k = 2; k <= M; k++) {
      int a = sat16(mn[k-1] + p->wMD[k-1]), b2 = sat16(dn[k-1] + p->wDD[k-1]);
      dn[k] = (int16_t)(a > b2 ? a : b2);
    }
    if (xE >= 32767) { free(vbase); *ret_xC = 32767; *ret_sc = INFINITY; return 1; }
    /* xN loop = 0, xC/xJ loops = 0 (the -3 nat NN/CC/JJ approximation) */
    { int a = xC, b2 = xE + p->wE_move; xC = a > b2 ? a : b2; }
    { int a = xJ, b2 = xE + p->wE_loop; xJ = a > b2 ? a : b2; }
    { int a = xJ + lc->w_move, b2 = xN + lc->w_move; xB = a > b2 ? a : b2; }
    { int16_t *t; t = mm; mm = mn; mn = t; t = im; im = in; in = t; t = dm; dm = dn; dn = t; }
  }
  free(vbase);
  *ret_xC = xC;
  if (xC > -32768) {
    float sc = (float)xC + (float)lc->w_move - (float)p->base_w;
    sc /= p->scale_w;
    sc -= 3.0f;
    *ret_sc = sc;
  } else *ret_sc = NEGINF;
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * Forward / Backward in probability space, canonical 64-lane blocked evaluation order
 * (lane-local folds, lane scans by lane_scan_up/lane_scan_down, sums by the xor butterfly 1,2,4,8,16,32).
 * Cell idx = z*Q + q  (lane z, slot q)  <->  model node k = idx+1.
 * ------------------------------------------------------------------------------------------ */
typedef struct { float E_loop, E_move, loop, move; } XF;

static void xf_config(const PROF *p, int L, int multihit, XF *xf)
{
  LENCFG lc; lencfg(p, L, multihit, &lc);
  xf->loop = lc.loop; xf->move = lc.move;
  if (multihit) { xf->E_loop = p->fE_loop; xf->E_move = p->fE_move; }
  else          { xf->E_loop = 0.0f;       xf->E_move = 1.0f; }
}

/* (a,b) <- (a,b) after (oa,ob):  d -> a*(oa*d + ob) + b */
#define COMPOSE(a, b, oa, ob) do { float t_ = (a) * (ob); (b) = (b) + t_; (a) = (a) * (oa); } while (0)

/* Inclusive scan of the 64 lanes' affine maps towards higher lanes, canonical association: inside each row of 16 lanes
 * Kogge-Stone with offsets 1,2,4,8 (a lane without a partner composes with the identity), then the row totals are
 * composed one after the other and every lane applies the prefix of the rows below its own. */
static void lane_scan_up(float *A, float *B)
{
  float nA[P7O_NL], nB[P7O_NL];
  for (int s = 1; s < 16; s <<= 1) {
    for (int z = 0; z < P7O_NL; z++) {
      float oa = ((z & 15) >= s) ? A[z-s] : 1.0f, ob = ((z & 15) >= s) ? B[z-s] : 0.0f;
      float a = A[z], b = B[z]; COMPOSE(a, b, oa, ob); nA[z] = a; nB[z] = b;
    }
    memcpy(A, nA, sizeof(nA)); memcpy(B, nB, sizeof(nB));
  }
  float a0 = A[15], b0 = B[15], a1 = A[31], b1 = B[31], a2 = A[47], b2 = B[47];
  float pa2 = a1, pb2 = b1; COMPOSE(pa2, pb2, a0, b0);
  float pa3 = a2, pb3 = b2; COMPOSE(pa3, pb3, pa2, pb2);
  for (int z = 0; z < P7O_NL; z++) {
    int row = z >> 4;
    float pa = row == 0 ? 1.0f : row == 1 ? a0 : row == 2 ? pa2 : pa3;
    float pb = row == 0 ? 0.0f : row == 1 ? b0 : row == 2 ? pb2 : pb3;
    COMPOSE(A[z], B[z], pa, pb);
  }
}
/* mirror image: lane z ends with the composition of lanes z..63 (its own map applied last) */
static void lane_scan_down(float *A, float *B)
{
  float nA[P7O_NL], nB[P7O_NL];
  for (int s = 1; s < 16; s <<= 1) {
    for (int z = 0; z < P7O_NL; z++) {
      float oa = ((z & 15) + s <= 15) ? A[z+s] : 1.0f, ob = ((z & 15) + s <= 15) ? B[z+s] : 0.0f;
      float a = A[z], b = B[z]; COMPOSE(a, b, oa, ob); nA[z] = a; nB[z] = b;
    }
    memcpy(A, nA, sizeof(nA)); memcpy(B, nB, sizeof(nB));
  }
  float a3 = A[48], b3 = B[48], a2 = A[32], b2 = B[32], a1 = A[16], b1 = B[16];
  float pa1 = a2, pb1 = b2; COMPOSE(pa1, pb1, a3, b3);
  float pa0 = a1, pb0 = b1; COMPOSE(pa0, pb0, pa1, pb1);
  for (int z = 0; z < P7O_NL; z++) {
    int row = z >> 4;
    float pa = row == 3 ? 1.0f : row == 2 ? a3 : row == 1 ? pa1 : pa0;
    float pb = row == 3 ? 0.0f : row == 2 ? b3 : row == 1 ? pb1 : pb0;
    COMPOSE(A[z], B[z], pa, pb);
  }
}

/* one Forward row.  prev/cur hold M,I,D arrays of Mp floats each. returns xE */
static float fwd_row(const PROF *p, const float *rfx, float xB,
                     const float *Mp_, const float *Ip_, const float *Dp_, float *Mc, float *Ic, float *Dc)
{
  int Q = p->Q, Mp = p->Mp;
  for (int idx = 0; idx < Mp; idx++) {
    float mp = idx ? Mp_[idx-1] : 0.f, ip = idx ? Ip_[idx-1] : 0.f, dp = idx ? Dp_[idx-1] : 0.f;
    float sv = xB * p->fBM[idx];
    sv = sv + mp * p->fMM[idx];
    sv = sv + ip * p->fIM[idx];
    sv = sv + dp * p->fDM[idx];
    Mc[idx] = sv * rfx[idx];
    float a = Mp_[idx] * p->fMI[idx]; float b = Ip_[idx] * p->fII[idx];
    Ic[idx] = a + b;
  }
  /* D chain: D[idx+1] = Mc[idx]*fMD[idx] + fDD[idx]*D[idx]; lane-local affine maps, Kogge-Stone across lanes */
  float A[P7O_NL], B[P7O_NL];
  for (int z = 0; z < P7O_NL; z++) {
    float a = 1.0f, b = 0.0f;
    for (int q = 0; q < Q; q++) { int idx = z*Q+q; float md = Mc[idx] * p->fMD[idx]; float dd = p->fDD[idx]; float t = dd * b; b = md + t; a = dd * a; }
    A[z] = a; B[z] = b;
  }
  lane_scan_up(A, B);
  for (int z = 0; z < P7O_NL; z++) {
    float d = z ? B[z-1] : 0.0f;
    for (int q = 0; q < Q; q++) { int idx = z*Q+q; Dc[idx] = d; float md = Mc[idx] * p->fMD[idx]; float t = p->fDD[idx] * d; d = md + t; }
  }
  /* xE = sum_k M + D: lane partials then xor butterfly */
  float S[P7O_NL], nS[P7O_NL];
  for (int z = 0; z < P7O_NL; z++) { float s = 0.f; for (int q = 0; q < Q; q++) { int idx = z*Q+q; s = s + Mc[idx]; s = s + Dc[idx]; } S[z] = s; }
  for (int w = 1; w <= 32; w <<= 1) { for (int z = 0; z < P7O_NL; z++) nS[z] = S[z] + S[z ^ w]; memcpy(S, nS, sizeof(S)); }
  return S[0];
}

/* Forward over dsq[0..L-1].  If mx != NULL it receives (L+1) rows of 3*Mp floats (row 0 = zeros).
 * xs (optional): (L+1)*6 floats per row: E N J B C scale (post-rescale values).
 * Returns score in nats; *ret_xC scaled xC(L); *ret_nscale count of rescale events. */
static float forward(const PROF *p, const XF *xf, const uint8_t *dsq, int L, float *mx, float *xs, float *ret_xC, int *ret_nscale)
{
  int Mp = p->Mp;
  float *buf = calloc((size_t)6 * Mp, sizeof(float));
  float *pm = buf, *pi = buf + Mp, *pd = buf + 2*Mp, *cm = buf + 3*Mp, *ci = buf + 4*Mp, *cd = buf + 5*Mp;
  float xN = 1.0f, xB = xN * xf->move, xE = 0.f, xJ = 0.f, xC = 0.f, totscale = 0.f; int nscale = 0;
  if (mx) memset(mx, 0, sizeof(float) * 3 * (size_t)Mp);
  if (xs) { xs[0] = 0.f; xs[1] = xN; xs[2] = 0.f; xs[3] = xB; xs[4] = 0.f; xs[5] = 1.0f; }
  for (int i = 1; i <= L; i++) {
    const float *rfx = p->rf + (size_t)dsq[i-1] * Mp;
    xE = fwd_row(p, rfx, xB, pm, pi, pd, cm, ci, cd);
    xN = xN * xf->loop;
    { float a = xC * xf->loop, b = xE * xf->E_move; xC = a + b; }
    { float a = xJ * xf->loop, b = xE * xf->E_loop; xJ = a + b; }
    { float a = xJ * xf->move, b = xN * xf->move; xB = a + b; }
    float scale = 1.0f;
    if (xE > 1.0e4f) {
      float inv = 1.0f / xE;
      xN = xN / xE; xC = xC / xE; xJ = xJ / xE; xB = xB / xE;
      for (int k = 0; k < Mp; k++) { cm[k] *= inv; ci[k] *= inv; cd[k] *= inv; }
      scale = xE; totscale = (float)((double)totscale + log((double)xE)); xE = 1.0f; nscale++;
    }
    if (mx) { float *r = mx + (size_t)i * 3 * Mp; memcpy(r, cm, sizeof(float)*Mp); memcpy(r+Mp, ci, sizeof(float)*Mp); memcpy(r+2*Mp, cd, sizeof(float)*Mp); }
    if (xs) { float *r = xs + (size_t)i * 6; r[0] = xE; r[1] = xN; r[2] = xJ; r[3] = xB; r[4] = xC; r[5] = scale; }
    { float *t; t = pm; pm = cm; cm = t; t = pi; pi = ci; ci = t; t = pd; pd = cd; cd = t; }
  }
  free(buf);
  if (ret_xC) *ret_xC = xC; if (ret_nscale) *ret_nscale = nscale;
  return (float)((double)totscale + log((double)(xC * xf->move)));
}

/* one Backward row (row i from row i+1), canonical order. */
static void bwd_row(const PROF *p, const XF *xf, const float *rfx /* residue i+1 */,
                    const float *Mn, const float *In, float *xspec /* in: J C N of row i+1 ; out: E N J B C of row i */,
                    float *Mc, float *Ic, float *Dc, float *mnext /* scratch Mp */)
{
  int Q = p->Q, Mp = p->Mp;
  for (int idx = 0; idx < Mp; idx++) mnext[idx] = Mn[idx] * rfx[idx];
  float S[P7O_NL], nS[P7O_NL];
  for (int z = 0; z < P7O_NL; z++) { float s = 0.f; for (int q = 0; q < Q; q++) { int idx = z*Q+q; float t = p->fBM[idx] * mnext[idx]; s = s + t; } S[z] = s; }
  for (int w = 1; w <= 32; w <<= 1) { for (int z = 0; z < P7O_NL; z++) nS[z] = S[z] + S[z ^ w]; memcpy(S, nS, sizeof(S)); }
  float xB = S[0];
  float xJn = xspec[2], xCn = xspec[4], xNn = xspec[1];
  float xJ, xC, xE, xN;
  { float a = xB * xf->move, b = xJn * xf->loop; xJ = a + b; }
  xC = xCn * xf->loop;
  { float a = xC * xf->E_move, b = xJ * xf->E_loop; xE = a + b; }
  { float a = xB * xf->move, b = xNn * xf->loop; xN = a + b; }
  xspec[0] = xE; xspec[1] = xN; xspec[2] = xJ; xspec[3] = xB; xspec[4] = xC;
  /* D chain, reverse: D[idx] = (xE + fDM[idx+1]*mnext[idx+1]) + fDD[idx]*D[idx+1] */
  float A[P7O_NL], B[P7O_NL];
#define AVAL(idx) (xE + (((idx)+1 < Mp) ? p->fDM[(idx)+1] * mnext[(idx)+1] : 0.0f))
  for (int z = 0; z < P7O_NL; z++) {
    float a = 1.0f, b = 0.0f;
    for (int q = Q-1; q >= 0; q--) { int idx = z*Q+q; float av = AVAL(idx); float dd = p->fDD[idx]; float t = dd * b; b = av + t; a = dd * a; }
    A[z] = a; B[z] = b;
  }
  lane_scan_down(A, B);
  for (int z = 0; z < P7O_NL; z++) {
    float d = (z < P7O_NL-1) ? B[z+1] : 0.0f;
    for (int q = Q-1; q >= 0; q--) { int idx = z*Q+q; float av = AVAL(idx); float t = p->fDD[idx] * d; d = av + t; Dc[idx] = d; }
  }
#undef AVAL
  for (int idx = 0; idx < Mp; idx++) {
    float mn1 = (idx+1 < Mp) ? mnext[idx+1] : 0.0f;
    float dn1 = (idx+1 < Mp) ? Dc[idx+1] : 0.0f;
    float tim = (idx+1 < Mp) ? p->fIM[idx+1] : 0.0f, tmm = (idx+1 < Mp) ? p->fMM[idx+1] : 0.0f;
    { float a = tim * mn1, b = p->fII[idx] * In[idx]; Ic[idx] = a + b; }
    float m = xE + tmm * mn1;
    m = m + p->fMI[idx] * In[idx];
    m = m + p->fMD[idx] * dn1;
    Mc[idx] = m;
  }
}

/* Backward over dsq[0..L-1] using the Forward scale factors fscale[i] (i=1..L; xs rows' [5]).
 * mx (optional) gets (L+1) rows of 3*Mp; xb gets (L+1)*5 rows E N J B C. */
static void backward(const PROF *p, const XF *xf, const uint8_t *dsq, int L, const float *fxs, float *mx, float *xb)
{
  int Mp = p->Mp;
  float *buf = calloc((size_t)7 * Mp, sizeof(float));
  float *nm = buf, *ni = buf + Mp, *nd = buf + 2*Mp, *cm = buf + 3*Mp, *ci = buf + 4*Mp, *cd = buf + 5*Mp, *scr = buf + 6*Mp;
  float sp[5];
  /* row L */
  float xC = xf->move, xE = xC * xf->E_move;
  sp[0] = xE; sp[1] = 0.f; sp[2] = 0.f; sp[3] = 0.f; sp[4] = xC;
  { /* M(L,k) = xE + fMD*D(L,k+1); D(L,k) = xE + fDD*D(L,k+1): same recurrences with mnext = 0, In = 0 */
    float zero_rf_dummy = 0.f; (void)zero_rf_dummy;
    int Q = p->Q;
    float A[P7O_NL], B[P7O_NL];
    for (int z = 0; z < P7O_NL; z++) { float a = 1.f, b = 0.f; for (int q = Q-1; q >= 0; q--) { int idx = z*Q+q; float av = xE + 0.0f; float dd = p->fDD[idx]; float t = dd * b; b = av + t; a = dd * a; } A[z] = a; B[z] = b; }
    lane_scan_down(A, B);
    for (int z = 0; z < P7O_NL; z++) { float d = (z < P7O_NL-1) ? B[z+1] : 0.f; for (int q = Q-1; q >= 0; q--) { int idx = z*Q+q; float av = xE + 0.0f; float t = p->fDD[idx] * d; d = av + t; nd[idx] = d; } }
    for (int idx = 0; idx < Mp; idx++) { float dn1 = (idx+1 < Mp) ? nd[idx+1] : 0.f; float m = xE + 0.0f; m = m + 0.0f; m = m + p->fMD[idx] * dn1; nm[idx] = m; ni[idx] = 0.f; }
  }
  if (mx) { float *r = mx + (size_t)L * 3 * Mp; memcpy(r, nm, sizeof(float)*Mp); memcpy(r+Mp, ni, sizeof(float)*Mp); memcpy(r+2*Mp, nd, sizeof(float)*Mp); }
  memcpy(xb + (size_t)L*5, sp, sizeof(sp));
  for (int i = L-1; i >= 0; i--) {
    const float *rfx = p->rf + (size_t)dsq[i] * Mp;     /* residue i+1 */
    bwd_row(p, xf, rfx, nm, ni, sp, cm, ci, cd, scr);
    float sc = fxs[(size_t)(i+1)*6 + 5];
    if (sc != 1.0f) {
      float inv = 1.0f / sc;
      for (int k = 0; k < Mp; k++) { cm[k] *= inv; ci[k] *= inv; cd[k] *= inv; }
      for (int s = 0; s < 5; s++) sp[s] *= inv;
    }
    if (mx) { float *r = mx + (size_t)i * 3 * Mp; memcpy(r, cm, sizeof(float)*Mp); memcpy(r+Mp, ci, sizeof(float)*Mp); memcpy(r+2*Mp, cd, sizeof(float)*Mp); }
    memcpy(xb + (size_t)i*5, sp, sizeof(sp));
    { float *t; t = nm; nm = cm; cm = t; t = ni; ni = ci; ci = t; t = nd; nd = cd; cd = t; }
  }
  free(buf);
}

/* ------------------------------------------------------------------------------------------
 * Envelope rescoring: unihit Forward/Backward, decoding, null2 by expectation, OA fill + trace
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int ienv, jenv;
  float envsc, oasc, domcorrection;
  int hmm_from, hmm_to, ali_from, ali_to;
  int ok;
  /* filled by the pipeline */
  float dombias, bitscore; double lnP; int is_reported;
} DOMAIN;

/* null2 odds of the 20 residues from state usage (match me[], insert ie[], flanks xfactor), canonical order */
static void null2_from_usage(const PROF *p, const float *me, const float *ie, float xfactor, float *null2)
{
  int Q = p->Q, Mp = p->Mp;
  for (int x = 0; x < 20; x++) {
    const float *rfx = p->rf + (size_t)x * Mp;
    float S[P7O_NL], nS[P7O_NL];
    for (int z = 0; z < P7O_NL; z++) { float s = 0.f; for (int q = 0; q < Q; q++) { int idx = z*Q+q; float t = me[idx] * rfx[idx]; s = s + t; s = s + ie[idx]; } S[z] = s; }
    for (int w = 1; w <= 32; w <<= 1) { for (int z = 0; z < P7O_NL; z++) nS[z] = S[z] + S[z ^ w]; memcpy(S, nS, sizeof(S)); }
    null2[x] = S[0] + xfactor;
  }
}

static void null2_fill_degenerate(float *null2)
{
  for (int x = 21; x <= 26; x++) { float r = 0.f; int n = 0; for (int y = 0; y < 20; y++) if (degen(x, y)) { r += null2[y]; n++; } null2[x] = r / (float)n; }
  null2[20] = null2[27] = null2[28] = 1.0f;
}

static int32_t *g_oa_path = NULL;    /* when set: receives, per model node (0-based), the envelope-local residue its match state emits (0 = none) */
static float *g_oa_pp = NULL;        /* when set: receives, per envelope-local residue on the path, its posterior probability in the emitting state (M or I):
                                      * what hmmsearch's OA trace carries as tr->pp and prints as the PP line (p7_OATrace, p7_alidisplay_Create) */

static int rescore_envelope(const PROF *p, const uint8_t *dsq_full, int L_full, int ienv, int jenv,
                            float *n2sc /* per-position, 1-based, may be NULL */, int null2_done, DOMAIN *dom,
                            float *out_null2, float *out_xC, int *out_nscale)
{
  int Mp = p->Mp, M = p->M, Ld = jenv - ienv + 1;
  const uint8_t *dsq = dsq_full + (ienv - 1);
  XF xf; xf_config(p, L_full, 0, &xf);
  size_t rowsz = (size_t)3 * Mp;
  float *fmx = malloc(sizeof(float) * rowsz * (Ld+1)), *bmx = malloc(sizeof(float) * rowsz * (Ld+1));
  float *fxs = malloc(sizeof(float) * 6 * (Ld+1)), *bxs = malloc(sizeof(float) * 5 * (Ld+1));
  float xCL; int nscale;
  float envsc = forward(p, &xf, dsq, Ld, fmx, fxs, &xCL, &nscale);
  backward(p, &xf, dsq, Ld, fxs, bmx, bxs);
  if (out_xC) *out_xC = xCL; if (out_nscale) *out_nscale = nscale;
  /* decoding: pp overwrites bmx rows 1..Ld (M,I; D zeroed); specials N J C per row */
  float Zs = xCL * xf.move; float invZ = 1.0f / Zs;
  float *ppN = calloc((size_t)Ld+1, sizeof(float)), *ppJ = calloc((size_t)Ld+1, sizeof(float)), *ppC = calloc((size_t)Ld+1, sizeof(float));
  int range_err = 0;
  for (int i = 1; i <= Ld; i++) {
    float *f = fmx + rowsz * i, *b = bmx + rowsz * i;
    for (int k = 0; k < Mp; k++) {
      float pm = f[k] * b[k]; pm = pm * invZ; float pi = f[Mp+k] * b[Mp+k]; pi = pi * invZ;
      b[k] = pm; b[Mp+k] = pi; b[2*Mp+k] = 0.f;
      if (!isfinite(pm) || !isfinite(pi)) range_err = 1;
    }
    float w = invZ / fxs[(size_t)i*6+5];
    { float t = fxs[(size_t)(i-1)*6+1] * bxs[(size_t)i*5+1]; t = t * xf.loop; ppN[i] = t * w; }
    { float t = fxs[(size_t)(i-1)*6+2] * bxs[(size_t)i*5+2]; t = t * xf.loop; ppJ[i] = t * w; }
    { float t = fxs[(size_t)(i-1)*6+4] * bxs[(size_t)i*5+4]; t = t * xf.loop; ppC[i] = t * w; }
  }
  dom->ok = !range_err; dom->ienv = ienv; dom->jenv = jenv; dom->envsc = envsc;
  if (range_err) { free(fmx); free(bmx); free(fxs); free(bxs); free(ppN); free(ppJ); free(ppC); return -1; }
  /* null2 by expectation */
  float null2[P7O_KP];
  if (!null2_done) {
    float *me = calloc((size_t)2*Mp, sizeof(float)), *ie = me + Mp; float xN = 0.f, xJ = 0.f, xC = 0.f;
    for (int i = 1; i <= Ld; i++) { float *b = bmx + rowsz * i; for (int k = 0; k < Mp; k++) { me[k] = me[k] + b[k]; ie[k] = ie[k] + b[Mp+k]; } xN = xN + ppN[i]; xJ = xJ + ppJ[i]; xC = xC + ppC[i]; }
    float norm = 1.0f / (float)Ld;
    for (int k = 0; k < Mp; k++) { me[k] *= norm; ie[k] *= norm; }
    float xfactor = ((xN + xC) + xJ) * norm;
    null2_from_usage(p, me, ie, xfactor, null2);
    free(me);
    if (out_null2) memcpy(out_null2, null2, sizeof(float)*20);
    null2_fill_degenerate(null2);
  }
  float domcorrection = 0.f;
  if (null2_done) { for (int pos = ienv; pos <= jenv; pos++) domcorrection += n2sc[pos]; }     /* set by the trace ensemble */
  else for (int pos = ienv; pos <= jenv; pos++) { float v = logf(null2[dsq_full[pos-1]]); if (n2sc) n2sc[pos] = v; domcorrection += v; }
  dom->domcorrection = domcorrection;
  /* optimal accuracy fill (DEV2: -inf gating); matrix overwrites fmx */
  {
    float *oN = malloc(sizeof(float) * 5 * (Ld+1)), *oB = oN + (Ld+1), *oE = oB + (Ld+1), *oJ = oE + (Ld+1), *oC = oJ + (Ld+1);
    oN[0] = 0.f; oB[0] = 0.f; oE[0] = NEGINF; oJ[0] = NEGINF; oC[0] = NEGINF;
    float *r0 = fmx; for (size_t k = 0; k < rowsz; k++) r0[k] = NEGINF;
    int Eloop_ok = xf.E_loop > 0.f;
    for (int i = 1; i <= Ld; i++) {
      float *pr = fmx + rowsz*(i-1), *cr = fmx + rowsz*i, *pp = bmx + rowsz*i;
      float e = NEGINF;
      for (int idx = 0; idx < Mp; idx++) {
        float best = NEGINF;
        if (idx < M) {
          if (idx > 0) {
            if (p->fMM[idx] > 0.f && pr[idx-1] > best) best = pr[idx-1];
            if (p->fIM[idx] > 0.f && pr[Mp+idx-1] > best) best = pr[Mp+idx-1];
            if (p->fDM[idx] > 0.f && pr[2*Mp+idx-1] > best) best = pr[2*Mp+idx-1];
          }
          if (p->fBM[idx] > 0.f && oB[i-1] > best) best = oB[i-1];
          cr[idx] = best + pp[idx];
          float bi = NEGINF;
          if (p->fMI[idx] > 0.f && pr[idx] > bi) bi = pr[idx];
          if (p->fII[idx] > 0.f && pr[Mp+idx] > bi) bi = pr[Mp+idx];
          cr[Mp+idx] = bi + pp[Mp+idx];
          if (cr[idx] > e) e = cr[idx];
        } else { cr[idx] = NEGINF; cr[Mp+idx] = NEGINF; }
      }
      cr[2*Mp] = NEGINF;
      for (int idx = 1; idx < Mp; idx++) {
        float d = NEGINF;
        if (idx < M) { if (p->fMD[idx-1] > 0.f && cr[idx-1] > d) d = cr[idx-1]; if (p->fDD[idx-1] > 0.f && cr[2*Mp+idx-1] > d) d = cr[2*Mp+idx-1]; }
        cr[2*Mp+idx] = d;
      }
      oE[i] = e;
      { float a = oJ[i-1] + ppJ[i]; float b = Eloop_ok ? e : NEGINF; oJ[i] = a > b ? a : b; }
      { float a = oC[i-1] + ppC[i]; oC[i] = a > e ? a : e; }
      oN[i] = oN[i-1] + ppN[i];
      { float a = oN[i], b = oJ[i]; oB[i] = a > b ? a : b; }
    }
    dom->oasc = oC[Ld];
    /* traceback: C(Ld) back to the first match state */
    int i = Ld, k = 0, st = 0;   /* 0=C 1=E 2=M 3=I 4=D */
    int firstM_i = 0, firstM_k = 0, lastM_i = 0, lastM_k = 0, done = 0;
    while (!done) {
      float *cr = fmx + rowsz*i, *pr = (i > 0) ? fmx + rowsz*(i-1) : NULL;
      switch (st) {
      case 0: { if (i == 0) { done = 1; break; } float a = oC[i-1] + ppC[i], b = oE[i]; if (a >= b) i--; else st = 1; } break;
      case 1: { k = -1; for (int idx = 0; idx < M; idx++) if (cr[idx] == oE[i]) { k = idx; break; } if (k < 0) { done = 1; break; } st = 2; lastM_i = i; lastM_k = k+1; } break;
      case 2: {
        firstM_i = i; firstM_k = k+1;
        if (g_oa_path) g_oa_path[k] = i;
        if (g_oa_pp) g_oa_pp[i] = bmx[rowsz*i + k];
        float path[4] = { NEGINF, NEGINF, NEGINF, NEGINF };
        if (k > 0) { if (p->fMM[k] > 0.f) path[0] = pr[k-1]; if (p->fIM[k] > 0.f) path[1] = pr[Mp+k-1]; if (p->fDM[k] > 0.f) path[2] = pr[2*Mp+k-1]; }
        if (p->fBM[k] > 0.f) path[3] = oB[i-1];
        int best = 0; for (int c = 1; c < 4; c++) if (path[c] > path[best]) best = c;
        i--;
        if (best == 0) { k--; st = 2; } else if (best == 1) { k--; st = 3; } else if (best == 2) { k--; st = 4; } else done = 1;
      } break;
      case 3: { if (g_oa_pp) g_oa_pp[i] = bmx[rowsz*i + Mp + k]; float a = (p->fMI[k] > 0.f) ? pr[k] : NEGINF, b = (p->fII[k] > 0.f) ? pr[Mp+k] : NEGINF; i--; st = (a >= b) ? 2 : 3; } break;
      case 4: { float a = (p->fMD[k-1] > 0.f) ? cr[k-1] : NEGINF, b = (p->fDD[k-1] > 0.f) ? cr[2*Mp+k-1] : NEGINF; k--; st = (a >= b) ? 2 : 4; } break;
      }
    }
    dom->hmm_from = firstM_k; dom->hmm_to = lastM_k;
    dom->ali_from = firstM_i + ienv - 1; dom->ali_to = lastM_i + ienv - 1;
    free(oN);
  }
  free(fmx); free(bmx); free(fxs); free(bxs); free(ppN); free(ppJ); free(ppC);
  return 0;
}

int p7o_envelope(const P7O_HMM *hmm, const uint8_t *dsq, int L_full, int ienv, int jenv,
                 float *envsc, float *oasc, float *null2, int32_t *coords, float *fwd_xC, int32_t *nscale)
{
  PROF *p = prof_create(hmm); DOMAIN d; int ns;
  int rc = rescore_envelope(p, dsq, L_full, ienv, jenv, NULL, 0, &d, null2, fwd_xC, &ns);
  *envsc = d.envsc; *oasc = d.oasc; coords[0] = d.hmm_from; coords[1] = d.hmm_to; coords[2] = d.ali_from; coords[3] = d.ali_to;
  if (nscale) *nscale = ns;
  prof_free(p); return rc;
}

/* The alignment hmmsearch prints for a domain (hmmsearch without --noali: checkm/markerGeneFinder.py:138-142 under bKeepAlignment): the
 * optimal-accuracy path of the envelope ienv..jenv and the posterior probability of every residue on it.  path[k] (k = 0..M-1) = residue
 * (1-based within the envelope) emitted by match state k+1, 0 = none; pp[i] (i = 1..Ld, pp[0] unused) = posterior probability of residue i
 * in the state that emits it on the path, 0 off the path.  Not thread-safe (file-scope hooks into the traceback). */
int p7o_envelope_alignment(const P7O_HMM *hmm, const uint8_t *dsq, int L_full, int ienv, int jenv, int32_t *path, float *pp)
{
  PROF *p = prof_create(hmm); DOMAIN d;
  memset(path, 0, sizeof(int32_t) * hmm->M);
  memset(pp, 0, sizeof(float) * (size_t)(jenv - ienv + 2));
  g_oa_path = path; g_oa_pp = pp;
  int rc = rescore_envelope(p, dsq, L_full, ienv, jenv, NULL, 0, &d, NULL, NULL, NULL);
  g_oa_path = NULL; g_oa_pp = NULL;
  prof_free(p);
  return rc;
}

/* hmmalign's per-sequence computation (checkm/hmmer.py:76-95 runs `hmmalign --outformat Pfam`): unihit local profile with the
 * length model of the sequence itself, Forward/Backward/decoding, optimal-accuracy alignment of the WHOLE sequence.  path[k]
 * (k = 0..M-1) = 1-based residue emitted by match state k+1, 0 if the node is deleted or outside the local alignment -- all
 * that HmmerAligner._maskAlignment keeps of the alignment (checkm/hmmerAligner.py:325-352: the '#=GC RF' x columns).
 * Not thread-safe (uses a file-scope hook into the traceback). */
int p7o_align(const P7O_HMM *hmm, const uint8_t *dsq, int L, int32_t *path)
{
  PROF *p = prof_create(hmm); DOMAIN d;
  memset(path, 0, sizeof(int32_t) * hmm->M);
  if (L < 1) { prof_free(p); return -1; }
  g_oa_path = path;
  int rc = rescore_envelope(p, dsq, L, 1, L, NULL, 0, &d, NULL, NULL, NULL);
  g_oa_path = NULL;
  prof_free(p);
  return rc;
}

/* ------------------------------------------------------------------------------------------
 * Multi-domain regions: stochastic trace ensemble + single-linkage clustering (see DEV3)
 * ------------------------------------------------------------------------------------------ */
#define ENS_NSAMPLES 200
#define ENS_STRIDE   15485863u
#define ENS_MIN_OVERLAP   0.8f
#define ENS_MAX_DIAGDIFF  4
#define ENS_MIN_POSTERIOR 0.25f
#define ENS_MIN_ENDPOINTP 0.02f

static uint32_t mix3(uint32_t a, uint32_t b, uint32_t c)
{
  a -= b; a -= c; a ^= (c >> 13);  b -= c; b -= a; b ^= (a << 8);   c -= a; c -= b; c ^= (b >> 13);
  a -= b; a -= c; a ^= (c >> 12);  b -= c; b -= a; b ^= (a << 16);  c -= a; c -= b; c ^= (b >> 5);
  a -= b; a -= c; a ^= (c >> 3);   b -= c; b -= a; b ^= (a << 10);  c -= a; c -= b; c ^= (b >> 15);
  return c;
}
/* state of x -> x*69069+1 after n steps */
static uint32_t lcg_jump(uint32_t x, uint64_t n)
{
  uint32_t A = 69069u, C = 1u, ra = 1u, rc = 0u;
  while (n) { if (n & 1) { ra = A * ra; rc = A * rc + C; } C = A * C + C; A = A * A; n >>= 1; }
  return ra * x + rc;
}
static int g_ens_sequential = 1;     /* one stream per region (hmmsearch's own use of its generator); 0: one sub-stream per trace */
void p7o_set_ensemble_stream(int sequential) { g_ens_sequential = sequential; }

uint32_t p7o_ensemble_seed(int t)
{
  uint32_t x0 = mix3(42u, 87654321u, 12345678u); if (x0 == 0) x0 = 42u;
  return lcg_jump(x0, (uint64_t)t * ENS_STRIDE);
}
static inline double roll_next(uint32_t *x) { *x = *x * 69069u + 1u; return (double)*x / 4294967296.0; }

/* first index whose cumulative weight exceeds roll * total; weights summed in index order */
static int choose(double roll, const float *pth, int n)
{
  float norm = pth[0]; for (int i = 1; i < n; i++) norm = norm + pth[i];
  if (!(norm > 0.0f)) return 0;
  double target = roll * (double)norm, sum = 0.0;
  for (int i = 0; i < n; i++) { sum += (double)pth[i]; if (target < sum) return i; }
  for (int i = n-1; i > 0; i--) if (pth[i] > 0.0f) return i;
  return 0;
}

/* One stochastic traceback of the region's Forward matrix (rows 0..Ld of 3*Mp, specials xs rows of 6).
 * code[i], i=1..Ld: 0 = residue emitted outside a domain (N/J/C); 0x4000|k match state k; 0x8000|k insert state k;
 * bit 0x2000... not used; the first match state of each domain is additionally recorded in seg[].
 * Segments are produced last-domain-first; returns their number (or -1 if more than cap). */
static int stochastic_trace(const PROF *p, const XF *xf, int Ld, const float *mx, const float *xs, uint32_t *rng,
                            uint16_t *code, P7O_SEG *seg, int cap)
{
  int Mp = p->Mp; size_t rowsz = (size_t)3 * Mp;
  enum { sC, sE, sM, sI, sD, sB, sJ, sN } st = sC;
  int i = Ld, k = 0, nseg = 0, sqto = 0, hmmto = 0;
  float pth[4];
  for (;;) {
    const float *cr = mx + rowsz * i, *pr = (i > 0) ? mx + rowsz * (i-1) : mx;
    switch (st) {
    case sC:
      pth[0] = xs[(size_t)(i-1)*6+4] * xf->loop;
      pth[1] = (xs[(size_t)i*6+0] * xf->E_move) * xs[(size_t)i*6+5];
      if (choose(roll_next(rng), pth, 2) == 0) { code[i] = 0; i--; } else st = sE;
      break;
    case sJ:
      pth[0] = xs[(size_t)(i-1)*6+2] * xf->loop;
      pth[1] = (xs[(size_t)i*6+0] * xf->E_loop) * xs[(size_t)i*6+5];
      if (choose(roll_next(rng), pth, 2) == 0) { code[i] = 0; i--; } else st = sE;
      break;
    case sE: {
      /* any M(i,k), D(i,k).  Canonical order: lane z owns cells z*Q..z*Q+Q-1; weights in cell order are M(c), D(c);
       * local[z] = sequential double sum of the lane's weights, base[z] = local[0]+..+local[z-1] in lane order,
       * total = base[64]; first position whose cumulative weight base[z] + running sum exceeds roll*total. */
      int Q = p->Q; double local[P7O_NL], base[P7O_NL], run = 0.0;
      for (int z = 0; z < P7O_NL; z++) { double s = 0.0; for (int q = 0; q < Q; q++) { int c = z*Q+q; s += (double)cr[c]; s += (double)cr[2*Mp+c]; } local[z] = s; }
      for (int z = 0; z < P7O_NL; z++) { base[z] = run; run += local[z]; }
      double target = roll_next(rng) * run; int pick = 0, isd = 0, hit = 0;
      for (int z = 0; z < P7O_NL && !hit; z++) {
        double acc = 0.0;
        for (int q = 0; q < Q && !hit; q++) {
          int c = z*Q+q;
          acc += (double)cr[c];      if (target < base[z] + acc) { pick = c; isd = 0; hit = 1; break; }
          acc += (double)cr[2*Mp+c]; if (target < base[z] + acc) { pick = c; isd = 1; hit = 1; break; }
        }
      }
      /* HMMER's p7_trace_Index sets sqto AND hmmto in its match-state case only: the domain's last model node is the last MATCH
       * state of the trace (the first one met on the way back); delete states between it and E do not count */
      k = pick + 1; st = isd ? sD : sM; sqto = 0; hmmto = 0;
    } break;
    case sM: {
      int c = k - 1;
      code[i] = (uint16_t)(0x4000 | k);
      if (!sqto) { sqto = i; hmmto = k; }
      pth[0] = xs[(size_t)(i-1)*6+3] * p->fBM[c];
      if (c > 0) { pth[1] = pr[c-1] * p->fMM[c]; pth[2] = pr[Mp+c-1] * p->fIM[c]; pth[3] = pr[2*Mp+c-1] * p->fDM[c]; }
      else pth[1] = pth[2] = pth[3] = 0.0f;
      int ch = choose(roll_next(rng), pth, 4);
      if (ch == 0) {
        if (nseg == cap) return -1;
        seg[nseg].sqfrom = i; seg[nseg].sqto = sqto; seg[nseg].hmmfrom = k; seg[nseg].hmmto = hmmto; nseg++;
        st = sB;
      } else st = (ch == 1) ? sM : (ch == 2) ? sI : sD;
      i--; k--;
    } break;
    case sI: {
      int c = k - 1;
      code[i] = (uint16_t)(0x8000 | k);
      pth[0] = pr[c] * p->fMI[c]; pth[1] = pr[Mp+c] * p->fII[c];
      st = (choose(roll_next(rng), pth, 2) == 0) ? sM : sI;
      i--;
    } break;
    case sD: {
      int c = k - 1;
      if (c > 0) { pth[0] = cr[c-1] * p->fMD[c-1]; pth[1] = cr[2*Mp+c-1] * p->fDD[c-1]; } else pth[0] = pth[1] = 0.0f;
      st = (choose(roll_next(rng), pth, 2) == 0) ? sM : sD;
      k--;
    } break;
    case sB:
      pth[0] = xs[(size_t)i*6+1] * xf->move; pth[1] = xs[(size_t)i*6+2] * xf->move;
      st = (choose(roll_next(rng), pth, 2) == 0) ? sN : sJ;
      break;
    case sN:
      for (; i >= 1; i--) code[i] = 0;
      return nseg;
    }
    if (i < 0 || k < 0 || (st == sM && (k < 1 || i < 1)) || (st == sI && (k < 1 || i < 1)) || (st == sD && k < 1) ||
        ((st == sC || st == sJ || st == sE) && i < 1)) {
      /* a numerically impossible move (all candidate paths zero): end the trace here */
      for (; i >= 1; i--) code[i] = 0;
      return nseg;
    }
  }
}

/* n2sum[pos-ireg] (pos = ireg..jreg) = sum over traces of the null2 odds ratio; seg_all/nseg_all: every trace's
 * segments in region-local coordinates, first-domain-first. */
static int trace_ensemble(const PROF *p, const uint8_t *dsq, int L, int ireg, int jreg,
                          float *n2sum, P7O_SEG *seg_all, int *nseg_all, int cap)
{
  int Mp = p->Mp, Ld = jreg - ireg + 1; size_t rowsz = (size_t)3 * Mp;
  const uint8_t *rd = dsq + (ireg - 1);
  XF xf; xf_config(p, L, 1, &xf);
  float *mx = malloc(sizeof(float) * rowsz * (Ld+1)), *xs = malloc(sizeof(float) * 6 * (Ld+1));
  forward(p, &xf, rd, Ld, mx, xs, NULL, NULL);
  uint16_t *code = malloc(sizeof(uint16_t) * (Ld+2));
  float *ratio = malloc(sizeof(float) * (size_t)ENS_NSAMPLES * (Ld+1));
  float *cm = malloc(sizeof(float) * 2 * Mp), *ci = cm + Mp;
  int rc = 0;
  /* default: ONE stream per region, carried from trace to trace; p7o_set_ensemble_stream(0): every trace has its own sub-stream
   * trace to trace as HMMER carries its generator */
  uint32_t stream_rng = p7o_ensemble_seed(0);
  for (int t = 0; t < ENS_NSAMPLES && rc == 0; t++) {
    uint32_t rng = g_ens_sequential ? stream_rng : p7o_ensemble_seed(t);
    P7O_SEG *seg = seg_all + (size_t)t * cap;
    int ns = stochastic_trace(p, &xf, Ld, mx, xs, &rng, code, seg, cap);
    stream_rng = rng;
    if (ns < 0) { rc = -1; break; }
    for (int a = 0, b = ns-1; a < b; a++, b--) { P7O_SEG tmp = seg[a]; seg[a] = seg[b]; seg[b] = tmp; }
    nseg_all[t] = ns;
    float *rt = ratio + (size_t)t * (Ld+1);
    for (int pos = 1; pos <= Ld; pos++) rt[pos] = 1.0f;
    for (int d = 0; d < ns; d++) {
      float null2[P7O_KP]; int nemit = 0;
      memset(cm, 0, sizeof(float) * 2 * Mp);
      for (int pos = seg[d].sqfrom; pos <= seg[d].sqto; pos++) {
        int kk = code[pos] & 0x3fff;
        if (code[pos] & 0x4000) { cm[kk-1] += 1.0f; nemit++; } else if (code[pos] & 0x8000) { ci[kk-1] += 1.0f; nemit++; }
      }
      float norm = 1.0f / (float)nemit;
      for (int c = 0; c < Mp; c++) { cm[c] *= norm; ci[c] *= norm; }
      null2_from_usage(p, cm, ci, 0.0f, null2);
      null2_fill_degenerate(null2);
      /* the first residue of a domain keeps ratio 1 (HMMER's loops use pos <= sqfrom for the flank) */
      for (int pos = seg[d].sqfrom + 1; pos <= seg[d].sqto; pos++) rt[pos] = null2[rd[pos-1]];
    }
  }
  if (rc == 0) for (int pos = 1; pos <= Ld; pos++) {
    float acc = 0.0f;
    for (int t = 0; t < ENS_NSAMPLES; t++) acc = acc + ratio[(size_t)t * (Ld+1) + pos];
    n2sum[pos-1] = acc;
  }
  free(mx); free(xs); free(code); free(ratio); free(cm);
  return rc;
}

/* single-linkage clustering of sampled segments -> envelopes sorted by start; returns their number */
static int seg_linked(const P7O_SEG *a, const P7O_SEG *b)
{
  int nov = (a->sqto < b->sqto ? a->sqto : b->sqto) - (a->sqfrom > b->sqfrom ? a->sqfrom : b->sqfrom) + 1;
  int la = a->sqto - a->sqfrom + 1, lb = b->sqto - b->sqfrom + 1, n = la < lb ? la : lb;
  if ((float)nov / (float)n < ENS_MIN_OVERLAP) return 0;
  nov = (a->hmmto < b->hmmto ? a->hmmto : b->hmmto) - (a->hmmfrom > b->hmmfrom ? a->hmmfrom : b->hmmfrom) + 1;
  la = a->hmmto - a->hmmfrom + 1; lb = b->hmmto - b->hmmfrom + 1; n = la < lb ? la : lb;
  if ((float)nov / (float)n < ENS_MIN_OVERLAP) return 0;
  int d1 = (a->sqfrom - a->hmmfrom + a->sqto - a->hmmto) / 2, d2 = (b->sqfrom - b->hmmfrom + b->sqto - b->hmmto) / 2;
  if (abs(d1 - d2) > ENS_MAX_DIAGDIFF) return 0;
  return 1;
}

static int cluster_ensemble(const P7O_SEG *seg_all, const int *nseg_all, int cap, P7O_SEG *env, int envcap)
{
  int n = 0;
  for (int t = 0; t < ENS_NSAMPLES; t++) n += nseg_all[t];
  if (!n) return 0;
  P7O_SEG *sg = malloc(sizeof(P7O_SEG) * n); int *tr = malloc(sizeof(int) * n * 3), *asg = tr + n, *stack = asg + n;
  { int h = 0; for (int t = 0; t < ENS_NSAMPLES; t++) for (int d = 0; d < nseg_all[t]; d++) { sg[h] = seg_all[(size_t)t * cap + d]; tr[h] = t; h++; } }
  for (int h = 0; h < n; h++) asg[h] = -1;
  int nc = 0;
  for (int h = 0; h < n; h++) if (asg[h] < 0) {
    int sp = 0; stack[sp++] = h; asg[h] = nc;
    while (sp) { int a = stack[--sp]; for (int b = 0; b < n; b++) if (asg[b] < 0 && seg_linked(&sg[a], &sg[b])) { asg[b] = nc; stack[sp++] = b; } }
    nc++;
  }
  int nenv = 0;
  for (int c = 0; c < nc; c++) {
    int ninc = 0, lastt = -1;
    for (int h = 0; h < n; h++) if (asg[h] == c && tr[h] != lastt) { ninc++; lastt = tr[h]; }   /* traces are contiguous in sg[] */
    if ((float)ninc / (float)ENS_NSAMPLES < ENS_MIN_POSTERIOR) continue;
    int lim[4][2];   /* min,max of sqfrom, sqto, hmmfrom, hmmto */
    for (int f = 0; f < 4; f++) { lim[f][0] = 1 << 30; lim[f][1] = -1; }
    for (int h = 0; h < n; h++) if (asg[h] == c) {
      int v[4] = { sg[h].sqfrom, sg[h].sqto, sg[h].hmmfrom, sg[h].hmmto };
      for (int f = 0; f < 4; f++) { if (v[f] < lim[f][0]) lim[f][0] = v[f]; if (v[f] > lim[f][1]) lim[f][1] = v[f]; }
    }
    int best[4];
    for (int f = 0; f < 4; f++) {
      int span = lim[f][1] - lim[f][0] + 1; int *epc = calloc(span, sizeof(int));
      for (int h = 0; h < n; h++) if (asg[h] == c) { int v = (f == 0) ? sg[h].sqfrom : (f == 1) ? sg[h].sqto : (f == 2) ? sg[h].hmmfrom : sg[h].hmmto; epc[v - lim[f][0]]++; }
      int b;
      if (f == 0 || f == 2) { for (b = lim[f][0]; b < lim[f][1]; b++) if ((float)epc[b - lim[f][0]] / (float)ninc >= ENS_MIN_ENDPOINTP) break; }
      else                  { for (b = lim[f][1]; b > lim[f][0]; b--) if ((float)epc[b - lim[f][0]] / (float)ninc >= ENS_MIN_ENDPOINTP) break; }
      best[f] = b; free(epc);
    }
    if (nenv < envcap) { env[nenv].sqfrom = best[0]; env[nenv].sqto = best[1]; env[nenv].hmmfrom = best[2]; env[nenv].hmmto = best[3]; nenv++; }
  }
  /* order by start (then end): insertion sort, clusters are few */
  for (int a = 1; a < nenv; a++) { P7O_SEG v = env[a]; int b = a - 1; while (b >= 0 && (env[b].sqfrom > v.sqfrom || (env[b].sqfrom == v.sqfrom && env[b].sqto > v.sqto))) { env[b+1] = env[b]; b--; } env[b+1] = v; }
  free(sg); free(tr);
  return nenv;
}

int p7o_region_ensemble(const P7O_HMM *hmm, const uint8_t *dsq, int L, int ireg, int jreg,
                        float *n2sum, P7O_SEG *seg_all, int32_t *nseg_all, int cap, P7O_SEG *env, int envcap, int32_t *nenv)
{
  PROF *p = prof_create(hmm);
  int rc = trace_ensemble(p, dsq, L, ireg, jreg, n2sum, seg_all, nseg_all, cap);
  if (rc == 0) *nenv = cluster_ensemble(seg_all, nseg_all, cap, env, envcap);
  prof_free(p); return rc;
}

/* ------------------------------------------------------------------------------------------
 * Domain definition by posterior heuristics
 * ------------------------------------------------------------------------------------------ */
typedef struct { DOMAIN *dcl; int ndom, cap; int nregions, nenvelopes, nclustered; float *n2sc; } DDEF;

static void ddef_add(DDEF *dd, const DOMAIN *d)
{ if (dd->ndom == dd->cap) { dd->cap = dd->cap ? dd->cap * 2 : 4; dd->dcl = realloc(dd->dcl, sizeof(DOMAIN) * dd->cap); } dd->dcl[dd->ndom++] = *d; }

static void domain_definition(const PROF *p, const uint8_t *dsq, int L, const float *fxs, const float *bxs, float fwd_xC, DDEF *dd)
{
  XF xf; xf_config(p, L, 1, &xf);
  float invZ = 1.0f / (fwd_xC * xf.move);
  float *btot = calloc((size_t)3*(L+1), sizeof(float)), *etot = btot + (L+1), *mocc = etot + (L+1);
  for (int i = 1; i <= L; i++) {
    float bt = fxs[(size_t)(i-1)*6+3] * bxs[(size_t)(i-1)*5+3]; bt = bt * invZ;
    float et = fxs[(size_t)i*6+0] * bxs[(size_t)i*5+0]; et = et * invZ;
    float w = invZ / fxs[(size_t)i*6+5];
    float a = fxs[(size_t)(i-1)*6+1] * bxs[(size_t)i*5+1]; a = a * xf.loop;
    float b = fxs[(size_t)(i-1)*6+2] * bxs[(size_t)i*5+2]; b = b * xf.loop;
    float c = fxs[(size_t)(i-1)*6+4] * bxs[(size_t)i*5+4]; c = c * xf.loop;
    float njcp = ((a + b) + c) * w;
    btot[i] = btot[i-1] + bt; etot[i] = etot[i-1] + et; mocc[i] = 1.0f - njcp;
  }
  int i = -1, triggered = 0;
  for (int j = 1; j <= L; j++) {
    if (!triggered) {
      if (mocc[j] - (btot[j] - btot[j-1]) < RT2) i = j;
      else if (i == -1) i = j;
      if (mocc[j] >= RT1) triggered = 1;
    } else if (mocc[j] - (etot[j] - etot[j-1]) < RT2) {
      dd->nregions++;
      /* multi-domain test (rt3) */
      float max = -1.0f;
      for (int z = i; z <= j; z++) { float a = etot[z] - etot[i-1], b = btot[j] - btot[z-1]; float en = a < b ? a : b; if (en > max) max = en; }
      if (max >= RT3) {
        /* the region holds more than one domain: resolve it by the trace ensemble (DEV3) */
        dd->nclustered++;
        if (getenv("P7O_TRACE_REGIONS")) fprintf(stderr, "p7o: multi-domain region %d..%d of L=%d (M=%d)\n", i, j, L, p->M);
        int Lr = j - i + 1, cap = Lr < 16 ? Lr : 16, ens_rc;
        float *n2sum = malloc(sizeof(float) * Lr);
        P7O_SEG *seg_all = malloc(sizeof(P7O_SEG) * (size_t)ENS_NSAMPLES * cap), *env = malloc(sizeof(P7O_SEG) * (size_t)Lr); int nseg_all[ENS_NSAMPLES];
        /* a trace with more domains than segment slots: repeat with a larger table (a trace holds at most Lr domains) */
        while ((ens_rc = trace_ensemble(p, dsq, L, i, j, n2sum, seg_all, nseg_all, cap)) != 0 && cap < Lr) {
          cap = cap * 8 < Lr ? cap * 8 : Lr; seg_all = realloc(seg_all, sizeof(P7O_SEG) * (size_t)ENS_NSAMPLES * cap);
        }
        if (ens_rc == 0) {
          for (int pos = i; pos <= j; pos++) dd->n2sc[pos] = logf(n2sum[pos-i] / (float)ENS_NSAMPLES);
          int nenv = cluster_ensemble(seg_all, nseg_all, cap, env, Lr);
          for (int e = 0; e < nenv; e++) {
            int i2 = env[e].sqfrom + i - 1, j2 = env[e].sqto + i - 1;
            /* an envelope that overlaps its predecessor is rescored like any other (HMMER only counts it, "noverlaps"; the
             * duplicate alignments this can produce are hidden at reporting time: workaround of its bug #h74 below) */
            DOMAIN d; memset(&d, 0, sizeof(d)); dd->nenvelopes++;
            if (rescore_envelope(p, dsq, L, i2, j2, dd->n2sc, 1, &d, NULL, NULL, NULL) == 0) ddef_add(dd, &d);
          }
        }
        free(n2sum); free(seg_all); free(env);
      } else {
        DOMAIN d; memset(&d, 0, sizeof(d)); dd->nenvelopes++;
        if (rescore_envelope(p, dsq, L, i, j, dd->n2sc, 0, &d, NULL, NULL, NULL) == 0) ddef_add(dd, &d);
      }
      i = -1; triggered = 0;
    }
  }
  free(btot);
}

/* ------------------------------------------------------------------------------------------
 * Per-target pipeline
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int seq_idx; int L;
  float pre_score, score, sum_score; double lnP;
  DOMAIN *dcl; int ndom; int nreported;
} HIT;

static int filters(const PROF *p, const BIASHMM *bh, const uint8_t *dsq, int L, P7O_STAGES *st, int all_stages)
{
  const float *ev = p->hmm->evparam;
  LENCFG lc; lencfg(p, L, 1, &lc);
  memset(st, 0, sizeof(*st));
  st->null_sc = lc.nullsc;
  float usc, vfsc, fwdsc, filtersc, seq_score; double P; int xJ, xC;
  msv_filter(p, &lc, dsq, L, &xJ, &usc);
  st->msv_xJ = xJ; st->msv_sc = usc;
  seq_score = (float)((double)(usc - lc.nullsc) / LOG2C);
  P = gumbel_surv(seq_score, ev[P7O_MMU], ev[P7O_MLAMBDA]);
  st->pass_msv = !(P > F1);
  if (!st->pass_msv && !all_stages) return 0;
  filtersc = bias_filter(bh, &lc, dsq, L);
  st->bias_sc = filtersc;
  seq_score = (float)((double)(usc - filtersc) / LOG2C);
  P = gumbel_surv(seq_score, ev[P7O_MMU], ev[P7O_MLAMBDA]);
  st->pass_bias = st->pass_msv && !(P > F1);
  if (!st->pass_bias && !all_stages) return 0;
  st->pass_vit = st->pass_bias;
  if (P > F2 || all_stages) {
    vit_filter(p, &lc, dsq, L, &xC, &vfsc);
    st->vit_xC = xC; st->vit_sc = vfsc;
    if (P > F2) {
      seq_score = (float)((double)(vfsc - filtersc) / LOG2C);
      double P2 = gumbel_surv(seq_score, ev[P7O_VMU], ev[P7O_VLAMBDA]);
      if (P2 > F2) st->pass_vit = 0;
    }
  }
  if (!st->pass_vit && !all_stages) return 0;
  XF xf; xf_config(p, L, 1, &xf);
  float xCL; int ns;
  fwdsc = forward(p, &xf, dsq, L, NULL, NULL, &xCL, &ns);
  st->fwd_sc = fwdsc; st->fwd_xC = xCL; st->fwd_nscale = ns;
  seq_score = (float)((double)(fwdsc - filtersc) / LOG2C);
  P = exp_surv(seq_score, ev[P7O_FTAU], ev[P7O_FLAMBDA]);
  st->pass_fwd = st->pass_vit && !(P > F3);
  return st->pass_fwd;
}

int p7o_stages(const P7O_HMM *hmm, const uint8_t *dsq, int L, P7O_STAGES *out)
{
  PROF *p = prof_create(hmm); BIASHMM bh; bias_setup(hmm, &bh);
  filters(p, &bh, dsq, L, out, 1);
  prof_free(p); return 0;
}

/* returns 1 and fills *hit if the target is (provisionally) reportable */
static int pipeline_target(const PROF *p, const BIASHMM *bh, const uint8_t *dsq, int L, HIT *hit)
{
  P7O_STAGES st; const float *ev = p->hmm->evparam;
  if (L == 0) return 0;
  if (!filters(p, bh, dsq, L, &st, 0)) return 0;
  float fwdsc = st.fwd_sc, nullsc = st.null_sc;
  XF xf; xf_config(p, L, 1, &xf);
  float *fxs = malloc(sizeof(float) * 6 * (size_t)(L+1)), *bxs = malloc(sizeof(float) * 5 * (size_t)(L+1));
  float xCL; int ns;
  forward(p, &xf, dsq, L, NULL, fxs, &xCL, &ns);
  backward(p, &xf, dsq, L, fxs, NULL, bxs);
  DDEF dd; memset(&dd, 0, sizeof(dd)); dd.n2sc = calloc((size_t)L + 2, sizeof(float));
  domain_definition(p, dsq, L, fxs, bxs, xCL, &dd);
  free(fxs); free(bxs);
  if (dd.nregions == 0 || dd.nenvelopes == 0 || dd.ndom == 0) { free(dd.n2sc); free(dd.dcl); return 0; }
  float seqbias = 0.f;
  for (int i = 0; i <= L; i++) seqbias += dd.n2sc[i];
  seqbias = flogsum(0.0f, logf(OMEGA) + seqbias);
  float pre_score = (float)((double)(fwdsc - nullsc) / LOG2C);
  float seq_score = (float)((double)(fwdsc - (nullsc + seqbias)) / LOG2C);
  float sum_score = 0.f; int Ld = 0; seqbias = 0.f;
  for (int d = 0; d < dd.ndom; d++) if (dd.dcl[d].envsc - dd.dcl[d].domcorrection > 0.0f) {
    sum_score += dd.dcl[d].envsc; Ld += dd.dcl[d].jenv - dd.dcl[d].ienv + 1; seqbias += dd.dcl[d].domcorrection;
  }
  seqbias = flogsum(0.0f, logf(OMEGA) + seqbias);
  sum_score += (float)((double)(L - Ld) * log((double)((float)L / (float)(L+3))));
  float pre2_score = (float)((double)(sum_score - nullsc) / LOG2C);
  sum_score = (float)((double)(sum_score - (nullsc + seqbias)) / LOG2C);
  if (Ld > 0 && sum_score > seq_score) { seq_score = sum_score; pre_score = pre2_score; }
  double lnP = exp_logsurv(seq_score, ev[P7O_FTAU], ev[P7O_FLAMBDA]);
  hit->L = L; hit->pre_score = pre_score; hit->score = seq_score; hit->sum_score = sum_score; hit->lnP = lnP;
  hit->dcl = dd.dcl; hit->ndom = dd.ndom; hit->nreported = 0;
  for (int d = 0; d < hit->ndom; d++) {
    DOMAIN *dm = &hit->dcl[d];
    int ld = dm->jenv - dm->ienv + 1;
    float bs = dm->envsc + (float)((double)(L - ld) * log((double)((float)L / (float)(L+3))));
    dm->dombias = flogsum(0.0f, logf(OMEGA) + dm->domcorrection);
    dm->bitscore = (float)((double)(bs - (nullsc + dm->dombias)) / LOG2C);
    dm->lnP = exp_logsurv(dm->bitscore, ev[P7O_FTAU], ev[P7O_FLAMBDA]);
  }
  free(dd.n2sc);
  return 1;
}

typedef struct { HIT *h; const char *name; } HSORT;
static int hit_cmp(const void *a, const void *b)
{
  const HSORT *x = a, *y = b;
  double kx = -x->h->lnP, ky = -y->h->lnP;       /* sortkey = -lnP (inclusion by E-value) */
  if (kx < ky) return 1; if (kx > ky) return -1;
  int c = strcmp(x->name, y->name); if (c) return c;
  return (x->h->seq_idx > y->h->seq_idx) - (x->h->seq_idx < y->h->seq_idx);
}

int p7o_search(const P7O_HMMSET *set, const int32_t *model_idx, int nmodels,
               const uint8_t *dsq, const int64_t *offsets, int nseq, const char *const *names,
               double E, double domE, P7O_ROW **rows_out, int *nrows_out)
{
  P7O_ROW *rows = NULL; int nrows = 0, cap = 0;
  double Z = (double)nseq;
  for (int mi = 0; mi < nmodels; mi++) {
    const P7O_HMM *hmm = set->hmm[model_idx[mi]];
    PROF *p = prof_create(hmm); BIASHMM bh; bias_setup(hmm, &bh);
    HIT *hits = NULL; int nh = 0, hcap = 0;
    for (int s = 0; s < nseq; s++) {
      int L = (int)(offsets[s+1] - offsets[s]); HIT h; memset(&h, 0, sizeof(h));
      if (pipeline_target(p, &bh, dsq + offsets[s], L, &h)) {
        h.seq_idx = s;
        if (nh == hcap) { hcap = hcap ? hcap*2 : 16; hits = realloc(hits, sizeof(HIT)*hcap); }
        hits[nh++] = h;
      }
    }
    /* thresholding: sequences by E <= E (Z = #targets), then domZ = #reported sequences */
    HSORT *hs = malloc(sizeof(HSORT) * (nh ? nh : 1)); int nrep = 0;
    for (int i = 0; i < nh; i++) { hs[i].h = &hits[i]; hs[i].name = names ? names[hits[i].seq_idx] : ""; }
    qsort(hs, nh, sizeof(HSORT), hit_cmp);
    for (int i = 0; i < nh; i++) if (exp(hs[i].h->lnP) * Z <= E) nrep++;
    double domZ = (double)nrep;
    for (int i = 0; i < nh; i++) {
      HIT *h = hs[i].h;
      if (!(exp(h->lnP) * Z <= E)) continue;
      for (int d = 0; d < h->ndom; d++) { h->dcl[d].is_reported = (exp(h->dcl[d].lnP) * domZ <= domE); if (h->dcl[d].is_reported) h->nreported++; }
      /* bug-h74 workaround: hide the weaker of two domains with identical alignment coordinates */
      for (int d1 = 0; d1 < h->ndom; d1++)
        for (int d2 = d1 + 1; d2 < h->ndom; d2++) {      /* every pair, sequence coordinates only; the lower bit score goes (the later one on a tie) */
          DOMAIN *a = &h->dcl[d1], *b = &h->dcl[d2];
          if (a->ali_from == b->ali_from && a->ali_to == b->ali_to) {
            DOMAIN *w = (a->bitscore >= b->bitscore) ? b : a;
            if (w->is_reported) { w->is_reported = 0; h->nreported--; }
          }
        }
      int nd = 0;
      for (int d = 0; d < h->ndom; d++) if (h->dcl[d].is_reported) {
        DOMAIN *dm = &h->dcl[d]; nd++;
        if (nrows == cap) { cap = cap ? cap*2 : 64; rows = realloc(rows, sizeof(P7O_ROW)*cap); }
        P7O_ROW *r = &rows[nrows++]; memset(r, 0, sizeof(*r));
        r->model_idx = model_idx[mi]; r->seq_idx = h->seq_idx; r->tlen = h->L; r->qlen = hmm->M;
        r->full_evalue = exp(h->lnP) * Z; r->full_score = h->score; r->full_bias = h->pre_score - h->score;
        r->dom_idx = nd; r->ndom = h->nreported;
        r->c_evalue = exp(dm->lnP) * domZ; r->i_evalue = exp(dm->lnP) * Z;
        r->dom_score = dm->bitscore; r->dom_bias = (float)((double)dm->dombias * LOG2RC);
        r->hmm_from = dm->hmm_from; r->hmm_to = dm->hmm_to; r->ali_from = dm->ali_from; r->ali_to = dm->ali_to;
        r->env_from = dm->ienv; r->env_to = dm->jenv;
        r->acc = (float)((double)dm->oasc / (1.0 + fabs((double)(float)(dm->jenv - dm->ienv))));
        r->full_lnP = h->lnP; r->dom_lnP = dm->lnP;
      }
    }
    for (int i = 0; i < nh; i++) free(hits[i].dcl);
    free(hits); free(hs); prof_free(p);
  }
  *rows_out = rows; *nrows_out = nrows;
  return 0;
}

/* The MSV filter alone over every (model, sequence) pair: returns the cells (residues x model nodes) it scored and, in *checksum, the sum of
 * the final bytes (so that the scalar and the striped form can be seen to agree on the sample they are timed on).  bench.py's cpu_baseline:
 * GCUPS of the stage every pair goes through. */
int64_t p7o_msv_probe(const P7O_HMMSET *set, const int32_t *model_idx, int nmodels, const uint8_t *dsq, const int64_t *offsets, int nseq, int64_t *checksum)
{
  int64_t cells = 0, sum = 0;
  for (int mi = 0; mi < nmodels; mi++) {
    const P7O_HMM *hmm = set->hmm[model_idx[mi]];
    PROF *p = prof_create(hmm);
    for (int s = 0; s < nseq; s++) {
      int L = (int)(offsets[s+1] - offsets[s]); if (L == 0) continue;
      LENCFG lc; lencfg(p, L, 1, &lc);
      int xJ; float sc;
      msv_filter(p, &lc, dsq + offsets[s], L, &xJ, &sc);
      sum += xJ; cells += (int64_t)L * hmm->M;
    }
    prof_free(p);
  }
  if (checksum) *checksum = sum;
  return cells;
}

/* ------------------------------------------------------------------------------------------
 * domtblout text (column contract: checkm/hmmer.py:184-200, 255-285)
 * ------------------------------------------------------------------------------------------ */
char *p7o_format_domtblout(const P7O_HMMSET *set, const P7O_ROW *rows, int nrows,
                           const char *const *names, const char *const *descs)
{
  size_t cap = 4096, len = 0; char *out = malloc(cap);
#define EMIT(...) do { for (;;) { int n_ = snprintf(out + len, cap - len, __VA_ARGS__); if ((size_t)n_ < cap - len) { len += n_; break; } cap = cap * 2 + n_; out = realloc(out, cap); } } while (0)
  int tnamew = 20, qnamew = 20, taccw = 10, qaccw = 10;
  for (int i = 0; i < nrows; i++) {
    int n = (int)strlen(names[rows[i].seq_idx]); if (n > tnamew) tnamew = n;
    const P7O_HMM *h = set->hmm[rows[i].model_idx];
    n = (int)strlen(h->name); if (n > qnamew) qnamew = n;
    if (h->acc) { n = (int)strlen(h->acc); if (n > qaccw) qaccw = n; }
  }
  EMIT("#%*s %22s %40s %11s %11s %11s\n", tnamew+qnamew-1+15+taccw+qaccw, "", "--- full sequence ---", "-------------- this domain -------------", "hmm coord", "ali coord", "env coord");
  EMIT("#%-*s %-*s %5s %-*s %-*s %5s %9s %6s %5s %3s %3s %9s %9s %6s %5s %5s %5s %5s %5s %5s %5s %4s %s\n",
       tnamew-1, " target name", taccw, "accession", "tlen", qnamew, "query name", qaccw, "accession", "qlen", "E-value", "score", "bias", "#", "of", "c-Evalue", "i-Evalue", "score", "bias", "from", "to", "from", "to", "from", "to", "acc", "description of target");
  { char dash[512];
#define DASH(n) (memset(dash, '-', (n)), dash[(n)] = 0, dash)
    EMIT("#%s ", DASH(tnamew-1)); EMIT("%s ", DASH(taccw)); EMIT("%s ", DASH(5)); EMIT("%s ", DASH(qnamew)); EMIT("%s ", DASH(qaccw));
    EMIT("----- --------- ------ ----- --- --- --------- --------- ------ ----- ----- ----- ----- ----- ----- ----- ---- ---------------------\n");
  }
  for (int i = 0; i < nrows; i++) {
    const P7O_ROW *r = &rows[i]; const P7O_HMM *h = set->hmm[r->model_idx];
    EMIT("%-*s %-*s %5d %-*s %-*s %5d %9.2g %6.1f %5.1f %3d %3d %9.2g %9.2g %6.1f %5.1f %5d %5d %5ld %5ld %5ld %5ld %4.2f %s\n",
         tnamew, names[r->seq_idx], taccw, "-", r->tlen, qnamew, h->name, qaccw, (h->acc && h->acc[0]) ? h->acc : "-", r->qlen,
         r->full_evalue, r->full_score, r->full_bias, r->dom_idx, r->ndom, r->c_evalue, r->i_evalue, r->dom_score, r->dom_bias,
         r->hmm_from, r->hmm_to, (long)r->ali_from, (long)r->ali_to, (long)r->env_from, (long)r->env_to, r->acc,
         (descs && descs[r->seq_idx] && descs[r->seq_idx][0]) ? descs[r->seq_idx] : "-");
  }
  EMIT("#\n# Program:         hmmsearch\n# Pipeline mode:   SEARCH\n# [ok]\n");
  return out;
}
