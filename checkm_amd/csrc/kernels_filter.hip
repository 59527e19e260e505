// kernels_filter.hip -- the rest of the acceleration-filter cascade, gfx950 only:
//   msv_full_kernel    full multi-hit MSV for the rare pairs whose J state could be used
//   bias_kernel        2-state composition filter (Forward, power-of-two rescaling); F1/F2 tests follow on the host
//   vit_kernel<QH>     16-bit Viterbi filter, one wavefront per pair, packed words, lazy-F D->D passes
// Reference stage being replaced: the MSV -> bias -> Viterbi part of hmmsearch's per-target pipeline
// (process launched at checkm/hmmer.py:70 with the options of checkm/markerGeneFinder.py:141).
// Every decision is taken on IEEE basic operations only (no device libm), so it is bit-identical
// to the host formulation: thresholds were converted to score space on the host (host_profile.cpp).
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <cstring>
#include "dev_types.h"
#include "cascade_dev.h"
#include "xlane.h"

namespace ckm {



constexpr int KP_SYMS = 29;    // rows of the byte cost table (one per alphabet symbol)

constexpr double LN2D = 0.69314718055994529;
constexpr int NEG16 = -32768;

// --------------------------------------------------------------------------------------------
// MSV score arithmetic of the exact kernels (the SSV kernels finish their own pairs: kernels_ssv.hip, ssv_finish)
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ float msv_score(int xJ, int tjb, int base, float scale_b) {
  float sc = (float)(xJ - tjb) - (float)base;
  sc = sc / scale_b;
  sc = sc - 3.0f;
  return sc;
}
__device__ __forceinline__ float to_bits(float sc, float nullsc) {
  return (float)((double)(sc - nullsc) / LN2D);
}

// --------------------------------------------------------------------------------------------
// full multi-hit MSV: one wavefront (= one workgroup) per pair.  The model's byte costs for all 29 symbols are copied
// into LDS once per pair and the residues ride in registers, 64 at a time, so no global load sits on the row-to-row
// chain (rows of the longest sequence bound the launch).  Cells k = lane + 64*j, previous/current row in LDS.
// --------------------------------------------------------------------------------------------
// `queue.list` is unused: entry k of the queue is pairs[k].  With `decide` the kernel is the exact-MSV stage of the device-driven
// cascade: a pair whose exact score passes F1 (the bit-space form of the test the SSV epilogue takes in nats) joins the candidate table.
__global__ void __launch_bounds__(64) msv_full_kernel(WorkQueue queue, const PairRec *__restrict__ pairs, const DevModel *__restrict__ models,
                                                     const LenEntry *__restrict__ lentab, const uint8_t *__restrict__ res,
                                                     const uint64_t *__restrict__ seq_off, const int32_t *__restrict__ seq_len,
                                                     int32_t *__restrict__ out_xJ /* -1 overflow */, float *__restrict__ out_usc, int maxMp,
                                                     CascadeDev cd, int decide) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
 const uint32_t nqueue = queue_len(queue);
 for (uint32_t pi = blockIdx.x; pi < nqueue; pi += gridDim.x) {
  __syncthreads();
  const PairRec pr = pairs[pi];
  const DevModel &md = models[pr.model];
  const int M = md.M, L = seq_len[pr.seq], W = M + 1;
  uint8_t *tab = reinterpret_cast<uint8_t *>(smem);
  int16_t *row = reinterpret_cast<int16_t *>(smem + (((size_t)KP_SYMS * (maxMp + 1) + 15) & ~(size_t)15));
  {
    const gp<uint32_t> src = gptr(reinterpret_cast<const uint32_t *>(md.rbv));     // table start is 256-byte aligned
    uint32_t *dst = reinterpret_cast<uint32_t *>(tab);
    const int nw32 = (KP_SYMS * W + 3) >> 2;
    for (int j = lane; j < nw32; j += 64) dst[j] = src[j];
  }
  const uint8_t *rp = res + seq_off[pr.seq];
  const LenEntry le = lentab[L];
  const int tjbm = (le.tjb_b + md.tbm_b) & 0xff;
  int16_t *dp = row, *nw = row + maxMp;
  for (int k = lane; k < M; k += 64) dp[k] = 0;
  __syncthreads();
  // (copies: after a barrier the compiler has to assume the struct in global memory changed, and would fetch these again in every row)
  const int bias_b = md.bias_b, tec_b = md.tec_b, base_b = md.base_b;
  int xJ = 0, xB = max(base_b - tjbm, 0);
  bool overflow = false;
  int chunk = (lane < L) ? (int)rp[lane] : 0;
  for (int i0 = 0; i0 < L && !overflow; i0 += 64) {
    const int nxt = (i0 + 64 + lane < L) ? (int)rp[i0 + 64 + lane] : 0;     // next 64 residues, a whole chunk ahead
    const int n = min(64, L - i0);
    for (int r = 0; r < n; ++r) {
      const int x = __builtin_amdgcn_readlane(chunk, r);
      const uint8_t *cost = tab + x * W + 1;
      int xE = 0;
#pragma unroll 4
      for (int k = lane; k < M; k += 64) {
        const int mp = (k > 0) ? (int)dp[k - 1] : 0;
        int sv = max(mp, xB);
        sv = min(sv + bias_b, 255);
        sv = max(sv - (int)cost[k], 0);
        xE = max(xE, sv);
        nw[k] = (int16_t)sv;
      }
      xE = wave_max(xE);
      if (min(xE + bias_b, 255) == 255) { overflow = true; break; }
      xE = max(xE - tec_b, 0);
      xJ = max(xJ, xE);
      xB = max(max(base_b, xJ) - tjbm, 0);
      int16_t *t = dp; dp = nw; nw = t;
      __syncthreads();
    }
    chunk = nxt;
  }
  if (lane == 0) {
    const float usc = overflow ? __builtin_inff() : msv_score(xJ, le.tjb_b, md.base_b, md.scale_b);
    if (out_xJ) out_xJ[pi] = overflow ? -1 : xJ;
    if (out_usc) out_usc[pi] = usc;
    if (decide && to_bits(usc, le.nullsc) >= md.thr_msv_f1) {
      const uint32_t k = atomicAdd(&cd.cnt[CC_CAND], 1u);
      if (k < cd.cap_cand) { PairRec r = pr; r.usc = usc; r.filtersc = 0.f; cd.cand[k] = r; } else atomicOr(&cd.gcnt[CC_STATUS], (uint32_t)CS_CAND);
    }
  }
 }
}

// --------------------------------------------------------------------------------------------
// bias filter: one thread per pair
// --------------------------------------------------------------------------------------------
// Output per pair: [0] d0+d1 (the two states' scaled Forward mass), [1] power-of-two exponent, [2] unused.
// The logarithm and the F1/F2 tests are taken on the host (same libm as every other score).
__global__ void bias_kernel(const PairRec *__restrict__ pairs, uint32_t npairs, const DevModel *__restrict__ models,
                            const uint8_t *__restrict__ res, const uint64_t *__restrict__ seq_off, const int32_t *__restrict__ seq_len,
                            float *__restrict__ dbg_d) {
  const uint32_t pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= npairs) return;
  PairRec pr = pairs[pi];
  const DevModel &md = models[pr.model];
  const int L = seq_len[pr.seq];
  const uint8_t *rp = res + seq_off[pr.seq];
  float d0 = md.bpi0, d1 = md.beo1[rp[0]] * md.bpi1;
  int nexp = 0;
  for (int i = 1; i < L; ++i) {
    const float n0 = d0 * md.bt00 + d1 * md.bt10;
    const float n1 = (d0 * md.bt01 + d1 * md.bt11) * md.beo1[rp[i]];
    d0 = n0; d1 = n1;
    const float mx = fmaxf(d0, d1);
    if (mx < 0x1p-40f) { d0 *= 0x1p64f; d1 *= 0x1p64f; nexp -= 64; }
    else if (mx > 0x1p40f) { d0 *= 0x1p-64f; d1 *= 0x1p-64f; nexp += 64; }
  }
  dbg_d[(size_t)pi * 3 + 0] = d0 + d1;
  dbg_d[(size_t)pi * 3 + 1] = (float)nexp;
  dbg_d[(size_t)pi * 3 + 2] = 0.f;
}

// The bias filter of the device-driven cascade: same recurrence, one thread per candidate of a table whose length is only known on
// the device (grid-stride loop), followed by the F1 / F2 decisions in their conservative form.  filtersc is stored in its
// approximate form (the later device tests use it); the host recomputes it with libm from the two raw numbers.
//   bits(usc, filtersc) <  F1 - margin            dead
//   bits >= F2 + margin                           Viterbi filter skipped (HMMER runs it only when the MSV P-value is above F2)
//   bits <  F2 - margin                           Viterbi filter, fast kernel first
//   otherwise (within the margin of F2)           the exact Viterbi kernel runs and the pair goes on whatever it says: the host decides
__global__ void __launch_bounds__(128) bias_filter_kernel(CascadeDev cd, const DevModel *__restrict__ models, const LenEntry *__restrict__ lentab,
                                                         const uint8_t *__restrict__ res, const uint64_t *__restrict__ seq_off) {
  // the emission odds of the pair's model sit in the thread's own LDS row (stride 31 words: lanes asking for the same symbol hit 32
  // different banks) and the four transition odds in registers: nothing on the residue-to-residue chain touches global memory
  __shared__ float beo_s[128 * 31];
  float *be = beo_s + threadIdx.x * 31;
  const uint32_t n = min(cd.cnt[CC_CAND], cd.cap_cand);
  for (uint32_t pi = blockIdx.x * blockDim.x + threadIdx.x; pi < n; pi += gridDim.x * blockDim.x) {
    PairRec pr = cd.cand[pi];
    const DevModel &md = models[pr.model];
    const int L = cd.seq_len[pr.seq];
    const uint8_t *rp = res + seq_off[pr.seq];
#pragma unroll
    for (int x = 0; x < 30; ++x) be[x] = md.beo1[x];
    const float t00 = md.bt00, t01 = md.bt01, t10 = md.bt10, t11 = md.bt11;
    // residues 16 at a time (sequences are 16-byte aligned and padded)
    const uint4 *rp4 = reinterpret_cast<const uint4 *>(rp);
    uint32_t w0, w1, w2, w3;
    { const uint4 c = rp4[0]; w0 = c.x; w1 = c.y; w2 = c.z; w3 = c.w; }
    float d0 = md.bpi0, d1 = be[w0 & 0xffu] * md.bpi1;
    int nexp = 0;
#define BIAS_STEP(x)                                                                  \
    {                                                                                 \
      const float n0 = d0 * t00 + d1 * t10;                                           \
      const float n1 = (d0 * t01 + d1 * t11) * be[(x)];                               \
      d0 = n0; d1 = n1;                                                               \
      const float mx = fmaxf(d0, d1);                                                 \
      if (mx < 0x1p-40f) { d0 *= 0x1p64f; d1 *= 0x1p64f; nexp -= 64; }                \
      else if (mx > 0x1p40f) { d0 *= 0x1p-64f; d1 *= 0x1p-64f; nexp += 64; }          \
    }
#define BIAS_WORD(wv, i0)                                                             \
    _Pragma("unroll") for (int b = 0; b < 4; ++b) { const int i = (i0) + b; if (i >= 1 && i < L) BIAS_STEP(((wv) >> (8 * b)) & 0xffu) }
    for (int c0 = 0; c0 < L; c0 += 16) {
      uint32_t n0w = w0, n1w = w1, n2w = w2, n3w = w3;
      if (c0 + 16 < L) { const uint4 c = rp4[(c0 >> 4) + 1]; n0w = c.x; n1w = c.y; n2w = c.z; n3w = c.w; }
      BIAS_WORD(w0, c0) BIAS_WORD(w1, c0 + 4) BIAS_WORD(w2, c0 + 8) BIAS_WORD(w3, c0 + 12)
      w0 = n0w; w1 = n1w; w2 = n2w; w3 = n3w;
    }
#undef BIAS_WORD
#undef BIAS_STEP
    const float dsum = d0 + d1;
    cd.bias_raw[2 * (size_t)pi] = dsum; cd.bias_raw[2 * (size_t)pi + 1] = (float)nexp;
    const float filtersc = (approx_ln(dsum) + (float)nexp * LN2_F) + lentab[L].bias_tail;
    cd.cand[pi].filtersc = filtersc;
    cd.vit_fast[pi] = 0.f; cd.vit_exact[pi] = 0.f; cd.vit_flag[pi] = 0u;
    const float sc = (pr.usc - filtersc) * LOG2E_F;
    uint8_t route = 0;
    if (!(sc >= md.thr_msv_f1 - cd.margin_msv)) { cd.route[pi] = 0xffu; continue; }              // dead
    if (sc >= md.thr_msv_f2 + cd.margin_msv) { cd.route[pi] = 0; pass_to_forward(cd, md, pi, pr.model, pr.seq); continue; }
    if (sc < md.thr_msv_f2 - cd.margin_msv) { route = 1; queue_push(cd, cd.vq, CC_VQ, md.vit_cls, cd.cap_vq, pi, (uint32_t)CS_VQ); }
    else { route = 2 | 0x10; queue_push(cd, cd.vxq, CC_VXQ, md.vitx_cls, cd.cap_vq, pi, (uint32_t)CS_VQ); }
    cd.route[pi] = route;
  }
}

// --------------------------------------------------------------------------------------------
// Viterbi filter: one wavefront per pair, packed 2 x i16 saturating arithmetic (v_pk_add_i16 clamp,
// v_pk_max_i16) -- the same word arithmetic HMMER's striped filter performs, so results are
// identical whatever the layout.  Lane z owns the 2*QH consecutive cells z*2QH .. z*2QH+2QH-1;
// register j packs (cell j, cell j+QH) of the lane, i.e. the wave holds 128 stripes of QH cells
// (lane z low half = stripe 2z, high half = stripe 2z+1).  The move (i-1,k-1)->(i,k) is a register
// rename plus ONE wave_shr:1 DPP + alignbit per state array.  D->D: one in-stripe pass, then
// "lazy-F" passes that carry stripe ends forward until no D cell improves (exact: max / saturating
// add with non-positive addends).
// --------------------------------------------------------------------------------------------
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32;
__device__ __forceinline__ u32 pk_adds(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_elementwise_add_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b))); }
__device__ __forceinline__ u32 pk_max(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b))); }
constexpr u32 NEG2 = 0x80008000u;     // two -32768 words
// value of the previous stripe: low half <- previous lane's high half, high half <- own low half.  `hold` is a persistent register
// that starts as NEG2 in every lane: the DPP move (bound_ctrl off) never writes lane 0, which has no predecessor, so lane 0 keeps
// -32768 for ever and no constant has to be materialised per shift (one v_mov less per call: 5 per row).
__device__ __forceinline__ u32 stripe_shift(u32 v, u32 &hold) {
  hold = (u32)__builtin_amdgcn_update_dpp((int)hold, (int)v, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
  return __builtin_amdgcn_alignbit(v, hold, 16);
}

// FAST: the J state is assumed unused (xJ <= xN throughout), which makes xB a constant and removes every per-row
// reduction: the rows only feed a running element-wise maximum.  The result is exact when max xE + E->J <= base (flag 0);
// otherwise it is a LOWER bound of the exact score (max / saturating add are monotone in xB) and flag = 1: the caller
// accepts the pair if the bound already passes F2 and re-runs the exact kernel if it does not.
// Every wavefront of the 4-wave workgroups takes its share of the queue's candidate ids.  With `decide` the kernel is a stage of the
// device-driven cascade and its epilogue takes the F2 decision (conservative band of cd.margin_vit bits around the threshold; the
// host repeats the test exactly on the recorded score):
//   FAST   score + F2 margin passes -> Forward;  J flag set (score is only a lower bound) -> exact queue of the same class;
//          within the margin -> Forward (the host decides);  else dead
//   exact  score within the margin or above -> Forward; pairs whose NEED for the filter was itself within the margin go on regardless
template <int QH, bool FAST>
__global__ void __launch_bounds__(256) vit_kernel(WorkQueue queue, const PairRec *__restrict__ pairs,
                                                  const DevModel *__restrict__ models, const LenEntry *__restrict__ lentab,
                                                  const uint8_t *__restrict__ res, const uint64_t *__restrict__ seq_off,
                                                  const int32_t *__restrict__ seq_len, int32_t *__restrict__ out_xC, float *__restrict__ out_sc,
                                                  uint32_t *__restrict__ out_flag, CascadeDev cd, int decide) {
  const int lane = threadIdx.x & 63;
 const uint32_t nqueue = queue_len(queue);
 for (uint32_t qk = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); qk < nqueue; qk += gridDim.x * (blockDim.x >> 6)) {
  const uint32_t pi = queue.list[qk];
  const PairRec pr = pairs[pi];
  const DevModel &md = models[pr.model];
  constexpr int ROW = QH * 64;                      // u32 words per table row
  const int L = __builtin_amdgcn_readfirstlane(seq_len[pr.seq]);     // (one pair per wavefront: the length is uniform, and as an SGPR it keeps the row counter and the residue feed's bookkeeping off the VALU)
  const uint8_t *rp = res + seq_off[pr.seq];
  const LenEntry le = lentab[L];
  u32 tBM[QH], tMM[QH], tIM[QH], tDM[QH], tMD[QH], tMI[QH], tII[QH], tDD[QH];
#pragma unroll
  for (int j = 0; j < QH; ++j) {
    const gp<u32> t = gptr(md.vit_t) + j * 64 + lane;
    tBM[j] = t[0 * ROW]; tMM[j] = t[1 * ROW]; tIM[j] = t[2 * ROW]; tDM[j] = t[3 * ROW];
    tMD[j] = t[4 * ROW]; tMI[j] = t[5 * ROW]; tII[j] = t[6 * ROW]; tDD[j] = t[7 * ROW];
  }
  u32 Mv[QH], Iv[QH], Dv[QH];
#pragma unroll
  for (int j = 0; j < QH; ++j) Mv[j] = Iv[j] = Dv[j] = NEG2;
  int xN = md.base_w, xB = xN + le.w_move, xJ = NEG16, xC = NEG16;
  bool overflow = false;
  u32 xEv = NEG2;
  // emission words one row ahead; the residues come 64 at a time in one register (xlane.h: ResUp), so the address of a row's words
  // is an SGPR and no load of the row loop depends on another
  const gp<u32> vit_e = gptr(md.vit_e);
  ResUp feed; feed.init(rp, L, lane);
  u32 e[QH], e2[QH];
  {
    const gp<u32> er = vit_e + (size_t)feed.get(0) * ROW + lane;
#pragma unroll
    for (int j = 0; j < QH; ++j) e[j] = er[j * 64];
  }
  u32 hm = NEG2, hi = NEG2, hd = NEG2, hc = NEG2;          // stripe_shift's persistent registers
  // one row; `ec` holds this row's emission words, `en` receives the next row's (the two arrays swap roles from row to row, so no
  // register copies are needed).  Returns true when the exact variant overflowed.
  auto row = [&](int i, u32 (&ec)[QH], u32 (&en)[QH]) -> bool {
    {
      const int xn = feed.get((i + 1 < L) ? i + 1 : L - 1);
      const gp<u32> er = vit_e + (size_t)xn * ROW + lane;
#pragma unroll
      for (int j = 0; j < QH; ++j) en[j] = er[j * 64];
    }
    const u32 ms0 = stripe_shift(Mv[QH - 1], hm), is0 = stripe_shift(Iv[QH - 1], hi), ds0 = stripe_shift(Dv[QH - 1], hd);
    const u32 xBv = ((u32)(xB & 0xffff)) * 0x10001u;        // FAST: loop-invariant
    u32 mdv[QH];
    if (!FAST) xEv = NEG2;
#pragma unroll
    for (int j = QH - 1; j >= 0; --j) {
      const u32 mp = j ? Mv[j - 1] : ms0, ip = j ? Iv[j - 1] : is0, dp = j ? Dv[j - 1] : ds0;
      u32 sv = pk_adds(xBv, tBM[j]);
      sv = pk_max(sv, pk_adds(mp, tMM[j]));
      sv = pk_max(sv, pk_adds(ip, tIM[j]));
      sv = pk_max(sv, pk_adds(dp, tDM[j]));
      sv = pk_adds(sv, ec[j]);
      const u32 ni = pk_max(pk_adds(Mv[j], tMI[j]), pk_adds(Iv[j], tII[j]));
      Iv[j] = ni; Mv[j] = sv;
      xEv = pk_max(xEv, sv);
      mdv[j] = pk_adds(sv, tMD[j]);
    }
    // D, pass 1: the first cell of a stripe takes the M->D word of the previous stripe's last cell
    Dv[0] = stripe_shift(mdv[QH - 1], hd);
#pragma unroll
    for (int j = 1; j < QH; ++j) Dv[j] = pk_max(mdv[j - 1], pk_adds(Dv[j - 1], tDD[j - 1]));
    u32 carry = pk_adds(Dv[QH - 1], tDD[QH - 1]);
    // lazy-F passes: carry stripe ends forward while some first cell still improves
    for (int pass = 0; pass < 128; ++pass) {
      u32 cs = stripe_shift(carry, hc);
      const s16x2 c2 = __builtin_bit_cast(s16x2, cs), d2 = __builtin_bit_cast(s16x2, Dv[0]);
      if (!__any((c2.x > d2.x) || (c2.y > d2.y))) break;
#pragma unroll
      for (int j = 0; j < QH; ++j) { Dv[j] = pk_max(Dv[j], cs); cs = pk_adds(cs, tDD[j]); }
      carry = cs;
    }
    if (!FAST) {
      const s16x2 x2 = __builtin_bit_cast(s16x2, xEv);
      int xE = max((int)x2.x, (int)x2.y);
      xE = wave_max(xE);
      if (xE >= 32767) { overflow = true; return true; }
      xC = max(xC, xE + md.wE_move);
      xJ = max(xJ, xE + md.wE_loop);
      xB = max(xJ + le.w_move, xN + le.w_move);
    }
    return false;
  };
  {
    int i = 0;
    bool stop = false;
    for (; i + 1 < L; i += 2) {
      if (row(i, e, e2)) { stop = true; break; }
      if (row(i + 1, e2, e)) { stop = true; break; }
    }
    if (!stop && i < L) (void)row(i, e, e2);
  }
  bool jflag = false;
  if (FAST) {
    const s16x2 x2 = __builtin_bit_cast(s16x2, xEv);
    int xE = max((int)x2.x, (int)x2.y);
    xE = wave_max(xE);
    overflow = xE >= 32767;
    xC = max(xC, xE + md.wE_move);
    jflag = (xE + md.wE_loop) > xN;
  }
  if (lane == 0) {
    const uint32_t flag = (jflag && !overflow) ? 1u : 0u;
    float vsc;
    if (overflow) vsc = __builtin_inff();
    else if (xC > NEG16) { float sc = (float)xC + (float)le.w_move - (float)md.base_w; sc = sc / md.scale_w; sc = sc - 3.0f; vsc = sc; }
    else vsc = -__builtin_inff();
    if (!decide) {
      if (out_flag) out_flag[pi] = flag;
      out_xC[pi] = overflow ? 32767 : xC; out_sc[pi] = vsc;
    } else {
      const float v = (vsc - pr.filtersc) * LOG2E_F;
      if (FAST) {
        cd.vit_fast[pi] = vsc; cd.vit_flag[pi] = flag;
        if (v >= md.thr_vit_f2 + cd.margin_vit) pass_to_forward(cd, md, pi, pr.model, pr.seq);
        else if (flag) { cd.route[pi] = 2; queue_push(cd, cd.vxq, CC_VXQ, md.vitx_cls, cd.cap_vq, pi, (uint32_t)CS_VQ); }
        else if (v >= md.thr_vit_f2 - cd.margin_vit) pass_to_forward(cd, md, pi, pr.model, pr.seq);
      } else {
        cd.vit_exact[pi] = vsc;
        if ((cd.route[pi] & 0x10) || v >= md.thr_vit_f2 - cd.margin_vit) pass_to_forward(cd, md, pi, pr.model, pr.seq);
      }
    }
  }
 }
}

// --------------------------------------------------------------------------------------------
// The FAST Viterbi filter on 16 lanes per pair, FOUR pairs per wavefront (round 3): a model of cfg3's median length (190 nodes) fills 74 % of
// the 128 stripes a wavefront holds and pays the per-row fixed part alone; here it takes 32 stripes of Q cells (lane z of a DPP row
// owns cells z*2Q .. z*2Q+2Q-1, register j = (cell j, cell j+Q): M <= 32 Q <= 512) and four pairs share every instruction.  Same word
// arithmetic, so the same score whatever the layout.  The four pairs of a wavefront are independent (own model, own sequence, own
// tables: per-lane pointers); rows run to the longest of the four sequences, a pair that has ended reads the all-impossible pad symbol
// (its running maximum no longer moves) -- the queue is close to length-sorted, because the SSV blocks it descends from run longest
// slices first.  stripe moves are row_shr:1 inside the DPP row (lane 0 of a row has no predecessor and keeps -32768).
// Only the device-driven cascade uses it (decide): pairs whose bound needs the exact kernel go to the wave-per-pair queue of their model.
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 stripe_shift16(u32 v, u32 &hold) {
  hold = (u32)__builtin_amdgcn_update_dpp((int)hold, (int)v, 0x111 /*row_shr:1*/, 0xf, 0xf, false);
  return __builtin_amdgcn_alignbit(v, hold, 16);
}

template <int Q>
__global__ void __launch_bounds__(256) vit16_kernel(WorkQueue queue, const PairRec *__restrict__ pairs, const DevModel *__restrict__ models,
                                                    const LenEntry *__restrict__ lentab, const uint8_t *__restrict__ res, const uint64_t *__restrict__ seq_off,
                                                    const int32_t *__restrict__ seq_len, CascadeDev cd) {
  const int lane = threadIdx.x & 63, z = lane & 15, g = lane >> 4;
  constexpr int ROW = Q * 16;                       // u32 words per table row
  const uint32_t nqueue = queue_len(queue);
  // Round 6: the four pairs of a wavefront run to the LONGEST of their sequences, and the queue -- appended to by atomics from many
  // blocks -- holds lengths in no particular order (the mean of the longest of four log-normal lengths is ~1.5 x the mean length).  A
  // wavefront therefore takes a CHUNK of 64 queue entries, sorts them by length in its lanes (a bitonic network over ds_swizzle /
  // ds_bpermute moves: ~130 instructions per chunk against ~36,000 per quad of pairs) and runs neighbours of the sorted order together.
  // S wavefronts share a chunk (each sorts it and takes every S-th quad) when the queue has fewer chunks than the launch has wavefronts.
  const uint32_t nchunks = (nqueue + 63u) >> 6, nwaves = gridDim.x * (blockDim.x >> 6), wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  uint32_t S = 1;
  while (S < 16u && nchunks * S < nwaves) S <<= 1;
  for (uint32_t u = wid; u < nchunks * S; u += nwaves) {
    const uint32_t chunk = u / S, sub0 = u % S;
    const uint32_t qe = chunk * 64u + (uint32_t)lane;
    const uint32_t pi_l = qe < nqueue ? queue.list[qe] : 0u;
    // key: (length + 1) << 6 | lane for a real entry, the bare lane for one beyond the queue's end (sorts behind every real one)
    uint32_t key = (uint32_t)lane;
    if (qe < nqueue) key |= ((uint32_t)seq_len[pairs[pi_l].seq] + 1u) << 6;
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
      for (int j = k >> 1; j > 0; j >>= 1) {
        const uint32_t other = (uint32_t)__shfl_xor((int)key, j);
        const bool desc = (lane & k) == 0;                    // (descending overall: the last merge, k = 64, has every lane in a descending block)
        const bool lower = (lane & j) == 0;
        const uint32_t mx = key > other ? key : other, mn = key > other ? other : key;
        key = (lower == desc) ? mx : mn;
      }
    }
   for (uint32_t sub = sub0; sub < 16u; sub += S) {
    const uint32_t kq = (uint32_t)__shfl((int)key, (int)(sub * 4u));              // the quad's first (longest) entry
    if ((kq >> 6) == 0u) break;                                                   // (uniform: nothing real from here on)
    const uint32_t kg = (uint32_t)__shfl((int)key, (int)(sub * 4u) + g);
    const bool valid = (kg >> 6) != 0u;
    const uint32_t pi = (uint32_t)__shfl((int)pi_l, (int)((valid ? kg : kq) & 63u));   // (a group beyond the queue's end repeats the quad's first pair and reports nothing)
    const PairRec pr = pairs[pi];
    const DevModel &md = models[pr.model];
    const int L = seq_len[pr.seq];
    int Lmax = __builtin_amdgcn_readlane(L, 0);
    Lmax = max(Lmax, __builtin_amdgcn_readlane(L, 16));
    Lmax = max(Lmax, __builtin_amdgcn_readlane(L, 32));
    Lmax = max(Lmax, __builtin_amdgcn_readlane(L, 48));
    const uint8_t *rp = res + seq_off[pr.seq];
    const LenEntry le = lentab[L];
    u32 tBM[Q], tMM[Q], tIM[Q], tDM[Q], tMD[Q], tMI[Q], tII[Q], tDD[Q];
    {
      const gp<u32> t = gptr(md.vit16_t) + z;
#pragma unroll
      for (int j = 0; j < Q; ++j) {
        tBM[j] = t[0 * ROW + j * 16]; tMM[j] = t[1 * ROW + j * 16]; tIM[j] = t[2 * ROW + j * 16]; tDM[j] = t[3 * ROW + j * 16];
        tMD[j] = t[4 * ROW + j * 16]; tMI[j] = t[5 * ROW + j * 16]; tII[j] = t[6 * ROW + j * 16]; tDD[j] = t[7 * ROW + j * 16];
      }
    }
    u32 Mv[Q], Iv[Q], Dv[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) Mv[j] = Iv[j] = Dv[j] = NEG2;
    const int xN = md.base_w, xB = xN + le.w_move;
    const u32 xBv = ((u32)(xB & 0xffff)) * 0x10001u;
    u32 xEv = NEG2;
    const gp<u32> vit_e = gptr(md.vit16_e) + z;
    // residues four at a time (sequences are 16-byte aligned); rows beyond the pair's own length read the pad symbol
    auto sym = [&](u32 word, int i) -> u32 { return (i < L) ? ((word >> (8 * (i & 3))) & 0xffu) : 29u; };
    u32 rw = *reinterpret_cast<const u32 *>(rp);
    u32 e[Q], e2[Q];
    {
      const gp<u32> er = vit_e + (size_t)sym(rw, 0) * ROW;
#pragma unroll
      for (int j = 0; j < Q; ++j) e[j] = er[j * 16];
    }
    u32 hm = NEG2, hi = NEG2, hd = NEG2, hc = NEG2;
    auto row = [&](int i, u32 (&ec)[Q], u32 (&en)[Q]) {
      {
        const int i1 = i + 1;
        if ((i1 & 3) == 0 && i1 < L) rw = *reinterpret_cast<const u32 *>(rp + i1);      // (i1 < L: never reads past the pair's own sequence)
        const gp<u32> er = vit_e + (size_t)sym(rw, i1) * ROW;
#pragma unroll
        for (int j = 0; j < Q; ++j) en[j] = er[j * 16];
      }
      const u32 ms0 = stripe_shift16(Mv[Q - 1], hm), is0 = stripe_shift16(Iv[Q - 1], hi), ds0 = stripe_shift16(Dv[Q - 1], hd);
      u32 mdv[Q];
#pragma unroll
      for (int j = Q - 1; j >= 0; --j) {
        const u32 mp = j ? Mv[j - 1] : ms0, ip = j ? Iv[j - 1] : is0, dp = j ? Dv[j - 1] : ds0;
        u32 sv = pk_adds(xBv, tBM[j]);
        sv = pk_max(sv, pk_adds(mp, tMM[j]));
        sv = pk_max(sv, pk_adds(ip, tIM[j]));
        sv = pk_max(sv, pk_adds(dp, tDM[j]));
        sv = pk_adds(sv, ec[j]);
        const u32 ni = pk_max(pk_adds(Mv[j], tMI[j]), pk_adds(Iv[j], tII[j]));
        Iv[j] = ni; Mv[j] = sv;
        xEv = pk_max(xEv, sv);
        mdv[j] = pk_adds(sv, tMD[j]);
      }
      Dv[0] = stripe_shift16(mdv[Q - 1], hd);
#pragma unroll
      for (int j = 1; j < Q; ++j) Dv[j] = pk_max(mdv[j - 1], pk_adds(Dv[j - 1], tDD[j - 1]));
      u32 carry = pk_adds(Dv[Q - 1], tDD[Q - 1]);
      for (int pass = 0; pass < 32; ++pass) {                       // lazy-F: at most 32 stripes to cross
        u32 cs = stripe_shift16(carry, hc);
        const s16x2 c2 = __builtin_bit_cast(s16x2, cs), d2 = __builtin_bit_cast(s16x2, Dv[0]);
        if (!__any((c2.x > d2.x) || (c2.y > d2.y))) break;
#pragma unroll
        for (int j = 0; j < Q; ++j) { Dv[j] = pk_max(Dv[j], cs); cs = pk_adds(cs, tDD[j]); }
        carry = cs;
      }
    };
    {
      int i = 0;
      for (; i + 1 < Lmax; i += 2) { row(i, e, e2); row(i + 1, e2, e); }
      if (i < Lmax) row(i, e, e2);
    }
    const s16x2 x2 = __builtin_bit_cast(s16x2, xEv);
    int xE = max((int)x2.x, (int)x2.y);
    xE = max(xE, __shfl_xor(xE, 1, 16)); xE = max(xE, __shfl_xor(xE, 2, 16)); xE = max(xE, __shfl_xor(xE, 4, 16)); xE = max(xE, __shfl_xor(xE, 8, 16));
    if (valid && z == 0) {
      const bool overflow = xE >= 32767;
      const int xC = max((int)NEG16, xE + md.wE_move);
      const bool jflag = (xE + md.wE_loop) > xN;
      const uint32_t flag = (jflag && !overflow) ? 1u : 0u;
      float vsc;
      if (overflow) vsc = __builtin_inff();
      else if (xC > NEG16) { float sc = (float)xC + (float)le.w_move - (float)md.base_w; sc = sc / md.scale_w; sc = sc - 3.0f; vsc = sc; }
      else vsc = -__builtin_inff();
      const float v = (vsc - pr.filtersc) * LOG2E_F;
      cd.vit_fast[pi] = vsc; cd.vit_flag[pi] = flag;
      if (v >= md.thr_vit_f2 + cd.margin_vit) pass_to_forward(cd, md, pi, pr.model, pr.seq);
      else if (flag) { cd.route[pi] = 2; queue_push(cd, cd.vxq, CC_VXQ, md.vitx_cls, cd.cap_vq, pi, (uint32_t)CS_VQ); }
      else if (v >= md.thr_vit_f2 - cd.margin_vit) pass_to_forward(cd, md, pi, pr.model, pr.seq);
    }
   }
  }
}

int launch_vit16(int Q, uint32_t nblocks, hipStream_t stream, WorkQueue queue, const PairRec *pairs, const DevModel *models, const LenEntry *lentab,
                 const uint8_t *res, const uint64_t *seq_off, const int32_t *seq_len, const CascadeDev &cd) {
  if (nblocks == 0) return 0;
  switch (Q) {
#define X(QV) case QV: hipLaunchKernelGGL(vit16_kernel<QV>, dim3(nblocks), dim3(256), 0, stream, queue, pairs, models, lentab, res, seq_off, seq_len, cd); break;
    X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(10) X(12) X(14) X(16)
#undef X
    default: return -1;
  }
  return 0;
}

// --------------------------------------------------------------------------------------------
// Exact multi-hit MSV for the pairs SSV cannot decide, packed (round 3): 16 lanes per pair, four pairs per wavefront, SSV's lane mapping
// and the byte recurrence carried in full -- sv = max(prev, xB) + (bias - cost), floored by the clamped add (offset -32768), every row
// ending with the 16-lane maximum that feeds xJ and xB.  The four pairs of a wavefront are whatever the queue holds -- own model, own
// sequence -- so the emission words come from the model's i16 image in global memory (L2: a group's models are few), one row ahead.
// 3 packed ops per register per row + ~25, for four pairs: ~6x fewer instructions than the wave-per-pair msv_full_kernel and no 50 KB
// LDS image per pair; msv_full_kernel stays for models beyond 2048 nodes (no 16-lane image) and for the diagnostics.
// --------------------------------------------------------------------------------------------
constexpr u32 U_ZERO16 = 0x80008000u;
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
constexpr int MSV16_GSTRIDE = 30 * 256;       // bytes per register group of the image: [group][30 symbols][16 lanes][16 B]

template <int Q>
__global__ void __launch_bounds__(256) msv16_kernel(WorkQueue queue, const PairRec *__restrict__ pairs, const DevModel *__restrict__ models,
                                                    const LenEntry *__restrict__ lentab, const uint8_t *__restrict__ res, const uint64_t *__restrict__ seq_off,
                                                    const int32_t *__restrict__ seq_len, CascadeDev cd,
                                                    int32_t *__restrict__ out_xJ /* scores only: xJ (-1 overflow) and the score of entry pi, no decision */, float *__restrict__ out_usc) {
  constexpr int Qg = (Q + 3) / 4;
  const int lane = threadIdx.x & 63, z = lane & 15, g = lane >> 4;
  const uint32_t nqueue = queue_len(queue);
  for (uint32_t q4 = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); q4 * 4u < nqueue; q4 += gridDim.x * (blockDim.x >> 6)) {
    const uint32_t pi = q4 * 4u + (uint32_t)g;
    const bool valid = pi < nqueue;
    const PairRec pr = pairs[valid ? pi : q4 * 4u];
    const DevModel &md = models[pr.model];
    const int L = seq_len[pr.seq];
    int Lmax = __builtin_amdgcn_readlane(L, 0);
    Lmax = max(Lmax, __builtin_amdgcn_readlane(L, 16));
    Lmax = max(Lmax, __builtin_amdgcn_readlane(L, 32));
    Lmax = max(Lmax, __builtin_amdgcn_readlane(L, 48));
    const uint8_t *rp = res + seq_off[pr.seq];
    const LenEntry le = lentab[L];
    const int base = md.base_b, bias = md.bias_b, tec = md.tec_b;
    const int tjbm = (le.tjb_b + md.tbm_b) & 0xff;
    const gp<char> img = gptr(reinterpret_cast<const char *>(md.ssv_tbl)) + z * 16;
    u32 U[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) U[q] = U_ZERO16;
    u32 prev = U_ZERO16;
    int xJ = 0, xB = max(base - tjbm, 0);
    bool overflow = false;
    auto sym = [&](u32 word, int i) -> u32 { return (i < L) ? ((word >> (8 * (i & 3))) & 0xffu) : 29u; };
    auto load_row = [&](u32 (&e)[Qg * 4], u32 x) {
#pragma unroll
      for (int gq = 0; gq < Qg; ++gq) {
        const u32x4v v = *(gp<u32x4v>)(img + (size_t)gq * MSV16_GSTRIDE + (size_t)x * 256u);
        e[gq * 4 + 0] = v.x; e[gq * 4 + 1] = v.y; e[gq * 4 + 2] = v.z; e[gq * 4 + 3] = v.w;
      }
    };
    u32 rw = *reinterpret_cast<const u32 *>(rp);
    u32 ea[Qg * 4], eb[Qg * 4];
    load_row(ea, sym(rw, 0));
    auto row = [&](int i, u32 (&ec)[Qg * 4], u32 (&en)[Qg * 4]) {
      {
        const int i1 = i + 1;
        if ((i1 & 3) == 0 && i1 < L) rw = *reinterpret_cast<const u32 *>(rp + i1);
        load_row(en, sym(rw, i1));
      }
      const u32 xBv = (0x8000u + (u32)xB) * 0x10001u;
      const u32 last = U[Q - 1];
      prev = (u32)__builtin_amdgcn_update_dpp((int)prev, (int)last, 0x111 /*row_shr:1*/, 0xf, 0xf, false);
      const u32 carry = __builtin_amdgcn_alignbit(last, prev, 16);
      u32 xE = U_ZERO16;
#pragma unroll
      for (int q = Q - 1; q >= 1; --q) {
        const u32 v = pk_adds(pk_max(U[q - 1], xBv), ec[q]);
        xE = pk_max(xE, v);
        U[q] = v;
      }
      {
        const u32 v = pk_adds(pk_max(carry, xBv), ec[0]);
        xE = pk_max(xE, v);
        U[0] = v;
      }
      xE = pk_max(xE, (u32)__builtin_amdgcn_update_dpp((int)xE, (int)xE, 0xB1, 0xf, 0xf, false));
      xE = pk_max(xE, (u32)__builtin_amdgcn_update_dpp((int)xE, (int)xE, 0x4E, 0xf, 0xf, false));
      xE = pk_max(xE, (u32)__builtin_amdgcn_update_dpp((int)xE, (int)xE, 0x141, 0xf, 0xf, false));
      xE = pk_max(xE, (u32)__builtin_amdgcn_update_dpp((int)xE, (int)xE, 0x140, 0xf, 0xf, false));
      xE = pk_max(xE, __builtin_amdgcn_alignbit(xE, xE, 16));
      if (i < L) {                                  // (a pair that has ended keeps its score: the pad rows change nothing it reports)
        const int xe = (int)(short)(xE & 0xffffu) + 32768;
        overflow = overflow || (xe + bias >= 255);
        const int xe2 = max(xe - tec, 0);
        xJ = max(xJ, xe2);
        xB = max(max(base, xJ) - tjbm, 0);
      }
    };
    {
      int i = 0;
      for (; i + 1 < Lmax; i += 2) { row(i, ea, eb); row(i + 1, eb, ea); }
      if (i < Lmax) row(i, ea, eb);
    }
    if (valid && z == 0) {
      const float usc = overflow ? __builtin_inff() : msv_score(xJ, le.tjb_b, md.base_b, md.scale_b);
      if (out_usc) { out_xJ[pi] = overflow ? -1 : xJ; out_usc[pi] = usc; }
      else if (to_bits(usc, le.nullsc) >= md.thr_msv_f1) {
        const uint32_t k = atomicAdd(&cd.cnt[CC_CAND], 1u);
        if (k < cd.cap_cand) { PairRec r = pr; r.usc = usc; r.filtersc = 0.f; cd.cand[k] = r; } else atomicOr(&cd.gcnt[CC_STATUS], (uint32_t)CS_CAND);
      }
    }
  }
}

int launch_msv16(int Q, uint32_t nblocks, hipStream_t stream, WorkQueue queue, const PairRec *pairs, const DevModel *models, const LenEntry *lentab,
                 const uint8_t *res, const uint64_t *seq_off, const int32_t *seq_len, const CascadeDev &cd, int32_t *out_xJ, float *out_usc) {
  if (nblocks == 0) return 0;
  switch (Q) {
#define X(QV) case QV: hipLaunchKernelGGL(msv16_kernel<QV>, dim3(nblocks), dim3(256), 0, stream, queue, pairs, models, lentab, res, seq_off, seq_len, cd, out_xJ, out_usc); break;
    X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(32) X(36) X(40) X(48) X(56) X(64)
#undef X
    default: return -1;
  }
  return 0;
}

#define CKM_VIT_CASE(QV) case QV: \
    if (fast) hipLaunchKernelGGL((vit_kernel<QV, true>), dim3(nblocks), dim3(256), 0, stream, queue, pairs, models, lentab, res, seq_off, seq_len, out_xC, out_sc, out_flag, c, cd ? 1 : 0); \
    else hipLaunchKernelGGL((vit_kernel<QV, false>), dim3(nblocks), dim3(256), 0, stream, queue, pairs, models, lentab, res, seq_off, seq_len, out_xC, out_sc, out_flag, c, cd ? 1 : 0); \
    break;
int launch_vit(int QH, uint32_t nblocks, hipStream_t stream, WorkQueue queue, const PairRec *pairs, const DevModel *models,
               const LenEntry *lentab, const uint8_t *res, const uint64_t *seq_off, const int32_t *seq_len, int32_t *out_xC, float *out_sc,
               uint32_t *out_flag, bool fast, const CascadeDev *cd) {
  if (nblocks == 0) return 0;
  CascadeDev c; memset(&c, 0, sizeof(c));
  if (cd) c = *cd;
  switch (QH) {
    CKM_VIT_CASE(1) CKM_VIT_CASE(2) CKM_VIT_CASE(3) CKM_VIT_CASE(4) CKM_VIT_CASE(5) CKM_VIT_CASE(6) CKM_VIT_CASE(7) CKM_VIT_CASE(8)
    CKM_VIT_CASE(10) CKM_VIT_CASE(12) CKM_VIT_CASE(14) CKM_VIT_CASE(16) CKM_VIT_CASE(24) CKM_VIT_CASE(32)
    default: return -1;
  }
  return 0;
}

void launch_msv_full(hipStream_t stream, uint32_t nblocks, WorkQueue queue, const PairRec *pairs, const DevModel *models, const LenEntry *lentab,
                     const uint8_t *res, const uint64_t *seq_off, const int32_t *seq_len, int32_t *out_xJ, float *out_usc, int maxMp,
                     const CascadeDev *cd) {
  if (!nblocks) return;
  CascadeDev c; memset(&c, 0, sizeof(c));
  if (cd) c = *cd;
  const size_t lds = (((size_t)KP_SYMS * (maxMp + 1) + 15) & ~(size_t)15) + (size_t)2 * maxMp * sizeof(int16_t);
  {
    // (searches of several contexts launch from their own host threads: the limit only ever grows, so check + set are one critical section and
    //  a launch that follows sees a limit at least as large as it needs; the attribute belongs to the CURRENT DEVICE's code object, hence per device)
    static std::mutex attr_mutex; static size_t attr_bytes[16] = {0};
    int dev = 0; (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(attr_mutex);
    if (lds > 48 * 1024 && lds > attr_bytes[dev & 15]) { (void)hipFuncSetAttribute((const void *)msv_full_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_bytes[dev & 15] = lds; }
  }
  hipLaunchKernelGGL(msv_full_kernel, dim3(nblocks), dim3(64), lds, stream, queue, pairs, models, lentab, res, seq_off, seq_len, out_xJ, out_usc, maxMp, c, cd ? 1 : 0);
}
void launch_bias(hipStream_t stream, PairRec *pairs, uint32_t npairs, const DevModel *models, const LenEntry *lentab,
                 const uint8_t *res, const uint64_t *seq_off, const int32_t *seq_len, float *raw) {
  if (!npairs) return;
  (void)lentab;
  hipLaunchKernelGGL(bias_kernel, dim3((npairs + 127) / 128), dim3(128), 0, stream, pairs, npairs, models, res, seq_off, seq_len, raw);
}
void launch_bias_filter(hipStream_t stream, uint32_t nblocks, const CascadeDev &cd, const DevModel *models, const LenEntry *lentab,
                        const uint8_t *res, const uint64_t *seq_off) {
  if (nblocks) hipLaunchKernelGGL(bias_filter_kernel, dim3(nblocks), dim3(128), 0, stream, cd, models, lentab, res, seq_off);
}

}  // namespace ckm
