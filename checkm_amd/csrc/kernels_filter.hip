// kernels_filter.hip -- the rest of the acceleration-filter cascade, gfx950 only:
//   msv_finish_kernel  turns SSV maxV into the MSV byte score, applies F1, appends survivors
//   msv_full_kernel    full multi-hit MSV for the rare pairs whose J state could be used
//   bias_kernel        2-state composition filter (Forward, power-of-two rescaling), F1 again, F2 shortcut
//   vit_kernel<Q>      16-bit Viterbi filter, one wavefront per pair, D->D by integer prefix-max scan
// Reference stage being replaced: the MSV -> bias -> Viterbi part of hmmsearch's per-target pipeline
// (process launched at checkm/hmmer.py:70 with the options of checkm/markerGeneFinder.py:141).
// Every decision is taken on IEEE basic operations only (no device libm), so it is bit-identical
// to the host formulation: thresholds were converted to score space on the host (host_profile.cpp).
#include <hip/hip_runtime.h>
#include "dev_types.h"

namespace ckm {

constexpr double LN2D = 0.69314718055994529;
constexpr int NEG16 = -32768;

// --------------------------------------------------------------------------------------------
// MSV finish
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

__global__ void msv_finish_kernel(FinishArgs a, uint32_t nblocks_work) {
  const uint32_t wb = blockIdx.x;
  if (wb >= nblocks_work) return;
  const SsvBlockWork w = a.work[wb];
  const DevModel &md = a.models[w.model];
  for (uint32_t li = threadIdx.x; li < w.count; li += blockDim.x) {
    const uint32_t sid = a.lists[w.list_start + li];
    const int L = a.seq_len[sid];
    if (L <= 0) continue;
    const LenEntry le = a.lentab[L];
    const int maxV = a.maxv[w.pair_start + li];
    const int tjbm = (le.tjb_b + md.tbm_b) & 0xff;
    const int xB = max(md.base_b - tjbm, 0);
    const int xEi = xB + maxV;
    PairRec r; r.model = w.model; r.seq = sid; r.filtersc = 0.f;
    if (maxV == 0) {                             // no cell ever rose above xB: the floored recurrence lost max V; exact kernel
      r.usc = 0.f;
      const uint32_t k = atomicAdd(a.nnores, 1u);
      if (k < a.cap_nores) a.noresult[k] = r;
      continue;
    }
    if (xEi + md.bias_b >= 255) {               // byte overflow: score is +inf, passes every MSV test
      r.usc = __builtin_inff();
      const uint32_t k = atomicAdd(a.nsurv, 1u);
      if (k < a.cap_surv) a.survivors[k] = r;
      continue;
    }
    const int xE = max(xEi, 0);
    const int xJ = max(xE - md.tec_b, 0);
    if (xJ > md.base_b) {                        // J could have been used: exact multi-hit MSV needed
      r.usc = 0.f;
      const uint32_t k = atomicAdd(a.nnores, 1u);
      if (k < a.cap_nores) a.noresult[k] = r;
      continue;
    }
    const float usc = msv_score(xJ, le.tjb_b, md.base_b, md.scale_b);
    if (to_bits(usc, le.nullsc) >= md.thr_msv_f1) {
      r.usc = usc;
      const uint32_t k = atomicAdd(a.nsurv, 1u);
      if (k < a.cap_surv) a.survivors[k] = r;
    }
  }
}

// --------------------------------------------------------------------------------------------
// full multi-hit MSV: one wavefront per pair, cells k = lane + 64*j, row held in LDS as ints
// --------------------------------------------------------------------------------------------
__global__ void msv_full_kernel(const PairRec *__restrict__ pairs, uint32_t npairs, const DevModel *__restrict__ models,
                                const LenEntry *__restrict__ lentab, const uint8_t *__restrict__ res,
                                const uint64_t *__restrict__ seq_off, const int32_t *__restrict__ seq_len,
                                int32_t *__restrict__ out_xJ /* -1 overflow */, float *__restrict__ out_usc, int maxMp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t pi = blockIdx.x * (blockDim.x >> 6) + wave;
  if (pi >= npairs) return;
  int16_t *row = reinterpret_cast<int16_t *>(smem) + (size_t)wave * 2 * maxMp;   // two buffers
  const PairRec pr = pairs[pi];
  const DevModel &md = models[pr.model];
  const int M = md.M, L = seq_len[pr.seq];
  const uint8_t *rp = res + seq_off[pr.seq];
  const LenEntry le = lentab[L];
  const int tjbm = (le.tjb_b + md.tbm_b) & 0xff;
  int16_t *dp = row, *nw = row + maxMp;
  for (int k = lane; k < M; k += 64) dp[k] = 0;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  int xJ = 0, xB = max(md.base_b - tjbm, 0);
  bool overflow = false;
  for (int i = 0; i < L && !overflow; ++i) {
    const uint8_t *cost = md.rbv + (size_t)rp[i] * (M + 1) + 1;
    int xE = 0;
    for (int k = lane; k < M; k += 64) {
      const int mp = (k > 0) ? (int)dp[k - 1] : 0;
      int sv = max(mp, xB);
      sv = min(sv + md.bias_b, 255);
      sv = max(sv - (int)cost[k], 0);
      xE = max(xE, sv);
      nw[k] = (int16_t)sv;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) xE = max(xE, __shfl_xor(xE, s));
    if (min(xE + md.bias_b, 255) == 255) { overflow = true; break; }
    xE = max(xE - md.tec_b, 0);
    xJ = max(xJ, xE);
    xB = max(max(md.base_b, xJ) - tjbm, 0);
    int16_t *t = dp; dp = nw; nw = t;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) {
    if (overflow) { out_xJ[pi] = -1; out_usc[pi] = __builtin_inff(); }
    else { out_xJ[pi] = xJ; out_usc[pi] = msv_score(xJ, le.tjb_b, md.base_b, md.scale_b); }
  }
}

// --------------------------------------------------------------------------------------------
// bias filter: one thread per pair
// --------------------------------------------------------------------------------------------
// out flags: bit0 pass F1 after bias; bit1 needs Viterbi (P > F2)
__global__ void bias_kernel(PairRec *__restrict__ pairs, uint32_t npairs, const DevModel *__restrict__ models,
                            const LenEntry *__restrict__ lentab, const uint8_t *__restrict__ res,
                            const uint64_t *__restrict__ seq_off, const int32_t *__restrict__ seq_len,
                            uint8_t *__restrict__ flags, float *__restrict__ dbg_d /* optional [npairs*3]: d0 d1 nexp */) {
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
  // the log of (d0+d1) is taken on the host (same libm as every other score); here only the raw state
  dbg_d[(size_t)pi * 3 + 0] = d0 + d1;
  dbg_d[(size_t)pi * 3 + 1] = (float)nexp;
  dbg_d[(size_t)pi * 3 + 2] = 0.f;
  (void)flags; (void)lentab;
}

// --------------------------------------------------------------------------------------------
// Viterbi filter: one wavefront per pair; lane z owns cells c = z*Q+q (node k = c+1)
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ int sat_lo(int v) { return max(v, NEG16); }

template <int Q>
__global__ void __launch_bounds__(256) vit_kernel(const PairRec *__restrict__ pairs, const uint32_t *__restrict__ idx, uint32_t n,
                                                  const DevModel *__restrict__ models, const LenEntry *__restrict__ lentab,
                                                  const uint8_t *__restrict__ res, const uint64_t *__restrict__ seq_off,
                                                  const int32_t *__restrict__ seq_len, int32_t *__restrict__ out_xC, float *__restrict__ out_sc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t wi = blockIdx.x * (blockDim.x >> 6) + wave;
  if (wi >= n) return;
  const uint32_t pi = idx[wi];
  const PairRec pr = pairs[pi];
  const DevModel &md = models[pr.model];
  constexpr int Mp = Q * 64;
  const int L = seq_len[pr.seq];
  const uint8_t *rp = res + seq_off[pr.seq];
  const LenEntry le = lentab[L];
  const int c0 = lane * Q;
  int tBM[Q], tMM[Q], tIM[Q], tDM[Q], tMD[Q], tMI[Q], tII[Q], Cc[Q], Cn[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int c = c0 + q;
    tBM[q] = md.wtr[0 * Mp + c]; tMM[q] = md.wtr[1 * Mp + c]; tIM[q] = md.wtr[2 * Mp + c]; tDM[q] = md.wtr[3 * Mp + c];
    tMD[q] = md.wtr[4 * Mp + c]; tMI[q] = md.wtr[5 * Mp + c]; tII[q] = md.wtr[6 * Mp + c];
    Cc[q] = md.wddc[c]; Cn[q] = md.wddc[c + 1];
  }
  int Mv[Q], Iv[Q], Dv[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) Mv[q] = Iv[q] = Dv[q] = NEG16;
  int xN = md.base_w, xB = xN + le.w_move, xJ = NEG16, xC = NEG16;
  bool overflow = false;
  constexpr int NEGBIG = -(1 << 30);
  // emission words are fetched one row ahead, the residue byte two rows ahead (dependent loads)
  int e[Q];
  {
    const int16_t *__restrict__ er = md.rwv + (size_t)rp[0] * Mp + lane;
#pragma unroll
    for (int q = 0; q < Q; ++q) e[q] = er[q * 64];
  }
  int xn = (L > 1) ? rp[1] : rp[0];
  for (int i = 0; i < L; ++i) {
    int en[Q];
    {
      const int16_t *__restrict__ er = md.rwv + (size_t)xn * Mp + lane;
#pragma unroll
      for (int q = 0; q < Q; ++q) en[q] = er[q * 64];
    }
    xn = (i + 2 < L) ? rp[i + 2] : rp[L - 1];
    int mpi = __shfl_up(Mv[Q - 1], 1), ipi = __shfl_up(Iv[Q - 1], 1), dpi = __shfl_up(Dv[Q - 1], 1);
    if (lane == 0) { mpi = NEG16; ipi = NEG16; dpi = NEG16; }
    int xE = NEG16;
#pragma unroll
    for (int q = Q - 1; q >= 0; --q) {
      const int mp = q ? Mv[q - 1] : mpi, ip = q ? Iv[q - 1] : ipi, dp = q ? Dv[q - 1] : dpi;
      int sv = sat_lo(xB + tBM[q]);
      sv = max(sv, sat_lo(mp + tMM[q]));
      sv = max(sv, sat_lo(ip + tIM[q]));
      sv = max(sv, sat_lo(dp + tDM[q]));
      sv = sat_lo(sv + e[q]);
      const int ni = max(sat_lo(Mv[q] + tMI[q]), sat_lo(Iv[q] + tII[q]));
      Iv[q] = ni; Mv[q] = sv;
      xE = max(xE, sv);
    }
    // D(c) = max(-32768, C[c] + max_{j<c}(md(j) - C[j+1])),  md(j) = sat(M(j) + tMD(j))
    int g[Q]; int run = NEGBIG;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int mdv = sat_lo(Mv[q] + tMD[q]);
      g[q] = (mdv <= NEG16) ? NEGBIG : mdv - Cn[q];
      run = max(run, g[q]);
    }
    int incl = run;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) { const int o = __shfl_up(incl, s); if (lane >= s) incl = max(incl, o); }
    int excl = __shfl_up(incl, 1); if (lane == 0) excl = NEGBIG;
    int pm = excl;
#pragma unroll
    for (int q = 0; q < Q; ++q) { Dv[q] = (pm <= NEGBIG) ? NEG16 : sat_lo(Cc[q] + pm); pm = max(pm, g[q]); }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) xE = max(xE, __shfl_xor(xE, s));
    if (xE >= 32767) { overflow = true; break; }
    xC = max(xC, xE + md.wE_move);
    xJ = max(xJ, xE + md.wE_loop);
    xB = max(xJ + le.w_move, xN + le.w_move);
#pragma unroll
    for (int q = 0; q < Q; ++q) e[q] = en[q];
  }
  if (lane == 0) {
    if (overflow) { out_xC[pi] = 32767; out_sc[pi] = __builtin_inff(); }
    else {
      out_xC[pi] = xC;
      if (xC > NEG16) { float sc = (float)xC + (float)le.w_move - (float)md.base_w; sc = sc / md.scale_w; sc = sc - 3.0f; out_sc[pi] = sc; }
      else out_sc[pi] = -__builtin_inff();
    }
  }
}

#define CKM_VIT_CASE(QV) case QV: hipLaunchKernelGGL(vit_kernel<QV>, dim3((n + 3) / 4), dim3(256), 0, stream, pairs, idx, n, models, lentab, res, seq_off, seq_len, out_xC, out_sc); break;
int launch_vit(int Q, hipStream_t stream, const PairRec *pairs, const uint32_t *idx, uint32_t n, const DevModel *models,
               const LenEntry *lentab, const uint8_t *res, const uint64_t *seq_off, const int32_t *seq_len, int32_t *out_xC, float *out_sc) {
  if (n == 0) return 0;
  switch (Q) {
    CKM_VIT_CASE(1) CKM_VIT_CASE(2) CKM_VIT_CASE(3) CKM_VIT_CASE(4) CKM_VIT_CASE(6) CKM_VIT_CASE(8)
    CKM_VIT_CASE(12) CKM_VIT_CASE(16) CKM_VIT_CASE(24) CKM_VIT_CASE(32)
    default: return -1;
  }
  return 0;
}

void launch_msv_finish(hipStream_t stream, const FinishArgs &a, uint32_t nblocks) {
  if (nblocks) hipLaunchKernelGGL(msv_finish_kernel, dim3(nblocks), dim3(256), 0, stream, a, nblocks);
}
void launch_msv_full(hipStream_t stream, const PairRec *pairs, uint32_t npairs, const DevModel *models, const LenEntry *lentab,
                     const uint8_t *res, const uint64_t *seq_off, const int32_t *seq_len, int32_t *out_xJ, float *out_usc, int maxMp) {
  if (!npairs) return;
  hipLaunchKernelGGL(msv_full_kernel, dim3((npairs + 3) / 4), dim3(256), (size_t)4 * 2 * maxMp * sizeof(int16_t), stream,
                     pairs, npairs, models, lentab, res, seq_off, seq_len, out_xJ, out_usc, maxMp);
}
void launch_bias(hipStream_t stream, PairRec *pairs, uint32_t npairs, const DevModel *models, const LenEntry *lentab,
                 const uint8_t *res, const uint64_t *seq_off, const int32_t *seq_len, float *raw) {
  if (!npairs) return;
  hipLaunchKernelGGL(bias_kernel, dim3((npairs + 127) / 128), dim3(128), 0, stream, pairs, npairs, models, lentab, res, seq_off, seq_len,
                     (uint8_t *)nullptr, raw);
}

}  // namespace ckm
