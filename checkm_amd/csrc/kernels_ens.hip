// kernels_ens.hip -- multi-domain regions: stochastic trace ensemble of a region's Forward matrix, gfx950 only.
// Replaces, inside the hmmsearch process launched at checkm/hmmer.py:70, HMMER's "region_trace_ensemble":
// 200 stochastic tracebacks, null2 by trace, per-position null2 odds; the sampled segments go back to the host,
// which clusters them (integer work on a few hundred segments).
//   ens_trace_seq_kernel  (default) one WAVEFRONT per region: the 200 traces of a region run one after the other on ONE generator
//                      stream, re-seeded for the region and carried from trace to trace exactly as hmmsearch carries its generator
//                      (the number of draws a trace takes is only known when it ends).  The parallelism is over REGIONS.
//   ens_trace_kernel   (CKM_ENS_STREAM=substream, opt-in) one workgroup per region, one LANE per trace: every trace draws from
//                      its own substream of the generator -- independent lanes, but not hmmsearch's stream.
//   ens_null2_kernel   one wavefront per (region, trace): state usage counts in LDS, null2 odds in the canonical
//                      64-lane order, per-position odds ratio of this trace.
//   ens_sum_kernel     one thread per region position: sum of the ratios over traces, trace order.
// Float results follow the CPU restatement (trace_ensemble in the test oracle) operation by operation; -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include "dev_types.h"
#include "xlane.h"

namespace ckm {

typedef float f32x4 __attribute__((ext_vector_type(4)));     // a node of a cell-major region row: {M, I, D, 0}

constexpr int ENS_N = ENS_NSAMPLES;

__device__ __forceinline__ double ens_roll(uint32_t &x) { x = x * 69069u + 1u; return (double)x / 4294967296.0; }

// first index whose cumulative weight exceeds roll * total (weights summed in index order)
__device__ __forceinline__ int ens_choose(double roll, const float *pth, int n) {
  float norm = pth[0];
  for (int i = 1; i < n; ++i) norm = norm + pth[i];
  if (!(norm > 0.0f)) return 0;
  const double target = roll * (double)norm;
  double sum = 0.0;
  for (int i = 0; i < n; ++i) { sum += (double)pth[i]; if (target < sum) return i; }
  for (int i = n - 1; i > 0; --i) if (pth[i] > 0.0f) return i;
  return 0;
}

// E-state choice, evaluated by the whole wavefront for one of its traces.  Canonical order: lane z owns cells
// z*Q .. z*Q+Q-1 (Q consecutive float4 of the cell-major row); weights in cell order are M(c) then D(c); local[z] = sequential double sum of the lane's 2Q weights;
// base[z] = local[0] + .. + local[z-1] accumulated in lane order; total = base[64]; the choice is the first position
// (cell order) whose cumulative weight base[z] + running sum exceeds roll * total; none -> node 1, match.
__device__ __forceinline__ int ens_select_e(const f32x4 *__restrict__ row4 /* the row's Mp nodes {M, I, D, 0} */, int Q, double roll, int lane) {
  const f32x4 *__restrict__ mine = row4 + lane * Q;
  double local = 0.0;
  for (int q = 0; q < Q; ++q) { const f32x4 v = mine[q]; local += (double)v.x; local += (double)v.z; }
  double base = 0.0, run = 0.0;
#pragma unroll
  for (int z = 0; z < 64; ++z) { const double tz = __shfl(local, z); if (lane == z) base = run; run += tz; }
  const double target = roll * run;
  int found = -1;
  double acc = 0.0;
  for (int q = 0; q < Q; ++q) {
    const f32x4 v = mine[q];
    acc += (double)v.x; if (found < 0 && target < base + acc) found = 2 * q;
    acc += (double)v.z; if (found < 0 && target < base + acc) found = 2 * q + 1;
  }
  const unsigned long long any = __ballot(found >= 0);
  if (!any) return 0;                                   // cell 0, match
  const int zf = __ffsll((long long)any) - 1;
  const int f = __shfl(found, zf);
  return ((zf * Q + (f >> 1)) << 1) | (f & 1);          // (cell << 1) | is_delete
}

__global__ void __launch_bounds__(256) ens_trace_kernel(const EnsWork *__restrict__ work, const uint32_t *__restrict__ list, const DevModel *__restrict__ models,
                                                       const LenEntry *__restrict__ lentab, float *__restrict__ ws,
                                                       const uint32_t *__restrict__ seeds, const uint32_t *__restrict__ count, uint32_t cap) {
 const uint32_t nregions = min(*count, cap);
 for (uint32_t region = blockIdx.x; region < nregions; region += gridDim.x) {
  const EnsWork w = work[list ? list[region] : region];
  const int t = threadIdx.x, lane = threadIdx.x & 63;
  const DevModel &md = models[w.model];
  const int Q = md.fbQ, Mp = Q * 64, Ld = w.Ld;
  const size_t rowsz = (size_t)4 * Mp;                  // cell-major rows: f32x4 {M, I, D, 0} per node
  const float *__restrict__ mx = ws + w.mx_off;
  const float *__restrict__ xs = ws + w.xs_off;
  const gp<float> ftr = gptr(md.ftr);
  const gp<float> tBM = ftr, tMM = ftr + Mp, tIM = ftr + 2 * Mp, tDM = ftr + 3 * Mp,
                  tMI = ftr + 4 * Mp, tII = ftr + 5 * Mp, tMD = ftr + 6 * Mp, tDD = ftr + 7 * Mp;
  const LenEntry le = lentab[w.Lcfg];
  const float loop = le.loop_m, move = le.move_m, Eloop = md.fE_loop, Emove = md.fE_move;
  const bool live = t < ENS_N;
  uint16_t *__restrict__ code = reinterpret_cast<uint16_t *>(ws + w.code_off) + (size_t)(live ? t : 0) * (Ld + 1);
  int32_t *__restrict__ seg = reinterpret_cast<int32_t *>(ws + w.seg_off) + (size_t)(live ? t : 0) * w.cap * 4;
  int32_t *__restrict__ nsegp = reinterpret_cast<int32_t *>(ws + w.nseg_off);
#define CELL(c) (4 * (c))
  enum { sC, sE, sM, sI, sD, sB, sJ, sN };
  const bool mine = live;
  uint32_t rng = seeds[live ? t : 0];
  int st = sC, i = Ld, k = 0, nseg = 0, sqto = 0, hmmto = 0;
  bool overflow = false, done = !mine;
  float pth[4];
  for (;;) {
    // E-state choices are made by the whole wavefront, one requesting trace at a time (uniform control flow)
    unsigned long long need = __ballot(!done && st == sE);
    while (need) {
      const int l = __ffsll((long long)need) - 1;
      need &= need - 1;
      uint32_t xr = rng * 69069u + 1u;
      xr = __shfl(xr, l);
      const int row = __shfl(i, l);
      const int r = ens_select_e(reinterpret_cast<const f32x4 *>(mx + rowsz * row), Q, (double)xr / 4294967296.0, lane);
      if (lane == l) { rng = xr; k = (r >> 1) + 1; st = (r & 1) ? sD : sM; sqto = 0; hmmto = 0; }    // (coordinates come from the first MATCH state met on the way back)
    }
    if (!done) {
      const float *cr = mx + rowsz * i, *pr = (i > 0) ? mx + rowsz * (i - 1) : mx;
      switch (st) {
      case sC:
        pth[0] = xs[(size_t)(i - 1) * 6 + 4] * loop;
        pth[1] = (xs[(size_t)i * 6 + 0] * Emove) * xs[(size_t)i * 6 + 5];
        if (ens_choose(ens_roll(rng), pth, 2) == 0) { code[i] = 0; --i; } else st = sE;
        break;
      case sJ:
        pth[0] = xs[(size_t)(i - 1) * 6 + 2] * loop;
        pth[1] = (xs[(size_t)i * 6 + 0] * Eloop) * xs[(size_t)i * 6 + 5];
        if (ens_choose(ens_roll(rng), pth, 2) == 0) { code[i] = 0; --i; } else st = sE;
        break;
      case sM: {
        const int c = k - 1;
        code[i] = (uint16_t)(0x4000 | k);
        if (!sqto) { sqto = i; hmmto = k; }      // HMMER's p7_trace_Index takes sqto/hmmto from the last M state; trailing D states do not count
        pth[0] = xs[(size_t)(i - 1) * 6 + 3] * tBM[c];
        if (c > 0) { const int a = CELL(c - 1); pth[1] = pr[a] * tMM[c]; pth[2] = pr[a + 1] * tIM[c]; pth[3] = pr[a + 2] * tDM[c]; }
        else pth[1] = pth[2] = pth[3] = 0.0f;
        const int ch = ens_choose(ens_roll(rng), pth, 4);
        if (ch == 0) {
          if (nseg == w.cap) { overflow = true; done = true; break; }
          seg[nseg * 4 + 0] = i; seg[nseg * 4 + 1] = sqto; seg[nseg * 4 + 2] = k; seg[nseg * 4 + 3] = hmmto; ++nseg;
          st = sB;
        } else st = (ch == 1) ? sM : (ch == 2) ? sI : sD;
        --i; --k;
      } break;
      case sI: {
        const int a = CELL(k - 1);
        code[i] = (uint16_t)(0x8000 | k);
        pth[0] = pr[a] * tMI[k - 1]; pth[1] = pr[a + 1] * tII[k - 1];
        st = (ens_choose(ens_roll(rng), pth, 2) == 0) ? sM : sI;
        --i;
      } break;
      case sD: {
        const int c = k - 1;
        if (c > 0) { const int a = CELL(c - 1); pth[0] = cr[a] * tMD[c - 1]; pth[1] = cr[a + 2] * tDD[c - 1]; } else pth[0] = pth[1] = 0.0f;
        st = (ens_choose(ens_roll(rng), pth, 2) == 0) ? sM : sD;
        --k;
      } break;
      case sB:
        pth[0] = xs[(size_t)i * 6 + 1] * move; pth[1] = xs[(size_t)i * 6 + 2] * move;
        st = (ens_choose(ens_roll(rng), pth, 2) == 0) ? sN : sJ;
        break;
      default:   // sN: the rest of the region is flank
        done = true;
        break;
      }
      // a numerically impossible move ends the trace
      if (i < 0 || k < 0 || ((st == sM || st == sI) && (k < 1 || i < 1)) || (st == sD && k < 1) || ((st == sC || st == sJ || st == sE) && i < 1)) done = true;
    }
    if (!__ballot(!done)) break;
  }
  if (mine) {
    for (; i >= 1; --i) code[i] = 0;
    nsegp[t] = overflow ? -1 : nseg;
  }
 }
#undef CELL
}

// ---- default: ONE generator stream per region (hmmsearch's own use of its generator) ----------------------------------------------
// One wavefront per region; the 200 traces of a region run one after the other on one generator stream, re-seeded for the region and
// carried from trace to trace (every state of a trace but N takes exactly one draw).  A single walk is a chain of dependent choices, but
// most of it is RUNS whose states are known in advance as long as every choice comes out the likely way:
//   * a run of match states down a diagonal: M(i,k) -> M(i-1,k-1) -> ...  Lane l evaluates the step at (i-l, k-l) with the draw the
//     sequential walk would use there (the generator is linear: x[n+l+1] = A^(l+1) x[n] + C(l+1), one multiply-add per lane), on operands
//     it GATHERS itself -- the region's Forward matrix is cell-major (kernels_fb.hip writes float4 {M, I, D, 0} per node), so a step's
//     three predecessor values are one 16-byte load; the first lane whose choice is not "match again" ends the run, and the walk advances
//     by that many steps at once.  Exactly the sequential walk: each lane's step only assumes that all earlier steps stayed on the
//     diagonal, which is what the first deviating lane decides.  One memory round trip per up to 64 steps;
//   * a run of C (or J) states over the residues outside the domains, the same way.
// Insert / delete / begin states and the E state's choice over the 2M exit weights of a row (ens_select_e, all lanes) are single steps.
// (History of this kernel, per launch = per longest region: one step at a time from global memory 96 ms (up to 400); one step at a time
// from LDS windows 45 ms; lockstep runs over LDS windows 33 rows deep ~25 ms -- the windows cost more than they saved: a diagonal uses
// 33 of the 2112 nodes staged.)
__device__ __forceinline__ int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }

// first index whose cumulative weight exceeds roll * total (ens_choose's order of operations: float sum of the weights, double running sum)
__device__ __forceinline__ int ens_choice4(float p0, float p1, float p2, float p3, double roll) {
  float norm = p0 + p1; norm = norm + p2; norm = norm + p3;
  if (!(norm > 0.0f)) return 0;
  const double target = roll * (double)norm;
  const double s0 = (double)p0, s1 = s0 + (double)p1, s2 = s1 + (double)p2, s3 = s2 + (double)p3;
  return (target < s0) ? 0 : (target < s1) ? 1 : (target < s2) ? 2 : (target < s3) ? 3 : (p3 > 0.0f) ? 3 : (p2 > 0.0f) ? 2 : (p1 > 0.0f) ? 1 : 0;
}
__device__ __forceinline__ int ens_choice2(float p0, float p1, double roll) {
  const float norm = p0 + p1;
  if (!(norm > 0.0f)) return 0;
  const double target = roll * (double)norm;
  const double s0 = (double)p0, s1 = s0 + (double)p1;
  return (target < s0) ? 0 : (target < s1) ? 1 : (p1 > 0.0f) ? 1 : 0;
}

__global__ void __launch_bounds__(64) ens_trace_seq_kernel(const EnsWork *__restrict__ work, const uint32_t *__restrict__ list, const DevModel *__restrict__ models,
                                                          const LenEntry *__restrict__ lentab, float *__restrict__ ws,
                                                          const uint32_t *__restrict__ seeds, const uint32_t *__restrict__ count, uint32_t cap) {
 const uint32_t nregions = min(*count, cap);
 const int lane = threadIdx.x;
 __builtin_amdgcn_s_setprio(3);          // one latency-bound wavefront among the VALU-bound ones of the SSV launches
 // generator jump of this lane: x[n + lane + 1] = jA * x[n] + jC
 uint32_t jA = 69069u, jC = 1u;
 for (int l = 0; l < lane; ++l) { jC = jC * 69069u + 1u; jA = jA * 69069u; }
 for (uint32_t region = blockIdx.x; region < nregions; region += gridDim.x) {
  const EnsWork w = work[list ? list[region] : region];
  const DevModel &md = models[w.model];
  const int Q = uni_i(md.fbQ), Mp = Q * 64, Ld = uni_i(w.Ld), segcap = uni_i(w.cap);
  const gp<f32x4> mx4 = (gp<f32x4>)gptr(ws + w.mx_off);
  const gp<float> xs = gptr(ws + w.xs_off);
  const gp<float> ftr = gptr(md.ftr);
  const gp<float> tBM = ftr, tMM = ftr + Mp, tIM = ftr + 2 * Mp, tDM = ftr + 3 * Mp, tMI = ftr + 4 * Mp, tII = ftr + 5 * Mp, tMD = ftr + 6 * Mp, tDD = ftr + 7 * Mp;
  const LenEntry le = lentab[w.Lcfg];
  const float loop = le.loop_m, move = le.move_m, Eloop = md.fE_loop, Emove = md.fE_move;
  uint16_t *__restrict__ codes = reinterpret_cast<uint16_t *>(ws + w.code_off);
  int32_t *__restrict__ segs = reinterpret_cast<int32_t *>(ws + w.seg_off);
  int32_t *__restrict__ nsegp = reinterpret_cast<int32_t *>(ws + w.nseg_off);
  enum { sC, sE, sM, sI, sD, sB, sJ, sN };
  uint32_t rng = (uint32_t)uni_i((int)seeds[0]);            // the region's stream: re-seeded here, carried from trace to trace below
#define IMPOSSIBLE() (i < 0 || k < 0 || ((st == sM || st == sI) && (k < 1 || i < 1)) || (st == sD && k < 1) || ((st == sC || st == sJ || st == sE) && i < 1))
  for (int t = 0; t < ENS_N; ++t) {
    uint16_t *__restrict__ code = codes + (size_t)t * (Ld + 1);
    int32_t *__restrict__ seg = segs + (size_t)t * segcap * 4;
    int st = sC, i = Ld, k = 0, nseg = 0, sqto = 0, hmmto = 0;
    bool overflow = false;
    for (;;) {
      // ---- a run of match states down the diagonal, one step per lane ----
      if (st == sM && k >= 2 && i >= 1) {
        const int R = min(min(i, k - 1), 64);                 // steps that stay inside the matrix with a predecessor node (k - l >= 2)
        const bool in = lane < R;
        const int l = in ? lane : 0;
        const int ri = i - l, c = k - 1 - l;                  // lane l's state: M(ri, c + 1)
        const f32x4 g = mx4[(size_t)(ri - 1) * Mp + (c - 1)];
        const float xB = xs[(size_t)(ri - 1) * 6 + 3];
        const float p0 = xB * tBM[c], p1 = g.x * tMM[c], p2 = g.y * tIM[c], p3 = g.z * tDM[c];
        const uint32_t x = jA * rng + jC;
        const int ch = ens_choice4(p0, p1, p2, p3, (double)x / 4294967296.0);
        const unsigned long long stay = __ballot(in && ch == 1);
        const int r = (stay == ~0ull) ? 64 : (int)__builtin_ctzll(~stay);      // first lane that does not continue the run (r >= R: all R stayed)
        const int last = min(r, R - 1);                       // lane of the last step taken
        if (!sqto) { sqto = i; hmmto = k; }
        if (lane <= last) code[i - lane] = (uint16_t)(0x4000u | (uint32_t)(k - lane));
        rng = (uint32_t)__builtin_amdgcn_readlane((int)x, last);
        if (r < R) {
          const int chr = __builtin_amdgcn_readlane(ch, r);
          const int ir = i - r, kr = k - r;
          if (chr == 0) {
            if (nseg == segcap) { overflow = true; break; }
            if (lane == 0) { seg[nseg * 4 + 0] = ir; seg[nseg * 4 + 1] = sqto; seg[nseg * 4 + 2] = kr; seg[nseg * 4 + 3] = hmmto; }
            ++nseg;
            st = sB;
          } else st = (chr == 2) ? sI : sD;
          i = ir - 1; k = kr - 1;
        } else { i -= R; k -= R; }
        if (IMPOSSIBLE()) break;
        continue;
      }
      // ---- a run of C (or J) states over the special rows ----
      if ((st == sC || st == sJ) && i >= 1) {
        const int R = min(i, 64);
        const bool in = lane < R;
        const int l = in ? lane : 0;
        const int col = (st == sC) ? 4 : 2; const float em = (st == sC) ? Emove : Eloop;
        const gp<float> X1 = xs + (size_t)(i - l) * 6;
        const uint32_t x = jA * rng + jC;
        const int ch = ens_choice2(X1[col - 6] * loop, (X1[0] * em) * X1[5], (double)x / 4294967296.0);
        const unsigned long long stay = __ballot(in && ch == 0);
        const int r = (stay == ~0ull) ? 64 : (int)__builtin_ctzll(~stay);
        const int nstay = min(r, R);
        if (lane < nstay) code[i - lane] = 0;
        rng = (uint32_t)__builtin_amdgcn_readlane((int)x, min(r, R - 1));
        i -= nstay;
        if (r < R) st = sE;
        if (IMPOSSIBLE()) break;
        continue;
      }
      if (st == sN) break;
      if (st == sE) {
        rng = rng * 69069u + 1u;
        const int r = uni_i(ens_select_e((const f32x4 *)(mx4 + (size_t)i * Mp), Q, (double)rng / 4294967296.0, lane));
        k = (r >> 1) + 1; st = (r & 1) ? sD : sM; sqto = 0; hmmto = 0;
      } else {
        // single steps: M at node 1, I, D, B
        float pth0 = 0.0f, pth1 = 0.0f;
        if (st == sM) {                                      // k == 1: no predecessor node, only the entry from B
          pth0 = xs[(size_t)(i - 1) * 6 + 3] * tBM[0];
        } else if (st == sI) {
          const f32x4 g = mx4[(size_t)(i - 1) * Mp + (k - 1)];
          pth0 = g.x * tMI[k - 1]; pth1 = g.y * tII[k - 1];
        } else if (st == sD) {
          if (k > 1) { const f32x4 g = mx4[(size_t)i * Mp + (k - 2)]; pth0 = g.x * tMD[k - 2]; pth1 = g.z * tDD[k - 2]; }
        } else {
          pth0 = xs[(size_t)i * 6 + 1] * move; pth1 = xs[(size_t)i * 6 + 2] * move;
        }
        rng = rng * 69069u + 1u;
        const double roll = (double)rng / 4294967296.0;
        int ch = (st == sM) ? ens_choice4(pth0, 0.0f, 0.0f, 0.0f, roll) : ens_choice2(pth0, pth1, roll);
        ch = uni_i(ch);
        if (st == sM) {
          if (lane == 0) code[i] = (uint16_t)(0x4000u | (uint32_t)k);
          if (!sqto) { sqto = i; hmmto = k; }
          if (ch == 0) {
            if (nseg == segcap) { overflow = true; break; }
            if (lane == 0) { seg[nseg * 4 + 0] = i; seg[nseg * 4 + 1] = sqto; seg[nseg * 4 + 2] = k; seg[nseg * 4 + 3] = hmmto; }
            ++nseg;
            st = sB;
          } else st = (ch == 1) ? sM : (ch == 2) ? sI : sD;
          --i; --k;
        } else if (st == sI) {
          if (lane == 0) code[i] = (uint16_t)(0x8000u | (uint32_t)k);
          st = (ch == 0) ? sM : sI;
          --i;
        } else if (st == sD) {
          st = (ch == 0) ? sM : sD;
          --k;
        } else {
          st = (ch == 0) ? sN : sJ;
        }
      }
      // a numerically impossible move ends the trace
      if (IMPOSSIBLE()) break;
    }
    // residues i .. 1 lie outside every domain: code 0
    for (int pos_ = 1 + lane; pos_ <= i; pos_ += 64) code[pos_] = 0;
    if (lane == 0) nsegp[t] = overflow ? -1 : nseg;
  }
#undef IMPOSSIBLE
 }
}

// grid (ENS_N, nregions), 64 threads; dynamic LDS: 2*Mp counters/floats + 32 floats
__global__ void __launch_bounds__(64) ens_null2_kernel(const EnsWork *__restrict__ work, const uint32_t *__restrict__ list, const DevModel *__restrict__ models,
                                                      const uint8_t *__restrict__ res, const uint64_t *__restrict__ seq_off,
                                                      float *__restrict__ ws, const uint32_t *__restrict__ count, uint32_t cap) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
 const uint32_t nregions = min(*count, cap);
 for (uint32_t region = blockIdx.y; region < nregions; region += gridDim.y) {
  __syncthreads();
  const EnsWork w = work[list ? list[region] : region];
  const int t = blockIdx.x, lane = threadIdx.x;
  const DevModel &md = models[w.model];
  const int Q = md.fbQ, Mp = Q * 64, Ld = w.Ld;
  uint32_t *cnt = reinterpret_cast<uint32_t *>(lds);
  float *n2 = lds + 2 * Mp;
  const uint16_t *__restrict__ code = reinterpret_cast<const uint16_t *>(ws + w.code_off) + (size_t)t * (Ld + 1);
  const int32_t *__restrict__ seg = reinterpret_cast<const int32_t *>(ws + w.seg_off) + (size_t)t * w.cap * 4;
  const int ns = reinterpret_cast<const int32_t *>(ws + w.nseg_off)[t];
  float *__restrict__ rt = ws + w.ratio_off + (size_t)t * (Ld + 1);
  const uint8_t *__restrict__ rd = res + seq_off[w.seq] + w.i0;
  for (int pos = 1 + lane; pos <= Ld; pos += 64) rt[pos] = 1.0f;
  for (int d = 0; d < ns; ++d) {
    const int sqfrom = seg[d * 4 + 0], sqto = seg[d * 4 + 1];
    for (int c = lane; c < 2 * Mp; c += 64) cnt[c] = 0u;
    __syncthreads();
    int nemit = 0;
    for (int pos = sqfrom + lane; pos <= sqto; pos += 64) {
      const uint16_t cd = code[pos]; const int kk = cd & 0x3fff;
      if (cd & 0x4000) { atomicAdd(&cnt[kk - 1], 1u); ++nemit; } else if (cd & 0x8000) { atomicAdd(&cnt[Mp + kk - 1], 1u); ++nemit; }
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) nemit += __shfl_xor(nemit, s);
    __syncthreads();
    const float norm = 1.0f / (float)nemit;
    for (int c = lane; c < 2 * Mp; c += 64) { const float v = (float)cnt[c] * norm; lds[c] = v; }
    __syncthreads();
    for (int x = 0; x < 20; ++x) {
      const gp<float> rfx = gptr(md.rf) + (size_t)x * Mp + lane;
      float s = 0.f;
      for (int q = 0; q < Q; ++q) { const int idx = lane * Q + q; const float tt = lds[idx] * rfx[q * 64]; s = s + tt; s = s + lds[Mp + idx]; }
      s = wave_sum(s);
      if (lane == 0) n2[x] = s + 0.0f;
    }
    __syncthreads();
    if (lane == 0) {
      // degenerate symbols: plain average over their residues (alphabet order ACDEFGHIKLMNPQRSTVWY-BJZOUX*~)
      const float B = (n2[2] + n2[11]) / 2.0f, J = (n2[7] + n2[9]) / 2.0f, Z = (n2[3] + n2[13]) / 2.0f;
      float r = 0.f;
      for (int y = 0; y < 20; ++y) r += n2[y];
      n2[20] = 1.0f; n2[21] = B; n2[22] = J; n2[23] = Z; n2[24] = n2[8] / 1.0f; n2[25] = n2[1] / 1.0f; n2[26] = r / 20.0f;
      n2[27] = 1.0f; n2[28] = 1.0f; n2[29] = 1.0f;
    }
    __syncthreads();
    for (int pos = sqfrom + 1 + lane; pos <= sqto; pos += 64) rt[pos] = n2[rd[pos - 1]];
    __syncthreads();
  }
 }
}

// grid (x, regions): one thread per region position (strided), sum over the traces in trace order; then the region's results
// (200 counts, the segment table, the sums: contiguous in the workspace) are copied to `host_res` when the region has a place there
__global__ void __launch_bounds__(256) ens_sum_kernel(const EnsWork *__restrict__ work, const uint32_t *__restrict__ list, float *__restrict__ ws, const uint32_t *__restrict__ count, uint32_t cap) {
  const uint32_t nregions = min(*count, cap);
  for (uint32_t region = blockIdx.y; region < nregions; region += gridDim.y) {
    const EnsWork w = work[list ? list[region] : region];
    for (int pos = 1 + blockIdx.x * 256 + threadIdx.x; pos <= w.Ld; pos += gridDim.x * 256) {
      const float *__restrict__ rt = ws + w.ratio_off + pos;
      float acc = 0.0f;
      for (int t = 0; t < ENS_N; ++t) acc = acc + rt[(size_t)t * (w.Ld + 1)];
      (ws + w.n2_off)[pos - 1] = acc;
    }
  }
}

__global__ void __launch_bounds__(256) ens_export_kernel(const EnsWork *__restrict__ work, const uint32_t *__restrict__ list, const float *__restrict__ ws, float *__restrict__ host_res,
                                                        const uint32_t *__restrict__ count, uint32_t cap) {
  const uint32_t nregions = min(*count, cap);
  for (uint32_t region = blockIdx.y; region < nregions; region += gridDim.y) {
    const EnsWork w = work[list ? list[region] : region];
    if (w.host_off == ~0ull) continue;
    const size_t n = (size_t)(w.n2_off - w.nseg_off) + (size_t)w.Ld;
    const float *__restrict__ src = ws + w.nseg_off;
    float *__restrict__ dst = host_res + w.host_off;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) dst[k] = src[k];
  }
}

// `count` (device memory) regions, at most cap: entries list[0..count) of `work`, or -- without a list -- work[0..count);
// grid_regions workgroups share them
void launch_ensemble(hipStream_t stream, const EnsWork *work, const uint32_t *list, const uint32_t *count, uint32_t cap, uint32_t grid_regions, int max_Mp,
                     const DevModel *models, const LenEntry *lentab, const uint8_t *res, const uint64_t *seq_off, float *ws, const uint32_t *seeds,
                     float *host_res) {
  if (!grid_regions) return;
  // CKM_ENS_STREAM=substream: one generator sub-stream per trace (round 1-3's default; NOT hmmsearch's stream).  Default: one stream per region.
  static const int substream = [] { const char *e = getenv("CKM_ENS_STREAM"); return (e && !strcmp(e, "substream")) ? 1 : 0; }();
  if (substream) hipLaunchKernelGGL(ens_trace_kernel, dim3(grid_regions), dim3(256), 0, stream, work, list, models, lentab, ws, seeds, count, cap);
  else hipLaunchKernelGGL(ens_trace_seq_kernel, dim3(std::min<uint32_t>(cap, 4096u)), dim3(64), 0, stream, work, list, models, lentab, ws, seeds, count, cap);
  hipLaunchKernelGGL(ens_null2_kernel, dim3(ENS_N, grid_regions), dim3(64), (size_t)(2 * max_Mp + 32) * 4, stream, work, list, models, res, seq_off, ws, count, cap);
  hipLaunchKernelGGL(ens_sum_kernel, dim3(4, grid_regions), dim3(256), 0, stream, work, list, ws, count, cap);
  if (host_res) hipLaunchKernelGGL(ens_export_kernel, dim3(4, grid_regions), dim3(256), 0, stream, work, list, ws, host_res, count, cap);
}

}  // namespace ckm
