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
// z*Q .. z*Q+Q-1; weights in cell order are M(c) then D(c); local[z] = sequential double sum of the lane's 2Q weights;
// base[z] = local[0] + .. + local[z-1] accumulated in lane order; total = base[64]; the choice is the first position
// (cell order) whose cumulative weight base[z] + running sum exceeds roll * total; none -> node 1, match.
__device__ __forceinline__ int ens_select_e(const float *__restrict__ cr, int Q, int Mp, double roll, int lane) {
  const float *__restrict__ mrow = cr + lane, *__restrict__ drow = cr + 2 * Mp + lane;
  double local = 0.0;
  for (int q = 0; q < Q; ++q) { local += (double)mrow[q * 64]; local += (double)drow[q * 64]; }
  double base = 0.0, run = 0.0;
#pragma unroll
  for (int z = 0; z < 64; ++z) { const double tz = __shfl(local, z); if (lane == z) base = run; run += tz; }
  const double target = roll * run;
  int found = -1;
  double acc = 0.0;
  for (int q = 0; q < Q; ++q) {
    acc += (double)mrow[q * 64]; if (found < 0 && target < base + acc) found = 2 * q;
    acc += (double)drow[q * 64]; if (found < 0 && target < base + acc) found = 2 * q + 1;
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
  const size_t rowsz = (size_t)3 * Mp;
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
#define CELL(c) (((c) % Q) * 64 + (c) / Q)
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
      const int r = ens_select_e(mx + rowsz * row, Q, Mp, (double)xr / 4294967296.0, lane);
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
        if (c > 0) { const int a = CELL(c - 1); pth[1] = pr[a] * tMM[c]; pth[2] = pr[Mp + a] * tIM[c]; pth[3] = pr[2 * Mp + a] * tDM[c]; }
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
        pth[0] = pr[a] * tMI[k - 1]; pth[1] = pr[Mp + a] * tII[k - 1];
        st = (ens_choose(ens_roll(rng), pth, 2) == 0) ? sM : sI;
        --i;
      } break;
      case sD: {
        const int c = k - 1;
        if (c > 0) { const int a = CELL(c - 1); pth[0] = cr[a] * tMD[c - 1]; pth[1] = cr[2 * Mp + a] * tDD[c - 1]; } else pth[0] = pth[1] = 0.0f;
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
// One wavefront per region; every value of the walk (state, row, node, generator) is wave-uniform -- forced through
// v_readfirstlane after each choice so that the state machine branches on SGPRs -- and the 64 lanes are used for what a single walk
// can share: the E-state choice over the 2M exit weights of a row (ens_select_e), the state codes of 64 residues riding in one register
// (lane = residue & 63, stored 128 contiguous bytes at a time), and a LOOK-AHEAD load every sixteen rows that touches the cache lines
// the walk will most likely read next (rows i-16 .. i-31 around the diagonal it is on, M / I / D planes and the special rows): a step is
// one dependent load round, and the 200 walks of ~500 regions do not stay in L2 from one trace to the next, so without the look-ahead
// a step costs an HBM latency.  Same draws, same choices as the oracle's single-stream mode: every state but N takes exactly one draw.
__device__ __forceinline__ int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }

__global__ void __launch_bounds__(64) ens_trace_seq_kernel(const EnsWork *__restrict__ work, const uint32_t *__restrict__ list, const DevModel *__restrict__ models,
                                                          const LenEntry *__restrict__ lentab, float *__restrict__ ws,
                                                          const uint32_t *__restrict__ seeds, const uint32_t *__restrict__ count, uint32_t cap, int warm) {
 const uint32_t nregions = min(*count, cap);
 const int lane = threadIdx.x;
 float sink = 0.0f, pf0 = 0.0f, pf1 = 0.0f, pf2 = 0.0f;
 for (uint32_t region = blockIdx.x; region < nregions; region += gridDim.x) {
  const EnsWork w = work[list ? list[region] : region];
  const DevModel &md = models[w.model];
  const int Q = uni_i(md.fbQ), Mp = Q * 64, Ld = uni_i(w.Ld), segcap = uni_i(w.cap);
  const size_t rowsz = (size_t)3 * Mp;
  const gp<float> mx = gptr(ws + w.mx_off);
  const gp<float> xs = gptr(ws + w.xs_off);
  const gp<float> ftr = gptr(md.ftr);
  const gp<float> tBM = ftr, tMM = ftr + Mp, tIM = ftr + 2 * Mp, tDM = ftr + 3 * Mp,
                  tMI = ftr + 4 * Mp, tII = ftr + 5 * Mp, tMD = ftr + 6 * Mp, tDD = ftr + 7 * Mp;
  const LenEntry le = lentab[w.Lcfg];
  const float loop = le.loop_m, move = le.move_m, Eloop = md.fE_loop, Emove = md.fE_move;
  uint16_t *__restrict__ codes = reinterpret_cast<uint16_t *>(ws + w.code_off);
  int32_t *__restrict__ segs = reinterpret_cast<int32_t *>(ws + w.seg_off);
  int32_t *__restrict__ nsegp = reinterpret_cast<int32_t *>(ws + w.nseg_off);
  const float invQ = 1.0f / (float)Q;
  auto cellq = [&](int c) { int z = (int)((float)c * invQ); z += ((z + 1) * Q <= c) ? 1 : 0; z -= (z * Q > c) ? 1 : 0; return (c - z * Q) * 64 + z; };
  enum { sC, sE, sM, sI, sD, sB, sJ, sN };
  uint32_t rng = (uint32_t)uni_i((int)seeds[0]);            // the region's stream: re-seeded here, carried from trace to trace below
  for (int t = 0; t < ENS_N; ++t) {
    uint16_t *__restrict__ code = codes + (size_t)t * (Ld + 1);
    int32_t *__restrict__ seg = segs + (size_t)t * segcap * 4;
    int st = sC, i = Ld, k = 0, nseg = 0, sqto = 0, hmmto = 0;
    int kq = 0, kz = 0;                                      // cell k-1 of the striped rows is float kq*64 + kz: kept up to date as k falls, no division per step
    bool overflow = false;
    uint32_t mycode = 0;                                     // lane (r & 63) holds the code of residue r of the 64-block the walk is in
    // residue i is done: its block goes to memory when the walk leaves it
#define CELL_K()  (kq * 64 + kz)                                         /* cell k-1 */
#define CELL_KM() (kq > 0 ? (kq - 1) * 64 + kz : (Q - 1) * 64 + kz - 1)   /* cell k-2 */
#define DEC_K()   { if (kq > 0) --kq; else { kq = Q - 1; --kz; } --k; }
#define SET_CODE(v) { if (lane == (i & 63)) mycode = (v); }
#define LEAVE_ROW() { if ((i & 63) == 0) { const int pos_ = i + lane; if (pos_ >= 1 && pos_ <= Ld) code[pos_] = (uint16_t)mycode; } --i; }
    for (;;) {
      float pth0, pth1, pth2 = 0.0f, pth3 = 0.0f; int n = 2;
      if (st == sE) {
        rng = rng * 69069u + 1u;
        const int r = uni_i(ens_select_e((const float *)(mx + rowsz * i), Q, Mp, (double)rng / 4294967296.0, lane));
        k = (r >> 1) + 1; st = (r & 1) ? sD : sM; sqto = 0; hmmto = 0;
        kz = (r >> 1) / Q; kq = (r >> 1) - kz * Q;
      } else if (st == sN) {
        break;
      } else {
        const gp<float> cr = mx + rowsz * i, pr = (i > 0) ? mx + rowsz * (i - 1) : mx;
        if (st == sC) {
          pth0 = xs[(size_t)(i - 1) * 6 + 4] * loop;
          pth1 = (xs[(size_t)i * 6 + 0] * Emove) * xs[(size_t)i * 6 + 5];
        } else if (st == sJ) {
          pth0 = xs[(size_t)(i - 1) * 6 + 2] * loop;
          pth1 = (xs[(size_t)i * 6 + 0] * Eloop) * xs[(size_t)i * 6 + 5];
        } else if (st == sM) {
          const int c = k - 1;
          n = 4;
          pth0 = xs[(size_t)(i - 1) * 6 + 3] * tBM[c];
          if (c > 0) { const int a = CELL_KM(); pth1 = pr[a] * tMM[c]; pth2 = pr[Mp + a] * tIM[c]; pth3 = pr[2 * Mp + a] * tDM[c]; }
          else pth1 = pth2 = pth3 = 0.0f;
          if (warm && (i & 15) == 0 && i > 16) {
            // rows i-16-d (d = lane >> 2 = 0..15) of the diagonal the walk is on, and the diagonals one insert / one delete away;
            // lane & 3: M, I, D plane of the predecessor cell, special row.  The values are consumed at the NEXT look-ahead (they
            // have long arrived by then: loads return in order and sixteen rows of demand loads were waited for in between).
            sink = sink + pf0; sink = sink + pf1; sink = sink + pf2;
            const int d = 16 + (lane >> 2), ri = max(i - d, 0), what = lane & 3;
            const gp<float> rowp = mx + rowsz * ri + (what == 3 ? 0 : what) * Mp;
            const int c0 = max(c - 1 - d, 0), c1 = min(c0 + 1, Mp - 1), c2 = max(c0 - 1, 0);
            pf0 = (what == 3) ? xs[(size_t)ri * 6] : rowp[cellq(c0)];
            pf1 = rowp[cellq(c1)];
            pf2 = rowp[cellq(c2)];
          }
        } else if (st == sI) {
          const int a = CELL_K();
          pth0 = pr[a] * tMI[k - 1]; pth1 = pr[Mp + a] * tII[k - 1];
        } else if (st == sD) {
          const int c = k - 1;
          if (c > 0) { const int a = CELL_KM(); pth0 = cr[a] * tMD[c - 1]; pth1 = cr[2 * Mp + a] * tDD[c - 1]; } else pth0 = pth1 = 0.0f;
        } else {   // sB
          pth0 = xs[(size_t)i * 6 + 1] * move; pth1 = xs[(size_t)i * 6 + 2] * move;
        }
        // the draw and the choice (ens_choose's order of operations: float sum of the weights, double running sum)
        rng = rng * 69069u + 1u;
        const double roll = (double)rng / 4294967296.0;
        int ch;
        {
          float norm = pth0 + pth1;
          if (n == 4) { norm = norm + pth2; norm = norm + pth3; }
          if (!(norm > 0.0f)) ch = 0;
          else {
            const double target = roll * (double)norm;
            double sum = (double)pth0;
            if (target < sum) ch = 0;
            else { sum += (double)pth1;
              if (target < sum) ch = 1;
              else if (n == 2) ch = (pth1 > 0.0f) ? 1 : 0;
              else { sum += (double)pth2;
                if (target < sum) ch = 2;
                else { sum += (double)pth3;
                  if (target < sum) ch = 3;
                  else ch = (pth3 > 0.0f) ? 3 : (pth2 > 0.0f) ? 2 : (pth1 > 0.0f) ? 1 : 0; } } }
          }
        }
        ch = uni_i(ch);
        if (st == sC || st == sJ) {
          if (ch == 0) { SET_CODE(0u) LEAVE_ROW() } else st = sE;
        } else if (st == sM) {
          SET_CODE(0x4000u | (uint32_t)k)
          if (!sqto) { sqto = i; hmmto = k; }
          if (ch == 0) {
            if (nseg == segcap) { overflow = true; break; }
            if (lane == 0) { seg[nseg * 4 + 0] = i; seg[nseg * 4 + 1] = sqto; seg[nseg * 4 + 2] = k; seg[nseg * 4 + 3] = hmmto; }
            ++nseg;
            st = sB;
          } else st = (ch == 1) ? sM : (ch == 2) ? sI : sD;
          LEAVE_ROW() DEC_K()
        } else if (st == sI) {
          SET_CODE(0x8000u | (uint32_t)k)
          st = (ch == 0) ? sM : sI;
          LEAVE_ROW()
        } else if (st == sD) {
          st = (ch == 0) ? sM : sD;
          DEC_K()
        } else {
          st = (ch == 0) ? sN : sJ;
        }
      }
      // a numerically impossible move ends the trace
      if (i < 0 || k < 0 || ((st == sM || st == sI) && (k < 1 || i < 1)) || (st == sD && k < 1) || ((st == sC || st == sJ || st == sE) && i < 1)) break;
    }
    // residues i .. 1 lie outside every domain: code 0 (the block the walk stopped in, then everything below it)
    if (i >= 0) {
      const int b = i & ~63;
      if (lane <= (i & 63)) mycode = 0;
      { const int pos_ = b + lane; if (pos_ >= 1 && pos_ <= Ld) code[pos_] = (uint16_t)mycode; }
      for (int pos_ = 1 + lane; pos_ < b; pos_ += 64) code[pos_] = 0;
    }
    if (lane == 0) nsegp[t] = overflow ? -1 : nseg;
#undef SET_CODE
#undef CELL_K
#undef CELL_KM
#undef DEC_K
#undef LEAVE_ROW
  }
 }
 if (nregions == 0xffffffffu) ws[0] = sink + pf0 + pf1 + pf2;      // (never true) keeps the look-ahead loads alive
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
  static const int warm = [] { const char *e = getenv("CKM_ENS_WARM"); return (e && !strcmp(e, "0")) ? 0 : 1; }();
  if (substream) hipLaunchKernelGGL(ens_trace_kernel, dim3(grid_regions), dim3(256), 0, stream, work, list, models, lentab, ws, seeds, count, cap);
  else hipLaunchKernelGGL(ens_trace_seq_kernel, dim3(std::min<uint32_t>(cap, 4096u)), dim3(64), 0, stream, work, list, models, lentab, ws, seeds, count, cap, warm);
  hipLaunchKernelGGL(ens_null2_kernel, dim3(ENS_N, grid_regions), dim3(64), (size_t)(2 * max_Mp + 32) * 4, stream, work, list, models, res, seq_off, ws, count, cap);
  hipLaunchKernelGGL(ens_sum_kernel, dim3(4, grid_regions), dim3(256), 0, stream, work, list, ws, count, cap);
  if (host_res) hipLaunchKernelGGL(ens_export_kernel, dim3(4, grid_regions), dim3(256), 0, stream, work, list, ws, host_res, count, cap);
}

}  // namespace ckm
