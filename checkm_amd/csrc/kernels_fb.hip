// kernels_fb.hip -- float32 Forward / Backward / posterior decoding / null2 / optimal-accuracy
// kernels, gfx950 only.  One wavefront per (model, sequence) pair or per envelope; lane z owns the
// Q consecutive model nodes c = z*Q+q ("canonical 64-lane blocked order"), so
// every order-dependent float reduction has ONE defined evaluation order:
//   D->D chain   lane-local affine map (a,b), scan over lanes (xlane.h: rows of 16 by offsets 1,2,4,8, then row prefixes), lane-local replay
//   E-state sum  lane-local fold over q, then xor butterfly 1,2,4,8,16,32
// Compiled with -ffp-contract=off: no fused multiply-add may be formed.  No transcendental is
// evaluated on the device; rescale factors go back to the host as events and are logged there.
// Reference stage replaced: Forward/Backward/domain definition of the hmmsearch per-target pipeline
// (process launched at checkm/hmmer.py:70).
#include <hip/hip_runtime.h>
#include <cstring>
#include "dev_types.h"
#include "cascade_dev.h"
#include "cascade_regions.h"
#include "xlane.h"

namespace ckm {



constexpr float NEGINF_F = -__builtin_inff();

// Transition odds in LDS, transposed to [array][q][lane] so that the 64 lanes of a wave read 64
// consecutive floats (conflict-free); arrays: BM MM IM DM MI II MD DD.  Cell c = lane*Q + q.
template <int Q>
__device__ __forceinline__ int lds_cell(int c) { return (c % Q) * 64 + c / Q; }

template <int Q>
struct Tr {
  const float *t; int lane;
  __device__ __forceinline__ float BM(int q) const { return t[0 * Q * 64 + q * 64 + lane]; }
  __device__ __forceinline__ float MM(int q) const { return t[1 * Q * 64 + q * 64 + lane]; }
  __device__ __forceinline__ float IM(int q) const { return t[2 * Q * 64 + q * 64 + lane]; }
  __device__ __forceinline__ float DM(int q) const { return t[3 * Q * 64 + q * 64 + lane]; }
  __device__ __forceinline__ float MI(int q) const { return t[4 * Q * 64 + q * 64 + lane]; }
  __device__ __forceinline__ float II(int q) const { return t[5 * Q * 64 + q * 64 + lane]; }
  __device__ __forceinline__ float MD(int q) const { return t[6 * Q * 64 + q * 64 + lane]; }
  __device__ __forceinline__ float DD(int q) const { return t[7 * Q * 64 + q * 64 + lane]; }
  // any cell of any array (neighbour cells, traceback)
  __device__ __forceinline__ float at(int arr, int c) const { return t[arr * Q * 64 + lds_cell<Q>(c)]; }
};

// A workgroup is ONE wavefront that takes work items from a queue (dev_types.h: WorkQueue) until the queue is empty: the queue
// may have been written by the host (alignment requests, diagnostics, second envelope rounds) or by the kernels of the previous
// stage (the device-driven cascade, ckm_cascade.hip), in which case its length is only known on the device.  The wavefront
// keeps the transition image of its current model in LDS and reloads it when the model changes.
template <int Q>
__device__ __forceinline__ void load_tr(float *lds, const float *ftr) {
  constexpr int Mp = Q * 64;
  __syncthreads();
  const gp<float> g = gptr(ftr);
  for (int i = threadIdx.x; i < 8 * Mp; i += blockDim.x) { const int arr = i / Mp, c = i % Mp; lds[arr * Mp + lds_cell<Q>(c)] = g[i]; }
  __syncthreads();
}

// same image, but 0 for a possible transition and -inf for an impossible one (optimal-accuracy max-plus gates)
template <int Q>
__device__ __forceinline__ void load_gates(float *lds, const float *ftr) {
  constexpr int Mp = Q * 64;
  __syncthreads();
  const gp<float> g = gptr(ftr);
  for (int i = threadIdx.x; i < 8 * Mp; i += blockDim.x) { const int arr = i / Mp, c = i % Mp; lds[arr * Mp + lds_cell<Q>(c)] = (g[i] > 0.f) ? 0.0f : -__builtin_inff(); }
  __syncthreads();
}

// ---- one Forward row --------------------------------------------------------------------------
template <int Q>
__device__ __forceinline__ float fwd_row(float (&Mv)[Q], float (&Iv)[Q], float (&Dv)[Q], const Tr<Q> &tr,
                                         const float (&rfx)[Q], float xB, int lane) {
  const float mpi = lane_up1(Mv[Q - 1], 0.f), ipi = lane_up1(Iv[Q - 1], 0.f), dpi = lane_up1(Dv[Q - 1], 0.f);
  float Mn[Q], In[Q], Dn[Q], md[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const float mp = q ? Mv[q - 1] : mpi, ip = q ? Iv[q - 1] : ipi, dp = q ? Dv[q - 1] : dpi;
    float sv = xB * tr.BM(q);
    sv = sv + mp * tr.MM(q);
    sv = sv + ip * tr.IM(q);
    sv = sv + dp * tr.DM(q);
    Mn[q] = sv * rfx[q];
    const float a = Mv[q] * tr.MI(q), b = Iv[q] * tr.II(q);
    In[q] = a + b;
    md[q] = Mn[q] * tr.MD(q);
  }
  float a = 1.0f, b = 0.0f;
#pragma unroll
  for (int q = 0; q < Q; ++q) { const float dd = tr.DD(q); const float t = dd * b; b = md[q] + t; a = dd * a; }
  scan_up(a, b, lane);
  float d = lane_up1(b, 0.0f);
#pragma unroll
  for (int q = 0; q < Q; ++q) { Dn[q] = d; const float t = tr.DD(q) * d; d = md[q] + t; }
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < Q; ++q) { s = s + Mn[q]; s = s + Dn[q]; }
  s = wave_sum(s);
#pragma unroll
  for (int q = 0; q < Q; ++q) { Mv[q] = Mn[q]; Iv[q] = In[q]; Dv[q] = Dn[q]; }
  return s;
}

// One Forward pass of one work item by one wavefront (the LDS image of the item's model is in place): special rows, optional matrix
// rows, rescale events; returns the scaled xC, the number of rescales and the approximate log of their product.
template <int Q>
__device__ __forceinline__ void fwd_item(const FbWork &w, const DevModel &md, float *lds, int lane, const LenEntry *__restrict__ lentab,
                                         const uint8_t *__restrict__ res, const uint64_t *__restrict__ seq_off, float *__restrict__ ws,
                                         ScaleEvent *__restrict__ events, uint32_t *__restrict__ nevents, uint32_t cap_events,
                                         float &o_xC, int &o_nscale, float &o_lsum, float &o_move) {
  constexpr int Mp = Q * 64;
  Tr<Q> tr{lds, lane};
  const uint8_t *rp = res + seq_off[w.seq] + w.i0;
  const LenEntry le = lentab[w.Lcfg];
  const float loop = w.multihit ? le.loop_m : le.loop_u, move = w.multihit ? le.move_m : le.move_u;
  const float Eloop = w.multihit ? md.fE_loop : 0.0f, Emove = w.multihit ? md.fE_move : 1.0f;
  float Mv[Q], Iv[Q], Dv[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) Mv[q] = Iv[q] = Dv[q] = 0.f;
  float xN = 1.0f, xB = xN * move, xE = 0.f, xJ = 0.f, xC = 0.f;
  int nscale = 0;
  float lsum = 0.f;                  // approximate log of the product of the rescale factors (conservative F3 test only)
  float *xs = ws + w.xs_off;
  float *mx = w.full ? ws + w.mxf_off : nullptr;
  if (lane == 0) { xs[0] = 0.f; xs[1] = xN; xs[2] = 0.f; xs[3] = xB; xs[4] = 0.f; xs[5] = 1.0f; }
  // Matrix rows.  full == 1 (envelopes): three (two used) q-major planes per row, [plane][q][lane].  full == 2 (multihit Forward of a
  // multi-domain region, read only by the trace ensemble): CELL-major, one float4 {M, I, D, 0} per node, rows of 4 * Mp floats -- a
  // stochastic traceback follows a diagonal, and with the node index running fastest the cells it will visit next lie side by side
  // (kernels_ens.hip stages windows of them in LDS); lane z owns the Q consecutive nodes z*Q .. z*Q+Q-1, so every store is 16 B and the
  // lanes' stores are contiguous.
  if (mx) {
    if (w.full == 2) {
      float4 *r4 = reinterpret_cast<float4 *>(mx) + lane * Q;
#pragma unroll
      for (int q = 0; q < Q; ++q) r4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
      for (int q = 0; q < Q; ++q) { mx[q * 64 + lane] = 0.f; mx[Mp + q * 64 + lane] = 0.f; }
    }
  }
  // emission odds of the next row are fetched one row ahead; the residues come 64 at a time (xlane.h: ResUp)
  const gp<float> rf = gptr(md.rf);
  ResUp feed; feed.init(rp, w.Ld, lane);
  float rfc[Q];
  {
    const gp<float> r0 = rf + (size_t)feed.get(0) * Mp + lane;
#pragma unroll
    for (int q = 0; q < Q; ++q) rfc[q] = r0[q * 64];
  }
  for (int i = 1; i <= w.Ld; ++i) {
    float rfn[Q];
    {
      const int xn = feed.get((i < w.Ld) ? i : i - 1);
      const gp<float> r1 = rf + (size_t)xn * Mp + lane;
#pragma unroll
      for (int q = 0; q < Q; ++q) rfn[q] = r1[q * 64];
    }
    xE = fwd_row<Q>(Mv, Iv, Dv, tr, rfc, xB, lane);
#pragma unroll
    for (int q = 0; q < Q; ++q) rfc[q] = rfn[q];
    xN = xN * loop;
    { const float a = xC * loop, b = xE * Emove; xC = a + b; }
    { const float a = xJ * loop, b = xE * Eloop; xJ = a + b; }
    { const float a = xJ * move, b = xN * move; xB = a + b; }
    float scale = 1.0f;
    if (xE > 1.0e4f) {
      const float inv = 1.0f / xE;
      xN = xN / xE; xC = xC / xE; xJ = xJ / xE; xB = xB / xE;
#pragma unroll
      for (int q = 0; q < Q; ++q) { Mv[q] *= inv; Iv[q] *= inv; Dv[q] *= inv; }
      scale = xE; xE = 1.0f; ++nscale;
      lsum += approx_ln(scale);
      if (lane == 0) {
        const uint32_t e = atomicAdd(nevents, 1u);
        if (e < cap_events) { ScaleEvent ev; ev.slot = w.slot; ev.row = i; ev.scale = scale; ev.pad = 0; events[e] = ev; }
      }
    }
    if (lane == 0) { float *r = xs + (size_t)i * 6; r[0] = xE; r[1] = xN; r[2] = xJ; r[3] = xB; r[4] = xC; r[5] = scale; }
    if (mx) {
      if (w.full == 2) {      // the trace ensemble also walks delete states
        float4 *r4 = reinterpret_cast<float4 *>(mx) + (size_t)i * Mp + lane * Q;
#pragma unroll
        for (int q = 0; q < Q; ++q) r4[q] = make_float4(Mv[q], Iv[q], Dv[q], 0.f);
      } else {
        float *r = mx + (size_t)i * 3 * Mp + lane;
#pragma unroll
        for (int q = 0; q < Q; ++q) { r[q * 64] = Mv[q]; r[Mp + q * 64] = Iv[q]; }
      }
    }
  }
  o_xC = xC; o_nscale = nscale; o_lsum = lsum; o_move = move;
}

template <int Q>
__global__ void __launch_bounds__(64) fwd_kernel(WorkQueue queue, FbWork *__restrict__ work,
                                                const DevModel *__restrict__ models, const LenEntry *__restrict__ lentab,
                                                const uint8_t *__restrict__ res, const uint64_t *__restrict__ seq_off,
                                                float *__restrict__ ws, FwdOut *__restrict__ out,
                                                ScaleEvent *__restrict__ events, uint32_t *__restrict__ nevents, uint32_t cap_events,
                                                CascadeDev cd, int decide) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  uint32_t cur_model = 0xffffffffu;
 const uint32_t nqueue = queue_len(queue);
 for (uint32_t qk = blockIdx.x; qk < nqueue; qk += gridDim.x) {
  const uint32_t item = queue.list[qk];
  const FbWork w = work[item];
  const DevModel &md = models[w.model];
  if (w.model != cur_model) { load_tr<Q>(lds, md.ftr); cur_model = w.model; }
  float xC, lsum, move; int nscale;
  fwd_item<Q>(w, md, lds, lane, lentab, res, seq_off, ws, events, nevents, cap_events, xC, nscale, lsum, move);
  if (lane == 0) { out[w.slot].xC = xC; out[w.slot].nscale = nscale; }
  // ---- device-driven cascade: the F3 decision of a whole-sequence parser item, taken conservatively (the threshold is lowered by a
  // margin that covers the approximate logarithms; the host repeats the test exactly).  A passer gets its pass record, the workspace
  // of its decoding terms, and a place in the Backward queue of this register class.
  if (decide && lane == 0) {
    const PairRec pr = cd.cand[w.cand];
    const float fwdsc = lsum + approx_ln(xC * move);
    const float sc = (fwdsc - pr.filtersc) * LOG2E_F;
    if (sc >= md.thr_fwd_f3 - cd.margin_fwd) {
      unsigned long long off;
      if (ws_alloc(cd, (unsigned long long)(w.Ld + 1) * 3ull, off)) {
        const uint32_t pid = atomicAdd(&cd.gcnt[CC_PASS], 1u);
        if (pid < cd.cap_pass) {
          PassRec r;
          r.cand = w.cand; r.fwork = item; r.model = w.model; r.seq = w.seq; r.usc = pr.usc;
          r.bias_d = cd.bias_raw[2 * (size_t)w.cand]; r.bias_e = cd.bias_raw[2 * (size_t)w.cand + 1];
          r.vit_fast = cd.vit_fast[w.cand]; r.vit_exact = cd.vit_exact[w.cand]; r.vit_flag = cd.vit_flag[w.cand]; r.route = cd.route[w.cand];
          r.fwd_xC = xC; r.nscale = nscale;
          cd.h_pass[pid] = r;
          work[item].aux_off = off; work[item].pass = pid;
          queue_push(cd, cd.bq, CC_BQ, md.fb_cls, cd.cap_fq, item, (uint32_t)CS_FWORK);
        } else atomicOr(&cd.gcnt[CC_STATUS], (uint32_t)CS_PASS);
      }
    }
  }
 }
}

#define CKM_LD2(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)   /* L2-served: this wave's own stores are visible */
// ---- Backward -----------------------------------------------------------------------------------
// D chain, reverse direction: D[c] = av[c] + DD[c]*D[c+1]
template <int Q>
__device__ __forceinline__ void bwd_dchain(float (&Dn)[Q], const float (&av)[Q], const Tr<Q> &tr, int lane) {
  float a = 1.0f, b = 0.0f;
#pragma unroll
  for (int q = Q - 1; q >= 0; --q) { const float dd = tr.DD(q); const float t = dd * b; b = av[q] + t; a = dd * a; }
  scan_down(a, b, lane);
  float d = lane_down1(b, 0.0f);
#pragma unroll
  for (int q = Q - 1; q >= 0; --q) { const float t = tr.DD(q) * d; d = av[q] + t; Dn[q] = d; }
}

// One Backward pass of one work item by one wavefront.  XL: the Forward pass of the item ran in this very wavefront just before (fused
// kernels), so the special rows lane 0 stored are read with L2-scope loads.  Parser items (full = 0) leave the three decoding terms
// per row; envelope items (full = 1) the posterior rows M, I and the N/J/C terms, and report a non-finite posterior through `bad`.
template <int Q, bool XL>
__device__ __forceinline__ void bwd_item(const FbWork &w, const DevModel &md, float *lds, int lane, const LenEntry *__restrict__ lentab,
                                         const uint8_t *__restrict__ res, const uint64_t *__restrict__ seq_off, float *__restrict__ ws,
                                         float fwd_xC, bool &o_bad) {
  constexpr int Mp = Q * 64;
  Tr<Q> tr{lds, lane};
  const uint8_t *rp = res + seq_off[w.seq] + w.i0;
  const LenEntry le = lentab[w.Lcfg];
  const float loop = w.multihit ? le.loop_m : le.loop_u, move = w.multihit ? le.move_m : le.move_u;
  const float Eloop = w.multihit ? md.fE_loop : 0.0f, Emove = w.multihit ? md.fE_move : 1.0f;
  const float *xs = ws + w.xs_off;
  float *aux = ws + w.aux_off;
  const float *fm = w.full ? ws + w.mxf_off : nullptr;
  float *bm = w.full ? ws + w.mxb_off : nullptr;
  // posterior rows either have a matrix of their own (2 arrays per row) or overwrite the Forward rows they were computed from (same
  // offset: the Forward matrix's 3 arrays per row; row r's Forward values are in registers before its posterior is stored)
  const size_t pst = (w.mxb_off == w.mxf_off) ? (size_t)3 * Mp : (size_t)2 * Mp;
  const int L = w.Ld;
  const float invZ = 1.0f / (fwd_xC * move);
  // boundary transition odds of the right-hand neighbour cell (c+1)
  float tIMn[Q], tMMn[Q], tDMn[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int c = lane * Q + q + 1;
    tMMn[q] = (c < Mp) ? tr.at(1, c) : 0.f; tIMn[q] = (c < Mp) ? tr.at(2, c) : 0.f; tDMn[q] = (c < Mp) ? tr.at(3, c) : 0.f;
  }
  float Mv[Q], Iv[Q], Dv[Q];
  // row L
  float xC = move, xE = xC * Emove, xJ = 0.f, xB = 0.f, xN = 0.f;
  {
    float av[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) av[q] = xE + 0.0f;
    bwd_dchain<Q>(Dv, av, tr, lane);
    const float dnx = lane_down1(Dv[0], 0.f);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const float dn1 = (q + 1 < Q) ? Dv[q + 1] : dnx;
      float m = xE + 0.0f; m = m + 0.0f; m = m + tr.MD(q) * dn1;
      Mv[q] = m; Iv[q] = 0.f;
    }
  }
  bool bad = false;
  // forward rows are pulled into registers one row ahead of their use: the loads must not sit behind the
  // (possibly aliasing, same workspace) stores of the posterior rows
  float fMr[Q], fIr[Q];
  auto fetch_f = [&](int r) {
    if (w.full && r >= 1) {
      const float *__restrict__ f = fm + (size_t)r * 3 * Mp + lane;
#pragma unroll
      for (int q = 0; q < Q; ++q) { fMr[q] = f[q * 64]; fIr[q] = f[Mp + q * 64]; }
    }
  };
  fetch_f(L);
  // forward special rows r+1 (rowU), r (rowC), r-1 (rowD) ride in registers; the next one is requested a row ahead
  float rowU[6], rowC[6], rowD[6];
  auto load_row = [&](float (&dst)[6], int r) {
    const float *x = xs + (size_t)(r < 0 ? 0 : r) * 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) dst[k] = XL ? CKM_LD2(&x[k]) : x[k];      // (XL: lane 0 of THIS wavefront wrote them a moment ago)
  };
  load_row(rowC, L); load_row(rowD, L - 1);
#pragma unroll
  for (int k = 0; k < 6; ++k) rowU[k] = 1.0f;
  auto emit = [&](int r) {
    // decoding terms that become available once backward row r is final
    if (w.full) {
      if (r >= 1) {
        float *__restrict__ b = bm + (size_t)r * pst + lane;        // posterior rows hold M and I only
        float pmv[Q], piv[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          float pm = fMr[q] * Mv[q]; pm = pm * invZ;
          float pi = fIr[q] * Iv[q]; pi = pi * invZ;
          pmv[q] = pm; piv[q] = pi;
          if (!(__builtin_isfinite(pm) && __builtin_isfinite(pi))) bad = true;
        }
        fetch_f(r - 1);
#pragma unroll
        for (int q = 0; q < Q; ++q) { b[q * 64] = pmv[q]; b[Mp + q * 64] = piv[q]; }
        if (lane == 0) {
          const float wgt = invZ / rowC[5];
          float t;
          t = rowD[1] * xN; t = t * loop; aux[(size_t)r * 3 + 0] = t * wgt;
          t = rowD[2] * xJ; t = t * loop; aux[(size_t)r * 3 + 1] = t * wgt;
          t = rowD[4] * xC; t = t * loop; aux[(size_t)r * 3 + 2] = t * wgt;
        }
      }
    } else if (lane == 0) {
      if (r >= 1) {
        float et = rowC[0] * xE; et = et * invZ;
        const float wgt = invZ / rowC[5];
        float a = rowD[1] * xN; a = a * loop;
        float b = rowD[2] * xJ; b = b * loop;
        float c = rowD[4] * xC; c = c * loop;
        aux[(size_t)r * 3 + 1] = et;
        aux[(size_t)r * 3 + 2] = ((a + b) + c) * wgt;
      }
      if (r < L) { float bt = rowC[3] * xB; bt = bt * invZ; aux[(size_t)(r + 1) * 3 + 0] = bt; }
    }
  };
  emit(L);
  const gp<float> rf = gptr(md.rf);
  ResDown feed; feed.init(rp, max(L, 1), lane);
  float rfc[Q];
  if (L >= 1) {
    const gp<float> r0 = rf + (size_t)feed.get(L - 1) * Mp + lane;    // residue L
#pragma unroll
    for (int q = 0; q < Q; ++q) rfc[q] = r0[q * 64];
  }
  for (int i = L - 1; i >= 0; --i) {
#pragma unroll
    for (int k = 0; k < 6; ++k) { rowU[k] = rowC[k]; rowC[k] = rowD[k]; }
    load_row(rowD, i - 1);
    float rfn[Q];                                                     // residue i, needed by the next iteration
    {
      const int xn = feed.get((i >= 1) ? i - 1 : 0);
      const gp<float> r1 = rf + (size_t)xn * Mp + lane;
#pragma unroll
      for (int q = 0; q < Q; ++q) rfn[q] = r1[q * 64];
    }
    float mn[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) mn[q] = Mv[q] * rfc[q];
#pragma unroll
    for (int q = 0; q < Q; ++q) rfc[q] = rfn[q];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q) { const float t = tr.BM(q) * mn[q]; s = s + t; }
    xB = wave_sum(s);
    { const float a = xB * move, b = xJ * loop; xJ = a + b; }
    xC = xC * loop;
    { const float a = xC * Emove, b = xJ * Eloop; xE = a + b; }
    { const float a = xB * move, b = xN * loop; xN = a + b; }
    const float mnx = lane_down1(mn[0], 0.f);
    float av[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) { const float mn1 = (q + 1 < Q) ? mn[q + 1] : mnx; av[q] = xE + tDMn[q] * mn1; }
    float Dn[Q];
    bwd_dchain<Q>(Dn, av, tr, lane);
    const float dnx = lane_down1(Dn[0], 0.f);
    float Mn[Q], In[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const float mn1 = (q + 1 < Q) ? mn[q + 1] : mnx;
      const float dn1 = (q + 1 < Q) ? Dn[q + 1] : dnx;
      { const float a = tIMn[q] * mn1, b = tr.II(q) * Iv[q]; In[q] = a + b; }
      float m = xE + tMMn[q] * mn1;
      m = m + tr.MI(q) * Iv[q];
      m = m + tr.MD(q) * dn1;
      Mn[q] = m;
    }
    const float sc = rowU[5];
    if (sc != 1.0f) {
      const float inv = 1.0f / sc;
#pragma unroll
      for (int q = 0; q < Q; ++q) { Mn[q] *= inv; In[q] *= inv; Dn[q] *= inv; }
      xE *= inv; xN *= inv; xJ *= inv; xB *= inv; xC *= inv;
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) { Mv[q] = Mn[q]; Iv[q] = In[q]; Dv[q] = Dn[q]; }
    emit(i);
  }
  o_bad = bad;
}

template <int Q>
__global__ void __launch_bounds__(64) bwd_kernel(WorkQueue queue, const FbWork *__restrict__ work,
                                                const DevModel *__restrict__ models, const LenEntry *__restrict__ lentab,
                                                const uint8_t *__restrict__ res, const uint64_t *__restrict__ seq_off,
                                                float *__restrict__ ws, const FwdOut *__restrict__ fout, int32_t *__restrict__ range_err) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  uint32_t cur_model = 0xffffffffu;
 const uint32_t nqueue = queue_len(queue);
 for (uint32_t qk = blockIdx.x; qk < nqueue; qk += gridDim.x) {
  const uint32_t item = queue.list[qk];
  const FbWork w = work[item];
  const DevModel &md = models[w.model];
  if (w.model != cur_model) { load_tr<Q>(lds, md.ftr); cur_model = w.model; }
  bool bad = false;
  bwd_item<Q, false>(w, md, lds, lane, lentab, res, seq_off, ws, fout[w.slot].xC, bad);
  if (w.full) {
    const unsigned long long any = __ballot(bad);
    if (lane == 0) range_err[w.slot] = any ? 1 : 0;
  }
 }
}

// ---- null2 by expectation + optimal accuracy fill + traceback -------------------------------------
// null2 by expectation + optimal-accuracy fill + traceback of one envelope by one wavefront (the LDS image holds the 0 / -inf GATES of
// the item's model).  XL: Backward ran in this very wavefront just before (fused kernel): the N/J/C terms lane 0 stored are read with
// L2-scope loads.
#define LD2X(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
template <int Q, bool XL>
__device__ __forceinline__ void oa_item(const FbWork &w, const DevModel &md, float *lds, int lane, float *__restrict__ ws, EnvOut *outp) {
  constexpr int Mp = Q * 64;
  Tr<Q> tr{lds, lane};
  const int M = md.M, L = w.Ld, c0 = lane * Q;
  float *pp = ws + w.mxb_off;       // posterior rows (M, I; D = 0)
  const size_t pst = (w.mxb_off == w.mxf_off) ? (size_t)3 * Mp : (size_t)2 * Mp;     // in place: the OA rows below overwrite them row by row, each after it was read
  float *oa = ws + w.mxf_off;       // OA rows overwrite the forward matrix
  const float *aux = ws + w.aux_off;                                   // [ppN ppJ ppC] per row (written by bwd_kernel)
  float *oax = ws + w.aux_off + (((size_t)(w.Ld + 1) * 3 + 31) & ~(size_t)31);   // [oN oB oE oJ oC] per row, own cache lines
  const bool Eloop_ok = false;      // envelopes are rescored unihit
  // ---------------- OA fill ----------------
  // Max-plus with additive gates: from here on the LDS image holds 0 for a possible transition and -inf for an
  // impossible one (cells beyond M and node-0 predecessors have zero odds, so they gate themselves).
  float Mv[Q], Iv[Q], Dv[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) { Mv[q] = Iv[q] = Dv[q] = NEGINF_F; oa[q * 64 + lane] = NEGINF_F; oa[Mp + q * 64 + lane] = NEGINF_F; oa[2 * Mp + q * 64 + lane] = NEGINF_F; }
  float oN = 0.f, oB = 0.f, oE = NEGINF_F, oJ = NEGINF_F, oC = NEGINF_F;
  if (lane == 0) { oax[0] = oN; oax[1] = oB; oax[2] = oE; oax[3] = oJ; oax[4] = oC; }
  // null2 by expectation rides along: per-cell posterior sums in ascending row order (the oracle's order)
  float me[Q], ie[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) me[q] = ie[q] = 0.f;
  float n2N = 0.f, n2J = 0.f, n2C = 0.f;
  float ppM[Q], ppI[Q];
  float ax0 = 0.f, ax1 = 0.f, ax2 = 0.f;          // ppN ppJ ppC of the current row
  if (L >= 1) { ax0 = XL ? LD2X(&aux[3]) : aux[3]; ax1 = XL ? LD2X(&aux[4]) : aux[4]; ax2 = XL ? LD2X(&aux[5]) : aux[5]; }
  if (L >= 1) {
    const float *__restrict__ p1 = pp + (size_t)1 * pst + lane;
#pragma unroll
    for (int q = 0; q < Q; ++q) { ppM[q] = p1[q * 64]; ppI[q] = p1[Mp + q * 64]; }
  }
  for (int i = 1; i <= L; ++i) {
    // posterior row i is in registers; row i+1 is requested now, before this row's stores
    float ppMn[Q], ppIn[Q];
    const float *__restrict__ axn = aux + (size_t)((i < L) ? i + 1 : i) * 3;
    const float an0 = XL ? LD2X(&axn[0]) : axn[0], an1 = XL ? LD2X(&axn[1]) : axn[1], an2 = XL ? LD2X(&axn[2]) : axn[2];
    {
      const float *__restrict__ pn = pp + (size_t)((i < L) ? i + 1 : i) * pst + lane;
#pragma unroll
      for (int q = 0; q < Q; ++q) { ppMn[q] = pn[q * 64]; ppIn[q] = pn[Mp + q * 64]; }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) { me[q] = me[q] + ppM[q]; ie[q] = ie[q] + ppI[q]; }
    n2N = n2N + ax0; n2J = n2J + ax1; n2C = n2C + ax2;
    const float mpi = lane_up1(Mv[Q - 1], NEGINF_F), ipi = lane_up1(Iv[Q - 1], NEGINF_F), dpi = lane_up1(Dv[Q - 1], NEGINF_F);
    float Mn[Q], In[Q], Dn[Q];
    float e = NEGINF_F;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const float mp = q ? Mv[q - 1] : mpi, ip = q ? Iv[q - 1] : ipi, dp = q ? Dv[q - 1] : dpi;
      const float best = fmaxf(fmaxf(mp + tr.MM(q), ip + tr.IM(q)), fmaxf(dp + tr.DM(q), oB + tr.BM(q)));
      const float bi = fmaxf(Mv[q] + tr.MI(q), Iv[q] + tr.II(q));
      Mn[q] = best + ppM[q]; In[q] = bi + ppI[q];
      e = fmaxf(e, Mn[q]);
    }
    // D chain: D(c) = max(M(c-1) + gMD(c-1), D(c-1) + gDD(c-1)): lane-local fold, max-plus Kogge-Stone over lanes
    {
      float dloc[Q], pw[Q];
      dloc[0] = NEGINF_F; pw[0] = 0.0f;
#pragma unroll
      for (int q = 1; q < Q; ++q) { dloc[q] = fmaxf(Mn[q - 1] + tr.MD(q - 1), dloc[q - 1] + tr.DD(q - 1)); pw[q] = pw[q - 1] + tr.DD(q - 1); }
      float outv = fmaxf(Mn[Q - 1] + tr.MD(Q - 1), dloc[Q - 1] + tr.DD(Q - 1));
      float wgt = pw[Q - 1] + tr.DD(Q - 1);
      // max-plus scan over lanes: exact whatever the association (weights are sums of 0 / -inf gates)
#define MP(ov, ow) { outv = fmaxf(outv, (ov) + wgt); wgt = wgt + (ow); }
#define STEP(S) { const float ov = dpp_f<DPP_ROW_SHR + S>(NEGINF_F, outv), ow = dpp_f<DPP_ROW_SHR + S>(0.0f, wgt); MP(ov, ow) }
      STEP(1) STEP(2) STEP(4) STEP(8)
#undef STEP
      {
        const float v0 = read_lane(outv, 15), w0 = read_lane(wgt, 15), v1 = read_lane(outv, 31), w1 = read_lane(wgt, 31), v2 = read_lane(outv, 47), w2 = read_lane(wgt, 47);
        const float pv2 = fmaxf(v1, v0 + w1), pw2 = w1 + w0;
        const float pv3 = fmaxf(v2, pv2 + w2), pw3 = w2 + pw2;
        const int row = lane >> 4;
        const float pv = row == 0 ? NEGINF_F : row == 1 ? v0 : row == 2 ? pv2 : pv3;
        const float pw_ = row == 0 ? 0.0f : row == 1 ? w0 : row == 2 ? pw2 : pw3;
        MP(pv, pw_)
      }
#undef MP
      const float carry = lane_up1(outv, NEGINF_F);
#pragma unroll
      for (int q = 0; q < Q; ++q) Dn[q] = fmaxf(dloc[q], carry + pw[q]);
    }
    e = wave_max(e);
    oE = e;
    { const float a = oJ + ax1; const float b = Eloop_ok ? e : NEGINF_F; oJ = a > b ? a : b; }
    { const float a = oC + ax2; oC = a > e ? a : e; }
    oN = oN + ax0;
    ax0 = an0; ax1 = an1; ax2 = an2;
    oB = oN > oJ ? oN : oJ;
    float *__restrict__ r = oa + (size_t)i * 3 * Mp + lane;
#pragma unroll
    for (int q = 0; q < Q; ++q) { Mv[q] = Mn[q]; Iv[q] = In[q]; Dv[q] = Dn[q]; r[q * 64] = Mn[q]; r[Mp + q * 64] = In[q]; r[2 * Mp + q * 64] = Dn[q]; ppM[q] = ppMn[q]; ppI[q] = ppIn[q]; }
    if (lane == 0) { float *a = oax + (size_t)i * 5; a[0] = oN; a[1] = oB; a[2] = oE; a[3] = oJ; a[4] = oC; }
  }
  {
    const float norm = 1.0f / (float)L;
#pragma unroll
    for (int q = 0; q < Q; ++q) { me[q] *= norm; ie[q] *= norm; }
    const float xfactor = ((n2N + n2C) + n2J) * norm;
    for (int x = 0; x < 20; ++x) {
      const gp<float> rfx = gptr(md.rf) + (size_t)x * Mp + lane;
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < Q; ++q) { const float t = me[q] * rfx[q * 64]; s = s + t; s = s + ie[q]; }
      s = wave_sum(s);
      if (lane == 0) outp->null2[x] = s + xfactor;
    }
  }
  __threadfence();
  __builtin_amdgcn_wave_barrier();
#define LD2(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)   /* L2-served: this wave's own stores are visible */
  // ---------------- traceback (uniform control flow; select_e is lane-parallel) ----------------
  int i = L, k = 0, st = 0;      // 0=C 1=E 2=M 3=I 4=D
  int fi = 0, fk = 0, li = 0, lk = 0;
  bool done = false;
  const uint64_t path_at = w.path_off & ~FB_PATH_WITH_PP;
  int32_t *path = path_at ? reinterpret_cast<int32_t *>(ws) + (path_at - 1) : nullptr;    // alignment requests: residue of every match state
  // ... followed, when the posterior rows have a matrix of their own (they survive the OA fill), by L + 1 floats: the posterior probability of
  // every residue on the path in the state that emits it (what hmmsearch prints as the PP line of a domain alignment)
  // (the L + 1 posterior floats behind the path exist only where the host reserved them -- FB_PATH_WITH_PP -- and need rows that outlive the fill)
  float *ppres = (path && (w.path_off & FB_PATH_WITH_PP) && w.mxb_off != w.mxf_off) ? reinterpret_cast<float *>(path + Mp) : nullptr;
  if (path) { for (int c = lane; c < Mp; c += 64) path[c] = 0; if (ppres) for (int c = lane; c <= L; c += 64) ppres[c] = 0.f; __threadfence(); __builtin_amdgcn_wave_barrier(); }
#define tBM_(c) tr.at(0, (c))
#define tMM_(c) tr.at(1, (c))
#define tIM_(c) tr.at(2, (c))
#define tDM_(c) tr.at(3, (c))
#define tMI_(c) tr.at(4, (c))
#define tII_(c) tr.at(5, (c))
#define tMD_(c) tr.at(6, (c))
#define tDD_(c) tr.at(7, (c))
  int guard = 0;
  while (!done && guard++ < 4 * (L + Mp) + 16) {
    const float *cr = oa + (size_t)i * 3 * Mp;
    const float *pr = (i > 0) ? oa + (size_t)(i - 1) * 3 * Mp : oa;
    if (st == 0) {
      if (i == 0) { done = true; }
      else { const float a = LD2(&oax[(size_t)(i - 1) * 5 + 4]) + (XL ? LD2X(&aux[(size_t)i * 3 + 2]) : aux[(size_t)i * 3 + 2]), b = LD2(&oax[(size_t)i * 5 + 2]); if (a >= b) --i; else st = 1; }
    } else if (st == 1) {
      const float e = LD2(&oax[(size_t)i * 5 + 2]);
      int best = Mp;
#pragma unroll
      for (int q = Q - 1; q >= 0; --q) { const int c = c0 + q; if (c < M && LD2(&cr[q * 64 + lane]) == e) best = c; }
      best = wave_min(best);
      if (best >= Mp) done = true; else { k = best; st = 2; li = i; lk = k + 1; }
    } else if (st == 2) {
      fi = i; fk = k + 1;
      if (path && lane == 0) { path[k] = i; if (ppres) ppres[i] = LD2(&pp[(size_t)i * pst + lds_cell<Q>(k)]); }
      float p0 = NEGINF_F, p1 = NEGINF_F, p2 = NEGINF_F, p3 = NEGINF_F;
      if (k > 0) { if (tMM_(k) == 0.f) p0 = LD2(&pr[lds_cell<Q>(k - 1)]); if (tIM_(k) == 0.f) p1 = LD2(&pr[Mp + lds_cell<Q>(k - 1)]); if (tDM_(k) == 0.f) p2 = LD2(&pr[2 * Mp + lds_cell<Q>(k - 1)]); }
      if (tBM_(k) == 0.f) p3 = LD2(&oax[(size_t)(i - 1) * 5 + 1]);
      int best = 0; float bv = p0;
      if (p1 > bv) { bv = p1; best = 1; }
      if (p2 > bv) { bv = p2; best = 2; }
      if (p3 > bv) { bv = p3; best = 3; }
      --i;
      if (best == 0) { --k; st = 2; } else if (best == 1) { --k; st = 3; } else if (best == 2) { --k; st = 4; } else done = true;
    } else if (st == 3) {
      if (ppres && lane == 0) ppres[i] = LD2(&pp[(size_t)i * pst + Mp + lds_cell<Q>(k)]);
      const float a = (tMI_(k) == 0.f) ? LD2(&pr[lds_cell<Q>(k)]) : NEGINF_F, b = (tII_(k) == 0.f) ? LD2(&pr[Mp + lds_cell<Q>(k)]) : NEGINF_F;
      --i; st = (a >= b) ? 2 : 3;
    } else {
      const float a = (tMD_(k - 1) == 0.f) ? LD2(&cr[lds_cell<Q>(k - 1)]) : NEGINF_F, b = (tDD_(k - 1) == 0.f) ? LD2(&cr[2 * Mp + lds_cell<Q>(k - 1)]) : NEGINF_F;
      --k; st = (a >= b) ? 2 : 4;
    }
  }
  if (lane == 0) {
    EnvOut &o = *outp;
    o.range_err = 0; o.oasc = oC;
    o.hmm_from = fk; o.hmm_to = lk; o.ali_from = fi + w.i0; o.ali_to = li + w.i0;
  }
}

template <int Q>
__global__ void __launch_bounds__(64) oa_kernel(WorkQueue queue, const FbWork *__restrict__ work,
                                               const DevModel *__restrict__ models, float *__restrict__ ws,
                                               const int32_t *__restrict__ range_err, const FwdOut *__restrict__ fout, EnvOut *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  uint32_t cur_model = 0xffffffffu;
 const uint32_t nqueue = queue_len(queue);
 for (uint32_t qk = blockIdx.x; qk < nqueue; qk += gridDim.x) {
  const uint32_t item = queue.list[qk];
  const FbWork w = work[item];
  const DevModel &md = models[w.model];
  if (w.model != cur_model) { load_gates<Q>(lds, md.ftr); cur_model = w.model; }
  if (lane == 0) { out[w.slot].xC = fout[w.slot].xC; out[w.slot].nscale = fout[w.slot].nscale; }     // the envelope's Forward result travels with its record
  if (range_err[w.slot]) { if (lane == 0) out[w.slot].range_err = 1; continue; }
  oa_item<Q, false>(w, md, lds, lane, ws, &out[w.slot]);
 }
}

#define CKM_FB_QS(X) X(1) X(2) X(3) X(4) X(6) X(8) X(12) X(16) X(24) X(32) X(48) X(64)

// the transition image of a model beyond 2048 nodes (Q = 48, 64) is 96 / 128 KB of LDS: above the 64 KB a launch may ask for by default
template <class K>
static void allow_big_lds(K kernel, int Q) {
  if (Q >= 48) (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * Q * 64 * 4);
}

int launch_fwd(int Q, uint32_t nblocks, hipStream_t stream, WorkQueue queue, FbWork *work, const DevModel *models,
               const LenEntry *lentab, const uint8_t *res, const uint64_t *seq_off, float *ws, FwdOut *out,
               ScaleEvent *events, uint32_t *nevents, uint32_t cap_events, const CascadeDev *cd) {
  if (!nblocks) return 0;
  CascadeDev c; memset(&c, 0, sizeof(c));
  if (cd) c = *cd;
  switch (Q) {
#define X(QV) case QV: allow_big_lds(fwd_kernel<QV>, QV); hipLaunchKernelGGL(fwd_kernel<QV>, dim3(nblocks), dim3(64), (size_t)8 * QV * 64 * 4, stream, queue, work, models, lentab, res, seq_off, ws, out, events, nevents, cap_events, c, cd ? 1 : 0); break;
    CKM_FB_QS(X)
#undef X
    default: return -1;
  }
  return 0;
}
int launch_bwd(int Q, uint32_t nblocks, hipStream_t stream, WorkQueue queue, const FbWork *work, const DevModel *models,
               const LenEntry *lentab, const uint8_t *res, const uint64_t *seq_off, float *ws, const FwdOut *fout, int32_t *range_err) {
  if (!nblocks) return 0;
  switch (Q) {
#define X(QV) case QV: allow_big_lds(bwd_kernel<QV>, QV); hipLaunchKernelGGL(bwd_kernel<QV>, dim3(nblocks), dim3(64), (size_t)8 * QV * 64 * 4, stream, queue, work, models, lentab, res, seq_off, ws, fout, range_err); break;
    CKM_FB_QS(X)
#undef X
    default: return -1;
  }
  return 0;
}
int launch_oa(int Q, uint32_t nblocks, hipStream_t stream, WorkQueue queue, const FbWork *work, const DevModel *models,
              float *ws, const int32_t *range_err, const FwdOut *fout, EnvOut *out) {
  if (!nblocks) return 0;
  switch (Q) {
#define X(QV) case QV: allow_big_lds(oa_kernel<QV>, QV); hipLaunchKernelGGL(oa_kernel<QV>, dim3(nblocks), dim3(64), (size_t)8 * QV * 64 * 4, stream, queue, work, models, ws, range_err, fout, out); break;
    CKM_FB_QS(X)
#undef X
    default: return -1;
  }
  return 0;
}


}  // namespace ckm
