// cascade_regions.h -- the posterior domain heuristics of the device-driven cascade as device functions (used by the fused parser kernel
// in kernels_fb.hip and by region_kernel in kernels_cascade.hip).  See kernels_cascade.hip for what is replaced and why.
#pragma once
#include <hip/hip_runtime.h>
#include "dev_types.h"
#include "cascade_dev.h"
#include "xlane.h"

namespace ckm {

#ifndef CKM_LD2
#define CKM_LD2(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)   /* L2-served: this wave's own stores are visible */
#endif

constexpr float RT1_F = 0.25f, RT2_F = 0.10f, RT3_F = 0.20f;
constexpr int ENS_CAP0 = 16;         // segment slots per trace on the first attempt (the host repeats a region that needs more)

__device__ __forceinline__ unsigned long long al32(unsigned long long v) { return (v + 31ull) & ~31ull; }

__device__ __forceinline__ void emit_region(const CascadeDev &cd, const DevModel &md, const FbWork &pw, int L, int ri, int rj, bool multi) {
  const uint32_t rid = atomicAdd(&cd.gcnt[CC_REG], 1u);
  if (rid >= cd.cap_reg) { atomicOr(&cd.gcnt[CC_STATUS], (uint32_t)CS_REG); return; }
  RegionRec rec; rec.pass = pw.pass; rec.i = ri; rec.j = rj; rec.multi = multi ? 1 : 0; rec.target = 0xffffffffu; rec.pad = 0;
  const unsigned long long Mp = (unsigned long long)md.fbQ * 64ull, Ld = (unsigned long long)(rj - ri + 1);
  unsigned long long off;
  if (!multi) {
    // layout of env_floats() (ckm_stages.hip): specials, decoding terms + OA specials, Forward matrix (3 arrays), posterior matrix (2 arrays)
    unsigned long long pos = 0;
    const unsigned long long xs = pos; pos = al32(pos + (Ld + 1) * 6);
    const unsigned long long aux = pos; pos = al32(al32(pos + (Ld + 1) * 3) + (Ld + 1) * 5);
    const unsigned long long mf = pos; pos = al32(pos + (Ld + 1) * 3 * Mp);
    unsigned long long mb = mf;                                                    // posterior rows in place over the Forward rows ...
    if (!cd.env_inplace) { mb = pos; pos = al32(pos + (Ld + 1) * 2 * Mp); }           // ... or in a matrix of their own
    if (!ws2_alloc(cd, pos, off)) rec.target = REGION_DEFERRED;
    else {
      const uint32_t e = atomicAdd(&cd.gcnt[CC_EWORK], 1u);
      if (e < cd.cap_ework) {
        FbWork w;
        w.model = pw.model; w.seq = pw.seq; w.i0 = ri - 1; w.Ld = (int32_t)Ld; w.Lcfg = L; w.multihit = 0;
        w.xs_off = off + xs; w.aux_off = off + aux; w.mxf_off = off + mf; w.mxb_off = off + mb; w.path_off = 0;
        w.slot = e; w.full = 1; w.cand = pw.cand; w.pass = pw.pass;
        cd.ework[e] = w;
        queue_push(cd, cd.eq, CC_EQ, md.fb_cls, cd.cap_eq, e, (uint32_t)CS_EWORK);
        rec.target = e;
      } else atomicOr(&cd.gcnt[CC_STATUS], (uint32_t)CS_EWORK);
    }
  } else {
    const unsigned long long cap = Ld < (unsigned long long)ENS_CAP0 ? Ld : (unsigned long long)ENS_CAP0;
    // results (counts, segments, sums) contiguous, then the multihit Forward of the region and the per-trace tables (ens_queue_batch)
    unsigned long long pos = 0;
    const unsigned long long nseg = pos; pos += 256;
    const unsigned long long seg = pos; pos += (unsigned long long)ENS_NSAMPLES * cap * 4;
    const unsigned long long n2 = pos; pos = al32(pos + Ld);
    const unsigned long long nres = pos;
    const unsigned long long xs = pos; pos = al32(pos + (Ld + 1) * 6);
    const unsigned long long mx = pos; pos = al32(pos + (Ld + 1) * 4 * Mp);      // cell-major rows of float4 {M, I, D, 0}
    const unsigned long long code = pos; pos = al32(pos + ((unsigned long long)ENS_NSAMPLES * (Ld + 1) + 1) / 2);
    const unsigned long long ratio = pos; pos = al32(pos + (unsigned long long)ENS_NSAMPLES * (Ld + 1));
    if (!ws2_alloc(cd, pos, off)) rec.target = REGION_DEFERRED;
    else {
      const uint32_t r = atomicAdd(&cd.gcnt[CC_RWORK], 1u);
      if (r < cd.cap_rwork) {
        EnsWork e;
        e.model = pw.model; e.seq = pw.seq; e.i0 = ri - 1; e.Ld = (int32_t)Ld; e.Lcfg = L; e.cap = (int32_t)cap;
        e.xs_off = off + xs; e.mx_off = off + mx; e.code_off = off + code; e.ratio_off = off + ratio;
        e.seg_off = off + seg; e.nseg_off = off + nseg; e.n2_off = off + n2;
        const unsigned long long hoff = atomicAdd(cd.hens_top, nres);
        e.host_off = (hoff + nres <= cd.hens_cap) ? hoff : ~0ull;
        if (e.host_off == ~0ull) atomicOr(&cd.gcnt[CC_STATUS], (uint32_t)CS_RWORK);
        rec.pad = (e.host_off == ~0ull) ? 0xffffffffu : (uint32_t)hoff;
        cd.ens[r] = e;
        FbWork w;
        w.model = pw.model; w.seq = pw.seq; w.i0 = ri - 1; w.Ld = (int32_t)Ld; w.Lcfg = L; w.multihit = 1;
        w.xs_off = e.xs_off; w.aux_off = 0; w.mxf_off = e.mx_off; w.mxb_off = 0; w.path_off = 0;
        w.slot = r; w.full = 2; w.cand = pw.cand; w.pass = pw.pass;
        cd.rwork[r] = w;
        queue_push(cd, cd.rq, CC_RQ, md.fb_cls, cd.cap_rq, r, (uint32_t)CS_RWORK);
        if (cd.ensq) { const uint32_t k = atomicAdd(cd.ensq_cnt, 1u); if (k < cd.cap_rwork) cd.ensq[k] = r; }      // the part's trace-ensemble launch
        rec.target = r;
      } else atomicOr(&cd.gcnt[CC_STATUS], (uint32_t)CS_RWORK);
    }
  }
  cd.h_reg[rid] = rec;
}

// One wavefront scans the decoding terms of one parser item (64 rows at a time; the running sums and the trigger logic in residue
// order on values broadcast from their lanes, uniform control flow), writes the prefix sums back in place, and emits the regions.
// XL: the Backward pass that left the terms ran in this very wavefront (fused parser kernel): L2-scope loads.
template <bool XL>
__device__ __forceinline__ void region_scan(const CascadeDev &cd, const DevModel &md, const FbWork &w, float *__restrict__ ws, int lane) {
  const int L = w.Ld;
  float *aux = ws + w.aux_off;           // row r (1..L): [begin term, end term, N/J/C occupancy]; becomes [btot, etot, .]
  float btp = 0.f, etp = 0.f;            // btot / etot of the row before the current one
  int ri = -1; bool trig = false;
  for (int base = 1; base <= L; base += 64) {
    const int row = base + lane;
    const bool ok = row <= L;
    const float bt = ok ? (XL ? CKM_LD2(&aux[(size_t)row * 3]) : aux[(size_t)row * 3]) : 0.f, et = ok ? (XL ? CKM_LD2(&aux[(size_t)row * 3 + 1]) : aux[(size_t)row * 3 + 1]) : 0.f,
                nj = ok ? (XL ? CKM_LD2(&aux[(size_t)row * 3 + 2]) : aux[(size_t)row * 3 + 2]) : 0.f;
    const int nrow = min(64, L - base + 1);
    // running sums in residue order; lane r keeps row base + r's
    const float btp0 = btp, etp0 = etp;
    float pb = 0.f, pe = 0.f;
    for (int r = 0; r < nrow; ++r) {
      btp = btp + read_lane(bt, r); etp = etp + read_lane(et, r);
      if (lane == r) { pb = btp; pe = etp; }
    }
    if (ok) { aux[(size_t)row * 3] = pb; aux[(size_t)row * 3 + 1] = pe; }
    __threadfence();
    __builtin_amdgcn_wave_barrier();
    float pbm = btp0, pem = etp0;
    for (int r = 0; r < nrow; ++r) {
      const float btn = read_lane(pb, r), etn = read_lane(pe, r), mo = 1.0f - read_lane(nj, r);
      const int j = base + r;
      if (!trig) {
        if (mo - (btn - pbm) < RT2_F) ri = j; else if (ri == -1) ri = j;
        if (mo >= RT1_F) trig = true;
      } else if (mo - (etn - pem) < RT2_F) {
        const float e0 = (ri - 1 >= 1) ? CKM_LD2(&aux[(size_t)(ri - 1) * 3 + 1]) : 0.f;       // etot[ri-1]
        float mx = -1.0f;
        for (int z = ri + lane; z <= j; z += 64) {
          const float bz = (z - 1 >= 1) ? CKM_LD2(&aux[(size_t)(z - 1) * 3]) : 0.f;           // btot[z-1]
          const float a = CKM_LD2(&aux[(size_t)z * 3 + 1]) - e0, b = btn - bz;
          const float en = a < b ? a : b;
          if (en > mx) mx = en;
        }
        mx = wave_max(mx);
        if (lane == 0) emit_region(cd, md, w, L, ri, j, mx >= RT3_F);
        ri = -1; trig = false;
      }
      pbm = btn; pem = etn;
    }
  }
}

}  // namespace ckm
