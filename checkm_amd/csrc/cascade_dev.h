// cascade_dev.h -- device-side plumbing of the device-driven filter cascade (gfx950 only): work queues consumed by persistent
// wavefronts, the bump allocator of the float workspace, and the hand-over of a pair from one stage to the next.  The stages
// themselves are the epilogues of the kernels in kernels_filter.hip / kernels_fb.hip and the small kernels of kernels_cascade.hip.
//
// Decisions taken on the device are CONSERVATIVE: a P-value test `bits(score, null) >= threshold` whose null score depends on a
// logarithm (bias filter, Forward) is evaluated with the hardware's approximate log2 and the threshold lowered by a margin that
// covers the approximation, so no pair the exact test lets through can be lost here; the host takes every decision again with
// libm on the recorded integers / floats before a row is reported (ckm_cascade.hip), so rows cannot change either.
#pragma once
#include <hip/hip_runtime.h>
#include "dev_types.h"

namespace ckm {

constexpr float LOG2E_F = 1.44269504088896341f, LN2_F = 0.69314718055994529f;

// Entries of a queue: wavefront w of the W a launch starts takes entries w, w + W, w + 2W, ...  (The index comes from blockIdx /
// threadIdx alone, so it is uniform by construction and lives in scalar registers; a dynamic take with atomicAdd + readfirstlane
// was tried first and the compiler's structurizer turned it into a loop that never left the first entry.)
__device__ __forceinline__ uint32_t queue_len(const WorkQueue &q) { return min(*q.count, q.cap); }

// natural logarithm for the conservative tests only (v_log_f32, ~1 ulp in log2)
__device__ __forceinline__ float approx_ln(float x) { return __log2f(x) * LN2_F; }

// bump allocation of n floats (rounded up to 32) from the cascade's workspace; false when it is exhausted (status bit CS_WS)
__device__ __forceinline__ bool ws_alloc(const CascadeDev &cd, unsigned long long n, unsigned long long &off) {
  n = (n + 31ull) & ~31ull;
  off = atomicAdd(cd.ws_top, n);
  if (off + n > cd.ws_cap) { atomicOr(&cd.gcnt[CC_STATUS], (uint32_t)CS_WS); return false; }
  return true;
}

// the second zone (envelope / region matrices): no status bit, the caller defers the region to the host
__device__ __forceinline__ bool ws2_alloc(const CascadeDev &cd, unsigned long long n, unsigned long long &off) {
  n = (n + 31ull) & ~31ull;
  const unsigned long long at = atomicAdd(cd.ws2_top, n);
  if (at + n > cd.ws2_cap) return false;
  off = cd.ws2_base + at;
  return true;
}

__device__ __forceinline__ void queue_push(const CascadeDev &cd, uint32_t *lists, int counter, int cls, uint32_t cap, uint32_t value, uint32_t overflow_bit) {
  const uint32_t pos = atomicAdd(&cd.cnt[counter + cls], 1u);
  if (pos < cap) lists[(size_t)cls * cap + pos] = value; else atomicOr(&cd.gcnt[CC_STATUS], overflow_bit);
}

// A candidate that passed the Viterbi stage (or skipped it) becomes a whole-sequence Forward parser item: special rows in the
// workspace, an FbWork record, a place in the Forward queue of its model's register class.
__device__ __forceinline__ void pass_to_forward(const CascadeDev &cd, const DevModel &md, uint32_t pi, uint32_t model, uint32_t seq) {
  const int L = cd.seq_len[seq];
  unsigned long long off;
  if (!ws_alloc(cd, (unsigned long long)(L + 1) * 6ull, off)) return;
  const uint32_t t = atomicAdd(&cd.gcnt[CC_FWORK], 1u);
  if (t >= cd.cap_fwork) { atomicOr(&cd.gcnt[CC_STATUS], (uint32_t)CS_FWORK); return; }
  FbWork w;
  w.model = model; w.seq = seq; w.i0 = 0; w.Ld = L; w.Lcfg = L; w.multihit = 1;
  w.xs_off = off; w.aux_off = 0; w.mxf_off = 0; w.mxb_off = 0; w.path_off = 0;
  w.slot = t; w.full = 0; w.cand = pi; w.pass = 0xffffffffu;
  cd.fwork[t] = w;
  queue_push(cd, cd.fq, CC_FQ, md.fb_cls, cd.cap_fq, t, (uint32_t)CS_FWORK);
}

}  // namespace ckm
