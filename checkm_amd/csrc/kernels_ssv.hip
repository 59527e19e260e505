// kernels_ssv.hip -- the dominant kernel: single-segment ungapped Viterbi (SSV) over every
// (model, sequence) pair, gfx950 only.
//
// What it replaces: HMMER's MSV/SSV filter stage of the hmmsearch process CheckM launches per bin
// (checkm/markerGeneFinder.py:140-142 -> checkm/hmmer.py:70).  MSV in unsigned bytes is
//     sv(i,k) = sat_sub(sat_add(max(sv(i-1,k-1), xB), bias), cost[x_i][k])
// and while the running xJ never exceeds `base`, xB is constant, so with U = max(sv - xB, 0):
//     U(i,k) = max(U(i-1,k-1) + (bias - cost[x_i][k]), 0);   Smax = max U
// is independent of the target length; xE = xB + Smax reproduces the byte arithmetic exactly whenever
// Smax > 0 (sat_add cannot saturate before the overflow test fires).  Pairs
// with Smax == 0 (no cell ever scored above xB: degenerate targets) and pairs for which the J state
// could have been used (xJ > base) are re-run by the exact MSV kernels (kernels_filter.hip).
//
// VALU budget: on gfx950 every VALU op except f32 add/mul/fma issues at 4 cycles per wave64
// (tools/ubench/valu_rates.hip), so the kernel is bound by instruction COUNT.  A byte score k is held as
// the HALF float k/256 (exact), v_pk_add_f16 clamp is the floored, saturating add, v_pk_maximum3_f16 takes
// the running maximum of two rows at once: 1.5 packed ops per register per row.
//
// Mapping: 16 lanes per sequence (one DPP row), 4 sequences per wavefront -- or 8 lanes per sequence, 8 per
// wavefront for models of <= 512 nodes -- Q packed registers of two cells per lane; position
// p = q + Q*(2*lane16 + half) so the diagonal move i-1,k-1 -> i,k is a register rename plus ONE row_shr DPP
// move per row.  Emission words of the model live in LDS as [Q/4][symbol][16 lanes][16 B]: each row step is
// ceil(Q/4) conflict-free ds_read_b128 per lane (address = one v_perm_b32 of the residue word over the lane
// offset, the group index in the offset field).
//
// Round 4: the lane that holds a pair's Smax finishes the MSV stage itself (ssv_finish below) -- byte score,
// overflow / J-state / Smax == 0 routing, F1 -- and appends the pair to the survivor or the exact-MSV table;
// rounds 1-3 wrote Smax to HBM and launched a thread-per-pair kernel over it (msv_finish_kernel: 97 % of its
// cycles waiting, as many waves as SSV itself).  Smax goes to memory only for ckm_debug_stages.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include "dev_types.h"

namespace ckm {

typedef unsigned int u32;

constexpr int SSV_NROWS = 30;
constexpr u32 PAD4 = 0x1d1d1d1du;   // four PADCODE (29) residues

// LDS address of a residue's emission words: symbol * 256 + lane16 * 16, formed by ONE v_perm_b32 that drops byte B of the
// residue word into byte 1 above the lane offset (symbols < 32, so the image of register group g starts at g * SSV_GSTRIDE
// and the group index travels in the ds_read offset field).
constexpr int SSV_GSTRIDE = SSV_NROWS * 256;
template <int B>
__device__ __forceinline__ u32 ssv_addr(u32 word, u32 lane_off) { return __builtin_amdgcn_perm(word, lane_off, 0x0c0c0400u + ((u32)B << 8)); }

// The emission image is addressed ABSOLUTELY in LDS (it is the only LDS these kernels use, so the dynamic segment starts
// at 0; lds_image_at_zero() traps otherwise): the permuted word IS the ds_read address, no base has to be added per row.
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x4 lds_cu4;
typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ void lds_image_at_zero(char *smem) { if ((u32)(size_t)(lds_char *)smem != 0u) __builtin_trap(); }

// --------------------------------------------------------------------------------------------------------------------
// Round 3: the same recurrence in packed HALF floats, 1.5 VALU ops per register per row instead of 2.
// A byte score k (0..255, and 256 = "saturated") is held as the f16 value k/256: every such value, and every sum of two of them,
// is exact in f16 (11 significant bits), so this is still the integer arithmetic.  v_pk_add_f16 with the clamp modifier clamps to
// [0, 1]: the floor at 0 of the recurrence AND the byte ceiling (a score that reaches 256 units has overflowed the byte MSV, which
// passes the filter whatever happens afterwards -- ssv_finish's overflow test fires for every Smax >= 255 - bias - xB).
// gfx950 has v_pk_maximum3_f16, so the running maximum is taken once per TWO rows: Smax = max3(Smax, U(row a), U(row b)).
// Rows go in pairs; sequences are padded to a multiple of 16 rows with the all-impossible symbol, so no remainder exists.
// --------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 pk_add_h_clamp(u32 a, u32 b) { u32 r; asm("v_pk_add_f16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ u32 pk_max3_h(u32 a, u32 b, u32 c) { u32 r; asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

template <int Q>
__device__ __forceinline__ void ssv_load_row(u32 (&e)[((Q + 3) / 4) * 4], u32 addr) {
  constexpr int Qg = (Q + 3) / 4;
#pragma unroll
  for (int g = 0; g < Qg; ++g) {
    const u32x4 v = *(lds_cu4 *)(size_t)(addr + (u32)(g * SSV_GSTRIDE));
    e[g * 4 + 0] = v.x; e[g * 4 + 1] = v.y; e[g * 4 + 2] = v.z; e[g * 4 + 3] = v.w;
  }
}

template <int Q, int SHR = 0x111 /* row_shr:1; 0x112 = row_shr:2 for the 8-lane mapping */>
__device__ __forceinline__ void ssv_rows2_h(u32 (&U)[Q], u32 &xE, u32 &prev, u32 addr_a, u32 addr_b) {
  constexpr int Qg = (Q + 3) / 4;
  u32 e[Qg * 4];
  // row a: in place, no maximum
  ssv_load_row<Q>(e, addr_a);
  {
    const u32 last = U[Q - 1];
    prev = (u32)__builtin_amdgcn_update_dpp((int)prev, (int)last, SHR, 0xf, 0xf, false);
    const u32 carry = __builtin_amdgcn_alignbit(last, prev, 16);
#pragma unroll
    for (int q = Q - 1; q >= 1; --q) U[q] = pk_add_h_clamp(U[q - 1], e[q]);
    U[0] = pk_add_h_clamp(carry, e[0]);
  }
  // row b: U[q] still holds row a's value when its successor is formed, so one max3 covers both rows
  ssv_load_row<Q>(e, addr_b);
  {
    const u32 last = U[Q - 1];
    prev = (u32)__builtin_amdgcn_update_dpp((int)prev, (int)last, SHR, 0xf, 0xf, false);
    const u32 carry = __builtin_amdgcn_alignbit(last, prev, 16);
#pragma unroll
    for (int q = Q - 1; q >= 1; --q) {
      const u32 v = pk_add_h_clamp(U[q - 1], e[q]);
      xE = pk_max3_h(xE, U[q], v);
      U[q] = v;
    }
    const u32 v = pk_add_h_clamp(carry, e[0]);
    xE = pk_max3_h(xE, U[0], v);
    U[0] = v;
  }
}

// --------------------------------------------------------------------------------------------------------------------
// The MSV stage's finish, run by the lane that holds the pair's Smax (one per sequence: 4 or 8 lanes of a wavefront at a time).
// Integer routing exactly as the byte filter decides it; the score in nats is formed with the same IEEE operations the host uses
// (cvt, sub, correctly rounded div, sub); the F1 test "bits >= thr" is taken in nats: DevModel::thr_msv_f1_nat is the smallest
// float v with (float)((double)v / ln 2) >= thr_msv_f1 (found on the host by bisection over float bit patterns), and
// v -> (float)((double)v / ln 2) is monotone, so (usc - nullsc) >= thr_msv_f1_nat IS the host's test -- without the double
// divide per pair.  Survivors and undecided pairs are appended by atomicAdd (order arbitrary, as before: every later stage is
// per pair and the rows are ordered at the end).
// --------------------------------------------------------------------------------------------------------------------
struct SsvModelScalars { int base_b, bias_b, tbm_b, tec_b; float scale_b, thr_nat; };
__device__ __forceinline__ SsvModelScalars ssv_scalars(const DevModel &md) {
  return SsvModelScalars{md.base_b, md.bias_b, md.tbm_b, md.tec_b, md.scale_b, md.thr_msv_f1_nat};
}
__device__ __forceinline__ void ssv_finish(const SsvEpi &a, const SsvModelScalars &ms, u32 model, u32 sid, int L, int maxV) {
  const LenEntry *le = a.lentab + L;
  const int tjb = le->tjb_b;
  const float nullsc = le->nullsc;
  const int tjbm = (tjb + ms.tbm_b) & 0xff;
  const int xB = max(ms.base_b - tjbm, 0);
  const int xEi = xB + maxV;
  const int xJ = max(xEi - ms.tec_b, 0);
  float usc = (float)(xJ - tjb) - (float)ms.base_b;
  usc = usc / ms.scale_b;
  usc = usc - 3.0f;
  const bool overflow = xEi + ms.bias_b >= 255;               // byte overflow: score is +inf, passes every MSV test
  PairRec r; r.model = model; r.seq = sid; r.filtersc = 0.f;
  if (maxV == 0 || (!overflow && xJ > ms.base_b)) {           // no cell rose above xB, or J could have been used: exact kernel
    r.usc = 0.f;
    const u32 k = atomicAdd(a.nnores, 1u);
    if (k < a.cap_nores) a.noresult[k] = r;
  } else if (overflow || (usc - nullsc) >= ms.thr_nat) {
    r.usc = overflow ? __builtin_inff() : usc;
    const u32 k = atomicAdd(a.nsurv, 1u);
    if (k < a.cap_surv) a.survivors[k] = r;
  }
}

template <int Q>
__global__ void __launch_bounds__(Q > 40 ? 512 : 1024) ssv_kernel_h(const SsvBlockWork *__restrict__ work, const DevModel *__restrict__ models,
                           const uint8_t *__restrict__ res, const uint64_t *__restrict__ seq_off,
                           const int32_t *__restrict__ seq_len, const uint32_t *__restrict__ lists,
                           SsvEpi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int Qg = (Q + 3) / 4;
  constexpr int ROWB = Qg * 256;
  lds_image_at_zero(smem);
  const SsvBlockWork w = work[blockIdx.x];
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(models[w.model].ssv_tbl_h);
    uint4 *dst = reinterpret_cast<uint4 *>(smem);
    for (int i = threadIdx.x; i < SSV_NROWS * ROWB / 16; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int seg = lane >> 4, z = lane & 15;
  const u32 lane_off = (u32)z * 16u;
  for (u32 g = wave; g * 4 < w.count; g += nwaves) {
    const u32 li = g * 4 + seg;
    const bool valid = li < w.count;
    const u32 sid = valid ? lists[w.list_start + li] : 0u;
    const int L = valid ? seq_len[sid] : 0;
    const uint8_t *rp = res + seq_off[sid];
    int Lmax = __builtin_amdgcn_readlane(L, 0);
    Lmax = max(Lmax, __builtin_amdgcn_readlane(L, 16));
    Lmax = max(Lmax, __builtin_amdgcn_readlane(L, 32));
    Lmax = max(Lmax, __builtin_amdgcn_readlane(L, 48));
    u32 U[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) U[q] = 0u;
    u32 xE = 0u, prev = 0u;
    const int nchunk = (Lmax + 15) >> 4;
    const uint4 padv = make_uint4(PAD4, PAD4, PAD4, PAD4);
    uint4 cur = padv;
    if (0 < L) cur = *reinterpret_cast<const uint4 *>(rp);
    for (int c = 0; c < nchunk; ++c) {
      uint4 nxt = padv;
      if ((c + 1) * 16 < L) nxt = *reinterpret_cast<const uint4 *>(rp + (c + 1) * 16);
#define CKM_SSV_WORD(wd)                                                                     \
  ssv_rows2_h<Q>(U, xE, prev, ssv_addr<0>(wd, lane_off), ssv_addr<1>(wd, lane_off));           \
  ssv_rows2_h<Q>(U, xE, prev, ssv_addr<2>(wd, lane_off), ssv_addr<3>(wd, lane_off));
      CKM_SSV_WORD(cur.x) CKM_SSV_WORD(cur.y) CKM_SSV_WORD(cur.z) CKM_SSV_WORD(cur.w)
#undef CKM_SSV_WORD
      cur = nxt;
    }
    // k/256 back to k: the f16 bit patterns of non-negative values order like integers, so the maximum is taken on the bits
    u32 mb = max(xE & 0xffffu, xE >> 16);
    mb = max(mb, (u32)__shfl_xor((int)mb, 1, 16));
    mb = max(mb, (u32)__shfl_xor((int)mb, 2, 16));
    mb = max(mb, (u32)__shfl_xor((int)mb, 4, 16));
    mb = max(mb, (u32)__shfl_xor((int)mb, 8, 16));
    const _Float16 hv = __builtin_bit_cast(_Float16, (unsigned short)mb);
    if (valid && z == 0 && L > 0) {
      const int maxV = (int)((float)hv * 256.0f);
      if (epi.maxv) epi.maxv[w.pair_start + li] = (uint16_t)maxV;
      else ssv_finish(epi, ssv_scalars(models[w.model]), w.model, lists[w.list_start + li] /* (read again: not carried through the row loop) */, L, maxV);
    }
  }
}

// --------------------------------------------------------------------------------------------------------------------
// The packed-half row on EIGHT lanes per sequence, eight sequences per wavefront, for models of up to 512 nodes (round 3).  The row's
// fixed part (address, DPP move, alignbit) and the padding to a whole register per lane are what short models pay most for: 1.5 Q + 3.25
// instructions per four sequence-rows at Q = ceil(M / 32).  Here a DPP row holds TWO sequences, interleaved (even lanes one, odd lanes the
// other), so "the previous lane of my sequence" is row_shr:2 -- lanes 0 and 1 of a row have no source and keep U = 0 -- and a sequence's
// 16 stripes of Q8 = ceil(M / 16) cells give 1.5 Q8 + 3.25 instructions per EIGHT sequence-rows.  The LDS image keeps 256 bytes per
// (register group, symbol): the eight 16-byte columns of the model, twice, so that the odd lanes read the upper copy and the 16 lanes of a
// row still hit 16 different bank groups whatever the two residues are.
// --------------------------------------------------------------------------------------------------------------------
template <int Q>
__global__ void __launch_bounds__(Q > 40 ? 512 : 1024) ssv_kernel_h8(const SsvBlockWork *__restrict__ work, const DevModel *__restrict__ models,
                           const uint8_t *__restrict__ res, const uint64_t *__restrict__ seq_off,
                           const int32_t *__restrict__ seq_len, const uint32_t *__restrict__ lists,
                           SsvEpi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int Qg = (Q + 3) / 4;
  constexpr int ROWB = Qg * 256;
  lds_image_at_zero(smem);
  const SsvBlockWork w = work[blockIdx.x];
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(models[w.model].ssv8_tbl_h);
    uint4 *dst = reinterpret_cast<uint4 *>(smem);
    for (int i = threadIdx.x; i < SSV_NROWS * ROWB / 16; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int z16 = lane & 15, sub = z16 & 1, z = z16 >> 1;
  const u32 lane_off = (u32)sub * 128u + (u32)z * 16u;
  for (u32 g = wave; g * 8 < w.count; g += nwaves) {
    const u32 li = g * 8 + (u32)(lane >> 4) * 2u + (u32)sub;
    const bool valid = li < w.count;
    const u32 sid = valid ? lists[w.list_start + li] : 0u;
    const int L = valid ? seq_len[sid] : 0;
    const uint8_t *rp = res + seq_off[sid];
    int Lmax = max(__builtin_amdgcn_readlane(L, 0), __builtin_amdgcn_readlane(L, 1));
    Lmax = max(Lmax, max(__builtin_amdgcn_readlane(L, 16), __builtin_amdgcn_readlane(L, 17)));
    Lmax = max(Lmax, max(__builtin_amdgcn_readlane(L, 32), __builtin_amdgcn_readlane(L, 33)));
    Lmax = max(Lmax, max(__builtin_amdgcn_readlane(L, 48), __builtin_amdgcn_readlane(L, 49)));
    u32 U[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) U[q] = 0u;
    u32 xE = 0u, prev = 0u;
    const int nchunk = (Lmax + 15) >> 4;
    const uint4 padv = make_uint4(PAD4, PAD4, PAD4, PAD4);
    uint4 cur = padv;
    if (0 < L) cur = *reinterpret_cast<const uint4 *>(rp);
    for (int c = 0; c < nchunk; ++c) {
      uint4 nxt = padv;
      if ((c + 1) * 16 < L) nxt = *reinterpret_cast<const uint4 *>(rp + (c + 1) * 16);
#define CKM_SSV_WORD(wd)                                                                            \
  ssv_rows2_h<Q, 0x112>(U, xE, prev, ssv_addr<0>(wd, lane_off), ssv_addr<1>(wd, lane_off));           \
  ssv_rows2_h<Q, 0x112>(U, xE, prev, ssv_addr<2>(wd, lane_off), ssv_addr<3>(wd, lane_off));
      CKM_SSV_WORD(cur.x) CKM_SSV_WORD(cur.y) CKM_SSV_WORD(cur.z) CKM_SSV_WORD(cur.w)
#undef CKM_SSV_WORD
      cur = nxt;
    }
    u32 mb = max(xE & 0xffffu, xE >> 16);
    mb = max(mb, (u32)__shfl_xor((int)mb, 2, 16));
    mb = max(mb, (u32)__shfl_xor((int)mb, 4, 16));
    mb = max(mb, (u32)__shfl_xor((int)mb, 8, 16));
    const _Float16 hv = __builtin_bit_cast(_Float16, (unsigned short)mb);
    if (valid && z == 0 && L > 0) {
      const int maxV = (int)((float)hv * 256.0f);
      if (epi.maxv) epi.maxv[w.pair_start + li] = (uint16_t)maxV;
      else ssv_finish(epi, ssv_scalars(models[w.model]), w.model, lists[w.list_start + li] /* (read again: not carried through the row loop) */, L, maxV);
    }
  }
}

// Models beyond 2048 nodes have no SSV instance (the LDS image of their emission words would not fit): every one of their pairs goes
// straight to the exact-MSV table (what Smax == 0, "no cell rose above xB: recompute exactly", does for any other pair); the exact kernel
// takes any model length that fits 160 KB of LDS.  Exact, just slow -- such models are a handful of a full Pfam / TIGRFAM file.
__global__ void ssv_none_kernel(const SsvBlockWork *__restrict__ work, const int32_t *__restrict__ seq_len, const uint32_t *__restrict__ lists, SsvEpi epi) {
  const SsvBlockWork w = work[blockIdx.x];
  for (uint32_t li = threadIdx.x; li < w.count; li += blockDim.x) {
    if (epi.maxv) { epi.maxv[w.pair_start + li] = 0; continue; }
    const uint32_t sid = lists[w.list_start + li];
    if (seq_len[sid] <= 0) continue;
    PairRec r; r.model = w.model; r.seq = sid; r.usc = 0.f; r.filtersc = 0.f;
    const u32 k = atomicAdd(epi.nnores, 1u);
    if (k < epi.cap_nores) epi.noresult[k] = r;
  }
}

// the dynamic-LDS limit of an instance is raised once PER DEVICE (the attribute belongs to the current device's code object), under a
// lock: searches of several contexts launch from different host threads
static std::mutex g_attr_mutex;
template <class K>
static void raise_lds_limit(K kernel, size_t bytes, uint32_t &done_devices) {
  if (bytes <= 48 * 1024) return;
  int dev = 0; (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(g_attr_mutex);
  if (!(done_devices & (1u << (dev & 31)))) { (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); done_devices |= 1u << (dev & 31); }
}

#define CKM_SSV_CASE(QV)                                                                                         \
  case QV: {                                                                                                     \
    static uint32_t attr_set = 0;                                                                                  \
    raise_lds_limit(ssv_kernel_h<QV>, (size_t)SSV_NROWS * ((QV + 3) / 4) * 256, attr_set);                       \
    hipLaunchKernelGGL(ssv_kernel_h<QV>, dim3(nblocks), dim3(threads), (size_t)SSV_NROWS * ((QV + 3) / 4) * 256, stream, \
                       work, models, res, seq_off, seq_len, lists, epi);                                         \
  } break;

int launch_ssv(int Q, int nblocks, int threads, hipStream_t stream, const SsvBlockWork *work, const DevModel *models,
               const uint8_t *res, const uint64_t *seq_off, const int32_t *seq_len, const uint32_t *lists, const SsvEpi &epi) {
  if (nblocks <= 0) return 0;
  if (Q >= 100) {                      // launch classes 100 + Q8: eight lanes per sequence
    switch (Q - 100) {
#define X(QV) case QV: {                                                                                         \
      static uint32_t attr8 = 0;                                                                                 \
      raise_lds_limit(ssv_kernel_h8<QV>, (size_t)SSV_NROWS * ((QV + 3) / 4) * 256, attr8);                       \
      hipLaunchKernelGGL(ssv_kernel_h8<QV>, dim3(nblocks), dim3(threads), (size_t)SSV_NROWS * ((QV + 3) / 4) * 256, stream, work, models, res, seq_off, seq_len, lists, epi); \
    } break;
      X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(32)
#undef X
      default: return -1;
    }
    return 0;
  }
  if (Q > 64) { hipLaunchKernelGGL(ssv_none_kernel, dim3(nblocks), dim3(64), 0, stream, work, seq_len, lists, epi); return 0; }
  switch (Q) {
    CKM_SSV_CASE(1) CKM_SSV_CASE(2) CKM_SSV_CASE(3) CKM_SSV_CASE(4) CKM_SSV_CASE(5) CKM_SSV_CASE(6) CKM_SSV_CASE(7)
    CKM_SSV_CASE(8) CKM_SSV_CASE(9) CKM_SSV_CASE(10) CKM_SSV_CASE(11) CKM_SSV_CASE(12) CKM_SSV_CASE(13) CKM_SSV_CASE(14)
    CKM_SSV_CASE(15) CKM_SSV_CASE(16) CKM_SSV_CASE(18) CKM_SSV_CASE(20) CKM_SSV_CASE(22) CKM_SSV_CASE(24) CKM_SSV_CASE(26)
    CKM_SSV_CASE(28) CKM_SSV_CASE(30) CKM_SSV_CASE(32) CKM_SSV_CASE(36) CKM_SSV_CASE(40) CKM_SSV_CASE(48) CKM_SSV_CASE(56)
    CKM_SSV_CASE(64)
    default: return -1;
  }
  return 0;
}

}  // namespace ckm
