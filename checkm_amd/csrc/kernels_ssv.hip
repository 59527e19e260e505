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
// could have been used (xJ > base) are re-run by msv_full_kernel.
//
// VALU budget: on gfx950 every VALU op except f32 add/mul/fma issues at 4 cycles per wave64
// (tools/ubench/valu_rates.hip), so the kernel is bound by instruction COUNT.  U is held with a
// -32768 offset: the saturating v_pk_add_i16 then performs the floor at 0 for free, leaving
// 2 packed ops per register per row (add-with-clamp, running max) instead of 3.
//
// Mapping: 16 lanes per sequence (one DPP row), 4 sequences per wavefront, Q packed 2 x i16
// registers per lane; position p = q + Q*(2*lane16 + half) so the diagonal move i-1,k-1 -> i,k is a
// register rename plus ONE row_shr:1 DPP move per row.  Emission words for the model live in LDS
// as [Q/4][symbol][16 lanes][16 B]: each row step is ceil(Q/4) conflict-free ds_read_b128 per lane
// (address = one v_perm_b32 of the residue word over the lane offset, the group index in the offset
// field) and 2 packed-i16 VALU ops per register (v_pk_add_i16 clamp, v_pk_max_i16); per row the
// overhead on top of 2Q is 3 VALU ops (address, DPP move, alignbit) plus a quarter op of chunk bookkeeping.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include "dev_types.h"

namespace ckm {

typedef short  s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32;

__device__ __forceinline__ s16x2 as_s16x2(u32 v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ u32   as_u32(s16x2 v) { return __builtin_bit_cast(u32, v); }

constexpr int SSV_NROWS = 30;
constexpr u32 PAD4 = 0x1d1d1d1du;   // four PADCODE (29) residues
constexpr u32 U_ZERO = 0x80008000u; // two cells with U = 0 in the -32768-offset representation

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

template <int Q>
__device__ __forceinline__ void ssv_row(u32 (&U)[Q], u32 &xE, u32 &prev, u32 addr) {
  constexpr int Qg = (Q + 3) / 4;
  u32 e[Qg * 4];
#pragma unroll
  for (int g = 0; g < Qg; ++g) {
    const u32x4 v = *(lds_cu4 *)(size_t)(addr + (u32)(g * SSV_GSTRIDE));
    e[g * 4 + 0] = v.x; e[g * 4 + 1] = v.y; e[g * 4 + 2] = v.z; e[g * 4 + 3] = v.w;
  }
  // cell p=0 of each lane's first register comes from the previous lane's last register (high half)
  // and this lane's own last register (low half -> high half)
  const u32 last = U[Q - 1];
  // lane 0 of each 16-lane row has no predecessor: it keeps `old`.  `prev` is a persistent register that starts as the
  // offset representation of U = 0 and is only ever overwritten in lanes 1..15, so no per-row re-initialisation is needed
  prev = (u32)__builtin_amdgcn_update_dpp((int)prev, (int)last, 0x111 /*row_shr:1*/, 0xf, 0xf, false);
  const u32 carry = __builtin_amdgcn_alignbit(last, prev, 16);
#pragma unroll
  for (int q = Q - 1; q >= 1; --q) {
    const s16x2 v = __builtin_elementwise_add_sat(as_s16x2(U[q - 1]), as_s16x2(e[q]));   // clamps at -32768 == U 0
    xE = as_u32(__builtin_elementwise_max(as_s16x2(xE), v));
    U[q] = as_u32(v);
  }
  {
    const s16x2 v = __builtin_elementwise_add_sat(as_s16x2(carry), as_s16x2(e[0]));
    xE = as_u32(__builtin_elementwise_max(as_s16x2(xE), v));
    U[0] = as_u32(v);
  }
}

template <int Q>
__global__ void __launch_bounds__(Q > 40 ? 512 : 1024) ssv_kernel(const SsvBlockWork *__restrict__ work, const DevModel *__restrict__ models,
                           const uint8_t *__restrict__ res, const uint64_t *__restrict__ seq_off,
                           const int32_t *__restrict__ seq_len, const uint32_t *__restrict__ lists,
                           uint16_t *__restrict__ maxv) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int Qg = (Q + 3) / 4;
  constexpr int ROWB = Qg * 256;
  lds_image_at_zero(smem);
  const SsvBlockWork w = work[blockIdx.x];
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(models[w.model].ssv_tbl);
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
    for (int q = 0; q < Q; ++q) U[q] = U_ZERO;
    u32 xE = U_ZERO, prev = U_ZERO;
    const int nchunk = (Lmax + 15) >> 4;
    const uint4 padv = make_uint4(PAD4, PAD4, PAD4, PAD4);
    uint4 cur = padv;
    if (0 < L) cur = *reinterpret_cast<const uint4 *>(rp);     // (not a ?: -- that selects between ADDRESSES and spills padv to scratch)
    for (int c = 0; c < nchunk; ++c) {
      uint4 nxt = padv;
      if ((c + 1) * 16 < L) nxt = *reinterpret_cast<const uint4 *>(rp + (c + 1) * 16);
#define CKM_SSV_WORD(wd)                                   \
  ssv_row<Q>(U, xE, prev, ssv_addr<0>(wd, lane_off));            \
  ssv_row<Q>(U, xE, prev, ssv_addr<1>(wd, lane_off));            \
  ssv_row<Q>(U, xE, prev, ssv_addr<2>(wd, lane_off));            \
  ssv_row<Q>(U, xE, prev, ssv_addr<3>(wd, lane_off));
      CKM_SSV_WORD(cur.x) CKM_SSV_WORD(cur.y) CKM_SSV_WORD(cur.z) CKM_SSV_WORD(cur.w)
#undef CKM_SSV_WORD
      cur = nxt;
    }
    const s16x2 xv = as_s16x2(xE);
    int m = max((int)xv.x, (int)xv.y) + 32768;   // back to Smax >= 0
    m = max(m, __shfl_xor(m, 1, 16));
    m = max(m, __shfl_xor(m, 2, 16));
    m = max(m, __shfl_xor(m, 4, 16));
    m = max(m, __shfl_xor(m, 8, 16));
    if (valid && z == 0) maxv[w.pair_start + li] = (uint16_t)m;
  }
}

// --------------------------------------------------------------------------------------------------------------------
// Round 3: the same recurrence in packed HALF floats, 1.5 VALU ops per register per row instead of 2.
// A byte score k (0..255, and 256 = "saturated") is held as the f16 value k/256: every such value, and every sum of two of them,
// is exact in f16 (11 significant bits), so this is still the integer arithmetic.  v_pk_add_f16 with the clamp modifier clamps to
// [0, 1]: the floor at 0 of the recurrence AND the byte ceiling (a score that reaches 256 units has overflowed the byte MSV, which
// passes the filter whatever happens afterwards -- msv_finish_kernel's overflow test fires for every Smax >= 255 - bias - xB).
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

template <int Q>
__global__ void __launch_bounds__(Q > 40 ? 512 : 1024) ssv_kernel_h(const SsvBlockWork *__restrict__ work, const DevModel *__restrict__ models,
                           const uint8_t *__restrict__ res, const uint64_t *__restrict__ seq_off,
                           const int32_t *__restrict__ seq_len, const uint32_t *__restrict__ lists,
                           uint16_t *__restrict__ maxv) {
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
    if (valid && z == 0) maxv[w.pair_start + li] = (uint16_t)(int)((float)hv * 256.0f);
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
                           uint16_t *__restrict__ maxv) {
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
    if (valid && z == 0) maxv[w.pair_start + li] = (uint16_t)(int)((float)hv * 256.0f);
  }
}

// --------------------------------------------------------------------------------------------------------------------
// Exact multi-hit MSV for the pairs SSV cannot decide (xJ may exceed base, or Smax == 0): same lane mapping, same LDS
// emission image, but the byte recurrence is carried in full -- sv = max(prev, xB) + (bias - cost), floored at 0 by the
// clamped add (offset -32768 again) -- and every row ends with the 16-lane maximum that feeds xJ and xB.  sat_add(., bias)
// cannot saturate on a row whose predecessors passed the overflow test, so adding (bias - cost) in one step is the byte
// arithmetic exactly; the overflow flag is sticky and turns the score into +inf, as HMMER's early return does.
// 3 packed ops per register per row + ~25 for the row maximum and the specials.
// --------------------------------------------------------------------------------------------------------------------
template <int Q>
__device__ __forceinline__ void msv_row(u32 (&U)[Q], u32 &xE, u32 &prev, u32 xBv, const char *lds_base, u32 lane_off, u32 x) {
  constexpr int Qg = (Q + 3) / 4;
  const char *rowp = lds_base + x * 256u + lane_off;
  u32 e[Qg * 4];
#pragma unroll
  for (int g = 0; g < Qg; ++g) {
    const uint4 v = *reinterpret_cast<const uint4 *>(rowp + g * SSV_GSTRIDE);
    e[g * 4 + 0] = v.x; e[g * 4 + 1] = v.y; e[g * 4 + 2] = v.z; e[g * 4 + 3] = v.w;
  }
  const u32 last = U[Q - 1];
  prev = (u32)__builtin_amdgcn_update_dpp((int)prev, (int)last, 0x111 /*row_shr:1*/, 0xf, 0xf, false);
  const u32 carry = __builtin_amdgcn_alignbit(last, prev, 16);
#pragma unroll
  for (int q = Q - 1; q >= 1; --q) {
    const s16x2 m = __builtin_elementwise_max(as_s16x2(U[q - 1]), as_s16x2(xBv));
    const s16x2 v = __builtin_elementwise_add_sat(m, as_s16x2(e[q]));
    xE = as_u32(__builtin_elementwise_max(as_s16x2(xE), v));
    U[q] = as_u32(v);
  }
  {
    const s16x2 m = __builtin_elementwise_max(as_s16x2(carry), as_s16x2(xBv));
    const s16x2 v = __builtin_elementwise_add_sat(m, as_s16x2(e[0]));
    xE = as_u32(__builtin_elementwise_max(as_s16x2(xE), v));
    U[0] = as_u32(v);
  }
}

__device__ __forceinline__ u32 pkmax_u(u32 a, u32 b) { return as_u32(__builtin_elementwise_max(as_s16x2(a), as_s16x2(b))); }

template <int Q>
__global__ void __launch_bounds__(256) msv_kernel(const SsvBlockWork *__restrict__ work, const DevModel *__restrict__ models,
                                                  const LenEntry *__restrict__ lentab, const uint8_t *__restrict__ res,
                                                  const uint64_t *__restrict__ seq_off, const int32_t *__restrict__ seq_len,
                                                  const uint32_t *__restrict__ lists, int32_t *__restrict__ out_xJ, float *__restrict__ out_usc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int Qg = (Q + 3) / 4;
  constexpr int ROWB = Qg * 256;
  const SsvBlockWork w = work[blockIdx.x];
  const DevModel &md = models[w.model];
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(md.ssv_tbl);
    uint4 *dst = reinterpret_cast<uint4 *>(smem);
    for (int i = threadIdx.x; i < SSV_NROWS * ROWB / 16; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int seg = lane >> 4, z = lane & 15;
  const u32 lane_off = (u32)z * 16u;
  const int base = md.base_b, bias = md.bias_b, tec = md.tec_b;
  for (u32 g = wave; g * 4 < w.count; g += nwaves) {
    const u32 li = g * 4 + seg;
    const bool valid = li < w.count;
    const u32 sid = valid ? lists[w.list_start + li] : 0u;
    const int L = valid ? seq_len[sid] : 0;
    const uint8_t *rp = res + seq_off[sid];
    const int tjb = lentab[L].tjb_b;
    const int tjbm = (tjb + md.tbm_b) & 0xff;
    int Lmax = __builtin_amdgcn_readlane(L, 0);
    Lmax = max(Lmax, __builtin_amdgcn_readlane(L, 16));
    Lmax = max(Lmax, __builtin_amdgcn_readlane(L, 32));
    Lmax = max(Lmax, __builtin_amdgcn_readlane(L, 48));
    u32 U[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) U[q] = U_ZERO;
    u32 prev = U_ZERO;
    int xJ = 0, xB = max(base - tjbm, 0);
    u32 xBv = (0x8000u + (u32)xB) * 0x10001u;
    bool overflow = false;
    const int nchunk = (Lmax + 15) >> 4;
    const uint4 padv = make_uint4(PAD4, PAD4, PAD4, PAD4);
    uint4 cur = padv;
    if (0 < L) cur = *reinterpret_cast<const uint4 *>(rp);
    for (int c = 0; c < nchunk; ++c) {
      uint4 nxt = padv;
      if ((c + 1) * 16 < L) nxt = *reinterpret_cast<const uint4 *>(rp + (c + 1) * 16);
      const u32 wd[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
#pragma unroll
        for (int b4 = 0; b4 < 4; ++b4) {
          u32 xE = U_ZERO;
          msv_row<Q>(U, xE, prev, xBv, smem, lane_off, (wd[k4] >> (8 * b4)) & 0xffu);
          // maximum over the 16 lanes of the sequence, then over the two halves
          xE = pkmax_u(xE, (u32)__builtin_amdgcn_update_dpp((int)xE, (int)xE, 0xB1, 0xf, 0xf, false));
          xE = pkmax_u(xE, (u32)__builtin_amdgcn_update_dpp((int)xE, (int)xE, 0x4E, 0xf, 0xf, false));
          xE = pkmax_u(xE, (u32)__builtin_amdgcn_update_dpp((int)xE, (int)xE, 0x141, 0xf, 0xf, false));
          xE = pkmax_u(xE, (u32)__builtin_amdgcn_update_dpp((int)xE, (int)xE, 0x140, 0xf, 0xf, false));
          xE = pkmax_u(xE, __builtin_amdgcn_alignbit(xE, xE, 16));
          const int xe = (int)(short)(xE & 0xffffu) + 32768;
          overflow = overflow || (xe + bias >= 255);
          const int xe2 = max(xe - tec, 0);
          xJ = max(xJ, xe2);
          xB = max(max(base, xJ) - tjbm, 0);
          xBv = (0x8000u + (u32)xB) * 0x10001u;
        }
      }
      cur = nxt;
    }
    if (valid && z == 0) {
      float sc = (float)(xJ - tjb) - (float)base;
      sc = sc / md.scale_b;
      sc = sc - 3.0f;
      out_xJ[w.pair_start + li] = overflow ? -1 : xJ;
      out_usc[w.pair_start + li] = overflow ? __builtin_inff() : sc;
    }
  }
}

#define CKM_MSV_CASE(QV)                                                                                         \
  case QV:                                                                                                       \
    if ((size_t)SSV_NROWS * ((QV + 3) / 4) * 256 > 48 * 1024) {                                                  \
      static bool attr_set = false;                                                                              \
      if (!attr_set) { (void)hipFuncSetAttribute((const void *)msv_kernel<QV>, hipFuncAttributeMaxDynamicSharedMemorySize, SSV_NROWS * ((QV + 3) / 4) * 256); attr_set = true; } \
    }                                                                                                            \
    hipLaunchKernelGGL(msv_kernel<QV>, dim3(nblocks), dim3(256), (size_t)SSV_NROWS * ((QV + 3) / 4) * 256, stream, \
                       work, models, lentab, res, seq_off, seq_len, lists, out_xJ, out_usc);                     \
    break;

int launch_msv(int Q, int nblocks, hipStream_t stream, const SsvBlockWork *work, const DevModel *models, const LenEntry *lentab,
               const uint8_t *res, const uint64_t *seq_off, const int32_t *seq_len, const uint32_t *lists, int32_t *out_xJ, float *out_usc) {
  if (nblocks <= 0) return 0;
  switch (Q) {
    CKM_MSV_CASE(1) CKM_MSV_CASE(2) CKM_MSV_CASE(3) CKM_MSV_CASE(4) CKM_MSV_CASE(5) CKM_MSV_CASE(6) CKM_MSV_CASE(7)
    CKM_MSV_CASE(8) CKM_MSV_CASE(9) CKM_MSV_CASE(10) CKM_MSV_CASE(11) CKM_MSV_CASE(12) CKM_MSV_CASE(13) CKM_MSV_CASE(14)
    CKM_MSV_CASE(15) CKM_MSV_CASE(16) CKM_MSV_CASE(18) CKM_MSV_CASE(20) CKM_MSV_CASE(22) CKM_MSV_CASE(24) CKM_MSV_CASE(26)
    CKM_MSV_CASE(28) CKM_MSV_CASE(30) CKM_MSV_CASE(32) CKM_MSV_CASE(36) CKM_MSV_CASE(40) CKM_MSV_CASE(48) CKM_MSV_CASE(56)
    CKM_MSV_CASE(64)
    default: return -1;
  }
  return 0;
}

#define CKM_SSV_CASE(QV)                                                                                         \
  case QV:                                                                                                       \
    if ((size_t)SSV_NROWS * ((QV + 3) / 4) * 256 > 48 * 1024) {                                                  \
      static bool attr_set = false;                                                                              \
      if (!attr_set) {                                                                                           \
        (void)hipFuncSetAttribute((const void *)ssv_kernel<QV>, hipFuncAttributeMaxDynamicSharedMemorySize, SSV_NROWS * ((QV + 3) / 4) * 256);   \
        (void)hipFuncSetAttribute((const void *)ssv_kernel_h<QV>, hipFuncAttributeMaxDynamicSharedMemorySize, SSV_NROWS * ((QV + 3) / 4) * 256); \
        attr_set = true;                                                                                         \
      }                                                                                                          \
    }                                                                                                            \
    if (g_ssv_half)                                                                                              \
      hipLaunchKernelGGL(ssv_kernel_h<QV>, dim3(nblocks), dim3(threads), (size_t)SSV_NROWS * ((QV + 3) / 4) * 256, stream, \
                         work, models, res, seq_off, seq_len, lists, maxv);                                      \
    else                                                                                                         \
      hipLaunchKernelGGL(ssv_kernel<QV>, dim3(nblocks), dim3(threads), (size_t)SSV_NROWS * ((QV + 3) / 4) * 256, stream, \
                         work, models, res, seq_off, seq_len, lists, maxv);                                      \
    break;

// CKM_SSV=i16 keeps round 1's packed-integer row (2 ops per register per row); the default is the packed-half row (1.5).
static const bool g_ssv_half = [] { const char *e = getenv("CKM_SSV"); return !(e && strcmp(e, "i16") == 0); }();
bool ssv_half_mode() { return g_ssv_half; }

// Models beyond 2048 nodes have no SSV instance (the LDS image of their emission words would not fit): Smax = 0 for all of their
// pairs sends every one of them to the exact MSV kernel ("no cell rose above xB: recompute exactly" in msv_finish_kernel), which takes
// any model length that fits 160 KB of LDS.  Exact, just slow -- such models are a handful of a full Pfam / TIGRFAM file.
__global__ void ssv_none_kernel(const SsvBlockWork *__restrict__ work, uint16_t *__restrict__ maxv) {
  const SsvBlockWork w = work[blockIdx.x];
  for (uint32_t li = threadIdx.x; li < w.count; li += blockDim.x) maxv[w.pair_start + li] = 0;
}

int launch_ssv(int Q, int nblocks, int threads, hipStream_t stream, const SsvBlockWork *work, const DevModel *models,
               const uint8_t *res, const uint64_t *seq_off, const int32_t *seq_len, const uint32_t *lists, uint16_t *maxv) {
  if (Q >= 100) {                      // launch classes 100 + Q8: eight lanes per sequence
    switch (Q - 100) {
#define X(QV) case QV:                                                                                           \
      if ((size_t)SSV_NROWS * ((QV + 3) / 4) * 256 > 48 * 1024) {                                                \
        static bool attr8 = false;                                                                               \
        if (!attr8) { (void)hipFuncSetAttribute((const void *)ssv_kernel_h8<QV>, hipFuncAttributeMaxDynamicSharedMemorySize, SSV_NROWS * ((QV + 3) / 4) * 256); attr8 = true; } \
      }                                                                                                          \
      hipLaunchKernelGGL(ssv_kernel_h8<QV>, dim3(nblocks), dim3(threads), (size_t)SSV_NROWS * ((QV + 3) / 4) * 256, stream, work, models, res, seq_off, seq_len, lists, maxv); \
      break;
      X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(32)
#undef X
      default: return -1;
    }
    return 0;
  }
  if (Q > 64) { if (nblocks > 0) hipLaunchKernelGGL(ssv_none_kernel, dim3(nblocks), dim3(64), 0, stream, work, maxv); return 0; }
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
