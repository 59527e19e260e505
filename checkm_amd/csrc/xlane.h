// xlane.h -- cross-lane primitives of the float kernels, gfx950 only.  Everything here is a VALU data-parallel-primitive
// (DPP) move, a v_permlane*_swap or a v_readlane: no LDS round trip (ds_bpermute, what __shfl compiles to) sits on the
// row-to-row dependency chain of the Forward/Backward recurrences.
//
// The ORDER of the float operations below is part of the canonical evaluation order:
//   wave_sum      xor butterfly with partners 1, 2, 4, 8, 16, 32 in THAT order (ascending)
//   scan_up/down  affine maps (a, b): 16-lane rows are scanned with offsets 1, 2, 4, 8 (Kogge-Stone inside a row), the
//                 three row totals are composed one after the other, and every lane applies the prefix of its row
#pragma once
#include <hip/hip_runtime.h>

namespace ckm {

constexpr int DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;
constexpr int DPP_ROW_SHL = 0x100, DPP_ROW_SHR = 0x110, DPP_WAVE_SHL1 = 0x130, DPP_WAVE_SHR1 = 0x138;

// A table pointer that was read from a struct in memory (DevModel::rf, ::vit_e, ...) is a generic pointer to the compiler, and a load
// through it is a FLAT load: it counts against lgkmcnt as well as vmcnt, so every wait for an LDS read also waits for the emission row
// requested a moment ago -- the whole global latency lands on the row-to-row chain.  gptr() types the pointer as global, and loads
// through the result (keep it in an `auto` / gp<T> variable) are global_load (vmcnt only).
template <class T> using gp = const __attribute__((address_space(1))) T *;
template <class T>
__device__ __forceinline__ gp<T> gptr(const T *p) { return (gp<T>)p; }

// The residues of a segment, 64 at a time in ONE register of the wavefront (lane l holds residue base + l) and handed out through
// v_readlane: the row loop of a latency-bound kernel then has no load whose address depends on another load of the same row (residue
// byte -> table row), and the table row's address is an SGPR.  ResUp serves ascending positions, ResDown descending ones; a position
// may be asked for repeatedly but never further back (forward) than the current chunk.
struct ResUp {
  const uint8_t *rp; int n, base, lane, cur, nxt;
  __device__ __forceinline__ void init(const uint8_t *p, int n_, int lane_) {
    rp = p; n = n_; lane = lane_; base = 0;
    cur = p[min(lane, n - 1)]; nxt = p[min(64 + lane, n - 1)];
  }
  __device__ __forceinline__ int get(int r) {
    if (r >= base + 64) { base += 64; cur = nxt; nxt = rp[min(base + 64 + lane, n - 1)]; }
    return __builtin_amdgcn_readlane(cur, r - base);
  }
};
struct ResDown {
  const uint8_t *rp; int n, base, lane, cur, prv;
  __device__ __forceinline__ void init(const uint8_t *p, int n_, int lane_) {
    rp = p; n = n_; lane = lane_; base = ((n - 1) >> 6) << 6;
    cur = p[min(base + lane, n - 1)]; prv = p[max(base - 64, 0) + lane];        // (a first chunk is always 64 readable bytes: sequences are padded)
  }
  __device__ __forceinline__ int get(int r) {
    if (r < base) { base -= 64; cur = prv; prv = rp[max(base - 64, 0) + lane]; }
    return __builtin_amdgcn_readlane(cur, r - base);
  }
};

// lane <- dpp-selected lane of src; lanes whose source is out of range keep `old`
template <int CTRL>
__device__ __forceinline__ float dpp_f(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xf, 0xf, false); }

// lane z <- lane z-1 (lane 0 <- fill) / lane z <- lane z+1 (lane 63 <- fill)
__device__ __forceinline__ float lane_up1(float v, float fill) { return dpp_f<DPP_WAVE_SHR1>(fill, v); }
__device__ __forceinline__ float lane_down1(float v, float fill) { return dpp_f<DPP_WAVE_SHL1>(fill, v); }

// partner exchange across 16-lane rows (xor 16) and across half-waves (xor 32): a <-> b halves swap, a + b is the pair sum
__device__ __forceinline__ void swap16(float &a, float &b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap32(float &a, float &b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap16(int &a, int &b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap32(int &a, int &b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b)); }

// sum over the 64 lanes, every lane gets it.  After the step with partner w all lanes of a 2w-group hold the same value, so
// the mirror moves (partner 7-z, 15-z) pair exactly the operands the xor-4 / xor-8 butterfly pairs.
__device__ __forceinline__ float wave_sum(float s) {
  s = s + dpp_f<DPP_QUAD_XOR1>(s, s);
  s = s + dpp_f<DPP_QUAD_XOR2>(s, s);
  s = s + dpp_f<DPP_ROW_HALF_MIRROR>(s, s);
  s = s + dpp_f<DPP_ROW_MIRROR>(s, s);
  { float a = s, b = s; swap16(a, b); s = a + b; }
  { float a = s, b = s; swap32(a, b); s = a + b; }
  return s;
}
__device__ __forceinline__ float wave_max(float s) {
  s = fmaxf(s, dpp_f<DPP_QUAD_XOR1>(s, s));
  s = fmaxf(s, dpp_f<DPP_QUAD_XOR2>(s, s));
  s = fmaxf(s, dpp_f<DPP_ROW_HALF_MIRROR>(s, s));
  s = fmaxf(s, dpp_f<DPP_ROW_MIRROR>(s, s));
  { float a = s, b = s; swap16(a, b); s = fmaxf(a, b); }
  { float a = s, b = s; swap32(a, b); s = fmaxf(a, b); }
  return s;
}
__device__ __forceinline__ int wave_max(int s) {
  s = max(s, dpp_i<DPP_QUAD_XOR1>(s, s));
  s = max(s, dpp_i<DPP_QUAD_XOR2>(s, s));
  s = max(s, dpp_i<DPP_ROW_HALF_MIRROR>(s, s));
  s = max(s, dpp_i<DPP_ROW_MIRROR>(s, s));
  { int a = s, b = s; swap16(a, b); s = max(a, b); }
  { int a = s, b = s; swap32(a, b); s = max(a, b); }
  return s;
}
__device__ __forceinline__ int wave_min(int s) {
  s = min(s, dpp_i<DPP_QUAD_XOR1>(s, s));
  s = min(s, dpp_i<DPP_QUAD_XOR2>(s, s));
  s = min(s, dpp_i<DPP_ROW_HALF_MIRROR>(s, s));
  s = min(s, dpp_i<DPP_ROW_MIRROR>(s, s));
  { int a = s, b = s; swap16(a, b); s = min(a, b); }
  { int a = s, b = s; swap32(a, b); s = min(a, b); }
  return s;
}

__device__ __forceinline__ float read_lane(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

// (a, b) <- (a, b) after (oa, ob):  d -> a*(oa*d + ob) + b
#define CKM_COMPOSE(a, b, oa, ob) { const float t_ = (a) * (ob); (b) = (b) + t_; (a) = (a) * (oa); }

// Inclusive scan of affine maps towards higher lanes: on return lane z holds the composition of the maps of lanes 0..z
// (lane z's own map applied last).
__device__ __forceinline__ void scan_up(float &a, float &b, int lane) {
#define STEP(S) { const float oa = dpp_f<DPP_ROW_SHR + S>(1.0f, a), ob = dpp_f<DPP_ROW_SHR + S>(0.0f, b); CKM_COMPOSE(a, b, oa, ob) }
  STEP(1) STEP(2) STEP(4) STEP(8)
#undef STEP
  const float a0 = read_lane(a, 15), b0 = read_lane(b, 15), a1 = read_lane(a, 31), b1 = read_lane(b, 31), a2 = read_lane(a, 47), b2 = read_lane(b, 47);
  float pa2 = a1, pb2 = b1; CKM_COMPOSE(pa2, pb2, a0, b0)           // rows 0..1
  float pa3 = a2, pb3 = b2; CKM_COMPOSE(pa3, pb3, pa2, pb2)         // rows 0..2
  const int row = lane >> 4;
  const float pa = row == 0 ? 1.0f : row == 1 ? a0 : row == 2 ? pa2 : pa3;
  const float pb = row == 0 ? 0.0f : row == 1 ? b0 : row == 2 ? pb2 : pb3;
  CKM_COMPOSE(a, b, pa, pb)
}
// Mirror image: lane z holds the composition of the maps of lanes z..63 (lane z's own map applied last).
__device__ __forceinline__ void scan_down(float &a, float &b, int lane) {
#define STEP(S) { const float oa = dpp_f<DPP_ROW_SHL + S>(1.0f, a), ob = dpp_f<DPP_ROW_SHL + S>(0.0f, b); CKM_COMPOSE(a, b, oa, ob) }
  STEP(1) STEP(2) STEP(4) STEP(8)
#undef STEP
  const float a3 = read_lane(a, 48), b3 = read_lane(b, 48), a2 = read_lane(a, 32), b2 = read_lane(b, 32), a1 = read_lane(a, 16), b1 = read_lane(b, 16);
  float pa1 = a2, pb1 = b2; CKM_COMPOSE(pa1, pb1, a3, b3)           // rows 2..3
  float pa0 = a1, pb0 = b1; CKM_COMPOSE(pa0, pb0, pa1, pb1)         // rows 1..3
  const int row = lane >> 4;
  const float pa = row == 3 ? 1.0f : row == 2 ? a3 : row == 1 ? pa1 : pa0;
  const float pb = row == 3 ? 0.0f : row == 2 ? b3 : row == 1 ? pb1 : pb0;
  CKM_COMPOSE(a, b, pa, pb)
}

}  // namespace ckm
