// gene_exec.h -- what gene_pipe.h runs on: the device (HIP streams, device buffers, kernels) in libcheckm_hip.so, or plain loops over host
// memory in the test-only emulation (tests/emu, CKM_GENE_EMU).  A "map" is one thread per index with no cooperation between threads; the
// few kernels that do cooperate (wave ballots, LDS histograms, ordered sums, the dynamic program, scans) are declared here and
// implemented twice: kernels_genes.hip for the device, tests/emu/gene_emu.cpp as scalar loops.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>
#include "gene_dev.h"

#ifdef CKM_GENE_EMU
#include <stdexcept>
namespace ckm {
namespace gene {
struct GExec { int dummy = 0; std::vector<std::vector<uint8_t>> host_blocks; };
// host memory the training rounds exchange with the device every round (page-locked in the library); g_up / g_down move it
inline void g_host_reserve(GExec &, int, size_t) {}
inline void *g_host_raw(GExec &e, int, size_t bytes) { e.host_blocks.emplace_back(bytes + 64); return e.host_blocks.back().data(); }
inline void g_up(GExec &, void *dst, const void *src, size_t n) { if (n) memcpy(dst, src, n); }
inline void g_down(GExec &, void *dst, const void *src, size_t n) { if (n) memcpy(dst, src, n); }
struct GTimer { void begin(GExec &) {} double end(GExec &) { return 0.0; } };
struct GBuf {
  std::vector<uint8_t> v; void *p = nullptr;
  void ensure(size_t bytes) { if (v.size() < bytes + 64) { v.resize(bytes + 64); } p = v.data(); }
  template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};
inline void g_zero(GExec &, void *p, int byte, size_t n) { memset(p, byte, n); }
inline void g_h2d(GExec &, void *dst, const void *src, size_t n) { if (n) memcpy(dst, src, n); }
inline void g_d2h(GExec &, void *dst, const void *src, size_t n) { if (n) memcpy(dst, src, n); }
inline void g_sync(GExec &) {}
template <class F> inline void g_map(GExec &, size_t n, F f) { for (size_t i = 0; i < n; ++i) f(i); }
template <class F> inline void g_map_waves(GExec &, size_t n, F f) { for (size_t i = 0; i < n; ++i) f(i); }
template <class T> inline T *g_host(GExec &e, int which, size_t n) { return reinterpret_cast<T *>(g_host_raw(e, which, n * sizeof(T))); }
template <class T> inline T g_atomic_add(T *p, T v) { const T o = *p; *p = o + v; return o; }
inline void g_count(uint32_t *p) { ++*p; }
inline void g_count_n(uint32_t *p, uint32_t n) { *p += n; }
inline void g_atomic_or(unsigned long long *p, unsigned long long v) { *p |= v; }
[[noreturn]] inline void g_fail(const char *msg) { throw std::runtime_error(msg); }
}  // namespace gene
}  // namespace ckm
#else
#include "ckm_host.h"
namespace ckm {
namespace gene {
// pin[0], pin[1]: page-locked, device-mapped host arenas of the call (the stream's slot keeps them from call to call): what the training
// rounds exchange with the device -- weights up, counts down, thirty times -- moves by a copy KERNEL on the call's stream (g_up / g_down)
// instead of the runtime's copy path: with two dozen calls in flight every hipMemcpyAsync / hipMemsetAsync of a round stood 1.3-1.5 ms in
// the queue of the runtime's own copy kernels (profiles/r06c), ten to twenty-five of them per round.
struct GExec { hipStream_t st = nullptr; PinnedBuf *pin[2] = {nullptr, nullptr}; size_t pin_top[2] = {0, 0}; };
inline void g_host_reserve(GExec &e, int which, size_t bytes) { e.pin[which]->ensure(bytes + 4096); e.pin_top[which] = 0; }      // (before the first g_host_raw of that arena: growing moves it)
inline void *g_host_raw(GExec &e, int which, size_t bytes) {
  const size_t at = (e.pin_top[which] + 255) & ~(size_t)255;
  if (at + bytes > e.pin[which]->cap) throw Error(CKM_ENOMEM, "gene calling: the page-locked exchange area was reserved too small");
  e.pin_top[which] = at + bytes;
  return e.pin[which]->as<uint8_t>() + at;
}
inline void g_up(GExec &e, void *dst, const void *src_pinned, size_t n) { launch_upload(e.st, dst, src_pinned, n); }
inline void g_down(GExec &e, void *dst_pinned, const void *src, size_t n) { launch_upload(e.st, dst_pinned, src, n); }
struct GTimer {          // HIP events on the stream the kernels run on; end() waits for the stream
  hipEvent_t e0 = nullptr, e1 = nullptr;
  void begin(GExec &e) { HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventRecord(e0, e.st)); }
  double end(GExec &e) {
    HIPCHK(hipEventRecord(e1, e.st)); HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(e.st));
    float t = 0.f; HIPCHK(hipEventElapsedTime(&t, e0, e1)); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); e0 = e1 = nullptr;
    return t;
  }
  ~GTimer() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
};
using GBuf = DevBuf;
inline void g_zero(GExec &e, void *p, int byte, size_t n) { if (n) HIPCHK(hipMemsetAsync(p, byte, n, e.st)); }
inline void g_h2d(GExec &e, void *dst, const void *src, size_t n) { if (n) HIPCHK(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, e.st)); }
inline void g_d2h(GExec &e, void *dst, const void *src, size_t n) { if (n) HIPCHK(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, e.st)); }
inline void g_sync(GExec &e) { HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(e.st)); }
template <class F> __global__ void __launch_bounds__(256) g_map_kernel(size_t n, F f) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) f(i);
}
template <class F> inline void g_map(GExec &e, size_t n, F f) {
  if (!n) return;
  if (n > (size_t)0x7fffffff * 256) throw Error(CKM_ERANGE, "gene-calling batch too large for one launch");
  hipLaunchKernelGGL(g_map_kernel<F>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e.st, n, f);
}
// one index per WAVEFRONT (lane 0 works): the ordered walks over a sequence's nodes are chains of dependent loads -- a walk that shares its
// wavefront with 63 others moves in lockstep with the slowest of them
template <class F> __global__ void __launch_bounds__(64) g_map_waves_kernel(size_t n, F f) {
  if (threadIdx.x == 0 && blockIdx.x < n) f((size_t)blockIdx.x);
}
template <class F> inline void g_map_waves(GExec &e, size_t n, F f) {
  if (!n) return;
  if (n > (size_t)0x7fffffff) throw Error(CKM_ERANGE, "gene-calling batch too large for one launch");
  hipLaunchKernelGGL(g_map_waves_kernel<F>, dim3((unsigned)n), dim3(64), 0, e.st, n, f);
}
template <class T> inline T *g_host(GExec &e, int which, size_t n) { return reinterpret_cast<T *>(g_host_raw(e, which, n * sizeof(T))); }
template <class T> __device__ __forceinline__ T g_atomic_add(T *p, T v) { return atomicAdd(p, v); }
// ++*p for counters that most lanes of a wavefront share (the few dozen histogram bins of a bin's start-site training): one atomic per
// distinct address and wavefront, carrying the number of lanes that named it.  The lanes that are active here may be any subset.
__device__ __forceinline__ void g_count(uint32_t *p) {
  const unsigned long long pv = (unsigned long long)p;
  const unsigned me = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  for (;;) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pv), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(pv >> 32));
    const bool mine = (unsigned)pv == lo && (unsigned)(pv >> 32) == hi;
    const unsigned long long same = __ballot(mine);
    if (mine) { if (me == (unsigned)(__ffsll((long long)same) - 1)) atomicAdd(p, (uint32_t)__popcll(same)); break; }
  }
}
// *p += n (n < 32) the same way: the lanes that name one address add their sum with one atomic (five ballots, one per bit of n)
__device__ __forceinline__ void g_count_n(uint32_t *p, uint32_t n) {
  const unsigned long long pv = (unsigned long long)p;
  const unsigned me = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  for (;;) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pv), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(pv >> 32));
    const bool mine = (unsigned)pv == lo && (unsigned)(pv >> 32) == hi;
    const unsigned long long same = __ballot(mine);
    if (mine) {
      uint32_t sum = 0;
#pragma unroll
      for (int b = 0; b < 5; ++b) sum += (uint32_t)__popcll(__ballot((n >> b) & 1u)) << b;
      if (me == (unsigned)(__ffsll((long long)same) - 1) && sum) atomicAdd(p, sum);
      break;
    }
  }
}
__device__ __forceinline__ void g_atomic_or(unsigned long long *p, unsigned long long v) { atomicOr(p, v); }
[[noreturn]] inline void g_fail(const char *msg) { throw Error(CKM_ERANGE, msg); }
}  // namespace gene
}  // namespace ckm
#endif

namespace ckm {
namespace gene {

// ---- cooperating kernels (kernels_genes.hip / tests/emu/gene_emu.cpp) ----
// eight codon-flag bit planes of the ascii text (kernels_orf.hip)
void x_orf_flags(GExec &e, const uint8_t *ascii, unsigned long long *planes, uint64_t body);
// in-place exclusive prefix sum of n + 1 entries (entry n receives the total)
void x_scan_u32(GExec &e, uint32_t *a, size_t n, GBuf &scratch);

// One scan of x_chain: the codons of one reading frame of one strand of a sequence from strand position `top` down to `bottom`.  A whole
// chain starts at the sequence's last codon with the open-end state of add_nodes (after_stop = 0); a bin's training sequence is cut at
// the TTAATTAATTAA separators between its contigs -- every frame of both strands has a stop there, so the scan state behind one is known:
// such a piece starts just below that stop (top = the stop's position, after_stop = 1) and ends with the stop event of the next separator
// (bottom = that stop's position).  Every piece is a chain of its own in the chain array (an open reading frame never spans a stop).
struct SubChain { uint32_t seq, chain; int32_t top, bottom; uint8_t rev, frame, after_stop, pad; };
struct ChainArgs {
  const unsigned long long *planes; uint64_t nwin;                    // codon flags
  const uint64_t *seq_off; const int32_t *seq_len; uint32_t nbins;    // sequences [0, nbins): training, the rest contigs
  const SubChain *sc; uint32_t nsc;
  int tt4;
  const unsigned long long *r50; const uint32_t *pr50;                // starts of 50-runs of unknown bases and their prefix counts (null: no masking)
  unsigned long long *node_planes;                                    // [set][strand][nwin] bit per node position (atomic OR)
  uint32_t *chain_cnt;                                                // [chains] events of the chain
  void *rec; uint32_t *rec_t, *rec_c; unsigned long long *nrec; unsigned long long cap;      // unsorted 16-byte records, their event numbers and chains
};
// start / stop nodes of every chain piece (node.c: add_nodes with -m masks), open ends
void x_chain(GExec &e, const ChainArgs &a);

// t.bias of every bin: the ordered sum over the bin's start nodes (node.c: record_gc_bias), then scaled to a sum of 3
void x_gc_bias(GExec &e, const Nodes &nd, const uint32_t *seq_lo, const uint32_t *seq_n, uint32_t nbins, double *bias /* [nbins][3] */);

// the dynamic program (dprog.c: the forward sweep) for sequences s = 0 .. nseq-1 with nodes [seq_lo[s], seq_lo[s] + seq_n[s]); traceb comes back
// RELATIVE to the sequence's first node; score / traceb / ov_mark must arrive 0 / -1 / -1
void x_dp(GExec &e, const Nodes &nd, const uint32_t *seq_lo, const uint32_t *seq_n, const uint32_t *seq_bin, const double *st_wt, uint32_t nseq, int flag);

// where the trace-back of every sequence begins (dprog.c: the highest-scoring node that may end a path -- not a forward start, not a
// reverse stop -- the LAST of equals in node order); -1: none
void x_path_ends(GExec &e, const Nodes &nd, const uint32_t *seq_lo, const uint32_t *seq_n, uint32_t nseq, int32_t *end_rel);

// hexamer counts of both strands of every bin's training sequence: hist[b][f] = number of positions whose forward hexamer is f
void x_hexamer_background(GExec &e, const uint8_t *code, const uint64_t *seq_off, const int32_t *seq_len, uint32_t nbins, int max_len, uint32_t *hist /* [nbins][4096] */);

// the background word counts of a round of the upstream-motif training (node.c: update_motif_counts over every start node that is not an
// edge).  Stage 0: every word of 3-6 bases at the 13 positions of the upstream window, into tab[slot][length - 3][word] (52 increments
// per start); stages 1 and 2: the start's current motif (nd.mot) and, in stage 1, the shorter words inside it, into
// tab[slot][length - 3][spacer class][word].  On the device a workgroup per part with the counters in LDS.
struct MotifPart { uint32_t lo, hi, slot, pad; };
void x_motif_bg(GExec &e, int stage, const Nodes &nd, const int32_t *seq_len, const MotifPart *parts, uint32_t nparts, uint32_t *tab);

// hexamer sums (bin tables staged in LDS on the device) and Shine-Dalgarno bins of every start node
void x_cscore(GExec &e, const uint8_t *code, const uint64_t *seq_off, const int32_t *seq_len, const Nodes &nd, const double *gene_dc, uint32_t n);
void x_rbs(GExec &e, const uint8_t *code, const uint64_t *seq_off, const int32_t *seq_len, const Nodes &nd, const double *rbs_wt, uint32_t n);

}  // namespace gene
}  // namespace ckm
