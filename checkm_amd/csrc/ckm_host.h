// ckm_host.h -- what the host-side translation units of libcheckm_hip.so share (not part of the ABI): device and pinned buffers,
// the host thread pool, workers and the context, the opaque handle types, and the stage drivers.
//   ckm_api.hip     context, profiles, sequences, hit columns, domtblout writer
//   ckm_stages.hip  drivers of the rare stages: Forward/Backward/OA batches, envelope rescoring, exact MSV, trace ensembles
//   ckm_search.hip  the filter cascade of one worker and ckm_search
//   ckm_debug.hip   diagnostics entries used by the parity tests
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <exception>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <thread>
#include "ckm_internal.h"
#include "dev_types.h"
#include "host_pool.h"

namespace ckm {

// ---- kernel launchers (kernels_*.hip) ------------------------------------------------------------

// ---- kernel launchers (kernels_*.hip) ------------------------------------------------------------
int launch_ssv(int Q, int nblocks, int threads, hipStream_t stream, const SsvBlockWork *work, const DevModel *models,
               const uint8_t *res, const uint64_t *seq_off, const int32_t *seq_len, const uint32_t *lists, const SsvEpi &epi /* where the fused MSV finish appends */);
void launch_msv_full(hipStream_t stream, uint32_t nblocks, WorkQueue queue, const PairRec *pairs, const DevModel *models, const LenEntry *lentab,
                     const uint8_t *res, const uint64_t *seq_off, const int32_t *seq_len, int32_t *out_xJ, float *out_usc, int maxMp,
                     const CascadeDev *cd /* null: scores only */);
void launch_bias(hipStream_t stream, PairRec *pairs, uint32_t npairs, const DevModel *models, const LenEntry *lentab,
                 const uint8_t *res, const uint64_t *seq_off, const int32_t *seq_len, float *raw);
int launch_msv16(int Q, uint32_t nblocks, hipStream_t stream, WorkQueue queue, const PairRec *pairs, const DevModel *models, const LenEntry *lentab,
                 const uint8_t *res, const uint64_t *seq_off, const int32_t *seq_len, const CascadeDev &cd, int32_t *out_xJ = nullptr /* scores only, no decision */, float *out_usc = nullptr);
int launch_vit16(int Q, uint32_t nblocks, hipStream_t stream, WorkQueue queue, const PairRec *pairs, const DevModel *models, const LenEntry *lentab,
                 const uint8_t *res, const uint64_t *seq_off, const int32_t *seq_len, const CascadeDev &cd);
int launch_vit(int QH, uint32_t nblocks, hipStream_t stream, WorkQueue queue, const PairRec *pairs, const DevModel *models,
               const LenEntry *lentab, const uint8_t *res, const uint64_t *seq_off, const int32_t *seq_len, int32_t *out_xC, float *out_sc,
               uint32_t *out_flag, bool fast, const CascadeDev *cd /* null: scores only */);
int launch_fwd(int Q, uint32_t nblocks, hipStream_t stream, WorkQueue queue, FbWork *work, const DevModel *models,
               const LenEntry *lentab, const uint8_t *res, const uint64_t *seq_off, float *ws, FwdOut *out,
               ScaleEvent *events, uint32_t *nevents, uint32_t cap_events, const CascadeDev *cd /* null: no F3 epilogue */);
int launch_bwd(int Q, uint32_t nblocks, hipStream_t stream, WorkQueue queue, const FbWork *work, const DevModel *models,
               const LenEntry *lentab, const uint8_t *res, const uint64_t *seq_off, float *ws, const FwdOut *fout, int32_t *range_err);
int launch_oa(int Q, uint32_t nblocks, hipStream_t stream, WorkQueue queue, const FbWork *work, const DevModel *models,
              float *ws, const int32_t *range_err, const FwdOut *fout, EnvOut *out);

void launch_ensemble(hipStream_t stream, const EnsWork *work, const uint32_t *list /* indices into work, or null */, const uint32_t *count, uint32_t cap, uint32_t grid_regions, int max_Mp,
                     const DevModel *models, const LenEntry *lentab, const uint8_t *res, const uint64_t *seq_off, float *ws, const uint32_t *seeds,
                     float *host_res /* pinned buffer the results are exported to, or null */);
void launch_bias_filter(hipStream_t stream, uint32_t nblocks, const CascadeDev &cd, const DevModel *models, const LenEntry *lentab,
                        const uint8_t *res, const uint64_t *seq_off);
void launch_regions(hipStream_t stream, uint32_t nblocks, const uint32_t *list, const uint32_t *count, uint32_t cap, const FbWork *fwork,
                    const CascadeDev &cd, const DevModel *models, float *ws);
#define HIPCHK(expr)                                                                                         \
  do {                                                                                                       \
    hipError_t e_ = (expr);                                                                                  \
    if (e_ != hipSuccess) throw Error(CKM_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
  } while (0)

// Device blocks given back by a DevBuf are kept for the next DevBuf of about that size instead of going through hipFree / hipMalloc:
// hipFree waits for every kernel running on the device, and a lineage_wf pass over a thousand bins frees (and re-allocates) a few
// thousand sequence, hit and table buffers -- 1.7 s per 1000-bin step once the release is counted.  Per device; blocks of >= 2 GB (the
// float workspace) and anything beyond 64 GB cached are freed as before.  dev_cache_trim() empties it (ckm_ctx_destroy).
struct DevCache {
  std::mutex m;
  std::multimap<size_t, void *> free_[16];
  size_t cached[16] = {0};
  static DevCache &get() { static DevCache c; return c; }
  void *take(int dev, size_t want, size_t &got) {
    std::lock_guard<std::mutex> lock(m);
    auto &f = free_[dev & 15];
    auto it = f.lower_bound(want);
    if (it != f.end() && it->first <= want + want / 2 + (1 << 20)) { void *p = it->second; got = it->first; cached[dev & 15] -= got; f.erase(it); return p; }
    return nullptr;
  }
  bool give(int dev, void *p, size_t cap) {
    if (cap >= ((size_t)2 << 30)) return false;
    std::lock_guard<std::mutex> lock(m);
    if (cached[dev & 15] + cap > ((size_t)64 << 30)) return false;
    free_[dev & 15].emplace(cap, p); cached[dev & 15] += cap;
    return true;
  }
  void trim(int dev) {
    std::multimap<size_t, void *> f;
    { std::lock_guard<std::mutex> lock(m); f.swap(free_[dev & 15]); cached[dev & 15] = 0; }
    for (auto &kv : f) (void)hipFree(kv.second);
  }
};
inline void dev_cache_trim(int dev) { DevCache::get().trim(dev); }

struct DevBuf {
  void *p = nullptr; size_t cap = 0; int dev = -1;
  // vmm (the float workspace, CKM_WS_VMM=1): an address range is reserved once and physical 1 GB chunks are mapped into it as the buffer
  // grows -- measured on MI355X (tools/ubench/vmm_probe.hip, profiles/r03z_vmm_probe.txt): 0.2 ms per GB against hipMalloc's 30 ms per GB,
  // and growth keeps the contents and the address (no free + malloc).  Mapping waits for kernels already running, like hipFree.
  bool vmm = false; size_t va_bytes = (size_t)128 << 30, chunk_bytes = 0; std::vector<hipMemGenericAllocationHandle_t> chunks;
  // A block goes to the cache only from places where its owner's work is known to have drained: the destructor on the normal path (the
  // free entry points and the ends of the calls, all behind their stream synchronisations).  Growth while work may still be queued, and
  // destructors that run while an exception unwinds a call (kernels and copies of the call possibly still in flight), use hipFree, which
  // waits for the device -- the cache hands a block to the next taker at once, with no synchronisation of its own.
  void drop(bool drained = false) {
    if (!p) return;
    if (!drained || dev < 0 || !DevCache::get().give(dev, p, cap)) (void)hipFree(p);
    p = nullptr; cap = 0;
  }
  void ensure(size_t bytes) {
    if (bytes <= cap) return;
    if (vmm) {
      if (p) { grow_mapped(bytes); return; }
      try { grow_mapped(bytes); return; }
      catch (const Error &) { if (p || cap) throw; vmm = false; }      // no address range to be had (nothing is mapped yet): a plain allocation instead
    }
    drop();
    (void)hipGetDevice(&dev);
    size_t want = bytes + std::min<size_t>(bytes / 4, (size_t)1 << 30) + 256;      // (growth slack, bounded: the float workspace is tens of GB)
    size_t got = 0;
    if (void *q = DevCache::get().take(dev, want, got)) { p = q; cap = got; return; }
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      // device memory held by the cache may be what is missing
      DevCache::get().trim(dev);
      e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) { p = nullptr; throw Error(CKM_ENOMEM, "hipMalloc of " + std::to_string(want) + " bytes failed: " + hipGetErrorString(e)); }
    cap = want;
  }
  void grow_mapped(size_t bytes) {
    int dev = 0; (void)hipGetDevice(&dev);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    auto fail = [&](const char *what, hipError_t e) { throw Error(CKM_ENOMEM, std::string(what) + " failed while growing a mapped buffer to " + std::to_string(bytes) + " bytes: " + hipGetErrorString(e)); };
    if (!p) {
      size_t gran = 0;
      hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
      if (e != hipSuccess || !gran) fail("hipMemGetAllocationGranularity", e);
      chunk_bytes = (((size_t)1 << 30) + gran - 1) / gran * gran;
      va_bytes = (va_bytes + chunk_bytes - 1) / chunk_bytes * chunk_bytes;
      e = hipMemAddressReserve(&p, va_bytes, 0, nullptr, 0);
      if (e != hipSuccess) { p = nullptr; fail("hipMemAddressReserve", e); }
    }
    const size_t want = (bytes + 256 + chunk_bytes - 1) / chunk_bytes * chunk_bytes;
    if (want > va_bytes) throw Error(CKM_ENOMEM, "a mapped buffer cannot grow beyond its " + std::to_string(va_bytes) + " reserved bytes");
    hipMemAccessDesc acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = dev; acc.flags = hipMemAccessFlagsProtReadWrite;
    while (cap < want) {
      hipMemGenericAllocationHandle_t h;
      hipError_t e = hipMemCreate(&h, chunk_bytes, &prop, 0);
      if (e != hipSuccess) { DevCache::get().trim(dev); e = hipMemCreate(&h, chunk_bytes, &prop, 0); }      // (device memory held by the block cache may be what is missing)
      if (e != hipSuccess) fail("hipMemCreate", e);
      e = hipMemMap(static_cast<char *>(p) + cap, chunk_bytes, 0, h, 0);
      if (e != hipSuccess) { (void)hipMemRelease(h); fail("hipMemMap", e); }
      e = hipMemSetAccess(static_cast<char *>(p) + cap, chunk_bytes, &acc, 1);
      if (e != hipSuccess) { (void)hipMemUnmap(static_cast<char *>(p) + cap, chunk_bytes); (void)hipMemRelease(h); fail("hipMemSetAccess", e); }
      chunks.push_back(h); cap += chunk_bytes;
    }
  }
  template <class T> T *as() const { return reinterpret_cast<T *>(p); }
  ~DevBuf() {
    if (vmm) {
      if (p && cap) (void)hipMemUnmap(p, cap);
      for (auto h : chunks) (void)hipMemRelease(h);
      if (p) (void)hipMemAddressFree(p, va_bytes);
    } else drop(std::uncaught_exceptions() == 0);
  }
  DevBuf() = default; DevBuf(const DevBuf &) = delete; DevBuf &operator=(const DevBuf &) = delete;
};

struct PinnedBuf {      // grow-only page-locked host staging buffer (pageable D2H copies run at a fraction of PCIe speed)
  void *p = nullptr; size_t cap = 0;
  void ensure(size_t bytes) {
    if (bytes <= cap) return;
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipHostMalloc(&p, want, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { p = nullptr; throw Error(CKM_ENOMEM, "hipHostMalloc failed"); }
    cap = want;
  }
  template <class T> T *as() const { return reinterpret_cast<T *>(p); }
  ~PinnedBuf() { if (p) (void)hipHostFree(p); }
};


// Host -> device through a page-locked staging area and a copy KERNEL on the caller's stream.  The runtime's own copy path
// (hipMemcpyAsync, pageable or pinned source alike) was seen to hold a 3 ms upload for 370 ms -- until another context's SSV launches and
// chains had drained -- about one time in three (profiles/r03t_lane_trace.txt); kernels on the high-priority streams start at once.
void launch_upload(hipStream_t st, void *dst, const void *src_pinned, size_t bytes);
struct Stager {
  // two page-locked halves of 16 MB, filled in turn: while the kernel that reads one half is in flight the host copies the next piece
  // into the other (a staging area as large as a batch -- 150 MB and growing with the batches -- cost seconds of page locking in the
  // first pass of a process).  All puts between begin() and the caller's synchronisation of the stream go to ONE stream.
  static constexpr size_t HALF = (size_t)16 << 20;
  PinnedBuf buf; hipEvent_t ev[2] = {nullptr, nullptr}; bool busy[2] = {false, false}; int half = 0; size_t top = 0;
  void begin(size_t /*total*/) {                                              // (the previous round's stream has been synchronised by its caller)
    if (!buf.p) { buf.ensure(2 * HALF); for (auto &e : ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); }
    busy[0] = busy[1] = false; half = 0; top = 0;
  }
  void put(hipStream_t st, void *dst, const void *src, size_t bytes) {
    const uint8_t *from = static_cast<const uint8_t *>(src); uint8_t *to = static_cast<uint8_t *>(dst);
    while (bytes) {
      if (top == HALF) {                                                        // this half is full: mark its readers, take the other one
        HIPCHK(hipEventRecord(ev[half], st)); busy[half] = true;
        half ^= 1; top = 0;
        if (busy[half]) { HIPCHK(hipEventSynchronize(ev[half])); busy[half] = false; }
      }
      const size_t n = std::min(bytes, HALF - top);
      uint8_t *stage = buf.as<uint8_t>() + (size_t)half * HALF + top;
      memcpy(stage, from, n);
      launch_upload(st, to, stage, n);
      top = std::min(HALF, (top + n + 255) & ~(size_t)255);
      from += n; to += n; bytes -= n;
    }
  }
  ~Stager() { for (auto &e : ev) if (e) (void)hipEventDestroy(e); }
};

}  // namespace ckm

using namespace ckm;      // internal header: every includer is one of the four files above

extern std::atomic<uint64_t> g_uid;        // identity of every profile DB / sequence set ever created (pointers get reused)

// One worker = one host thread's view of the device: its own streams, events and scratch buffers.
// ckm_search splits the models of a call over the workers so that the latency-bound rare stages and the
// host glue of one chunk overlap the VALU-bound SSV / Viterbi kernels of the other.
struct Worker {
  int device = 0;
  int id = 0;
  hipStream_t stream = nullptr;
  hipStream_t ens_stream = nullptr;       // trace ensembles run beside the envelope stage
  hipStream_t side[16];                    // per-register-class launches of the rare stages overlap on these (the first side_streams() are distinct, the rest alias them)
  int nside = 0;                          // distinct side streams owned
  hipEvent_t ev[8];
  ckm_search_stats stats;
  // reusable device scratch
  std::vector<uint64_t> plan_key;         // identifies the SSV block tables currently resident in `work` / `idx`
  std::vector<std::pair<int, std::pair<size_t, size_t>>> plan_groups;
  uint64_t plan_npairs = 0, plan_nblocks = 0, plan_residue_hmm = 0, plan_cells = 0, plan_pairs = 0;
  PinnedBuf h_a, h_b, h_ens;              // D2H staging
  DevBuf work, maxv, surv, nores, counters, cand, raw, idx, vitx, vits, vitf, fbwork, fbidx, fbmodel, ws, fout, events, rerr, envout, fullx, fullu, msvwork, msvlist, enswork, ensseeds, ws_ens, enscount, vitq;
  size_t ws_budget = (size_t)8 << 30;     // float workspace budget (bytes) for Forward/Backward matrices
  std::unique_ptr<HostPool> pool;         // host threads of this worker
  // ---- device-driven cascade (ckm_cascade.hip): tables, queues and result buffers of this lane; capacities only grow ----
  struct CascadeCaps { uint32_t fwork = 0, ework = 0, rwork = 0, pass = 0, reg = 0, events_f = 0, events_e = 0, seen_rwork = 0; uint64_t hens = 0;
                       uint32_t div_cand = 12, div_nores = 48, div_fwork = 160, div_ework = 256, div_rwork = 32768;        // per-group tables hold max(pairs / div, floor) entries
                       uint32_t fl_cand = 4096, fl_nores = 2048, fl_fwork = 2048, fl_ework = 1024, fl_rwork = 256; float ws_per_cell = 11.f; } caps;
  DevBuf c_cnt, c_cand, c_nores, c_bias, c_vfast, c_vexact, c_vflag, c_route, c_vq, c_vxq, c_fq, c_bq, c_eq, c_rq, c_fwork, c_ework, c_rwork, c_ens, c_ensq,
         c_fout_f, c_fout_e, c_fout_r, c_rerr_e, c_rerr_r, c_tops, c_events_r, c_pass, c_reg, c_hens, c_envout, c_events_f, c_events_e;
  PinnedBuf h_cnt, h_pass, h_reg, h_envout, h_events_f, h_events_e, h_hens, h_tops;
  hipEvent_t cev[4] = {nullptr, nullptr, nullptr, nullptr};      // fork / join points of the lane's chain
  hipEvent_t cls_ev[16] = {};                                    // one per side stream
  bool ens_pending = false;
  hipEvent_t ens_ev[17] = {};                                    // chain streams -> the trace-ensemble launch of a sequence part; [16]: that launch -> main stream
  PinnedBuf wstage; std::mutex wstage_mutex;                     // wcopy's staging buffer
  Stager stager;                                                 // staging of the per-search uploads (plan, late rounds)
  hipStream_t late[4] = {};                                      // high-priority streams of the short rounds that follow a search's drain (run_fb with late_round set)
  bool late_round = false;
  hipEvent_t grp_ev[192] = {};                                    // end of the SSV launch of each model-length group
};

// register classes of the Viterbi-filter and Forward/Backward kernels, in queue order (DevModel::vit_cls / fb_cls index these)
constexpr int kVitQH[ckm::NVW] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 24, 32};
constexpr int kVit16Q[ckm::NV16] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16};
constexpr int kFbQ[ckm::NFC] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64};
constexpr int kSsvNone = 65;          // "SSV class" of a model beyond the 2048 nodes the SSV kernel's LDS image holds: every pair goes to the exact MSV kernel
inline int vit_class_id(int QH) { for (int i = 0; i < ckm::NVW; ++i) if (kVitQH[i] == QH) return ckm::NV16 + i; return -1; }
inline int vit16_class_id(int Q16) { for (int i = 0; i < ckm::NV16; ++i) if (kVit16Q[i] == Q16) return i; return -1; }
inline int fb_class_id(int Q) { for (int i = 0; i < ckm::NFC; ++i) if (kFbQ[i] == Q) return i; return -1; }

constexpr int NWORKERS = 8;          // upper bound; CKM_WORKERS (default 3) selects how many a large search uses

struct ckm_ctx {
  int device = 0;
  int nworkers = 1;                // = nclasses * ngroups
  int nclasses = 1, ngroups = 1;   // lanes: length classes (CKM_WORKERS).  One lane is the default: the device-driven
                                   // cascade overlaps the stages of its model-length groups by itself; the host-driven one (CKM_CASCADE=host) gains from 3
  DevBuf reduce_scratch;                  // grow-only device buffer of the reduce kernels
  Worker w[NWORKERS];
  ckm_search_stats stats;
  std::mutex ssv_mutex;                   // SSV phases are VALU-bound: two of them side by side gain nothing
  std::condition_variable ssv_cv; int ssv_turn = 0;    // workers take their first SSV phase in worker order (largest chunk first)
  std::mutex upload_mutex; Stager upload;               // staging of the sequence uploads (finish_seqs)
  std::thread reserve_thread;                          // ckm_ctx_reserve: background allocation of lane 0's float workspace
  std::string reserve_error;
  std::mutex reserve_mutex;
  void settle() { std::lock_guard<std::mutex> lock(reserve_mutex); if (reserve_thread.joinable()) reserve_thread.join(); }      // every entry point that touches the workspace calls this first
  std::atomic<uint64_t> fallbacks{0};                  // lanes the device-driven cascade handed back to the host-driven one (tables / workspace too small)
  hipEvent_t ssv_prev_done = nullptr;                  // device-driven cascade: end of the previous lane's SSV launches (the next lane's wait on it)
};

struct ckm_profiles {
  ckm_ctx *ctx = nullptr;
  uint64_t uid = g_uid++;
  std::vector<HostHMM> hmm;
  std::vector<HostProfile> prof;
  std::vector<DevModel> dm;
  DevBuf d_models;
  std::vector<std::unique_ptr<DevBuf>> tables;
  std::vector<uint8_t> too_long;       // model is longer than the kernels are instantiated for: kept in the database (headers, order), never searched
  int maxMp = 0;
};


struct ckm_seqs {
  ckm_ctx *ctx = nullptr;
  uint64_t uid = 0;
  uint32_t nseq = 0, nbins = 0;
  std::vector<uint32_t> bin_off, seq_bin;
  std::vector<int32_t> len;
  std::vector<uint64_t> off;          // offsets into the padded digital buffer
  std::vector<uint8_t> dsq;           // host copy (null2 needs the residues)
  std::vector<std::string> names, descs;
  std::vector<LenEntry> lentab;
  DevBuf d_res, d_off, d_len, d_lentab;
  uint64_t total_res = 0;
  int maxL = 0;
  // ONE order of all non-empty sequences: grouped by bin, longest first inside a bin.  SSV blocks index ranges of it,
  // so per-bin model subsets (lineage_wf) need no per-model lists.
  std::vector<uint32_t> order, order_off;      // order_off[b] .. order_off[b+1]
  std::vector<uint64_t> bin_res;               // residues of bin b
  DevBuf d_order;
};

struct ckm_hits {
  std::vector<uint64_t> bin_row_off;
  std::vector<uint32_t> seq, model;
  std::vector<int32_t> tlen, qlen, dom_idx, ndom, hmm_from, hmm_to, ali_from, ali_to, env_from, env_to;
  std::vector<double> full_evalue, c_evalue, i_evalue;
  std::vector<float> full_score, full_bias, dom_score, dom_bias, acc;
  uint32_t nbins = 0;
};

// An entry point that fails may have kernels and copies queued (a HIPCHK that throws in the middle of a cascade, with trace ensembles on
// their own stream, chains behind SSV launches ...): nothing of the failed call may still be running when the caller frees or reuses the
// objects it passed in, so the device is drained before the error code goes back.  Argument errors (CKM_EINVAL: thrown before any launch)
// and a missing device skip the wait.
static inline void drain_after_error(int code) {
  if (code != CKM_EINVAL && code != CKM_ENODEV) (void)hipDeviceSynchronize();
}
template <class F>
static inline int guarded(F &&f) {
  try { f(); return CKM_OK; }
  catch (const Error &e) { set_last_error(e.what()); drain_after_error(e.code); return e.code; }
  catch (const std::bad_alloc &) { set_last_error("out of host memory"); drain_after_error(CKM_ENOMEM); return CKM_ENOMEM; }
  catch (const std::exception &e) { set_last_error(e.what()); return CKM_EINVAL; }
}

namespace ckm {

constexpr double kLn2 = 0.69314718055994529;
constexpr double kLog2R = 1.44269504088896341;
constexpr float kOmega = 1.0f / 256.0f;
constexpr float RT1 = 0.25f, RT2 = 0.10f, RT3 = 0.20f;

struct Domain {
  int ienv, jenv; float envsc, oasc, domcorrection; int hmm_from, hmm_to, ali_from, ali_to;
  float dombias, bitscore; double lnP; bool reported;
};
struct Hit {
  uint32_t model, seq; int L; float pre_score, score; double lnP; std::vector<Domain> dom; int nreported;
};
struct Cand {            // a pair that survived the MSV stage
  PairRec r; float fwdsc; float fwd_xC; uint32_t slot; bool alive;
};

struct EventIndex {      // rescale events grouped by slot, rows ascending (flat: one counting sort, no per-slot allocation)
  std::vector<uint32_t> first;                    // [nslots + 1]
  std::vector<std::pair<int, float>> ev;          // (row, scale), slot-major
  void build(const ScaleEvent *e, size_t n, size_t nslots) {
    first.assign(nslots + 1, 0);
    for (size_t k = 0; k < n; ++k) if (e[k].slot < nslots) first[e[k].slot + 1]++;
    for (size_t k = 0; k < nslots; ++k) first[k + 1] += first[k];
    ev.resize(first[nslots]);
    std::vector<uint32_t> at(first.begin(), first.end() - 1);
    for (size_t k = 0; k < n; ++k) if (e[k].slot < nslots) ev[at[e[k].slot]++] = {e[k].row, e[k].scale};
    for (size_t k = 0; k < nslots; ++k) if (first[k + 1] - first[k] > 1) std::sort(ev.begin() + first[k], ev.begin() + first[k + 1]);
  }
  void build(const std::vector<ScaleEvent> &v, size_t nslots) { build(v.data(), v.size(), nslots); }
  std::vector<float> scales(uint32_t slot) const { std::vector<float> r; for (uint32_t k = first[slot]; k < first[slot + 1]; ++k) r.push_back(ev[k].second); return r; }
};

// Runs fwd/bwd(/oa) for a list of work items, grouped by the model's canonical Q.
struct FbBatch {
  std::vector<FbWork> work;
  std::vector<FwdOut> fout;
  std::vector<ScaleEvent> events;
  std::vector<int32_t> rerr;
  std::vector<EnvOut> envout;
};

struct EnvReq { uint32_t model, seq; int ienv, jenv; };

struct EnvRes { bool ok; float envsc, oasc, xC; int nscale; float null2[KP]; int hmm_from, hmm_to, ali_from, ali_to; };

struct Seg { int32_t sqfrom, sqto, hmmfrom, hmmto; };

struct RegionReq { uint32_t model, seq; int ireg, jreg; };

struct RegionRes {
  std::vector<float> n2sum;        // per region position: sum over traces of the null2 odds ratio
  std::vector<Seg> segs;           // [200][cap], first domain first
  std::vector<int32_t> nseg;       // [200]
  int cap = 0;
  std::vector<Seg> env;            // clustered envelopes, region-local coordinates, sorted by start
};

// ---- stage drivers (ckm_stages.hip) ----------------------------------------------------------------------------------
void pool_run(Worker *w, size_t n, size_t chunk, const std::function<void(size_t, size_t)> &f);
void wcopy(Worker *w, void *dst, const void *src, size_t bytes, hipMemcpyKind kind);     // blocking copy on the worker's own stream
double now_ms();
float bits(float sc, float nullsc);
float finish_forward(float xC, float move, const std::vector<float> &scales);
int ssv_threads_for(int Q);
int ssv_class(const HostProfile &hp);
uint32_t ssv_per_block(int cls);
int side_streams();
int choose_side_streams(int nworkers);
void trace_pt(const Worker *w, const char *label);      // CKM_TRACE=1: worker / ms since the search began / label on stderr
void trace_begin();
void run_fb(Worker *ctx, const ckm_profiles *p, const ckm_seqs *s, FbBatch &b, bool do_fwd, bool do_bwd, bool do_oa,
            const std::vector<uint32_t> *subset /* indices into b.work, or null = all */, float *ws_other = nullptr);
void rescore_envelopes(Worker *ctx, const ckm_profiles *p, const ckm_seqs *s, const std::vector<EnvReq> &req, std::vector<EnvRes> &out,
                       std::vector<std::vector<int32_t>> *paths = nullptr, std::vector<std::vector<float>> *pps = nullptr);
void run_msv_exact(Worker *ctx, const ckm_profiles *p, const ckm_seqs *s, const std::vector<PairRec> &pairs, std::vector<float> &usc, std::vector<int32_t> *xJ);
void run_ensembles(Worker *ctx, const ckm_profiles *p, const ckm_seqs *s, const std::vector<RegionReq> &req, std::vector<RegionRes> &out);
void fill_null2(float *null2);
uint32_t fb_grid(size_t n);
uint32_t ens_seed(int t);
void cluster_ensemble(RegionRes &r);
constexpr bool kEnvInplaceDefault = true;
bool env_inplace();
void ensure_ens_seeds(Worker *ctx);

// what the domain stage hands to the row assembly (both cascades fill it)
struct PassInfo { uint32_t model, seq; float fwdsc; };
struct DomItem { uint32_t pass; int i, j, region; };       // regions in sequence order; region >= 0: resolved by the trace ensemble (index into regres)
struct DomStage {
  std::vector<PassInfo> pass;
  std::vector<int> nregions;
  std::vector<DomItem> items;
  std::vector<RegionRes> regres;
  std::vector<EnvReq> envreq; std::vector<int> env_region; std::vector<EnvRes> envres;
  std::vector<std::pair<size_t, size_t>> env_of_pass;       // [first, count) into envreq
};

}  // namespace ckm
