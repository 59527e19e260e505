// ckm_api.hip -- C ABI of libcheckm_hip.so: context, profile upload, sequence packing and the
// search orchestration (host glue between the gfx950 kernels).  See include/checkm_hip.h for the
// reference interfaces each entry point replaces.
//
// The host side owns every transcendental (log/exp): per-length specials, null scores, E-values.
// The device side owns every per-cell operation.  There is no CPU implementation of any kernel in
// this library: if HIP is unusable, ckm_ctx_create fails and nothing else can run.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <atomic>
#include <thread>
#include <exception>
#include <numeric>
#include <array>
#include <condition_variable>
#include <functional>
#include "ckm_internal.h"
#include "dev_types.h"

namespace ckm {

// ---- kernel launchers (kernels_*.hip) ------------------------------------------------------------
int launch_ssv(int Q, int nblocks, int threads, hipStream_t stream, const SsvBlockWork *work, const DevModel *models,
               const uint8_t *res, const uint64_t *seq_off, const int32_t *seq_len, const uint32_t *lists, uint16_t *maxv);
void launch_msv_finish(hipStream_t stream, const FinishArgs &a, uint32_t nblocks);
int launch_msv(int Q, int nblocks, hipStream_t stream, const SsvBlockWork *work, const DevModel *models, const LenEntry *lentab,
               const uint8_t *res, const uint64_t *seq_off, const int32_t *seq_len, const uint32_t *lists, int32_t *out_xJ, float *out_usc);
void launch_msv_full(hipStream_t stream, const PairRec *pairs, uint32_t npairs, const DevModel *models, const LenEntry *lentab,
                     const uint8_t *res, const uint64_t *seq_off, const int32_t *seq_len, int32_t *out_xJ, float *out_usc, int maxMp);
void launch_bias(hipStream_t stream, PairRec *pairs, uint32_t npairs, const DevModel *models, const LenEntry *lentab,
                 const uint8_t *res, const uint64_t *seq_off, const int32_t *seq_len, float *raw);
int launch_vit(int Q, hipStream_t stream, const PairRec *pairs, const uint32_t *idx, uint32_t n, const DevModel *models,
               const LenEntry *lentab, const uint8_t *res, const uint64_t *seq_off, const int32_t *seq_len, int32_t *out_xC, float *out_sc, uint32_t *out_flag, bool fast);
int launch_fwd(int Q, uint32_t n, hipStream_t stream, const FbWork *work, const uint32_t *idx, const uint32_t *blk_model, const DevModel *models,
               const LenEntry *lentab, const uint8_t *res, const uint64_t *seq_off, float *ws, FwdOut *out,
               ScaleEvent *events, uint32_t *nevents, uint32_t cap_events);
int launch_bwd(int Q, uint32_t n, hipStream_t stream, const FbWork *work, const uint32_t *idx, const uint32_t *blk_model, const DevModel *models,
               const LenEntry *lentab, const uint8_t *res, const uint64_t *seq_off, float *ws, const FwdOut *fout, int32_t *range_err);
int launch_oa(int Q, uint32_t n, hipStream_t stream, const FbWork *work, const uint32_t *idx, const uint32_t *blk_model, const DevModel *models,
              float *ws, const int32_t *range_err, EnvOut *out);

void launch_ensemble(hipStream_t stream, const EnsWork *work, uint32_t nregions, int max_Ld, int max_Mp, const DevModel *models,
                     const LenEntry *lentab, const uint8_t *res, const uint64_t *seq_off, float *ws, const uint32_t *seeds);
static thread_local std::string g_err;
void set_last_error(const std::string &m) { g_err = m; }

#define HIPCHK(expr)                                                                                         \
  do {                                                                                                       \
    hipError_t e_ = (expr);                                                                                  \
    if (e_ != hipSuccess) throw Error(CKM_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
  } while (0)

struct DevBuf {
  void *p = nullptr; size_t cap = 0;
  void ensure(size_t bytes) {
    if (bytes <= cap) return;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { p = nullptr; throw Error(CKM_ENOMEM, "hipMalloc of " + std::to_string(want) + " bytes failed: " + hipGetErrorString(e)); }
    cap = want;
  }
  template <class T> T *as() const { return reinterpret_cast<T *>(p); }
  ~DevBuf() { if (p) (void)hipFree(p); }
};

struct PinnedBuf {      // grow-only page-locked host staging buffer (pageable D2H copies run at a fraction of PCIe speed)
  void *p = nullptr; size_t cap = 0;
  void ensure(size_t bytes) {
    if (bytes <= cap) return;
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; throw Error(CKM_ENOMEM, "hipHostMalloc failed"); }
    cap = want;
  }
  template <class T> T *as() const { return reinterpret_cast<T *>(p); }
  ~PinnedBuf() { if (p) (void)hipHostFree(p); }
};


// A few host threads for the per-pair / per-sequence glue between the kernel stages (logs of rescale factors, region
// scans over the decoding terms, segment clustering, bit scores): the device idles while that glue runs.
class HostPool {
 public:
  explicit HostPool(int nthreads) {
    for (int i = 1; i < nthreads; ++i) th_.emplace_back([this] { loop(); });
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> g(m_); stop_ = true; }
    cv_.notify_all();
    for (auto &t : th_) t.join();
  }
  // f(lo, hi) over [0, n) in chunks; the caller works too; returns when every chunk is done and no thread is still inside
  void run(size_t n, size_t chunk, const std::function<void(size_t, size_t)> &f) {
    if (!n) return;
    if (th_.empty() || n <= chunk) { f(0, n); return; }
    {
      std::lock_guard<std::mutex> g(m_);
      job_ = &f; n_ = n; chunk_ = chunk; next_.store(0); err_ = nullptr; open_ = true; ++active_;
    }
    cv_.notify_all();
    work(f, n, chunk);
    std::unique_lock<std::mutex> g(m_);
    open_ = false;                                    // late wakers must not join a job whose chunks are all handed out
    --active_;
    done_.wait(g, [this] { return active_ == 0; });   // every helper has left work(): the fields may change again
    job_ = nullptr;
    if (err_) std::rethrow_exception(err_);
  }
 private:
  // the job's description travels by value: a helper never reads fields the next run() may be rewriting
  void work(const std::function<void(size_t, size_t)> &f, size_t n, size_t chunk) {
    for (;;) {
      const size_t lo = next_.fetch_add(chunk);
      if (lo >= n) return;
      try { f(lo, std::min(n, lo + chunk)); } catch (...) { std::lock_guard<std::mutex> g(m_); if (!err_) err_ = std::current_exception(); }
    }
  }
  void loop() {
    for (;;) {
      const std::function<void(size_t, size_t)> *f; size_t n, chunk;
      {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return stop_ || (open_ && job_ && next_.load() < n_); });
        if (stop_) return;
        f = job_; n = n_; chunk = chunk_; ++active_;
      }
      work(*f, n, chunk);
      std::lock_guard<std::mutex> g(m_);
      if (--active_ == 0) done_.notify_all();
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_; std::condition_variable cv_, done_;
  const std::function<void(size_t, size_t)> *job_ = nullptr;
  size_t n_ = 0, chunk_ = 1; std::atomic<size_t> next_{0};
  int active_ = 0; bool open_ = false, stop_ = false; std::exception_ptr err_;
};

}  // namespace ckm

using namespace ckm;

// One worker = one host thread's view of the device: its own streams, events and scratch buffers.
// ckm_search splits the models of a call over the workers so that the latency-bound rare stages and the
// host glue of one chunk overlap the VALU-bound SSV / Viterbi kernels of the other.
struct Worker {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t ens_stream = nullptr;       // trace ensembles run beside the envelope stage
  hipStream_t side[8];                    // per-register-class launches of the rare stages overlap on these
  hipEvent_t ev[8];
  ckm_search_stats stats;
  // reusable device scratch
  std::vector<uint64_t> plan_key;         // identifies the SSV block tables currently resident in `work` / `idx`
  std::vector<std::pair<int, std::pair<size_t, size_t>>> plan_groups;
  uint64_t plan_npairs = 0, plan_nblocks = 0, plan_residue_hmm = 0, plan_cells = 0, plan_pairs = 0;
  PinnedBuf h_a, h_b, h_ens;              // D2H staging
  DevBuf work, maxv, surv, nores, counters, cand, raw, idx, vitx, vits, vitf, fbwork, fbidx, fbmodel, ws, fout, events, rerr, envout, fullx, fullu, msvwork, msvlist, enswork, ensseeds, ws_ens;
  size_t ws_budget = (size_t)8 << 30;     // float workspace budget (bytes) for Forward/Backward matrices
  std::unique_ptr<HostPool> pool;         // host threads of this worker
};

constexpr int NWORKERS = 4;          // upper bound; CKM_WORKERS (default 3) selects how many a large search uses

struct ckm_ctx {
  int device = 0;
  int nworkers = 3;
  DevBuf reduce_scratch;                  // grow-only device buffer of the reduce kernels
  Worker w[NWORKERS];
  ckm_search_stats stats;
  std::mutex ssv_mutex;                   // SSV phases are VALU-bound: two of them side by side gain nothing
  std::condition_variable ssv_cv; int ssv_turn = 0;    // workers take their first SSV phase in worker order (largest chunk first)
};

static std::atomic<uint64_t> g_uid{1};     // identity of every profile DB / sequence set / list ever created (pointers get reused)

struct ckm_profiles {
  ckm_ctx *ctx = nullptr;
  uint64_t uid = g_uid++;
  std::vector<HostHMM> hmm;
  std::vector<HostProfile> prof;
  std::vector<DevModel> dm;
  DevBuf d_models;
  std::vector<std::unique_ptr<DevBuf>> tables;
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

template <class F>
static int guarded(F &&f) {
  try { f(); return CKM_OK; }
  catch (const Error &e) { set_last_error(e.what()); return e.code; }
  catch (const std::bad_alloc &) { set_last_error("out of host memory"); return CKM_ENOMEM; }
  catch (const std::exception &e) { set_last_error(e.what()); return CKM_EINVAL; }
}

// accessors for ckm_reduce.hip
const std::string &ckm_seq_name(const ckm_seqs *s, uint32_t i) { return s->names.at(i); }
int ckm_ctx_device(const ckm_ctx *ctx) { return ctx->device; }
void ckm_ctx_parallel_for(ckm_ctx *ctx, size_t n, size_t chunk, const std::function<void(size_t, size_t)> &f) {
  if (ctx->w[0].pool) ctx->w[0].pool->run(n, chunk, f); else if (n) f(0, n);
}
void *ckm_ctx_reduce_scratch(ckm_ctx *ctx, size_t bytes) { ctx->reduce_scratch.ensure(bytes); return ctx->reduce_scratch.p; }

extern "C" const char *ckm_last_error(void) { return g_err.c_str(); }
extern "C" int ckm_abi_version(void) { return CKM_ABI_VERSION; }

extern "C" int ckm_device_count(int *n) {
  return guarded([&] {
    if (!n) throw Error(CKM_EINVAL, "n is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; throw Error(CKM_ENODEV, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *n = c;
  });
}

extern "C" int ckm_ctx_create(int device, ckm_ctx **out) {
  // The per-register-class launches of the rare stages overlap on up to 8 streams; the runtime's default of 4 hardware
  // queues would serialise half of them.  Only effective if HIP has not been initialised in this process yet.
  setenv("GPU_MAX_HW_QUEUES", "16", 0);
  return guarded([&] {
    if (!out) throw Error(CKM_EINVAL, "out is NULL");
    *out = nullptr;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c <= 0) throw Error(CKM_ENODEV, "no HIP device visible: libcheckm_hip has no CPU path");
    if (device < 0 || device >= c) throw Error(CKM_ENODEV, "device index out of range");
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).compare(0, 6, "gfx950") != 0)
      throw Error(CKM_ENODEV, std::string("device is ") + prop.gcnArchName + "; this library carries gfx950 code objects only");
    std::unique_ptr<ckm_ctx> ctx(new ckm_ctx());
    ctx->device = device;
    memset(&ctx->stats, 0, sizeof(ctx->stats));
    if (const char *e = getenv("CKM_WORKERS")) ctx->nworkers = std::max(1, std::min(NWORKERS, atoi(e)));
    int host_threads = std::max(1, std::min(8, (int)std::thread::hardware_concurrency()));
    if (const char *e = getenv("CKM_HOST_THREADS")) host_threads = std::max(1, std::min(64, atoi(e)));
    size_t fre = 0, tot = 0;
    size_t budget = (size_t)8 << 30;
    if (hipMemGetInfo(&fre, &tot) == hipSuccess) budget = std::min<size_t>((size_t)96 << 30, fre / 2) / ctx->nworkers;
    if (const char *e = getenv("CKM_WS_BUDGET_MB")) budget = std::max<size_t>(16, strtoull(e, nullptr, 10)) << 20;   // tests: force several envelope batches
    for (auto &w : ctx->w) {
      w.device = device;
      // non-blocking streams: nothing here may synchronise implicitly with the null stream or with another worker's streams
      HIPCHK(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
      HIPCHK(hipStreamCreateWithFlags(&w.ens_stream, hipStreamNonBlocking));
      for (auto &st : w.side) HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      for (auto &e : w.ev) HIPCHK(hipEventCreate(&e));
      memset(&w.stats, 0, sizeof(w.stats));
      w.ws_budget = budget;
      if (&w - ctx->w < ctx->nworkers) w.pool.reset(new HostPool(host_threads));
    }
    *out = ctx.release();
  });
}

extern "C" void ckm_ctx_destroy(ckm_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  for (auto &w : ctx->w) {
    for (auto &e : w.ev) (void)hipEventDestroy(e);
    for (auto &st : w.side) (void)hipStreamDestroy(st);
    (void)hipStreamDestroy(w.stream);
    (void)hipStreamDestroy(w.ens_stream);
  }
  delete ctx;
}

// ---- profiles -----------------------------------------------------------------------------------
template <class T>
static const T *upload(ckm_profiles *p, const std::vector<T> &v) {
  std::unique_ptr<DevBuf> b(new DevBuf());
  b->ensure(std::max<size_t>(16, v.size() * sizeof(T)));
  if (!v.empty()) HIPCHK(hipMemcpy(b->p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  const T *r = b->as<T>();
  p->tables.push_back(std::move(b));
  return r;
}

extern "C" int ckm_profiles_load(ckm_ctx *ctx, const char *hmm_path, ckm_profiles **out) {
  return guarded([&] {
    if (!ctx || !hmm_path || !out) throw Error(CKM_EINVAL, "NULL argument");
    *out = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    std::unique_ptr<ckm_profiles> p(new ckm_profiles());
    p->ctx = ctx;
    p->hmm = read_hmm_file(hmm_path);
    for (const auto &h : p->hmm) {
      p->prof.push_back(configure_profile(h));
      const HostProfile &hp = p->prof.back();
      DevModel d;
      memset(&d, 0, sizeof(d));
      d.M = hp.M; d.ssvQ = hp.ssvQ; d.fbQ = hp.fbQ; d.vitQH = hp.vitQH;
      d.base_b = hp.base_b; d.bias_b = hp.bias_b; d.tbm_b = hp.tbm_b; d.tec_b = hp.tec_b; d.scale_b = hp.scale_b;
      d.scale_w = hp.scale_w; d.base_w = hp.base_w; d.wE_loop = hp.wE_loop; d.wE_move = hp.wE_move;
      d.fE_loop = hp.fE_loop; d.fE_move = hp.fE_move;
      d.bt00 = hp.bt00; d.bt01 = hp.bt01; d.bt10 = hp.bt10; d.bt11 = hp.bt11; d.bpi0 = hp.bpi0; d.bpi1 = hp.bpi1;
      for (int x = 0; x < NROWS; ++x) d.beo1[x] = hp.beo1[x];
      d.thr_msv_f1 = hp.thr_msv_f1; d.thr_msv_f2 = hp.thr_msv_f2; d.thr_vit_f2 = hp.thr_vit_f2; d.thr_fwd_f3 = hp.thr_fwd_f3;
      d.ssv_tbl = upload(p.get(), hp.ssv_tbl); d.rbv = upload(p.get(), hp.rbv); d.vit_e = upload(p.get(), hp.vit_e);
      d.vit_t = upload(p.get(), hp.vit_t); d.rf = upload(p.get(), hp.rf); d.ftr = upload(p.get(), hp.ftr);
      p->dm.push_back(d);
      p->maxMp = std::max(p->maxMp, hp.fbQ * NL);
    }
    p->d_models.ensure(p->dm.size() * sizeof(DevModel));
    HIPCHK(hipMemcpy(p->d_models.p, p->dm.data(), p->dm.size() * sizeof(DevModel), hipMemcpyHostToDevice));
    *out = p.release();
  });
}

extern "C" int ckm_profiles_count(const ckm_profiles *p, int32_t *n) {
  if (!p || !n) { set_last_error("NULL argument"); return CKM_EINVAL; }
  *n = (int32_t)p->hmm.size();
  return CKM_OK;
}

extern "C" int ckm_profiles_header(const ckm_profiles *p, int32_t i, ckm_model_header *o) {
  if (!p || !o || i < 0 || i >= (int32_t)p->hmm.size()) { set_last_error("bad argument"); return CKM_EINVAL; }
  const HostHMM &h = p->hmm[i];
  o->name = h.name.c_str(); o->acc = h.has_acc ? h.acc.c_str() : nullptr; o->desc = h.has_desc ? h.desc.c_str() : nullptr;
  o->leng = h.M; o->has_ga = h.has_ga; o->has_tc = h.has_tc; o->has_nc = h.has_nc;
  for (int k = 0; k < 2; ++k) { o->ga[k] = h.ga[k]; o->tc[k] = h.tc[k]; o->nc[k] = h.nc[k]; }
  for (int k = 0; k < 6; ++k) o->evparam[k] = h.evparam[k];
  return CKM_OK;
}

extern "C" void ckm_profiles_free(ckm_profiles *p) {
  if (!p) return;
  if (p->ctx) (void)hipSetDevice(p->ctx->device);
  delete p;
}

// ---- sequences ----------------------------------------------------------------------------------
static void build_lentab(ckm_seqs *s) {
  HostProfile dummy;    // scale_b / scale_w are model independent constants
  dummy.scale_b = (float)(3.0 / 0.69314718055994529);
  dummy.scale_w = (float)(500.0 / 0.69314718055994529);
  s->lentab.resize((size_t)s->maxL + 1);
  for (int L = 0; L <= s->maxL; ++L) {
    const LenCfg m = len_config(dummy, L, true), u = len_config(dummy, L, false);
    LenEntry e;
    e.loop_m = m.loop; e.move_m = m.move; e.loop_u = u.loop; e.move_u = u.move;
    e.nullsc = m.nullsc; e.bias_tail = m.bias_tail; e.w_move = m.w_move; e.tjb_b = m.tjb_b;
    s->lentab[L] = e;
  }
}

// shared tail of the two constructors: s->len / s->off / s->dsq / names are filled; build the order, tables and upload
static void finish_seqs(ckm_seqs *s) {
  const uint32_t nbins = s->nbins, nseq = s->nseq;
  s->seq_bin.resize(nseq);
  for (uint32_t b = 0; b < nbins; ++b) for (uint32_t i = s->bin_off[b]; i < s->bin_off[b + 1]; ++i) s->seq_bin[i] = b;
  s->order_off.assign(nbins + 1, 0); s->bin_res.assign(nbins, 0);
  for (uint32_t b = 0; b < nbins; ++b) {
    s->order_off[b] = (uint32_t)s->order.size();
    const size_t first = s->order.size();
    for (uint32_t i = s->bin_off[b]; i < s->bin_off[b + 1]; ++i) if (s->len[i] > 0) { s->order.push_back(i); s->bin_res[b] += (uint64_t)s->len[i]; }
    std::stable_sort(s->order.begin() + first, s->order.end(), [&](uint32_t x, uint32_t y) { return s->len[x] > s->len[y]; });
  }
  s->order_off[nbins] = (uint32_t)s->order.size();
  build_lentab(s);
  s->d_res.ensure(s->dsq.size()); HIPCHK(hipMemcpy(s->d_res.p, s->dsq.data(), s->dsq.size(), hipMemcpyHostToDevice));
  s->d_off.ensure(std::max<size_t>(8, (size_t)nseq * 8)); if (nseq) HIPCHK(hipMemcpy(s->d_off.p, s->off.data(), (size_t)nseq * 8, hipMemcpyHostToDevice));
  s->d_len.ensure(std::max<size_t>(4, (size_t)nseq * 4)); if (nseq) HIPCHK(hipMemcpy(s->d_len.p, s->len.data(), (size_t)nseq * 4, hipMemcpyHostToDevice));
  s->d_order.ensure(std::max<size_t>(4, s->order.size() * 4));
  if (!s->order.empty()) HIPCHK(hipMemcpy(s->d_order.p, s->order.data(), s->order.size() * 4, hipMemcpyHostToDevice));
  s->d_lentab.ensure(s->lentab.size() * sizeof(LenEntry));
  HIPCHK(hipMemcpy(s->d_lentab.p, s->lentab.data(), s->lentab.size() * sizeof(LenEntry), hipMemcpyHostToDevice));
}

extern "C" int ckm_seqs_pack(ckm_ctx *ctx, const char *text, const uint64_t *seq_off, uint32_t nseq,
                             const uint32_t *bin_off, uint32_t nbins, const char *const *names,
                             const char *const *descs, ckm_seqs **out) {
  return guarded([&] {
    if (!ctx || !text || !seq_off || !bin_off || !out) throw Error(CKM_EINVAL, "NULL argument");
    *out = nullptr;
    if (nbins == 0 || bin_off[0] != 0 || bin_off[nbins] != nseq) throw Error(CKM_EINVAL, "bin_off must start at 0 and end at nseq");
    HIPCHK(hipSetDevice(ctx->device));
    std::unique_ptr<ckm_seqs> s(new ckm_seqs());
    s->ctx = ctx; s->nseq = nseq; s->nbins = nbins; s->uid = g_uid++;
    s->bin_off.assign(bin_off, bin_off + nbins + 1);
    for (uint32_t b = 0; b < nbins; ++b) if (bin_off[b + 1] < bin_off[b]) throw Error(CKM_EINVAL, "bin_off not monotone");
    s->len.resize(nseq); s->off.resize(nseq);
    uint64_t pos = 0;
    for (uint32_t i = 0; i < nseq; ++i) {
      if (seq_off[i + 1] < seq_off[i]) throw Error(CKM_EINVAL, "seq_off not monotone");
      const uint64_t L = seq_off[i + 1] - seq_off[i];
      if (L > 100000) throw Error(CKM_ERANGE, "sequence longer than 100000 residues");
      s->len[i] = (int32_t)L; s->off[i] = pos;
      pos += (L + 15) & ~(uint64_t)15;
      s->total_res += L; s->maxL = std::max(s->maxL, (int)L);
    }
    s->dsq.assign(pos + 16, (uint8_t)PADCODE);
    for (uint32_t i = 0; i < nseq; ++i) digitize(text + seq_off[i], (uint64_t)s->len[i], s->dsq.data() + s->off[i]);
    s->names.resize(nseq); s->descs.resize(nseq);
    for (uint32_t i = 0; i < nseq; ++i) {
      if (names && names[i]) s->names[i] = names[i]; else s->names[i] = "seq" + std::to_string(i);
      if (descs && descs[i]) s->descs[i] = descs[i];
    }
    finish_seqs(s.get());
    *out = s.release();
  });
}

extern "C" int ckm_seqs_from_fasta(ckm_ctx *ctx, const char *const *paths, uint32_t nbins, ckm_seqs **out) {
  return guarded([&] {
    if (!ctx || !paths || !out) throw Error(CKM_EINVAL, "NULL argument");
    *out = nullptr;
    if (nbins == 0) throw Error(CKM_EINVAL, "no bins");
    HIPCHK(hipSetDevice(ctx->device));
    std::unique_ptr<ckm_seqs> s(new ckm_seqs());
    s->ctx = ctx; s->nbins = nbins; s->uid = g_uid++;
    s->bin_off.assign(nbins + 1, 0);
    std::vector<char> buf;
    uint64_t pos = 0;
    auto close_seq = [&](uint64_t start) {            // pad the record that just ended to a 16-byte boundary
      const uint64_t L = pos - start;
      if (L > 100000) throw Error(CKM_ERANGE, "sequence longer than 100000 residues");
      s->len.push_back((int32_t)L); s->total_res += L; s->maxL = std::max(s->maxL, (int)L);
      const uint64_t padded = (L + 15) & ~(uint64_t)15;
      s->dsq.resize(start + padded, (uint8_t)PADCODE);
      pos = start + padded;
    };
    for (uint32_t b = 0; b < nbins; ++b) {
      FILE *f = fopen(paths[b], "rb");
      if (!f) throw Error(CKM_EIO, std::string("cannot open FASTA file ") + paths[b]);
      fseek(f, 0, SEEK_END); const long sz = ftell(f); fseek(f, 0, SEEK_SET);
      buf.resize((size_t)std::max<long>(sz, 0));
      const size_t got = sz > 0 ? fread(buf.data(), 1, (size_t)sz, f) : 0;
      fclose(f);
      if ((long)got != sz) throw Error(CKM_EIO, std::string("short read on ") + paths[b]);
      size_t i = 0; bool open = false; uint64_t start = 0;
      while (i < got) {
        size_t e = i; while (e < got && buf[e] != '\n') ++e;
        size_t le = e; if (le > i && buf[le - 1] == '\r') --le;
        if (le > i && buf[i] == '>') {
          if (open) close_seq(start);
          size_t n0 = i + 1, n1 = n0; while (n1 < le && !isspace((unsigned char)buf[n1])) ++n1;
          size_t d0 = n1; while (d0 < le && isspace((unsigned char)buf[d0])) ++d0;
          s->names.emplace_back(buf.data() + n0, n1 - n0);
          s->descs.emplace_back(buf.data() + d0, le - d0);
          start = pos; s->off.push_back(start); open = true;
        } else if (open && le > i) {
          size_t a = i, z = le;                           // strip blanks at both ends of the line
          while (a < z && isspace((unsigned char)buf[a])) ++a;
          while (z > a && isspace((unsigned char)buf[z - 1])) --z;
          if (z > a) { s->dsq.resize(pos + (z - a)); digitize(buf.data() + a, z - a, s->dsq.data() + pos); pos += z - a; }
        }
        i = e + 1;
      }
      if (open) close_seq(start);
      s->bin_off[b + 1] = (uint32_t)s->names.size();
    }
    s->nseq = (uint32_t)s->names.size();
    s->dsq.resize(pos + 16, (uint8_t)PADCODE);
    finish_seqs(s.get());
    *out = s.release();
  });
}

extern "C" int ckm_seqs_count(const ckm_seqs *s, uint32_t *nseq, uint32_t *nbins) {
  if (!s || !nseq || !nbins) { set_last_error("NULL argument"); return CKM_EINVAL; }
  *nseq = s->nseq; *nbins = s->nbins;
  return CKM_OK;
}
extern "C" int ckm_seqs_bin_offsets(const ckm_seqs *s, const uint32_t **bin_off) {
  if (!s || !bin_off) { set_last_error("NULL argument"); return CKM_EINVAL; }
  *bin_off = s->bin_off.data();
  return CKM_OK;
}
extern "C" int ckm_seqs_name(const ckm_seqs *s, uint32_t i, const char **name, const char **desc, int32_t *len) {
  if (!s || i >= s->nseq) { set_last_error("bad argument"); return CKM_EINVAL; }
  if (name) *name = s->names[i].c_str();
  if (desc) *desc = s->descs[i].c_str();
  if (len) *len = s->len[i];
  return CKM_OK;
}

extern "C" int ckm_seqs_residues(const ckm_seqs *s, uint64_t *total) {
  if (!s || !total) { set_last_error("NULL argument"); return CKM_EINVAL; }
  *total = s->total_res;
  return CKM_OK;
}

extern "C" void ckm_seqs_free(ckm_seqs *s) {
  if (!s) return;
  if (s->ctx) (void)hipSetDevice(s->ctx->device);
  delete s;
}

// ---- the search -----------------------------------------------------------------------------------
namespace {

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

void pool_run(Worker *w, size_t n, size_t chunk, const std::function<void(size_t, size_t)> &f) { if (w->pool) w->pool->run(n, chunk, f); else if (n) f(0, n); }

// blocking copy on the worker's own stream (a plain hipMemcpy would wait for every blocking stream of the device)
void wcopy(Worker *w, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
  HIPCHK(hipMemcpyAsync(dst, src, bytes, kind, w->stream));
  HIPCHK(hipStreamSynchronize(w->stream));
}

static const bool g_trace = getenv("CKM_TRACE") != nullptr;     // per-worker stage timestamps on stderr
static double g_trace_t0 = 0;
#define CKM_TRACE_PT(label) do { if (g_trace) fprintf(stderr, "ckm-trace w%d %8.3f %s\n", my_turn, now_ms_() - g_trace_t0, label); } while (0)
double now_ms_();

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

double now_ms_() { return now_ms(); }

float bits(float sc, float nullsc) { return (float)((double)(sc - nullsc) / kLn2); }

// host-side completion of a Forward score from the device's scaled xC and its rescale events
float finish_forward(float xC, float move, const std::vector<float> &scales) {
  float totscale = 0.f;
  for (float sc : scales) totscale = (float)((double)totscale + log((double)sc));
  return (float)((double)totscale + log((double)(xC * move)));
}

struct EventIndex {      // rescale events grouped by slot, rows ascending
  std::vector<std::vector<std::pair<int, float>>> by_slot;
  void build(const std::vector<ScaleEvent> &ev, size_t nslots) {
    by_slot.assign(nslots, {});
    for (const auto &e : ev) if (e.slot < nslots) by_slot[e.slot].push_back({e.row, e.scale});
    for (auto &v : by_slot) std::sort(v.begin(), v.end());
  }
  std::vector<float> scales(uint32_t slot) const { std::vector<float> r; for (auto &p : by_slot[slot]) r.push_back(p.second); return r; }
};

int ssv_threads_for(int Q) {
  const size_t lds = (size_t)NROWS * ((Q + 3) / 4) * 256;
  if (lds <= 40 * 1024) return 256;
  if (lds <= 80 * 1024 || Q > 40) return 512;   // kernels with Q > 40 are compiled for <= 512 threads (256 VGPRs)
  return 1024;
}

// Runs fwd/bwd(/oa) for a list of work items, grouped by the model's canonical Q.
struct FbBatch {
  std::vector<FbWork> work;
  std::vector<FwdOut> fout;
  std::vector<ScaleEvent> events;
  std::vector<int32_t> rerr;
  std::vector<EnvOut> envout;
};

void run_fb(Worker *ctx, const ckm_profiles *p, const ckm_seqs *s, FbBatch &b, bool do_fwd, bool do_bwd, bool do_oa,
            const std::vector<uint32_t> *subset /* indices into b.work, or null = all */, float *ws_other = nullptr) {
  const size_t n = b.work.size();
  if (!n) return;
  ctx->fbwork.ensure(n * sizeof(FbWork));
  wcopy(ctx, ctx->fbwork.p, b.work.data(), n * sizeof(FbWork), hipMemcpyHostToDevice);
  // group by canonical Q; inside a group, blocks of 4 wavefronts take 4 items of ONE model (shared LDS table)
  std::map<int, std::map<uint32_t, std::vector<uint32_t>>> byQ;
  auto add = [&](uint32_t i) { byQ[p->prof[b.work[i].model].fbQ][b.work[i].model].push_back(i); };
  if (subset) for (uint32_t i : *subset) add(i); else for (uint32_t i = 0; i < n; ++i) add(i);
  struct Group { int Q; size_t blk0, nblk; };
  std::vector<uint32_t> items, blk_model; std::vector<Group> groups;
  for (auto it = byQ.rbegin(); it != byQ.rend(); ++it) {      // heaviest register class first: its chain is the longest
    auto &kq = *it;
    Group g{kq.first, blk_model.size(), 0};
    // longest items first inside a model so the four wavefronts of a block finish together
    for (auto &km : kq.second) {
      std::vector<uint32_t> &v = km.second;
      std::stable_sort(v.begin(), v.end(), [&](uint32_t x, uint32_t y) { return b.work[x].Ld > b.work[y].Ld; });
      for (size_t i = 0; i < v.size(); i += 4) {
        for (size_t j = 0; j < 4; ++j) items.push_back(i + j < v.size() ? v[i + j] : 0xffffffffu);
        blk_model.push_back(km.first);
      }
    }
    g.nblk = blk_model.size() - g.blk0;
    groups.push_back(g);
  }
  ctx->fbidx.ensure(items.size() * 4); ctx->fbmodel.ensure(blk_model.size() * 4);
  wcopy(ctx, ctx->fbidx.p, items.data(), items.size() * 4, hipMemcpyHostToDevice);
  wcopy(ctx, ctx->fbmodel.p, blk_model.data(), blk_model.size() * 4, hipMemcpyHostToDevice);
  ctx->fout.ensure(n * sizeof(FwdOut));
  ctx->rerr.ensure(n * 4);
  ctx->envout.ensure(n * sizeof(EnvOut));
  const uint32_t cap_events = (uint32_t)std::max<size_t>(1 << 20, n * 64);
  ctx->events.ensure((size_t)cap_events * sizeof(ScaleEvent));
  ctx->counters.ensure(64);
  const DevModel *dm = p->d_models.as<DevModel>();
  const LenEntry *lt = s->d_lentab.as<LenEntry>();
  const uint8_t *res = s->d_res.as<uint8_t>();
  const uint64_t *off = s->d_off.as<uint64_t>();
  float *ws = ws_other ? ws_other : ctx->ws.as<float>();
  if (do_fwd) HIPCHK(hipMemsetAsync(ctx->counters.p, 0, 64, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  // every register class runs its stages in order on its own stream; classes overlap each other
  size_t gi = 0;
  for (auto &g : groups) {
    static const int nfb = [] { const char *e = getenv("CKM_FB_STREAMS"); return e ? std::max(1, std::min(8, atoi(e))) : 8; }();
    hipStream_t st = ctx->side[gi++ % nfb];
    const uint32_t *ix = ctx->fbidx.as<uint32_t>() + g.blk0 * 4, *bm = ctx->fbmodel.as<uint32_t>() + g.blk0;
    if (do_fwd && launch_fwd(g.Q, (uint32_t)g.nblk, st, ctx->fbwork.as<FbWork>(), ix, bm, dm, lt, res, off, ws, ctx->fout.as<FwdOut>(),
                             ctx->events.as<ScaleEvent>(), ctx->counters.as<uint32_t>(), cap_events))
      throw Error(CKM_ERANGE, "no Forward kernel instance for this model length");
    if (do_bwd && launch_bwd(g.Q, (uint32_t)g.nblk, st, ctx->fbwork.as<FbWork>(), ix, bm, dm, lt, res, off, ws, ctx->fout.as<FwdOut>(), ctx->rerr.as<int32_t>()))
      throw Error(CKM_ERANGE, "no Backward kernel instance for this model length");
    if (do_oa && launch_oa(g.Q, (uint32_t)g.nblk, st, ctx->fbwork.as<FbWork>(), ix, bm, dm, ws, ctx->rerr.as<int32_t>(), ctx->envout.as<EnvOut>()))
      throw Error(CKM_ERANGE, "no OA kernel instance for this model length");
  }
  HIPCHK(hipGetLastError());
  for (auto &st : ctx->side) HIPCHK(hipStreamSynchronize(st));
  if (do_fwd) {
    b.fout.resize(n);
    uint32_t nev = 0;
    wcopy(ctx, b.fout.data(), ctx->fout.p, n * sizeof(FwdOut), hipMemcpyDeviceToHost);
    wcopy(ctx, &nev, ctx->counters.p, 4, hipMemcpyDeviceToHost);
    if (nev > cap_events) throw Error(CKM_ERANGE, "rescale event buffer overflow");
    b.events.resize(nev);
    if (nev) wcopy(ctx, b.events.data(), ctx->events.p, (size_t)nev * sizeof(ScaleEvent), hipMemcpyDeviceToHost);
  }
  if (do_oa) {
    b.envout.resize(n);
    wcopy(ctx, b.envout.data(), ctx->envout.p, n * sizeof(EnvOut), hipMemcpyDeviceToHost);
  }
}

size_t env_floats(int Mp, int Ld, uint64_t &xs, uint64_t &aux, uint64_t &mf, uint64_t &mb, uint64_t base) {
  auto al = [](uint64_t v) { return (v + 31) & ~(uint64_t)31; };
  uint64_t pos = al(base);
  xs = pos; pos = al(pos + (uint64_t)(Ld + 1) * 6);
  aux = pos; pos = al(al(pos + (uint64_t)(Ld + 1) * 3) + (uint64_t)(Ld + 1) * 5);
  mf = pos; pos = al(pos + (uint64_t)(Ld + 1) * 3 * Mp);
  mb = pos; pos = al(pos + (uint64_t)(Ld + 1) * 2 * Mp);     // posterior rows: M and I only
  return pos;
}

// Rescore envelopes on the device; returns one Domain per envelope (ok flag via envsc NaN on range error)
struct EnvReq { uint32_t model, seq; int ienv, jenv; };
struct EnvRes { bool ok; float envsc, oasc, xC; int nscale; float null2[KP]; int hmm_from, hmm_to, ali_from, ali_to; };

void rescore_envelopes(Worker *ctx, const ckm_profiles *p, const ckm_seqs *s, const std::vector<EnvReq> &req, std::vector<EnvRes> &out) {
  out.resize(req.size());
  size_t done = 0;
  const uint64_t budget_floats = ctx->ws_budget / 4;
  while (done < req.size()) {
    FbBatch b; uint64_t pos = 0; size_t j = done;
    for (; j < req.size(); ++j) {
      const EnvReq &r = req[j];
      const int Mp = p->prof[r.model].fbQ * NL, Ld = r.jenv - r.ienv + 1;
      FbWork w; memset(&w, 0, sizeof(w));
      uint64_t end = env_floats(Mp, Ld, w.xs_off, w.aux_off, w.mxf_off, w.mxb_off, pos);
      if (end > budget_floats && j > done) break;
      if (end > budget_floats) throw Error(CKM_ENOMEM, "one envelope needs more workspace than the device budget allows");
      w.model = r.model; w.seq = r.seq; w.i0 = r.ienv - 1; w.Ld = Ld; w.Lcfg = s->len[r.seq]; w.multihit = 0; w.slot = (uint32_t)(j - done); w.full = 1;
      b.work.push_back(w); pos = end;
    }
    ctx->ws.ensure(pos * 4 + 256);
    run_fb(ctx, p, s, b, true, true, true, nullptr);
    EventIndex ei; ei.build(b.events, b.work.size());
    for (size_t k = 0; k < b.work.size(); ++k) {
      const EnvReq &r = req[done + k]; EnvRes &o = out[done + k];
      const EnvOut &eo = b.envout[k];
      const LenEntry &le = s->lentab[s->len[r.seq]];
      o.ok = eo.range_err == 0;
      o.xC = b.fout[k].xC; o.nscale = b.fout[k].nscale;
      o.envsc = finish_forward(b.fout[k].xC, le.move_u, ei.scales((uint32_t)k));
      o.oasc = eo.oasc; o.hmm_from = eo.hmm_from; o.hmm_to = eo.hmm_to; o.ali_from = eo.ali_from; o.ali_to = eo.ali_to;
      for (int x = 0; x < K; ++x) o.null2[x] = eo.null2[x];
    }
    done = j;
  }
}



// Exact multi-hit MSV of an arbitrary list of pairs with the packed (SSV-style) kernel: pairs are grouped by model (one LDS
// emission image per workgroup), longest sequences first, 16 sequences per workgroup (4 wavefronts x 4).  Results land in
// usc/xJ in the order of `pairs`.
void run_msv_exact(Worker *ctx, const ckm_profiles *p, const ckm_seqs *s, const std::vector<PairRec> &pairs, std::vector<float> &usc, std::vector<int32_t> *xJ) {
  const size_t n = pairs.size();
  usc.assign(n, 0.f); if (xJ) xJ->assign(n, 0);
  if (!n) return;
  std::map<int, std::map<uint32_t, std::vector<uint32_t>>> byQ;            // Q -> model -> indices into pairs
  for (uint32_t i = 0; i < n; ++i) byQ[p->prof[pairs[i].model].ssvQ][pairs[i].model].push_back(i);
  std::vector<SsvBlockWork> work; std::vector<uint32_t> lists, slot_of(n); std::vector<std::pair<int, std::pair<size_t, size_t>>> groups;
  constexpr uint32_t PER_BLOCK = 16;
  for (auto it = byQ.rbegin(); it != byQ.rend(); ++it) {
    const size_t first = work.size();
    std::vector<SsvBlockWork> blocks;
    for (auto &km : it->second) {
      std::vector<uint32_t> &v = km.second;
      std::stable_sort(v.begin(), v.end(), [&](uint32_t a, uint32_t b) { return s->len[pairs[a].seq] > s->len[pairs[b].seq]; });
      for (size_t a = 0; a < v.size(); a += PER_BLOCK) {
        SsvBlockWork w; w.model = km.first; w.list_start = (uint32_t)lists.size(); w.count = (uint32_t)std::min<size_t>(PER_BLOCK, v.size() - a); w.pair_start = w.list_start;
        for (uint32_t k = 0; k < w.count; ++k) { slot_of[v[a + k]] = (uint32_t)lists.size(); lists.push_back(pairs[v[a + k]].seq); }
        blocks.push_back(w);
      }
    }
    // longest workgroups first inside a launch
    std::stable_sort(blocks.begin(), blocks.end(), [&](const SsvBlockWork &x, const SsvBlockWork &y) { return s->len[lists[x.list_start]] > s->len[lists[y.list_start]]; });
    work.insert(work.end(), blocks.begin(), blocks.end());
    groups.push_back({it->first, {first, work.size() - first}});
  }
  ctx->msvwork.ensure(work.size() * sizeof(SsvBlockWork)); ctx->msvlist.ensure(lists.size() * 4);
  ctx->fullx.ensure(n * 4); ctx->fullu.ensure(n * 4);
  HIPCHK(hipMemcpyAsync(ctx->msvwork.p, work.data(), work.size() * sizeof(SsvBlockWork), hipMemcpyHostToDevice, ctx->stream));
  wcopy(ctx, ctx->msvlist.p, lists.data(), lists.size() * 4, hipMemcpyHostToDevice);
  int gi = 0;
  for (auto &g : groups) {
    if (launch_msv(g.first, (int)g.second.second, ctx->side[gi++ % 8], ctx->msvwork.as<SsvBlockWork>() + g.second.first, p->d_models.as<DevModel>(), s->d_lentab.as<LenEntry>(),
                   s->d_res.as<uint8_t>(), s->d_off.as<uint64_t>(), s->d_len.as<int32_t>(), ctx->msvlist.as<uint32_t>(), ctx->fullx.as<int32_t>(), ctx->fullu.as<float>()))
      throw Error(CKM_ERANGE, "no MSV kernel instance for this model length");
  }
  HIPCHK(hipGetLastError());
  for (auto &st : ctx->side) HIPCHK(hipStreamSynchronize(st));
  std::vector<float> raw(n); std::vector<int32_t> rawx(n);
  HIPCHK(hipMemcpyAsync(raw.data(), ctx->fullu.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (xJ) HIPCHK(hipMemcpyAsync(rawx.data(), ctx->fullx.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (size_t i = 0; i < n; ++i) { usc[i] = raw[slot_of[i]]; if (xJ) (*xJ)[i] = rawx[slot_of[i]]; }
}

// ---- multi-domain regions: trace ensemble on the device, clustering of the sampled segments here ----------------
struct Seg { int32_t sqfrom, sqto, hmmfrom, hmmto; };
struct RegionReq { uint32_t model, seq; int ireg, jreg; };
struct RegionRes {
  std::vector<float> n2sum;        // per region position: sum over traces of the null2 odds ratio
  std::vector<Seg> segs;           // [200][cap], first domain first
  std::vector<int32_t> nseg;       // [200]
  int cap = 0;
  std::vector<Seg> env;            // clustered envelopes, region-local coordinates, sorted by start
};

constexpr uint32_t kEnsStride = 15485863u;
constexpr float kEnsMinOverlap = 0.8f, kEnsMinPosterior = 0.25f, kEnsMinEndpointP = 0.02f;
constexpr int kEnsMaxDiagDiff = 4;

uint32_t ens_mix3(uint32_t a, uint32_t b, uint32_t c) {
  a -= b; a -= c; a ^= (c >> 13);  b -= c; b -= a; b ^= (a << 8);   c -= a; c -= b; c ^= (b >> 13);
  a -= b; a -= c; a ^= (c >> 12);  b -= c; b -= a; b ^= (a << 16);  c -= a; c -= b; c ^= (b >> 5);
  a -= b; a -= c; a ^= (c >> 3);   b -= c; b -= a; b ^= (a << 10);  c -= a; c -= b; c ^= (b >> 15);
  return c;
}
// generator state of trace t: Easel's fast generator x -> 69069x+1, seeded 42 (mixed as esl_randomness_Init does), advanced t*stride steps
uint32_t ens_seed(int t) {
  uint32_t x = ens_mix3(42u, 87654321u, 12345678u); if (x == 0) x = 42u;
  uint32_t A = 69069u, C = 1u, ra = 1u, rc = 0u;
  for (uint64_t n = (uint64_t)t * kEnsStride; n; n >>= 1) { if (n & 1) { ra = A * ra; rc = A * rc + C; } C = A * C + C; A = A * A; }
  return ra * x + rc;
}

bool seg_linked(const Seg &a, const Seg &b) {
  int nov = std::min(a.sqto, b.sqto) - std::max(a.sqfrom, b.sqfrom) + 1;
  int n = std::min(a.sqto - a.sqfrom + 1, b.sqto - b.sqfrom + 1);
  if ((float)nov / (float)n < kEnsMinOverlap) return false;
  nov = std::min(a.hmmto, b.hmmto) - std::max(a.hmmfrom, b.hmmfrom) + 1;
  n = std::min(a.hmmto - a.hmmfrom + 1, b.hmmto - b.hmmfrom + 1);
  if ((float)nov / (float)n < kEnsMinOverlap) return false;
  const int d1 = (a.sqfrom - a.hmmfrom + a.sqto - a.hmmto) / 2, d2 = (b.sqfrom - b.hmmfrom + b.sqto - b.hmmto) / 2;
  return std::abs(d1 - d2) <= kEnsMaxDiagDiff;
}

// single linkage over all sampled segments; clusters seen in >= 25% of the traces become envelopes whose ends are the
// outermost endpoints sampled in >= 2% of those traces.  Most of the 200 traces sample the same few segments, so the
// linkage runs over the DISTINCT segments (numbered in order of first appearance, which keeps the cluster order).
void cluster_ensemble(RegionRes &r) {
  struct Uniq { Seg g; int count; std::vector<uint8_t> in_trace; };
  std::vector<Uniq> u;
  std::map<std::array<int32_t, 4>, int> index;
  for (int t = 0; t < ENS_NSAMPLES; ++t) for (int d = 0; d < r.nseg[t]; ++d) {
    const Seg &g = r.segs[(size_t)t * r.cap + d];
    auto ins = index.insert({{g.sqfrom, g.sqto, g.hmmfrom, g.hmmto}, (int)u.size()});
    if (ins.second) u.push_back({g, 0, std::vector<uint8_t>(ENS_NSAMPLES, 0)});
    Uniq &x = u[ins.first->second]; x.count++; x.in_trace[t] = 1;
  }
  const int n = (int)u.size();
  std::vector<int> asg(n, -1), stack;
  int nc = 0;
  for (int h = 0; h < n; ++h) if (asg[h] < 0) {
    stack.assign(1, h); asg[h] = nc;
    while (!stack.empty()) { const int a = stack.back(); stack.pop_back(); for (int b = 0; b < n; ++b) if (asg[b] < 0 && seg_linked(u[a].g, u[b].g)) { asg[b] = nc; stack.push_back(b); } }
    ++nc;
  }
  for (int c = 0; c < nc; ++c) {
    int ninc = 0;
    for (int t = 0; t < ENS_NSAMPLES; ++t) { bool any = false; for (int h = 0; h < n && !any; ++h) any = asg[h] == c && u[h].in_trace[t]; ninc += any; }
    if ((float)ninc / (float)ENS_NSAMPLES < kEnsMinPosterior) continue;
    int best[4];
    for (int f = 0; f < 4; ++f) {
      auto val = [&](int h) { return f == 0 ? u[h].g.sqfrom : f == 1 ? u[h].g.sqto : f == 2 ? u[h].g.hmmfrom : u[h].g.hmmto; };
      int lo = 1 << 30, hi = -1;
      for (int h = 0; h < n; ++h) if (asg[h] == c) { lo = std::min(lo, val(h)); hi = std::max(hi, val(h)); }
      std::vector<int> epc(hi - lo + 1, 0);
      for (int h = 0; h < n; ++h) if (asg[h] == c) epc[val(h) - lo] += u[h].count;
      int b;
      if (f == 0 || f == 2) { for (b = lo; b < hi; ++b) if ((float)epc[b - lo] / (float)ninc >= kEnsMinEndpointP) break; }
      else                  { for (b = hi; b > lo; --b) if ((float)epc[b - lo] / (float)ninc >= kEnsMinEndpointP) break; }
      best[f] = b;
    }
    r.env.push_back({best[0], best[1], best[2], best[3]});
  }
  std::stable_sort(r.env.begin(), r.env.end(), [](const Seg &a, const Seg &b) { return a.sqfrom != b.sqfrom ? a.sqfrom < b.sqfrom : a.sqto < b.sqto; });
}

// The ensembles of a list of regions, in two halves so that the device works on them while the host drives the
// envelope stage of the single-domain regions: ens_begin queues Forward + trace kernels + one result copy of the first
// workspace-sized batch on the worker's ensemble stream; ens_end waits, clusters, and runs what is left.
struct EnsJob {
  std::vector<RegionReq> req; std::vector<int> cap;
  std::vector<std::pair<size_t, size_t>> batches;      // [first, last) of req
  std::vector<EnsWork> ew;                              // work of the batch in flight
  uint64_t res_floats = 0; bool in_flight = false; size_t next_batch = 0;
};

void ens_queue_batch(Worker *ctx, const ckm_profiles *p, const ckm_seqs *s, EnsJob &job) {
  auto al = [](uint64_t v) { return (v + 31) & ~(uint64_t)31; };
  const auto range = job.batches[job.next_batch++];
  job.ew.clear();
  FbBatch b; uint64_t pos = 0; int maxLd = 0, maxMp = 0;
  // results first (counts, segments, sums of every region: ONE copy back), then the matrices
  for (size_t j = range.first; j < range.second; ++j) {
    const RegionReq &r = job.req[j];
    const int Ld = r.jreg - r.ireg + 1, cap = job.cap[j];
    EnsWork e; memset(&e, 0, sizeof(e));
    e.model = r.model; e.seq = r.seq; e.i0 = r.ireg - 1; e.Ld = Ld; e.Lcfg = s->len[r.seq]; e.cap = cap;
    e.nseg_off = pos; pos += 256;
    e.seg_off = pos;  pos += (uint64_t)ENS_NSAMPLES * cap * 4;
    e.n2_off = pos;   pos = al(pos + (uint64_t)Ld);
    job.ew.push_back(e);
  }
  job.res_floats = pos;
  for (size_t k = 0; k < job.ew.size(); ++k) {
    EnsWork &e = job.ew[k];
    const int Mp = p->prof[e.model].fbQ * NL, Ld = e.Ld;
    e.xs_off = pos;    pos = al(pos + (uint64_t)(Ld + 1) * 6);
    e.mx_off = pos;    pos = al(pos + (uint64_t)(Ld + 1) * 3 * Mp);
    e.code_off = pos;  pos = al(pos + ((uint64_t)ENS_NSAMPLES * (Ld + 1) + 1) / 2);
    e.ratio_off = pos; pos = al(pos + (uint64_t)ENS_NSAMPLES * (Ld + 1));
    FbWork w; memset(&w, 0, sizeof(w));
    w.model = e.model; w.seq = e.seq; w.i0 = e.i0; w.Ld = Ld; w.Lcfg = e.Lcfg; w.multihit = 1; w.slot = (uint32_t)k; w.full = 2;
    w.xs_off = e.xs_off; w.mxf_off = e.mx_off;
    b.work.push_back(w);
    maxLd = std::max(maxLd, Ld); maxMp = std::max(maxMp, Mp);
  }
  ctx->ws_ens.ensure(pos * 4 + 256);
  run_fb(ctx, p, s, b, true, false, false, nullptr, ctx->ws_ens.as<float>());   // multihit Forward of every region, M, I and D rows kept
  ctx->enswork.ensure(job.ew.size() * sizeof(EnsWork));
  HIPCHK(hipMemcpyAsync(ctx->enswork.p, job.ew.data(), job.ew.size() * sizeof(EnsWork), hipMemcpyHostToDevice, ctx->ens_stream));
  launch_ensemble(ctx->ens_stream, ctx->enswork.as<EnsWork>(), (uint32_t)job.ew.size(), maxLd, maxMp, p->d_models.as<DevModel>(), s->d_lentab.as<LenEntry>(),
                  s->d_res.as<uint8_t>(), s->d_off.as<uint64_t>(), ctx->ws_ens.as<float>(), ctx->ensseeds.as<uint32_t>());
  HIPCHK(hipGetLastError());
  ctx->h_ens.ensure(job.res_floats * 4);
  HIPCHK(hipMemcpyAsync(ctx->h_ens.p, ctx->ws_ens.p, job.res_floats * 4, hipMemcpyDeviceToHost, ctx->ens_stream));
  job.in_flight = true;
}

void ens_begin(Worker *ctx, const ckm_profiles *p, const ckm_seqs *s, EnsJob &job) {
  if (job.req.empty()) return;
  if (!ctx->ensseeds.p) {
    std::vector<uint32_t> seeds(ENS_NSAMPLES);
    for (int t = 0; t < ENS_NSAMPLES; ++t) seeds[t] = ens_seed(t);
    ctx->ensseeds.ensure(seeds.size() * 4);
    wcopy(ctx, ctx->ensseeds.p, seeds.data(), seeds.size() * 4, hipMemcpyHostToDevice);
  }
  auto al = [](uint64_t v) { return (v + 31) & ~(uint64_t)31; };
  const uint64_t budget_floats = ctx->ws_budget / 4;
  job.batches.clear(); job.next_batch = 0;
  uint64_t pos = 0; size_t first = 0;
  for (size_t j = 0; j < job.req.size(); ++j) {
    const RegionReq &r = job.req[j];
    const uint64_t Mp = p->prof[r.model].fbQ * NL, Ld = r.jreg - r.ireg + 1;
    const uint64_t need = 256 + (uint64_t)ENS_NSAMPLES * job.cap[j] * 4 + al(Ld) + al((Ld + 1) * 6) + al((Ld + 1) * 3 * Mp) +
                          al((ENS_NSAMPLES * (Ld + 1) + 1) / 2) + al(ENS_NSAMPLES * (Ld + 1)) + 64;
    if (need > budget_floats) throw Error(CKM_ENOMEM, "one multi-domain region needs more workspace than the device budget allows");
    if (pos + need > budget_floats) { job.batches.push_back({first, j}); first = j; pos = 0; }
    pos += need;
  }
  job.batches.push_back({first, job.req.size()});
  ens_queue_batch(ctx, p, s, job);
}

void ens_end(Worker *ctx, const ckm_profiles *p, const ckm_seqs *s, EnsJob &job, std::vector<RegionRes> &out) {
  out.clear(); out.resize(job.req.size());
  if (job.req.empty()) return;
  std::vector<size_t> again;
  size_t base = 0;
  for (;;) {
    HIPCHK(hipStreamSynchronize(ctx->ens_stream));
    const float *raw = ctx->h_ens.as<float>();
    for (size_t k = 0; k < job.ew.size(); ++k) {
      const EnsWork &e = job.ew[k]; RegionRes &o = out[base + k];
      const int32_t *ns = reinterpret_cast<const int32_t *>(raw + e.nseg_off);
      const int32_t *sg = reinterpret_cast<const int32_t *>(raw + e.seg_off);
      bool overflow = false;
      for (int t = 0; t < ENS_NSAMPLES; ++t) overflow |= ns[t] < 0;
      if (overflow) { again.push_back(base + k); continue; }      // more domains in one trace than slots: redo with a larger table
      o.cap = e.cap; o.nseg.assign(ns, ns + ENS_NSAMPLES); o.segs.assign((size_t)ENS_NSAMPLES * e.cap, Seg{0, 0, 0, 0});
      for (int t = 0; t < ENS_NSAMPLES; ++t)
        for (int d = 0; d < ns[t]; ++d) {          // the device walks backwards: last domain first
          const int32_t *q4 = sg + ((size_t)t * e.cap + (ns[t] - 1 - d)) * 4;
          o.segs[(size_t)t * e.cap + d] = Seg{q4[0], q4[1], q4[2], q4[3]};
        }
      o.n2sum.assign(raw + e.n2_off, raw + e.n2_off + e.Ld);
    }
    pool_run(ctx, job.ew.size(), 1, [&](size_t lo, size_t hi) { for (size_t k = lo; k < hi; ++k) if (!out[base + k].nseg.empty()) cluster_ensemble(out[base + k]); });
    base += job.ew.size();
    if (job.next_batch >= job.batches.size()) break;
    ens_queue_batch(ctx, p, s, job);
  }
  if (!again.empty()) {
    EnsJob redo; std::vector<RegionRes> r2;
    for (size_t j : again) { redo.req.push_back(job.req[j]); redo.cap.push_back(std::min(job.req[j].jreg - job.req[j].ireg + 1, job.cap[j] * 8)); }
    ens_begin(ctx, p, s, redo); ens_end(ctx, p, s, redo, r2);
    for (size_t k = 0; k < again.size(); ++k) out[again[k]] = std::move(r2[k]);
  }
}

constexpr int kEnsCap0 = 16;      // segment slots per trace on the first attempt

void run_ensembles(Worker *ctx, const ckm_profiles *p, const ckm_seqs *s, const std::vector<RegionReq> &req, std::vector<RegionRes> &out) {
  EnsJob job; job.req = req;
  for (const auto &r : req) job.cap.push_back(std::min(r.jreg - r.ireg + 1, kEnsCap0));
  ens_begin(ctx, p, s, job);
  ens_end(ctx, p, s, job, out);
}

void fill_null2(float *null2) {   // degenerate symbols: plain average of the odds of their residues
  static const char *sym = "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~";
  auto member = [&](int x, int y) {
    switch (sym[x]) { case 'B': return sym[y] == 'D' || sym[y] == 'N'; case 'J': return sym[y] == 'I' || sym[y] == 'L';
                      case 'Z': return sym[y] == 'E' || sym[y] == 'Q'; case 'O': return sym[y] == 'K'; case 'U': return sym[y] == 'C'; default: return true; } };
  for (int x = 21; x <= 26; ++x) { float r = 0.f; int n = 0; for (int y = 0; y < K; ++y) if (member(x, y)) { r += null2[y]; ++n; } null2[x] = r / (float)n; }
  null2[20] = null2[27] = null2[28] = 1.0f;
}

}  // namespace

struct SearchPlan {        // which models run against which sequence lists
  std::vector<std::vector<uint32_t>> model_bins;   // per model: bins (sorted)
};

typedef std::map<std::pair<uint32_t, uint32_t>, std::vector<Hit>> HitMap;     // (bin, model) -> hits

// The whole filter cascade + domain stage for a subset of the models, on one worker.
struct SeqRange { std::vector<uint32_t> lo, hi; std::vector<uint64_t> res; uint64_t tag = 0; };   // per bin: [lo, hi) of s->order, residues in it

static void cascade(Worker *ctx, ckm_ctx *owner, int my_turn, const ckm_profiles *p, const ckm_seqs *s, const SeqRange &rng, const std::vector<uint32_t> &my_models,
                    const std::vector<std::vector<uint32_t>> &model_bins, HitMap &by_bin_model) {
  HIPCHK(hipSetDevice(ctx->device));
  const double t_start = now_ms();
  ckm_search_stats &st = ctx->stats;
  memset(&st, 0, sizeof(st));
  const DevModel *dm = p->d_models.as<DevModel>();
  const LenEntry *lt = s->d_lentab.as<LenEntry>();
  const uint8_t *res = s->d_res.as<uint8_t>();
  const uint64_t *off = s->d_off.as<uint64_t>();
  const int32_t *dlen = s->d_len.as<int32_t>();

  // ---- stage 1: SSV over every pair, chunked by a pair budget ----
  std::vector<Cand> cands;
  bool took_turn = false;
  struct TurnGuard {      // a worker that never reaches an SSV phase (no pairs, or an error) still passes the turn on
    ckm_ctx *o; int t; bool *took;
    ~TurnGuard() { if (*took) return; std::unique_lock<std::mutex> l(o->ssv_mutex); o->ssv_cv.wait(l, [&] { return o->ssv_turn == t; }); o->ssv_turn++; o->ssv_cv.notify_all(); }
  } turn_guard{owner, my_turn, &took_turn};
  {
    uint64_t pair_budget = (uint64_t)1 << 29;                  // pairs per SSV chunk (2 B of maxV each); CKM_PAIR_BUDGET overrides (tests)
    if (const char *e = getenv("CKM_PAIR_BUDGET")) pair_budget = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
    size_t i0 = 0;
    while (i0 < my_models.size()) {
      // gather models of this chunk
      struct MW { uint32_t model; uint64_t pair_base; uint64_t npairs; };
      const bool i0_was_first = (i0 == 0);
      std::vector<MW> mws; uint64_t npairs = 0; size_t i1 = i0;
      for (; i1 < my_models.size(); ++i1) {
        const uint32_t m1 = my_models[i1];
        uint64_t n = 0;
        for (uint32_t b : model_bins[m1]) n += rng.hi[b] - rng.lo[b];
        if (n == 0) continue;
        if (npairs + n > pair_budget && !mws.empty()) break;
        mws.push_back({m1, npairs, n}); npairs += n;
      }
      i0 = i1;
      if (mws.empty() || npairs == 0) continue;
      std::unique_lock<std::mutex> ssv_lock(owner->ssv_mutex);     // one SSV phase at a time (VALU-bound); released after the finish kernel
      if (!took_turn) owner->ssv_cv.wait(ssv_lock, [&] { return owner->ssv_turn == my_turn; });
      CKM_TRACE_PT("ssv turn taken");
      // The block table depends only on (profiles, sequences, models and their bins): reuse the resident one when the
      // previous call on this worker had the same plan (lineage_wf scans the same bins twice; bench repeats steps).
      std::vector<uint64_t> key{p->uid, s->uid, pair_budget, (uint64_t)i0, rng.tag};
      for (auto &mw : mws) { key.push_back(0xffffffffull + mw.model); for (uint32_t b : model_bins[mw.model]) key.push_back(b); }
      std::vector<std::pair<int, std::pair<size_t, size_t>>> groups;
      size_t nblocks_total = 0;
      const bool single_chunk = (i1 == my_models.size() && i0_was_first);
      if (single_chunk && key == ctx->plan_key) {
        groups = ctx->plan_groups; nblocks_total = ctx->plan_nblocks;
        st.pairs_ssv += ctx->plan_pairs; st.residue_hmm += ctx->plan_residue_hmm; st.cells_ssv += ctx->plan_cells;
      } else {
        std::map<int, std::vector<SsvBlockWork>> byQ;
        uint64_t c_pairs = 0, c_res = 0, c_cells = 0;
        for (auto &mw : mws) {
          const int Q = p->prof[mw.model].ssvQ; const int threads = ssv_threads_for(Q); const uint32_t per_block = (uint32_t)threads / 64 * 4 * 4;
          uint64_t pb = mw.pair_base;
          for (uint32_t b : model_bins[mw.model]) {
            const uint32_t o0 = rng.lo[b], n = rng.hi[b] - o0;
            for (uint32_t a = 0; a < n; a += per_block) {
              SsvBlockWork w; w.model = mw.model; w.list_start = o0 + a; w.count = std::min(per_block, n - a); w.pair_start = (uint32_t)(pb + a);
              byQ[Q].push_back(w);
            }
            pb += n; c_res += rng.res[b]; c_cells += rng.res[b] * (uint64_t)p->prof[mw.model].M;
          }
          c_pairs += mw.npairs;
        }
        st.pairs_ssv += c_pairs; st.residue_hmm += c_res; st.cells_ssv += c_cells;
        std::vector<SsvBlockWork> allw;
        for (auto &kv : byQ) {
          // longest blocks first inside a launch (a block's time is set by its first = longest sequence): without this
          // the few very long sequences of each bin start late and leave most CUs idle at the end of every launch
          std::stable_sort(kv.second.begin(), kv.second.end(), [&](const SsvBlockWork &x, const SsvBlockWork &y) {
            return s->len[s->order[x.list_start]] > s->len[s->order[y.list_start]]; });
          groups.push_back({kv.first, {allw.size(), kv.second.size()}}); allw.insert(allw.end(), kv.second.begin(), kv.second.end());
        }
        nblocks_total = allw.size();
        ctx->work.ensure(allw.size() * sizeof(SsvBlockWork));
        HIPCHK(hipMemcpyAsync(ctx->work.p, allw.data(), allw.size() * sizeof(SsvBlockWork), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));            // allw goes out of scope
        if (single_chunk) { ctx->plan_key = key; ctx->plan_groups = groups; ctx->plan_nblocks = nblocks_total; ctx->plan_pairs = c_pairs; ctx->plan_residue_hmm = c_res; ctx->plan_cells = c_cells; }
        else ctx->plan_key.clear();
      }
      ctx->maxv.ensure(npairs * 2 + 64);
      uint32_t cap_surv = (uint32_t)std::max<uint64_t>(1 << 16, npairs / 8), cap_nores = (uint32_t)std::max<uint64_t>(1 << 14, npairs / 64);
      for (int attempt = 0;; ++attempt) {
        ctx->surv.ensure((size_t)cap_surv * sizeof(PairRec)); ctx->nores.ensure((size_t)cap_nores * sizeof(PairRec)); ctx->counters.ensure(64);
        HIPCHK(hipMemsetAsync(ctx->counters.p, 0, 64, ctx->stream));
        HIPCHK(hipEventRecord(ctx->ev[0], ctx->stream));
        if (attempt == 0) {
          // register classes go round-robin over 4 streams (heaviest first) so the tail of one launch -- a few very long
          // sequences -- is covered by the next launch; ev[0]..ev[1] on the main stream brackets all of them
          constexpr int NS = 4;
          for (int k = 0; k < NS; ++k) HIPCHK(hipStreamWaitEvent(ctx->side[k], ctx->ev[0], 0));
          int gi = 0;
          for (auto it = groups.rbegin(); it != groups.rend(); ++it, ++gi) {
            auto &g = *it;
            if (launch_ssv(g.first, (int)g.second.second, ssv_threads_for(g.first), ctx->side[gi % NS], ctx->work.as<SsvBlockWork>() + g.second.first, dm, res, off, dlen,
                           s->d_order.as<uint32_t>(), ctx->maxv.as<uint16_t>()))
              throw Error(CKM_ERANGE, "no SSV kernel instance for this model length");
            st.ssv_launches++;
          }
          for (int k = 0; k < NS; ++k) { HIPCHK(hipEventRecord(ctx->ev[2 + k], ctx->side[k])); HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev[2 + k], 0)); }
        }
        HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
        FinishArgs fa{dm, lt, dlen, s->d_order.as<uint32_t>(), ctx->work.as<SsvBlockWork>(), ctx->maxv.as<uint16_t>(),
                      ctx->surv.as<PairRec>(), ctx->counters.as<uint32_t>(), cap_surv, ctx->nores.as<PairRec>(), ctx->counters.as<uint32_t>() + 1, cap_nores};
        launch_msv_finish(ctx->stream, fa, (uint32_t)nblocks_total);
        HIPCHK(hipGetLastError());
        uint32_t cnt[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(cnt, ctx->counters.p, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (attempt == 0) { float ms = 0.f; HIPCHK(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1])); st.ms_ssv += ms; }
        CKM_TRACE_PT("ssv + msv_finish done");
        if (cnt[0] > cap_surv || cnt[1] > cap_nores) { cap_surv = std::max(cap_surv, cnt[0]); cap_nores = std::max(cap_nores, cnt[1]); continue; }
        std::vector<PairRec> nr(cnt[1]);
        ctx->h_a.ensure((size_t)cnt[0] * sizeof(PairRec) + 16);
        const PairRec *sv = ctx->h_a.as<PairRec>();
        if (cnt[0]) wcopy(ctx, ctx->h_a.p, ctx->surv.p, (size_t)cnt[0] * sizeof(PairRec), hipMemcpyDeviceToHost);
        if (cnt[1]) wcopy(ctx, nr.data(), ctx->nores.p, (size_t)cnt[1] * sizeof(PairRec), hipMemcpyDeviceToHost);
        if (!took_turn) { took_turn = true; owner->ssv_turn++; owner->ssv_cv.notify_all(); }
        ssv_lock.unlock();
        cands.reserve(cands.size() + cnt[0]);
        for (uint32_t k = 0; k < cnt[0]; ++k) { Cand c; c.r = sv[k]; c.alive = true; c.fwdsc = 0; c.fwd_xC = 0; c.slot = 0; cands.push_back(c); }
        if (!nr.empty()) {     // exact multi-hit MSV for the pairs where J could be used
          st.pairs_msv_full += nr.size();
          std::vector<float> usc;
          run_msv_exact(ctx, p, s, nr, usc, nullptr);
          for (size_t i = 0; i < nr.size(); ++i) {
            const float nullsc = s->lentab[s->len[nr[i].seq]].nullsc;
            if (bits(usc[i], nullsc) >= p->prof[nr[i].model].thr_msv_f1) { Cand c; c.r = nr[i]; c.r.usc = usc[i]; c.alive = true; c.fwdsc = 0; c.fwd_xC = 0; c.slot = 0; cands.push_back(c); }
          }
        }
        break;
      }
    }
  }
  // (the atomic append order of the survivors is arbitrary; every later stage is per pair, and the rows are
  //  ordered at the end, so no sort is needed here)
  const double t_filters0 = now_ms();

  CKM_TRACE_PT("stage1 done (ssv, msv_finish, msv_full)");
  // ---- stage 2: bias filter ----
  std::vector<PairRec> cr(cands.size());
  for (size_t i = 0; i < cands.size(); ++i) cr[i] = cands[i].r;
  st.pairs_bias = cands.size();
  std::vector<uint8_t> need_vit(cands.size(), 0);
  if (!cands.empty()) {
    ctx->cand.ensure(cr.size() * sizeof(PairRec)); ctx->raw.ensure(cr.size() * 12);
    HIPCHK(hipMemcpyAsync(ctx->cand.p, cr.data(), cr.size() * sizeof(PairRec), hipMemcpyHostToDevice, ctx->stream));
    launch_bias(ctx->stream, ctx->cand.as<PairRec>(), (uint32_t)cr.size(), dm, lt, res, off, dlen, ctx->raw.as<float>());
    HIPCHK(hipGetLastError());
    ctx->h_a.ensure(cr.size() * 12 + 16);
    const float *raw = ctx->h_a.as<float>();
    HIPCHK(hipMemcpyAsync(ctx->h_a.p, ctx->raw.p, cr.size() * 12, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    pool_run(ctx, cands.size(), 4096, [&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; ++i) {
        Cand &c = cands[i];
        const int L = s->len[c.r.seq];
        const float p1 = (float)L / (float)(L + 1);
        const float nullsc = (float)(log((double)raw[i * 3]) + (double)raw[i * 3 + 1] * kLn2);
        c.r.filtersc = nullsc + (float)L * logf(p1) + logf(1.0f - p1);
        const float sc = bits(c.r.usc, c.r.filtersc);
        const HostProfile &hp = p->prof[c.r.model];
        if (!(sc >= hp.thr_msv_f1)) { c.alive = false; continue; }
        need_vit[i] = !(sc >= hp.thr_msv_f2);
      }
    });
  }
  // ---- stage 3: Viterbi filter ----
  CKM_TRACE_PT("bias done");
  {
    std::map<int, std::vector<uint32_t>> byQ;
    for (size_t i = 0; i < cands.size(); ++i) if (cands[i].alive && need_vit[i]) byQ[p->prof[cands[i].r.model].vitQH].push_back((uint32_t)i);
    std::vector<uint32_t> flat; std::vector<std::pair<int, std::pair<size_t, size_t>>> groups;
    for (auto &kv : byQ) {
      // (survivors were appended by SSV blocks that ran longest-first, so these lists are already roughly length-ordered)
      groups.push_back({kv.first, {flat.size(), kv.second.size()}}); flat.insert(flat.end(), kv.second.begin(), kv.second.end());
    }
    st.pairs_vit = flat.size();
    if (!flat.empty()) {
      for (size_t i = 0; i < cands.size(); ++i) cr[i] = cands[i].r;
      HIPCHK(hipMemcpyAsync(ctx->cand.p, cr.data(), cr.size() * sizeof(PairRec), hipMemcpyHostToDevice, ctx->stream));
      ctx->fbidx.ensure(flat.size() * 4); ctx->vitx.ensure(cands.size() * 4); ctx->vits.ensure(cands.size() * 4); ctx->vitf.ensure(cands.size() * 4);
      HIPCHK(hipMemcpyAsync(ctx->fbidx.p, flat.data(), flat.size() * 4, hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));            // uploads done; the launches go to the side streams
      // pass 1: the J-free fast kernel (exact, or a lower bound with its flag set); pass 2: the exact kernel for the pairs
      // whose bound fails F2 although the J state could have lifted them
      auto run_vit = [&](const std::vector<std::pair<int, std::pair<size_t, size_t>>> &grp, bool fast) {
        int gi = 0;
        for (auto it = grp.rbegin(); it != grp.rend(); ++it, ++gi) {
          auto &g = *it;
          if (launch_vit(g.first, ctx->side[gi % 4], ctx->cand.as<PairRec>(), ctx->fbidx.as<uint32_t>() + g.second.first, (uint32_t)g.second.second, dm, lt, res, off, dlen,
                         ctx->vitx.as<int32_t>(), ctx->vits.as<float>(), ctx->vitf.as<uint32_t>(), fast))
            throw Error(CKM_ERANGE, "no Viterbi kernel instance for this model length");
        }
        HIPCHK(hipGetLastError());
        for (int k = 0; k < 4; ++k) HIPCHK(hipStreamSynchronize(ctx->side[k]));
      };
      run_vit(groups, true);
      ctx->h_a.ensure(cands.size() * 4 + 16); ctx->h_b.ensure(cands.size() * 4 + 16);
      const float *vsc = ctx->h_a.as<float>(); const uint32_t *vfl = ctx->h_b.as<uint32_t>();
      HIPCHK(hipMemcpyAsync(ctx->h_a.p, ctx->vits.p, cands.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
      wcopy(ctx, ctx->h_b.p, ctx->vitf.p, cands.size() * 4, hipMemcpyDeviceToHost);
      std::map<int, std::vector<uint32_t>> redoQ;
      for (uint32_t i : flat) {
        Cand &c = cands[i];
        if (bits(vsc[i], c.r.filtersc) >= p->prof[c.r.model].thr_vit_f2) continue;          // passes already on the bound
        if (vfl[i]) redoQ[p->prof[c.r.model].vitQH].push_back(i); else c.alive = false;
      }
      if (!redoQ.empty()) {
        std::vector<uint32_t> flat2; std::vector<std::pair<int, std::pair<size_t, size_t>>> groups2;
        for (auto &kv : redoQ) { groups2.push_back({kv.first, {flat2.size(), kv.second.size()}}); flat2.insert(flat2.end(), kv.second.begin(), kv.second.end()); }
        st.pairs_vit_exact = flat2.size();
        wcopy(ctx, ctx->fbidx.p, flat2.data(), flat2.size() * 4, hipMemcpyHostToDevice);
        run_vit(groups2, false);
        wcopy(ctx, ctx->h_a.p, ctx->vits.p, cands.size() * 4, hipMemcpyDeviceToHost);
        for (uint32_t i : flat2) { Cand &c = cands[i]; if (!(bits(vsc[i], c.r.filtersc) >= p->prof[c.r.model].thr_vit_f2)) c.alive = false; }
      }
    }
  }
  st.ms_filters = now_ms() - t_filters0;
  CKM_TRACE_PT("viterbi done");
  const double t_fb0 = now_ms();
  // ---- stage 4: Forward parser (multihit, whole sequence), F3 ----
  FbBatch fb;
  std::vector<uint32_t> fb_cand;
  uint64_t aux_base = 0;
  {
    uint64_t pos = 0, aux_total = 0;
    for (size_t i = 0; i < cands.size(); ++i) if (cands[i].alive) {
      const int L = s->len[cands[i].r.seq];
      FbWork w; memset(&w, 0, sizeof(w));
      w.model = cands[i].r.model; w.seq = cands[i].r.seq; w.i0 = 0; w.Ld = L; w.Lcfg = L; w.multihit = 1; w.full = 0; w.slot = (uint32_t)fb.work.size();
      w.xs_off = pos; pos += ((uint64_t)(L + 1) * 6 + 31) & ~(uint64_t)31;
      aux_total += ((uint64_t)(L + 1) * 3 + 31) & ~(uint64_t)31;
      fb.work.push_back(w); fb_cand.push_back((uint32_t)i);
    }
    st.pairs_fwd = fb.work.size();
    aux_base = pos;      // decoding terms of the F3 survivors are laid out compactly from here after the Forward pass
    if ((pos + aux_total) * 4 > ctx->ws_budget) throw Error(CKM_ENOMEM, "Forward special-row workspace exceeds the device budget; search fewer bins per call");
    ctx->ws.ensure((pos + aux_total) * 4 + 256);
    run_fb(ctx, p, s, fb, true, false, false, nullptr);
  }
  EventIndex fev; fev.build(fb.events, fb.work.size());
  CKM_TRACE_PT("fwd parser kernels+copies done, event index built");
  std::vector<uint32_t> passers;
  pool_run(ctx, fb.work.size(), 512, [&](size_t lo, size_t hi) {
    for (size_t k = lo; k < hi; ++k) {
      Cand &c = cands[fb_cand[k]];
      const LenEntry &le = s->lentab[s->len[c.r.seq]];
      c.fwd_xC = fb.fout[k].xC; c.slot = (uint32_t)k;
      c.fwdsc = finish_forward(fb.fout[k].xC, le.move_m, fev.scales((uint32_t)k));
      if (!(bits(c.fwdsc, c.r.filtersc) >= p->prof[c.r.model].thr_fwd_f3)) c.alive = false;
    }
  });
  for (size_t k = 0; k < fb.work.size(); ++k) if (cands[fb_cand[k]].alive) passers.push_back((uint32_t)k);
  st.pairs_dom = passers.size();
  // ---- stage 5: Backward parser + posterior domain heuristics ----
  CKM_TRACE_PT("fwd post done");
  std::vector<EnvReq> envreq; std::vector<std::pair<size_t, size_t>> env_of_pass(passers.size());   // [first, count)
  std::vector<int> nregions(passers.size(), 0);
  struct Item { uint32_t pass; int i, j, region; };       // regions in sequence order; region >= 0: resolved by the trace ensemble
  std::vector<Item> items; std::vector<RegionReq> regreq; std::vector<RegionRes> regres;
  std::vector<int> env_region;                            // per envelope: index into regres or -1
  if (!passers.empty()) {
    uint64_t ap = aux_base;
    for (uint32_t k : passers) { fb.work[k].aux_off = ap; ap += ((uint64_t)(fb.work[k].Ld + 1) * 3 + 31) & ~(uint64_t)31; }
    run_fb(ctx, p, s, fb, false, true, false, &passers);
    // pull the decoding terms of the passers in one copy
    std::vector<float> dec_all(ap - aux_base);     // pageable on purpose: the region scan below re-reads it; pinned memory reads slowly from the CPU
    const float *dec_all_p = dec_all.data();
    if (!dec_all.empty()) wcopy(ctx, dec_all.data(), ctx->ws.as<float>() + aux_base, dec_all.size() * 4, hipMemcpyDeviceToHost);
    std::vector<std::vector<Item>> found(passers.size());
    pool_run(ctx, passers.size(), 64, [&](size_t qlo, size_t qhi) {
      std::vector<float> btot, etot, mocc;
      for (size_t q = qlo; q < qhi; ++q) {
        const FbWork &w = fb.work[passers[q]];
        const int L = w.Ld;
        const float *dec = dec_all_p + (w.aux_off - aux_base);
        btot.assign(L + 1, 0.f); etot.assign(L + 1, 0.f); mocc.assign(L + 1, 0.f);
        for (int i = 1; i <= L; ++i) { btot[i] = btot[i - 1] + dec[(size_t)i * 3]; etot[i] = etot[i - 1] + dec[(size_t)i * 3 + 1]; mocc[i] = 1.0f - dec[(size_t)i * 3 + 2]; }
        int i = -1; bool triggered = false;
        for (int j = 1; j <= L; ++j) {
          if (!triggered) {
            if (mocc[j] - (btot[j] - btot[j - 1]) < RT2) i = j; else if (i == -1) i = j;
            if (mocc[j] >= RT1) triggered = true;
          } else if (mocc[j] - (etot[j] - etot[j - 1]) < RT2) {
            nregions[q]++;
            float mx = -1.0f;
            for (int z = i; z <= j; ++z) { const float a = etot[z] - etot[i - 1], b = btot[j] - btot[z - 1]; const float en = a < b ? a : b; if (en > mx) mx = en; }
            found[q].push_back({(uint32_t)q, i, j, (mx >= RT3) ? 0 : -1});       // region >= 0: multi-domain, numbered below
            i = -1; triggered = false;
          }
        }
      }
    });
    for (size_t q = 0; q < passers.size(); ++q) for (Item im : found[q]) {
      const FbWork &w = fb.work[passers[q]];
      if (im.region >= 0) { im.region = (int)regreq.size(); regreq.push_back({w.model, w.seq, im.i, im.j}); }
      items.push_back(im);
    }
  }
  // multi-domain regions: 200 stochastic tracebacks each, clustered into envelopes.  (Queueing them beside the envelope
  // stage of the single-domain regions was tried: the second envelope pass it needs costs more than it hides.)
  st.regions_multi = regreq.size();
  CKM_TRACE_PT("bwd parser + region scan done");
  run_ensembles(ctx, p, s, regreq, regres);
  {
    size_t it = 0;
    for (size_t q = 0; q < passers.size(); ++q) {
      const FbWork &w = fb.work[passers[q]];
      env_of_pass[q].first = envreq.size();
      for (; it < items.size() && items[it].pass == q; ++it) {
        const Item &im = items[it];
        if (im.region < 0) { envreq.push_back({w.model, w.seq, im.i, im.j}); env_region.push_back(-1); continue; }
        int last_j2 = 0;
        for (const Seg &e : regres[im.region].env) {
          const int i2 = e.sqfrom + im.i - 1, j2 = e.sqto + im.i - 1;
          if (i2 <= last_j2) continue;        // overlapping envelopes: the later one is skipped, as HMMER does
          envreq.push_back({w.model, w.seq, i2, j2}); env_region.push_back(im.region);
          last_j2 = j2;
        }
      }
      env_of_pass[q].second = envreq.size() - env_of_pass[q].first;
    }
  }
  st.ms_fwdbwd = now_ms() - t_fb0;
  CKM_TRACE_PT("ensembles done");
  const double t_dom0 = now_ms();
  // ---- stage 6: envelope rescoring ----
  std::vector<EnvRes> envres;
  rescore_envelopes(ctx, p, s, envreq, envres);
  st.envelopes = envreq.size();
  st.ms_domains = now_ms() - t_dom0;
  CKM_TRACE_PT("envelopes done");
  const double t_host0 = now_ms();
  // ---- stage 7: scores, thresholds, rows ----
  std::vector<std::pair<size_t, size_t>> items_of(passers.size(), {0, 0});
  { size_t it = 0; for (size_t q = 0; q < passers.size(); ++q) { items_of[q].first = it; while (it < items.size() && items[it].pass == q) ++it; items_of[q].second = it; } }
  std::vector<Hit> hit_of(passers.size()); std::vector<uint8_t> has_hit(passers.size(), 0);
  pool_run(ctx, passers.size(), 32, [&](size_t qlo, size_t qhi) {
  std::vector<float> n2sc;
  for (size_t q = qlo; q < qhi; ++q) {
    const Cand &c = cands[fb_cand[passers[q]]];
    const HostHMM &hm = p->hmm[c.r.model];
    const int L = s->len[c.r.seq];
    const uint8_t *dsq = s->dsq.data() + s->off[c.r.seq];
    const float nullsc = s->lentab[L].nullsc;
    n2sc.assign((size_t)L + 2, 0.f);
    Hit h; h.model = c.r.model; h.seq = c.r.seq; h.L = L; h.nreported = 0;
    int nenv = 0;
    for (size_t item_at = items_of[q].first; item_at < items_of[q].second; ++item_at) if (items[item_at].region >= 0) {
      // null2 of an ensemble region: log of the mean odds ratio over the traces, for every residue of the region
      const Item &im = items[item_at]; const RegionRes &rr = regres[im.region];
      for (int pos = im.i; pos <= im.j; ++pos) n2sc[pos] = logf(rr.n2sum[pos - im.i] / (float)ENS_NSAMPLES);
    }
    for (size_t e = env_of_pass[q].first; e < env_of_pass[q].first + env_of_pass[q].second; ++e) {
      ++nenv;
      EnvRes &er = envres[e];
      if (!er.ok) continue;
      float null2[KP]; for (int x = 0; x < K; ++x) null2[x] = er.null2[x];
      fill_null2(null2);
      Domain d; memset(&d, 0, sizeof(d));
      d.ienv = envreq[e].ienv; d.jenv = envreq[e].jenv; d.envsc = er.envsc; d.oasc = er.oasc;
      d.hmm_from = er.hmm_from; d.hmm_to = er.hmm_to; d.ali_from = er.ali_from; d.ali_to = er.ali_to;
      float ln2[KP + 1];
      for (int x = 0; x < KP; ++x) ln2[x] = logf(null2[x]);          // same value the per-position logf would give
      ln2[KP] = 0.f;
      float dc = 0.f;
      if (env_region[e] >= 0) { for (int pos = d.ienv; pos <= d.jenv; ++pos) dc += n2sc[pos]; }
      else for (int pos = d.ienv; pos <= d.jenv; ++pos) { const float v = ln2[dsq[pos - 1]]; n2sc[pos] = v; dc += v; }
      d.domcorrection = dc;
      h.dom.push_back(d);
    }
    if (nregions[q] == 0 || nenv == 0 || h.dom.empty()) continue;
    float seqbias = 0.f;
    for (int i = 0; i <= L; ++i) seqbias += n2sc[i];
    seqbias = flogsum(0.0f, logf(kOmega) + seqbias);
    float pre_score = (float)((double)(c.fwdsc - nullsc) / kLn2);
    float seq_score = (float)((double)(c.fwdsc - (nullsc + seqbias)) / kLn2);
    float sum_score = 0.f; int Ld = 0; seqbias = 0.f;
    for (auto &d : h.dom) if (d.envsc - d.domcorrection > 0.0f) { sum_score += d.envsc; Ld += d.jenv - d.ienv + 1; seqbias += d.domcorrection; }
    seqbias = flogsum(0.0f, logf(kOmega) + seqbias);
    sum_score += (float)((double)(L - Ld) * log((double)((float)L / (float)(L + 3))));
    const float pre2 = (float)((double)(sum_score - nullsc) / kLn2);
    sum_score = (float)((double)(sum_score - (nullsc + seqbias)) / kLn2);
    if (Ld > 0 && sum_score > seq_score) { seq_score = sum_score; pre_score = pre2; }
    h.pre_score = pre_score; h.score = seq_score;
    h.lnP = exp_logsurv(seq_score, hm.evparam[4], hm.evparam[5]);
    for (auto &d : h.dom) {
      const int ld = d.jenv - d.ienv + 1;
      const float bs = d.envsc + (float)((double)(L - ld) * log((double)((float)L / (float)(L + 3))));
      d.dombias = flogsum(0.0f, logf(kOmega) + d.domcorrection);
      d.bitscore = (float)((double)(bs - (nullsc + d.dombias)) / kLn2);
      d.lnP = exp_logsurv(d.bitscore, hm.evparam[4], hm.evparam[5]);
      d.reported = false;
    }
    hit_of[q] = std::move(h); has_hit[q] = 1;
  }
  });
  for (size_t q = 0; q < passers.size(); ++q) if (has_hit[q]) by_bin_model[{s->seq_bin[hit_of[q].seq], hit_of[q].model}].push_back(std::move(hit_of[q]));
  st.ms_host = now_ms() - t_host0;
  st.ms_total = now_ms() - t_start;
  CKM_TRACE_PT("cascade done");
}

static void do_search(ckm_ctx *c, const ckm_profiles *p, const ckm_seqs *s, const uint32_t *model_off, const uint32_t *model_idx,
                      double E, double domE, ckm_hits *hits) {
  HIPCHK(hipSetDevice(c->device));
  const double t_start = now_ms();
  const uint32_t nmodels = (uint32_t)p->hmm.size(), nbins = s->nbins;
  // ---- plan ----
  std::vector<std::vector<uint32_t>> bin_models(nbins);
  for (uint32_t b = 0; b < nbins; ++b) {
    if (model_off) { for (uint32_t k = model_off[b]; k < model_off[b + 1]; ++k) { if (model_idx[k] >= nmodels) throw Error(CKM_EINVAL, "model index out of range"); bin_models[b].push_back(model_idx[k]); } }
    else { bin_models[b].resize(nmodels); std::iota(bin_models[b].begin(), bin_models[b].end(), 0u); }
  }
  std::vector<std::vector<uint32_t>> model_bins(nmodels);
  for (uint32_t b = 0; b < nbins; ++b) {
    std::vector<uint32_t> uniq = bin_models[b]; std::sort(uniq.begin(), uniq.end()); uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    for (uint32_t m : uniq) model_bins[m].push_back(b);
  }
  // models -> workers: deal them out by decreasing work (pairs x M) so both chunks cost about the same
  std::vector<uint32_t> active; std::vector<double> cost(nmodels, 0.0);
  for (uint32_t m = 0; m < nmodels; ++m) if (!model_bins[m].empty()) { active.push_back(m); double n = 0; for (uint32_t b : model_bins[m]) n += (double)s->bin_res[b]; cost[m] = n * p->prof[m].M; }
  std::stable_sort(active.begin(), active.end(), [&](uint32_t x, uint32_t y) { return cost[x] > cost[y]; });
  // several workers only pay off on a large search (every one of them adds its own launches and host threads)
  uint64_t total_pairs = 0;
  for (uint32_t m : active) for (uint32_t b : model_bins[m]) total_pairs += s->order_off[b + 1] - s->order_off[b];
  uint64_t min_pairs = 300000;
  if (const char *e = getenv("CKM_WORKER_MIN_PAIRS")) min_pairs = strtoull(e, nullptr, 10);      // tests: small searches on several workers
  const int nw = (total_pairs >= min_pairs * c->nworkers) ? c->nworkers : 1;
  // Two workers split the SEQUENCES, not the models: every stage behind SSV is bound by the row-by-row chain of the longest
  // sequence it holds, so the few long sequences (a prefix of each bin's length-sorted order) go to worker 0, whose short SSV
  // phase runs first and whose long chains then run underneath the SSV phase of everything else (worker 1).
  std::vector<std::vector<uint32_t>> chunk(nw);
  std::vector<SeqRange> ranges(nw);
  // cut lengths, descending: class k holds the sequences with cut[k-1] >= L > cut[k].  Default: the cuts that give the
  // classes fixed shares of the residues (measured best on cfg2: 16 / 45 / 39 % for three workers, 60 / 40 for two).
  std::vector<int> cuts;
  if (const char *e = getenv("CKM_LEN_SPLIT")) {
    const std::string spec = e; size_t pos = 0;
    while (pos < spec.size()) { size_t q = spec.find(',', pos); if (q == std::string::npos) q = spec.size(); cuts.push_back(atoi(spec.substr(pos, q - pos).c_str())); pos = q + 1; }
    std::sort(cuts.begin(), cuts.end(), std::greater<int>());
  } else if (nw >= 2) {
    static const double shares2[] = {0.60}, shares3[] = {0.16, 0.61}, shares4[] = {0.10, 0.35, 0.65};
    const double *sh = nw == 2 ? shares2 : nw == 3 ? shares3 : shares4;
    std::vector<uint64_t> by_len((size_t)s->maxL + 2, 0);
    uint64_t total = 0;
    for (uint32_t i = 0; i < s->nseq; ++i) { by_len[s->len[i]] += (uint64_t)s->len[i]; total += (uint64_t)s->len[i]; }
    uint64_t acc = 0; int k = 0;
    for (int L = s->maxL; L >= 1 && k < nw - 1; --L) { acc += by_len[L]; if ((double)acc >= sh[k] * (double)total) { cuts.push_back(L - 1); ++k; } }
    while ((int)cuts.size() < nw - 1) cuts.push_back(0);
  }
  if (nw >= 2 && (int)cuts.size() == nw - 1 && cuts.back() > 0) {
    for (int k = 0; k < nw; ++k) { chunk[k] = active; ranges[k].lo.resize(nbins); ranges[k].hi.resize(nbins); ranges[k].res.assign(nbins, 0); ranges[k].tag = 1000 + (uint64_t)k; for (int cv : cuts) ranges[k].tag = ranges[k].tag * 4099 + (uint64_t)cv; }
    for (uint32_t b = 0; b < nbins; ++b) {
      uint32_t at = s->order_off[b];
      for (int k = 0; k < nw; ++k) {
        const int cut = (k < nw - 1) ? cuts[k] : -1;
        uint64_t r = 0; const uint32_t lo = at;
        while (at < s->order_off[b + 1] && s->len[s->order[at]] > cut) { r += (uint64_t)s->len[s->order[at]]; ++at; }
        ranges[k].lo[b] = lo; ranges[k].hi[b] = at; ranges[k].res[b] = r;
      }
    }
  } else {
    // models -> workers, by decreasing work, shares ~ ratio^k
    double ratio = 1.0;
    if (const char *e = getenv("CKM_SPLIT_RATIO")) ratio = std::min(1.0, std::max(0.01, atof(e)));
    std::vector<double> share(nw, 1.0), load(nw, 0.0);
    for (int k = 1; k < nw; ++k) share[k] = share[k - 1] * ratio;
    for (uint32_t m : active) {
      int k = 0;
      for (int j = 1; j < nw; ++j) if (load[j] / share[j] < load[k] / share[k]) k = j;
      chunk[k].push_back(m); load[k] += cost[m];
    }
    for (int k = 0; k < nw; ++k) {
      ranges[k].lo.assign(s->order_off.begin(), s->order_off.end() - 1); ranges[k].hi.assign(s->order_off.begin() + 1, s->order_off.end());
      ranges[k].res = s->bin_res; ranges[k].tag = 0;
    }
  }
  for (auto &ch : chunk) std::sort(ch.begin(), ch.end());
  std::vector<HitMap> maps(nw); std::vector<std::exception_ptr> errs(nw);
  c->ssv_turn = 0;
  g_trace_t0 = now_ms();
  auto run = [&](int k) { try { cascade(&c->w[k], c, k, p, s, ranges[k], chunk[k], model_bins, maps[k]); } catch (...) { errs[k] = std::current_exception(); } };
  std::vector<std::thread> threads;
  for (int k = 1; k < nw; ++k) threads.emplace_back(run, k);
  run(0);
  for (auto &t : threads) t.join();
  for (auto &e : errs) if (e) std::rethrow_exception(e);
  HitMap by_bin_model;
  for (auto &m : maps) for (auto &kv : m) { auto &dst = by_bin_model[kv.first]; for (auto &h : kv.second) dst.push_back(std::move(h)); }
  ckm_search_stats &st = c->stats;
  memset(&st, 0, sizeof(st));
  for (int k = 0; k < nw; ++k) {
    const ckm_search_stats &w = c->w[k].stats;
    st.pairs_ssv += w.pairs_ssv; st.pairs_msv_full += w.pairs_msv_full; st.pairs_bias += w.pairs_bias; st.pairs_vit += w.pairs_vit; st.pairs_vit_exact += w.pairs_vit_exact; st.pairs_fwd += w.pairs_fwd;
    st.pairs_dom += w.pairs_dom; st.envelopes += w.envelopes; st.regions_multi += w.regions_multi; st.cells_ssv += w.cells_ssv; st.residue_hmm += w.residue_hmm; st.ssv_launches += w.ssv_launches;
    st.ms_ssv += w.ms_ssv;                                   // SSV phases are serialised by the mutex: the sum is the kernel time
    st.ms_filters = std::max(st.ms_filters, w.ms_filters); st.ms_fwdbwd = std::max(st.ms_fwdbwd, w.ms_fwdbwd);
    st.ms_domains = std::max(st.ms_domains, w.ms_domains); st.ms_host = std::max(st.ms_host, w.ms_host);
  }
  const double t_host0 = now_ms();
  // rows, bin by bin, models in the bin's own order
  hits->nbins = nbins;
  hits->bin_row_off.assign(nbins + 1, 0);
  for (uint32_t b = 0; b < nbins; ++b) {
    hits->bin_row_off[b] = hits->seq.size();
    const double Z = (double)(s->bin_off[b + 1] - s->bin_off[b]);
    for (uint32_t m : bin_models[b]) {
      auto it = by_bin_model.find({b, m});
      if (it == by_bin_model.end()) continue;
      std::vector<Hit> hs = it->second;      // copy: a model listed twice in one bin reports twice, as two records in the HMM file would
      std::sort(hs.begin(), hs.end(), [&](const Hit &a, const Hit &c) {
        if (a.lnP != c.lnP) return a.lnP < c.lnP;
        const int cmp = s->names[a.seq].compare(s->names[c.seq]);
        if (cmp) return cmp < 0;
        return a.seq < c.seq;
      });
      int nrep = 0;
      for (auto &h : hs) if (exp(h.lnP) * Z <= E) ++nrep;
      const double domZ = (double)nrep;
      for (auto &h : hs) {
        if (!(exp(h.lnP) * Z <= E)) continue;
        for (auto &d : h.dom) { d.reported = exp(d.lnP) * domZ <= domE; if (d.reported) h.nreported++; }
        for (size_t d = 1; d < h.dom.size(); ++d) {
          Domain &a = h.dom[d - 1], &c = h.dom[d];
          if (a.reported && c.reported && a.ali_from == c.ali_from && a.ali_to == c.ali_to && a.hmm_from == c.hmm_from && a.hmm_to == c.hmm_to) {
            Domain &w = (a.bitscore >= c.bitscore) ? c : a; w.reported = false; h.nreported--;
          }
        }
        int nd = 0;
        for (auto &d : h.dom) if (d.reported) {
          ++nd;
          hits->seq.push_back(h.seq); hits->model.push_back(h.model); hits->tlen.push_back(h.L); hits->qlen.push_back(p->hmm[h.model].M);
          hits->full_evalue.push_back(exp(h.lnP) * Z); hits->full_score.push_back(h.score); hits->full_bias.push_back(h.pre_score - h.score);
          hits->dom_idx.push_back(nd); hits->ndom.push_back(h.nreported);
          hits->c_evalue.push_back(exp(d.lnP) * domZ); hits->i_evalue.push_back(exp(d.lnP) * Z);
          hits->dom_score.push_back(d.bitscore); hits->dom_bias.push_back((float)((double)d.dombias * kLog2R));
          hits->hmm_from.push_back(d.hmm_from); hits->hmm_to.push_back(d.hmm_to); hits->ali_from.push_back(d.ali_from); hits->ali_to.push_back(d.ali_to);
          hits->env_from.push_back(d.ienv); hits->env_to.push_back(d.jenv);
          hits->acc.push_back((float)((double)d.oasc / (1.0 + fabs((double)(float)(d.jenv - d.ienv)))));
        }
      }
    }
  }
  hits->bin_row_off[nbins] = hits->seq.size();
  st.ms_host += now_ms() - t_host0;
  st.ms_total = now_ms() - t_start;
}

extern "C" int ckm_search(ckm_ctx *ctx, const ckm_profiles *p, const ckm_seqs *s, const uint32_t *model_off,
                          const uint32_t *model_idx, double E, double domE, ckm_hits **out) {
  return guarded([&] {
    if (!ctx || !p || !s || !out) throw Error(CKM_EINVAL, "NULL argument");
    if ((model_off == nullptr) != (model_idx == nullptr)) throw Error(CKM_EINVAL, "model_off and model_idx must both be given or both be NULL");
    *out = nullptr;
    std::unique_ptr<ckm_hits> h(new ckm_hits());
    do_search(ctx, p, s, model_off, model_idx, E, domE, h.get());
    *out = h.release();
  });
}

extern "C" int ckm_hits_columns(const ckm_hits *h, ckm_hit_columns *o) {
  if (!h || !o) { set_last_error("NULL argument"); return CKM_EINVAL; }
  o->n = h->seq.size(); o->nbins = h->nbins; o->bin_row_off = h->bin_row_off.data();
  o->seq = h->seq.data(); o->model = h->model.data(); o->tlen = h->tlen.data(); o->qlen = h->qlen.data();
  o->full_evalue = h->full_evalue.data(); o->full_score = h->full_score.data(); o->full_bias = h->full_bias.data();
  o->dom_idx = h->dom_idx.data(); o->ndom = h->ndom.data(); o->c_evalue = h->c_evalue.data(); o->i_evalue = h->i_evalue.data();
  o->dom_score = h->dom_score.data(); o->dom_bias = h->dom_bias.data();
  o->hmm_from = h->hmm_from.data(); o->hmm_to = h->hmm_to.data(); o->ali_from = h->ali_from.data(); o->ali_to = h->ali_to.data();
  o->env_from = h->env_from.data(); o->env_to = h->env_to.data(); o->acc = h->acc.data();
  o->target_name = nullptr; o->full_score_d = nullptr; o->dom_score_d = nullptr;
  return CKM_OK;
}

extern "C" void ckm_hits_free(ckm_hits *h) { delete h; }

extern "C" int ckm_last_search_stats(const ckm_ctx *ctx, ckm_search_stats *out) {
  if (!ctx || !out) { set_last_error("NULL argument"); return CKM_EINVAL; }
  *out = ctx->stats;
  return CKM_OK;
}

// ---- domtblout writer -------------------------------------------------------------------------------
extern "C" int ckm_hits_write_domtblout(const ckm_hits *h, const ckm_profiles *p, const ckm_seqs *s, uint32_t bin, const char *path) {
  return guarded([&] {
    if (!h || !p || !s || !path) throw Error(CKM_EINVAL, "NULL argument");
    if (bin >= h->nbins) throw Error(CKM_EINVAL, "bin out of range");
    FILE *f = fopen(path, "w");
    if (!f) throw Error(CKM_EIO, std::string("cannot write ") + path);
    const uint64_t r0 = h->bin_row_off[bin], r1 = h->bin_row_off[bin + 1];
    int tnamew = 20, qnamew = 20, taccw = 10, qaccw = 10;
    for (uint64_t r = r0; r < r1; ++r) {
      tnamew = std::max(tnamew, (int)s->names[h->seq[r]].size());
      const HostHMM &hm = p->hmm[h->model[r]];
      qnamew = std::max(qnamew, (int)hm.name.size());
      if (hm.has_acc) qaccw = std::max(qaccw, (int)hm.acc.size());
    }
    fprintf(f, "#%*s %22s %40s %11s %11s %11s\n", tnamew + qnamew - 1 + 15 + taccw + qaccw, "", "--- full sequence ---",
            "-------------- this domain -------------", "hmm coord", "ali coord", "env coord");
    fprintf(f, "#%-*s %-*s %5s %-*s %-*s %5s %9s %6s %5s %3s %3s %9s %9s %6s %5s %5s %5s %5s %5s %5s %5s %4s %s\n", tnamew - 1, " target name", taccw,
            "accession", "tlen", qnamew, "query name", qaccw, "accession", "qlen", "E-value", "score", "bias", "#", "of", "c-Evalue", "i-Evalue",
            "score", "bias", "from", "to", "from", "to", "from", "to", "acc", "description of target");
    auto dashes = [](int n) { return std::string((size_t)n, '-'); };
    fprintf(f, "#%s %s %s %s %s ", dashes(tnamew - 1).c_str(), dashes(taccw).c_str(), dashes(5).c_str(), dashes(qnamew).c_str(), dashes(qaccw).c_str());
    fprintf(f, "----- --------- ------ ----- --- --- --------- --------- ------ ----- ----- ----- ----- ----- ----- ----- ---- ---------------------\n");
    for (uint64_t r = r0; r < r1; ++r) {
      const HostHMM &hm = p->hmm[h->model[r]];
      const std::string &desc = s->descs[h->seq[r]];
      fprintf(f, "%-*s %-*s %5d %-*s %-*s %5d %9.2g %6.1f %5.1f %3d %3d %9.2g %9.2g %6.1f %5.1f %5d %5d %5ld %5ld %5ld %5ld %4.2f %s\n", tnamew,
              s->names[h->seq[r]].c_str(), taccw, "-", h->tlen[r], qnamew, hm.name.c_str(), qaccw, (hm.has_acc && !hm.acc.empty()) ? hm.acc.c_str() : "-",
              h->qlen[r], h->full_evalue[r], h->full_score[r], h->full_bias[r], h->dom_idx[r], h->ndom[r], h->c_evalue[r], h->i_evalue[r], h->dom_score[r],
              h->dom_bias[r], h->hmm_from[r], h->hmm_to[r], (long)h->ali_from[r], (long)h->ali_to[r], (long)h->env_from[r], (long)h->env_to[r], h->acc[r],
              desc.empty() ? "-" : desc.c_str());
    }
    fprintf(f, "#\n# Program:         hmmsearch\n# Pipeline mode:   SEARCH\n# [ok]\n");
    if (fclose(f) != 0) throw Error(CKM_EIO, std::string("error closing ") + path);
  });
}

// ---- diagnostics -------------------------------------------------------------------------------------
extern "C" int ckm_debug_stages(ckm_ctx *ctx_, const ckm_profiles *p, const ckm_seqs *s, const uint32_t *model, const uint32_t *seq,
                                uint32_t npairs, ckm_stage_scores *out) {
  return guarded([&] {
    if (!ctx_ || !p || !s || !model || !seq || !out) throw Error(CKM_EINVAL, "NULL argument");
    Worker *ctx = &ctx_->w[0];
    ctx->plan_key.clear();                 // this entry overwrites the worker's SSV tables
    HIPCHK(hipSetDevice(ctx->device));
    const DevModel *dm = p->d_models.as<DevModel>();
    const LenEntry *lt = s->d_lentab.as<LenEntry>();
    const uint8_t *res = s->d_res.as<uint8_t>();
    const uint64_t *off = s->d_off.as<uint64_t>();
    const int32_t *dlen = s->d_len.as<int32_t>();
    memset(out, 0, sizeof(*out) * npairs);
    // SSV: one block per pair (count = 1)
    std::vector<SsvBlockWork> work(npairs); std::vector<uint32_t> ids(seq, seq + npairs);
    std::map<int, std::vector<uint32_t>> byQ;
    for (uint32_t i = 0; i < npairs; ++i) {
      if (model[i] >= p->hmm.size() || seq[i] >= s->nseq) throw Error(CKM_EINVAL, "pair index out of range");
      work[i].model = model[i]; work[i].list_start = i; work[i].count = 1; work[i].pair_start = i; byQ[p->prof[model[i]].ssvQ].push_back(i);
    }
    std::vector<SsvBlockWork> sorted; std::vector<std::pair<int, std::pair<size_t, size_t>>> groups;
    for (auto &kv : byQ) { groups.push_back({kv.first, {sorted.size(), kv.second.size()}}); for (uint32_t i : kv.second) sorted.push_back(work[i]); }
    ctx->work.ensure(npairs * sizeof(SsvBlockWork)); ctx->idx.ensure(npairs * 4); ctx->maxv.ensure(npairs * 2 + 64);
    wcopy(ctx, ctx->work.p, sorted.data(), npairs * sizeof(SsvBlockWork), hipMemcpyHostToDevice);
    wcopy(ctx, ctx->idx.p, ids.data(), npairs * 4, hipMemcpyHostToDevice);
    for (auto &g : groups)
      if (launch_ssv(g.first, (int)g.second.second, ssv_threads_for(g.first), ctx->stream, ctx->work.as<SsvBlockWork>() + g.second.first, dm, res, off, dlen,
                     ctx->idx.as<uint32_t>(), ctx->maxv.as<uint16_t>()))
        throw Error(CKM_ERANGE, "no SSV kernel instance");
    HIPCHK(hipGetLastError());
    std::vector<uint16_t> maxv(npairs);
    HIPCHK(hipMemcpyAsync(maxv.data(), ctx->maxv.p, npairs * 2, hipMemcpyDeviceToHost, ctx->stream));
    // full MSV on every pair: first with the packed kernel the search uses, then with the plain reference kernel
    std::vector<PairRec> pr(npairs);
    for (uint32_t i = 0; i < npairs; ++i) { pr[i].model = model[i]; pr[i].seq = seq[i]; pr[i].usc = 0; pr[i].filtersc = 0; }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::vector<float> uscp; std::vector<int32_t> xJp;
    run_msv_exact(ctx, p, s, pr, uscp, &xJp);
    ctx->cand.ensure(npairs * sizeof(PairRec)); ctx->fullx.ensure(npairs * 4); ctx->fullu.ensure(npairs * 4); ctx->raw.ensure(npairs * 12);
    HIPCHK(hipMemcpyAsync(ctx->cand.p, pr.data(), npairs * sizeof(PairRec), hipMemcpyHostToDevice, ctx->stream));
    launch_msv_full(ctx->stream, ctx->cand.as<PairRec>(), npairs, dm, lt, res, off, dlen, ctx->fullx.as<int32_t>(), ctx->fullu.as<float>(), p->maxMp);
    launch_bias(ctx->stream, ctx->cand.as<PairRec>(), npairs, dm, lt, res, off, dlen, ctx->raw.as<float>());
    HIPCHK(hipGetLastError());
    std::vector<int32_t> xJ(npairs); std::vector<float> usc(npairs), raw(npairs * 3);
    HIPCHK(hipMemcpyAsync(xJ.data(), ctx->fullx.p, npairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(usc.data(), ctx->fullu.p, npairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(raw.data(), ctx->raw.p, npairs * 12, hipMemcpyDeviceToHost, ctx->stream));
    // Viterbi on every pair
    std::map<int, std::vector<uint32_t>> vq;
    for (uint32_t i = 0; i < npairs; ++i) vq[p->prof[model[i]].vitQH].push_back(i);
    std::vector<uint32_t> flat; std::vector<std::pair<int, std::pair<size_t, size_t>>> vg;
    for (auto &kv : vq) { vg.push_back({kv.first, {flat.size(), kv.second.size()}}); flat.insert(flat.end(), kv.second.begin(), kv.second.end()); }
    ctx->fbidx.ensure(npairs * 4); ctx->vitx.ensure(npairs * 4); ctx->vits.ensure(npairs * 4);
    HIPCHK(hipMemcpyAsync(ctx->fbidx.p, flat.data(), npairs * 4, hipMemcpyHostToDevice, ctx->stream));
    for (auto &g : vg)
      if (launch_vit(g.first, ctx->stream, ctx->cand.as<PairRec>(), ctx->fbidx.as<uint32_t>() + g.second.first, (uint32_t)g.second.second, dm, lt, res, off, dlen,
                     ctx->vitx.as<int32_t>(), ctx->vits.as<float>(), nullptr, false))
        throw Error(CKM_ERANGE, "no Viterbi kernel instance");
    HIPCHK(hipGetLastError());
    std::vector<int32_t> vx(npairs); std::vector<float> vs(npairs);
    HIPCHK(hipMemcpyAsync(vx.data(), ctx->vitx.p, npairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(vs.data(), ctx->vits.p, npairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    // Forward parser on every pair
    FbBatch fb; uint64_t pos = 0;
    for (uint32_t i = 0; i < npairs; ++i) {
      const int L = s->len[seq[i]];
      FbWork w; memset(&w, 0, sizeof(w));
      w.model = model[i]; w.seq = seq[i]; w.i0 = 0; w.Ld = L; w.Lcfg = L; w.multihit = 1; w.full = 0; w.slot = i;
      w.xs_off = pos; pos += ((uint64_t)(L + 1) * 6 + 31) & ~(uint64_t)31;
      w.aux_off = pos; pos += ((uint64_t)(L + 1) * 3 + 31) & ~(uint64_t)31;
      fb.work.push_back(w);
    }
    ctx->ws.ensure(pos * 4 + 256);
    run_fb(ctx, p, s, fb, true, false, false, nullptr);
    EventIndex ei; ei.build(fb.events, npairs);
    for (uint32_t i = 0; i < npairs; ++i) {
      const int L = s->len[seq[i]];
      const LenEntry &le = s->lentab[L];
      ckm_stage_scores &o = out[i];
      o.ssv_maxv = maxv[i]; o.msv_xJ = xJ[i]; o.msv_sc = usc[i]; o.null_sc = le.nullsc;
      o.msvp_xJ = xJp[i]; o.msvp_sc = uscp[i];
      const float p1 = (float)L / (float)(L + 1);
      const float nullsc = (float)(log((double)raw[(size_t)i * 3]) + (double)raw[(size_t)i * 3 + 1] * kLn2);
      o.bias_sc = nullsc + (float)L * logf(p1) + logf(1.0f - p1);
      o.vit_xC = vx[i]; o.vit_sc = vs[i];
      o.fwd_xC = fb.fout[i].xC; o.fwd_nscale = fb.fout[i].nscale;
      o.fwd_sc = finish_forward(fb.fout[i].xC, le.move_m, ei.scales(i));
    }
  });
}

extern "C" int ckm_debug_envelopes(ckm_ctx *ctx_, const ckm_profiles *p, const ckm_seqs *s, const uint32_t *model, const uint32_t *seq,
                                   const int32_t *ienv, const int32_t *jenv, uint32_t n, ckm_envelope_result *out) {
  return guarded([&] {
    if (!ctx_ || !p || !s || !model || !seq || !ienv || !jenv || !out) throw Error(CKM_EINVAL, "NULL argument");
    Worker *ctx = &ctx_->w[0];
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<EnvReq> req(n); std::vector<EnvRes> res;
    for (uint32_t i = 0; i < n; ++i) {
      if (model[i] >= p->hmm.size() || seq[i] >= s->nseq || ienv[i] < 1 || jenv[i] > s->len[seq[i]] || jenv[i] < ienv[i]) throw Error(CKM_EINVAL, "bad envelope");
      req[i] = {model[i], seq[i], ienv[i], jenv[i]};
    }
    rescore_envelopes(ctx, p, s, req, res);
    for (uint32_t i = 0; i < n; ++i) {
      out[i].envsc = res[i].envsc; out[i].oasc = res[i].oasc; out[i].fwd_xC = res[i].xC; out[i].nscale = res[i].nscale; out[i].ok = res[i].ok;
      for (int x = 0; x < 20; ++x) out[i].null2[x] = res[i].null2[x];
      out[i].hmm_from = res[i].hmm_from; out[i].hmm_to = res[i].hmm_to; out[i].ali_from = res[i].ali_from; out[i].ali_to = res[i].ali_to;
    }
  });
}

extern "C" int ckm_debug_region(ckm_ctx *ctx_, const ckm_profiles *p, const ckm_seqs *s, uint32_t model, uint32_t seq, int32_t ireg, int32_t jreg,
                                float *n2sum, int32_t *segs, int32_t *nseg, int32_t cap, int32_t *env, int32_t envcap, int32_t *nenv) {
  return guarded([&] {
    if (!ctx_ || !p || !s || !n2sum || !segs || !nseg || !env || !nenv) throw Error(CKM_EINVAL, "NULL argument");
    if (model >= p->hmm.size() || seq >= s->nseq || ireg < 1 || jreg > s->len[seq] || jreg < ireg) throw Error(CKM_EINVAL, "bad region");
    Worker *ctx = &ctx_->w[0];
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<RegionReq> req{{model, seq, ireg, jreg}}; std::vector<RegionRes> res;
    run_ensembles(ctx, p, s, req, res);
    const RegionRes &r = res[0];
    for (size_t i = 0; i < r.n2sum.size(); ++i) n2sum[i] = r.n2sum[i];
    for (int t = 0; t < ENS_NSAMPLES; ++t) {
      if (r.nseg[t] > cap) throw Error(CKM_ERANGE, "segment table too small");
      nseg[t] = r.nseg[t];
      for (int d = 0; d < r.nseg[t]; ++d) { const Seg &g = r.segs[(size_t)t * r.cap + d]; int32_t *o = segs + ((size_t)t * cap + d) * 4; o[0] = g.sqfrom; o[1] = g.sqto; o[2] = g.hmmfrom; o[3] = g.hmmto; }
    }
    if ((int)r.env.size() > envcap) throw Error(CKM_ERANGE, "envelope table too small");
    *nenv = (int32_t)r.env.size();
    for (size_t e = 0; e < r.env.size(); ++e) { env[e * 4] = r.env[e].sqfrom; env[e * 4 + 1] = r.env[e].sqto; env[e * 4 + 2] = r.env[e].hmmfrom; env[e * 4 + 3] = r.env[e].hmmto; }
  });
}
