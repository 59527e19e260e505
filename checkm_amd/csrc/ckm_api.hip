// ckm_api.hip -- C ABI of libcheckm_hip.so, part 1: context, profile upload, sequence packing, hit columns and the domtblout
// writer.  See include/checkm_hip.h for the reference interfaces each entry point replaces.
//
// The host side owns every transcendental (log/exp): per-length specials, null scores, E-values.
// The device side owns every per-cell operation.  There is no CPU implementation of any kernel in
// this library: if HIP is unusable, ckm_ctx_create fails and nothing else can run.
#include "ckm_host.h"

namespace ckm {
static thread_local std::string g_err;
void set_last_error(const std::string &m) { g_err = m; }

}  // namespace ckm

std::atomic<uint64_t> g_uid{1};     // identity of every profile DB / sequence set / list ever created (pointers get reused)

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
    if (const char *e = getenv("CKM_WORKERS")) ctx->nclasses = std::max(1, std::min(NWORKERS, atoi(e)));
    ctx->nworkers = ctx->nclasses * ctx->ngroups;
    // 8 threads, or an eighth of the machine up to 16: eight contexts (one per GPU) must fit the node's cores together
    const int hw = (int)std::thread::hardware_concurrency();
    int host_threads = std::max(1, std::min(hw, std::max(8, std::min(16, hw / 8))));
    if (const char *e = getenv("CKM_HOST_THREADS")) host_threads = std::max(1, std::min(64, atoi(e)));
    size_t fre = 0, tot = 0;
    size_t budget = (size_t)8 << 30;
    // (a quarter of what is free: MarkerGeneFinder.find keeps up to three contexts per device in flight, each with its own workspace)
    if (hipMemGetInfo(&fre, &tot) == hipSuccess) budget = std::min<size_t>((size_t)96 << 30, fre / 4) / ctx->nworkers;
    if (const char *e = getenv("CKM_WS_BUDGET_MB")) budget = std::max<size_t>(16, strtoull(e, nullptr, 10)) << 20;   // tests: force several envelope batches
    const int nside = choose_side_streams(ctx->nworkers);
    for (auto &w : ctx->w) {
      w.device = device; w.id = (int)(&w - ctx->w);
      if (&w - ctx->w >= ctx->nworkers) continue;
      // non-blocking streams: nothing here may synchronise implicitly with the null stream or with another worker's streams
      HIPCHK(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
      w.nside = nside;
      for (int k = 0; k < 16; ++k) {
        if (k >= nside) { w.side[k] = w.side[k % nside]; continue; }
        HIPCHK(hipStreamCreateWithFlags(&w.side[k], hipStreamNonBlocking));
      }
      // the trace ensembles of a search get a stream of their own: a launch lasts as long as its longest region (tens of milliseconds), and
      // neither a chain stream nor the priority stream (which carries the NEXT batch's uploads) may queue behind it
      HIPCHK(hipStreamCreateWithFlags(&w.ens_stream, hipStreamNonBlocking));
      {
        // the rounds after a search's drain (envelopes of the ensembles' clustering, deferred regions: milliseconds of device work the
        // host waits for) must not queue behind ANOTHER context's SSV launches, whose backlog holds the normal-priority hardware queues
        // for the whole SSV phase (measured: 5 ms of kernels returned after 490 ms, profiles/r03t_lane_trace.txt)
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        // ONE such stream per worker (late[1..3] are the same stream): every high-priority stream is a hardware queue of its own, and with
        // four per worker a process holding four contexts oversubscribed the device's queue slots -- every idle -> busy transition then
        // waits for the scheduler's time slice (cfg2 measured after cfg3 in one process: 75.9 -> 98 ms per step)
        HIPCHK(hipStreamCreateWithPriority(&w.late[0], hipStreamNonBlocking, hi));
        for (int k = 1; k < 4; ++k) w.late[k] = w.late[0];
      }
      for (auto &e : w.ev) HIPCHK(hipEventCreate(&e));
      for (auto &e : w.cev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      for (auto &e : w.cls_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      for (auto &e : w.grp_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      for (auto &e : w.ens_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      memset(&w.stats, 0, sizeof(w.stats));
      w.ws_budget = budget;
      // the float workspace is an address reservation with 1 GB chunks mapped in as it grows (DevBuf::grow_mapped): find() reserves the
      // largest batch's size before its first search, when the device is idle, and mapping then costs 0.2 ms per GB against hipMalloc's
      // 30 -- measured on the first full-size step of a process: 0.9 s over a steady step instead of 3.4 s (profiles/r04v_*; the whole
      // GPU suite ran in this mode).  CKM_WS_VMM=0: one hipMalloc per growth, as rounds 1-3 had it.
      if (!(getenv("CKM_WS_VMM") && atoi(getenv("CKM_WS_VMM")) == 0)) { w.ws.vmm = true; w.ws.va_bytes = budget + ((size_t)8 << 30); }
      if (const char *e = getenv("CKM_WS_PER_CELL")) w.caps.ws_per_cell = (float)atof(e);  // first workspace size of the device-driven cascade (bytes per expected cell, ckm_search.hip)
      w.pool.reset(new HostPool(host_threads));
    }
    *out = ctx.release();
  });
}

extern "C" int ckm_ctx_reserve(ckm_ctx *ctx, uint64_t pairs, double cells) {
  return guarded([&] {
    if (!ctx) throw Error(CKM_EINVAL, "ctx is NULL");
    ctx->settle();
    Worker &w = ctx->w[0];
    // the estimate ckm_search itself makes (ckm_search.hip: cascade_dev), for one lane
    const uint64_t est = (uint64_t)(1.05 * cells * (double)w.caps.ws_per_cell + (double)pairs * 24.0) + ((uint64_t)256 << 20);
    const size_t want = (size_t)std::min<uint64_t>(std::max<uint64_t>(est, (uint64_t)1 << 30), (uint64_t)w.ws_budget);
    if (w.ws.cap >= want) return;
    const int device = ctx->device;
    std::lock_guard<std::mutex> lock(ctx->reserve_mutex);
    ctx->reserve_thread = std::thread([&w, want, device] {
      if (hipSetDevice(device) != hipSuccess) return;
      try { w.ws.ensure(want); } catch (...) { /* whatever went wrong (out of memory, a driver error): the search will ask again and report the failure itself */ }
    });
  });
}

extern "C" void ckm_ctx_destroy(ckm_ctx *ctx) {
  if (!ctx) return;
  ctx->settle();
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  for (auto &w : ctx->w) {
    if (!w.stream) continue;
    for (auto &e : w.ev) (void)hipEventDestroy(e);
    for (auto &e : w.cev) if (e) (void)hipEventDestroy(e);
    for (auto &e : w.cls_ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : w.grp_ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : w.ens_ev) if (e) (void)hipEventDestroy(e);
    for (int k = 0; k < w.nside; ++k) (void)hipStreamDestroy(w.side[k]);
    if (w.late[0]) (void)hipStreamDestroy(w.late[0]);
    if (w.ens_stream) (void)hipStreamDestroy(w.ens_stream);
    (void)hipStreamDestroy(w.stream);
  }
  const int device = ctx->device;
  delete ctx;
  dev_cache_trim(device);                 // (blocks the context's buffers left in the cache)
}

// ---- profiles -----------------------------------------------------------------------------------
// All device tables of a database live in ONE allocation, filled by ONE copy: a 2000-model database has 16 000 tables, and a
// hipMalloc + synchronous hipMemcpy for each of them used to cost more than the arithmetic of configuring the profiles.
namespace {
struct Arena {
  std::vector<uint8_t> host;
  template <class T> size_t add(const std::vector<T> &v) {
    const size_t off = (host.size() + 255) & ~(size_t)255;
    host.resize(off + std::max<size_t>(16, v.size() * sizeof(T)));
    if (!v.empty()) memcpy(host.data() + off, v.data(), v.size() * sizeof(T));
    return off;
  }
};
}  // namespace

extern "C" int ckm_profiles_load(ckm_ctx *ctx, const char *hmm_path, ckm_profiles **out) {
  return guarded([&] {
    if (!ctx || !hmm_path || !out) throw Error(CKM_EINVAL, "NULL argument");
    *out = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    std::unique_ptr<ckm_profiles> p(new ckm_profiles());
    p->ctx = ctx;
    p->hmm = read_hmm_file(hmm_path);
    const size_t n = p->hmm.size();
    p->prof.resize(n);
    p->too_long.assign(n, 0);
    {   // configuration is per model (logs, exps, the three striped layouts): host threads
      std::vector<std::unique_ptr<Error>> errs(n);
      auto one = [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
          try { p->prof[i] = configure_profile(p->hmm[i]); }
          catch (const Error &e) {
            // A model beyond the instantiated kernel classes (M > 4096) does not take the database down: it keeps its place and its
            // header, and only a search that actually selects it is refused, by name (ckm_search).  CheckM's marker sets hold no such
            // model; a full Pfam / TIGRFAM file holds a handful.
            if (e.code == CKM_ERANGE) { p->too_long[i] = 1; p->prof[i] = HostProfile(); p->prof[i].M = p->hmm[i].M; p->prof[i].ssvQ = p->prof[i].fbQ = p->prof[i].vitQH = 1; }
            else errs[i].reset(new Error(e));
          }
        }
      };
      if (ctx->w[0].pool && n > 8) ctx->w[0].pool->run(n, 4, one); else one(0, n);
      for (size_t i = 0; i < n; ++i) if (errs[i]) throw Error(*errs[i]);
    }
    Arena ar;
    struct Off { size_t ssv, ssvh, rbv, vit_e, vit_t, rf, ftr, v16e, v16t, ssv8; };
    std::vector<Off> off(n);
    for (size_t i = 0; i < n; ++i) {
      const HostProfile &hp = p->prof[i];
      off[i] = Off{ar.add(hp.ssv_tbl), ar.add(hp.ssv_tbl_h), ar.add(hp.rbv), ar.add(hp.vit_e), ar.add(hp.vit_t), ar.add(hp.rf), ar.add(hp.ftr), ar.add(hp.vit16_e), ar.add(hp.vit16_t), ar.add(hp.ssv8_tbl_h)};
    }
    std::unique_ptr<DevBuf> tables(new DevBuf());
    tables->ensure(std::max<size_t>(256, ar.host.size()));
    HIPCHK(hipMemcpy(tables->p, ar.host.data(), ar.host.size(), hipMemcpyHostToDevice));
    const uint8_t *base = tables->as<uint8_t>();
    p->tables.push_back(std::move(tables));
    for (size_t i = 0; i < n; ++i) {
      const HostHMM &h = p->hmm[i];
      const HostProfile &hp = p->prof[i];
      DevModel d;
      memset(&d, 0, sizeof(d));
      d.M = hp.M; d.ssvQ = hp.ssvQ; d.fbQ = hp.fbQ; d.vitQH = hp.vitQH;
      d.fb_cls = fb_class_id(hp.fbQ); d.vitx_cls = vit_class_id(hp.vitQH); d.vit16Q = hp.vit16Q;
      d.vit_cls = hp.vit16Q ? vit16_class_id(hp.vit16Q) : d.vitx_cls;      // short models take the 16-lane FAST kernel
      if (!p->too_long[i] && (d.fb_cls < 0 || d.vit_cls < 0)) throw Error(CKM_ERANGE, "model " + h.name + " is longer than the instantiated kernel classes (DESIGN.md limits)");
      d.base_b = hp.base_b; d.bias_b = hp.bias_b; d.tbm_b = hp.tbm_b; d.tec_b = hp.tec_b; d.scale_b = hp.scale_b;
      d.scale_w = hp.scale_w; d.base_w = hp.base_w; d.wE_loop = hp.wE_loop; d.wE_move = hp.wE_move;
      d.fE_loop = hp.fE_loop; d.fE_move = hp.fE_move;
      d.bt00 = hp.bt00; d.bt01 = hp.bt01; d.bt10 = hp.bt10; d.bt11 = hp.bt11; d.bpi0 = hp.bpi0; d.bpi1 = hp.bpi1;
      for (int x = 0; x < NROWS; ++x) d.beo1[x] = hp.beo1[x];
      d.thr_msv_f1 = hp.thr_msv_f1; d.thr_msv_f1_nat = hp.thr_msv_f1_nat; d.thr_msv_f2 = hp.thr_msv_f2; d.thr_vit_f2 = hp.thr_vit_f2; d.thr_fwd_f3 = hp.thr_fwd_f3;
      d.ssv_tbl = reinterpret_cast<const int16_t *>(base + off[i].ssv); d.ssv_tbl_h = reinterpret_cast<const uint16_t *>(base + off[i].ssvh);
      d.rbv = base + off[i].rbv; d.vit_e = reinterpret_cast<const uint32_t *>(base + off[i].vit_e);
      d.vit_t = reinterpret_cast<const uint32_t *>(base + off[i].vit_t); d.rf = reinterpret_cast<const float *>(base + off[i].rf);
      d.ftr = reinterpret_cast<const float *>(base + off[i].ftr);
      d.ssv8_tbl_h = hp.ssv8Q ? reinterpret_cast<const uint16_t *>(base + off[i].ssv8) : nullptr;
      d.vit16_e = reinterpret_cast<const uint32_t *>(base + off[i].v16e); d.vit16_t = reinterpret_cast<const uint32_t *>(base + off[i].v16t);
      p->dm.push_back(d);
      if (!p->too_long[i]) p->maxMp = std::max(p->maxMp, hp.fbQ * NL);
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
  o->searchable = p->too_long[(size_t)i] ? 0 : 1;
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

static SeqColumns seq_columns(ckm_seqs *s) {
  return SeqColumns{&s->bin_off, &s->seq_bin, &s->order, &s->order_off, &s->len, &s->off, &s->bin_res, &s->dsq, &s->names, &s->descs, &s->total_res, &s->maxL};
}

// shared tail of the two constructors: s->len / s->off / s->dsq / names are filled; build the order, tables and upload
static void finish_seqs(ckm_seqs *s) {
  const uint32_t nseq = s->nseq;
  build_seq_order(ingest_threads(), seq_columns(s));
  build_lentab(s);
  // uploads on one of the context's high-priority streams: a plain hipMemcpy travels on the null stream, whose hardware queue it shares
  // with whatever streams were mapped onto it -- behind another context's SSV backlog the 40 ms ingest of a batch took 420 ms
  // (profiles/r03t_lane_trace.txt)
  hipStream_t up = s->ctx->w[0].late[3];
  trace_pt(&s->ctx->w[0], "seqs: tables built");
  s->d_res.ensure(s->dsq.size() + 16); s->d_off.ensure((size_t)nseq * 8 + 16); s->d_len.ensure((size_t)nseq * 4 + 16);       // (+16: the copy kernel moves 16-byte words)
  s->d_order.ensure(s->order.size() * 4 + 16); s->d_lentab.ensure(s->lentab.size() * sizeof(LenEntry) + 16);
  trace_pt(&s->ctx->w[0], "seqs: device buffers allocated");
  // ... and by a KERNEL that reads a page-locked staging buffer of the context over the bus: the runtime's copy path (pageable or
  // pinned source alike) was seen to wait until the device had drained -- one upload in three took 370 ms instead of 3 -- while kernels
  // on the high-priority streams start within a millisecond
  {
    std::lock_guard<std::mutex> lock(s->ctx->upload_mutex);
    struct Part { DevBuf *d; const void *src; size_t bytes; };
    const Part parts[5] = {{&s->d_res, s->dsq.data(), s->dsq.size()}, {&s->d_off, s->off.data(), (size_t)nseq * 8}, {&s->d_len, s->len.data(), (size_t)nseq * 4},
                           {&s->d_order, s->order.data(), s->order.size() * 4}, {&s->d_lentab, s->lentab.data(), s->lentab.size() * sizeof(LenEntry)}};
    size_t tot = 0;
    for (auto &pt : parts) tot += (pt.bytes + 255) & ~(size_t)255;
    s->ctx->upload.begin(tot);
    for (auto &pt : parts) s->ctx->upload.put(up, pt.d->p, pt.src, pt.bytes);
    HIPCHK(hipStreamSynchronize(up));
  }
  trace_pt(&s->ctx->w[0], "seqs: uploaded");
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
    trace_pt(&ctx->w[0], "seqs: begin");
    if (nbins == 0) throw Error(CKM_EINVAL, "no bins");
    HIPCHK(hipSetDevice(ctx->device));
    std::unique_ptr<ckm_seqs> s(new ckm_seqs());
    s->ctx = ctx; s->nbins = nbins; s->uid = g_uid++;
    s->bin_off.assign(nbins + 1, 0);
    // the files are read and digitised a file per thread, then laid end to end a bin per thread (fasta_ingest.cpp)
    std::vector<FastaBin> bins = read_fasta_bins(paths, nbins, ingest_threads());
    trace_pt(&ctx->w[0], "seqs: files read");
    merge_fasta_bins(bins, ingest_threads(), seq_columns(s.get()));
    s->nseq = (uint32_t)s->names.size();
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

// HMMER's amino-acid background (Swiss-Prot 50.8 composition; the same table host_profile.cpp configures profiles with)
static const float kBgAmino[ckm::K] = {0.0787945f, 0.0151600f, 0.0535222f, 0.0668298f, 0.0397062f, 0.0695071f, 0.0229198f, 0.0590092f, 0.0594422f, 0.0963728f,
                                       0.0237718f, 0.0414386f, 0.0482904f, 0.0395639f, 0.0540978f, 0.0683364f, 0.0540687f, 0.0673417f, 0.0114135f, 0.0304133f};

// ---- hmmsearch-style report with the domain alignments (bKeepAlignment) ------------------------------------------------
// What `hmmsearch -o <hmmerOut>` leaves when CheckM keeps alignments (checkm/markerGeneFinder.py:138-142, `--ali`): per query model the
// score table of the reported sequences, and per domain the alignment of the envelope's optimal-accuracy path.  Nothing in CheckM
// reads this file back; it is for the user.  The paths come from the same envelope kernels that produced the row's coordinates
// (rescore_envelopes with traces), the consensus line from the model's match emissions (upper case above 0.5, HMMER's rule for amino
// acids), the middle line marks identities by the consensus letter and positive log-odds by '+', the PP line under each alignment is the
// posterior probability of every aligned residue in the state that emits it, in hmmsearch's code (0-9: tenths, rounded; '*': >= 0.95;
// '.': a deleted node), read from the posterior matrix of the same envelope computation.  NOT reproduced: the `exp` column (expected
// number of domains, a by-product of the domain definition that the rows do not carry) and hmmsearch's line wrapping (the file is
// written as with --notextw).
extern "C" int ckm_hits_write_alignments(ckm_ctx *ctx_, const ckm_hits *h, const ckm_profiles *p, const ckm_seqs *s, uint32_t bin, const char *path) {
  return guarded([&] {
    if (!ctx_ || !h || !p || !s || !path) throw Error(CKM_EINVAL, "NULL argument");
    if (bin >= h->nbins) throw Error(CKM_EINVAL, "bin out of range");
    ctx_->settle();
    Worker *ctx = &ctx_->w[0];
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t r0 = h->bin_row_off[bin], r1 = h->bin_row_off[bin + 1];
    std::vector<EnvReq> req;
    for (uint64_t r = r0; r < r1; ++r) req.push_back({h->model[r], h->seq[r], h->env_from[r], h->env_to[r]});
    std::vector<EnvRes> res; std::vector<std::vector<int32_t>> paths; std::vector<std::vector<float>> pps;
    if (!req.empty()) rescore_envelopes(ctx, p, s, req, res, &paths, &pps);
    FILE *f = fopen(path, "w");
    if (!f) throw Error(CKM_EIO, std::string("cannot write ") + path);
    fprintf(f, "# hmmsearch-style report written by libcheckm_hip (MI355X scan; options -E 0.1 --domE 0.1 --notextw).\n"
               "# Alignments: optimal-accuracy path of each domain's envelope with its posterior-probability (PP) line; the `exp` column of hmmsearch is not produced.\n");
    static const char kSym[] = "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~";
    uint64_t r = r0;
    while (r < r1) {
      const uint32_t m = h->model[r];
      uint64_t re = r;
      while (re < r1 && h->model[re] == m) ++re;
      const HostHMM &hm = p->hmm[m];
      // consensus line of the model
      std::string cons((size_t)hm.M + 1, 'x');
      for (int k = 1; k <= hm.M; ++k) {
        int best = 0;
        for (int x = 1; x < K; ++x) if (hm.mat[(size_t)k * K + x] > hm.mat[(size_t)k * K + best]) best = x;
        cons[k] = hm.mat[(size_t)k * K + best] > 0.5f ? kSym[best] : (char)tolower(kSym[best]);
      }
      fprintf(f, "\nQuery:       %s  [M=%d]\n", hm.name.c_str(), hm.M);
      if (hm.has_acc && !hm.acc.empty()) fprintf(f, "Accession:   %s\n", hm.acc.c_str());
      if (hm.has_desc && !hm.desc.empty()) fprintf(f, "Description: %s\n", hm.desc.c_str());
      fprintf(f, "Scores for complete sequences (score includes all domains):\n   --- full sequence ---   --- best 1 domain ---    -#dom-\n"
                 "    E-value  score  bias    E-value  score  bias    exp  N  Sequence Description\n"
                 "    ------- ------ -----    ------- ------ -----   ---- --  -------- -----------\n");
      for (uint64_t a = r; a < re; ++a) {
        if (h->dom_idx[a] != 1) continue;
        uint64_t best = a;                                   // best-scoring domain of this sequence
        for (uint64_t b = a; b < re && h->seq[b] == h->seq[a]; ++b) if (h->dom_score[b] > h->dom_score[best]) best = b;
        fprintf(f, "  %9.2g %6.1f %5.1f  %9.2g %6.1f %5.1f   %4s %2d  %s  %s\n", h->full_evalue[a], h->full_score[a], h->full_bias[a], h->i_evalue[best],
                h->dom_score[best], h->dom_bias[best], "-", h->ndom[a], s->names[h->seq[a]].c_str(), s->descs[h->seq[a]].c_str());
      }
      fprintf(f, "\n\nDomain annotation for each sequence (and alignments):\n");
      for (uint64_t a = r; a < re; ++a) {
        const uint32_t sq = h->seq[a];
        if (h->dom_idx[a] == 1) {
          fprintf(f, ">> %s  %s\n   #    score  bias  c-Evalue  i-Evalue hmmfrom  hmm to    alifrom  ali to    envfrom  env to     acc\n"
                     " ---   ------ ----- --------- --------- ------- -------    ------- -------    ------- -------    ----\n", s->names[sq].c_str(), s->descs[sq].c_str());
          for (uint64_t b = a; b < re && h->seq[b] == sq; ++b)
            fprintf(f, " %3d %c %6.1f %5.1f %9.2g %9.2g %7d %7d %c%c %7d %7d %c%c %7d %7d %c%c %4.2f\n", h->dom_idx[b], '!', h->dom_score[b], h->dom_bias[b], h->c_evalue[b],
                    h->i_evalue[b], h->hmm_from[b], h->hmm_to[b], h->hmm_from[b] == 1 ? '[' : '.', h->hmm_to[b] == hm.M ? ']' : '.', h->ali_from[b], h->ali_to[b],
                    h->ali_from[b] == 1 ? '[' : '.', h->ali_to[b] == h->tlen[b] ? ']' : '.', h->env_from[b], h->env_to[b], h->env_from[b] == 1 ? '[' : '.',
                    h->env_to[b] == h->tlen[b] ? ']' : '.', h->acc[b]);
          fprintf(f, "\n  Alignments for each domain:\n");
        }
        fprintf(f, "  == domain %d  score: %.1f bits;  conditional E-value: %.2g\n", h->dom_idx[a], h->dom_score[a], h->c_evalue[a]);
        const std::vector<int32_t> &path = paths[a - r0];
        const uint8_t *dsq = s->dsq.data() + s->off[sq];
        const std::vector<float> &ppv = pps[a - r0];
        auto pp_code = [&](int pr) {                           // p7_alidisplay_EncodePostProb (hmmer/src/p7_alidisplay.c)
          const float pv = (pr >= 0 && (size_t)pr < ppv.size()) ? ppv[(size_t)pr] : 0.f;
          return (pv + 0.05 >= 1.0) ? '*' : (char)((int)((pv + 0.05) * 10.0) + '0');
        };
        std::string ml, mid, tl, ppl;
        if (res[a - r0].ok && (int)path.size() >= hm.M) {
          const int base = h->env_from[a] - 1;                // path entries are 1-based within the envelope
          int prev_res = 0;
          for (int k = h->hmm_from[a]; k <= h->hmm_to[a]; ++k) {
            const int pr = path[(size_t)k - 1];
            if (pr == 0) { ml += cons[k]; mid += ' '; tl += '-'; ppl += '.'; continue; }
            const int i = base + pr;                           // sequence coordinate, 1-based
            if (prev_res) for (int j = prev_res + 1; j < i; ++j) { ml += '.'; mid += ' '; tl += (char)tolower(kSym[dsq[j - 1]]); ppl += pp_code(j - base); }
            const int x = dsq[i - 1];
            ml += cons[k]; tl += kSym[x]; ppl += pp_code(pr);
            if (x < K && toupper(cons[k]) == kSym[x]) mid += cons[k];
            else if (x < K && hm.mat[(size_t)k * K + x] > kBgAmino[x]) mid += '+';
            else mid += ' ';
            prev_res = i;
          }
        } else ml = mid = tl = ppl = "(no alignment: the envelope could not be rescored)";
        const int w = (int)std::max(hm.name.size(), s->names[sq].size());
        fprintf(f, "  %*s %7d %s %-7d\n  %*s %7s %s\n  %*s %7d %s %-7d\n  %*s %7s %s PP\n\n", w, hm.name.c_str(), h->hmm_from[a], ml.c_str(), h->hmm_to[a], w, "", "", mid.c_str(), w,
                s->names[sq].c_str(), h->ali_from[a], tl.c_str(), h->ali_to[a], w, "", "", ppl.c_str());
      }
      r = re;
    }
    fprintf(f, "\n//\n[ok]\n");
    if (fclose(f) != 0) throw Error(CKM_EIO, std::string("error closing ") + path);
  });
}

// ---- alignment of marker genes to their models ---------------------------------------------------------
extern "C" int ckm_align(ckm_ctx *ctx_, const ckm_profiles *p, const ckm_seqs *s, const uint32_t *model, const uint32_t *seq, uint32_t n,
                         const uint64_t *out_off, int32_t *node_residue) {
  return guarded([&] {
    if (!ctx_ || !p || !s || (n && (!model || !seq || !out_off || !node_residue))) throw Error(CKM_EINVAL, "NULL argument");
    ctx_->settle();
    Worker *ctx = &ctx_->w[0];
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<EnvReq> req; std::vector<uint32_t> which;
    for (uint32_t j = 0; j < n; ++j) {
      if (model[j] >= p->hmm.size() || seq[j] >= s->nseq) throw Error(CKM_EINVAL, "pair index out of range");
      if (p->too_long[model[j]]) throw Error(CKM_ERANGE, "model " + p->hmm[model[j]].name + " is longer than the 4096 nodes the kernels are instantiated for");
      const uint64_t M = (uint64_t)p->hmm[model[j]].M;
      if (out_off[j + 1] - out_off[j] != M) throw Error(CKM_EINVAL, "out_off does not match the model lengths");
      std::fill(node_residue + out_off[j], node_residue + out_off[j + 1], 0);
      if (s->len[seq[j]] < 1) continue;                 // nothing to align: every node stays unmatched
      req.push_back({model[j], seq[j], 1, s->len[seq[j]]}); which.push_back(j);
    }
    std::vector<EnvRes> res; std::vector<std::vector<int32_t>> paths;
    rescore_envelopes(ctx, p, s, req, res, &paths);
    for (size_t k = 0; k < req.size(); ++k) if (res[k].ok) std::copy(paths[k].begin(), paths[k].end(), node_residue + out_off[which[k]]);
  });
}
