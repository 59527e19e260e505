// ckm_stages.hip -- host drivers of the stages behind the SSV filter: batches of Forward / Backward / optimal-accuracy work,
// envelope rescoring, the exact multi-hit MSV and the trace ensembles of multi-domain regions (host glue between the gfx950
// kernels; the per-cell work is in kernels_*.hip).
#include "ckm_host.h"

namespace ckm {

void pool_run(Worker *w, size_t n, size_t chunk, const std::function<void(size_t, size_t)> &f) { if (w->pool) w->pool->run(n, chunk, f); else if (n) f(0, n); }

// blocking copy on the worker's own stream (a plain hipMemcpy would wait for every blocking stream of the device)
// Through the worker's own page-locked staging buffer: a copy to or from pageable memory has the runtime pin or stage that memory per call
// (measured in a process that holds a few hundred device allocations: 1.5 ms per small copy, 7.6 ms for the four of a late round).
void wcopy(Worker *w, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
  if (!bytes) return;
  std::lock_guard<std::mutex> lock(w->wstage_mutex);
  w->wstage.ensure(bytes);
  // in the rounds after a search's drain, on the priority stream like the kernels of those rounds: the hardware queue underneath the
  // worker's normal stream is shared with other contexts' streams and held by their queued launches for a whole SSV phase (measured:
  // the four result copies of a 12 ms round returned 190 ms later, every search of one lane, profiles/r04p_lane_trace.txt)
  hipStream_t st = w->late_round ? w->late[0] : w->stream;
  if (kind == hipMemcpyDeviceToHost) {
    HIPCHK(hipMemcpyAsync(w->wstage.p, src, bytes, kind, st));
    HIPCHK(hipStreamSynchronize(st));
    memcpy(dst, w->wstage.p, bytes);
  } else {
    memcpy(w->wstage.p, src, bytes);
    HIPCHK(hipMemcpyAsync(dst, w->wstage.p, bytes, kind, st));
    HIPCHK(hipStreamSynchronize(st));
  }
}

typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
__global__ void upload_kernel(u32x4s *__restrict__ dst, const u32x4s *__restrict__ src, size_t n16, uint32_t tail) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
  if (blockIdx.x == 0 && threadIdx.x < tail) reinterpret_cast<uint8_t *>(dst + n16)[threadIdx.x] = reinterpret_cast<const uint8_t *>(src + n16)[threadIdx.x];
}
void launch_upload(hipStream_t st, void *dst, const void *src_pinned, size_t bytes) {
  if (!bytes) return;
  const size_t n16 = bytes / 16;
  upload_kernel<<<(unsigned)std::max<size_t>(1, std::min<size_t>(1024, (n16 + 255) / 256)), 256, 0, st>>>(
      reinterpret_cast<u32x4s *>(dst), reinterpret_cast<const u32x4s *>(src_pinned), n16, (uint32_t)(bytes % 16));
  HIPCHK(hipGetLastError());
}

static const bool g_trace_on = getenv("CKM_TRACE") != nullptr;
static double g_trace_origin = 0;
void trace_begin() { g_trace_origin = now_ms(); }
static const bool g_trace_abs = g_trace_on && atoi(getenv("CKM_TRACE")) >= 2;    // CKM_TRACE=2: the monotonic clock itself (ms, as Python's time.monotonic()) + the worker's address
void trace_pt(const Worker *w, const char *label) {
  if (g_trace_abs) fprintf(stderr, "ckm-trace %p %12.3f %s\n", (const void *)w, now_ms(), label);
  else if (g_trace_on) fprintf(stderr, "ckm-trace w%d %8.3f %s\n", w->id, now_ms() - g_trace_origin, label);
}

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

float bits(float sc, float nullsc) { return (float)((double)(sc - nullsc) / kLn2); }

// host-side completion of a Forward score from the device's scaled xC and its rescale events
float finish_forward(float xC, float move, const std::vector<float> &scales) {
  float totscale = 0.f;
  for (float sc : scales) totscale = (float)((double)totscale + log((double)sc));
  return (float)((double)totscale + log((double)(xC * move)));
}

// Side streams a worker spreads the register classes of one stage over.  All workers together must stay within the
// hardware queues of the device (GPU_MAX_HW_QUEUES = 16): streams that share a queue run their kernels one behind the
// other, and a short launch of one worker then waits behind a long one of another (measured: the Forward parser of the
// short class, 1 ms of work, finished 12 ms late behind the envelope kernels of the long class).  So a worker creates
// main stream + NS side streams with nworkers * (1 + NS) <= 15 and aliases the rest.
static int g_side_streams = 8;
int side_streams() { return g_side_streams; }
int choose_side_streams(int nworkers) {
  int n = std::max(1, std::min(14, 15 / std::max(1, nworkers) - 1));
  g_side_streams = n;
  return n;
}

// The SSV LAUNCH CLASS of a model: its 16-lane register class (1..64), 100 + its 8-lane class for models of up to 512 nodes,
// kSsvNone (65) when no SSV instance holds it.
int ssv_class(const HostProfile &hp) { return hp.ssv8Q ? 100 + hp.ssv8Q : hp.ssvQ; }
uint32_t ssv_per_block(int cls) { return (uint32_t)ssv_threads_for(cls) / 64u * (cls >= 100 ? 8u : 4u) * 4u; }      // sequences a workgroup takes: four rounds of its wavefronts

int ssv_threads_for(int Q) {
  if (Q >= 100) Q -= 100;
  const size_t lds = (size_t)NROWS * ((Q + 3) / 4) * 256;
  // (other limits -- 16 / 32 KB, 8 / 16 KB -- were measured in round 3, profiles/r03m_ssv_threads.txt: no difference beyond noise)
  if (lds <= 40 * 1024) return 256;
  if (lds <= 80 * 1024 || Q > 40) return 512;   // kernels with Q > 40 are compiled for <= 512 threads (256 VGPRs)
  return 1024;
}

// Work-groups a launch of the persistent Forward/Backward kernels starts for n queued items: one wavefront each, at most what the
// device holds at once (256 CUs x a few wavefronts per SIMD); the wavefronts share the queue by striding.
uint32_t fb_grid(size_t n) { return (uint32_t)std::min<size_t>(n, 8192); }

void run_fb(Worker *ctx, const ckm_profiles *p, const ckm_seqs *s, FbBatch &b, bool do_fwd, bool do_bwd, bool do_oa,
            const std::vector<uint32_t> *subset /* indices into b.work, or null = all */, float *ws_other) {
  const size_t n = b.work.size();
  if (!n) return;
  trace_pt(ctx, "  fb begin");
  const bool late = ctx->late_round;          // (set by the device-driven cascade for the rounds after its drain: high-priority streams)
  hipStream_t su = late ? ctx->late[3] : ctx->stream;       // uploads: staged + copied by a kernel in the late rounds (see Stager), the runtime's copies otherwise
  ctx->fbwork.ensure(n * sizeof(FbWork));
  if (!late) HIPCHK(hipMemcpyAsync(ctx->fbwork.p, b.work.data(), n * sizeof(FbWork), hipMemcpyHostToDevice, su));
  // one queue per canonical Q (register class), longest items first; a wavefront reloads its LDS image when the model changes
  std::map<int, std::vector<uint32_t>> byQ;
  auto add = [&](uint32_t i) { byQ[p->prof[b.work[i].model].fbQ].push_back(i); };
  if (subset) for (uint32_t i : *subset) add(i); else for (uint32_t i = 0; i < n; ++i) add(i);
  struct Group { int Q; size_t first, count; };
  std::vector<uint32_t> items; std::vector<Group> groups;
  for (auto it = byQ.rbegin(); it != byQ.rend(); ++it) {      // heaviest register class first: its chain is the longest
    std::vector<uint32_t> &v = it->second;
    std::stable_sort(v.begin(), v.end(), [&](uint32_t x, uint32_t y) {
      if (b.work[x].Ld != b.work[y].Ld) return b.work[x].Ld > b.work[y].Ld;
      return b.work[x].model < b.work[y].model; });
    groups.push_back({it->first, items.size(), v.size()});
    items.insert(items.end(), v.begin(), v.end());
  }
  std::vector<uint32_t> qc(groups.size(), 0u);           // queue lengths
  for (size_t g = 0; g < groups.size(); ++g) qc[g] = (uint32_t)groups[g].count;
  ctx->fbidx.ensure(items.size() * 4 + 16); ctx->fbmodel.ensure(qc.size() * 4 + 16);
  if (late) {
    ctx->stager.begin(n * sizeof(FbWork) + items.size() * 4 + qc.size() * 4 + 1024);
    ctx->stager.put(su, ctx->fbwork.p, b.work.data(), n * sizeof(FbWork));
    ctx->stager.put(su, ctx->fbidx.p, items.data(), items.size() * 4);
    ctx->stager.put(su, ctx->fbmodel.p, qc.data(), qc.size() * 4);
  } else {
    HIPCHK(hipMemcpyAsync(ctx->fbidx.p, items.data(), items.size() * 4, hipMemcpyHostToDevice, su));
    HIPCHK(hipMemcpyAsync(ctx->fbmodel.p, qc.data(), qc.size() * 4, hipMemcpyHostToDevice, su));
  }
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
  if (do_fwd) HIPCHK(hipMemsetAsync(ctx->counters.p, 0, 64, su));
  HIPCHK(hipStreamSynchronize(su));
  trace_pt(ctx, "  fb tables uploaded");
  // every register class runs its stages in order on its own stream; classes overlap each other
  size_t gi = 0;
  for (size_t g = 0; g < groups.size(); ++g) {
    const Group &gr = groups[g];
    hipStream_t st = late ? ctx->late[gi++ % 4] : ctx->side[gi++ % side_streams()];
    uint32_t *qcd = ctx->fbmodel.as<uint32_t>() + g;
    const uint32_t *lst = ctx->fbidx.as<uint32_t>() + gr.first;
    const uint32_t nb = fb_grid(gr.count);
    if (do_fwd && launch_fwd(gr.Q, nb, st, WorkQueue{lst, qcd, (uint32_t)gr.count}, ctx->fbwork.as<FbWork>(), dm, lt, res, off, ws, ctx->fout.as<FwdOut>(),
                             ctx->events.as<ScaleEvent>(), ctx->counters.as<uint32_t>(), cap_events, nullptr))
      throw Error(CKM_ERANGE, "no Forward kernel instance for this model length");
    if (do_bwd && launch_bwd(gr.Q, nb, st, WorkQueue{lst, qcd, (uint32_t)gr.count}, ctx->fbwork.as<FbWork>(), dm, lt, res, off, ws, ctx->fout.as<FwdOut>(), ctx->rerr.as<int32_t>()))
      throw Error(CKM_ERANGE, "no Backward kernel instance for this model length");
    if (do_oa && launch_oa(gr.Q, nb, st, WorkQueue{lst, qcd, (uint32_t)gr.count}, ctx->fbwork.as<FbWork>(), dm, ws, ctx->rerr.as<int32_t>(), ctx->fout.as<FwdOut>(), ctx->envout.as<EnvOut>()))
      throw Error(CKM_ERANGE, "no OA kernel instance for this model length");
  }
  HIPCHK(hipGetLastError());
  trace_pt(ctx, "  fb launched");
  if (late) { for (auto &st : ctx->late) HIPCHK(hipStreamSynchronize(st)); }
  else for (auto &st : ctx->side) HIPCHK(hipStreamSynchronize(st));
  trace_pt(ctx, "  fb kernels done");
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

// posterior rows in place over the Forward rows (3 arrays per row; kEnvInplaceDefault = false at compile time restores a matrix of their own, 5 per row)
bool env_inplace() { return kEnvInplaceDefault; }

size_t env_floats(int Mp, int Ld, uint64_t &xs, uint64_t &aux, uint64_t &mf, uint64_t &mb, uint64_t base, bool inplace) {
  auto al = [](uint64_t v) { return (v + 31) & ~(uint64_t)31; };
  uint64_t pos = al(base);
  xs = pos; pos = al(pos + (uint64_t)(Ld + 1) * 6);
  aux = pos; pos = al(al(pos + (uint64_t)(Ld + 1) * 3) + (uint64_t)(Ld + 1) * 5);
  mf = pos; pos = al(pos + (uint64_t)(Ld + 1) * 3 * Mp);
  if (inplace) mb = mf;                                       // posterior rows (M and I) overwrite the Forward rows they come from, OA rows overwrite them in turn
  else { mb = pos; pos = al(pos + (uint64_t)(Ld + 1) * 2 * Mp); }
  return pos;
}

// Rescore envelopes on the device; returns one Domain per envelope (ok flag via envsc NaN on range error)

void rescore_envelopes(Worker *ctx, const ckm_profiles *p, const ckm_seqs *s, const std::vector<EnvReq> &req, std::vector<EnvRes> &out,
                       std::vector<std::vector<int32_t>> *paths, std::vector<std::vector<float>> *pps) {
  // pps (with paths): the posterior probability of every residue on the OA path, [0..Ld] per envelope -- these items keep their posterior rows
  // in a matrix of their own, so that the rows outlive the OA fill (the scan's items overwrite them in place)
  out.resize(req.size());
  if (!paths) pps = nullptr;
  if (paths) paths->assign(req.size(), {});
  if (pps) pps->assign(req.size(), {});
  size_t done = 0;
  const uint64_t budget_floats = ctx->ws_budget / 4;
  while (done < req.size()) {
    FbBatch b; uint64_t pos = 0, path_total = 0; size_t j = done;
    for (; j < req.size(); ++j) {
      const EnvReq &r = req[j];
      const int Mp = p->prof[r.model].fbQ * NL, Ld = r.jenv - r.ienv + 1;
      FbWork w; memset(&w, 0, sizeof(w));
      const uint64_t end = env_floats(Mp, Ld, w.xs_off, w.aux_off, w.mxf_off, w.mxb_off, pos, pps ? false : env_inplace());
      const uint64_t path_len = paths ? (uint64_t)Mp + (pps ? (uint64_t)Ld + 1 : 0) : 0;
      const uint64_t need = end + path_total + path_len;
      if (need > budget_floats && j > done) break;
      if (need > budget_floats) throw Error(CKM_ENOMEM, "one envelope needs more workspace than the device budget allows");
      w.model = r.model; w.seq = r.seq; w.i0 = r.ienv - 1; w.Ld = Ld; w.Lcfg = s->len[r.seq]; w.multihit = 0; w.slot = (uint32_t)(j - done); w.full = 1;
      if (paths) { w.path_off = path_total + 1; path_total += path_len; }      // relative for now: the zone starts behind the last item
      b.work.push_back(w); pos = end;
    }
    const uint64_t zone = (pos + 31) & ~(uint64_t)31;          // match-state residues of the OA paths (alignment requests), one copy back
    if (paths) for (auto &w : b.work) { w.path_off += zone; if (pps) w.path_off |= FB_PATH_WITH_PP; }
    ctx->ws.ensure((zone + path_total) * 4 + 256);
    run_fb(ctx, p, s, b, true, true, true, nullptr);
    std::vector<int32_t> zone_host;
    if (paths && path_total) { zone_host.resize(path_total); wcopy(ctx, zone_host.data(), ctx->ws.as<int32_t>() + zone, path_total * 4, hipMemcpyDeviceToHost); }
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
      if (paths) {
        std::vector<int32_t> &pv = (*paths)[done + k];
        pv.assign((size_t)p->hmm[r.model].M, 0);
        const size_t at = (size_t)((b.work[k].path_off & ~FB_PATH_WITH_PP) - 1 - zone);
        if (o.ok) std::copy(zone_host.begin() + at, zone_host.begin() + at + pv.size(), pv.begin());
        if (pps) {
          std::vector<float> &qv = (*pps)[done + k];
          const int Mp = p->prof[r.model].fbQ * NL, Ld = r.jenv - r.ienv + 1;
          qv.assign((size_t)Ld + 1, 0.f);
          if (o.ok) memcpy(qv.data(), zone_host.data() + at + Mp, sizeof(float) * ((size_t)Ld + 1));
        }
      }
    }
    done = j;
  }
}

// Exact multi-hit MSV of an arbitrary list of pairs with the packed (SSV-style) kernel: pairs are grouped by model (one LDS
// emission image per workgroup), longest sequences first, 16 sequences per workgroup (4 wavefronts x 4).  Results land in
// usc/xJ in the order of `pairs`.
void run_msv_exact(Worker *ctx, const ckm_profiles *p, const ckm_seqs *s, const std::vector<PairRec> &pairs, std::vector<float> &usc, std::vector<int32_t> *xJ) {
  // The exact multi-hit MSV bytes of arbitrary pairs (host-driven cascade, diagnostics): the kernels of the device-driven cascade in their
  // scores-only mode -- msv16_kernel (four pairs per wavefront) per 16-lane register class, msv_full_kernel (wave per pair) for models
  // beyond 2048 nodes -- over host-built queues: the pairs sorted by class, inside a class by model and longest sequence first.
  const size_t n = pairs.size();
  usc.assign(n, 0.f); if (xJ) xJ->assign(n, 0);
  if (!n) return;
  std::map<int, std::vector<uint32_t>> byQ;                                 // class -> indices into pairs (1000: wave-per-pair kernel)
  for (uint32_t i = 0; i < n; ++i) { const int q = p->prof[pairs[i].model].ssvQ; byQ[q > 64 ? 1000 : q].push_back(i); }
  std::vector<PairRec> sorted; sorted.reserve(n);
  std::vector<uint32_t> slot_of(n), counts;
  std::vector<std::pair<int, std::pair<size_t, size_t>>> groups;             // class, (first, count)
  for (auto &kv : byQ) {
    std::vector<uint32_t> &v = kv.second;
    std::stable_sort(v.begin(), v.end(), [&](uint32_t a, uint32_t b) {
      if (pairs[a].model != pairs[b].model) return pairs[a].model < pairs[b].model;
      return s->len[pairs[a].seq] > s->len[pairs[b].seq]; });
    groups.push_back({kv.first, {sorted.size(), v.size()}});
    counts.push_back((uint32_t)v.size());
    for (uint32_t i : v) { slot_of[i] = (uint32_t)sorted.size(); sorted.push_back(pairs[i]); }
  }
  ctx->msvwork.ensure(n * sizeof(PairRec)); ctx->msvlist.ensure(counts.size() * 4 + 16);
  ctx->fullx.ensure(n * 4); ctx->fullu.ensure(n * 4);
  wcopy(ctx, ctx->msvwork.p, sorted.data(), n * sizeof(PairRec), hipMemcpyHostToDevice);
  wcopy(ctx, ctx->msvlist.p, counts.data(), counts.size() * 4, hipMemcpyHostToDevice);
  CascadeDev none; memset(&none, 0, sizeof(none));
  int gi = 0;
  for (auto &g : groups) {
    const size_t first = g.second.first; const uint32_t cnt = (uint32_t)g.second.second;
    hipStream_t st = ctx->side[gi % side_streams()];
    const WorkQueue q{nullptr, ctx->msvlist.as<uint32_t>() + gi, cnt};
    ++gi;
    if (g.first == 1000)
      launch_msv_full(st, std::min<uint32_t>(cnt, 1024), q, ctx->msvwork.as<PairRec>() + first, p->d_models.as<DevModel>(), s->d_lentab.as<LenEntry>(),
                      s->d_res.as<uint8_t>(), s->d_off.as<uint64_t>(), s->d_len.as<int32_t>(), ctx->fullx.as<int32_t>() + first, ctx->fullu.as<float>() + first, p->maxMp, nullptr);
    else if (launch_msv16(g.first, std::min<uint32_t>((cnt + 15) / 16, 1024), st, q, ctx->msvwork.as<PairRec>() + first, p->d_models.as<DevModel>(), s->d_lentab.as<LenEntry>(),
                          s->d_res.as<uint8_t>(), s->d_off.as<uint64_t>(), s->d_len.as<int32_t>(), none, ctx->fullx.as<int32_t>() + first, ctx->fullu.as<float>() + first))
      throw Error(CKM_ERANGE, "no exact-MSV kernel instance for this model length");
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
    EnsWork e; memset(&e, 0, sizeof(e)); e.host_off = ~0ull;
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
    e.mx_off = pos;    pos = al(pos + (uint64_t)(Ld + 1) * 4 * Mp);        // cell-major rows of float4 {M, I, D, 0}
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
  {
    const uint32_t nreg = (uint32_t)job.ew.size();
    ctx->enscount.ensure(16);
    HIPCHK(hipMemcpyAsync(ctx->enscount.p, &nreg, 4, hipMemcpyHostToDevice, ctx->ens_stream));
    (void)maxLd;
    launch_ensemble(ctx->ens_stream, ctx->enswork.as<EnsWork>(), nullptr, ctx->enscount.as<uint32_t>(), nreg, std::min<uint32_t>(nreg, 256), maxMp, p->d_models.as<DevModel>(),
                    s->d_lentab.as<LenEntry>(), s->d_res.as<uint8_t>(), s->d_off.as<uint64_t>(), ctx->ws_ens.as<float>(), ctx->ensseeds.as<uint32_t>(), nullptr);
  }
  HIPCHK(hipGetLastError());
  ctx->h_ens.ensure(job.res_floats * 4);
  HIPCHK(hipMemcpyAsync(ctx->h_ens.p, ctx->ws_ens.p, job.res_floats * 4, hipMemcpyDeviceToHost, ctx->ens_stream));
  job.in_flight = true;
}

void ensure_ens_seeds(Worker *ctx) {
  if (ctx->ensseeds.p) return;
  std::vector<uint32_t> seeds(ENS_NSAMPLES);
  for (int t = 0; t < ENS_NSAMPLES; ++t) seeds[t] = ens_seed(t);
  ctx->ensseeds.ensure(seeds.size() * 4);
  wcopy(ctx, ctx->ensseeds.p, seeds.data(), seeds.size() * 4, hipMemcpyHostToDevice);
}

void ens_begin(Worker *ctx, const ckm_profiles *p, const ckm_seqs *s, EnsJob &job) {
  if (job.req.empty()) return;
  ensure_ens_seeds(ctx);
  auto al = [](uint64_t v) { return (v + 31) & ~(uint64_t)31; };
  const uint64_t budget_floats = ctx->ws_budget / 4;
  job.batches.clear(); job.next_batch = 0;
  uint64_t pos = 0; size_t first = 0;
  for (size_t j = 0; j < job.req.size(); ++j) {
    const RegionReq &r = job.req[j];
    const uint64_t Mp = p->prof[r.model].fbQ * NL, Ld = r.jreg - r.ireg + 1;
    const uint64_t need = 256 + (uint64_t)ENS_NSAMPLES * job.cap[j] * 4 + al(Ld) + al((Ld + 1) * 6) + al((Ld + 1) * 4 * Mp) +
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

}  // namespace ckm
