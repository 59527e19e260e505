// ckm_search.hip -- the filter cascade of one worker (SSV -> MSV -> bias -> Viterbi -> Forward -> Backward -> domain definition ->
// envelope rescoring -> scores) and ckm_search, which spreads a search over the workers and assembles the rows.
// Replaces the per-target pipeline of the hmmsearch process CheckM launches per bin (checkm/hmmer.py:70).
#include "ckm_host.h"

namespace {

#define CKM_TRACE_PT(label) trace_pt(ctx, label)     // CKM_TRACE=1: per-worker stage timestamps on stderr

}  // namespace


typedef std::map<std::pair<uint32_t, uint32_t>, std::vector<Hit>> HitMap;     // (bin, model) -> hits

// Scores, thresholds and rows of the pairs that reached the domain stage: null2 corrections, bit scores, P-values (host libm), one Hit
// per pair with at least one envelope.  Shared by the device-driven cascade and the host-driven one.
void assemble_hits(Worker *ctx, const ckm_profiles *p, const ckm_seqs *s, DomStage &ds, HitMap &by_bin_model) {
  const std::vector<DomItem> &items = ds.items; const std::vector<RegionRes> &regres = ds.regres; const std::vector<EnvReq> &envreq = ds.envreq;
  const std::vector<int> &env_region = ds.env_region; std::vector<EnvRes> &envres = ds.envres; const std::vector<int> &nregions = ds.nregions;
  const std::vector<std::pair<size_t, size_t>> &env_of_pass = ds.env_of_pass;
  std::vector<std::pair<size_t, size_t>> items_of(ds.pass.size(), {0, 0});
  { size_t it = 0; for (size_t q = 0; q < ds.pass.size(); ++q) { items_of[q].first = it; while (it < items.size() && items[it].pass == q) ++it; items_of[q].second = it; } }
  std::vector<Hit> hit_of(ds.pass.size()); std::vector<uint8_t> has_hit(ds.pass.size(), 0);
  pool_run(ctx, ds.pass.size(), 32, [&](size_t qlo, size_t qhi) {
  std::vector<float> n2sc;
  for (size_t q = qlo; q < qhi; ++q) {
    const PassInfo &c = ds.pass[q];
    const HostHMM &hm = p->hmm[c.model];
    const int L = s->len[c.seq];
    const uint8_t *dsq = s->dsq.data() + s->off[c.seq];
    const float nullsc = s->lentab[L].nullsc;
    n2sc.assign((size_t)L + 2, 0.f);
    Hit h; h.model = c.model; h.seq = c.seq; h.L = L; h.nreported = 0;
    int nenv = 0;
    for (size_t item_at = items_of[q].first; item_at < items_of[q].second; ++item_at) if (items[item_at].region >= 0) {
      // null2 of an ensemble region: log of the mean odds ratio over the traces, for every residue of the region
      const DomItem &im = items[item_at]; const RegionRes &rr = regres[im.region];
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
  for (size_t q = 0; q < ds.pass.size(); ++q) if (has_hit[q]) by_bin_model[{s->seq_bin[hit_of[q].seq], hit_of[q].model}].push_back(std::move(hit_of[q]));
}

// The whole filter cascade + domain stage for a subset of the models, on one worker.
struct SeqRange { std::vector<uint32_t> lo, hi; std::vector<uint64_t> res; uint64_t tag = 0; };   // per bin: [lo, hi) of s->order, residues in it

static void cascade(Worker *ctx, ckm_ctx *owner, int my_turn, const ckm_profiles *p, const ckm_seqs *s, const SeqRange &rng, const std::vector<uint32_t> &my_models,
                    const std::vector<std::vector<uint32_t>> &model_bins, HitMap &by_bin_model, bool turn_done = false) {
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
  bool took_turn = turn_done;          // (the lane's turn was already taken and passed on by the device-driven attempt this call replaces)
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
          const int Q = ssv_class(p->prof[mw.model]); const uint32_t per_block = ssv_per_block(Q);
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
      uint32_t cap_surv = (uint32_t)std::max<uint64_t>(1 << 16, npairs / 8), cap_nores = (uint32_t)std::max<uint64_t>(1 << 14, npairs / 64);
      for (int attempt = 0;; ++attempt) {
        ctx->surv.ensure((size_t)cap_surv * sizeof(PairRec)); ctx->nores.ensure((size_t)cap_nores * sizeof(PairRec)); ctx->counters.ensure(64);
        HIPCHK(hipMemsetAsync(ctx->counters.p, 0, 64, ctx->stream));
        HIPCHK(hipEventRecord(ctx->ev[0], ctx->stream));
        {
          // register classes go round-robin over 4 streams (heaviest first) so the tail of one launch -- a few very long
          // sequences -- is covered by the next launch; ev[0]..ev[1] on the main stream brackets all of them.  The SSV lanes finish
          // the MSV stage themselves (survivor / exact-MSV tables); a table that overflowed means the launches run again with larger ones.
          const SsvEpi epi{lt, ctx->surv.as<PairRec>(), ctx->counters.as<uint32_t>(), cap_surv, ctx->nores.as<PairRec>(), ctx->counters.as<uint32_t>() + 1, cap_nores, nullptr};
          const int NS = std::min(4, side_streams());
          for (int k = 0; k < NS; ++k) HIPCHK(hipStreamWaitEvent(ctx->side[k], ctx->ev[0], 0));
          int gi = 0;
          for (auto it = groups.rbegin(); it != groups.rend(); ++it, ++gi) {
            auto &g = *it;
            if (launch_ssv(g.first, (int)g.second.second, ssv_threads_for(g.first), ctx->side[gi % NS], ctx->work.as<SsvBlockWork>() + g.second.first, dm, res, off, dlen,
                           s->d_order.as<uint32_t>(), epi))
              throw Error(CKM_ERANGE, "no SSV kernel instance for this model length");
            if (attempt == 0) st.ssv_launches++;
          }
          for (int k = 0; k < NS; ++k) { HIPCHK(hipEventRecord(ctx->ev[2 + k], ctx->side[k])); HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev[2 + k], 0)); }
        }
        HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
        HIPCHK(hipGetLastError());
        uint32_t cnt[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(cnt, ctx->counters.p, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (attempt == 0) { float ms = 0.f; HIPCHK(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1])); st.ms_ssv += ms; }
        CKM_TRACE_PT("ssv (+ fused finish) done");
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

  CKM_TRACE_PT("stage1 done (ssv, exact msv)");
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
        std::vector<uint32_t> qctl(grp.size(), 0u);                 // queue length per register class
        for (size_t k = 0; k < grp.size(); ++k) qctl[k] = (uint32_t)grp[k].second.second;
        ctx->vitq.ensure(qctl.size() * 4 + 16);
        wcopy(ctx, ctx->vitq.p, qctl.data(), qctl.size() * 4, hipMemcpyHostToDevice);
        int gi = 0;
        for (size_t k = grp.size(); k-- > 0; ++gi) {
          auto &g = grp[k];
          uint32_t *qd = ctx->vitq.as<uint32_t>() + k;
          const uint32_t cnt = (uint32_t)g.second.second;
          if (launch_vit(g.first, std::min<uint32_t>((cnt + 3) / 4, 2048), ctx->side[gi % std::min(4, side_streams())],
                         WorkQueue{ctx->fbidx.as<uint32_t>() + g.second.first, qd, cnt}, ctx->cand.as<PairRec>(), dm, lt, res, off, dlen,
                         ctx->vitx.as<int32_t>(), ctx->vits.as<float>(), ctx->vitf.as<uint32_t>(), fast, nullptr))
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
        for (const Seg &e : regres[im.region].env) {
          const int i2 = e.sqfrom + im.i - 1, j2 = e.sqto + im.i - 1;
          // an envelope overlapping its predecessor is rescored like any other (HMMER only counts it); duplicate alignments are
          // hidden at reporting time (its bug #h74 workaround, below)
          envreq.push_back({w.model, w.seq, i2, j2}); env_region.push_back(im.region);
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
  DomStage ds;
  ds.pass.resize(passers.size());
  for (size_t q = 0; q < passers.size(); ++q) { const Cand &c = cands[fb_cand[passers[q]]]; ds.pass[q] = {c.r.model, c.r.seq, c.fwdsc}; }
  ds.nregions = nregions;
  for (const Item &im : items) ds.items.push_back({im.pass, im.i, im.j, im.region});
  ds.regres = std::move(regres); ds.envreq = std::move(envreq); ds.env_region = std::move(env_region); ds.envres = std::move(envres);
  ds.env_of_pass = std::move(env_of_pass);
  assemble_hits(ctx, p, s, ds, by_bin_model);
  st.ms_host = now_ms() - t_host0;
  st.ms_total = now_ms() - t_start;
  CKM_TRACE_PT("cascade done");
}

// ---------------------------------------------------------------------------------------------------------------------------
// The device-driven cascade of one lane (one worker = one length class of the sequences): every stage behind SSV is launched without
// waiting for the one before it -- survivors travel through device-side queues (dev_types.h: CascadeDev), the kernels' epilogues take
// the filter decisions conservatively, the region scan and the workspace allocation of the envelope stage run on the device -- and
// the host synchronises ONCE, when the chain has drained, to take the decisions again exactly (libm) and assemble the rows.
// A second, short round follows only for the envelopes that come out of the trace ensembles (clustered on the host).
// Returns 0 when done; 1 when a table or the workspace was too small for this search (the lane's SSV turn has been taken): the caller
// then runs the lane through the host-driven cascade (which batches by workspace) and the grown capacities serve the next call;
// 2 when the search was not attempted (more pairs than one SSV pass holds): host-driven cascade, turn still to be taken.
// ---------------------------------------------------------------------------------------------------------------------------
namespace {

template <class T> T *dev_table(DevBuf &b, size_t n) { b.ensure(std::max<size_t>(1, n) * sizeof(T)); return b.as<T>(); }
template <class T> T *pin_table(PinnedBuf &b, size_t n) { b.ensure(std::max<size_t>(1, n) * sizeof(T)); return b.as<T>(); }

}  // namespace

// One SSV phase at a time per device, across contexts (see cascade_dev).  The events belong to the baton and live as long as the process.
struct DeviceBaton { std::mutex m; hipEvent_t ev[2] = {nullptr, nullptr}; int k = 0; bool recorded = false; std::atomic<int> searches{0}; };
static DeviceBaton &device_baton(int dev) { static DeviceBaton b[64]; return b[(unsigned)dev & 63]; }
static bool late_on() { static const bool on = !(getenv("CKM_LATE_PRIO") && atoi(getenv("CKM_LATE_PRIO")) == 0); return on; }
static bool baton_on() { static const bool on = !(getenv("CKM_SSV_BATON") && atoi(getenv("CKM_SSV_BATON")) == 0); return on; }

static int cascade_dev(Worker *ctx, ckm_ctx *owner, int my_turn, const ckm_profiles *p, const ckm_seqs *s, const SeqRange &rng,
                        const std::vector<uint32_t> &my_models, const std::vector<std::vector<uint32_t>> &model_bins, HitMap &by_bin_model) {
  HIPCHK(hipSetDevice(ctx->device));
  const double t_start = now_ms();
  ckm_search_stats &st = ctx->stats;
  memset(&st, 0, sizeof(st));
  const DevModel *dm = p->d_models.as<DevModel>();
  const LenEntry *lt = s->d_lentab.as<LenEntry>();
  const uint8_t *res = s->d_res.as<uint8_t>();
  const uint64_t *off = s->d_off.as<uint64_t>();
  const int32_t *dlen = s->d_len.as<int32_t>();
  bool took_turn = false;
  struct TurnGuard {
    ckm_ctx *o; int t; bool *took;
    ~TurnGuard() { if (*took) return; std::unique_lock<std::mutex> l(o->ssv_mutex); o->ssv_cv.wait(l, [&] { return o->ssv_turn == t; }); o->ssv_turn++; o->ssv_cv.notify_all(); }
  } turn_guard{owner, my_turn, &took_turn};

  // ---- the lane's pairs; a search with more pairs than the SSV budget goes through the host-driven cascade (it works chunk by chunk) ----
  uint64_t pair_budget = (uint64_t)1 << 29;
  if (const char *e = getenv("CKM_PAIR_BUDGET")) pair_budget = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
  struct MW { uint32_t model; uint64_t pair_base; uint64_t npairs; };
  std::vector<MW> mws;
  uint64_t total_pairs = 0;
  for (uint32_t m1 : my_models) {
    uint64_t n = 0;
    for (uint32_t b : model_bins[m1]) n += rng.hi[b] - rng.lo[b];
    if (n == 0) continue;
    mws.push_back({m1, total_pairs, n}); total_pairs += n;
  }
  if (total_pairs == 0) { st.ms_total = now_ms() - t_start; return 0; }
  if (total_pairs > pair_budget) return 2;

  // ---- SSV block table, grouped by SSV register class = model length class (cached when the previous call on this worker had the same
  // plan: lineage_wf scans the same bins twice, bench repeats steps).  Each group is one SSV launch AND one sub-cascade: its survivors
  // go down their own chain of queues as soon as that launch is done, while the SSV launches of the other groups still run. ----
  hipStream_t ms = ctx->stream;
  // Parts by sequence length: the long sequences of every bin (a prefix of its length-sorted order) are scanned first, so that their
  // chains -- every stage lasts as long as its longest sequence -- run underneath the SSV launches of the parts behind them, and what
  // is left after the last SSV launch is the chain of the shortest sequences only.
  const uint32_t nbins = s->nbins;
  // cut[k][b] .. cut[k+1][b] = part k of bin b's length-sorted order (lengths descend along it); CKM_LONG_SHARE="0.25,0.5" gives the
  // shares of the residues of all parts but the last.  Two parts by default (the longest sequences holding a quarter of the residues,
  // then the rest); a third part of the shortest sequences, scanned last so that only short chains are left at the end, was measured
  // and does not pay: 18 more launches and chains cost what the shorter tail gains (cfg2: 86.4-88.0 ms against 85.3).
  std::vector<std::vector<uint32_t>> cut(1, rng.lo);
  int nparts = 1, Lcut = 0;
  std::vector<int> Lcuts;
  {
    uint64_t split_min = 2000000;
    if (const char *e = getenv("CKM_SPLIT_MIN_PAIRS")) split_min = strtoull(e, nullptr, 10);
    std::vector<double> shares{0.25};
    if (const char *e = getenv("CKM_LONG_SHARE")) {
      shares.clear();
      for (const char *q = e; *q;) {
        char *end = nullptr;
        const double v = strtod(q, &end);
        if (end == q) break;
        if (v > 0.0 && v < 1.0) shares.push_back(v);
        if (*end != ',') break;
        q = end + 1;
      }
    }
    if (total_pairs >= split_min && !shares.empty()) {
      std::vector<uint64_t> by_len((size_t)s->maxL + 2, 0);
      uint64_t total = 0;
      for (uint32_t b = 0; b < nbins; ++b) for (uint32_t k = rng.lo[b]; k < rng.hi[b]; ++k) { const int L = s->len[s->order[k]]; by_len[L] += (uint64_t)L; total += (uint64_t)L; }
      uint64_t acc = 0; double want = 0.0; size_t si = 0;
      want = shares[0];
      for (int L = s->maxL; L >= 1 && si < shares.size(); --L) {
        acc += by_len[L];
        if ((double)acc >= want * (double)total) {
          if (L - 1 > 0 && (Lcuts.empty() || L - 1 < Lcuts.back())) Lcuts.push_back(L - 1);      // part boundary: lengths > L-1 | <= L-1
          if (++si < shares.size()) want += shares[si];
        }
      }
      for (int lc : Lcuts) {
        std::vector<uint32_t> c(nbins);
        for (uint32_t b = 0; b < nbins; ++b) {
          uint32_t lo = rng.lo[b], hi = rng.hi[b];                 // first position with len <= lc
          while (lo < hi) { const uint32_t m = (lo + hi) / 2; if (s->len[s->order[m]] > lc) lo = m + 1; else hi = m; }
          c[b] = lo;
        }
        cut.push_back(std::move(c));
      }
      nparts = (int)Lcuts.size() + 1;
      if (!Lcuts.empty()) Lcut = Lcuts[0];
    }
  }
  cut.push_back(rng.hi);
  // key = part * 1000 + code of the SSV launch class, the code growing with the length of the models (groups are launched heaviest first):
  // 8-lane classes 100 + Q8 (models of <= 512 nodes) -> Q8, 16-lane classes Q -> 40 + Q, kSsvNone -> 105
  auto enc = [](int cls) { return cls >= 100 ? cls - 100 : 40 + cls; };
  auto dec = [](int code) { return code < 40 ? 100 + code : code - 40; };
  std::vector<std::pair<int, std::pair<size_t, size_t>>> groups;
  {
    std::vector<uint64_t> key{p->uid, s->uid, pair_budget, (uint64_t)Lcut, rng.tag, 0xdeull};
    for (int lc : Lcuts) key.push_back(0xc0000000ull + (uint64_t)lc);
    for (auto &mw : mws) { key.push_back(0xffffffffull + mw.model); for (uint32_t b : model_bins[mw.model]) key.push_back(b); }
    if (key == ctx->plan_key) {
      groups = ctx->plan_groups;
      st.pairs_ssv += ctx->plan_pairs; st.residue_hmm += ctx->plan_residue_hmm; st.cells_ssv += ctx->plan_cells;
    } else {
      // runs = the sequences of one (model, bin, part), already longest first; a launch takes the first block of every run, then the
      // second of every run, ...: its blocks come out (nearly) longest first without sorting a million of them (a block's time is set
      // by its first = longest sequence, and the few very long ones must not start last)
      struct Run { uint32_t model, first, count; uint64_t pair0; };
      std::map<int, std::vector<Run>> runs;
      uint64_t c_pairs = 0, c_res = 0, c_cells = 0;
      for (auto &mw : mws) {
        const int Q = ssv_class(p->prof[mw.model]);
        uint64_t pb = mw.pair_base;
        for (uint32_t b : model_bins[mw.model]) {
          const uint32_t o0 = rng.lo[b], n = rng.hi[b] - o0;
          for (int part = 0; part < nparts; ++part) {
            const uint32_t a0 = cut[part][b] - o0, a1 = cut[part + 1][b] - o0;
            if (a1 > a0) runs[part * 1000 + enc(Q)].push_back({mw.model, o0 + a0, a1 - a0, pb + a0});
          }
          pb += n; c_res += rng.res[b]; c_cells += rng.res[b] * (uint64_t)p->prof[mw.model].M;
        }
        c_pairs += mw.npairs;
      }
      st.pairs_ssv += c_pairs; st.residue_hmm += c_res; st.cells_ssv += c_cells;
      std::vector<SsvBlockWork> allw;
      for (auto &kv : runs) {
        const int Q = dec(kv.first % 1000); const uint32_t per_block = ssv_per_block(Q);
        const size_t first = allw.size();
        std::vector<Run> &rv = kv.second;
        std::stable_sort(rv.begin(), rv.end(), [&](const Run &x, const Run &y) { return s->len[s->order[x.first]] > s->len[s->order[y.first]]; });
        for (uint32_t a = 0;; a += per_block) {
          bool any = false;
          for (const Run &r : rv) if (a < r.count) {
            any = true;
            SsvBlockWork w; w.model = r.model; w.list_start = r.first + a; w.count = std::min(per_block, r.count - a); w.pair_start = (uint32_t)(r.pair0 + a);
            allw.push_back(w);
          }
          if (!any) break;
        }
        groups.push_back({kv.first, {first, allw.size() - first}});
      }
      ctx->work.ensure(allw.size() * sizeof(SsvBlockWork));
      // (staged, copied by a kernel on a high-priority stream: see Stager)
      ctx->stager.begin(allw.size() * sizeof(SsvBlockWork));
      ctx->stager.put(ctx->late[2], ctx->work.p, allw.data(), allw.size() * sizeof(SsvBlockWork));
      HIPCHK(hipStreamSynchronize(ctx->late[2]));  // the host waits, so everything queued below comes after
      ctx->plan_key = key; ctx->plan_groups = groups; ctx->plan_nblocks = allw.size(); ctx->plan_pairs = c_pairs; ctx->plan_residue_hmm = c_res; ctx->plan_cells = c_cells;
    }
  }
  // per group: its pairs and the Viterbi / Forward register classes of its models
  struct Sub { int Q; int maxM = 0; size_t first, nblocks; uint64_t pairs = 0; bool vit[NVC] = {false}, fb[NFC] = {false};
               uint32_t cap_cand = 0, cap_nores = 0, cap_f = 0, cap_e = 0, cap_r = 0; size_t o_cand = 0, o_nores = 0, o_vq = 0, o_f = 0, o_e = 0, o_r = 0; };
  std::vector<Sub> subs;
  {
    std::map<int, size_t> at;
    for (auto &g : groups) { Sub sb; sb.Q = dec(g.first % 1000); sb.first = g.second.first; sb.nblocks = g.second.second; at[g.first] = subs.size(); subs.push_back(sb); }
    for (auto &mw : mws) {
      for (int part = 0; part < nparts; ++part) {
        uint64_t np = 0;                                      // the model's pairs in this part
        for (uint32_t b : model_bins[mw.model]) np += cut[part + 1][b] - cut[part][b];
        auto it = at.find(part * 1000 + enc(ssv_class(p->prof[mw.model])));
        if (it == at.end()) continue;                         // (no sequence of this part in the model's bins)
        Sub &sb = subs[it->second]; sb.pairs += np; sb.vit[p->dm[mw.model].vit_cls] = true; sb.vit[p->dm[mw.model].vitx_cls] = true; sb.fb[p->dm[mw.model].fb_cls] = true; sb.maxM = std::max(sb.maxM, p->prof[mw.model].M);
      }
    }
  }
  const size_t NG = subs.size();
  if (NG > 192) return 2;                          // (grp_ev; 2 parts x (30 eight-lane + 29 sixteen-lane SSV classes + 1) at most)
  CKM_TRACE_PT("plan ready");

  // ---- capacities: shares of the pairs (the divisors halve when a table overflowed on an earlier call), tables ----
  Worker::CascadeCaps &cp = ctx->caps;
  size_t tot_cand = 0, tot_nores = 0, tot_f = 0, tot_e = 0, tot_r = 0;
  // (tests: CKM_CAP_SHRINK=n makes every table n times smaller than its share and drops the floors, to force the overflow -> host-driven path)
  const uint64_t shrink = getenv("CKM_CAP_SHRINK") ? std::max(1, atoi(getenv("CKM_CAP_SHRINK"))) : 1;
  auto capof = [&](uint64_t pairs, uint32_t div, uint64_t floor_) {
    return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(shrink > 1 ? 8 : floor_, pairs / ((uint64_t)div * shrink)), 0x7ffffff0ull); };
  for (Sub &sb : subs) {
    sb.cap_cand = capof(sb.pairs, cp.div_cand, cp.fl_cand);
    sb.cap_nores = sb.Q == kSsvNone ? (uint32_t)std::min<uint64_t>(sb.pairs + 64, 0x7ffffff0ull)     // no SSV for these models: every pair is recomputed exactly
                                    : capof(sb.pairs, cp.div_nores, cp.fl_nores);
    sb.cap_f = capof(sb.pairs, cp.div_fwork, cp.fl_fwork);
    sb.cap_e = capof(sb.pairs, cp.div_ework, cp.fl_ework);
    sb.cap_r = capof(sb.pairs, cp.div_rwork, cp.fl_rwork);
    sb.o_cand = tot_cand; tot_cand += sb.cap_cand; sb.o_nores = tot_nores; tot_nores += sb.cap_nores;
    sb.o_vq = sb.o_cand * NVC;
    sb.o_f = tot_f * NFC; tot_f += sb.cap_f; sb.o_e = tot_e * NFC; tot_e += sb.cap_e; sb.o_r = tot_r * NFC; tot_r += sb.cap_r;
  }
  auto grow = [](uint32_t &v, uint64_t want) { if (v < want) v = (uint32_t)std::min<uint64_t>(want, 0xfffffff0ull); };
  grow(cp.fwork, tot_f); grow(cp.ework, tot_e); grow(cp.rwork, tot_r);
  grow(cp.pass, std::max<uint64_t>(1 << 13, tot_e)); grow(cp.reg, (uint64_t)cp.ework + cp.rwork);
  grow(cp.events_f, std::max<uint64_t>(1 << 18, (uint64_t)cp.fwork * 16));
  grow(cp.events_e, std::max<uint64_t>(1 << 16, (uint64_t)cp.ework * 16));
  // (the export buffer of the ensembles is sized for the regions SEEN so far, with room -- not for the region tables' capacity, which is
  //  a floor per group times the groups; an overflow doubles it: CS_RWORK)
  cp.hens = std::max<uint64_t>(cp.hens, std::min<uint64_t>(cp.rwork, (uint64_t)4 * cp.seen_rwork + 8192) * (256 + ENS_NSAMPLES * 16 * 4 + 1024));
  // the float workspace: an estimate from the pairs (special rows of ~0.3 % of them, matrices of ~0.06 %), within the lane's budget
  // the float workspace, estimated from the cells the search is expected to fill: a marker model finds about one domain in a bin it is
  // scanned against, and a domain costs its envelope's matrix (about M rows of Mp floats, three arrays in place) -- so the demand follows
  // sum over models of (bins scanned against the model) x (Mp + 64) x Mp, whatever the number of ORFs in those bins and whichever kind
  // of search it is (43 phylogenetic markers against hundreds of bins, or a lineage's hundreds of models against a few bins).  Rounds
  // 2-4 priced it per (pair x model position), which follows the ORF count instead: one factor could not fit both kinds (round 3: 156 GB
  // allocated for 76 GB used), two factors still overshot 2x.  Measured on the 1000-bin workload (profiles/r04r_workspace_per_cell.txt):
  // 9.78-9.82 B per cell where every model is present in every bin (the 43 phylogenetic markers), 4.1-4.7 where about half of a
  // lineage's models are; the first estimate is 11 B (the first kind + 12 %).  A search that outgrows the estimate falls back once
  // (deferred regions) and leaves its measured bytes per cell (+15 %) for the next calls.
  double cell_sum = 0.0;
  {
    for (auto &mw : mws) { const double Mp = (double)(p->prof[mw.model].fbQ * NL); cell_sum += (double)model_bins[mw.model].size() * (Mp + 64.0) * Mp; }
    const uint64_t est = (uint64_t)(cell_sum * cp.ws_per_cell + (double)total_pairs * 24.0) + ((uint64_t)256 << 20);
    const size_t want = (size_t)std::min<uint64_t>(std::max<uint64_t>(est, (uint64_t)1 << 30), (uint64_t)ctx->ws_budget);
    if (ctx->ws.cap < want) ctx->ws.ensure(want);
  }
  const uint64_t ws_floats = ctx->ws.cap / 4;
  CKM_TRACE_PT("workspace ready");
  uint32_t *d_gcnt = dev_table<uint32_t>(ctx->c_cnt, (NG + 1) * CC_SIZE);        // block 0: counters shared by the groups; block 1 + g: group g's
  unsigned long long *d_tops = dev_table<unsigned long long>(ctx->c_tops, 4);
  CascadeDev cd0; memset(&cd0, 0, sizeof(cd0));
  PairRec *d_cand = dev_table<PairRec>(ctx->c_cand, tot_cand), *d_nores = dev_table<PairRec>(ctx->c_nores, tot_nores);
  float *d_bias = dev_table<float>(ctx->c_bias, tot_cand * 2), *d_vfast = dev_table<float>(ctx->c_vfast, tot_cand), *d_vexact = dev_table<float>(ctx->c_vexact, tot_cand);
  uint32_t *d_vflag = dev_table<uint32_t>(ctx->c_vflag, tot_cand); uint8_t *d_route = dev_table<uint8_t>(ctx->c_route, tot_cand);
  uint32_t *d_vq = dev_table<uint32_t>(ctx->c_vq, tot_cand * NVC), *d_vxq = dev_table<uint32_t>(ctx->c_vxq, tot_cand * NVC);
  uint32_t *d_fq = dev_table<uint32_t>(ctx->c_fq, tot_f * NFC), *d_bq = dev_table<uint32_t>(ctx->c_bq, tot_f * NFC);
  uint32_t *d_eq = dev_table<uint32_t>(ctx->c_eq, tot_e * NFC), *d_rq = dev_table<uint32_t>(ctx->c_rq, tot_r * NFC);
  cd0.gcnt = d_gcnt;
  cd0.fwork = dev_table<FbWork>(ctx->c_fwork, cp.fwork); cd0.cap_fwork = cp.fwork;
  cd0.ework = dev_table<FbWork>(ctx->c_ework, cp.ework); cd0.cap_ework = cp.ework;
  cd0.rwork = dev_table<FbWork>(ctx->c_rwork, cp.rwork); cd0.ens = dev_table<EnsWork>(ctx->c_ens, cp.rwork); cd0.cap_rwork = cp.rwork;
  uint32_t *d_ensq = dev_table<uint32_t>(ctx->c_ensq, (size_t)4 * cp.rwork);       // region ids by sequence part
  // zone 1: special rows and decoding terms of the parser items (~a few KB for ~0.3 % of the pairs); zone 2: everything else
  {
    const uint64_t z1 = std::min<uint64_t>(ws_floats / 2, ((uint64_t)total_pairs * 64 + ((uint64_t)64 << 20)) / 4) & ~(uint64_t)31;
    cd0.ws_top = d_tops; cd0.ws_cap = z1;
    cd0.ws2_top = d_tops + 2; cd0.ws2_base = z1; cd0.ws2_cap = ws_floats - z1;
  }
  // result tables live in device memory; once the counters are known their used prefixes are copied to pinned staging in one go
  cd0.h_pass = dev_table<PassRec>(ctx->c_pass, cp.pass); cd0.cap_pass = cp.pass;
  cd0.h_reg = dev_table<RegionRec>(ctx->c_reg, cp.reg); cd0.cap_reg = cp.reg;
  float *d_hens = dev_table<float>(ctx->c_hens, cp.hens);
  cd0.hens_top = d_tops + 1; cd0.hens_cap = cp.hens;
  cd0.seq_len = dlen;
  cd0.margin_msv = 0.01f; cd0.margin_vit = 0.01f; cd0.margin_fwd = 0.05f;
  cd0.env_inplace = env_inplace() ? 1u : 0u;
  FwdOut *d_fout_f = dev_table<FwdOut>(ctx->c_fout_f, cp.fwork), *d_fout_e = dev_table<FwdOut>(ctx->c_fout_e, cp.ework), *d_fout_r = dev_table<FwdOut>(ctx->c_fout_r, cp.rwork);
  int32_t *d_rerr_e = dev_table<int32_t>(ctx->c_rerr_e, cp.ework);
  ScaleEvent *d_events_r = dev_table<ScaleEvent>(ctx->c_events_r, 1 << 16);
  EnvOut *d_envout = dev_table<EnvOut>(ctx->c_envout, cp.ework);
  ScaleEvent *d_events_f = dev_table<ScaleEvent>(ctx->c_events_f, cp.events_f), *d_events_e = dev_table<ScaleEvent>(ctx->c_events_e, cp.events_e);
  uint32_t *h_cnt = pin_table<uint32_t>(ctx->h_cnt, (NG + 1) * CC_SIZE);
  ensure_ens_seeds(ctx);
  float *ws = ctx->ws.as<float>();
  CKM_TRACE_PT("tables ready");
  const int NS = side_streams();
  const int NSS = NS >= 8 ? 4 : std::max(1, NS / 3); // streams of the SSV launches
  const int NCH = NS - NSS;                        // streams of the groups' chains (none left: a chain follows its SSV launch on the same stream)

  // ---- wait for the turn (the lanes' SSV phases run one behind the other on the device: VALU-bound, nothing to gain side by side),
  // queue everything behind the previous lane's SSV launches, pass the turn on ----
  {
    std::unique_lock<std::mutex> lock(owner->ssv_mutex);
    owner->ssv_cv.wait(lock, [&] { return owner->ssv_turn == my_turn; });
  }
  CKM_TRACE_PT("ssv turn taken");
  // (diagnostics: CKM_CHAIN_STOP=n queues only the first n stages of every chain, prints the device counters and hands the lane to the
  //  host-driven cascade: 1 SSV + finish, 2 exact MSV, 3 bias filter, 4 Viterbi fast, 5 Viterbi exact, 6 Forward parser, 7 Backward
  //  parser, 8 regions, 9-11 envelope Forward / Backward / OA, 12 region Forward, 13 ensembles)
  const int stop = getenv("CKM_CHAIN_STOP") ? atoi(getenv("CKM_CHAIN_STOP")) : 99;
  constexpr uint32_t GRID_FB = 4096, GRID_VIT = 2048, GRID_MSV = 1024;      // workgroups of the persistent chain kernels (other sizes were measured in round 2: no gain)
  HIPCHK(hipMemsetAsync(d_gcnt, 0, (NG + 1) * CC_SIZE * sizeof(uint32_t), ms));
  HIPCHK(hipMemsetAsync(d_tops, 0, 4 * sizeof(unsigned long long), ms));
  if (owner->ssv_prev_done) HIPCHK(hipStreamWaitEvent(ms, owner->ssv_prev_done, 0));     // previous lane's SSV launches
  // ... and the SSV launches of whichever search was queued on this DEVICE before this one, another context's included (find() keeps two
  // contexts in flight): two SSV phases side by side finish together, the two host threads then do their between-searches work at the
  // same time and the device waits for both (measured: 18.6 % of a 1000-bin step idle, profiles/r03r_timeline_cfg3_1000bins.txt);
  // one behind the other, each search's tail, copies and host work lie underneath the other's SSV phase.  CKM_SSV_BATON=0: off.
  DeviceBaton &baton = device_baton(owner->device);
  struct InFlight { DeviceBaton &b; InFlight(DeviceBaton &x) : b(x) { b.searches++; } ~InFlight() { b.searches--; } } in_flight{baton};
  std::unique_lock<std::mutex> baton_lock(baton.m, std::defer_lock);
  if (baton_on()) {
    baton_lock.lock();
    if (!baton.ev[0]) for (auto &e : baton.ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (baton.recorded) HIPCHK(hipStreamWaitEvent(ms, baton.ev[baton.k], 0));
  }
  HIPCHK(hipEventRecord(ctx->ev[0], ms));
  HIPCHK(hipEventRecord(ctx->cev[0], ms));
  for (int k = 0; k < NS; ++k) HIPCHK(hipStreamWaitEvent(ctx->side[k], ctx->cev[0], 0));
  {
    std::vector<size_t> launch_order;                            // long part first; inside a part the heaviest register class first
    {
      size_t g0 = 0;                                             // groups are sorted by key = part * 1000 + class
      for (int part = 0; part < nparts; ++part) {
        size_t g1 = g0;
        while (g1 < NG && groups[g1].first / 1000 == part) ++g1;
        for (size_t g = g1; g-- > g0;) launch_order.push_back(g);
        g0 = g1;
      }
    }
    // The trace ensembles of the multi-domain regions: ONE set of launches per sequence PART, behind the chains of all of the part's
    // groups, on a stream of their own.  A region's 200 traces run one after the other (one generator stream per region, as
    // hmmsearch carries it), so a launch lasts as long as its longest region -- tens of milliseconds; per group and register class, on the
    // group's chain stream (round 3, when a launch took 2 ms), those latencies queued up behind each other (measured: 11.5 -> 14.1 s per
    // 1000 bins).  The long part's ensembles run underneath the SSV launches of the short part; what is left after the last SSV launch
    // are the regions of the short sequences.
    int ens_Mp = NL;
    for (const Sub &sb : subs) for (int c = 0; c < NFC; ++c) if (sb.fb[c]) ens_Mp = std::max(ens_Mp, kFbQ[c] * NL);
    hipStream_t es = ctx->ens_stream;
    bool ens_launched = false;
    auto ens_slot = [](int part) { return std::min(part, 3); };
    auto flush_part = [&](int part) {
      if (stop < 13) return;
      const int k0 = NCH > 0 ? NSS : 0, k1 = NCH > 0 ? NS : NSS;
      for (int k = k0; k < k1; ++k) { HIPCHK(hipEventRecord(ctx->ens_ev[k], ctx->side[k])); HIPCHK(hipStreamWaitEvent(es, ctx->ens_ev[k], 0)); }
      const int sl = ens_slot(part);
      launch_ensemble(es, cd0.ens, d_ensq + (size_t)sl * cp.rwork, d_gcnt + CC_ENSQ + sl, cp.rwork, 64, ens_Mp, dm, lt, res, off, ws, ctx->ensseeds.as<uint32_t>(), d_hens);
      ens_launched = true;
    };
    int gi = 0, cur_part = -1;
    for (size_t g : launch_order) {
      const int gi_now = gi++;
      const Sub &sb = subs[g];
      const int part = groups[g].first / 1000;
      if (cur_part >= 0 && ens_slot(part) != ens_slot(cur_part)) flush_part(cur_part);
      cur_part = part;
      hipStream_t sv = ctx->side[gi_now % NSS];
      // ---- the group's tables; its SSV launch appends survivors (-> candidate table) and undecided pairs (-> exact-MSV table) itself ----
      CascadeDev cd = cd0;
      uint32_t *cnt = d_gcnt + (1 + g) * CC_SIZE;
      cd.cnt = cnt;
      cd.cand = d_cand + sb.o_cand; cd.cap_cand = sb.cap_cand;
      cd.bias_raw = d_bias + 2 * sb.o_cand; cd.vit_fast = d_vfast + sb.o_cand; cd.vit_exact = d_vexact + sb.o_cand; cd.vit_flag = d_vflag + sb.o_cand; cd.route = d_route + sb.o_cand;
      cd.vq = d_vq + sb.o_vq; cd.vxq = d_vxq + sb.o_vq; cd.cap_vq = sb.cap_cand;
      cd.fq = d_fq + sb.o_f; cd.bq = d_bq + sb.o_f; cd.cap_fq = sb.cap_f;
      cd.eq = d_eq + sb.o_e; cd.cap_eq = sb.cap_e;
      cd.rq = d_rq + sb.o_r; cd.cap_rq = sb.cap_r;
      cd.ensq = d_ensq + (size_t)ens_slot(part) * cp.rwork; cd.ensq_cnt = d_gcnt + CC_ENSQ + ens_slot(part);
      PairRec *nores = d_nores + sb.o_nores;
      const SsvEpi epi{lt, cd.cand, cnt + CC_CAND, sb.cap_cand, nores, cnt + CC_NORES, sb.cap_nores, nullptr};
      if (launch_ssv(sb.Q, (int)sb.nblocks, ssv_threads_for(sb.Q), sv, ctx->work.as<SsvBlockWork>() + sb.first, dm, res, off, dlen,
                     s->d_order.as<uint32_t>(), epi))
        throw Error(CKM_ERANGE, "no SSV kernel instance for this model length");
      st.ssv_launches++;
      hipStream_t sc = sv;
      if (NCH > 0) { sc = ctx->side[NSS + gi_now % NCH]; HIPCHK(hipEventRecord(ctx->grp_ev[g], sv)); HIPCHK(hipStreamWaitEvent(sc, ctx->grp_ev[g], 0)); }
      // ---- the group's chain ----
      if (stop >= 2) {
        // exact MSV of the pairs SSV could not decide: packed, four pairs per wavefront, for every model that has a 16-lane image (an 8-lane
        // class 100 + Q8 holds models of exactly the 16-lane class ceil(Q8 / 2)); the wave-per-pair kernel for models beyond 2048 nodes
        const WorkQueue qn{nullptr, cnt + CC_NORES, sb.cap_nores};
        if (sb.Q == kSsvNone) launch_msv_full(sc, GRID_MSV, qn, nores, dm, lt, res, off, dlen, nullptr, nullptr, std::max(64, sb.maxM), &cd);
        else if (launch_msv16(sb.Q >= 100 ? (sb.Q - 100 + 1) / 2 : sb.Q, GRID_MSV, sc, qn, nores, dm, lt, res, off, dlen, cd)) throw Error(CKM_ERANGE, "no exact-MSV kernel instance for this model length");
      }
      if (stop >= 3) launch_bias_filter(sc, GRID_MSV, cd, dm, lt, res, off);
      int rc = 0;
      // FAST filter of every class first (short models: four pairs per wavefront on 16 lanes each), then the exact kernel for the pairs
      // whose bound did not decide -- those of the 16-lane classes join the wave-per-pair queue of their model's class
      for (int c = NVC - 1; c >= 0; --c) if (sb.vit[c] && stop >= 4) {
        const WorkQueue qv{cd.vq + (size_t)c * cd.cap_vq, cnt + CC_VQ + c, cd.cap_vq};
        if (c < NV16) rc |= launch_vit16(kVit16Q[c], GRID_VIT, sc, qv, cd.cand, dm, lt, res, off, dlen, cd);
        else rc |= launch_vit(kVitQH[c - NV16], GRID_VIT, sc, qv, cd.cand, dm, lt, res, off, dlen, nullptr, nullptr, nullptr, true, &cd);
      }
      for (int c = NVC - 1; c >= NV16; --c) if (sb.vit[c] && stop >= 5)
        rc |= launch_vit(kVitQH[c - NV16], std::max(64u, GRID_VIT / 4), sc, WorkQueue{cd.vxq + (size_t)c * cd.cap_vq, cnt + CC_VXQ + c, cd.cap_vq}, cd.cand, dm, lt, res, off, dlen, nullptr, nullptr, nullptr, false, &cd);
      for (int c = NFC - 1; c >= 0; --c) if (sb.fb[c]) {
        const int Q = kFbQ[c];
        const WorkQueue qf{cd.fq + (size_t)c * sb.cap_f, cnt + CC_FQ + c, sb.cap_f}, qb{cd.bq + (size_t)c * sb.cap_f, cnt + CC_BQ + c, sb.cap_f};
        const WorkQueue qe{cd.eq + (size_t)c * sb.cap_e, cnt + CC_EQ + c, sb.cap_e}, qr{cd.rq + (size_t)c * sb.cap_r, cnt + CC_RQ + c, sb.cap_r};
        // one launch per stage (a fused Forward -> F3 -> Backward -> regions kernel was measured in round 2: same rows, 2-3 % slower -- it
        // holds 1.5-2x the registers and the launch boundaries it removes are hidden underneath the SSV launches -- and removed in round 3)
        if (stop >= 6) rc |= launch_fwd(Q, GRID_FB, sc, qf, cd.fwork, dm, lt, res, off, ws, d_fout_f, d_events_f, d_gcnt + CC_EVENTS, cp.events_f, &cd);
        if (stop >= 7) rc |= launch_bwd(Q, GRID_FB, sc, qb, cd.fwork, dm, lt, res, off, ws, d_fout_f, nullptr);
        if (stop >= 8) launch_regions(sc, 1024, qb.list, qb.count, sb.cap_f, cd.fwork, cd, dm, ws);
        if (stop >= 9) rc |= launch_fwd(Q, GRID_FB, sc, qe, cd.ework, dm, lt, res, off, ws, d_fout_e, d_events_e, d_gcnt + CC_EVENTS_E, cp.events_e, nullptr);
        if (stop >= 10) rc |= launch_bwd(Q, GRID_FB, sc, qe, cd.ework, dm, lt, res, off, ws, d_fout_e, d_rerr_e);
        if (stop >= 11) rc |= launch_oa(Q, GRID_FB, sc, qe, cd.ework, dm, ws, d_rerr_e, d_fout_e, d_envout);
        if (stop >= 12) rc |= launch_fwd(Q, std::max(64u, GRID_FB / 8), sc, qr, cd.rwork, dm, lt, res, off, ws, d_fout_r, d_events_r, d_gcnt + CC_EVENTS_R, 1 << 16, nullptr);
      }
      if (rc) throw Error(CKM_ERANGE, "no kernel instance for this model length");
    }
    if (cur_part >= 0) flush_part(cur_part);
    if (ens_launched) HIPCHK(hipEventRecord(ctx->ens_ev[16], es));
    ctx->ens_pending = ens_launched;
  }
  // the SSV streams join the main stream first (end of the lane's SSV phase: the next lane may start its own), then the chains
  for (int k = 0; k < NSS; ++k) { HIPCHK(hipEventRecord(ctx->cls_ev[k], ctx->side[k])); HIPCHK(hipStreamWaitEvent(ms, ctx->cls_ev[k], 0)); }
  HIPCHK(hipEventRecord(ctx->ev[1], ms));                    // ev[0]..ev[1] brackets the lane's SSV launches (with NCH == 0: and the chains queued behind them)
  HIPCHK(hipEventRecord(ctx->cev[1], ms));
  if (baton_lock.owns_lock()) { baton.k ^= 1; HIPCHK(hipEventRecord(baton.ev[baton.k], ms)); baton.recorded = true; baton_lock.unlock(); }
  {
    std::unique_lock<std::mutex> lock(owner->ssv_mutex);
    owner->ssv_prev_done = ctx->cev[1];
    took_turn = true; owner->ssv_turn++; owner->ssv_cv.notify_all();
  }
  for (int k = NSS; k < NS; ++k) { HIPCHK(hipEventRecord(ctx->cls_ev[k], ctx->side[k])); HIPCHK(hipStreamWaitEvent(ms, ctx->cls_ev[k], 0)); }
  if (ctx->ens_pending) { HIPCHK(hipStreamWaitEvent(ms, ctx->ens_ev[16], 0)); ctx->ens_pending = false; }
  // ---- counters last ----
  HIPCHK(hipMemcpyAsync(h_cnt, d_gcnt, (NG + 1) * CC_SIZE * sizeof(uint32_t), hipMemcpyDeviceToHost, ms));
  unsigned long long *h_tops = pin_table<unsigned long long>(ctx->h_tops, 4);
  HIPCHK(hipMemcpyAsync(h_tops, d_tops, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ms));
  HIPCHK(hipGetLastError());
  CKM_TRACE_PT("chain queued");
  HIPCHK(hipStreamSynchronize(ms));                           // ---- the one synchronisation of the lane ----
  CKM_TRACE_PT("chain drained");
  if (stop < 99) {
    fprintf(stderr, "ckm-chain w%d stop=%d:", ctx->id, stop);
    for (int k = 0; k < CC_END; ++k) { uint64_t v = h_cnt[k]; for (size_t g = 0; g < NG; ++g) v += h_cnt[(1 + g) * CC_SIZE + k]; if (v) fprintf(stderr, " c%d=%llu", k, (unsigned long long)v); }
    fprintf(stderr, "\n");
    return 1;
  }
  { float msv = 0.f; HIPCHK(hipEventElapsedTime(&msv, ctx->ev[0], ctx->ev[1])); st.ms_ssv = msv; }
  const double t_host0 = now_ms();
  st.ms_filters = t_host0 - t_start;          // (queueing + the whole device chain: the stages are no longer separable by host clocks)

  // ---- did everything fit? ----
  const uint32_t n_fwork = h_cnt[CC_FWORK], n_ework = h_cnt[CC_EWORK], n_rwork = h_cnt[CC_RWORK],
                 n_pass = h_cnt[CC_PASS], n_reg = h_cnt[CC_REG], n_evf = h_cnt[CC_EVENTS], n_eve = h_cnt[CC_EVENTS_E], status = h_cnt[CC_STATUS];
  bool fits = status == 0;
  const bool tr = getenv("CKM_TRACE") != nullptr;
  auto over = [&](const char *what, int g, int k, uint64_t n, uint64_t cap) {
    fits = false; if (tr) fprintf(stderr, "ckm-trace w%d table %s (group %d, class %d) wanted %llu of %llu\n", ctx->id, what, g, k, (unsigned long long)n, (unsigned long long)cap); };
  auto need = [&](const char *what, uint32_t &cap, uint32_t n) { if (n > cap) { over(what, -1, -1, n, cap); cap = (uint32_t)std::min<uint64_t>((uint64_t)n + n / 4 + 1024, 0xfffffff0ull); } };
  cp.seen_rwork = std::max(cp.seen_rwork, n_rwork);
  need("fwork", cp.fwork, n_fwork); need("ework", cp.ework, n_ework); need("rwork", cp.rwork, n_rwork); need("pass", cp.pass, n_pass);
  need("reg", cp.reg, n_reg); need("events_f", cp.events_f, n_evf); need("events_e", cp.events_e, n_eve);
  // A per-group table that overflowed is sized from what the group ASKED for (round 6): its divisor follows the observed share of the
  // pairs and its floor the observed count (+25 %), so that the same kind of search fits the next time.  Rounds 2-5 halved the divisor
  // once per search -- which changes nothing where the floor is the larger term: on inputs with many multi-domain regions (paralog
  // families, low-complexity proteins: bench.py's hard_workload) the 256-entry region queues overflowed in EVERY search, and every search
  // paid the host-driven cascade (profiles/r06q_hard_workload_overflows.txt).
  // (the divisor still halves at most once per search, and only where it was the larger term: a small group's share says nothing about
  //  the large ones -- regions follow the domains present, not the pairs scanned)
  uint32_t halved = 0;
  auto learn = [&](int kind, uint32_t &div, uint32_t &floor_, uint64_t pairs, uint64_t want) {
    const uint64_t w = want + want / 4 + 16;
    if ((pairs / std::max<uint64_t>(1, (uint64_t)div * shrink) >= floor_ || shrink > 1) && !((halved >> kind) & 1u)) { div = std::max<uint32_t>(1, div / 2); halved |= 1u << kind; }
    if (shrink == 1) floor_ = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(floor_, w), 1u << 14);
  };
  uint64_t n_cand = 0, n_nores = 0;
  for (size_t g = 0; g < NG; ++g) {
    const uint32_t *c = h_cnt + (1 + g) * CC_SIZE; const Sub &sb = subs[g];
    n_cand += c[CC_CAND]; n_nores += c[CC_NORES];
    if (c[CC_CAND] > sb.cap_cand) { over("cand", (int)g, -1, c[CC_CAND], sb.cap_cand); learn(0, cp.div_cand, cp.fl_cand, sb.pairs, c[CC_CAND]); }
    if (c[CC_NORES] > sb.cap_nores) { over("nores", (int)g, -1, c[CC_NORES], sb.cap_nores); if (sb.Q != kSsvNone) learn(1, cp.div_nores, cp.fl_nores, sb.pairs, c[CC_NORES]); }
    for (int k = 0; k < NVC; ++k) {
      st.pairs_vit += c[CC_VQ + k]; st.pairs_vit_exact += c[CC_VXQ + k];
      const uint32_t v = std::max(c[CC_VQ + k], c[CC_VXQ + k]);
      if (v > sb.cap_cand) { over("vq", (int)g, k, v, sb.cap_cand); learn(0, cp.div_cand, cp.fl_cand, sb.pairs, v); }
    }
    for (int k = 0; k < NFC; ++k) {
      const uint32_t f = std::max(c[CC_FQ + k], c[CC_BQ + k]);
      if (f > sb.cap_f) { over("fq", (int)g, k, f, sb.cap_f); learn(2, cp.div_fwork, cp.fl_fwork, sb.pairs, f); }
      if (c[CC_EQ + k] > sb.cap_e) { over("eq", (int)g, k, c[CC_EQ + k], sb.cap_e); learn(3, cp.div_ework, cp.fl_ework, sb.pairs, c[CC_EQ + k]); }
      if (c[CC_RQ + k] > sb.cap_r) { over("rq", (int)g, k, c[CC_RQ + k], sb.cap_r); learn(4, cp.div_rwork, cp.fl_rwork, sb.pairs, c[CC_RQ + k]); }
    }
  }
  if (status & CS_RWORK) cp.hens *= 2;
  // zone 2 ran out (regions were deferred to the host): size the workspace from what this search asked for, for the next calls
  st.ws_cap_bytes = ctx->ws.cap; st.ws_used_bytes = (std::min<uint64_t>(h_tops[0], cd0.ws_cap) + h_tops[2]) * 4;
  // the bump allocator counts every request, granted or not: after an overflow the search's whole demand is known, and the next one of its
  // kind gets that (+15 %) instead of a blind x1.5
  if (h_tops[2] > cd0.ws2_cap)
    cp.ws_per_cell = std::max(cp.ws_per_cell, (float)(1.15 * (double)(h_tops[2] + cd0.ws_cap) * 4.0 / std::max(cell_sum, 1.0)));
  if (tr) fprintf(stderr, "ckm-trace w%d workspace: %.3f GB held, %.3f GB asked for by this search (zone 1 %.3f of %.3f, zone 2 %.3f of %.3f), %.3e cells -> %.2f B per cell (estimate %.2f)\n", ctx->id,
                  (double)ctx->ws.cap / 1e9, (double)(h_tops[0] + h_tops[2]) * 4 / 1e9, (double)h_tops[0] * 4 / 1e9, (double)cd0.ws_cap * 4 / 1e9, (double)h_tops[2] * 4 / 1e9,
                  (double)cd0.ws2_cap * 4 / 1e9, cell_sum, (double)(h_tops[0] + h_tops[2]) * 4 / std::max(cell_sum, 1.0), (double)cp.ws_per_cell);
  if (!fits) {
    if (getenv("CKM_TRACE")) fprintf(stderr, "ckm-trace w%d device cascade did not fit (status 0x%x): host-driven cascade for this lane\n", ctx->id, status);
    return 1;
  }
  st.pairs_msv_full = n_nores; st.pairs_bias = n_cand; st.pairs_fwd = n_fwork; st.pairs_dom = n_pass; st.regions_multi = n_rwork;
  const CascadeDev &cd = cd0;

  // ---- the used prefixes of the result tables come over in one batch of copies (second and last wait of the chain) ----
  const uint64_t n_hens = std::min<uint64_t>(h_tops[1], cp.hens);
  PassRec *h_pass = pin_table<PassRec>(ctx->h_pass, n_pass);
  RegionRec *h_reg = pin_table<RegionRec>(ctx->h_reg, n_reg);
  EnvOut *h_envout = pin_table<EnvOut>(ctx->h_envout, n_ework);
  ScaleEvent *h_events_f = pin_table<ScaleEvent>(ctx->h_events_f, n_evf), *h_events_e = pin_table<ScaleEvent>(ctx->h_events_e, n_eve);
  float *h_hens = pin_table<float>(ctx->h_hens, n_hens);
  // (on the priority stream, which is a hardware queue of this context's own: the normal-priority streams of all the contexts of a process
  //  share 16 hardware queues, and whichever of them `ms` shares with another context's streams is held by that context's queued SSV
  //  launches and waiting chain kernels for its whole SSV phase -- measured: these six copies ran 214 ms after the drain, at the end of
  //  the other context's SSV phase, and this context was then 77 ms late for its own turn, every turn: profiles/r04n_timeline_cfg3.txt.
  //  The host has just waited for `ms`, so the copies need no event.)
  hipStream_t rs = ctx->late[0];
  if (n_pass) HIPCHK(hipMemcpyAsync(h_pass, cd.h_pass, (size_t)n_pass * sizeof(PassRec), hipMemcpyDeviceToHost, rs));
  if (n_reg) HIPCHK(hipMemcpyAsync(h_reg, cd.h_reg, (size_t)n_reg * sizeof(RegionRec), hipMemcpyDeviceToHost, rs));
  if (n_ework) HIPCHK(hipMemcpyAsync(h_envout, d_envout, (size_t)n_ework * sizeof(EnvOut), hipMemcpyDeviceToHost, rs));
  if (n_evf) HIPCHK(hipMemcpyAsync(h_events_f, d_events_f, (size_t)n_evf * sizeof(ScaleEvent), hipMemcpyDeviceToHost, rs));
  if (n_eve) HIPCHK(hipMemcpyAsync(h_events_e, d_events_e, (size_t)n_eve * sizeof(ScaleEvent), hipMemcpyDeviceToHost, rs));
  if (n_hens) HIPCHK(hipMemcpyAsync(h_hens, d_hens, (size_t)n_hens * sizeof(float), hipMemcpyDeviceToHost, rs));
  HIPCHK(hipStreamSynchronize(rs));
  CKM_TRACE_PT("results copied");
  const PassRec *pass = h_pass;
  const RegionRec *reg = h_reg;
  // ---- ensembles first: their clustered domains are the only envelopes the device has not rescored yet.  Parse the exported traces of
  // every multi-domain region, cluster them on the host threads, and hand the resulting envelopes to a helper thread that drives the
  // short second round (Forward / Backward / OA of a few dozen envelopes) while this thread takes the exact decisions below. ----
  std::vector<RegionRes> pre(n_reg);                                    // by region record; nseg stays empty where nothing was exported
  std::vector<uint8_t> pre_overflow(n_reg, 0);
  {
    std::vector<uint32_t> multi_rec;
    for (uint32_t ro = 0; ro < n_reg; ++ro) {
      const RegionRec &rr = reg[ro];
      if (!rr.multi || rr.pass >= n_pass) continue;
      if (rr.target == 0xffffffffu || (rr.target != REGION_DEFERRED && rr.pad == 0xffffffffu)) return 1;   // no table entry / no export slot (status would have said so)
      if (rr.target != REGION_DEFERRED) multi_rec.push_back(ro);
    }
    pool_run(ctx, multi_rec.size(), 1, [&](size_t lo, size_t hi) {
      for (size_t k = lo; k < hi; ++k) {
        const uint32_t ro = multi_rec[k];
        const RegionRec &rr = reg[ro]; RegionRes &o = pre[ro];
        const int Ld = rr.j - rr.i + 1, cap = std::min(Ld, 16);
        const float *raw = h_hens + rr.pad;
        const int32_t *ns = reinterpret_cast<const int32_t *>(raw);
        const int32_t *sg = reinterpret_cast<const int32_t *>(raw + 256);
        bool overflow = false;
        for (int t = 0; t < ENS_NSAMPLES; ++t) overflow |= ns[t] < 0;
        if (overflow) { pre_overflow[ro] = 1; continue; }               // a trace needed more segment slots: the host-driven ensemble repeats the region
        o.cap = cap; o.nseg.assign(ns, ns + ENS_NSAMPLES); o.segs.assign((size_t)ENS_NSAMPLES * cap, Seg{0, 0, 0, 0});
        for (int t = 0; t < ENS_NSAMPLES; ++t)
          for (int d = 0; d < ns[t]; ++d) {                             // the device walks backwards: last domain first
            const int32_t *q4 = sg + ((size_t)t * cap + (ns[t] - 1 - d)) * 4;
            o.segs[(size_t)t * cap + d] = Seg{q4[0], q4[1], q4[2], q4[3]};
          }
        const float *n2 = raw + 256 + (size_t)ENS_NSAMPLES * cap * 4;
        o.n2sum.assign(n2, n2 + Ld);
        cluster_ensemble(o);
      }
    });
  }
  std::vector<EnvReq> early_req; std::vector<EnvRes> early_res;
  std::vector<int64_t> early_first(n_reg, -1);                          // region record -> first of its envelopes in early_req
  for (uint32_t ro = 0; ro < n_reg; ++ro) {
    if (pre[ro].nseg.empty()) continue;
    const RegionRec &rr = reg[ro];
    early_first[ro] = (int64_t)early_req.size();
    for (const Seg &e : pre[ro].env) early_req.push_back({pass[rr.pass].model, pass[rr.pass].seq, e.sqfrom + rr.i - 1, e.sqto + rr.i - 1});
  }
  if (getenv("CKM_TRACE")) {
    size_t nm = 0, nl = 0;
    for (uint32_t ro = 0; ro < n_reg; ++ro) if (reg[ro].multi && reg[ro].pass < n_pass) { ++nm; nl += s->len[pass[reg[ro].pass].seq] > Lcut; }
    fprintf(stderr, "ckm-trace w%d %zu multi-domain regions, %zu of them on sequences of the long part (L > %d); %zu ensemble envelopes\n", ctx->id, nm, nl, Lcut, early_req.size());
  }
  CKM_TRACE_PT("clustering done");
  std::exception_ptr early_err;
  // (only when another search is in flight on the device: alone, the normal streams run the round's register classes side by side)
  struct LateGuard { Worker *w; ~LateGuard() { w->late_round = false; } } late_guard{ctx};
  ctx->late_round = late_on() && baton.searches.load() > 1;
  std::thread early_thread;
  if (!early_req.empty())
    early_thread = std::thread([&] {
      try { HIPCHK(hipSetDevice(ctx->device)); rescore_envelopes(ctx, p, s, early_req, early_res); } catch (...) { early_err = std::current_exception(); }
    });
  struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } early_joiner{early_thread};

  // ---- the filter decisions again, exactly (host libm), for the pairs the device let through ----
  EventIndex fev; fev.build(h_events_f, n_evf, n_fwork);
  EventIndex eev; eev.build(h_events_e, n_eve, n_ework);
  std::vector<uint8_t> alive(n_pass, 0); std::vector<float> fwdsc(n_pass, 0.f);
  std::atomic<int> inconsistent{0};
  pool_run(ctx, n_pass, 256, [&](size_t lo, size_t hi) {
    for (size_t k = lo; k < hi; ++k) {
      const PassRec &r = pass[k];
      const HostProfile &hp = p->prof[r.model];
      const int L = s->len[r.seq];
      const LenEntry &le = s->lentab[L];
      const float p1 = (float)L / (float)(L + 1);
      const float nullb = (float)(log((double)r.bias_d) + (double)r.bias_e * kLn2);
      const float filtersc = nullb + (float)L * logf(p1) + logf(1.0f - p1);
      const float sc = bits(r.usc, filtersc);
      if (!(sc >= hp.thr_msv_f1)) continue;
      if (!(sc >= hp.thr_msv_f2)) {                    // the Viterbi filter applies
        const uint32_t route = r.route & 0x0fu;
        if (route == 0) { inconsistent++; continue; }   // the device skipped it with a margin the exact test contradicts: cannot happen
        if (route == 1) {
          if (!(bits(r.vit_fast, filtersc) >= hp.thr_vit_f2)) { if (r.vit_flag) inconsistent++; continue; }   // (a bound that fails with its flag set would have been re-run)
        } else if (!(bits(r.vit_exact, filtersc) >= hp.thr_vit_f2)) continue;
      }
      const float f = finish_forward(r.fwd_xC, le.move_m, fev.scales(r.fwork));
      if (!(bits(f, filtersc) >= hp.thr_fwd_f3)) continue;
      alive[k] = 1; fwdsc[k] = f;
    }
  });
  if (inconsistent.load()) throw Error(CKM_EHIP, "device-side filter decision contradicts the exact one (margin too small): please report");

  // ---- regions by pair, in sequence order; envelopes of single-domain regions are already rescored ----
  std::vector<uint32_t> rorder(n_reg);
  std::iota(rorder.begin(), rorder.end(), 0u);
  std::sort(rorder.begin(), rorder.end(), [&](uint32_t a, uint32_t b) { return reg[a].pass != reg[b].pass ? reg[a].pass < reg[b].pass : reg[a].i < reg[b].i; });
  DomStage ds;
  std::vector<int32_t> dsidx(n_pass, -1);
  for (uint32_t k = 0; k < n_pass; ++k) if (alive[k]) { dsidx[k] = (int32_t)ds.pass.size(); ds.pass.push_back({pass[k].model, pass[k].seq, fwdsc[k]}); }
  ds.nregions.assign(ds.pass.size(), 0);
  ds.env_of_pass.assign(ds.pass.size(), {0, 0});
  std::vector<RegionReq> redo_req; std::vector<size_t> redo_at;          // ensembles that need more segment slots or found no workspace: host-driven
  size_t n_deferred = 0;
  std::vector<uint32_t> item_rec;                                        // region record of every item of ds.items
  for (uint32_t ro : rorder) {
    const RegionRec &rr = reg[ro];
    if (rr.pass >= n_pass || dsidx[rr.pass] < 0) continue;
    const uint32_t q = (uint32_t)dsidx[rr.pass];
    if (rr.target == 0xffffffffu) return 1;                            // no table entry for it on the device (status would have said so)
    ds.nregions[q]++;
    item_rec.push_back(ro);
    if (!rr.multi) { ds.items.push_back({q, rr.i, rr.j, -1}); continue; }
    ds.items.push_back({q, rr.i, rr.j, (int)ds.regres.size()});
    if (rr.target == REGION_DEFERRED || pre_overflow[ro]) {
      redo_req.push_back({pass[rr.pass].model, pass[rr.pass].seq, rr.i, rr.j}); redo_at.push_back(ds.regres.size());
      if (rr.target == REGION_DEFERRED) ++n_deferred;
      ds.regres.emplace_back();
    } else ds.regres.push_back(std::move(pre[ro]));
  }
  // envelopes in pair order (as the host-driven cascade lists them); source of each: >= 0 index into h_envout (rescored by the chain),
  // <= -2 index -2 - k into early_res (ensemble envelope, second round under way), -1 still to be rescored (deferred / repeated regions)
  std::vector<int64_t> env_src;
  auto list_envelopes = [&]() {
    ds.envreq.clear(); ds.env_region.clear(); env_src.clear();
    size_t it = 0;
    for (size_t q = 0; q < ds.pass.size(); ++q) {
      ds.env_of_pass[q].first = ds.envreq.size();
      for (; it < ds.items.size() && ds.items[it].pass == q; ++it) {
        const DomItem &im = ds.items[it];
        const uint32_t ro = item_rec[it];
        const RegionRec &rr = reg[ro];
        if (im.region < 0) {
          ds.envreq.push_back({ds.pass[q].model, ds.pass[q].seq, im.i, im.j}); ds.env_region.push_back(-1);
          env_src.push_back(rr.target == REGION_DEFERRED ? -1 : (int64_t)rr.target);          // deferred: rescored with the late round, in workspace-sized batches
          continue;
        }
        size_t k = 0;
        for (const Seg &e : ds.regres[im.region].env) {
          const int i2 = e.sqfrom + im.i - 1, j2 = e.sqto + im.i - 1;
          ds.envreq.push_back({ds.pass[q].model, ds.pass[q].seq, i2, j2}); ds.env_region.push_back(im.region);
          env_src.push_back(early_first[ro] >= 0 && !pre_overflow[ro] && rr.target != REGION_DEFERRED ? -2 - (early_first[ro] + (int64_t)k) : -1);
          ++k;
        }
      }
      ds.env_of_pass[q].second = ds.envreq.size() - ds.env_of_pass[q].first;
    }
  };
  for (uint32_t ro = 0; ro < n_reg; ++ro) if (!reg[ro].multi && reg[ro].target == REGION_DEFERRED && reg[ro].pass < n_pass && dsidx[reg[ro].pass] >= 0) ++n_deferred;
  list_envelopes();
  ds.envres.resize(ds.envreq.size());
  // first-round results while the second round runs
  pool_run(ctx, ds.envreq.size(), 64, [&](size_t lo, size_t hi) {
    for (size_t e = lo; e < hi; ++e) {
      if (env_src[e] < 0) continue;
      const EnvOut &eo = h_envout[env_src[e]];
      EnvRes &o = ds.envres[e];
      const LenEntry &le = s->lentab[s->len[ds.envreq[e].seq]];
      o.ok = eo.range_err == 0;
      o.xC = eo.xC; o.nscale = eo.nscale;
      o.envsc = finish_forward(eo.xC, le.move_u, eev.scales((uint32_t)env_src[e]));
      o.oasc = eo.oasc; o.hmm_from = eo.hmm_from; o.hmm_to = eo.hmm_to; o.ali_from = eo.ali_from; o.ali_to = eo.ali_to;
      for (int x = 0; x < K; ++x) o.null2[x] = eo.null2[x];
    }
  });
  CKM_TRACE_PT("decisions done");
  if (early_thread.joinable()) early_thread.join();
  if (early_err) std::rethrow_exception(early_err);
  CKM_TRACE_PT("second round done");
  if (!redo_req.empty()) {                                              // rare: regions the device could not take (workspace, segment slots)
    std::vector<RegionRes> r2;
    run_ensembles(ctx, p, s, redo_req, r2);
    for (size_t k = 0; k < redo_at.size(); ++k) ds.regres[redo_at[k]] = std::move(r2[k]);
    list_envelopes();                                                   // the repeated regions now have envelopes: same order, more entries
    std::vector<EnvRes> keep(ds.envreq.size());
    {
      // results computed above stay valid for the entries whose source did not change: redo the cheap conversion instead of tracking moves
      for (size_t e = 0; e < ds.envreq.size(); ++e) {
        if (env_src[e] < 0) continue;
        const EnvOut &eo = h_envout[env_src[e]];
        EnvRes &o = keep[e];
        const LenEntry &le = s->lentab[s->len[ds.envreq[e].seq]];
        o.ok = eo.range_err == 0;
        o.xC = eo.xC; o.nscale = eo.nscale;
        o.envsc = finish_forward(eo.xC, le.move_u, eev.scales((uint32_t)env_src[e]));
        o.oasc = eo.oasc; o.hmm_from = eo.hmm_from; o.hmm_to = eo.hmm_to; o.ali_from = eo.ali_from; o.ali_to = eo.ali_to;
        for (int x = 0; x < K; ++x) o.null2[x] = eo.null2[x];
      }
    }
    ds.envres.swap(keep);
  }
  {
    std::vector<EnvReq> second; std::vector<size_t> second_at;
    for (size_t e = 0; e < ds.envreq.size(); ++e) {
      if (env_src[e] >= 0) continue;
      if (env_src[e] <= -2) { ds.envres[e] = early_res[(size_t)(-2 - env_src[e])]; continue; }
      second.push_back(ds.envreq[e]); second_at.push_back(e);
    }
    if (!second.empty()) {                                             // late round: deferred regions and the envelopes of repeated ensembles
      std::vector<EnvRes> r2;
      rescore_envelopes(ctx, p, s, second, r2);
      for (size_t k = 0; k < second.size(); ++k) ds.envres[second_at[k]] = r2[k];
    }
  }
  st.envelopes = ds.envreq.size();
  if (n_deferred && getenv("CKM_TRACE")) fprintf(stderr, "ckm-trace w%d %zu regions found no device workspace and were rescored by the host-driven rounds\n", ctx->id, n_deferred);
  st.ms_domains = now_ms() - t_host0;
  CKM_TRACE_PT("envelopes done");
  const double t_rows0 = now_ms();
  assemble_hits(ctx, p, s, ds, by_bin_model);
  st.ms_host = now_ms() - t_rows0;
  st.ms_total = now_ms() - t_start;
  CKM_TRACE_PT("cascade done");
  return 0;
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
  for (uint32_t b = 0; b < nbins; ++b) for (uint32_t m : bin_models[b]) if (p->too_long[m])
    throw Error(CKM_ERANGE, "model " + p->hmm[m].name + " (LENG " + std::to_string(p->hmm[m].M) + ") is selected for bin " + std::to_string(b) +
                            " but is longer than the 4096 nodes the kernels are instantiated for (DESIGN.md section 8); leave it out of the bin's model list");
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
  // cut lengths, descending: class k holds the sequences with cut[k-1] >= L > cut[k]: the cuts that give the classes fixed shares of
  // the residues (measured best on cfg2 in round 1: 16 / 45 / 39 % for three classes, 60 / 40 for two).
  const int ncl = nw > 1 ? c->nclasses : 1, ngr = 1;
  std::vector<int> cuts;
  if (ncl >= 2) {
    std::vector<double> sh;
    if (ncl == 2) sh = {0.60}; else if (ncl == 3) sh = {0.16, 0.61}; else if (ncl == 4) sh = {0.10, 0.35, 0.65};
    else { for (int k = 1; k < ncl; ++k) sh.push_back((double)k / ncl); }
    std::vector<uint64_t> by_len((size_t)s->maxL + 2, 0);
    uint64_t total = 0;
    for (uint32_t i = 0; i < s->nseq; ++i) { by_len[s->len[i]] += (uint64_t)s->len[i]; total += (uint64_t)s->len[i]; }
    uint64_t acc = 0; int k = 0;
    for (int L = s->maxL; L >= 1 && k < ncl - 1; --L) { acc += by_len[L]; if ((double)acc >= sh[k] * (double)total) { cuts.push_back(L - 1); ++k; } }
    while ((int)cuts.size() < ncl - 1) cuts.push_back(0);
  }
  if (nw >= 2 && (int)cuts.size() == ncl - 1 && (ncl == 1 || cuts.back() > 0)) {
    // bin groups: contiguous, about equal residue counts
    std::vector<int> group_of(nbins, 0);
    {
      uint64_t tot = 0, acc = 0; for (uint32_t b = 0; b < nbins; ++b) tot += s->bin_res[b];
      for (uint32_t b = 0; b < nbins; ++b) { group_of[b] = std::min(ngr - 1, (int)((double)acc * ngr / std::max<double>(1.0, (double)tot))); acc += s->bin_res[b]; }
    }
    for (int k = 0; k < nw; ++k) { chunk[k] = active; ranges[k].lo.resize(nbins); ranges[k].hi.resize(nbins); ranges[k].res.assign(nbins, 0); ranges[k].tag = 1000 + (uint64_t)k + 64 * (uint64_t)ngr; for (int cv : cuts) ranges[k].tag = ranges[k].tag * 4099 + (uint64_t)cv; }
    for (uint32_t b = 0; b < nbins; ++b) {
      uint32_t at = s->order_off[b];
      for (int cl = 0; cl < ncl; ++cl) {
        const int cut = (cl < ncl - 1) ? cuts[cl] : -1;
        uint64_t r = 0; const uint32_t lo = at;
        while (at < s->order_off[b + 1] && s->len[s->order[at]] > cut) { r += (uint64_t)s->len[s->order[at]]; ++at; }
        for (int g = 0; g < ngr; ++g) {
          const int k = g * ncl + cl;
          if (g == group_of[b]) { ranges[k].lo[b] = lo; ranges[k].hi[b] = at; ranges[k].res[b] = r; }
          else { ranges[k].lo[b] = lo; ranges[k].hi[b] = lo; ranges[k].res[b] = 0; }
        }
      }
    }
  } else {
    // models -> workers, by decreasing work, shares ~ ratio^k
    const double ratio = 1.0;
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
  trace_begin();
  // CKM_CASCADE=host keeps the host-driven cascade (a device phase, a copy and a host decision per stage) for comparison and as the
  // fallback of a lane whose tables or workspace the device-driven one outgrew
  const char *cascade_env = getenv("CKM_CASCADE");
  const bool host_cascade = cascade_env && !strcmp(cascade_env, "host");
  c->ssv_prev_done = nullptr;
  c->fallbacks = 0;
  auto run = [&](int k) {
    try {
      if (host_cascade) cascade(&c->w[k], c, k, p, s, ranges[k], chunk[k], model_bins, maps[k]);
      else if (cascade_dev(&c->w[k], c, k, p, s, ranges[k], chunk[k], model_bins, maps[k])) {
        c->fallbacks++;
        cascade(&c->w[k], c, k, p, s, ranges[k], chunk[k], model_bins, maps[k], true);      // (the attempt took and passed the lane's SSV turn, rc 1 or 2)
      }
    } catch (...) { errs[k] = std::current_exception(); }
  };
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
    st.ms_ssv += w.ms_ssv;                                   // the lanes' SSV phases run one behind the other: the sum is the kernel time
    st.ms_filters = std::max(st.ms_filters, w.ms_filters); st.ms_fwdbwd = std::max(st.ms_fwdbwd, w.ms_fwdbwd);
    st.ms_domains = std::max(st.ms_domains, w.ms_domains); st.ms_host = std::max(st.ms_host, w.ms_host);
  }
  st.cascade_fallback_lanes = (uint32_t)c->fallbacks.load();
  for (int k = 0; k < nw; ++k) { st.ws_cap_bytes = std::max(st.ws_cap_bytes, c->w[k].stats.ws_cap_bytes); st.ws_used_bytes = std::max(st.ws_used_bytes, c->w[k].stats.ws_used_bytes); }
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
        for (size_t d1 = 0; d1 < h.dom.size(); ++d1)          // HMMER's workaround of its bug #h74: every pair, sequence coordinates only
          for (size_t d2 = d1 + 1; d2 < h.dom.size(); ++d2) {
            Domain &a = h.dom[d1], &c = h.dom[d2];
            if (a.ali_from == c.ali_from && a.ali_to == c.ali_to) {
              Domain &w = (a.bitscore >= c.bitscore) ? c : a;
              if (w.reported) { w.reported = false; h.nreported--; }
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
    ctx->settle();                          // a workspace reservation still in flight (ckm_ctx_reserve)
    std::unique_ptr<ckm_hits> h(new ckm_hits());
    do_search(ctx, p, s, model_off, model_idx, E, domE, h.get());
    *out = h.release();
  });
}
