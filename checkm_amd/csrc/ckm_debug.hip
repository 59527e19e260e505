// ckm_debug.hip -- diagnostics entries used by the parity tests: every stage of chosen (model, sequence) pairs without filtering,
// envelope rescoring of chosen envelopes, the trace ensemble of a chosen region.
#include "ckm_host.h"

// ---- diagnostics -------------------------------------------------------------------------------------
extern "C" int ckm_debug_stages(ckm_ctx *ctx_, const ckm_profiles *p, const ckm_seqs *s, const uint32_t *model, const uint32_t *seq,
                                uint32_t npairs, ckm_stage_scores *out) {
  return guarded([&] {
    if (!ctx_ || !p || !s || !model || !seq || !out) throw Error(CKM_EINVAL, "NULL argument");
    for (uint32_t j = 0; j < npairs; ++j) if (model[j] < p->hmm.size() && p->too_long[model[j]]) throw Error(CKM_ERANGE, "model longer than the instantiated kernel classes");
    ctx_->settle();
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
      work[i].model = model[i]; work[i].list_start = i; work[i].count = 1; work[i].pair_start = i; byQ[ssv_class(p->prof[model[i]])].push_back(i);
    }
    std::vector<SsvBlockWork> sorted; std::vector<std::pair<int, std::pair<size_t, size_t>>> groups;
    for (auto &kv : byQ) { groups.push_back({kv.first, {sorted.size(), kv.second.size()}}); for (uint32_t i : kv.second) sorted.push_back(work[i]); }
    ctx->work.ensure(npairs * sizeof(SsvBlockWork)); ctx->idx.ensure(npairs * 4); ctx->maxv.ensure(npairs * 2 + 64);
    wcopy(ctx, ctx->work.p, sorted.data(), npairs * sizeof(SsvBlockWork), hipMemcpyHostToDevice);
    wcopy(ctx, ctx->idx.p, ids.data(), npairs * 4, hipMemcpyHostToDevice);
    SsvEpi epi; memset(&epi, 0, sizeof(epi));
    epi.lentab = lt; epi.maxv = ctx->maxv.as<uint16_t>();              // diagnostics: Smax per pair, no finish
    for (auto &g : groups)
      if (launch_ssv(g.first, (int)g.second.second, ssv_threads_for(g.first), ctx->stream, ctx->work.as<SsvBlockWork>() + g.second.first, dm, res, off, dlen,
                     ctx->idx.as<uint32_t>(), epi))
        throw Error(CKM_ERANGE, "no SSV kernel instance");
    HIPCHK(hipGetLastError());
    std::vector<uint16_t> maxv(npairs);
    HIPCHK(hipMemcpyAsync(maxv.data(), ctx->maxv.p, npairs * 2, hipMemcpyDeviceToHost, ctx->stream));
    // full MSV on every pair: first with the packed kernel the search uses (msv16_kernel, scores only), then with the wave-per-pair kernel
    std::vector<PairRec> pr(npairs);
    for (uint32_t i = 0; i < npairs; ++i) { pr[i].model = model[i]; pr[i].seq = seq[i]; pr[i].usc = 0; pr[i].filtersc = 0; }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::vector<float> uscp; std::vector<int32_t> xJp;
    run_msv_exact(ctx, p, s, pr, uscp, &xJp);
    ctx->cand.ensure(npairs * sizeof(PairRec)); ctx->fullx.ensure(npairs * 4); ctx->fullu.ensure(npairs * 4); ctx->raw.ensure(npairs * 12);
    HIPCHK(hipMemcpyAsync(ctx->cand.p, pr.data(), npairs * sizeof(PairRec), hipMemcpyHostToDevice, ctx->stream));
    std::map<int, std::vector<uint32_t>> vq;
    for (uint32_t i = 0; i < npairs; ++i) vq[p->prof[model[i]].vitQH].push_back(i);
    // queue lengths: the exact-MSV launch, then one per Viterbi register class
    std::vector<uint32_t> qctl(2 + 64, 0u);
    qctl[0] = npairs;
    { size_t k = 1; for (auto &kv : vq) { qctl[k] = (uint32_t)kv.second.size(); ++k; } }
    ctx->vitq.ensure(qctl.size() * 4);
    uint32_t *qd = ctx->vitq.as<uint32_t>();
    HIPCHK(hipMemcpyAsync(qd, qctl.data(), qctl.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    launch_msv_full(ctx->stream, std::min<uint32_t>(npairs, 4096), WorkQueue{nullptr, qd, npairs}, ctx->cand.as<PairRec>(), dm, lt, res, off, dlen,
                    ctx->fullx.as<int32_t>(), ctx->fullu.as<float>(), p->maxMp, nullptr);
    launch_bias(ctx->stream, ctx->cand.as<PairRec>(), npairs, dm, lt, res, off, dlen, ctx->raw.as<float>());
    HIPCHK(hipGetLastError());
    std::vector<int32_t> xJ(npairs); std::vector<float> usc(npairs), raw(npairs * 3);
    HIPCHK(hipMemcpyAsync(xJ.data(), ctx->fullx.p, npairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(usc.data(), ctx->fullu.p, npairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(raw.data(), ctx->raw.p, npairs * 12, hipMemcpyDeviceToHost, ctx->stream));
    // Viterbi on every pair
    std::vector<uint32_t> flat; std::vector<std::pair<int, std::pair<size_t, size_t>>> vg;
    for (auto &kv : vq) { vg.push_back({kv.first, {flat.size(), kv.second.size()}}); flat.insert(flat.end(), kv.second.begin(), kv.second.end()); }
    ctx->fbidx.ensure(npairs * 4); ctx->vitx.ensure(npairs * 4); ctx->vits.ensure(npairs * 4);
    HIPCHK(hipMemcpyAsync(ctx->fbidx.p, flat.data(), npairs * 4, hipMemcpyHostToDevice, ctx->stream));
    {
      size_t k = 1;
      for (auto &g : vg) {
        const uint32_t cnt = (uint32_t)g.second.second;
        if (launch_vit(g.first, std::min<uint32_t>((cnt + 3) / 4, 2048), ctx->stream, WorkQueue{ctx->fbidx.as<uint32_t>() + g.second.first, qd + k, cnt},
                       ctx->cand.as<PairRec>(), dm, lt, res, off, dlen, ctx->vitx.as<int32_t>(), ctx->vits.as<float>(), nullptr, false, nullptr))
          throw Error(CKM_ERANGE, "no Viterbi kernel instance");
        ++k;
      }
    }
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
    ctx_->settle();
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
    ctx_->settle();
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
