// ckm_genes.hip -- C ABI of gene calling on the device: nucleotide contigs of MANY bins in, genes and their proteins out.
// What it replaces: the two `prodigal -p single -m -f gff -g <11|4> -a genes.faa` runs per bin of checkm/prodigal.py:80-133 (`-p meta`,
// which CheckM uses below 100 kb, is NOT built: such a bin comes back untrained and the caller decides).
//
// One call = one translation table for a batch of bins; the pipeline is gene_pipe.h (nodes resident on the device from the codon flags to
// the gene records; the host takes the logarithms of the training tables and nothing else), its cooperating kernels kernels_genes.hip, its
// per-thread arithmetic gene_dev.h.  The arithmetic follows oracle/gene_full.c (the restatement of Prodigal 2.6.3's single-genome mode;
// parity unpinned) operation by operation; tests/test_gpu_genes.py compares gene for gene and score for score, tests/test_gene_emu.py
// runs the same pipeline source through a host executor on the CPU.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>
#include <memory>
#include <vector>
#include "ckm_host.h"
#include "gene_pipe.h"

struct ckm_genes {
  ckm::gene::GeneResult r;
  double ms_nodes = 0, ms_host = 0;
};

namespace {
// Several gene-calling calls may run side by side on one context (the pipeline is latency-bound: a workgroup per bin in the dynamic
// programs, a thread per contig in the trace-back walks -- throughput comes from calls in flight, checkm_amd/geneFinder.py keeps several):
// each takes a stream of its own from this per-device pool for its duration.
// A slot also keeps the call's two page-locked exchange areas (gene_exec.h: g_host_reserve) from call to call: page-locking tens of
// megabytes costs milliseconds.
struct GeneSlot { hipStream_t st = nullptr; bool busy = false; PinnedBuf pin[2]; };
struct GeneStreams {
  std::mutex m; std::deque<GeneSlot> s[16];
  static GeneStreams &get() { static GeneStreams g; return g; }
  GeneSlot *take(int dev) {
    std::lock_guard<std::mutex> lock(m);
    for (auto &p : s[dev & 15]) if (!p.busy) { p.busy = true; return &p; }
    hipStream_t st = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    s[dev & 15].emplace_back();
    GeneSlot &g = s[dev & 15].back();
    g.st = st; g.busy = true;
    return &g;
  }
  void give(GeneSlot *g) { std::lock_guard<std::mutex> lock(m); g->busy = false; }
};
struct StreamLease {
  GeneSlot *slot; hipStream_t st;
  explicit StreamLease(int d) : slot(GeneStreams::get().take(d)), st(slot->st) {}
  ~StreamLease() { (void)hipStreamSynchronize(st); GeneStreams::get().give(slot); }
};
}  // namespace

extern "C" int ckm_genes_call(ckm_ctx *ctx, const char *text, const uint64_t *contig_off, uint32_t ncontigs, const uint32_t *bin_first, uint32_t nbins,
                              int trans_table, int closed, int mask_runs, ckm_genes **out) {
  return guarded([&] {
    if (!ctx || !text || !contig_off || !bin_first || !out) throw Error(CKM_EINVAL, "NULL argument");
    if (trans_table != 11 && trans_table != 4) throw Error(CKM_EINVAL, "translation table must be 11 or 4 (checkm/prodigal.py:86-93)");
    if (bin_first[nbins] != ncontigs) throw Error(CKM_EINVAL, "bin_first[nbins] must equal the number of contigs");
    if (closed) throw Error(CKM_EINVAL, "closed ends (prodigal -c) are not built: CheckM never passes -c (checkm/prodigal.py:86-93)");
    *out = nullptr;
    ctx->settle();
    HIPCHK(hipSetDevice(ctx->device));
    StreamLease lease(ctx->device);
    const double t_begin = now_ms();
    int gthreads = std::max(2, std::min(6, (int)std::thread::hardware_concurrency() / 16));      // (table arithmetic of the training loops; many calls run side by side)
    if (const char *e = getenv("CKM_GENE_THREADS")) gthreads = std::max(1, std::min(128, atoi(e)));
    HostPool gpool(gthreads);
    const bool tr_on = getenv("CKM_TRACE") != nullptr;
    static std::atomic<int> call_no{0};
    const int call_id = call_no++;
    double t_nodes = 0.0;
    ckm::gene::PipeInput in;
    in.text = text; in.contig_off = contig_off; in.ncontigs = ncontigs; in.bin_first = bin_first; in.nbins = nbins; in.trans_table = trans_table; in.mask_runs = mask_runs;
    in.pfor = [&](size_t n, const std::function<void(size_t)> &f) { gpool.run(n, 1, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) f(i); }); };
    in.trace = [&](const char *label) {
      if (!strcmp(label, "nodes in working order")) t_nodes = now_ms();
      if (tr_on) fprintf(stderr, "ckm-trace genes call %d table %d %9.1f ms  %s\n", call_id, trans_table, now_ms() - t_begin, label);
    };
    std::unique_ptr<ckm_genes> o(new ckm_genes());
    ckm::gene::GExec ex; ex.st = lease.st; ex.pin[0] = &lease.slot->pin[0]; ex.pin[1] = &lease.slot->pin[1];
    ckm::gene::gene_pipeline(ex, in, o->r);
    HIPCHK(hipStreamSynchronize(lease.st));
    o->ms_nodes = t_nodes ? t_nodes - t_begin : 0.0; o->ms_host = now_ms() - t_begin;
    *out = o.release();
  });
}

extern "C" int ckm_genes_columns_get(const ckm_genes *gg, ckm_genes_columns *c) {
  if (!gg || !c) { set_last_error("NULL argument"); return CKM_EINVAL; }
  const ckm::gene::GeneResult *g = &gg->r;
  c->n = g->begin.size(); c->bin = g->bin.data(); c->contig = g->contig.data(); c->begin = g->begin.data(); c->end = g->end.data(); c->strand = g->strand.data();
  c->start_type = g->start_type.data(); c->partial_left = g->partial_left.data(); c->partial_right = g->partial_right.data();
  c->rbs_bin = g->rbs_bin.data(); c->mot_len = g->mot_len.data(); c->mot_ndx = g->mot_ndx.data(); c->mot_spacer = g->mot_spacer.data();
  c->gc_cont = g->gc_cont.data(); c->conf = g->conf.data(); c->score = g->score.data(); c->cscore = g->cscore.data(); c->sscore = g->sscore.data();
  c->rscore = g->rscore.data(); c->uscore = g->uscore.data(); c->tscore = g->tscore.data();
  c->prot_off = g->prot_off.data(); c->prot = g->prot.data();
  c->nbins = g->bin_trained.size(); c->bin_trained = g->bin_trained.data(); c->bin_uses_sd = g->bin_uses_sd.data(); c->bin_gc = g->bin_gc.data();
  c->bin_bases = g->bin_bases.data(); c->bin_coding = g->bin_coding.data(); c->bin_nodes = g->bin_nodes_find.data();
  c->ms_nodes = gg->ms_nodes; c->ms_dp_train = g->ms_dp_train; c->ms_score = g->ms_score; c->ms_dp_find = g->ms_dp_find; c->ms_total = gg->ms_host;
  return CKM_OK;
}
extern "C" void ckm_genes_free(ckm_genes *g) { delete g; }

// bases of every bin covered by at least one gene: what ProdigalGeneFeatureParser.codingBases sums over a bin's contigs
// (checkm/prodigal.py:246-274), the numerator of the coding density that chooses between the translation tables (:117-133)
extern "C" int ckm_genes_coding_union(const ckm_genes *gg, uint64_t *bases /* [nbins] */) {
  return guarded([&] {
    if (!gg || !bases) throw Error(CKM_EINVAL, "NULL argument");
    const ckm::gene::GeneResult &g = gg->r;
    const size_t n = g.begin.size(), nb = g.bin_trained.size();
    for (size_t b = 0; b < nb; ++b) bases[b] = 0;
    std::vector<uint32_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { if (g.contig[x] != g.contig[y]) return g.contig[x] < g.contig[y]; return g.begin[x] < g.begin[y]; });
    long long prev_end = -1; uint32_t prev_contig = 0xffffffffu;
    for (uint32_t k : order) {
      if (g.contig[k] != prev_contig) { prev_contig = g.contig[k]; prev_end = -1; }
      const long long s0 = (long long)g.begin[k] - 1, e0 = g.end[k];
      const long long from = std::max(s0, prev_end);
      if (e0 > from) bases[g.bin[k]] += (uint64_t)(e0 - from);
      prev_end = std::max(prev_end, e0);
    }
  });
}

namespace {
const char *kSdMotif[28] = {"None", "GGA/GAG/AGG", "3Base/5BMM", "4Base/6BMM", "AGxAG", "AGxAG", "GGA/GAG/AGG", "GGxGG", "GGxGG", "AGxAG", "AGGAG(G)/GGAGG",
                            "AGGA/GGAG/GAGG", "AGGA/GGAG/GAGG", "GGA/GAG/AGG", "GGxGG", "AGGA", "GGAG/GAGG", "AGxAGG/AGGxGG", "AGxAGG/AGGxGG", "AGxAGG/AGGxGG",
                            "AGGAG/GGAGG", "AGGAG", "AGGAG", "GGAGG", "GGAGG", "AGGAGG", "AGGAGG", "AGGAGG"};
const char *kSdSpacer[28] = {"None", "3-4bp", "13-15bp", "13-15bp", "11-12bp", "3-4bp", "11-12bp", "11-12bp", "3-4bp", "5-10bp", "13-15bp", "3-4bp", "11-12bp", "5-10bp",
                             "5-10bp", "5-10bp", "5-10bp", "11-12bp", "3-4bp", "5-10bp", "11-12bp", "3-4bp", "5-10bp", "3-4bp", "5-10bp", "11-12bp", "3-4bp", "5-10bp"};
const char *kStartType[4] = {"ATG", "GTG", "TTG", "Edge"};
struct OutFile {
  FILE *f = nullptr; std::string buf;
  bool open(const char *path) { f = fopen(path, "wb"); buf.reserve(1 << 20); return f != nullptr; }
  void flush() { if (f && !buf.empty()) { fwrite(buf.data(), 1, buf.size(), f); buf.clear(); } }
  void add(const char *s, size_t n) { buf.append(s, n); if (buf.size() > (1 << 20) - 4096) flush(); }
  void add(const std::string &s) { add(s.data(), s.size()); }
  bool close() { flush(); const bool ok = f && fclose(f) == 0; f = nullptr; return ok; }
  ~OutFile() { if (f) fclose(f); }
};
}  // namespace

// genes.faa / genes.gff (/ genes.fna) of ONE bin of a call in prodigal's layout (`>contig_n # begin # end # strand # attributes`, GFF3 CDS
// lines; the files ProdigalRunner.run leaves behind, checkm/prodigal.py:86-93,136-153).  contig_ids / text / contig_off / bin_first: the
// call's own arguments (ids of all contigs of the batch); nt_path may be NULL.
extern "C" int ckm_genes_write_bin(const ckm_genes *gg, uint32_t bin, int trans_table, const char *const *contig_ids, const char *text, const uint64_t *contig_off,
                                   const uint32_t *bin_first, const char *aa_path, const char *gff_path, const char *nt_path) {
  return guarded([&] {
    if (!gg || !contig_ids || !text || !contig_off || !bin_first || !aa_path || !gff_path) throw Error(CKM_EINVAL, "NULL argument");
    const ckm::gene::GeneResult &g = gg->r;
    if (bin >= g.bin_trained.size()) throw Error(CKM_EINVAL, "no such bin in this call");
    OutFile aa, gff, nt;
    if (!aa.open(aa_path)) throw Error(CKM_EIO, std::string("cannot write ") + aa_path);
    if (!gff.open(gff_path)) throw Error(CKM_EIO, std::string("cannot write ") + gff_path);
    if (nt_path && !nt.open(nt_path)) throw Error(CKM_EIO, std::string("cannot write ") + nt_path);
    const size_t k0 = std::lower_bound(g.bin.begin(), g.bin.end(), bin) - g.bin.begin(), k1 = std::upper_bound(g.bin.begin(), g.bin.end(), bin) - g.bin.begin();
    static const char *comp_from = "ACGTRYKMSWBDHVNacgtrykmswbdhvn", *comp_to = "TGCAYRMKSWVHDBNtgcayrmkswvhdbn";      // (IUPAC codes complement too)
    unsigned char comp[256];
    for (int i = 0; i < 256; ++i) comp[i] = (unsigned char)i;
    for (int i = 0; comp_from[i]; ++i) comp[(unsigned char)comp_from[i]] = (unsigned char)comp_to[i];
    char line[1024];
    gff.add("##gff-version  3\n", 17);
    size_t k = k0;
    for (uint32_t c = bin_first[bin]; c < bin_first[bin + 1]; ++c) {
      const char *cid = contig_ids[c]; const uint64_t clen = contig_off[c + 1] - contig_off[c]; const char *seq = text + contig_off[c];
      int n = snprintf(line, sizeof line, "# Sequence Data: seqnum=%u;seqlen=%llu;seqhdr=\"", c - bin_first[bin] + 1, (unsigned long long)clen);
      gff.add(line, n); gff.add(cid, strlen(cid)); gff.add("\"\n", 2);
      n = snprintf(line, sizeof line, "# Model Data: version=checkm_amd.device.gene_caller;run_type=Single;model=\"Ab initio\";gc_cont=%.2f;transl_table=%d;uses_sd=%d\n",
                   100.0 * g.bin_gc[bin], trans_table, (int)g.bin_uses_sd[bin]);
      gff.add(line, n);
      unsigned idx = 0;
      for (; k < k1 && g.contig[k] == c; ++k) {
        ++idx;
        char motif[16], spacer[16]; const char *mo, *sp;
        if (g.rbs_bin[k] >= 0) { mo = kSdMotif[g.rbs_bin[k]]; sp = kSdSpacer[g.rbs_bin[k]]; }
        else if (g.mot_len[k] > 0) {
          for (int i = 0; i < g.mot_len[k]; ++i) motif[i] = "ACGT"[(g.mot_ndx[k] >> (2 * i)) & 3];
          motif[g.mot_len[k]] = 0; snprintf(spacer, sizeof spacer, "%dbp", g.mot_spacer[k]); mo = motif; sp = spacer;
        } else { mo = "None"; sp = "None"; }
        char at[256];
        const int na = snprintf(at, sizeof at, "ID=%u_%u;partial=%d%d;start_type=%s;rbs_motif=%s;rbs_spacer=%s;gc_cont=%.3f", c - bin_first[bin] + 1, idx, (int)g.partial_left[k], (int)g.partial_right[k],
                                kStartType[g.start_type[k] & 3], mo, sp, g.gc_cont[k]);
        gff.add(cid, strlen(cid));
        n = snprintf(line, sizeof line, "\tcheckm_amd_device\tCDS\t%d\t%d\t%.1f\t%s\t0\t", g.begin[k], g.end[k], g.score[k], g.strand[k] == 1 ? "+" : "-");
        gff.add(line, n); gff.add(at, na);
        n = snprintf(line, sizeof line, ";conf=%.2f;score=%.2f;cscore=%.2f;sscore=%.2f;rscore=%.2f;uscore=%.2f;tscore=%.2f;\n", g.conf[k], g.score[k], g.cscore[k], g.sscore[k], g.rscore[k], g.uscore[k], g.tscore[k]);
        gff.add(line, n);
        std::string head = ">"; head += cid;
        n = snprintf(line, sizeof line, "_%u # %d # %d # %d # ", idx, g.begin[k], g.end[k], (int)g.strand[k]);
        head.append(line, n); head.append(at, na); head += '\n';
        aa.add(head);
        const char *p = g.prot.data() + g.prot_off[k]; const size_t pl = (size_t)(g.prot_off[k + 1] - g.prot_off[k]);
        for (size_t i = 0; i < pl; i += 60) { aa.add(p + i, std::min<size_t>(60, pl - i)); aa.add("\n", 1); }
        if (nt_path) {
          nt.add(head);
          const long long b0 = (long long)g.begin[k] - 1, e0 = std::min<long long>(g.end[k], (long long)clen);
          std::string s;
          if (e0 > b0 && b0 >= 0) {
            s.assign(seq + b0, (size_t)(e0 - b0));
            if (g.strand[k] != 1) { std::reverse(s.begin(), s.end()); for (char &ch : s) ch = (char)comp[(unsigned char)ch]; }
          }
          for (size_t i = 0; i < s.size(); i += 70) { nt.add(s.data() + i, std::min<size_t>(70, s.size() - i)); nt.add("\n", 1); }
        }
      }
    }
    if (!aa.close() || !gff.close() || (nt_path && !nt.close())) throw Error(CKM_EIO, "short write of a gene file");
  });
}
