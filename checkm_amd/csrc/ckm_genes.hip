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
#include <memory>
#include <vector>
#include "ckm_host.h"
#include "gene_pipe.h"

struct ckm_genes {
  ckm::gene::GeneResult r;
  double ms_nodes = 0, ms_host = 0;
};

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
    Worker *w = &ctx->w[0];
    const double t_begin = now_ms();
    int gthreads = std::max(4, std::min(16, (int)std::thread::hardware_concurrency() / 8));
    if (const char *e = getenv("CKM_GENE_THREADS")) gthreads = std::max(1, std::min(128, atoi(e)));
    HostPool gpool(gthreads);
    const bool tr_on = getenv("CKM_TRACE") != nullptr;
    double t_nodes = 0.0;
    ckm::gene::PipeInput in;
    in.text = text; in.contig_off = contig_off; in.ncontigs = ncontigs; in.bin_first = bin_first; in.nbins = nbins; in.trans_table = trans_table; in.mask_runs = mask_runs;
    in.pfor = [&](size_t n, const std::function<void(size_t)> &f) { gpool.run(n, 1, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) f(i); }); };
    in.trace = [&](const char *label) {
      if (!strcmp(label, "nodes in working order")) t_nodes = now_ms();
      if (tr_on) fprintf(stderr, "ckm-trace genes table %d %9.1f ms  %s\n", trans_table, now_ms() - t_begin, label);
    };
    std::unique_ptr<ckm_genes> o(new ckm_genes());
    ckm::gene::GExec ex; ex.st = w->stream;
    ckm::gene::gene_pipeline(ex, in, o->r);
    HIPCHK(hipStreamSynchronize(w->stream));
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
