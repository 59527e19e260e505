// ckm_orf.hip -- C ABI of the gene-calling front end (kernels_orf.hip): nucleotide contigs in, start / stop nodes of all six frames out.
#include <algorithm>
#include <memory>
#include <vector>
#include "ckm_host.h"

namespace ckm {
void launch_orf_flags(hipStream_t stream, const uint8_t *text, uint8_t *flags, uint64_t n);
void launch_orf_chain(hipStream_t stream, const uint8_t *planes, uint64_t nwin, const uint64_t *contig_off, const int32_t *contig_len, uint32_t ncontigs, int tt4, int closed,
                      void *nodes, unsigned long long *nnodes, unsigned long long cap);
void launch_orf_fill(hipStream_t stream, uint8_t *text, uint64_t n, uint32_t seed);
struct OrfNodeH { uint32_t contig; int32_t ndx, stop_val; uint8_t type, strand_rev, edge, pad; };
}  // namespace ckm
using namespace ckm;

struct ckm_orf {
  std::vector<uint32_t> contig;
  std::vector<int32_t> ndx, stop_val;
  std::vector<uint8_t> type, strand_rev, edge;
  double ms_flags = 0.0, ms_chain = 0.0;
  uint64_t bases = 0, padded_bytes = 0;
};

extern "C" int ckm_orf_scan(ckm_ctx *ctx, const char *text, const uint64_t *contig_off, uint32_t ncontigs, int trans_table, int closed, ckm_orf **out) {
  return guarded([&] {
    if (!ctx || !text || !contig_off || !out) throw Error(CKM_EINVAL, "NULL argument");
    if (trans_table != 11 && trans_table != 4) throw Error(CKM_EINVAL, "translation table must be 11 or 4 (checkm/prodigal.py:86-93)");
    *out = nullptr;
    ctx->settle();
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->w[0].stream;
    // layout: 64 bytes of 'N', then every contig followed by 'N' up to the next multiple of 16 (at least 2), then 64 + slack bytes of 'N'
    std::vector<uint64_t> off(ncontigs); std::vector<int32_t> len(ncontigs);
    uint64_t pos = 0, bases = 0;
    for (uint32_t c = 0; c < ncontigs; ++c) {
      const uint64_t n = contig_off[c + 1] - contig_off[c];
      if (n > 0x7ffffff0ull) throw Error(CKM_ERANGE, "contig longer than 2^31 bases");
      off[c] = pos; len[c] = (int32_t)n; bases += n;
      pos = (pos + n + 2 + 15) & ~(uint64_t)15;
    }
    const uint64_t body = (pos + 63) & ~(uint64_t)63;
    std::vector<uint8_t> host(64 + body + 128, (uint8_t)'N');
    for (uint32_t c = 0; c < ncontigs; ++c) memcpy(host.data() + 64 + off[c], text + contig_off[c], (size_t)len[c]);
    DevBuf d_text, d_flags, d_off, d_len, d_nodes, d_cnt;
    d_text.ensure(host.size()); d_flags.ensure(body + 256);                    // eight bit planes of body / 64 words: body bytes
    d_off.ensure(std::max<size_t>(8, ncontigs * 8)); d_len.ensure(std::max<size_t>(4, ncontigs * 4)); d_cnt.ensure(8);
    HIPCHK(hipMemcpyAsync(d_text.p, host.data(), host.size(), hipMemcpyHostToDevice, st));
    if (ncontigs) {
      HIPCHK(hipMemcpyAsync(d_off.p, off.data(), ncontigs * 8, hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(d_len.p, len.data(), ncontigs * 4, hipMemcpyHostToDevice, st));
    }
    HIPCHK(hipMemsetAsync(d_cnt.p, 0, 8, st));
    hipEvent_t e0, e1, e2;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventCreate(&e2));
    std::unique_ptr<ckm_orf> o(new ckm_orf());
    o->bases = bases; o->padded_bytes = body;
    // a bacterial genome carries about one start / stop node per 12 bases; a too-small table is detected and the chain kernel runs again
    unsigned long long cap = std::max<unsigned long long>(1 << 16, bases / 6);
    unsigned long long n = 0;
    const uint8_t *tx = d_text.as<uint8_t>() + 64; uint8_t *fl = d_flags.as<uint8_t>();
    HIPCHK(hipEventRecord(e0, st));
    launch_orf_flags(st, tx, fl, body);
    HIPCHK(hipEventRecord(e1, st));
    for (int attempt = 0; attempt < 2; ++attempt) {
      d_nodes.ensure((size_t)cap * sizeof(OrfNodeH));
      if (attempt) { HIPCHK(hipMemsetAsync(d_cnt.p, 0, 8, st)); HIPCHK(hipEventRecord(e1, st)); }
      launch_orf_chain(st, fl, body / 64, d_off.as<uint64_t>(), d_len.as<int32_t>(), ncontigs, trans_table == 4 ? 1 : 0, closed ? 1 : 0, d_nodes.p, d_cnt.as<unsigned long long>(), cap);
      HIPCHK(hipEventRecord(e2, st));
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpyAsync(&n, d_cnt.p, 8, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      if (n <= cap) break;
      cap = n + 1024;
    }
    float a = 0.f, b = 0.f;
    HIPCHK(hipEventElapsedTime(&a, e0, e1)); HIPCHK(hipEventElapsedTime(&b, e1, e2));
    o->ms_flags = a; o->ms_chain = b;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
    std::vector<OrfNodeH> nodes((size_t)n);
    if (n) HIPCHK(hipMemcpy(nodes.data(), d_nodes.p, (size_t)n * sizeof(OrfNodeH), hipMemcpyDeviceToHost));
    // prodigal's working order (node.c: compare_nodes -- position, then the forward strand first), made total
    std::sort(nodes.begin(), nodes.end(), [](const OrfNodeH &x, const OrfNodeH &y) {
      if (x.contig != y.contig) return x.contig < y.contig;
      if (x.ndx != y.ndx) return x.ndx < y.ndx;
      if (x.strand_rev != y.strand_rev) return x.strand_rev < y.strand_rev;
      if (x.type != y.type) return x.type < y.type;
      if (x.stop_val != y.stop_val) return x.stop_val < y.stop_val;
      return x.edge < y.edge;
    });
    o->contig.reserve(n); o->ndx.reserve(n); o->stop_val.reserve(n); o->type.reserve(n); o->strand_rev.reserve(n); o->edge.reserve(n);
    for (const OrfNodeH &x : nodes) { o->contig.push_back(x.contig); o->ndx.push_back(x.ndx); o->stop_val.push_back(x.stop_val); o->type.push_back(x.type); o->strand_rev.push_back(x.strand_rev); o->edge.push_back(x.edge); }
    *out = o.release();
  });
}

extern "C" int ckm_orf_columns_get(const ckm_orf *o, ckm_orf_columns *c) {
  if (!o || !c) { set_last_error("NULL argument"); return CKM_EINVAL; }
  c->n = o->ndx.size(); c->contig = o->contig.data(); c->ndx = o->ndx.data(); c->stop_val = o->stop_val.data();
  c->type = o->type.data(); c->strand_rev = o->strand_rev.data(); c->edge = o->edge.data();
  c->ms_flags = o->ms_flags; c->ms_chain = o->ms_chain; c->bases = o->bases; c->padded_bytes = o->padded_bytes;
  return CKM_OK;
}

extern "C" void ckm_orf_free(ckm_orf *o) { delete o; }

// measurement hook: the streaming flag kernel over `nbytes` of device-generated nucleotides (larger than the 256 MB last-level cache when
// an HBM figure is wanted), `reps` launches timed with HIP events on the stream they run on; *ms = average duration of one launch
extern "C" int ckm_debug_orf_flags(ckm_ctx *ctx, uint64_t nbytes, uint32_t reps, double *ms) {
  return guarded([&] {
    if (!ctx || !ms || !reps) throw Error(CKM_EINVAL, "bad argument");
    ctx->settle();
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->w[0].stream;
    const uint64_t n = (std::max<uint64_t>(nbytes, 4096) + 63) & ~(uint64_t)63;
    DevBuf d_text, d_flags;
    d_text.ensure(n + 256); d_flags.ensure(n + 256);
    launch_orf_fill(st, d_text.as<uint8_t>(), n + 192, 12345u);
    launch_orf_flags(st, d_text.as<uint8_t>() + 64, d_flags.as<uint8_t>(), n);        // warm-up
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, st));
    for (uint32_t r = 0; r < reps; ++r) launch_orf_flags(st, d_text.as<uint8_t>() + 64, d_flags.as<uint8_t>(), n);
    HIPCHK(hipEventRecord(e1, st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms = (double)t / reps;
  });
}
