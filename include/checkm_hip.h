/* checkm_hip.h -- C ABI of libcheckm_hip.so, the MI355X (gfx950) marker-gene hot path of CheckM.
 *
 * The reference has no FFI: its seam is a shell command plus a text file.  Each entry point below
 * names the reference interface it replaces (paths relative to the CheckM source tree).
 *
 * Conventions
 *   - every function returns 0 on success or a negative CKM_E* code; ckm_last_error() returns a
 *     thread-local message for the last failure.  No C++ exception crosses this boundary.
 *   - the caller owns every input buffer; the library owns every output object and frees it in
 *     the matching *_free().  Column pointers returned by ckm_hits_columns()/ckm_qa_columns() stay
 *     valid until that object is freed.
 *   - a ckm_ctx runs ONE ckm_search / ckm_reduce / ckm_align at a time (not re-entrant); different ctxs are independent, also on
 *     one device: MarkerGeneFinder.find keeps up to three on a device, one batch of bins in flight on each.  A ctx owns its streams,
 *     tables and float workspace (budget: a quarter of the device memory free at creation, at most 96 GB); ckm_profiles, ckm_seqs
 *     and ckm_hits are plain device / host memory and may be used with ANY ctx of the device they were created on (the contexts of
 *     a device share one resident copy of a profile database; ckm_reduce may be called with another ctx than the one that searched).  ckm_seqs_pack / ckm_seqs_from_fasta use only the ctx's upload
 *     staging area and high-priority stream (serialised by a mutex of the ctx), ckm_hits_write_domtblout touches no state of the ctx: all
 *     three may run on other threads while a search is in flight (MarkerGeneFinder.find reads the next batch of bins and writes the previous batch's
 *     tables that way).  Searches of different ctxs of ONE device run concurrently but take their SSV phases one after the other
 *     (a device-wide baton inside the library).  The library never falls back to a CPU implementation: without a usable HIP
 *     device ckm_ctx_create() fails with CKM_ENODEV and nothing else can be called.
 */
#ifndef CHECKM_HIP_H
#define CHECKM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CKM_ABI_VERSION 7

enum {
  CKM_OK      =  0,
  CKM_EINVAL  = -1,   /* bad argument */
  CKM_EIO     = -2,   /* file could not be read/written */
  CKM_EFORMAT = -3,   /* malformed HMMER3 profile / input */
  CKM_ENODEV  = -4,   /* no usable HIP device */
  CKM_EHIP    = -5,   /* HIP runtime error */
  CKM_ENOMEM  = -6,
  CKM_ERANGE  = -7    /* size limit exceeded (see DESIGN.md limits) */
};

typedef struct ckm_ctx      ckm_ctx;
typedef struct ckm_profiles ckm_profiles;
typedef struct ckm_seqs     ckm_seqs;
typedef struct ckm_hits     ckm_hits;
typedef struct ckm_qa       ckm_qa;

const char *ckm_last_error(void);
int         ckm_abi_version(void);

/* Replaces the "is hmmsearch on PATH" probe, checkm/hmmer.py:131-137 (HMMERRunner.checkForHMMER). */
int ckm_device_count(int *n);
int ckm_ctx_create(int device, ckm_ctx **out);
void ckm_ctx_destroy(ckm_ctx *ctx);

/* ---- profiles ------------------------------------------------------------------------------
 * Replaces hmmsearch's reading of <hmmfile> (checkm/hmmer.py:70) and the header skim of
 * checkm/hmmerModelParser.py:54-83.  Parses a HMMER3/f ASCII file completely (headers, COMPO,
 * emissions, transitions, STATS LOCAL), configures the multihit-local search profiles and uploads
 * the score tables to HBM.  The header view is the RAW per-record view; the sticky ACC/GA/TC/NC
 * carry-over quirk of HmmModelParser.simpleParse is applied by the Python mirror, not here. */
typedef struct {
  const char *name;      /* NAME */
  const char *acc;       /* ACC or NULL */
  const char *desc;      /* DESC or NULL */
  int32_t     leng;      /* LENG */
  int32_t     has_ga, has_tc, has_nc;
  double      ga[2], tc[2], nc[2];   /* float64: CheckM compares them as Python floats with text scores */
  float       evparam[6];   /* MSV mu,lambda; VITERBI mu,lambda; FORWARD tau,lambda */
  int32_t     searchable;   /* 0: the model is longer than the kernels are instantiated for (LENG > 4096; models of 2049..4096 nodes are searched, through the exact MSV kernel instead of SSV): it keeps its place in the database, but a
                               ckm_search / ckm_align that selects it fails with CKM_ERANGE naming it */
} ckm_model_header;

int  ckm_profiles_load(ckm_ctx *ctx, const char *hmm_path, ckm_profiles **out);
int  ckm_profiles_count(const ckm_profiles *p, int32_t *n);
int  ckm_profiles_header(const ckm_profiles *p, int32_t i, ckm_model_header *out);
void ckm_profiles_free(ckm_profiles *p);

/* ---- target sequences ----------------------------------------------------------------------
 * Replaces hmmsearch's reading of <seqfile> (the genes.faa written at
 * checkm/markerGeneFinder.py:113-127).  `text` holds the residues of all sequences of all bins
 * back to back (no separators); sequence s is text[seq_off[s] .. seq_off[s+1]).  Bin b owns
 * sequences bin_off[b] .. bin_off[b+1]; Z of a bin = its sequence count, as one hmmsearch
 * run per bin has it.  names/descs (nseq entries, descs may be NULL) are copied. */
int  ckm_seqs_pack(ckm_ctx *ctx, const char *text, const uint64_t *seq_off, uint32_t nseq,
                   const uint32_t *bin_off, uint32_t nbins,
                   const char *const *names, const char *const *descs, ckm_seqs **out);
/* Same, reading the sequences itself: one protein FASTA file per bin (the bins/<binId>/genes.faa files of
 * checkm/markerGeneFinder.py:113-127; what hmmsearch does with <seqfile>).  Name = header up to the first
 * whitespace, description = the rest; residues are digitized on the fly.  Plain text files only. */
int  ckm_seqs_from_fasta(ckm_ctx *ctx, const char *const *paths, uint32_t nbins, ckm_seqs **out);
int  ckm_seqs_count(const ckm_seqs *s, uint32_t *nseq, uint32_t *nbins);
int  ckm_seqs_bin_offsets(const ckm_seqs *s, const uint32_t **bin_off);          /* [nbins+1], owned by s */
int  ckm_seqs_name(const ckm_seqs *s, uint32_t i, const char **name, const char **desc, int32_t *len);
int  ckm_seqs_residues(const ckm_seqs *s, uint64_t *total);
void ckm_seqs_free(ckm_seqs *s);

/* ---- the scan ------------------------------------------------------------------------------
 * Replaces one `hmmsearch --domtblout T --notextw -E <E> --domE <domE> --noali` process per bin
 * (checkm/markerGeneFinder.py:134-142 -> checkm/hmmer.py:61-74).
 * Bin b is searched with models model_idx[model_off[b] .. model_off[b+1]) in that order
 * (the order of the temporary HMM file of checkm/markerSets.py:326-343); model_off == NULL
 * means every model of `p`, in file order, for every bin. */
int  ckm_search(ckm_ctx *ctx, const ckm_profiles *p, const ckm_seqs *s,
                const uint32_t *model_off, const uint32_t *model_idx,
                double E, double domE, ckm_hits **out);

/* One entry per reported domain = one domtblout row (column contract: checkm/hmmer.py:255-285). */
typedef struct {
  uint64_t        n;            /* rows */
  uint32_t        nbins;
  const uint64_t *bin_row_off;  /* [nbins+1] rows of bin b, already in domtblout order */
  const uint32_t *seq;          /* global sequence index (target_name / description / tlen) */
  const uint32_t *model;        /* profile index (query_name / accession / qlen) */
  const int32_t  *tlen, *qlen;
  const double   *full_evalue;  const float *full_score, *full_bias;
  const int32_t  *dom_idx, *ndom;
  const double   *c_evalue, *i_evalue; const float *dom_score, *dom_bias;
  const int32_t  *hmm_from, *hmm_to, *ali_from, *ali_to, *env_from, *env_to;
  const float    *acc;
  /* Only when the struct is INPUT to ckm_reduce (a table parsed from existing domtblout text):
   * target name of every row (prodigal `<contig>_<n>` form, resultsParser.py:410-422).  NULL in
   * the struct ckm_hits_columns() fills: those rows take their names from the ckm_seqs. */
  const char *const *target_name;
  /* INPUT only, optional: the two scores as float64 exactly as parsed from the text (Python floats in the
   * reference).  float32 cannot hold "25.3" exactly, and vetHit compares scores with cutoffs such as GA 25.30. */
  const double *full_score_d, *dom_score_d;
} ckm_hit_columns;

int  ckm_hits_columns(const ckm_hits *h, ckm_hit_columns *out);
void ckm_hits_free(ckm_hits *h);

/* Writes bin b's rows as hmmsearch --domtblout text, the file every later CheckM stage re-reads
 * (checkm/resultsParser.py:191-204; written today by hmmsearch itself, checkm/hmmer.py:70). */
int  ckm_hits_write_domtblout(const ckm_hits *h, const ckm_profiles *p, const ckm_seqs *s,
                              uint32_t bin, const char *path);

/* ---- per-stage counters of the last ckm_search on this ctx (bench.py) --- */
typedef struct {
  uint64_t pairs_ssv, pairs_msv_full, pairs_bias, pairs_vit, pairs_fwd, pairs_dom, envelopes;
  uint64_t regions_multi;       /* regions resolved by the stochastic trace ensemble */
  uint64_t pairs_vit_exact;     /* Viterbi pairs re-run by the exact kernel (bound failed F2 with the J flag set) */
  uint64_t cells_ssv;           /* sum over pairs of L*M: GCUPS denominator */
  uint64_t residue_hmm;         /* sum over pairs of L */
  double   ms_ssv, ms_filters, ms_fwdbwd, ms_domains, ms_host, ms_total;
  uint32_t ssv_launches;
  uint32_t cascade_fallback_lanes; /* lanes (length classes) of the last search that outgrew the device-side tables / workspace of the
                                      device-driven cascade and were run by the host-driven one instead (0 in the normal case) */
  uint64_t ws_cap_bytes, ws_used_bytes;  /* float workspace of the device-driven cascade: allocated, and the high-water mark the search asked for */
} ckm_search_stats;
/* Replaces the text `hmmsearch -o <hmmerOut>` leaves when CheckM keeps alignments (bKeepAlignment: checkm/markerGeneFinder.py:138-142
 * drops --noali): per query model the score table and, per reported domain, the alignment of the envelope's optimal-accuracy path
 * (model consensus / identity-or-'+' line / target with inserts in lower case / PP line: the posterior probability of every aligned
 * residue in the state that emits it, in hmmsearch's one-character code).  The `exp` column of hmmsearch's score table is not produced;
 * CheckM never reads this file back. */
int ckm_hits_write_alignments(ckm_ctx *ctx, const ckm_hits *h, const ckm_profiles *p, const ckm_seqs *s, uint32_t bin, const char *path);

int ckm_last_search_stats(const ckm_ctx *ctx, ckm_search_stats *out);

/* Starts allocating, on a background thread, the float workspace a search of `pairs` (ORF, model) pairs will ask for, so that the
 * allocation (tens of GB: seconds of hipMalloc) runs beside the caller's reading of FASTA files and profile databases instead of
 * inside the first ckm_search.  `cells` is what ckm_search sizes the workspace from: the sum, over the bins of the search and over the
 * models each bin is scanned against, of (Mp + 64) * Mp with Mp = 64 * ceil(M / 64) -- a marker model finds about one domain per bin
 * and a domain costs its envelope's matrices.  Returns at once; ckm_search / ckm_align / the diagnostics wait for it.  Replaces nothing
 * of the reference (one hmmsearch process per bin allocates per process, checkm/hmmer.py:70): it exists because one context serves a
 * whole batch of bins. */
int ckm_ctx_reserve(ckm_ctx *ctx, uint64_t pairs, double cells);

/* ---- the reduction --------------------------------------------------------------------------
 * Replaces ResultsParser.parseBinHits -> ResultsManager.{vetHit,addHit} -> PFAM.filterHitsFromSameClan
 * -> identifyAdjacentMarkerGenes -> geneCounts -> MarkerSet.genomeCheck
 * (checkm/resultsParser.py:76-119,340-537; checkm/util/pfam.py:86-147; checkm/markerSets.py:206-238).
 * Hits come either from ckm_search (h != NULL) or, for tables parsed from existing domtblout
 * text, through `ext` (same columns, text-rounded values).  */
typedef struct {
  /* per model (index = profile index, or caller's own model numbering when ext is used) */
  uint32_t        nmodels;
  const int32_t  *qlen;
  const uint8_t  *thr_kind;     /* 0 none, 1 NC(TIGR), 2 GA, 3 TC, 4 NC : the cascade of resultsParser.py:356-367, resolved by the caller from the (sticky) header view */
  const double   *thr_full, *thr_dom; /* Python floats in the reference: compared as float64 */
  const uint8_t  *is_pf;        /* marker id starts with 'PF' (pfam.py:97) */
  const int32_t  *clan;         /* clan id or -1 (pfam.py:116: None==None counts as same clan) */
  const uint32_t *nest_off, *nest_idx;   /* CSR: models nested with model m (pfam.py:135) */
  const uint32_t *key;          /* models sharing one accession share a key (markerHits dict key) */
} ckm_model_info;

typedef struct {
  int32_t ignore_thresholds, skip_pseudogene_correction, skip_adj_correction, individual_markers;
  double  evalue_threshold, length_threshold;
  const uint8_t *bin_select;    /* NULL = every bin; else only bins with a non-zero byte are reduced */
  /* Threshold VARIANTS: the reference resolves a model's GA/TC/NC from the header view of the BIN's own model file, and a sticky
   * header parse (hmmerModelParser.py:54-83) can give the same model different cutoffs in different subsets.  With nvariants > 1,
   * ckm_model_info.thr_kind / thr_full / thr_dom hold nvariants x nmodels entries (variant-major) and bin b uses variant
   * bin_variant[b]; nvariants <= 1 or bin_variant == NULL: one table for every bin. */
  uint32_t        nvariants;
  const uint32_t *bin_variant;  /* [nbins] or NULL */
} ckm_reduce_flags;

typedef struct {
  /* CSR over bins -> collocated sets -> marker keys */
  uint32_t        nbins;
  const uint32_t *set_off;      /* [nbins+1] */
  const uint32_t *marker_off;   /* [nsets+1] */
  const uint32_t *marker_key;   /* [nmarkers] key (see ckm_model_info.key) */
} ckm_marker_sets;

int  ckm_reduce(ckm_ctx *ctx, const ckm_hits *h, const ckm_hit_columns *ext, const ckm_seqs *s,
                const ckm_model_info *mi, const ckm_reduce_flags *fl, const ckm_marker_sets *ms, ckm_qa **out);

typedef struct {
  uint32_t        nbins;
  const int32_t  *hist;         /* [nbins*6] markers with 0,1,2,3,4,5+ hits (resultsParser.py:513-529) */
  const double   *completeness, *contamination;   /* markerSets.py:206-238 */
  const uint32_t *set_off;      /* [nbins+1] */
  const int32_t  *set_present, *set_multi;        /* per collocated set */
  /* surviving hits (ResultsManager.markerHits after all filters), grouped by bin then marker key in
   * the reference's dict/list order */
  uint64_t        nkept;
  const uint64_t *kept_bin_off; /* [nbins+1] */
  const uint32_t *kept_key;     /* marker key */
  const uint64_t *kept_row;     /* row index into the hit columns */
  const uint64_t *kept_row2;    /* second row when two adjacent ORFs were merged, else UINT64_MAX */
  const int32_t  *kept_tlen, *kept_hmm_from, *kept_hmm_to, *kept_ali_from, *kept_ali_to, *kept_env_from, *kept_env_to;
} ckm_qa_columns;

int  ckm_qa_columns_get(const ckm_qa *q, ckm_qa_columns *out);
void ckm_qa_free(ckm_qa *q);

/* The set-counting kernel alone: replaces the counting loops of MarkerSet.genomeCheck
 * (checkm/markerSets.py:206-238) and ResultsManager.geneCounts (checkm/resultsParser.py:513-529)
 * for caller-supplied copy numbers.  marker_count[i] = number of hits of marker i of the CSR
 * (i indexes ms->marker_key), marker_first[i] != 0 iff this is the first occurrence of that marker
 * within its bin.  Outputs: set_present/set_multi [nsets], hist [nbins*6], and per bin the
 * individual-marker totals present_total/multi_total [nbins].  The float64 division is left to the
 * caller, to be done in the reference's accumulation order. */
int  ckm_count_sets(ckm_ctx *ctx, const ckm_marker_sets *ms, const int32_t *marker_count, const uint8_t *marker_first,
                    int32_t *set_present, int32_t *set_multi, int32_t *hist, int32_t *present_total, int32_t *multi_total);

/* ---- alignment of marker genes to their models ---------------------------------------------------
 * Replaces `hmmalign --outformat Pfam <hmm> <seqs>` (checkm/hmmer.py:76-95, called from HmmerAligner._alignMarker,
 * checkm/hmmerAligner.py:275-302) as far as CheckM consumes it: _maskAlignment (hmmerAligner.py:325-352) keeps only the
 * match ('#=GC RF' x) columns, i.e. per model node the residue its match state emits on the optimal-accuracy path, or a gap.
 * Pair j aligns the WHOLE sequence seq[j] to model[j] (unihit local profile, length model of that sequence, Forward / Backward /
 * decoding / optimal accuracy -- hmmalign's own per-sequence computation).  node_residue[out_off[j] + k], k = 0..M-1, receives
 * the 1-based residue index emitted by match state k+1, or 0 (node deleted, or outside the local alignment).
 * out_off[j+1] - out_off[j] must equal the length of model[j].  A pair whose posterior decoding leaves the float range (the
 * eslERANGE case of HMMER's decoding: e.g. two strong copies of the domain in one sequence under this one-domain model) reports
 * no column at all (every entry 0). */
int  ckm_align(ckm_ctx *ctx, const ckm_profiles *p, const ckm_seqs *s, const uint32_t *model, const uint32_t *seq, uint32_t n,
               const uint64_t *out_off, int32_t *node_residue);

/* ---- tables written by an earlier command -----------------------------------------------------
 * Replaces HMMERParser.readHitsDOM / HmmerHitDOM (checkm/hmmer.py:184-200, 255-285) and the serial per-bin loop around them
 * (checkm/resultsParser.py:94, 191-204): the domtblout text of all bins is parsed once, on a few threads, into the column form
 * ckm_reduce takes as `ext`.  No device is needed.  A path that cannot be opened gives an empty bin with bin_missing = 1
 * (the reference prints the IOError and goes on, resultsParser.py:200-204); a malformed row is CKM_EFORMAT. */
typedef struct ckm_tables ckm_tables;
typedef struct {
  ckm_hit_columns    cols;               /* numeric columns, target_name, full_score_d/dom_score_d; seq = row index;
                                            model = slot set by ckm_tables_assign_models (UINT32_MAX = not in the list) */
  const char *const *target_accession, *const *query_name, *const *query_accession /* '-' replaced by the name */, *const *description;
  const double      *full_bias_d, *dom_bias_d, *acc_d;      /* the remaining text floats as float64 (Python floats in HmmerHitDOM) */
  const uint8_t     *bin_missing;        /* [nbins] */
} ckm_table_columns;
int  ckm_tables_read(const char *const *paths, uint32_t nbins, ckm_tables **out);
/* model slot of every row = index of its query accession in keys[] (the markerHits keys of the caller's ckm_model_info) */
int  ckm_tables_assign_models(ckm_tables *t, const char *const *keys, uint32_t nkeys, uint64_t *unknown_rows);
int  ckm_tables_get(const ckm_tables *t, ckm_table_columns *out);
void ckm_tables_free(ckm_tables *t);

/* ---- gene calling, first slice ------------------------------------------------------
 * The deterministic front end of the gene finder CheckM runs before the scan -- `prodigal -p single -m -f gff -g <11|4>`, twice per bin,
 * checkm/prodigal.py:74,86-93,131-133: start / stop codon flags of all six frames and the start / stop NODES the gene finder's dynamic
 * program works on (Prodigal 2.6.3 node.c: add_nodes), for all contigs of a bin in two kernel launches.  `text` holds the contigs'
 * nucleotides (ASCII, any case; anything but ACGTU is "no base"), contig c = text[contig_off[c] .. contig_off[c+1]).  The rest of the gene
 * finder is ckm_genes_call below. */
typedef struct ckm_orf ckm_orf;
typedef struct {
  uint64_t        n;            /* nodes, sorted by (contig, position, strand [forward first: node.c compare_nodes], type, stop_val, edge) */
  const uint32_t *contig;
  const int32_t  *ndx;          /* position of the codon's first base on the FORWARD strand's coordinates (node.c: ndx) */
  const int32_t  *stop_val;     /* start node: the stop that closes its ORF; stop node: the previous stop of its frame (node.c: stop_val) */
  const uint8_t  *type;         /* 0 ATG, 1 GTG, 2 TTG, 3 stop */
  const uint8_t  *strand_rev;   /* 0 forward, 1 reverse */
  const uint8_t  *edge;         /* the ORF runs off an end of the contig */
  double          ms_flags, ms_chain;   /* kernel times of this call (HIP events): the streaming flag kernel, the per-frame chains */
  uint64_t        bases, padded_bytes;  /* nucleotides given; bytes the flag kernel read (= wrote) */
} ckm_orf_columns;
int  ckm_orf_scan(ckm_ctx *ctx, const char *text, const uint64_t *contig_off, uint32_t ncontigs, int trans_table, int closed_ends, ckm_orf **out);
int  ckm_orf_columns_get(const ckm_orf *o, ckm_orf_columns *out);
void ckm_orf_free(ckm_orf *o);
/* measurement hook (bench.py: gene_front_end): the streaming flag kernel over nbytes of device-generated nucleotides, average ms of reps launches */
int  ckm_debug_orf_flags(ckm_ctx *ctx, uint64_t nbytes, uint32_t reps, double *ms);

/* ---- gene calling, the body -----------------------------------------------------------
 * Replaces the two `prodigal -p single -q -m -f gff -g <table> -a genes.faa -i <bin>` processes per bin of checkm/prodigal.py:80-93
 * (ProdigalRunner.run) for a BATCH of bins and ONE translation table per call: training on each bin's own sequence (GC-frame bias, hexamer
 * coding statistics, start-site model), node scoring, the dynamic program over the nodes, gene records and their translations -- the
 * single-genome mode of Prodigal 2.6.3 as restated in oracle/gene_full.c (parity unpinned: no prodigal exists next to the reference).
 * NOT built: `-p meta` (checkm/prodigal.py:80-83 uses it below 100 kb; prodigal refuses to train below 20 kb): such a bin comes back with
 * bin_trained = 0 and no genes, and the caller decides (checkm_amd/prodigal.py falls back to an external binary or reports the bin).
 * text: all contigs' nucleotides (ASCII, any case; anything but ACGTU is an unknown base); contig c = text[contig_off[c] .. contig_off[c+1]);
 * bin b owns contigs bin_first[b] .. bin_first[b+1]-1.  mask_n_runs = prodigal's -m (runs of >= 50 unknown bases hide the genes crossing them). */
typedef struct ckm_genes ckm_genes;
typedef struct {
  uint64_t        n;                 /* genes, bin by bin, contig by contig, in order along the contig */
  const uint32_t *bin, *contig;      /* contig: index into the call's contig table */
  const int32_t  *begin, *end;       /* 1-based, inclusive, on the contig's forward strand (GFF columns 4 and 5) */
  const int8_t   *strand;            /* +1 / -1 */
  const uint8_t  *start_type;        /* 0 ATG, 1 GTG, 2 TTG, 3 Edge */
  const uint8_t  *partial_left, *partial_right;
  const int32_t  *rbs_bin;           /* Shine-Dalgarno bin 0..27 the start was credited with, or -1 (upstream motif / none) */
  const int32_t  *mot_len, *mot_ndx, *mot_spacer;   /* upstream motif of organisms without SD: length 3..6 (0 none), its base-4 index, spacer */
  const double   *gc_cont, *conf, *score, *cscore, *sscore, *rscore, *uscore, *tscore;
  const uint64_t *prot_off;          /* [n + 1] */
  const char     *prot;              /* translations, '*' for the stop of a complete gene */
  uint64_t        nbins;
  const uint8_t  *bin_trained, *bin_uses_sd;
  const double   *bin_gc;
  const uint64_t *bin_bases, *bin_coding, *bin_nodes;   /* nucleotides, bases inside genes (sum of gene lengths), start/stop nodes of the contigs */
  double          ms_nodes, ms_dp_train, ms_score, ms_dp_find, ms_total;   /* wall to the nodes; kernel times (HIP events) of both dynamic programs and of the per-node scores; wall of the call */
} ckm_genes_columns;
int  ckm_genes_call(ckm_ctx *ctx, const char *text, const uint64_t *contig_off, uint32_t ncontigs, const uint32_t *bin_first, uint32_t nbins,
                    int trans_table, int closed_ends, int mask_n_runs, ckm_genes **out);
int  ckm_genes_columns_get(const ckm_genes *g, ckm_genes_columns *out);
void ckm_genes_free(ckm_genes *g);
/* Several ckm_genes_call may run at once on one context (each takes a stream of its own): the gene finder is latency-bound, its
 * throughput comes from calls in flight.  closed_ends must be 0 (CheckM never passes prodigal's -c; ABI 6 refuses it).
 * ckm_genes_coding_union: bases of every bin covered by at least one gene -- ProdigalGeneFeatureParser.codingBases summed over the
 * bin's contigs (checkm/prodigal.py:246-274), the numerator of the coding density that picks the translation table (:117-133).
 * ckm_genes_write_bin: genes.faa / genes.gff (/ genes.fna when nt_path is not NULL) of one bin in prodigal's layout, the files
 * ProdigalRunner.run leaves in bins/<binId>/ (checkm/prodigal.py:86-93,136-153); contig_ids[c] = header of contig c up to the first
 * white space (checkm/util/seqUtils.py:180-211), text / contig_off / bin_first as passed to ckm_genes_call. */
int  ckm_genes_coding_union(const ckm_genes *g, uint64_t *bases /* [nbins] */);
int  ckm_genes_write_bin(const ckm_genes *g, uint32_t bin, int trans_table, const char *const *contig_ids, const char *text, const uint64_t *contig_off,
                         const uint32_t *bin_first, const char *aa_path, const char *gff_path, const char *nt_path);

/* ---- the nucleotide files of a batch of bins, laid out for ckm_genes_call / ckm_genes_write_bin (ABI 7) --------------------------------
 * What checkm/prodigal.py:86-93 hands to prodigal by path (`-i <bin>`): plain (uncompressed) nucleotide FASTA files, one per bin, read by
 * host threads (a file per thread) with the record rules of CheckM's own reader (checkm/util/seqUtils.py:180-211): a record begins with
 * '>' at the start of a line, text before the first one is skipped, the contig id is the header's first blank-delimited word, the
 * sequence is what follows up to the next record without '\n', '\r', ' ', '\t'.  The view's arrays are the arguments of ckm_genes_call
 * (text, contig_off, ncontigs, bin_first, nbins) and ckm_genes_write_bin (contig_ids) and live until ckm_nuc_batch_free. */
typedef struct ckm_nuc_batch ckm_nuc_batch;
typedef struct {
  const char        *text;          /* all contigs' nucleotides end to end */
  const uint64_t    *contig_off;    /* [ncontigs + 1] */
  const uint32_t    *bin_first;     /* [nbins + 1] */
  const char *const *contig_ids;    /* [ncontigs] */
  const uint64_t    *bin_bases;     /* [nbins]: nucleotides of each bin */
  uint32_t           ncontigs, nbins;
} ckm_nuc_batch_view;
int  ckm_nuc_batch_read(const char *const *paths, uint32_t nbins, ckm_nuc_batch **out);
int  ckm_nuc_batch_view_get(const ckm_nuc_batch *b, ckm_nuc_batch_view *out);
void ckm_nuc_batch_free(ckm_nuc_batch *b);

/* ---- diagnostics used by the parity tests: every stage of one (model, sequence) pair, no filtering */
typedef struct {
  int32_t msv_xJ;  float msv_sc, null_sc, bias_sc;
  int32_t vit_xC;  float vit_sc, fwd_sc, fwd_xC;  int32_t fwd_nscale;
  int32_t ssv_maxv;
  int32_t msvp_xJ; float msvp_sc;      /* the packed exact-MSV kernel the search uses (must equal msv_xJ / msv_sc) */
} ckm_stage_scores;
int ckm_debug_stages(ckm_ctx *ctx, const ckm_profiles *p, const ckm_seqs *s,
                     const uint32_t *model, const uint32_t *seq, uint32_t npairs, ckm_stage_scores *out);
typedef struct {
  float envsc, oasc, fwd_xC; int32_t nscale; float null2[20];
  int32_t hmm_from, hmm_to, ali_from, ali_to; int32_t ok;
} ckm_envelope_result;
int ckm_debug_envelopes(ckm_ctx *ctx, const ckm_profiles *p, const ckm_seqs *s,
                        const uint32_t *model, const uint32_t *seq, const int32_t *ienv, const int32_t *jenv,
                        uint32_t n, ckm_envelope_result *out);

/* The trace ensemble of one multi-domain region ireg..jreg (1-based, inclusive) of sequence `seq`:
 * n2sum[jreg-ireg+1] = per residue, sum over the 200 traces of the null2 odds ratio; segs[200*cap*4] / nseg[200] = every
 * trace's sampled segments {sqfrom, sqto, hmmfrom, hmmto} in region-local coordinates, first domain first;
 * env[envcap*4] / *nenv = the clustered envelopes, sorted by start. */
int ckm_debug_region(ckm_ctx *ctx, const ckm_profiles *p, const ckm_seqs *s, uint32_t model, uint32_t seq,
                     int32_t ireg, int32_t jreg, float *n2sum, int32_t *segs, int32_t *nseg, int32_t cap,
                     int32_t *env, int32_t envcap, int32_t *nenv);

#ifdef __cplusplus
}
#endif
#endif
