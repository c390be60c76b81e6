/* airlift.h -- C-ABI of the MI355X-native AirLift re-alignment path (libairlift.so).
 *
 * Drop-in boundary for the one hot path AirLift delegates to an external aligner:
 *   src/0-align_reads.sh:13, src/0-align_singletons.sh:12, src/3-align_gaps/align_gaps.sh:14-15
 * (process level: `airlift-align` CLI), and, at library level, the C API of the bundled
 * minimap2 fork (src/minimap2-master_remapping/minimap.h) for the `-ax sr` path.  Each entry point
 * below names the reference declaration it replaces.  Plain pointers and sizes only.
 *
 * Ownership / threading rules are the reference's (minimap.h:296-331): the index is immutable and
 * shareable after build; one al_ctx_t per host thread (it owns a HIP stream and device workspaces);
 * hit arrays returned by al_map_frag() and each al_reg1_t::cigar are malloc()ed by the callee and
 * free()d by the caller.  Errors: NULL / negative return, message on stderr; there is NO CPU
 * fallback -- without a usable HIP device al_ctx_init() fails loudly.
 */
#ifndef AIRLIFT_H
#define AIRLIFT_H

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AL_VERSION "0.1.0"

/* flag bits: same values as MM_F_* (minimap.h:8-38) for the subset this path uses */
#define AL_F_CIGAR         0x004
#define AL_F_OUT_SAM       0x008
#define AL_F_SR            0x1000
#define AL_F_FRAG_MODE     0x2000
#define AL_F_NO_PRINT_2ND  0x4000
#define AL_F_HEAP_SORT     0x400000
#define AL_F_SAM_HIT_ONLY  0x40000000

/* replaces mm_idxopt_t (minimap.h:101-105) */
typedef struct {
	short k, w, flag, bucket_bits;
	int mini_batch_size;
	uint64_t batch_size;
} al_idxopt_t;

/* replaces mm_mapopt_t (minimap.h:107-152); same field names and meaning */
typedef struct {
	int64_t flag;
	int seed;
	int bw;
	int max_gap, max_gap_ref;
	int max_frag_len;
	int max_chain_skip, max_chain_iter;
	int min_cnt;
	int min_chain_score;
	float mask_level;
	float pri_ratio;
	int best_n;
	int a, b, q, e, q2, e2;
	int sc_ambi;
	int zdrop, zdrop_inv;
	int end_bonus;
	int min_dp_max;
	float max_clip_ratio;
	int pe_ori, pe_bonus;
	int32_t mid_occ;
	int32_t max_occ;
	int mini_batch_size;
} al_mapopt_t;

/* replaces mm_reg1_t + mm_extra_t (minimap.h:74-98) */
typedef struct {
	int32_t id, cnt, rid, score;
	int32_t qs, qe, rs, re;
	int32_t parent, subsc;
	int32_t mlen, blen;
	int32_t n_sub, score0;
	uint32_t mapq:8, split:2, rev:1, inv:1, sam_pri:1, proper_frag:1, pe_thru:1, seg_split:1, seg_id:8, split_inv:1, dummy:7;
	uint32_t hash;
	int32_t dp_score, dp_max, dp_max2;
	uint32_t n_ambi;
	uint32_t n_cigar;
	uint32_t *cigar;             /* BAM-encoded, malloc()ed; NULL if n_cigar == 0 */
} al_reg1_t;

typedef struct al_idx_s al_idx_t;   /* replaces mm_idx_t (minimap.h:63-72); opaque, own HBM-oriented layout */
typedef struct al_ctx_s al_ctx_t;   /* replaces mm_tbuf_t (minimap.h:296-310): per-thread buffer = HIP stream + workspaces */

/* mm_set_opt (minimap.h:180): preset NULL = defaults; "sr"/"short" = AirLift's preset; else -1 */
int  al_set_opt(const char *preset, al_idxopt_t *io, al_mapopt_t *mo);
/* mm_check_opt (minimap.h:181): 0 if usable on this path, negative otherwise (message on stderr) */
int  al_check_opt(const al_idxopt_t *io, const al_mapopt_t *mo);

/* mm_idx_reader_open + mm_idx_reader_read (minimap.h:206-232) for a FASTA(.gz) file: NULL on failure */
al_idx_t *al_idx_build(const char *fn, const al_idxopt_t *io, int n_threads);
/* Same index, built on the GPU (sketch, sort and table construction as kernels; the host only parses the FASTA): the
 * index stays resident on `device` (< 0: LOCAL_RANK or 0) and is copied device-to-device if a context on another GPU
 * asks for it.  Needs odd k.  NULL (with a message) if HIP is unusable -- there is no host path behind this entry point. */
al_idx_t *al_idx_build_device(const char *fn, const al_idxopt_t *io, int device);
/* mm_idx_reader_open / mm_idx_reader_read / mm_idx_reader_eof / mm_idx_reader_close (minimap.h:206-232): the reference's
 * iterator over index parts.  This path builds one part: the first read() returns the whole index (built on `device`, as
 * al_idx_build_device), the next NULL.  A file that starts with the index magic (mm_idx_is_idx, index.c:531) is a prebuilt index and is loaded instead; fn_out
 * (minimap2's -d) makes read() write the index it built there.  open() returns NULL if the file cannot be opened. */
typedef struct al_idx_reader_s al_idx_reader_t;
al_idx_reader_t *al_idx_reader_open(const char *fn, const al_idxopt_t *io, const char *fn_out);
al_idx_t *al_idx_reader_read(al_idx_reader_t *r, int device);
int       al_idx_reader_eof(const al_idx_reader_t *r);
void      al_idx_reader_close(al_idx_reader_t *r);
/* mm_idx_dump / mm_idx_load / mm_idx_is_idx (index.c:438-552): the reference's index file, both ways -- an .mmi the fork wrote is read here, one written here is read
 * by the fork (the files differ in the order of a bucket's hash pairs only).  dump copies a device-built index off the GPU first; 0 / -1. */
int       al_idx_dump(const char *fn, al_idx_t *mi);
al_idx_t *al_idx_load(const char *fn);
int64_t   al_idx_is_idx(const char *fn);
int       al_idx_k(const al_idx_t *mi);      /* mm_idx_t::k, ::w (minimap.h:57) */
int       al_idx_w(const al_idx_t *mi);
/* Batch API: the caller will not read the sorted-anchor taps (al_dbg_anchors after a run) of this context, so the alignment stage's per-mate anchor arrays take the
 * place of their copy: 16 bytes per seed hit less (20 GB on a 1 M-pair batch against a human-sized reference).  The file drivers set it for their contexts. */
void      al_ctx_set_no_taps(al_ctx_t *ctx, int on);
/* mm_idx_cal_max_occ (index.c:164-185): (1-f) quantile of the per-minimizer occurrence counts + 1; INT32_MAX for f <= 0 */
int32_t   al_idx_cal_max_occ(const al_idx_t *mi, float f);
/* mm_mapopt_update (minimap.h:183, options.c:51-61): mid_occ <= 0 is replaced by al_idx_cal_max_occ(mi, 2e-4) */
void      al_mapopt_update(al_mapopt_t *opt, const al_idx_t *mi);
/* mm_idx_str (minimap.h:269) */
al_idx_t *al_idx_str(int w, int k, int n, const char **seq, const char **name);
/* mm_idx_destroy (minimap.h:291) */
void      al_idx_destroy(al_idx_t *mi);
uint32_t  al_idx_n_seq(const al_idx_t *mi);
const char *al_idx_seq_name(const al_idx_t *mi, uint32_t rid);
uint32_t  al_idx_seq_len(const al_idx_t *mi, uint32_t rid);
/* mm_idx_stat-like numbers: distinct minimizers, total positions, total bases */
void      al_idx_stat(const al_idx_t *mi, uint64_t *n_keys, uint64_t *n_pos, uint64_t *n_bases);
/* the occurrence array (all positions, grouped by minimizer hash ascending, ascending inside a group): returns the
 * number of entries and copies up to cap of them; for tests that compare the two builders */
int64_t   al_idx_export_pos(const al_idx_t *mi, uint64_t *dst, int64_t cap);

/* mm_tbuf_init / mm_tbuf_destroy (minimap.h:303,310).  device < 0: use LOCAL_RANK or 0.
 * Uploads the index to that device's HBM on first use.  NULL (with a message) if HIP is unusable. */
al_ctx_t *al_ctx_init(const al_idx_t *mi, const al_mapopt_t *opt, int device);
void      al_ctx_destroy(al_ctx_t *ctx);
/* host worker threads this context may use for packing a batch before upload (default 1); the analogue of the
 * n_threads argument of mm_map_file_frag for callers that drive batches themselves */
void      al_ctx_set_threads(al_ctx_t *ctx, int n_threads);

/* mm_map_frag (minimap.h:334): one fragment of n_segs (1 or 2) reads, reads given in sequencing
 * orientation; output identical in meaning to the reference incl. the FR flip of worker_for (map.c:458-498). */
void al_map_frag(const al_idx_t *mi, int n_segs, const int *qlens, const char **seqs, int *n_regs,
                 al_reg1_t **regs, al_ctx_t *ctx, const al_mapopt_t *opt, const char *qname);

/* Batch entry point (no reference analogue; the reference's kt_for over fragments, map.c:592).
 * reads are listed fragment-major: n_segs[f] reads for fragment f.  On return n_regs[i] / regs[i]
 * (i over reads) are filled like al_map_frag; rep_len[f] gets the fragment's repeat length.
 * Returns 0, or negative on error. */
int  al_map_batch(al_ctx_t *ctx, int n_frag, const int *n_segs, const int *qlens, const char *const *seqs,
                  const char *const *qnames, int *n_regs, al_reg1_t **regs, int *rep_len);

/* mm_map_file_frag (minimap.h:348): map 1 or 2 FASTA/FASTQ(.gz) files, SAM to `out` (header included
 * when rg != (char*)-1; pass rg = NULL for no @RG).  Returns 0, -1 if a file cannot be read. */
int  al_map_file_frag(const al_idx_t *mi, int n_segs, const char **fn, const al_mapopt_t *opt, int n_threads,
                      FILE *out, const char *rg, int device);

/* Same mapping, BAM on `out` instead of SAM text (BGZF blocks deflated at `level`, 0-9; n_threads workers).  sorted == 0: input
 * order, the same records as the SAM output.  sorted != 0: mapped records only, coordinate-sorted -- what AirLift produces with
 * `| samtools view -h -F4 | samtools sort -l5` (src/0-align_reads.sh:13, run_pipeline.sh:82-87). */
int  al_map_file_frag_bam(const al_idx_t *mi, int n_segs, const char **fn, const al_mapopt_t *opt, int n_threads,
                          FILE *out, const char *rg, int device, int sorted, int level);

/* Several GPUs of one node (the reference's analogue is kt_for over the fragments of a mini-batch, map.c:592, with
 * its serial ordered writer, map.c:601-644): lane r is a mapping context on devices[r] (a device may appear twice: two lanes
 * overlap transfers and kernels on it); every mini-batch is cut into n_dev contiguous fragment ranges, lane r maps range r.
 * The index is built once and copied device to device.  The lanes exchange only {records, bytes} of their output blocks per
 * batch (a 16-byte ncclAllGather over xGMI when they sit on distinct GPUs), take the exclusive prefix sum as their file offset
 * and pwrite() their block when `out` is a regular file (ordered turns otherwise).  bam_mode 0 SAM, 1 BAM, 2 sorted BAM.
 * Output bytes are those of al_map_file_frag / al_map_file_frag_bam. */
int  al_map_file_frag_multi(const al_idx_t *mi, int n_segs, const char **fn, const al_mapopt_t *opt, int n_threads,
                            FILE *out, const char *rg, const int *devices, int n_dev, int bam_mode, int level);

/* One process per GPU: rank `rank` of `world` maps a contiguous range of the input's fragments -- it finds its own byte
 * ranges of the plain FASTQ files (line counts of every rank's share, exchanged once) -- and writes its SAM text into `out_path` at the
 * offset the final all-gather of {ok, bytes} per rank gives it (rank-major = input order).  The exchanges are 16-byte all-gathers: RCCL
 * over xGMI when every rank has its own GPU, files in `rendezvous` (default: the output's directory) otherwise; a rank that does not
 * arrive within timeout_s (<= 0: AL_RANK_TIMEOUT or 600) makes the others fail.  device < 0: LOCAL_RANK or `rank`.  Output bytes are
 * those of al_map_file_frag on the whole input. */
int  al_map_file_frag_ranked(const al_idx_t *mi, int n_segs, const char **fn, const al_mapopt_t *opt, int n_threads, const char *out_path,
                             const char *rg, int device, int rank, int world, const char *rendezvous, double timeout_s);
/* The same with unsorted BAM output: every rank deflates its own records into whole BGZF blocks (rank 0's part carries the header, the last
 * rank's the EOF block), the parts go behind each other at the exchanged offsets.  Decoded records = those of al_map_file_frag_bam. */
int  al_map_file_frag_ranked_bam(const al_idx_t *mi, int n_segs, const char **fn, const al_mapopt_t *opt, int n_threads, const char *out_path,
                                 const char *rg, int device, int rank, int world, const char *rendezvous, double timeout_s, int bam_level);
/* Self-test of the range finding of al_map_file_frag_ranked without a GPU: `world` threads act as the ranks; 0 = the ranges tile the
 * files, start at records and pair record for record. */
int  al_dbg_ranked_selftest(const char *fn1, const char *fn2, int world, const char *dir);

/* ---- read extraction feeding the path (host-side; replaces per-region `samtools view | convert2bed | awk`) ---- */
/* src/4-extract_reads/extract_reads.sh:8 (prune != 0) / extract_reads_noprune.sh:7 (prune == 0) for all lines of a BED file in
 * one pass over the BAM: rows "chrom start end name[.1|.2] MAPQ CIGAR" of the mapped records that lie inside a BED line
 * (start >= B-1, end <= E-1) and, when pruning, have MAPQ <= 10 or a CIGAR other than "<read_size>M"; one row per name,
 * sorted by name (`sort -uk4,4`).  Returns the number of rows, negative on error. */
int64_t al_extract_reads(const char *bam_fn, const char *bed_fn, int read_size, int prune, FILE *out);
/* src/4-extract_reads/extract_sequence.sh:17-19: FASTQ subsets by the names in column 4 of `rows_fn` (seqtk subseq), pairing
 * (BBMap repair.sh) and renaming (rename.sh: realigned_<n>, realigned_singleton_<n>) into out_dir/reads_1.fastq,
 * reads_2.fastq, singletons.fastq.  Returns 0, negative on error. */
int  al_extract_sequence(const char *fq1, const char *fq2, const char *rows_fn, const char *out_dir, int64_t *n_pairs, int64_t *n_single);
/* N1 fused: extract_reads.sh:8 + extract_sequence.sh:17-19 in one call, nothing written to disk.  The three FASTQ texts
 * (pairs file 1, pairs file 2, singletons) are left in anonymous memory files whose descriptors go to fds[0..2]; map them by path
 * ("/proc/self/fd/<n>") with al_map_file_frag -- `airlift-align remap` does exactly that -- and close() them.  0, negative on error. */
int  al_extract_to_memory(const char *bam_fn, const char *bed_fn, int read_size, int prune, const char *fq1, const char *fq2, int fds[3], int64_t *n_pairs, int64_t *n_single);

/* Tap (parity tests): the extension DP alone -- ksw_extd2_sse's result for n caller-supplied pairs, as the reference's --print-aln-seq
 * shows them (align.c:313-339).  seqs: nt4 codes (0..4); jobs6: {target offset, query offset, tlen, qlen, ksw flag, 0} per pair (the
 * sequences as passed to ksw, i.e. already reversed for left extensions); out9 per pair: score, max, max_q, max_t, mqe, mqe_t, zdropped,
 * reach_end, n_cigar (-1: pair larger than 1024 x 512); cig_out: cig_cap words per pair.  Uses the context's options. */
int  al_dbg_ksw(al_ctx_t *ctx, int n, const uint8_t *seqs, size_t n_seq_bytes, const int32_t *jobs6, int32_t *out9, uint32_t *cig_out, int cig_cap);

/* Self-test of the multi-lane output path (offset exchange + pwrite, or ordered turns) with synthetic blocks; needs no GPU. */
int  al_dbg_ordered_out_selftest(const char *path, int n_lanes, int n_batches, int use_offsets);
/* The radix-sort restatement (klib ksort.h radix_sort, unstable above 64 elements) in its serial and its wavefront form on the same
 * keys: both permutations of 0..n-1 come back; n <= 65535.  Test hook. */
int  al_dbg_rs_sort(int device, const uint64_t *keys, int n, uint16_t *order_serial, uint16_t *order_wave);
/* Self-test of the whole-file parallel FASTA loader of the index builders against the block reader: 0 = same names, lengths and
 * bytes, 1 = the loader declined the file (not a plain uncompressed FASTA), -1 = they differ; needs no GPU. */
int  al_dbg_fasta_selftest(const char *fn, int n_threads);
/* Self-test of the block-parallel FASTQ parser of the host driver against its serial kseq-grammar reader on one file: 0 = the same
 * records, -1 = they differ, -2 = the file cannot be opened; needs no GPU. */
int  al_dbg_fastq_selftest(const char *fn, int n_threads);
/* Self-test of the SAM record formatter the GPU runs (k_sam_len / k_sam_write; mm_write_sam3, format.c:387-544), compiled for the CPU:
 * n_frag random fragments formatted by it and by al_write_sam, `de:f:%.4f` checked against printf; returns the number of
 * differences (0 = identical); needs no GPU. */
int  al_dbg_sam_selftest(uint64_t seed, int n_frag);

/* ---- device-resident batch API (bench / multi-GPU harness; inputs already in HBM when timing starts) ---- */
/* Pack + upload a batch; returns 0.  The batch stays resident until the next upload. */
int  al_batch_upload(al_ctx_t *ctx, int n_frag, const int *n_segs, const int *qlens, const char *const *seqs,
                     const char *const *qnames);
/* Same, for flat buffers: sequences concatenated in read order (not NUL-terminated), every read of fragment f
 * named "<name_prefix><first_index+f>" (BBMap rename.sh naming, extract_sequence.sh:18). */
int  al_batch_upload_flat(al_ctx_t *ctx, int n_frag, const int *n_segs, const int *qlens, const char *seq_concat,
                          const char *name_prefix, int64_t first_index);
/* Run the whole hot path (sketch .. alignment records) on the resident batch; results stay on device.
 * Returns 0, or AL_ERR_NOMEM when a device workspace for this batch could not be allocated (workspaces grow with the
 * number of seed hits: a batch of reads from high-copy repeats can need hundreds of bytes per hit).  The context gives its
 * batch workspaces back and stays usable: upload fewer fragments and run again -- the file drivers below halve the batch and retry on their own, the
 * way the reference's per-read loop (map.c:229-400) is unaffected by how many reads share a mini-batch.  Any other
 * non-zero value is a device error. */
#define AL_ERR_NOMEM (-12)
int  al_batch_run(al_ctx_t *ctx);
/* Fetch results of the last al_batch_run into host reg arrays (same contract as al_map_batch). */
int  al_batch_fetch(al_ctx_t *ctx, int *n_regs, al_reg1_t **regs, int *rep_len);

/* Device -> host copy of the flat result block of the last al_batch_run into page-locked buffers kept by the library (what the file
 * drivers do per batch); reports the records and bytes moved.  For timing the PCIe-inclusive rate. */
int  al_batch_fetch_flat(al_ctx_t *ctx, uint64_t *n_records, uint64_t *n_bytes);

/* ---- token batches: reads that are fixed-length windows of longer sequences, cut on the device ---- */
typedef struct al_winsrc_s al_winsrc_t;     /* source sequences, concatenated, packed 4 bit/base in the context's GPU memory */
al_winsrc_t *al_winsrc_create(al_ctx_t *ctx, const char *ascii, uint64_t n_bases);
void         al_winsrc_destroy(al_winsrc_t *src);
/* n_tok single-segment reads of read_len bases; read i = source bases [start[i], start[i] + read_len); qnames as in
 * al_batch_upload.  Then al_batch_run / al_batch_fetch as usual. */
int  al_batch_upload_windows(al_ctx_t *ctx, const al_winsrc_t *src, int n_tok, const uint64_t *start, int read_len, const char *const *qnames);
/* Whole stage: what AirLift does with gaps_to_fasta.py (tile every sequence of `gaps_fn` into read_size-base tokens every `skip`
 * bases, named <sequence>_<start>) followed by the single-end alignment of the tokens (align_gaps.sh:14-15); SAM on `out`. */
int  al_map_tokens_file(const al_idx_t *mi, const char *gaps_fn, int read_size, int skip, const al_mapopt_t *opt, int n_threads,
                        FILE *out, const char *rg, int device);

/* per-batch work counters + per-kernel HIP-event times of the last al_batch_run */
typedef struct {
	uint64_t n_frag, n_reads, n_bases;
	uint64_t n_mini, n_anchor, n_chain, n_regs_aln, n_refbases, n_cigar, n_rechain, n_heap_fallback, n_sort_tie_flag;
	uint64_t bytes_in, bytes_out;
	double   algorithmic_bytes;       /* SURVEY.md §8(d) formula evaluated with the counters above */
	float    ms_total;                /* HIP events around the whole device pipeline */
	float    ms_kernel[40];           /* HIP-event time per pipeline interval, see al_stage_name() / al_stage_kernel() */
	int      n_stage;
	float    ms_side_stream;          /* exact (serial) heap merge + whole-fragment chaining of the fragments with equal-x anchors: runs on a side stream, overlapped with the intervals above */
	uint64_t n_chain_fallback;        /* fragments re-chained whole because of equal-x chain starts among more than 64 chains */
	uint64_t dp_jobs[10];             /* extension DP jobs per class (lane 16/32/64 targets, group DP of 1/2/4/8/22/32 16-column blocks, LDS rows) ... */
	uint64_t dp_target_bases[10];     /* ... and the reference-window bases of those jobs: the W term of the algorithmic bytes, per DP kernel */
} al_batch_stat_t;
void al_batch_stat(const al_ctx_t *ctx, al_batch_stat_t *st);
const char *al_stage_name(int i);
/* the kernel that runs in interval i (as rocprofv3 --kernel-trace names it); "" where an interval is a few small launches */
const char *al_stage_kernel(int i);

/* ---- the fork's as-shipped observable: number of seed clusters that would go on to alignment ---- */
/* Seed stages only on the resident batch (upload it with one segment per fragment, as the fork's main loop maps read by read,
 * main.c:384-391); *total gets the batch's count (map.c:299-312). */
int  al_batch_count_candidates(al_ctx_t *ctx, int64_t *total);
/* Whole file: the value the unmodified fork prints as "Total No. of Mappings before alignment (verification)" (main.c:417).
 * Returns 0, negative on error. */
int  al_count_candidates_file(const al_idx_t *mi, const char *fn, const al_mapopt_t *opt, int n_threads, int device, int64_t *total);

/* the pre-alignment filters of the bundled mrFAST fork applied to those candidates, on the device (al_prefilter.hip): the
 * adjacency filter (MrFAST.c:1741-1764: every other seed of the read looked up at the position the candidate predicts; more than adj_e
 * absent seeds reject) and GreedySnake (GreedySnake.c:52-200 with EditThreshold snake_e, KmerSize snake_k, IterationNo snake_iter).
 * out4: candidates (= the count above), kept by adjacency, kept by GreedySnake, kept by both.  Resident batch of single-segment
 * fragments (first, mid_occ, seeding pass); the alignment path does not use it. */
int  al_batch_prefilter(al_ctx_t *ctx, int adj_e, int snake_e, int snake_k, int snake_iter, int64_t *out4);
/* ... and every candidate's two decisions: dec[0 .. min(out4[0], dec_cap)) = fragment << 32 | first anchor of the candidate's cluster << 2 |
 * kept by the adjacency filter << 1 | kept by GreedySnake (no particular order) */
int  al_batch_prefilter_decisions(al_ctx_t *ctx, int adj_e, int snake_e, int snake_k, int snake_iter, int64_t *out4, uint64_t *dec, int64_t dec_cap);
int  al_count_candidates_file_filtered(const al_idx_t *mi, const char *fn, const al_mapopt_t *opt, int n_threads, int device,
                                       int adj_e, int snake_e, int snake_k, int snake_iter, int64_t *out4);

/* ---- stage taps for parity tests (analogue of --print-seeds, map.c:333-338,381-385) ---- */
/* minimizers of read i of the resident batch: returns count, writes up to cap records (x = hash<<8|span, y = i<<32|pos<<1|strand) */
int  al_dbg_minimizers(al_ctx_t *ctx, int read_idx, uint64_t *xy, int cap);
/* sorted anchors ("SD") and chained anchors ("CN") of fragment f after al_batch_run; x,y interleaved */
int  al_dbg_anchors(al_ctx_t *ctx, int frag_idx, uint64_t *xy, int cap, int *rep_len);
int  al_dbg_chains(al_ctx_t *ctx, int frag_idx, uint64_t *u, int cap_u, uint64_t *xy, int cap_a);
/* a8 ALSER counter (map.c:299-312) for every read of the resident batch mapped as single segments */
int  al_dbg_alser_count(al_ctx_t *ctx, int64_t *total);
/* bulk copy of a named device array of the resident batch (tests); returns bytes copied or -1 */
int64_t al_dbg_copy(al_ctx_t *ctx, const char *name, void *dst, int64_t max_bytes);

/* SAM text (format.c:116-135, 387-544) */
#define AL_MM_VERSION "2.17-r954-dirty"   /* MM_VERSION of the fork (main.c:16): the version whose records this path reproduces */
/* the `ver` and `argc, argv` arguments of mm_write_sam_hdr (format.c:116-135; main.c:369 passes MM_VERSION and its own
 * argv): `\tVN:<ver>` and `\tCL:minimap2 <argv[1..]>` appended to the @PG line of every header written afterwards.
 * Process-wide; NULL / argc <= 1 leave the respective part out (the default). */
void al_set_program_line(const char *ver, int argc, char *const *argv);
int  al_write_sam_hdr(FILE *out, const al_idx_t *mi, const char *rg, char *rg_id_out /* >=256 bytes or NULL */);
int  al_write_sam(char *buf, size_t cap, const al_idx_t *mi, const char *qname, int l_seq, const char *seq, const char *qual,
                  int seg_idx, int reg_idx, int n_seg, const int *n_regss, const al_reg1_t *const *regss,
                  const char *rg_id, int rep_len);

const char *al_version(void);

/* Device memory reserve (no reference analogue: kalloc's per-thread arenas, kalloc.c:38-205 / map.c:13-36, are the nearest thing).  Starts a
 * thread that obtains `bytes` of memory on `device` (< 0: LOCAL_RANK or 0) in a few large chunks and returns at once; every device range the
 * library uses afterwards -- index arrays, the index builder's temporaries, the mapping contexts' workspaces -- is served from those chunks
 * (requests that find no room fall back to the driver).  Meant to be called before al_idx_build_device so that obtaining the memory overlaps
 * the FASTA load and the index build.  One reserve per process; 0, or -1 if there is no usable device / a reserve on another device exists.
 * al_device_reserve_for_run sizes it from the reference and read files (bytes it asked for, 0 = run too small to be worth it, -1 = error). */
int al_device_reserve(int device, uint64_t bytes);
int64_t al_device_reserve_for_run(int device, const char *ref_fn, int n_fn, const char *const *fn);
uint64_t al_device_reserve_peak(void);        /* largest number of bytes in use at once so far (0 without a reserve) */
void al_device_reserve_reset_peak(void);
void al_device_reserve_report(FILE *fp);      /* one line: chunks, time, peak use */

#ifdef __cplusplus
}
#endif
#endif
