// al_stream_pipe.h -- the file driver with input parsing and SAM text on the GPU (al_stream_pipe.cpp)
#pragma once
#include <stdio.h>
#include "../../include/airlift.h"
#define AL_STREAM_NA 77          // not applicable: nothing was read or written, the caller uses the host driver
struct AlStreamResume { bool resume = false; long long off[2] = {0, 0}; char rg_id[256] = {0}; };
// a byte range of every input file (a record starts at start[i]; end[i] < 0: to the end of the file) and whether the SAM header is
// printed: what one process of a multi-process run takes (al_ranked.cpp)
// (round 6) one process of a multi-process run writing into the ONE output file: the driver tells the size of every batch it has mapped and learns
// where its text goes (al_ranked.cpp: one all-gather of {ok, bytes} per round over the ranks -- SURVEY.md 8e, map.c:601-644 keeps input order the same way).
// offset_of(ctx, round, bytes, pre_bytes, ok): bytes of this process's batch of that round (0: it has none), pre_bytes: what it has already written at the
// start of the file (the header; round 0 of the rank that prints it), ok = 0: this process failed; returns the file offset of the batch, < 0: stop (a peer
// failed or did not arrive).  Every process calls it once per round, n_rounds times.
struct AlStreamBatchSink { long long (*offset_of)(void *ctx, uint64_t round, uint64_t bytes, uint64_t pre_bytes, int ok); void *ctx; uint64_t n_rounds; };
struct AlStreamRange {
	long long start[2] = {0, 0}, end[2] = {-1, -1}; bool header = true;
	// (round 6) instead: a LIST of byte ranges of every file, each of which starts and ends at a record and is mapped as ONE batch -- the batches of the
	// common grid that are this process's -- and the sink that places each batch's text
	bool list = false; int n_ranges = 0; const long long *rstart[2] = {nullptr, nullptr}, *rend[2] = {nullptr, nullptr}; const AlStreamBatchSink *sink = nullptr;   // (list with n_ranges == 0: no batch of its own, the rounds of the sink only)
};
// mm_map_file_frag (map.c:672-700) for plain uncompressed four-line FASTQ files -> SAM text.  0 = done (rs->resume: the rest of the
// input, from rs->off, is for the general reader; the header is out), AL_STREAM_NA, or a negative error.
int al_stream_map_files(const al_idx_t *mi, int n_fn, const char **fn, const al_mapopt_t *opt, int n_threads, FILE *out, const char *rg,
                        const int *devices, int n_dev, AlStreamResume *rs, const AlStreamRange *range = nullptr);
// one part of a concatenated unsorted BAM (al_pipeline.cpp): the byte range [start, end) of the inputs through the host driver
int al_map_file_frag_bam_part(const al_idx_t *mi, int n_fn, const char **fn, const al_mapopt_t *opt, int n_threads, FILE *out, const char *rg, int device, int level,
                              const long long *start, const long long *end, bool header, bool eof);
