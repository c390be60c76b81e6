// al_stream_pipe.h -- the file driver with input parsing and SAM text on the GPU (al_stream_pipe.cpp)
#pragma once
#include <stdio.h>
#include "../../include/airlift.h"
#define AL_STREAM_NA 77          // not applicable: nothing was read or written, the caller uses the host driver
struct AlStreamResume { bool resume = false; long long off[2] = {0, 0}; char rg_id[256] = {0}; };
// a byte range of every input file (a record starts at start[i]; end[i] < 0: to the end of the file) and whether the SAM header is
// printed: what one process of a multi-process run takes (al_ranked.cpp)
struct AlStreamRange { long long start[2] = {0, 0}, end[2] = {-1, -1}; bool header = true; };
// mm_map_file_frag (map.c:672-700) for plain uncompressed four-line FASTQ files -> SAM text.  0 = done (rs->resume: the rest of the
// input, from rs->off, is for the general reader; the header is out), AL_STREAM_NA, or a negative error.
int al_stream_map_files(const al_idx_t *mi, int n_fn, const char **fn, const al_mapopt_t *opt, int n_threads, FILE *out, const char *rg,
                        const int *devices, int n_dev, AlStreamResume *rs, const AlStreamRange *range = nullptr);
// one part of a concatenated unsorted BAM (al_pipeline.cpp): the byte range [start, end) of the inputs through the host driver
int al_map_file_frag_bam_part(const al_idx_t *mi, int n_fn, const char **fn, const al_mapopt_t *opt, int n_threads, FILE *out, const char *rg, int device, int level,
                              const long long *start, const long long *end, bool header, bool eof);
