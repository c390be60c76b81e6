// al_bam.h -- BAM record / BGZF writer used by the file-level driver (product code)
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#include "al_internal.h"

struct AlBgzf {                 // BGZF stream: bytes in, 64 KB blocks deflated on n_threads workers, written in order
	FILE *out; int level, n_threads;
	std::vector<char> buf; size_t cap;
	AlBgzf(FILE *o, int lvl, int nt) : out(o), level(lvl), n_threads(nt > 1 ? nt : 1), cap((size_t)0xff00 * 64 * (size_t)(nt > 1 ? nt : 1)) { buf.reserve(cap); }
	int write(const char *p, size_t n);
	int finish();                // flushes the tail and appends the EOF block
	int flush_all();             // everything written so far goes out as whole blocks (the tail as a short one), no EOF block: what follows may come from other writers
private:
	int flush_full();
};

// n bytes of BAM records as a sequence of whole BGZF blocks (a record may span blocks), deflated on n_threads workers, appended to dst: a lane's
// share of a batch becomes a byte range that can be written at any offset of the output ("BGZF blocks are rank-local")
int al_bgzf_blocks(const char *src, size_t n, int level, int n_threads, std::vector<char> &dst);
extern const unsigned char AL_BGZF_EOF[28];
int al_bam_header(AlBgzf &z, const al_idx_t *mi, const char *rg, char *rg_id, bool sorted);
int al_write_bam_rec(std::vector<char> &out, const al_idx_t *mi, const char *qname, int l_seq, const char *seq, const char *qual,
                     int seg_idx, int reg_idx, int n_seg, const int *n_regss, const al_reg1_t *const *regss, const char *rg_id, int rep_len,
                     uint64_t *key, int *unmapped);
// stable sort permutation of n 64-bit keys, radix-sorted on the context's GPU (al_runtime.hip)
int al_sort_keys(al_ctx_t *c, const uint64_t *keys, uint32_t *perm, size_t n);
