// al_stream.h -- device-side input and output of the drop-in's file driver (product code).
//
// The reference's step 0 and step 2 (map.c:561-589: mm_bseq_read_frag2 + 4-bit packing, bseq.c:56-130, sketch.c:9-26;
// map.c:594-650: mm_write_sam3, format.c:387-544) are byte shuffling on every read's critical path.  Here the host only moves
// raw file bytes: FASTQ blocks go to HBM as they are, kernels find the records, pack the bases and hash the names (rows a1, a4 of
// SURVEY.md 8a), and after the mapping kernels the SAM text (row a22) is written by kernels as well; the host write()s it.
//
// One AlStreamSlot = the text and SAM buffers of one batch in flight, with a stream of its own for the transfers and the parsing
// kernels; the mapping kernels run in a mapping context (al_ctx_t: streams + workspaces, ~100 bytes per seed hit) that batches take
// turns on.  A lane (one GPU) has several slots per context, so H2D + parsing of batch n+1 and D2H + write() of batch n-1 overlap the
// mapping of batch n (the reference's three-step kt_pipeline, map.c:557-652, kthread.c:130-159) while only one batch's workspaces
// exist: what a process pays for device memory is per byte it ever touches (the driver maps and scrubs at 20-40 GB/s).
#pragma once
#include <stdint.h>
#include <vector>
#include "al_internal.h"
#include "al_runtime.h"

struct AlFqRec { uint32_t name, name_len, seq, len, qual; };      // one FASTQ record: offsets into its file's text
struct AlRdText { uint32_t name, name_len, seq, qual; };          // one read (fragment-major order): offsets into the text of its file
struct AlBulk { uint64_t dst; uint32_t src; uint32_t len_flags; }; // SEQ / QUAL field of a SAM record: copied by k_sam_bulk (len | file << 27 | u2t << 28 | comp << 29 | rev << 30)

// rd_info bits (per read)
#define AL_RI_FLIP   1u      // mapped reverse-complemented (map.c:468): un-flip at output (map.c:486-497)
#define AL_RI_SEG1   2u      // second read of its fragment
#define AL_RI_PAIRED 4u      // fragment has two reads

struct AlIngestResult {
	uint64_t lines[2];               // complete lines in each file's text
	uint64_t first_bad[2];           // first record that is not strict four-line FASTQ (~0: none)
	uint64_t n_rec[2];               // complete records available (before the first bad one)
	int      n_frag, n_reads;        // what this batch takes
	uint64_t consumed[2];            // bytes of each text the batch consumed (the rest is the next batch's carry)
};

struct AlStreamSlot {
	int device = 0; hipStream_t io = nullptr; hipEvent_t ev = nullptr;
	hipEvent_t ev_out[2] = {nullptr, nullptr};   // 'piece copied' events of the SAM drain, created on THIS slot's device (an event must be recorded on a stream of its own device: the writer's shared ring must not own them when the lanes sit on different GPUs)
	const al_idx_t *mi = nullptr;
	int n_files = 1;
	// input
	DevBuf<uint8_t> txt[2]; uint64_t txt_n[2] = {0, 0};
	DevBuf<uint32_t> ls[2];                                   // line starts: ls[j] = offset of line j; ls[lines] = end of the complete lines
	DevBuf<uint32_t> tile_cnt; DevBuf<uint64_t> tile_off;
	DevBuf<AlFqRec> frec[2];
	DevBuf<AlRdText> rtxt; DevBuf<uint8_t> rd_info; DevBuf<uint32_t> rd_frag, rd_words, rd_mcnt, se_key, se_run, se_fs, se_fidx;
	DevBuf<unsigned long long> st;                            // counters of the ingest kernels
	DevBuf<uint8_t> tabs;                                     // nt4 table (256 bytes) + complement table (256 bytes)
	DevBuf<uint8_t> scan_tmp;
	// output
	DevBuf<char> names; DevBuf<uint32_t> name_off; DevBuf<char> rg;
	DevBuf<uint32_t> sam_len, sam_nrec; DevBuf<uint64_t> sam_off, rec_off; DevBuf<AlBulk> bulk; DevBuf<char> sam;
	uint64_t sam_bytes = 0, sam_records = 0;
	bool cfg_ready = false; int rg_len = 0;
	void release();
};

int  al_stream_slot_init(AlStreamSlot &S, const al_idx_t *mi, int device, int n_files);
void al_stream_slot_destroy(AlStreamSlot &S);
// text of file i: begin (room for cap_bytes), then host pieces appended in file order (each call returns when its copy is done)
int  al_stream_begin_text(AlStreamSlot &S, int i, size_t cap_bytes);
int  al_stream_append_text(AlStreamSlot &S, int i, const char *p, size_t n);
// line index + record parse of the loaded text; decides what the batch takes (at most max_reads reads).  eof[i]: no more bytes of file i
// will follow (a trailing unpaired read is taken; an unterminated last line counts).  Blocks until the device has answered.
int  al_stream_parse(AlStreamSlot &S, const bool *eof, int max_reads, AlIngestResult *res);
// bytes [from, from + n) of file i's device text back to the host (the carry of the next batch)
int  al_stream_fetch_text(AlStreamSlot &S, int i, uint64_t from, uint64_t n, char *dst);
// fragment / read arrays of the mapping context from the parsed records [rec_lo, rec_hi) = fragments [frag_lo, frag_hi) (names hashed,
// bases packed 4 bit/base in mapping orientation); the whole batch is (0, n_reads or n_frag, 0, n_frag)
int  al_stream_setup(AlStreamSlot &S, al_ctx_t *c, uint32_t rec_lo, uint32_t rec_hi, uint32_t frag_lo, uint32_t frag_hi);
// single-file input: the record each fragment of the parsed text starts at (frag_first, n_frag + 1 entries), for cutting a batch
int  al_stream_frag_starts(AlStreamSlot &S, const AlIngestResult &res, std::vector<uint32_t> &first);
// SAM text of the batch context c has mapped (after al_batch_run) into S.sam (device): kernels on c's stream; the slot's stream is made
// to wait for them and the call returns (c is free for its next batch); sets sam_bytes / sam_records.
int  al_stream_sam(AlStreamSlot &S, al_ctx_t *c, const char *rg_id);
// bytes [off, off + n) of the text to a page-locked host buffer, on the slot's stream; `done` is recorded behind the copy
int  al_stream_sam_fetch(AlStreamSlot &S, uint64_t off, uint64_t n, char *dst, hipEvent_t done);

// device result arrays of the last al_batch_run (al_kernels_align.hip)
struct AlDevResult { const AlReg *out; const uint64_t *out_off; const uint32_t *arena; uint64_t out_total; };
int  al_align_result(al_ctx_t *c, AlDevResult *r);
