// al_stream.hip -- FASTQ text -> packed read batch and alignment records -> SAM text, both on the GPU (see al_stream.h).
//
// Input kernels (rows a1 / a4 of SURVEY.md 8a: kseq2bseq bseq.c:56-130, seq_nt4_table sketch.c:9-26, qname hash map.c:291-293):
//   k_nl_count / k_nl_fill   newline index of the raw text, 4 KB tiles: count, scan, fill              (HBM streaming, 16 B / lane)
//   k_fq_parse               lane per record: the four lines checked against the strict grammar, name / sequence / quality offsets
//   k_setup_pe / k_setup_se  fragment-major read arrays, fragment boundaries (adjacent equal names in single-file input, map.c:580-586)
//   k_pack_reads             32 lanes per read: 4-bit codes, mate 2 reverse-complemented for mapping (map.c:468, bseq.h:46-58)
// Output kernels (row a22: mm_write_sam3 format.c:387-544):
//   k_sam_len                lane per read: bytes and records of its SAM text (al_dev_sam.h, counting sink)
//   k_sam_write              lane per read: every field but SEQ / QUAL, which are left as copy descriptors
//   k_sam_bulk               wavefront per descriptor: SEQ / QUAL bytes, reversed / complemented as the record needs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <vector>
#include <string>
#include "al_internal.h"
#include "al_device.h"
#include "al_runtime.h"
#include "al_io.h"
#include "al_stream.h"
#include "al_dev_sam.h"

#define NL_TILE 4096

// ---- newline index ----------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t d_nl_mask(uint32_t x)
{   // 0x80 in every byte of x that equals '\n' (exact: no carries between bytes)
	const uint32_t y = x ^ 0x0a0a0a0au;
	return ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y | 0x7f7f7f7fu);
}
__device__ __forceinline__ uint4 d_load16(const uint8_t *__restrict__ t, uint64_t n, uint64_t at)
{   // 16 bytes at `at` (16-byte aligned), bytes at or beyond n read as zero
	uint4 v = make_uint4(0, 0, 0, 0);
	if (at + 16 <= n) v = *reinterpret_cast<const uint4 *>(t + at);
	else if (at < n) { uint32_t w[4] = {0, 0, 0, 0}; for (uint64_t j = at; j < n; ++j) w[(j - at) >> 2] |= (uint32_t)t[j] << (8 * ((j - at) & 3)); v = make_uint4(w[0], w[1], w[2], w[3]); }
	return v;
}
__global__ void __launch_bounds__(256)
k_nl_count(const uint8_t *__restrict__ t, uint64_t n, uint32_t *__restrict__ tile_cnt)
{
	const uint64_t at = (uint64_t)blockIdx.x * NL_TILE + threadIdx.x * 16;
	const uint4 v = d_load16(t, n, at);
	uint32_t c = __popc(d_nl_mask(v.x)) + __popc(d_nl_mask(v.y)) + __popc(d_nl_mask(v.z)) + __popc(d_nl_mask(v.w));
	for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
	__shared__ uint32_t s[4];
	if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
	__syncthreads();
	if (threadIdx.x == 0) tile_cnt[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ void __launch_bounds__(256)
k_nl_fill(const uint8_t *__restrict__ t, uint64_t n, const uint64_t *__restrict__ tile_off, uint32_t *__restrict__ ls)
{
	const uint64_t at = (uint64_t)blockIdx.x * NL_TILE + threadIdx.x * 16;
	const uint4 v = d_load16(t, n, at);
	const uint32_t m[4] = {d_nl_mask(v.x), d_nl_mask(v.y), d_nl_mask(v.z), d_nl_mask(v.w)};
	const uint32_t c = __popc(m[0]) + __popc(m[1]) + __popc(m[2]) + __popc(m[3]);
	uint32_t inc = c;                                                            // inclusive scan inside the wavefront
	for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d); if ((int)(threadIdx.x & 63) >= d) inc += o; }
	__shared__ uint32_t s[4];
	if ((threadIdx.x & 63) == 63) s[threadIdx.x >> 6] = inc;
	__syncthreads();
	uint32_t base = inc - c;
	for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += s[w];
	uint64_t o = tile_off[blockIdx.x] + base + 1;                                // ls[0] = 0 is line 0
	if (blockIdx.x == 0 && threadIdx.x == 0) ls[0] = 0;
	if (c == 0) return;
	for (int w = 0; w < 4; ++w) {
		uint32_t mm = m[w];
		while (mm) { const int b = __ffs(mm) - 1; mm &= mm - 1; ls[o++] = (uint32_t)(at + 4 * w + (b >> 3) + 1); }
	}
}

// ---- records ----------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool d_is_ws(uint8_t c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

// One strict record = lines 4r .. 4r+3: '@' name [comment], sequence, '+' [anything], quality of the sequence's length (a trailing
// '\r' of a line is dropped).  Anything else makes r the first bad record: the driver maps what lies in front of it and gives the
// rest of the input to the general (kseq grammar) reader.
__global__ void __launch_bounds__(256)
k_fq_parse(const uint8_t *__restrict__ t, const uint32_t *__restrict__ ls, uint32_t n_rec, AlFqRec *__restrict__ out, unsigned long long *__restrict__ first_bad)
{
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_rec) return;
	const uint32_t p0 = ls[4 * r], p1 = ls[4 * r + 1], p2 = ls[4 * r + 2], p3 = ls[4 * r + 3], p4 = ls[4 * r + 4];
	bool ok = t[p0] == '@' && t[p2] == '+';
	uint32_t s1 = p2 - 1, q1 = p4 - 1;                                           // ends without the newline
	if (s1 > p1 && t[s1 - 1] == '\r') --s1;
	if (q1 > p3 && t[q1 - 1] == '\r') --q1;
	if (s1 - p1 != q1 - p3) ok = false;
	uint32_t e = p0 + 1; const uint32_t le = p1 - 1;
	while (e < le && !d_is_ws(t[e])) ++e;
	AlFqRec o; o.name = p0 + 1; o.name_len = e - (p0 + 1); o.seq = p1; o.len = s1 - p1; o.qual = p3;
	out[r] = o;
	if (!ok) atomicMin(first_bad, (unsigned long long)r);
}

__device__ __forceinline__ uint32_t d_wang(uint32_t key)
{   // khash.h:398-408
	key += ~(key << 15); key ^= (key >> 10); key += (key << 3); key ^= (key >> 6); key += ~(key << 11); key ^= (key >> 16);
	return key;
}
__device__ __forceinline__ uint32_t d_qname_hash(const uint8_t *__restrict__ s, uint32_t l, int qlen_sum, int seed)
{   // map.c:291-293: X31 string hash (khash.h:383-389) ^ Wang(qlen_sum) + Wang(seed), Wang again
	uint32_t h = 0;
	if (l) { h = s[0]; for (uint32_t i = 1; i < l; ++i) h = (h << 5) - h + (uint32_t)s[i]; }
	h ^= d_wang((uint32_t)qlen_sum) + d_wang((uint32_t)seed);
	return d_wang(h);
}

struct SetupOut {
	AlRdText *rtxt; uint8_t *rd_info; uint32_t *rd_frag, *rd_len, *rd_words, *rd_mcnt, *frag_first, *frag_hash;
	unsigned long long *st;        // [4] longest read, [5] longest fragment, [6] bases, [7] input bytes (2 bits + N mask per base)
};
__device__ __forceinline__ void d_setup_read(const SetupOut &O, uint32_t i, uint32_t f, const AlFqRec &q, uint32_t info, int k)
{
	O.rtxt[i] = AlRdText{q.name, q.name_len, q.seq, q.qual};
	O.rd_info[i] = (uint8_t)info; O.rd_frag[i] = f; O.rd_len[i] = q.len;
	O.rd_words[i] = (q.len + 7) / 8 + 1;
	O.rd_mcnt[i] = (q.len >= (uint32_t)k ? q.len - (uint32_t)k + 1 : 0u) + 1u;
}
__device__ __forceinline__ void d_setup_stats(const SetupOut &O, uint32_t lmax, uint32_t qsum, uint64_t bases, uint64_t bin)
{
	for (int d = 32; d > 0; d >>= 1) { lmax = max(lmax, (uint32_t)__shfl_xor((int)lmax, d)); qsum = max(qsum, (uint32_t)__shfl_xor((int)qsum, d)); bases += __shfl_xor(bases, d); bin += __shfl_xor(bin, d); }
	if ((threadIdx.x & 63) == 0) { atomicMax(O.st + 4, (unsigned long long)lmax); atomicMax(O.st + 5, (unsigned long long)qsum); atomicAdd(O.st + 6, (unsigned long long)bases); atomicAdd(O.st + 7, (unsigned long long)bin); }
}
// two files in lock step (bseq.c:129-167): fragment f = record f of file 0 + record f of file 1
__global__ void __launch_bounds__(256)
k_setup_pe(const uint8_t *__restrict__ t0, const AlFqRec *__restrict__ r0, const AlFqRec *__restrict__ r1, uint32_t n_frag, SetupOut O, int k, int seed, int pe_ori)
{
	const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t lmax = 0, qsum = 0; uint64_t bases = 0, bin = 0;
	if (f < n_frag) {
		const AlFqRec a = r0[f], b = r1[f];
		d_setup_read(O, 2 * f, f, a, AL_RI_PAIRED | ((pe_ori >> 1 & 1) ? AL_RI_FLIP : 0u), k);
		d_setup_read(O, 2 * f + 1, f, b, AL_RI_PAIRED | AL_RI_SEG1 | ((pe_ori & 1) ? AL_RI_FLIP : 0u), k);
		qsum = a.len + b.len; lmax = max(a.len, b.len); bases = qsum; bin = (a.len * 3 + 7) / 8 + (b.len * 3 + 7) / 8;
		O.frag_first[f] = 2 * f;
		O.frag_hash[f] = d_qname_hash(t0 + a.name, a.name_len, (int)qsum, seed);
	}
	if (f == n_frag) { O.frag_first[f] = 2 * n_frag; O.rd_len[2 * n_frag] = 0; O.rd_words[2 * n_frag] = 0; O.rd_mcnt[2 * n_frag] = 0; }
	d_setup_stats(O, lmax, qsum, bases, bin);
}
// one file: adjacent records with the same name (a trailing /1 /2 ignored, bseq.h:31-36) form a fragment, greedily from the front
__device__ __forceinline__ uint32_t d_qname_len(const uint8_t *__restrict__ s, uint32_t l) { return l >= 3 && s[l - 1] >= '0' && s[l - 1] <= '9' && s[l - 2] == '/' ? l - 2 : l; }
__global__ void __launch_bounds__(256)
k_se_same(const uint8_t *__restrict__ t, const AlFqRec *__restrict__ rec, uint32_t n, uint32_t *__restrict__ key)
{   // key[r] = r if record r starts a run of equal names, else 0 (an inclusive max scan then gives every record its run's start)
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n) return;
	bool same = false;
	if (r > 0) {
		const AlFqRec a = rec[r - 1], b = rec[r];
		const uint32_t la = d_qname_len(t + a.name, a.name_len), lb = d_qname_len(t + b.name, b.name_len);
		if (la == lb) { same = true; for (uint32_t i = 0; i < la; ++i) if (t[a.name + i] != t[b.name + i]) { same = false; break; } }
	}
	key[r] = same ? 0u : r;
}
__global__ void __launch_bounds__(256)
k_se_flag(const uint32_t *__restrict__ run, uint32_t n, uint32_t *__restrict__ fs)
{   // fs[r] = 1 if record r is the first read of a fragment: even position in its run (runs of three or more are cut in twos from the front)
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r < n) fs[r] = ((r - run[r]) & 1u) ? 0u : 1u; else if (r == n) fs[r] = 1u;
}
__global__ void __launch_bounds__(256)
k_setup_se(const uint8_t *__restrict__ t, const AlFqRec *__restrict__ rec, const uint32_t *__restrict__ fs, const uint32_t *__restrict__ fidx, uint32_t n, uint32_t n_frag, uint32_t f0, SetupOut O, int k, int seed, int pe_ori)
{
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t lmax = 0, qsum = 0; uint64_t bases = 0, bin = 0;
	if (r < n) {
		const AlFqRec a = rec[r];
		const bool first = fs[r] != 0;
		const uint32_t f = (first ? fidx[r] : fidx[r] - 1) - f0;                 // fidx = exclusive sum of fs over the whole parsed text; f0 = its value at the range's first record
		const bool paired = first ? (r + 1 < n && fs[r + 1] == 0) : true;
		uint32_t info = 0;
		if (paired) info = AL_RI_PAIRED | (first ? 0u : AL_RI_SEG1) | (((first ? pe_ori >> 1 : pe_ori) & 1) ? AL_RI_FLIP : 0u);
		d_setup_read(O, r, f, a, info, k);
		lmax = a.len; bases = a.len; bin = (a.len * 3 + 7) / 8;
		if (first) {
			qsum = a.len + (paired ? rec[r + 1].len : 0u);
			O.frag_first[f] = r;
			O.frag_hash[f] = d_qname_hash(t + a.name, a.name_len, (int)qsum, seed);
		}
	}
	if (r == n) { O.frag_first[n_frag] = n; O.rd_len[n] = 0; O.rd_words[n] = 0; O.rd_mcnt[n] = 0; }
	d_setup_stats(O, lmax, qsum, bases, bin);
}

// 32 lanes per read: word w of the read = bases 8w .. 8w+7 in mapping orientation
__global__ void __launch_bounds__(256)
k_pack_reads(const uint8_t *__restrict__ t0, const uint8_t *__restrict__ t1, const AlRdText *__restrict__ rtxt, const uint8_t *__restrict__ rd_info, const uint32_t *__restrict__ rd_len,
             const uint64_t *__restrict__ rd_off, uint32_t n_reads, int two_files, const uint8_t *__restrict__ tabs, uint32_t *__restrict__ rd_seq)
{
	__shared__ uint8_t nt4[256];
	nt4[threadIdx.x] = tabs[threadIdx.x];
	__syncthreads();
	const uint32_t i = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
	if (i >= n_reads) return;
	const uint32_t info = rd_info[i], L = rd_len[i], nw = (L + 7) / 8 + 1;
	const uint8_t *s = ((two_files && (info & AL_RI_SEG1)) ? t1 : t0) + rtxt[i].seq;
	uint32_t *w = rd_seq + rd_off[i];
	const bool flip = (info & AL_RI_FLIP) != 0;
	for (uint32_t b = l; b < nw; b += 32) {
		uint32_t v = 0; const uint32_t j0 = b * 8, j1 = j0 + 8 < L ? j0 + 8 : L;
		if (!flip) for (uint32_t j = j0; j < j1; ++j) v |= (uint32_t)nt4[s[j]] << ((j & 7) << 2);
		else for (uint32_t j = j0; j < j1; ++j) { const uint32_t cd = nt4[s[L - 1 - j]]; v |= (cd < 4 ? 3u - cd : 4u) << ((j & 7) << 2); }
		w[b] = v;
	}
}

// ---- SAM text ---------------------------------------------------------------------------------------------------------------
struct SamIn {
	const AlReg *out; const uint64_t *out_off; const uint32_t *arena;
	const AlRdText *rtxt; const uint8_t *rd_info; const uint32_t *rd_frag, *rd_len; const int32_t *frag_rep;
	const char *t0, *t1; int two_files; uint32_t n_reads;
	AlSamCfg C;
};
__device__ __forceinline__ AlSamRead d_sam_read(const SamIn &I, uint32_t i)
{
	AlSamRead r; const uint64_t o0 = I.out_off[i], o1 = I.out_off[i + 1];
	r.regs = I.out + o0; r.n_regs = (int)(o1 - o0); r.arena = I.arena; r.qlen = (int)I.rd_len[i]; r.flip = (I.rd_info[i] & AL_RI_FLIP) ? 1 : 0;
	const AlRdText t = I.rtxt[i]; r.name = t.name; r.name_len = t.name_len; r.seq = t.seq; r.qual = t.qual;
	return r;
}
__global__ void __launch_bounds__(256)
k_sam_len(SamIn I, uint32_t *__restrict__ sam_len, uint32_t *__restrict__ sam_nrec)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i > I.n_reads) return;
	if (i == I.n_reads) { sam_len[i] = 0; sam_nrec[i] = 0; return; }
	const uint32_t info = I.rd_info[i];
	const AlSamRead me = d_sam_read(I, i);
	AlSamRead mt; const bool paired = (info & AL_RI_PAIRED) != 0;
	if (paired) mt = d_sam_read(I, (info & AL_RI_SEG1) ? i - 1 : i + 1);
	AlSamCountSink o; o.C = &I.C; o.text = (I.two_files && (info & AL_RI_SEG1)) ? I.t1 : I.t0;
	const int n = al_sam_read_records(o, I.C, me, paired ? &mt : nullptr, (info & AL_RI_SEG1) ? 1 : 0, paired ? 2 : 1, I.frag_rep[I.rd_frag[i]]);
	sam_len[i] = (uint32_t)o.n; sam_nrec[i] = (uint32_t)n;
}
struct SamWriteSink {
	const AlSamCfg *C; const char *text; char *p; char *base; AlBulk *bulk, *next; uint32_t file; int slot;
	__device__ __forceinline__ char peek(uint32_t off) const { return text[off]; }
	__device__ __forceinline__ void begin_record() { bulk = next; next += 2; slot = 0; }   // two descriptor slots per record
	__device__ __forceinline__ void ch(char c) { *p++ = c; }
	__device__ __forceinline__ void lit(const char *s) { while (*s) *p++ = *s++; }
	__device__ __forceinline__ void num(long long v)
	{
		unsigned long long x = v < 0 ? 0ULL - (unsigned long long)v : (unsigned long long)v;
		const int l = al_num_len(v); char *e = p + l;
		do { *--e = (char)('0' + (int)(x % 10)); x /= 10; } while (x);
		if (v < 0) *--e = '-';
		p += l;
	}
	__device__ __forceinline__ void txt(uint32_t off, uint32_t len) { for (uint32_t i = 0; i < len; ++i) p[i] = text[off + i]; p += len; }
	__device__ __forceinline__ void mem(const char *s, int len) { for (int i = 0; i < len; ++i) p[i] = s[i]; p += len; }
	__device__ __forceinline__ void cname(int rid) { const uint32_t a = C->name_off[rid], b = C->name_off[rid + 1]; for (uint32_t i = a; i < b; ++i) *p++ = C->names[i]; }
	__device__ __forceinline__ void seqfld(uint32_t off, int len, int rev, int comp, int is_seq)
	{   // SEQ prints U as T (kseq2bseq stores it so, bseq.c:72-74); QUAL bytes are copied as they are
		if (len <= 0) return;
		bulk[slot] = AlBulk{(uint64_t)(p - base), off, (uint32_t)len | file << 27 | (is_seq ? 1u << 28 : 0u) | (comp ? 1u << 29 : 0u) | (rev ? 1u << 30 : 0u)};
		++slot; p += len;
	}
};
__global__ void __launch_bounds__(256)
k_sam_write(SamIn I, const uint64_t *__restrict__ sam_off, const uint64_t *__restrict__ rec_off, char *__restrict__ sam, AlBulk *__restrict__ bulk)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= I.n_reads) return;
	const uint32_t info = I.rd_info[i];
	const AlSamRead me = d_sam_read(I, i);
	AlSamRead mt; const bool paired = (info & AL_RI_PAIRED) != 0;
	if (paired) mt = d_sam_read(I, (info & AL_RI_SEG1) ? i - 1 : i + 1);
	const uint32_t file = (I.two_files && (info & AL_RI_SEG1)) ? 1u : 0u;
	SamWriteSink o; o.C = &I.C; o.text = file ? I.t1 : I.t0; o.base = sam; o.p = sam + sam_off[i]; o.file = file;
	o.next = bulk + 2 * rec_off[i]; o.bulk = o.next; o.slot = 0;
	al_sam_read_records(o, I.C, me, paired ? &mt : nullptr, (info & AL_RI_SEG1) ? 1 : 0, paired ? 2 : 1, I.frag_rep[I.rd_frag[i]]);
}
// a SEQ / QUAL field: a wavefront per descriptor, one byte per lane and step (the destination is contiguous)
__global__ void __launch_bounds__(256)
k_sam_bulk(const AlBulk *__restrict__ bulk, uint64_t n_desc, const char *__restrict__ t0, const char *__restrict__ t1, const uint8_t *__restrict__ tabs, char *__restrict__ sam)
{
	const uint64_t d = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); const uint32_t l = threadIdx.x & 63;
	if (d >= n_desc) return;
	const AlBulk b = bulk[d];
	const uint32_t len = b.len_flags & 0x7ffffffu; if (len == 0) return;
	const char *s = ((b.len_flags >> 27 & 1u) ? t1 : t0) + b.src;
	const bool u2t = b.len_flags >> 28 & 1u, comp = b.len_flags >> 29 & 1u, rev = b.len_flags >> 30 & 1u;
	const uint8_t *ct = tabs + 256;
	char *o = sam + b.dst;
	for (uint32_t j = l; j < len; j += 64) {
		uint8_t c = (uint8_t)s[rev ? len - 1 - j : j];
		if (u2t && (c == 'u' || c == 'U')) --c;
		if (comp && c < 128) c = ct[c];
		o[j] = (char)c;
	}
}

// ---- host side ----------------------------------------------------------------------------------------------------------------
struct CastU64s { __host__ __device__ uint64_t operator()(const uint32_t &v) const { return (uint64_t)v; } };
static int scan_excl_u64(AlStreamSlot &S, const uint32_t *in, uint64_t *out, size_t n, hipStream_t s)
{   // out[0 .. n) = exclusive prefix sums of in[0 .. n)
	auto it = rocprim::make_transform_iterator(in, CastU64s());
	size_t bytes = 0;
	AL_HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, it, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), s));
	if (S.scan_tmp.ensure(bytes + 16)) return -1;
	AL_HIP_CHECK(rocprim::exclusive_scan(S.scan_tmp.p, bytes, it, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), s));
	return 0;
}

void AlStreamSlot::release()
{
	for (int i = 0; i < 2; ++i) { txt[i].release(); ls[i].release(); frec[i].release(); }
	tile_cnt.release(); tile_off.release(); rtxt.release(); rd_info.release(); rd_frag.release(); rd_words.release(); rd_mcnt.release();
	se_key.release(); se_run.release(); se_fs.release(); se_fidx.release(); st.release(); tabs.release(); scan_tmp.release();
	names.release(); name_off.release(); rg.release(); sam_len.release(); sam_nrec.release(); sam_off.release(); rec_off.release(); bulk.release(); sam.release();
}

int al_stream_slot_init(AlStreamSlot &S, const al_idx_t *mi, int device, int n_files)
{
	S.n_files = n_files; S.device = device; S.mi = mi;
	AL_HIP_CHECK(hipSetDevice(device));
	AL_HIP_CHECK(hipStreamCreateWithFlags(&S.io, hipStreamNonBlocking));
	AL_HIP_CHECK(hipEventCreateWithFlags(&S.ev, hipEventDisableTiming));
	for (int i = 0; i < 2; ++i) AL_HIP_CHECK(hipEventCreateWithFlags(&S.ev_out[i], hipEventDisableTiming));
	if (S.tabs.ensure(512) || S.st.ensure(16)) return -1;
	uint8_t h[512]; memcpy(h, al_nt4(), 256); memcpy(h + 256, al_comp(), 256);
	AL_HIP_CHECK(hipMemcpy(S.tabs.p, h, 512, hipMemcpyHostToDevice));
	std::string names; std::vector<uint32_t> off(mi->seq.size() + 1);
	for (size_t i = 0; i < mi->seq.size(); ++i) { off[i] = (uint32_t)names.size(); names += mi->seq[i].name; }
	off[mi->seq.size()] = (uint32_t)names.size();
	if (S.names.ensure(names.size() + 1) || S.name_off.ensure(off.size())) return -1;
	if (!names.empty()) AL_HIP_CHECK(hipMemcpy(S.names.p, names.data(), names.size(), hipMemcpyHostToDevice));
	AL_HIP_CHECK(hipMemcpy(S.name_off.p, off.data(), off.size() * 4, hipMemcpyHostToDevice));
	return 0;
}
void al_stream_slot_destroy(AlStreamSlot &S)
{
	if (!S.io) return;
	(void)hipSetDevice(S.device);
	(void)hipStreamSynchronize(S.io);
	S.release();
	(void)hipEventDestroy(S.ev); (void)hipStreamDestroy(S.io); S.io = nullptr; S.ev = nullptr;
	for (int i = 0; i < 2; ++i) if (S.ev_out[i]) { (void)hipEventDestroy(S.ev_out[i]); S.ev_out[i] = nullptr; }
}

int al_stream_begin_text(AlStreamSlot &S, int i, size_t cap_bytes)
{
	AL_HIP_CHECK(hipSetDevice(S.device));
	if (cap_bytes >= (1ULL << 31)) { fprintf(stderr, "[airlift] a text block of %zu bytes exceeds the 2 GB the device parser indexes\n", cap_bytes); return -1; }
	if (S.txt[i].ensure(cap_bytes + NL_TILE + 16)) return -1;
	S.txt_n[i] = 0;
	return 0;
}
int al_stream_append_text(AlStreamSlot &S, int i, const char *p, size_t n)
{   // host bytes (page-locked or not) to the end of file i's text; returns when the copy is done (the caller reuses the buffer)
	if (n == 0) return 0;
	if (S.txt_n[i] + n + NL_TILE + 16 > S.txt[i].cap) { fprintf(stderr, "[airlift] al_stream_append_text: text buffer too small\n"); return -1; }
	AL_HIP_CHECK(hipMemcpyAsync(S.txt[i].p + S.txt_n[i], p, n, hipMemcpyHostToDevice, S.io));
	AL_HIP_CHECK(hipStreamSynchronize(S.io));
	S.txt_n[i] += n;
	return 0;
}

int al_stream_fetch_text(AlStreamSlot &S, int i, uint64_t from, uint64_t n, char *dst)
{
	if (n == 0) return 0;
	AL_HIP_CHECK(hipSetDevice(S.device));
	AL_HIP_CHECK(hipMemcpyAsync(dst, S.txt[i].p + from, n, hipMemcpyDeviceToHost, S.io));
	AL_HIP_CHECK(hipStreamSynchronize(S.io));
	return 0;
}

int al_stream_parse(AlStreamSlot &S, const bool *eof, int max_reads, AlIngestResult *res)
{
	hipStream_t s = S.io;
	AL_HIP_CHECK(hipSetDevice(S.device));
	memset(res, 0, sizeof(*res));
	const int nf = S.n_files;
	unsigned long long h_st[16];
	AL_HIP_CHECK(hipMemsetAsync(S.st.p, 0, 16 * 8, s));
	AL_HIP_CHECK(hipMemsetAsync(S.st.p + 2, 0xff, 16, s));                     // first bad record of each file: none
	size_t tiles[2] = {0, 0}, tile_base[2] = {0, 0};
	for (int i = 0; i < nf; ++i) { tiles[i] = (S.txt_n[i] + NL_TILE - 1) / NL_TILE; tile_base[i] = i ? tiles[0] + 1 : 0; }
	if (S.tile_cnt.ensure(tiles[0] + tiles[1] + 4) || S.tile_off.ensure(tiles[0] + tiles[1] + 4)) return -1;
	for (int i = 0; i < nf; ++i) {
		if (tiles[i]) hipLaunchKernelGGL(k_nl_count, dim3((unsigned)tiles[i]), dim3(256), 0, s, S.txt[i].p, S.txt_n[i], S.tile_cnt.p + tile_base[i]);
		AL_HIP_CHECK(hipMemsetAsync(S.tile_cnt.p + tile_base[i] + tiles[i], 0, 4, s));
		if (scan_excl_u64(S, S.tile_cnt.p + tile_base[i], S.tile_off.p + tile_base[i], tiles[i] + 1, s)) return -1;
		AL_HIP_CHECK(hipMemcpyAsync(&h_st[i], S.tile_off.p + tile_base[i] + tiles[i], 8, hipMemcpyDeviceToHost, s));
	}
	AL_HIP_CHECK(hipStreamSynchronize(s));
	for (int i = 0; i < nf; ++i) {
		uint64_t lines = h_st[i];
		if (S.ls[i].ensure(lines + 8)) return -1;
		if (tiles[i]) hipLaunchKernelGGL(k_nl_fill, dim3((unsigned)tiles[i]), dim3(256), 0, s, S.txt[i].p, S.txt_n[i], S.tile_off.p + tile_base[i], S.ls[i].p);
		else AL_HIP_CHECK(hipMemsetAsync(S.ls[i].p, 0, 4, s));
		res->lines[i] = lines;
		const uint64_t nrec = lines / 4;
		if (S.frec[i].ensure(nrec + 1)) return -1;
		if (nrec) hipLaunchKernelGGL(k_fq_parse, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, s, S.txt[i].p, S.ls[i].p, (uint32_t)nrec, S.frec[i].p, S.st.p + 2 + i);
	}
	AL_HIP_CHECK(hipMemcpyAsync(h_st, S.st.p, 4 * 8, hipMemcpyDeviceToHost, s));
	AL_HIP_CHECK(hipStreamSynchronize(s));
	for (int i = 0; i < nf; ++i) { res->first_bad[i] = h_st[2 + i]; res->n_rec[i] = std::min<uint64_t>(res->lines[i] / 4, res->first_bad[i]); }
	uint64_t n = res->n_rec[0];
	if (nf == 2) {
		n = std::min(n, res->n_rec[1]); n = std::min<uint64_t>(n, (uint64_t)std::max(1, max_reads / 2));
		res->n_frag = (int)n; res->n_reads = (int)(2 * n);
	} else {
		n = std::min<uint64_t>(n, (uint64_t)std::max(2, max_reads));
		// fragments of the first n records: runs of equal names, cut in twos
		int n_frag = 0;
		if (n > 0) {
			if (S.se_key.ensure(n + 2) || S.se_run.ensure(n + 2) || S.se_fs.ensure(n + 2) || S.se_fidx.ensure(n + 2)) return -1;
			hipLaunchKernelGGL(k_se_same, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, S.txt[0].p, S.frec[0].p, (uint32_t)n, S.se_key.p);
			size_t bytes = 0;
			AL_HIP_CHECK(rocprim::inclusive_scan(nullptr, bytes, S.se_key.p, S.se_run.p, n, rocprim::maximum<uint32_t>(), s));
			if (S.scan_tmp.ensure(bytes + 16)) return -1;
			AL_HIP_CHECK(rocprim::inclusive_scan(S.scan_tmp.p, bytes, S.se_key.p, S.se_run.p, n, rocprim::maximum<uint32_t>(), s));
			hipLaunchKernelGGL(k_se_flag, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, s, S.se_run.p, (uint32_t)n, S.se_fs.p);
			bytes = 0;
			AL_HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, S.se_fs.p, S.se_fidx.p, 0u, n + 1, rocprim::plus<uint32_t>(), s));
			if (S.scan_tmp.ensure(bytes + 16)) return -1;
			AL_HIP_CHECK(rocprim::exclusive_scan(S.scan_tmp.p, bytes, S.se_fs.p, S.se_fidx.p, 0u, n + 1, rocprim::plus<uint32_t>(), s));
			uint32_t last_fs = 0, nfr = 0;
			AL_HIP_CHECK(hipMemcpyAsync(&last_fs, S.se_fs.p + (n - 1), 4, hipMemcpyDeviceToHost, s));
			AL_HIP_CHECK(hipMemcpyAsync(&nfr, S.se_fidx.p + n, 4, hipMemcpyDeviceToHost, s));
			AL_HIP_CHECK(hipStreamSynchronize(s));
			// a trailing single read may pair with the first read of the next batch: it stays in the text unless the input ends here
			const bool more = !(eof[0] && n == res->lines[0] / 4 && res->first_bad[0] == ~0ULL);
			if (last_fs && more) { --n; --nfr; }                                   // (fs[n] stays 1: the record after the last one taken starts a fragment)
			n_frag = (int)nfr;
		}
		res->n_frag = n_frag; res->n_reads = (int)n;
	}
	// bytes consumed = start of the first record not taken
	for (int i = 0; i < nf; ++i) {
		const uint64_t recs = nf == 2 ? (uint64_t)res->n_frag : (uint64_t)res->n_reads;
		uint32_t e = 0;
		AL_HIP_CHECK(hipMemcpyAsync(&e, S.ls[i].p + 4 * recs, 4, hipMemcpyDeviceToHost, s));
		AL_HIP_CHECK(hipStreamSynchronize(s));
		res->consumed[i] = e;
	}
	return 0;
}

int al_stream_setup(AlStreamSlot &S, al_ctx_t *c, uint32_t rec_lo, uint32_t rec_hi, uint32_t frag_lo, uint32_t frag_hi)
{   // records [rec_lo, rec_hi) of the parsed text (two files: record = fragment) = fragments [frag_lo, frag_hi)
	hipStream_t s = c->stream;
	AL_HIP_CHECK(hipSetDevice(c->device));
	const int n_frag = (int)(frag_hi - frag_lo), n_reads = S.n_files == 2 ? 2 * n_frag : (int)(rec_hi - rec_lo), k = c->mi->k;
	c->n_frag = n_frag; c->n_reads = n_reads; c->ran = false; c->dev_batch = true;
	if (S.rtxt.ensure((size_t)n_reads + 1) || S.rd_info.ensure((size_t)n_reads + 1) || S.rd_frag.ensure((size_t)n_reads + 1) || S.rd_words.ensure((size_t)n_reads + 2) || S.rd_mcnt.ensure((size_t)n_reads + 2)) return -1;
	if (c->rd_off.ensure(n_reads + 2) || c->rd_len.ensure(n_reads + 2) || c->frag_first.ensure(n_frag + 2) || c->frag_hash.ensure(n_frag + 2) || c->mini_off.ensure(n_reads + 2) || c->mini_cnt.ensure(n_reads + 1) ||
	    c->frag_nm.ensure(n_frag + 1) || c->frag_na.ensure(n_frag + 1) || c->frag_rep.ensure(n_frag + 1) || c->frag_nu.ensure(n_frag + 1) || c->a_off.ensure(n_frag + 2) ||
	    c->rechain_list.ensure(n_frag + 1) || c->tmp_u32.ensure(n_frag + 2) || c->tmp_u64.ensure(n_frag + 2) || c->counters.ensure(32)) return -1;
	SetupOut O{S.rtxt.p, S.rd_info.p, S.rd_frag.p, c->rd_len.p, S.rd_words.p, S.rd_mcnt.p, c->frag_first.p, c->frag_hash.p, S.st.p};
	AL_HIP_CHECK(hipMemsetAsync(S.st.p + 4, 0, 4 * 8, s));
	if (S.n_files == 2) hipLaunchKernelGGL(k_setup_pe, dim3((unsigned)((n_frag + 256) / 256)), dim3(256), 0, s, S.txt[0].p, S.frec[0].p + rec_lo, S.frec[1].p + rec_lo, (uint32_t)n_frag, O, k, c->opt.seed, c->opt.pe_ori);
	else hipLaunchKernelGGL(k_setup_se, dim3((unsigned)((n_reads + 256) / 256)), dim3(256), 0, s, S.txt[0].p, S.frec[0].p + rec_lo, S.se_fs.p + rec_lo, S.se_fidx.p + rec_lo, (uint32_t)n_reads, (uint32_t)n_frag, frag_lo, O, k, c->opt.seed, c->opt.pe_ori);
	if (scan_excl_u64(S, S.rd_words.p, c->rd_off.p, (size_t)n_reads + 1, s) || scan_excl_u64(S, S.rd_mcnt.p, c->mini_off.p, (size_t)n_reads + 1, s)) return -1;
	unsigned long long h[4]; uint64_t words = 0, mtot = 0;
	AL_HIP_CHECK(hipMemcpyAsync(h, S.st.p + 4, 32, hipMemcpyDeviceToHost, s));
	AL_HIP_CHECK(hipMemcpyAsync(&words, c->rd_off.p + n_reads, 8, hipMemcpyDeviceToHost, s));
	AL_HIP_CHECK(hipMemcpyAsync(&mtot, c->mini_off.p + n_reads, 8, hipMemcpyDeviceToHost, s));
	AL_HIP_CHECK(hipStreamSynchronize(s));
	c->max_rd_len = (int)h[0]; c->max_qlen_sum = (int)h[1]; c->n_bases = h[2]; c->stat_bytes_in = h[3]; c->seq_words = words; c->mini_total = mtot;
	{   // the same limit al_batch_upload enforces
		const int lim = AL_MAX_READ_LEN;
		if (c->max_rd_len >= lim) { fprintf(stderr, "[airlift] a read of %d bases exceeds the limit of the GPU path (%d bases at k = %d)\n", c->max_rd_len, lim - 1, k); return -3; }
	}
	if (c->rd_seq.ensure(words + 1) || c->mini.ensure(mtot + 1) || c->match.ensure(mtot + 1) || c->heap_ws.ensure(mtot + 1)) return -1;
	if (n_reads) hipLaunchKernelGGL(k_pack_reads, dim3((unsigned)((n_reads + 7) / 8)), dim3(256), 0, s, S.txt[0].p, S.n_files == 2 ? S.txt[1].p : S.txt[0].p, S.rtxt.p, S.rd_info.p, c->rd_len.p, c->rd_off.p, (uint32_t)n_reads, S.n_files == 2 ? 1 : 0, S.tabs.p, c->rd_seq.p);
	AL_HIP_CHECK(hipMemsetAsync(c->rd_seq.p + words, 0, 4, s));
	AL_HIP_CHECK(hipGetLastError());
	return 0;
}

int al_stream_sam(AlStreamSlot &S, al_ctx_t *c, const char *rg_id)
{
	hipStream_t s = c->stream;
	AL_HIP_CHECK(hipSetDevice(c->device));
	S.sam_bytes = 0; S.sam_records = 0;
	const uint32_t nr = (uint32_t)c->n_reads;
	if (nr == 0) return 0;
	AlDevResult R;
	if (al_align_result(c, &R)) return -4;
	if (!S.cfg_ready) {
		S.rg_len = rg_id ? (int)strlen(rg_id) : 0;
		if (S.rg.ensure((size_t)S.rg_len + 1)) return -1;
		if (S.rg_len) AL_HIP_CHECK(hipMemcpyAsync(S.rg.p, rg_id, (size_t)S.rg_len, hipMemcpyHostToDevice, s));
		S.cfg_ready = true;
	}
	SamIn I; I.out = R.out; I.out_off = R.out_off; I.arena = R.arena; I.rtxt = S.rtxt.p; I.rd_info = S.rd_info.p; I.rd_frag = S.rd_frag.p; I.rd_len = c->rd_len.p; I.frag_rep = c->frag_rep.p;
	I.t0 = (const char *)S.txt[0].p; I.t1 = (const char *)(S.n_files == 2 ? S.txt[1].p : S.txt[0].p); I.two_files = S.n_files == 2 ? 1 : 0; I.n_reads = nr;
	I.C.names = S.names.p; I.C.name_off = S.name_off.p; I.C.rg_id = S.rg.p; I.C.rg_len = S.rg_len;
	I.C.no_print_2nd = (c->opt.flag & AL_F_NO_PRINT_2ND) ? 1 : 0; I.C.hit_only = (c->opt.flag & AL_F_SAM_HIT_ONLY) ? 1 : 0; I.C.pe_ori = c->opt.pe_ori;
	if (S.sam_len.ensure((size_t)nr + 2) || S.sam_nrec.ensure((size_t)nr + 2) || S.sam_off.ensure((size_t)nr + 2) || S.rec_off.ensure((size_t)nr + 2)) return -1;
	hipLaunchKernelGGL(k_sam_len, dim3((nr + 256) / 256), dim3(256), 0, s, I, S.sam_len.p, S.sam_nrec.p);
	if (scan_excl_u64(S, S.sam_len.p, S.sam_off.p, (size_t)nr + 1, s) || scan_excl_u64(S, S.sam_nrec.p, S.rec_off.p, (size_t)nr + 1, s)) return -1;
	uint64_t tot[2] = {0, 0};
	AL_HIP_CHECK(hipMemcpyAsync(&tot[0], S.sam_off.p + nr, 8, hipMemcpyDeviceToHost, s));
	AL_HIP_CHECK(hipMemcpyAsync(&tot[1], S.rec_off.p + nr, 8, hipMemcpyDeviceToHost, s));
	AL_HIP_CHECK(hipStreamSynchronize(s));
	if (S.sam.ensure(tot[0] + 64) || S.bulk.ensure(2 * tot[1] + 2)) return -1;
	if (tot[1]) {
		AL_HIP_CHECK(hipMemsetAsync(S.bulk.p, 0, 2 * tot[1] * sizeof(AlBulk), s));
		hipLaunchKernelGGL(k_sam_write, dim3((nr + 255) / 256), dim3(256), 0, s, I, S.sam_off.p, S.rec_off.p, S.sam.p, S.bulk.p);
		hipLaunchKernelGGL(k_sam_bulk, dim3((unsigned)((2 * tot[1] + 3) / 4)), dim3(256), 0, s, S.bulk.p, 2 * tot[1], I.t0, I.t1, S.tabs.p, S.sam.p);
		// the text leaves on the slot's own stream (al_stream_sam_fetch): the context goes on with its next batch
		AL_HIP_CHECK(hipEventRecord(S.ev, s));
		AL_HIP_CHECK(hipStreamWaitEvent(S.io, S.ev, 0));
	}
	AL_HIP_CHECK(hipGetLastError());
	S.sam_bytes = tot[0]; S.sam_records = tot[1];
	return 0;
}
int al_stream_sam_fetch(AlStreamSlot &S, uint64_t off, uint64_t n, char *dst, hipEvent_t done)
{
	AL_HIP_CHECK(hipSetDevice(S.device));
	if (n) AL_HIP_CHECK(hipMemcpyAsync(dst, S.sam.p + off, n, hipMemcpyDeviceToHost, S.io));
	AL_HIP_CHECK(hipEventRecord(done, S.io));
	return 0;
}

int al_stream_frag_starts(AlStreamSlot &S, const AlIngestResult &res, std::vector<uint32_t> &first)
{   // single-file input: record index of every fragment start of the parsed batch (rare path: a batch that has to be cut)
	AL_HIP_CHECK(hipSetDevice(S.device));
	const size_t n = (size_t)res.n_reads;
	std::vector<uint32_t> fs(n + 1);
	if (n) AL_HIP_CHECK(hipMemcpy(fs.data(), S.se_fs.p, n * 4, hipMemcpyDeviceToHost));
	first.clear();
	for (size_t r = 0; r < n; ++r) if (fs[r]) first.push_back((uint32_t)r);
	first.push_back((uint32_t)n);
	return (int)first.size() - 1 == res.n_frag ? 0 : -1;
}
