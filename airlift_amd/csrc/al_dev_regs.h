// al_dev_regs.h -- device functions for the per-fragment region bookkeeping (chains -> hits -> primary/secondary
// -> MAPQ -> pairing).  Scalar integer/float code executed by one lane (KA) or redundantly by the 16 lanes of an
// extension group (K5).  Each function names the reference routine whose result it must reproduce.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "al_internal.h"
#include "al_device.h"

#define AL_D __device__ __forceinline__
#include "al_dev_sort.h"

struct AnchorAcc {
	typedef AlAnchor E; AlAnchor *a;
	AL_D uint64_t key(int i) const { return a[i].x; }
	AL_D uint64_t keyof(const AlAnchor &e) const { return e.x; }
	AL_D AlAnchor get(int i) const { return a[i]; }
	AL_D void set(int i, const AlAnchor &e) { a[i] = e; }
};
AL_D void d_isort128(AlAnchor *a, int n)
{
	for (int i = 1; i < n; ++i) if (a[i].x < a[i - 1].x) { AlAnchor t = a[i]; int j = i; for (; j > 0 && t.x < a[j - 1].x; --j) a[j] = a[j - 1]; a[j] = t; }
}
AL_D bool d_sort128(AlAnchor *a, int n, void *scratch)
{   // radix_sort_128x.  scratch: AL_RS_SCRATCH bytes, used only when n > 64.  Returns true if the order is not reproduced.
	if (n <= 64) { d_isort128(a, n); return false; }
	AnchorAcc acc{a};
	return d_rs_sort(acc, n, (uint16_t *)scratch);
}
AL_D void d_sort64(uint64_t *a, int n)
{   // plain values: ties are indistinguishable
	if (n <= 64) { for (int i = 1; i < n; ++i) if (a[i] < a[i - 1]) { uint64_t t = a[i]; int j = i; for (; j > 0 && t < a[j - 1]; --j) a[j] = a[j - 1]; a[j] = t; } return; }
	for (int s = (n >> 1) - 1; s >= 0; --s) { int i = s; uint64_t t = a[i]; for (;;) { int c = 2 * i + 1; if (c >= n) break; if (c + 1 < n && a[c + 1] > a[c]) ++c; if (a[c] <= t) break; a[i] = a[c]; i = c; } a[i] = t; }
	for (int e = n - 1; e > 0; --e) { uint64_t t = a[e]; a[e] = a[0]; int i = 0; for (;;) { int c = 2 * i + 1; if (c >= e) break; if (c + 1 < e && a[c + 1] > a[c]) ++c; if (a[c] <= t) break; a[i] = a[c]; i = c; } a[i] = t; }
}

AL_D uint64_t d_hash64(uint64_t key)
{   // hit.c:43-52
	key = (~key + (key << 21)); key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8)); key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4)); key = key ^ key >> 28;
	key = (key + (key << 31));
	return key;
}

AL_D void d_reg_set_coor(AlReg *r, int32_t qlen, const AlAnchor *a)
{   // mm_reg_set_coor + mm_cal_fuzzy_len, hit.c:8-41.  (round 6) Everything is computed in locals and stored once: through `r` every `r->blen += ...` of the fuzzy-length
	// loop was a load, an add and a store that the next iteration waited for (the compiler cannot know that `r` and `a` do not overlap) -- 6600 cycles per hit in k_regs_heavy.
	const int32_t k = r->as, cnt = r->cnt;
	const AlAnchor a0 = a[k], a1 = a[k + cnt - 1];
	const int32_t q_span = (int32_t)(a0.y >> 32 & 0xff);
	const int rev = (int)(a0.x >> 63);
	r->flags = (r->flags & ~ALR_REV) | (rev ? ALR_REV : 0);
	r->rid = (int32_t)(a0.x << 1 >> 33);
	r->rs = (int32_t)a0.x + 1 > q_span ? (int32_t)a0.x + 1 - q_span : 0;
	r->re = (int32_t)a1.x + 1;
	if (!rev) { r->qs = (int32_t)a0.y + 1 - q_span; r->qe = (int32_t)a1.y + 1; }
	else { r->qs = qlen - ((int32_t)a1.y + 1); r->qe = qlen - ((int32_t)a0.y + 1 - q_span); }
	int32_t mlen = 0, blen = 0;
	if (cnt > 0) {
		mlen = blen = q_span;
		AlAnchor p = a0;
		for (int i = k + 1; i < k + cnt; ++i) {
			const AlAnchor c = a[i];
			const int span = (int)(c.y >> 32 & 0xff);
			const int tl = (int32_t)c.x - (int32_t)p.x, ql = (int32_t)c.y - (int32_t)p.y;
			blen += tl > ql ? tl : ql;
			mlen += tl > span && ql > span ? span : tl < ql ? tl : ql;
			p = c;
		}
	}
	r->mlen = mlen; r->blen = blen;
}

AL_D void d_reg_clear(AlReg *r) { int32_t *p = (int32_t *)r; for (int i = 0; i < (int)(sizeof(AlReg) / 4); ++i) p[i] = 0; }

// mm_gen_regs, hit.c:52-88.  z: scratch of n_u AlAnchor, followed by AL_RS_SCRATCH bytes when n_u > 64 (the work areas
// hold 4*n_u+4 entries).  Returns true if the sort order could not be reproduced.
// uo: offset of every chain's first anchor in a[] (nullptr: the chains' anchors follow one another)
AL_D bool d_gen_regs(uint32_t hash, int qlen, int n_u, const uint64_t *u, const AlAnchor *a, AlReg *r, AlAnchor *z, const uint32_t *uo = nullptr)
{
	if (n_u == 0) return false;
	int k = 0;
	for (int i = 0; i < n_u; ++i) {
		if (uo) k = (int)uo[i];
		const uint32_t h = (uint32_t)d_hash64((d_hash64(a[k].x) + d_hash64(a[k].y)) ^ hash);
		z[i].x = u[i] ^ h;
		z[i].y = (uint64_t)k << 32 | (uint32_t)(int32_t)u[i];
		k += (int32_t)u[i];
	}
	const bool tie = d_sort128(z, n_u, z + n_u);
	for (int i = 0; i < n_u >> 1; ++i) { AlAnchor t = z[i]; z[i] = z[n_u - 1 - i]; z[n_u - 1 - i] = t; }
	for (int i = 0; i < n_u; ++i) {   // (the record is made in registers and stored once)
		const AlAnchor zi = z[i];
		AlReg R;
		d_reg_clear(&R);
		R.id = i; R.parent = AL_PARENT_UNSET;
		R.score = R.score0 = (int32_t)(zi.x >> 32);
		R.hash = (uint32_t)zi.x;
		R.cnt = (int32_t)zi.y; R.as = (int32_t)(zi.y >> 32);
		d_reg_set_coor(&R, qlen, a);
		r[i] = R;
	}
	return tie;
}

AL_D void d_split_reg(AlReg *r, AlReg *r2, int n, int qlen, const AlAnchor *a)
{   // mm_split_reg, hit.c:90-107
	if (n <= 0 || n >= r->cnt) return;
	*r2 = *r;
	r2->id = -1;
	r2->flags &= ~(ALR_SAM_PRI | ALR_SPLIT_INV | ALR_HAS_P);
	r2->n_cigar = 0; r2->dp_score = r2->dp_max = r2->dp_max2 = 0; r2->n_ambi = 0;
	r2->cnt = r->cnt - n;
	r2->score = (int32_t)((double)__fmul_rn((float)r->score, al_fdiv((float)r2->cnt, (float)r->cnt)) + .499);
	r2->as = r->as + n;
	if (r->parent == r->id) r2->parent = AL_PARENT_TMP_PRI;
	d_reg_set_coor(r2, qlen, a);
	r->cnt -= r2->cnt; r->score -= r2->score;
	d_reg_set_coor(r, qlen, a);
	r->flags |= 1u; r2->flags |= 2u;
}

// mm_set_parent, hit.c:109-167 (hard_mask_level = 0).  cov: n u64 scratch, w: n int scratch.
AL_D void d_set_parent(float mask_level, int n, AlReg *r, int sub_diff, uint64_t *cov, int *w)
{
	if (n <= 0) return;
	for (int i = 0; i < n; ++i) r[i].id = i;
	w[0] = 0; r[0].parent = 0;
	int k = 1;
	for (int i = 1; i < n; ++i) {
		AlReg *ri = &r[i];
		const int si = ri->qs, ei = ri->qe; int n_cov = 0, uncov_len = 0, j;
		for (j = 0; j < k; ++j) {
			const AlReg *rp = &r[w[j]]; int sj = rp->qs, ej = rp->qe;
			if (ej <= si || sj >= ei) continue;
			if (sj < si) sj = si;
			if (ej > ei) ej = ei;
			cov[n_cov++] = (uint64_t)(uint32_t)sj << 32 | (uint32_t)ej;
		}
		j = k;   // "goto set_parent_test" with no overlap leaves j == k
		if (n_cov > 0) {
			int x = si;
			d_sort64(cov, n_cov);
			for (int jj = 0; jj < n_cov; ++jj) {
				if ((int)(cov[jj] >> 32) > x) uncov_len += (int)(cov[jj] >> 32) - x;
				x = (int32_t)cov[jj] > x ? (int32_t)cov[jj] : x;
			}
			if (ei > x) uncov_len += ei - x;
			for (j = 0; j < k; ++j) {
				AlReg *rp = &r[w[j]]; const int sj = rp->qs, ej = rp->qe;
				if (ej <= si || sj >= ei) continue;
				const int mn = ej - sj < ei - si ? ej - sj : ei - si, mx = ej - sj > ei - si ? ej - sj : ei - si;
				const int ol = si < sj ? (ei < sj ? 0 : ei < ej ? ei - sj : ej - sj) : (ej < si ? 0 : ej < ei ? ej - si : ei - si);
				if (__fsub_rn(al_fdiv((float)ol, (float)mn), al_fdiv((float)uncov_len, (float)mx)) > mask_level) {
					int cnt_sub = 0;
					ri->parent = rp->parent;
					rp->subsc = rp->subsc > ri->score ? rp->subsc : ri->score;
					if (ri->cnt >= rp->cnt) cnt_sub = 1;
					if ((rp->flags & ALR_HAS_P) && (ri->flags & ALR_HAS_P) && (rp->rid != ri->rid || rp->rs != ri->rs || rp->re != ri->re || ol != mn)) {
						rp->dp_max2 = rp->dp_max2 > ri->dp_max ? rp->dp_max2 : ri->dp_max;
						if (rp->dp_max - ri->dp_max <= sub_diff) cnt_sub = 1;
					}
					if (cnt_sub) ++rp->n_sub;
					break;
				}
			}
		}
		if (j == k) { w[k++] = i; ri->parent = i; ri->n_sub = 0; }
	}
}
// The same with the primaries' query intervals kept beside their indices (w[n + j] = qs << 16 | qe of primary j; reads are at most 32 768 bases): the two loops over the
// primaries read one word per primary instead of following w[j] to the record's qs and qe (three dependent LDS reads).  k_regs_heavy (w: 2 n ints).
AL_D void d_set_parent_pq(float mask_level, int n, AlReg *r, int sub_diff, uint64_t *cov, int *w)
{
	if (n <= 0) return;
	for (int i = 0; i < n; ++i) r[i].id = i;
	w[0] = 0; r[0].parent = 0; w[n] = (int)((uint32_t)r[0].qs << 16 | (uint32_t)r[0].qe);
	int k = 1;
	for (int i = 1; i < n; ++i) {
		AlReg *ri = &r[i];
		const int si = ri->qs, ei = ri->qe; int n_cov = 0, uncov_len = 0, j;
		for (j = 0; j < k; ++j) {
			const uint32_t q = (uint32_t)w[n + j]; int sj = (int)(q >> 16), ej = (int)(q & 0xffffu);
			if (ej <= si || sj >= ei) continue;
			if (sj < si) sj = si;
			if (ej > ei) ej = ei;
			cov[n_cov++] = (uint64_t)(uint32_t)sj << 32 | (uint32_t)ej;
		}
		j = k;   // "goto set_parent_test" with no overlap leaves j == k
		if (n_cov > 0) {
			int x = si;
			d_sort64(cov, n_cov);
			for (int jj = 0; jj < n_cov; ++jj) {
				if ((int)(cov[jj] >> 32) > x) uncov_len += (int)(cov[jj] >> 32) - x;
				x = (int32_t)cov[jj] > x ? (int32_t)cov[jj] : x;
			}
			if (ei > x) uncov_len += ei - x;
			for (j = 0; j < k; ++j) {
				const uint32_t q = (uint32_t)w[n + j]; const int sj = (int)(q >> 16), ej = (int)(q & 0xffffu);
				if (ej <= si || sj >= ei) continue;
				const int mn = ej - sj < ei - si ? ej - sj : ei - si, mx = ej - sj > ei - si ? ej - sj : ei - si;
				const int ol = si < sj ? (ei < sj ? 0 : ei < ej ? ei - sj : ej - sj) : (ej < si ? 0 : ej < ei ? ej - si : ei - si);
				if (__fsub_rn(al_fdiv((float)ol, (float)mn), al_fdiv((float)uncov_len, (float)mx)) > mask_level) {
					AlReg *rp = &r[w[j]];
					int cnt_sub = 0;
					ri->parent = rp->parent;
					rp->subsc = rp->subsc > ri->score ? rp->subsc : ri->score;
					if (ri->cnt >= rp->cnt) cnt_sub = 1;
					if ((rp->flags & ALR_HAS_P) && (ri->flags & ALR_HAS_P) && (rp->rid != ri->rid || rp->rs != ri->rs || rp->re != ri->re || ol != mn)) {
						rp->dp_max2 = rp->dp_max2 > ri->dp_max ? rp->dp_max2 : ri->dp_max;
						if (rp->dp_max - ri->dp_max <= sub_diff) cnt_sub = 1;
					}
					if (cnt_sub) ++rp->n_sub;
					break;
				}
			}
		}
		if (j == k) { w[n + k] = (int)((uint32_t)si << 16 | (uint32_t)ei); w[k++] = i; ri->parent = i; ri->n_sub = 0; }
	}
}

AL_D int d_set_sam_pri(int n, AlReg *r)
{   // hit.c:203-212
	int n_pri = 0;
	for (int i = 0; i < n; ++i) {
		if (r[i].id == r[i].parent) { ++n_pri; r[i].flags = (r[i].flags & ~ALR_SAM_PRI) | (n_pri == 1 ? ALR_SAM_PRI : 0); }
		else r[i].flags &= ~ALR_SAM_PRI;
	}
	return n_pri;
}

AL_D void d_sync_regs(int n_regs, AlReg *regs, int *tmp)
{   // hit.c:214-236; tmp: (max id + 1) ints
	if (n_regs <= 0) return;
	int max_id = -1;
	for (int i = 0; i < n_regs; ++i) max_id = max_id > regs[i].id ? max_id : regs[i].id;
	const int n_tmp = max_id + 1;
	for (int i = 0; i < n_tmp; ++i) tmp[i] = -1;
	for (int i = 0; i < n_regs; ++i) if (regs[i].id >= 0) tmp[regs[i].id] = i;
	for (int i = 0; i < n_regs; ++i) {
		AlReg *r = &regs[i];
		r->id = i;
		if (r->parent == AL_PARENT_TMP_PRI) r->parent = i;
		else if (r->parent >= 0 && tmp[r->parent] >= 0) r->parent = tmp[r->parent];
		else r->parent = AL_PARENT_UNSET;
	}
	d_set_sam_pri(n_regs, regs);
}

AL_D void d_select_sub(float pri_ratio, int min_diff, int best_n, int *n_, AlReg *r, int *tmp)
{   // hit.c:238-255
	if (pri_ratio > 0.0f && *n_ > 0) {
		const int n = *n_; int k = 0, n_2nd = 0;
		for (int i = 0; i < n; ++i) {
			const int p = r[i].parent;
			if (p == i || (r[i].flags & ALR_INV)) r[k++] = r[i];
			else if (((float)r[i].score >= __fmul_rn((float)r[p].score, pri_ratio) || r[i].score + min_diff >= r[p].score) && n_2nd < best_n) {
				if (!(r[i].qs == r[p].qs && r[i].qe == r[p].qe && r[i].rid == r[p].rid && r[i].rs == r[p].rs && r[i].re == r[p].re)) { r[k++] = r[i]; ++n_2nd; }
			}
		}
		if (k != n) d_sync_regs(k, r, tmp);
		*n_ = k;
	}
}

AL_D void d_select_sub_multi(float pri_ratio, float pri1, float pri2, int max_gap_ref, int min_diff, int best_n, int n_segs, const int *qlens, int *n_, AlReg *r, int *tmp)
{   // pe.c:6-43
	if (pri_ratio > 0.0f && *n_ > 0) {
		const int n = *n_; int k = 0, n_2nd = 0;
		const int max_dist = n_segs == 2 ? qlens[0] + qlens[1] + max_gap_ref : 0;
		for (int i = 0; i < n; ++i) {
			int to_keep = 0;
			if (r[i].parent == i) to_keep = 1;
			else if (r[i].score + min_diff >= r[r[i].parent].score) to_keep = 1;
			else {
				const AlReg *p = &r[r[i].parent], *q = &r[i];
				if ((p->flags & ALR_REV) == (q->flags & ALR_REV) && p->rid == q->rid && q->re - p->rs < max_dist && p->re - q->rs < max_dist) {
					if ((float)q->score >= __fmul_rn((float)p->score, pri1)) to_keep = 1;
				} else {
					const int is_par_both = (n_segs == 2 && p->qs < qlens[0] && p->qe > qlens[0]);
					const int is_chi_both = (n_segs == 2 && q->qs < qlens[0] && q->qe > qlens[0]);
					if (is_chi_both || is_chi_both == is_par_both) { if ((float)q->score >= __fmul_rn((float)p->score, pri_ratio)) to_keep = 1; }
					else { if ((float)q->score >= __fmul_rn((float)p->score, pri2)) to_keep = 1; }
				}
			}
			if (to_keep && r[i].parent != i) { if (n_2nd++ >= best_n) to_keep = 0; }
			if (to_keep) r[k++] = r[i];
		}
		if (k != n) d_sync_regs(k, r, tmp);
		*n_ = k;
	}
}

AL_D void d_filter_regs(const AlParams &P, int qlen, int *n_regs, AlReg *regs)
{   // hit.c:257-276
	int k = 0;
	for (int i = 0; i < *n_regs; ++i) {
		AlReg *r = &regs[i]; int flt = 0;
		if (!(r->flags & ALR_INV) && !(r->flags & ALR_SEG_SPLIT) && r->cnt < P.min_cnt) flt = 1;
		if (r->flags & ALR_HAS_P) {
			if (r->mlen < P.min_chain_score) flt = 1;
			else if (r->dp_max < P.min_dp_max) flt = 1;
			else if ((float)r->qs > __fmul_rn((float)qlen, P.max_clip_ratio) && (float)(qlen - r->qe) > __fmul_rn((float)qlen, P.max_clip_ratio)) flt = 1;
		}
		if (!flt) { if (k < i) regs[k++] = regs[i]; else ++k; }
	}
	*n_regs = k;
}

// mm_hit_sort, hit.c:169-201.  aux: n AlAnchor, t: n AlReg scratch (doubles as the radix sort's work area before it
// receives the permuted hits).  Returns true if the sort order could not be reproduced.
AL_D bool d_hit_sort(int *n_regs, AlReg *r, AlAnchor *aux, AlReg *t)
{
	const int n = *n_regs; int n_aux = 0;
	if (n <= 1) return false;
	for (int i = 0; i < n; ++i) {
		if ((r[i].flags & ALR_INV) || r[i].cnt > 0) {
			if (r[i].flags & ALR_HAS_P) aux[n_aux].x = (uint64_t)(uint32_t)r[i].dp_max << 32 | r[i].hash;
			else aux[n_aux].x = (uint64_t)(uint32_t)r[i].score << 32 | r[i].hash;
			aux[n_aux++].y = (uint64_t)i;
		}
	}
	const bool tie = d_sort128(aux, n_aux, t);
	for (int i = n_aux - 1; i >= 0; --i) t[n_aux - 1 - i] = r[aux[i].y];
	for (int i = 0; i < n_aux; ++i) r[i] = t[i];
	*n_regs = n_aux;
	return tie;
}

AL_D int d_squeeze_a(int n_regs, AlReg *regs, AlAnchor *a, uint64_t *aux)
{   // hit.c:278-296
	int as = 0;
	for (int i = 0; i < n_regs; ++i) aux[i] = (uint64_t)(uint32_t)regs[i].as << 32 | (uint32_t)i;
	d_sort64(aux, n_regs);
	for (int i = 0; i < n_regs; ++i) {
		AlReg *r = &regs[(int32_t)aux[i]];
		const int ras = r->as, rcnt = r->cnt;                                   // (locals: `a` and `r` may overlap for all the compiler knows)
		if (ras != as) { for (int j = 0; j < rcnt; ++j) a[as + j] = a[ras + j]; r->as = as; }   // memmove to lower addresses
		as += rcnt;
	}
	return as;
}

// mm_set_mapq, hit.c:446-491 (is_sr = 1; no inversions on this path).  float32 + logf as the reference.
AL_D void d_set_mapq(int n_regs, AlReg *regs, int min_chain_sc, int match_sc, int rep_len, const AlLogTab &lt)
{
	const float q_coef = 40.0f; long long sum_sc = 0;
	if (n_regs == 0) return;
	for (int i = 0; i < n_regs; ++i) if (regs[i].parent == regs[i].id) sum_sc += regs[i].score;
	const float uniq_ratio = (float)((double)(float)sum_sc / (double)(float)(sum_sc + rep_len));
	for (int i = 0; i < n_regs; ++i) {
		AlReg *r = &regs[i];
		if (r->flags & ALR_INV) r->mapq = 0;
		else if (r->parent == r->id) {
			int mapq;
			const float pen_s1 = __fmul_rn((r->score > 100 ? 1.0f : __fmul_rn(0.01f, (float)r->score)), uniq_ratio);
			float pen_cm = r->cnt > 10 ? 1.0f : __fmul_rn(0.1f, (float)r->cnt);
			pen_cm = pen_s1 < pen_cm ? pen_s1 : pen_cm;
			const int subsc = r->subsc > min_chain_sc ? r->subsc : min_chain_sc;
			const bool has_p = (r->flags & ALR_HAS_P) != 0;
			if (has_p && r->dp_max2 > 0 && r->dp_max > 0) {
				const float identity = al_fdiv((float)r->mlen, (float)r->blen);
				const float x = al_fdiv(al_fdiv(__fmul_rn((float)r->dp_max2, (float)subsc), (float)r->dp_max), (float)r->score0);
				const float lg = al_logf_q(lt, r->dp_max);
				mapq = (int)__fmul_rn(__fmul_rn(__fmul_rn(__fmul_rn(identity, pen_cm), q_coef), __fsub_rn(1.0f, __fmul_rn(x, x))), lg);
			} else {
				const float x = al_fdiv((float)subsc, (float)r->score0);
				if (has_p) {
					const float identity = al_fdiv((float)r->mlen, (float)r->blen);
					const float lg = al_logf_q(lt, r->dp_max);
					mapq = (int)__fmul_rn(__fmul_rn(__fmul_rn(__fmul_rn(identity, pen_cm), q_coef), __fsub_rn(1.0f, x)), lg);
				} else mapq = (int)__fmul_rn(__fmul_rn(__fmul_rn(pen_cm, q_coef), __fsub_rn(1.0f, x)), al_logf_i(lt, r->score));
			}
			mapq -= (int)__fadd_rn(__fmul_rn(4.343f, al_logf_i(lt, r->n_sub + 1)), .499f);
			mapq = mapq > 0 ? mapq : 0;
			r->mapq = mapq < 60 ? mapq : 60;
			if (has_p && r->dp_max > r->dp_max2 && r->mapq == 0) r->mapq = 1;
		} else r->mapq = 0;
	}
}
