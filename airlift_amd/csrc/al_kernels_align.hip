// al_kernels_align.hip -- CDNA4 (gfx950) kernels of the second half of the re-alignment path:
//   KA k_regs   chains -> hits, primary/secondary selection, per-mate split   (mm_gen_regs hit.c:52, chain_post map.c:249,
//               one lane per fragment                                           mm_seg_gen hit.c:356, mm_set_parent hit.c:109)
//   K5 k_align  base-level extension + rescoring + MAPQ + pairing               (mm_align_skeleton align.c:857, mm_align1 align.c:565,
//               one 16-lane group per fragment, 4 groups per wavefront           ksw_extd2_sse ksw2_extd2_sse.c:26, mm_set_mapq hit.c:446,
//                                                                                 mm_pair pe.c:76)
// The SW band state (u,v,x,y,x2,y2,s: the reference's seven int8 vectors) and the exact-score row H live in LDS; a
// 16-lane group is one SSE vector of the reference, so the anti-diagonal sweep reproduces its 16-cell block geometry,
// wrapping int8 arithmetic and tie rules bit for bit.  Traceback bytes go to a per-group HBM scratch.
#include <hip/hip_runtime.h>
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <type_traits>
#include "al_internal.h"
#include "al_device.h"
#include "al_runtime.h"
#include "al_dev_regs.h"

#define AL_LREG 6                // hits per mate held in LDS
#define AL_LANC 64               // anchors per mate held in LDS
#define AL_LCIG 40               // CIGAR words held in LDS (region under construction / last DP call)
#define AL_LPTB 1280             // traceback bytes held in LDS
#define AL_GPB 4                 // extension groups per block (1: one fragment per wavefront -> no cross-group divergence)
#define GW 16                    // lanes per extension group (= one 128-bit SSE vector of int8)
#define KSW_NEG_INF -0x40000000
#define EZ_RIGHT      0x02
#define EZ_EXTZ_ONLY  0x40
#define EZ_REV_CIGAR  0x80
#define GSYNC() __syncthreads()  // blocks are one wavefront: this is a wave-level LDS/HBM fence, safe under group divergence

struct EzD { int max, zdropped, max_q, max_t, mqe, mqe_t, mte, score, n_cigar, reach_end; };

template <int TMAX, int QMAX> struct GroupLds {
	int8_t u[TMAX], v[TMAX], x[TMAX], y[TMAX], x2[TMAX], y2[TMAX], s[TMAX];
	int32_t H[TMAX];
	static constexpr bool kQrReady = false;   // d_ksw_reg builds the reversed query from qbuf
	static constexpr int kPtb = AL_LPTB;
	uint8_t sq[QMAX + 2 * (TMAX + 16) + 64];   // LDS-row DP: sf[tlen_*16] immediately followed by qr[] (one allocation in the reference: ksw2_extd2_sse.c:99-103); register DP: the reversed query between two pads of 16 bytes per block
	uint8_t tbuf[TMAX + 16];        // target of the current DP job / alignment window
	uint8_t qbuf[QMAX + 16];        // query of the current DP job
	uint8_t q0[QMAX + 16], q1[QMAX + 16];   // qseq0[0] (forward) and qseq0[1] (reverse complement), align.c:865-870
	// small per-fragment state kept on chip (HBM copies are used only when a fragment exceeds these tiles)
	AlReg regs[2][AL_LREG], rtmp[AL_LREG];
	AlAnchor aux128[AL_LREG], anc[AL_LANC];
	uint64_t aux64[AL_LREG], sc[AL_LREG * AL_LREG];
	int auxi[2 * AL_LREG];
	uint32_t cig[AL_LCIG], ezc[AL_LCIG];
	uint8_t ptb[AL_LPTB];
};

struct GroupWs {                    // per-group HBM scratch (+ the current location of the two CIGAR buffers)
	uint8_t *p;                     // traceback matrix (HBM)
	uint32_t *cig;                  // CIGAR of the region under construction (HBM spill area)
	uint32_t *ezc;                  // CIGAR returned by the last DP call (HBM spill area)
	uint64_t *sc;                   // pair scores (mm_pair), AL_PAIR_SC_CAP entries
	uint32_t *cur_cig; int cur_cig_cap;   // where the region CIGAR currently lives (LDS tile or ws.cig)
	uint32_t *cur_ezc;              // where the last DP call left its CIGAR
	unsigned long long *dbg;        // debug records (differential DP check)
	long long prof[8];              // AL_DBG bit 21: cycle accumulators [0] dp init [1] dp rows [2] backtrack [3] align1 other [4] post [5] stage-in [6] n_dp [7] n_rows
};
#ifndef AL_LB_DP8
#define AL_LB_DP8 4
#endif
#ifndef AL_LB_DP22
#define AL_LB_DP22 3
#endif
#ifndef AL_LB_REGS
#define AL_LB_REGS 5
#endif
#ifndef AL_LB_PREP
#define AL_LB_PREP 5
#endif
#ifndef AL_LB_FIN
#define AL_LB_FIN 4
#endif
#define PROF_ON(P) (((P).dbg >> 21) & 1)
#define AL_PAIR_SC_CAP 4096

__device__ __forceinline__ void d_ez_reset(EzD &ez)
{   // ksw2.h:153-158
	ez.max_q = ez.max_t = ez.mqe_t = -1; ez.max = 0; ez.score = ez.mqe = ez.mte = KSW_NEG_INF;
	ez.n_cigar = 0; ez.zdropped = 0; ez.reach_end = 0;
}

__device__ __forceinline__ void d_row_bounds(int r, int qlen, int tlen, int w, int &st, int &en)
{   // ksw2_extd2_sse.c:131-134
	st = 0; en = tlen - 1;
	if (st < r - qlen + 1) st = r - qlen + 1;
	if (en > r) en = r;
	if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
	if (en > (r + w) >> 1) en = (r + w) >> 1;
}

// cigar writer: every lane of the group holds the same state and stores the same words.  The run being extended is
// kept in registers (cur); c[] starts in LDS and migrates to the HBM scratch when it outgrows the tile.
struct CigW { uint32_t *c; int n; int cap; uint32_t *spill; uint32_t cur; };
__device__ __forceinline__ void d_cig_store(CigW &w, uint32_t v)
{
	if (w.n == w.cap) { for (int i = 0; i < w.n; ++i) w.spill[i] = w.c[i]; w.c = w.spill; w.cap = 0x7fffffff; }
	w.c[w.n++] = v;
}
__device__ __forceinline__ void d_push_cigar(CigW &w, uint32_t op, int len)
{   // ksw2.h:104-114
	if (w.cur != 0xffffffffu && op == (w.cur & 0xf)) w.cur += (uint32_t)len << 4;
	else { if (w.cur != 0xffffffffu) d_cig_store(w, w.cur); w.cur = (uint32_t)len << 4 | op; }
}
__device__ __forceinline__ void d_cig_flush(CigW &w) { if (w.cur != 0xffffffffu) { d_cig_store(w, w.cur); w.cur = 0xffffffffu; } }

__device__ __forceinline__ void d_backtrack(const uint8_t *p, int n_col, int qlen, int tlen, int w, int is_rev, int i0, int j0, CigW &cw)
{   // ksw_backtrack, ksw2.h:119-151 (is_rot = 1, min_intron_len = 0); off[]/off_end[] are recomputed from r
	int i = i0, j = j0, state = 0;
	cw.n = 0; cw.cur = 0xffffffffu;
	while (i >= 0 && j >= 0) {
		int force_state = -1, st, en; const int r = i + j;
		d_row_bounds(r, qlen, tlen, w, st, en);
		const int off = st / 16 * 16, off_end = (en + 16) / 16 * 16 - 1;
		if (i < off) force_state = 2;
		if (i > off_end) force_state = 1;
		const uint32_t tmp = force_state < 0 ? p[(size_t)r * n_col + i - off] : 0;
		if (state == 0) state = tmp & 7;
		else if (!(tmp >> (state + 2) & 1)) state = 0;
		if (state == 0) state = tmp & 7;
		if (force_state >= 0) state = force_state;
		if (state == 0) { d_push_cigar(cw, 0, 1); --i; --j; }
		else if (state == 1 || state == 3) { d_push_cigar(cw, 2, 1); --i; }
		else { d_push_cigar(cw, 1, 1); --j; }
	}
	if (i >= 0) d_push_cigar(cw, 2, i + 1);
	if (j >= 0) d_push_cigar(cw, 1, j + 1);
	d_cig_flush(cw);
	if (!is_rev) for (int k = 0; k < cw.n >> 1; ++k) { uint32_t t = cw.c[k]; cw.c[k] = cw.c[cw.n - 1 - k]; cw.c[cw.n - 1 - k] = t; }
}

// ksw_extd2_sse (ksw2_extd2_sse.c:26-393) for one 16-lane group, state rows in LDS (any size up to TMAX).
// Inputs: L.qbuf[0..qlen), L.tbuf[0..tlen).
template <int TMAX, int QMAX>
__device__ __forceinline__ void d_ksw_lds(GroupLds<TMAX, QMAX> &L, const int gl, GroupWs &ws, int qlen, int tlen, const AlParams &P,
                            int w, int zdrop, int end_bonus, int flag, EzD &ez)
{
	int q = P.q, e = P.e, q2 = P.q2, e2 = P.e2;
	d_ez_reset(ez);
	if (qlen <= 0 || tlen <= 0) return;
	if (q2 + e2 < q + e) { int t = q; q = q2; q2 = t; t = e; e = e2; e2 = t; }
	const int qe = q + e;
	const int8_t qe_ = (int8_t)(q + e), qe2_ = (int8_t)(q2 + e2);
	const int8_t sc_mch = (int8_t)P.a, sc_mis = (int8_t)(-P.b), sc_amb = (int8_t)(P.sc_ambi > 0 ? -P.sc_ambi : P.sc_ambi);
	const int8_t sc_N = sc_amb == 0 ? (int8_t)(-e2) : sc_amb;
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	const int tlen_ = (tlen + 15) / 16;
	int n_col_ = qlen < tlen ? qlen : tlen;
	n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
	const int qlen_ = (qlen + 15) / 16;
	{ int min_sc = sc_mis < sc_amb ? sc_mis : sc_amb; if (-min_sc > 2 * (q + e)) return; }
	int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
	uint8_t *sf = L.sq, *qr = L.sq + tlen_ * 16;
	// initialise (memset / kcalloc of the reference)
	for (int t = gl; t < tlen_ * 16; t += GW) {
		L.u[t] = L.v[t] = L.x[t] = L.y[t] = (int8_t)(-q - e);
		L.x2[t] = L.y2[t] = (int8_t)(-q2 - e2);
		L.s[t] = 0; L.H[t] = KSW_NEG_INF;
		sf[t] = t < tlen ? L.tbuf[t] : 0;
	}
	for (int t = gl; t < qlen_ * 16 + 32; t += GW) qr[t] = t < qlen ? L.qbuf[qlen - 1 - t] : 0;
	GSYNC();
	const size_t prow = (size_t)n_col_ * 16;
	uint8_t *const ptb = (size_t)(qlen + tlen - 1) * prow <= AL_LPTB ? L.ptb : ws.p;   // traceback bytes: LDS tile when they fit
	int last_st = -1, last_en = -1, r;
	for (r = 0; r < qlen + tlen - 1; ++r) {
		int st, en;
		d_row_bounds(r, qlen, tlen, w, st, en);
		if (st > en) { ez.zdropped = 1; break; }
		const int st0 = st, en0 = en;
		st = st / 16 * 16; en = (en + 16) / 16 * 16 - 1;
		int8_t x1, x21, v1;
		if (st > 0) {
			if (st - 1 >= last_st && st - 1 <= last_en) { x1 = L.x[st - 1]; x21 = L.x2[st - 1]; v1 = L.v[st - 1]; }
			else { x1 = (int8_t)(-q - e); x21 = (int8_t)(-q2 - e2); v1 = (int8_t)(-q - e); }
		} else {
			x1 = (int8_t)(-q - e); x21 = (int8_t)(-q2 - e2);
			v1 = r == 0 ? (int8_t)(-q - e) : r < long_thres ? (int8_t)(-e) : r == long_thres ? (int8_t)long_diff : (int8_t)(-e2);
		}
		if (en >= r) {
			L.y[r] = (int8_t)(-q - e); L.y2[r] = (int8_t)(-q2 - e2);
			L.u[r] = r == 0 ? (int8_t)(-q - e) : r < long_thres ? (int8_t)(-e) : r == long_thres ? (int8_t)long_diff : (int8_t)(-e2);
		}
		const uint8_t *qrr = qr + (qlen - 1 - r);
		for (int t = st0; t <= en0; t += 16) {                               // :158-176
			const uint8_t sq = sf[t + gl], sq2 = qrr[t + gl];
			int8_t sc = sq == sq2 ? sc_mch : sc_mis;
			if (sq == 4 || sq2 == 4) sc = sc_N;
			if (t + gl < tlen_ * 16) L.s[t + gl] = sc;    // the reference's 16-byte store may spill past s[] into bytes it never reads again
		}
		GSYNC();
		uint8_t *pr = ptb + (size_t)r * prow - st;
		int xc = x1, x2c = x21, vc = v1;                                      // carries across 16-cell blocks
		for (int tb = st; tb <= en; tb += 16) {                               // :182-306
			const int t = tb + gl;
			int8_t z = L.s[t];
			const int xo = L.x[t], vo = L.v[t], x2o = L.x2[t];
			const int8_t ut = L.u[t], yo = L.y[t], y2o = L.y2[t];
			int xt1 = __shfl_up(xo, 1, GW), vt1 = __shfl_up(vo, 1, GW), x2t1 = __shfl_up(x2o, 1, GW);
			if (gl == 0) { xt1 = xc; vt1 = vc; x2t1 = x2c; }
			xc = __shfl(xo, GW - 1, GW); vc = __shfl(vo, GW - 1, GW); x2c = __shfl(x2o, GW - 1, GW);
			int8_t a = (int8_t)(xt1 + vt1), b = (int8_t)(yo + ut), a2 = (int8_t)(x2t1 + vt1), b2 = (int8_t)(y2o + ut), d;
			if (!(flag & EZ_RIGHT)) {
				d = a > z ? 1 : 0;   z = z > a ? z : a;
				d = b > z ? 2 : d;   z = z > b ? z : b;
				d = a2 > z ? 3 : d;  z = z > a2 ? z : a2;
				d = b2 > z ? 4 : d;  z = z > b2 ? z : b2;
			} else {
				d = z > a ? 0 : 1;   z = z > a ? z : a;
				d = z > b ? d : 2;   z = z > b ? z : b;
				d = z > a2 ? d : 3;  z = z > a2 ? z : a2;
				d = z > b2 ? d : 4;  z = z > b2 ? z : b2;
			}
			z = z < sc_mch ? z : sc_mch;
			L.u[t] = (int8_t)(z - (int8_t)vt1); L.v[t] = (int8_t)(z - ut);
			int8_t tmp = (int8_t)(z - q); a = (int8_t)(a - tmp); b = (int8_t)(b - tmp);
			tmp = (int8_t)(z - q2); a2 = (int8_t)(a2 - tmp); b2 = (int8_t)(b2 - tmp);
			if (!(flag & EZ_RIGHT)) {
				L.x[t]  = (int8_t)((a  > 0 ? a  : 0) - qe_);  if (a  > 0) d |= 0x08;
				L.y[t]  = (int8_t)((b  > 0 ? b  : 0) - qe_);  if (b  > 0) d |= 0x10;
				L.x2[t] = (int8_t)((a2 > 0 ? a2 : 0) - qe2_); if (a2 > 0) d |= 0x20;
				L.y2[t] = (int8_t)((b2 > 0 ? b2 : 0) - qe2_); if (b2 > 0) d |= 0x40;
			} else {
				L.x[t]  = (int8_t)((a  >= 0 ? a  : 0) - qe_);  if (a  >= 0) d |= 0x08;
				L.y[t]  = (int8_t)((b  >= 0 ? b  : 0) - qe_);  if (b  >= 0) d |= 0x10;
				L.x2[t] = (int8_t)((a2 >= 0 ? a2 : 0) - qe2_); if (a2 >= 0) d |= 0x20;
				L.y2[t] = (int8_t)((b2 >= 0 ? b2 : 0) - qe2_); if (b2 >= 0) d |= 0x40;
			}
			pr[t] = (uint8_t)d;
		}
		GSYNC();
		{   // exact max (:307-361): H row update + argmax in the reference's evaluation order
			int max_H, max_t;
			if (r > 0) {
				const int Hen0 = en0 > 0 ? L.H[en0 - 1] + L.u[en0] : L.H[en0] + L.v[en0];
				const int en1 = st0 + (en0 - st0) / 4 * 4;
				int bh = Hen0, bo = 0, bt = en0;
				for (int t = st0 + gl; t < en0; t += GW) {
					const int h = L.H[t] + L.v[t];
					L.H[t] = h;
					const int ord = t < en1 ? 1 + ((t - st0) & 3) * 4096 + ((t - st0) >> 2) : 1 + 4 * 4096 + (t - en1);
					if (h > bh || (h == bh && ord < bo)) { bh = h; bo = ord; bt = t; }
				}
				for (int dlt = GW >> 1; dlt > 0; dlt >>= 1) {
					const int oh = __shfl_xor(bh, dlt, GW), oo = __shfl_xor(bo, dlt, GW), ot = __shfl_xor(bt, dlt, GW);
					if (oh > bh || (oh == bh && oo < bo)) { bh = oh; bo = oo; bt = ot; }
				}
				L.H[en0] = Hen0;
				max_H = bh; max_t = bt;
			} else { L.H[0] = L.v[0] - qe; max_H = L.H[0]; max_t = 0; }
			GSYNC();
			if (en0 == tlen - 1 && L.H[en0] > ez.mte) ez.mte = L.H[en0];
			if (r - st0 == qlen - 1 && L.H[st0] > ez.mqe) { ez.mqe = L.H[st0]; ez.mqe_t = st0; }
			// ksw_apply_zdrop, ksw2.h:160-176
			bool brk = false;
			if (max_H > ez.max) { ez.max = max_H; ez.max_t = max_t; ez.max_q = r - max_t; }
			else if (max_t >= ez.max_t && r - max_t >= ez.max_q) {
				const int tl = max_t - ez.max_t, ql = (r - max_t) - ez.max_q, l = tl > ql ? tl - ql : ql - tl;
				if (zdrop >= 0 && ez.max - max_H > zdrop + l * e2) { ez.zdropped = 1; brk = true; }
			}
			if (brk) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez.score = L.H[tlen - 1];
		}
		last_st = st; last_en = en;
	}
	GSYNC();
	{   // :384-392
		const int rev_cigar = !!(flag & EZ_REV_CIGAR);
		CigW cw{L.ezc, 0, AL_LCIG, ws.ezc, 0xffffffffu};
		if (!ez.zdropped && !(flag & EZ_EXTZ_ONLY)) d_backtrack(ptb, n_col_ * 16, qlen, tlen, w, rev_cigar, tlen - 1, qlen - 1, cw);
		else if (!ez.zdropped && (flag & EZ_EXTZ_ONLY) && ez.mqe + end_bonus > ez.max) { ez.reach_end = 1; d_backtrack(ptb, n_col_ * 16, qlen, tlen, w, rev_cigar, ez.mqe_t, qlen - 1, cw); }
		else if (ez.max_t >= 0 && ez.max_q >= 0) d_backtrack(ptb, n_col_ * 16, qlen, tlen, w, rev_cigar, ez.max_t, ez.max_q, cw);
		ez.n_cigar = cw.n; ws.cur_ezc = cw.c;
	}
	GSYNC();
}

#include "al_dev_ksw.h"
#include "al_dev_ksw2.h"
#include "al_dev_net.h"

// dispatcher: targets of up to 22 x 16 cells run register-resident, larger ones use the LDS rows
template <int TMAX, int QMAX>
__device__ __forceinline__ void d_ksw_extd2(GroupLds<TMAX, QMAX> &L, const int gl, GroupWs &ws, int qlen, int tlen, const AlParams &P,
                            int w, int zdrop, int end_bonus, int flag, EzD &ez)
{
	d_ez_reset(ez);
	if (qlen <= 0 || tlen <= 0) return;
	{ const int sc_mis = -P.b, sc_amb = P.sc_ambi > 0 ? -P.sc_ambi : P.sc_ambi; const int min_sc = sc_mis < sc_amb ? sc_mis : sc_amb;
	  const int qe1 = P.q + P.e, qe2 = P.q2 + P.e2; if (-min_sc > 2 * (qe1 < qe2 ? qe1 : qe2)) return; }
	const int tlen_ = (tlen + 15) / 16;
	const int maxnb = ((P.dbg >> 8) & 0xff) ? ((P.dbg >> 8) & 0xff) - 1 : 22;
   // AL_DBG>>8 = 1 + largest block count allowed on the register path (experiments)
	if ((P.dbg & 128) || tlen_ > maxnb) d_ksw_lds(L, gl, ws, qlen, tlen, P, w, zdrop, end_bonus, flag, ez);
	else if (tlen_ <= 1) d_ksw_reg<1>(L, gl, ws, qlen, tlen, P, w, zdrop, end_bonus, flag, ez);
	else if (tlen_ <= 2) d_ksw_reg<2>(L, gl, ws, qlen, tlen, P, w, zdrop, end_bonus, flag, ez);
	else if (tlen_ <= 4) d_ksw_reg<4>(L, gl, ws, qlen, tlen, P, w, zdrop, end_bonus, flag, ez);
	else if (tlen_ <= 8) d_ksw_reg<8>(L, gl, ws, qlen, tlen, P, w, zdrop, end_bonus, flag, ez);
	else if (tlen_ <= 22) d_ksw_reg<22>(L, gl, ws, qlen, tlen, P, w, zdrop, end_bonus, flag, ez);
	else d_ksw_lds(L, gl, ws, qlen, tlen, P, w, zdrop, end_bonus, flag, ez);
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int d_mat(const AlParams &P, int ct, int cq)
{   // ksw_gen_simple_mat, align.c:9-22 (m = 5)
	const int amb = P.sc_ambi > 0 ? -P.sc_ambi : P.sc_ambi;
	if (ct > 3 || cq > 3) return amb;
	return ct == cq ? (P.a < 0 ? -P.a : P.a) : (P.b > 0 ? -P.b : P.b);
}

// sequence accessors: the same scalar routines run on LDS byte tiles (monolithic kernel) or straight on the packed
// read / reference words in HBM (lane-per-fragment kernels)
struct BytesAcc { const uint8_t *p; __device__ __forceinline__ int operator()(int i) const { return p[i]; } __device__ __forceinline__ BytesAcc shift(int n) const { return BytesAcc{p + n}; } };
struct ReadAcc {     // qseq0[rev][off + i] of a read packed 4 bit/base in mapping orientation (align.c:865-870)
	const uint32_t *seq; int qlen, rev, off;
	mutable int cw = -1; mutable uint32_t cv = 0;          // last word fetched: sequential scans cost one load per 8 bases
	__device__ __forceinline__ int nib(int k) const { const int w = k >> 3; if (w != cw) { cw = w; cv = seq[w]; } return (int)((cv >> ((k & 7) << 2)) & 0xf); }
	__device__ __forceinline__ int operator()(int i) const {
		const int j = off + i;
		if (!rev) return nib(j);
		const int c = nib(qlen - 1 - j);
		return c < 4 ? 3 - c : 4;
	}
	__device__ __forceinline__ ReadAcc shift(int n) const { return ReadAcc{seq, qlen, rev, off + n}; }
	// 8 consecutive codes (nibble j = position i + j); the caller guarantees 0 <= off + i and off + i + 7 <= qlen - 1
	__device__ __forceinline__ uint32_t win8(int i) const {
		const int j = off + i, src = rev ? qlen - 8 - j : j, w = src >> 3, sh = (src & 7) << 2;
		uint32_t v = (uint32_t)((((uint64_t)seq[w + 1] << 32) | seq[w]) >> sh);
		if (rev) {                                     // reverse the nibble order, complement codes < 4
			v = ((v & 0x0f0f0f0fu) << 4) | ((v >> 4) & 0x0f0f0f0fu); v = __builtin_bswap32(v);
			const uint32_t amb = (v >> 2) & 0x11111111u;
			v ^= 0x33333333u & ~(amb * 3u);
		}
		return v;
	}
};
struct RefAcc {
	const uint32_t *S4; uint64_t base;
	mutable uint64_t cw = ~0ULL; mutable uint32_t cv = 0;
	__device__ __forceinline__ int operator()(int i) const { const uint64_t a = base + (uint64_t)(int64_t)i, w = a >> 3; if (w != cw) { cw = w; cv = S4[w]; } return (int)((cv >> ((a & 7) << 2)) & 0xf); }
	__device__ __forceinline__ RefAcc shift(int n) const { return RefAcc{S4, base + (uint64_t)(int64_t)n}; }
	__device__ __forceinline__ uint32_t win8(int i) const { const uint64_t a = base + (uint64_t)(int64_t)i, w = a >> 3; const int sh = (int)(a & 7) << 2; return (uint32_t)((((uint64_t)S4[w + 1] << 32) | S4[w]) >> sh); }
};
template <class A> struct HasWin8 { static const bool v = false; };
template <> struct HasWin8<ReadAcc> { static const bool v = true; };
template <> struct HasWin8<RefAcc> { static const bool v = true; };

template <class A> __device__ __forceinline__ uint32_t d_win8(const A &, int) { return 0; }
template <> __device__ __forceinline__ uint32_t d_win8<ReadAcc>(const ReadAcc &a, int i) { return a.win8(i); }
template <> __device__ __forceinline__ uint32_t d_win8<RefAcc>(const RefAcc &a, int i) { return a.win8(i); }
// four windows of eight bases of a query and a target, their sixteen loads in flight together (a loop of one window per step waits for its loads every step:
// 19 dependent round trips for a 150-base read)
template <class QA, class TA> __device__ __forceinline__ void d_win8x4(const QA &q, const int qo, const TA &t, const int to, uint32_t (&qw)[4], uint32_t (&tw)[4])
{
#pragma unroll
	for (int u = 0; u < 4; ++u) { qw[u] = d_win8(q, qo + 8 * u); tw[u] = d_win8(t, to + 8 * u); }
#pragma unroll
	for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(qw[u]), "+v"(tw[u]));
}

template <class QA, class TA>
__device__ int d_test_zdrop(const AlParams &P, QA qseq, TA tseq, int n_cigar, const uint32_t *cigar)
{   // mm_test_zdrop, align.c:47-89; the inversion branch is unreachable with MM_F_SR (align.c:72)
	int score = 0, max = INT32_MIN, max_i = -1, max_j = -1, i = 0, j = 0, max_zdrop = 0;
	for (int k = 0; k < n_cigar; ++k) {
		const int op = cigar[k] & 0xf, len = cigar[k] >> 4;
		if (op == 0) {
			for (int l = 0; l < len; ++l) {
				score += d_mat(P, tseq(i + l), qseq(j + l));
				if (score < max) { const int li = i + l - max_i, lj = j + l - max_j, diff = li > lj ? li - lj : lj - li, z = max - score - diff * P.e; if (z > max_zdrop) max_zdrop = z; }
				else { max = score; max_i = i + l; max_j = j + l; }
			}
			i += len; j += len;
		} else if (op == 1 || op == 2 || op == 3) {
			score -= P.q + P.e * len;
			if (op == 1) j += len; else i += len;
			if (score < max) { const int li = i - max_i, lj = j - max_j, diff = li > lj ? li - lj : lj - li, z = max - score - diff * P.e; if (z > max_zdrop) max_zdrop = z; }
			else { max = score; max_i = i; max_j = j; }
		}
	}
	return max_zdrop > P.zdrop ? 1 : 0;
}

template <class CG, class QA, class TA>
__device__ __forceinline__ void d_fix_cigar(AlReg *r, CG cigar, QA qseq, TA tseq, int *qshift, int *tshift)
{   // mm_fix_cigar, align.c:91-167
	int toff = 0, qoff = 0, to_shrink = 0; uint32_t k;
	*qshift = *tshift = 0;
	if (r->n_cigar <= 1) return;
	for (k = 0; k < r->n_cigar; ++k) {
		const uint32_t op = cigar[k] & 0xf, len = cigar[k] >> 4;
		if (len == 0) to_shrink = 1;
		if (op == 0) { toff += len; qoff += len; }
		else if (op == 1 || op == 2) {
			if (k > 0 && k < r->n_cigar - 1 && (cigar[k - 1] & 0xf) == 0 && (cigar[k + 1] & 0xf) == 0) {
				int l; const int prev_len = cigar[k - 1] >> 4;
				if (op == 1) { for (l = 0; l < prev_len; ++l) if (qseq(qoff - 1 - l) != qseq(qoff + len - 1 - l)) break; }
				else { for (l = 0; l < prev_len; ++l) if (tseq(toff - 1 - l) != tseq(toff + len - 1 - l)) break; }
				if (l > 0) { cigar[k - 1] -= l << 4; cigar[k + 1] += l << 4; qoff -= l; toff -= l; }
				if (l == prev_len) to_shrink = 1;
			}
			if (op == 1) qoff += len; else toff += len;
		} else if (op == 3) toff += len;
	}
	for (k = 0; k + 2 < r->n_cigar; ++k) {
		if ((cigar[k] & 0xf) > 0 && (cigar[k] & 0xf) + (cigar[k + 1] & 0xf) == 3) {
			uint32_t l, s1 = 0, s2 = 0;
			for (l = k; l < r->n_cigar; ++l) {
				const uint32_t cl = cigar[l], op = cl & 0xf;
				if (op == 1) s1 += cl >> 4; else if (op == 2) s2 += cl >> 4; else if (cl >> 4 != 0) break;
			}
			if (s1 > 0 && s2 > 0 && l - k > 2) {
				cigar[k] = s1 << 4 | 1; cigar[k + 1] = s2 << 4 | 2;
				for (k += 2; k < l; ++k) cigar[k] &= 0xf;
				to_shrink = 1;
			}
			k = l;
		}
	}
	if (to_shrink) {
		uint32_t l = 0;
		for (k = 0; k < r->n_cigar; ++k) if (cigar[k] >> 4 != 0) cigar[l++] = cigar[k];
		r->n_cigar = l;
		for (k = l = 0; k < r->n_cigar; ++k)
			if (k == r->n_cigar - 1 || (cigar[k] & 0xf) != (cigar[k + 1] & 0xf)) cigar[l++] = cigar[k];
			else cigar[k + 1] += cigar[k] >> 4 << 4;
		r->n_cigar = l;
	}
	if ((cigar[0] & 0xf) == 1 || (cigar[0] & 0xf) == 2) {
		const int l = cigar[0] >> 4;
		if ((cigar[0] & 0xf) == 1) { if (r->flags & ALR_REV) r->qe -= l; else r->qs += l; *qshift = l; }
		else { r->rs += l; *tshift = l; }
		--r->n_cigar;
		for (k = 0; k < r->n_cigar; ++k) cigar[k] = cigar[k + 1];
	}
}

template <bool BATCH = false /* four windows' loads at a time: k_ext_prep's one-run hits (in k_ext_finish it costs registers: 5.1 -> 6.3 ms) */, class CG, class QA, class TA>
__device__ __forceinline__ void d_update_extra(const AlParams &P, AlReg *r, CG cigar, QA qseq0_, TA tseq0_)
{   // mm_update_extra, align.c:240-286
	int s = 0, max = 0, qshift, tshift, toff = 0, qoff = 0;
	d_fix_cigar(r, cigar, qseq0_, tseq0_, &qshift, &tshift);
	const QA qseq = qseq0_.shift(qshift); const TA tseq = tseq0_.shift(tshift);
	r->blen = r->mlen = 0;
	for (uint32_t k = 0; k < r->n_cigar; ++k) {
		const uint32_t op = cigar[k] & 0xf, len = cigar[k] >> 4;
		if (op == 0) {
			int n_ambi = 0, n_diff = 0;
			uint32_t l = 0;
			if (HasWin8<QA>::v && HasWin8<TA>::v) {
				// packed sequences: eight bases per step; a window of eight unambiguous matches only raises s (and max with it)
				const int msc = P.a < 0 ? -P.a : P.a;
				auto win = [&](const uint32_t qw, const uint32_t tw) {
					if (qw == tw && !(qw & 0x44444444u)) { s += 8 * msc; max = max > s ? max : s; return; }
#pragma unroll
					for (int b = 0; b < 8; ++b) {
						const int cq = (int)(qw >> (4 * b) & 0xf), ct = (int)(tw >> (4 * b) & 0xf);
						if (ct > 3 || cq > 3) ++n_ambi; else if (ct != cq) ++n_diff;
						s += d_mat(P, ct, cq);
						if (s < 0) s = 0; else max = max > s ? max : s;
					}
				};
				if constexpr (BATCH) {
					for (; l + 32 <= len; l += 32) {
						uint32_t qw[4], tw[4]; d_win8x4(qseq, qoff + (int)l, tseq, toff + (int)l, qw, tw);
#pragma unroll
						for (int u = 0; u < 4; ++u) win(qw[u], tw[u]);
					}
					for (; l + 8 <= len; l += 8) win(d_win8(qseq, qoff + (int)l), d_win8(tseq, toff + (int)l));
				} else {
					for (; l + 8 <= len; l += 8) {
						const uint32_t qw = d_win8(qseq, qoff + (int)l), tw = d_win8(tseq, toff + (int)l);
						if (qw == tw && !(qw & 0x44444444u)) { s += 8 * msc; max = max > s ? max : s; continue; }
#pragma unroll
						for (int b = 0; b < 8; ++b) {
							const int cq = (int)(qw >> (4 * b) & 0xf), ct = (int)(tw >> (4 * b) & 0xf);
							if (ct > 3 || cq > 3) ++n_ambi; else if (ct != cq) ++n_diff;
							s += d_mat(P, ct, cq);
							if (s < 0) s = 0; else max = max > s ? max : s;
						}
					}
				}
			}
			for (; l < len; ++l) {
				const int cq = qseq(qoff + l), ct = tseq(toff + l);
				if (ct > 3 || cq > 3) ++n_ambi; else if (ct != cq) ++n_diff;
				s += d_mat(P, ct, cq);
				if (s < 0) s = 0; else max = max > s ? max : s;
			}
			r->blen += len - n_ambi; r->mlen += len - (n_ambi + n_diff); r->n_ambi += n_ambi;
			toff += len; qoff += len;
		} else if (op == 1) {
			int n_ambi = 0;
			for (uint32_t l = 0; l < len; ++l) if (qseq(qoff + l) > 3) ++n_ambi;
			r->blen += len - n_ambi; r->n_ambi += n_ambi;
			s -= P.q + P.e * len; if (s < 0) s = 0;
			qoff += len;
		} else if (op == 2) {
			int n_ambi = 0;
			for (uint32_t l = 0; l < len; ++l) if (tseq(toff + l) > 3) ++n_ambi;
			r->blen += len - n_ambi; r->n_ambi += n_ambi;
			s -= P.q + P.e * len; if (s < 0) s = 0;
			toff += len;
		} else if (op == 3) toff += len;
	}
	r->dp_max = max;
}

__device__ __forceinline__ void d_append_cigar(AlReg *r, GroupWs &ws, int n_cigar, const uint32_t *cigar)
{   // mm_append_cigar, align.c:288-311
	if (n_cigar == 0) return;
	if ((int)r->n_cigar + n_cigar > ws.cur_cig_cap) {      // outgrew the LDS tile: continue in the HBM scratch
		for (uint32_t i = 0; i < r->n_cigar; ++i) ws.cig[i] = ws.cur_cig[i];
		ws.cur_cig = ws.cig; ws.cur_cig_cap = 0x7fffffff;
	}
	uint32_t *cig = ws.cur_cig;
	r->flags |= ALR_HAS_P;
	if (r->n_cigar > 0 && (cig[r->n_cigar - 1] & 0xf) == (cigar[0] & 0xf)) {
		cig[r->n_cigar - 1] += cigar[0] >> 4 << 4;
		for (int i = 1; i < n_cigar; ++i) cig[r->n_cigar + i - 1] = cigar[i];
		r->n_cigar += n_cigar - 1;
	} else {
		for (int i = 0; i < n_cigar; ++i) cig[r->n_cigar + i] = cigar[i];
		r->n_cigar += n_cigar;
	}
}

__device__ __forceinline__ void d_max_stretch(const AlReg *r, const AlAnchor *a, int *as, int *cnt)
{   // mm_max_stretch, align.c:495-521
	*as = r->as; *cnt = r->cnt;
	if (r->cnt < 2) return;
	int max_score = -1, max_i = -1, max_len = 0, score = (int)(a[r->as].y >> 32 & 0xff), len = 1, i;
	for (i = r->as + 1; i < r->as + r->cnt; ++i) {
		const int q_span = (int)(a[i].y >> 32 & 0xff);
		const int lr = (int32_t)a[i].x - (int32_t)a[i - 1].x, lq = (int32_t)a[i].y - (int32_t)a[i - 1].y;
		if (lq == lr) { score += lq < q_span ? lq : q_span; ++len; }
		else { if (score > max_score) { max_score = score; max_len = len; max_i = i - len; } score = q_span; len = 1; }
	}
	if (score > max_score) { max_score = score; max_len = len; max_i = i - len; }
	*as = max_i; *cnt = max_len;
}

struct AlignShared {            // read-only kernel inputs
	const uint32_t *S4; const uint64_t *seq_off; const uint32_t *seq_len;
	uint32_t *arena; unsigned long long *arena_cnt; uint64_t arena_cap;
	unsigned long long *dbg;        // 1 + 32*16 words of debug records
	unsigned long long *counters;   // [4] regions aligned, [5] ref bases, [6] cigar ops, [7] errors, [8] logf misses, [9] arena overflow, [10] sort ties
};

// mm_align1 (align.c:565-788), short-read branch.  Executed by the 16 lanes of a group in lock step.
template <int TMAX, int QMAX>
__device__ __forceinline__ void d_align1(GroupLds<TMAX, QMAX> &L, const int gl, GroupWs &ws, const AlParams &P, const AlignShared &G,
                         int qlen, AlReg *r, AlReg *r2, const AlAnchor *a, EzD &ez)
{
	const int32_t rid = (int32_t)(a[r->as].x << 1 >> 33), rev = (int32_t)(a[r->as].x >> 63);
	int32_t as1, cnt1, l, dropped = 0, rs0, re0, qs0, qe0, rs, re, qs, qe, rs1, qs1, re1, qe1;
	r2->cnt = 0;
	if (r->cnt == 0) return;
	const int bw = (int)(P.bw * 1.5 + 1.);
	const int32_t ref_len = (int32_t)G.seq_len[rid]; const uint64_t ref_off = G.seq_off[rid];
	const uint8_t *qseq0 = rev ? L.q1 : L.q0;
	d_max_stretch(r, a, &as1, &cnt1);
	rs = (int32_t)a[as1].x + 1 - (int32_t)(a[as1].y >> 32 & 0xff);
	qs = (int32_t)a[as1].y + 1 - (int32_t)(a[as1].y >> 32 & 0xff);
	re = (int32_t)a[as1 + cnt1 - 1].x + 1;
	qe = (int32_t)a[as1 + cnt1 - 1].y + 1;
	qs0 = 0; qe0 = qlen;                                                      // align.c:613-620
	l = qs;
	l += l * P.a + P.end_bonus > P.q ? (l * P.a + P.end_bonus - P.q) / P.e : 0;
	rs0 = rs - l > 0 ? rs - l : 0;
	l = qlen - qe;
	l += l * P.a + P.end_bonus > P.q ? (l * P.a + P.end_bonus - P.q) / P.e : 0;
	re0 = re + l < ref_len ? re + l : ref_len;
	if (re0 - rs0 > TMAX || qlen > QMAX) { if (gl == 0) atomicAdd(&G.counters[7], 1ULL << 16); r->cnt = 0; return; }
	if (gl == 0) { atomicAdd(&G.counters[4], 1ULL); atomicAdd(&G.counters[5], (unsigned long long)(re0 - rs0)); }
	r->n_cigar = 0; r->dp_score = 0; r->dp_max = 0; r->dp_max2 = 0; r->n_ambi = 0;
	ws.cur_cig = L.cig; ws.cur_cig_cap = AL_LCIG;

	// The three DP opportunities of mm_align1 (left extension, z-drop re-alignment of the ungapped core, right extension)
	// go through ONE call site of the DP so that it is inlined once per kernel.
	rs1 = rs; qs1 = qs; re1 = rs; qe1 = qs;
	for (int ph = 0; ph < 3; ++ph) {
		bool run = false; int ql = 0, tl = 0, zd = P.zdrop, eb = P.end_bonus, fl = 0;
		if (ph == 0) {                                                        // left extension, align.c:690-705
			if (qs > 0 && rs > 0 && !(P.dbg & 1)) {
				ql = qs - qs0; tl = rs - rs0;
				for (int i = gl; i < ql; i += GW) L.qbuf[i] = qseq0[qs0 + (ql - 1 - i)];                     // mm_seq_rev of both
				for (int i = gl; i < tl; i += GW) L.tbuf[i] = (uint8_t)d_seq4(G.S4, ref_off + (uint64_t)(rs - 1 - i));
				zd = (r->flags & ALR_SPLIT_INV) ? P.zdrop_inv : P.zdrop; fl = EZ_EXTZ_ONLY | EZ_RIGHT | EZ_REV_CIGAR; run = true;
			}
		} else if (ph == 1) {                                                 // ungapped core, align.c:709-758 with is_sr (i = cnt1 - 1)
			const int i = cnt1 - 1;
			re = (int32_t)a[as1 + i].x + 1; qe = (int32_t)a[as1 + i].y + 1;
			re1 = re; qe1 = qe;
			const int len = qe - qs;
			for (int k = gl; k < len; k += GW) { L.qbuf[k] = qseq0[qs + k]; L.tbuf[k] = (uint8_t)d_seq4(G.S4, ref_off + (uint64_t)(rs + k)); }
			GSYNC();
			d_ez_reset(ez);
			int sc = 0;
			if (!(P.dbg & 4)) for (int k = 0; k < len; ++k) {
				if (L.qbuf[k] >= 4 || L.tbuf[k] >= 4) sc += P.e2;
				else sc += L.qbuf[k] == L.tbuf[k] ? P.a : -P.b;
			}
			ez.score = sc;
			L.ezc[0] = (uint32_t)len << 4; ez.n_cigar = 1; ws.cur_ezc = L.ezc;
			if (!(P.dbg & 4) && d_test_zdrop(P, BytesAcc{L.qbuf}, BytesAcc{L.tbuf}, ez.n_cigar, L.ezc) != 0) {   // second pass (align.c:736-737)
				ql = len; tl = re - rs; zd = P.zdrop; eb = -1; fl = 0; run = true;
			}
		} else {                                                              // right extension, align.c:760-771
			if (!dropped && qe < qe0 && re < re0 && !(P.dbg & 1)) {
				ql = qe0 - qe; tl = re0 - re;
				for (int i = gl; i < ql; i += GW) L.qbuf[i] = qseq0[qe + i];
				for (int i = gl; i < tl; i += GW) L.tbuf[i] = (uint8_t)d_seq4(G.S4, ref_off + (uint64_t)(re + i));
				fl = EZ_EXTZ_ONLY; run = true;
			}
		}
		if (run) { GSYNC(); d_ksw_extd2(L, gl, ws, ql, tl, P, bw, zd, eb, fl, ez); }
		if (ph == 0) {
			if (run) {
				if (ez.n_cigar > 0) { d_append_cigar(r, ws, ez.n_cigar, ws.cur_ezc); r->dp_score += ez.max; }
				rs1 = rs - (ez.reach_end ? ez.mqe_t + 1 : ez.max_t + 1);
				qs1 = qs - (ez.reach_end ? qs - qs0 : ez.max_q + 1);
			}
		} else if (ph == 1) {
			const int i = cnt1 - 1;
			if (ez.n_cigar > 0) d_append_cigar(r, ws, ez.n_cigar, ws.cur_ezc);
			if (ez.zdropped) {
				int j;
				for (j = i - 1; j >= 0; --j) if ((int32_t)a[as1 + j].x <= rs + ez.max_t) break;
				dropped = 1;
				if (j < 0) j = 0;
				r->dp_score += ez.max;
				re1 = rs + (ez.max_t + 1);
				qe1 = qs + (ez.max_q + 1);
				if (cnt1 - (j + 1) >= P.min_cnt) d_split_reg(r, r2, as1 + j + 1 - r->as, qlen, a);
			} else { r->dp_score += ez.score; rs = re; qs = qe; }
		} else if (run) {
			if (ez.n_cigar > 0) { d_append_cigar(r, ws, ez.n_cigar, ws.cur_ezc); r->dp_score += ez.max; }
			re1 = re + (ez.reach_end ? ez.mqe_t + 1 : ez.max_t + 1);
			qe1 = qe + (ez.reach_end ? qe0 - qe : ez.max_q + 1);
		}
	}
	r->rs = rs1; r->re = re1;
	if (rev) { r->qs = qlen - qe1; r->qe = qlen - qs1; }
	else { r->qs = qs1; r->qe = qe1; }
	if (r->flags & ALR_HAS_P) {
		const int tl = re1 - rs1;
		for (int i = gl; i < tl; i += GW) L.tbuf[i] = (uint8_t)d_seq4(G.S4, ref_off + (uint64_t)(rs1 + i));
		GSYNC();
		if (!(P.dbg & 2)) d_update_extra(P, r, ws.cur_cig, BytesAcc{qseq0 + qs1}, BytesAcc{L.tbuf}); else r->dp_max = 100;
		// publish the finished CIGAR: reserve words in the global arena (one atomic per group) and copy
		if (r->n_cigar <= 4) { for (uint32_t i = 0; i < r->n_cigar; ++i) r->cig_inl[i] = ws.cur_cig[i]; r->cigar_off = AL_CIG_INLINE; }
		else {
			unsigned long long off = 0;
			if (gl == 0) off = atomicAdd(G.arena_cnt, (unsigned long long)r->n_cigar);
			off = __shfl(off, 0, GW);
			if (off + r->n_cigar <= G.arena_cap) { for (uint32_t i = gl; i < r->n_cigar; i += GW) G.arena[off + i] = ws.cur_cig[i]; r->cigar_off = (uint32_t)off; }
			else { if (gl == 0) atomicAdd(&G.counters[9], 1ULL); r->cigar_off = 0xffffffffu; }
		}
		if (gl == 0) atomicAdd(&G.counters[6], (unsigned long long)r->n_cigar);
	}
	GSYNC();
}

// mm_pair, pe.c:76-177 (+ mm_set_pe_thru pe.c:45-64).  pa: scratch (n0+n1) x 3 words; sc: scratch u64
struct PairEnt { uint64_t key; int32_t s, rev, idx; int32_t pad; };
struct PairAcc {
	typedef PairEnt E; PairEnt *a;
	AL_D uint64_t key(int i) const { return a[i].key; }
	AL_D uint64_t keyof(const PairEnt &e) const { return e.key; }
	AL_D PairEnt get(int i) const { return a[i]; }
	AL_D void set(int i, const PairEnt &e) { a[i] = e; }
};
// mm_pair + mm_set_pe_thru (pe.c:45-177).  Mate ids, strands and the two hit arrays are selected with two-way selects
// instead of indexing small local arrays by run-time values: those arrays would be placed in scratch memory.
__device__ __forceinline__ void d_pair2(const AlParams &P, int max_gap_ref, const int ql0, const int ql1, const int n0, const int n1, AlReg *const regs0, AlReg *const regs1,
                                        PairEnt *pa, uint64_t *sc, int sc_cap, const AlLogTab &lt, bool *tie, bool *ovf)
{
	const int sub_diff = P.a * 2 + P.b, match_sc = P.a;
	int n = 0, dp_thres = 0, segs = 0;
#define RG(s_) ((s_) ? regs1 : regs0)
#define NR(s_) ((s_) ? n1 : n0)
#pragma unroll
	for (int s = 0; s < 2; ++s) {
		int mx = 0;
		for (int i = 0; i < NR(s); ++i) {
			const AlReg *r = &RG(s)[i];
			PairEnt e; e.s = s; e.idx = i; e.rev = (r->flags & ALR_REV) ? 1 : 0;
			e.key = (uint64_t)(uint32_t)r->rid << 32 | (uint64_t)(uint32_t)(r->rs << 1) | (uint64_t)(s ^ e.rev);
			pa[n] = e;
			mx = mx > r->dp_max ? mx : r->dp_max;
			++n; segs |= 1 << s;
		}
		dp_thres += mx;
	}
	if (segs == 3) {
		dp_thres -= P.pe_bonus; if (dp_thres < 0) dp_thres = 0;
		// radix_sort_pair (ksort.h:147-151): stable insertion sort up to 64 entries, the reference's radix permutation above
		// (work area: the rest of the hit scratch behind pa[n]; more than 64 entries implies a work area of >= 33 hits)
		if (n <= 64) { for (int i = 1; i < n; ++i) if (pa[i].key < pa[i - 1].key) { PairEnt t = pa[i]; int j = i; for (; j > 0 && t.key < pa[j - 1].key; --j) pa[j] = pa[j - 1]; pa[j] = t; } }
		else { PairAcc acc{pa}; if (d_rs_sort(acc, n, (uint16_t *)(pa + n))) *tie = true; }
		long long max = -1; int max_idx0 = -1, max_idx1 = -1, last0 = -1, last1 = -1; int n_sc = 0;
#define PR(e) (&RG((e).s)[(e).idx])
		for (int i = 0; i < n; ++i) {
			const PairEnt ei = pa[i];
			if (ei.key & 1) {
				const int lst = ei.rev ? last1 : last0;
				if (lst < 0) continue;
				const AlReg *r = PR(ei), *q = PR(pa[lst]);
				if (r->rid != q->rid || r->rs - q->re > max_gap_ref) continue;
				for (int j = lst; j >= 0; --j) {
					const PairEnt ej = pa[j];
					if (ej.rev != ei.rev || ej.s == ei.s) continue;
					q = PR(ej);
					if (r->rid != q->rid || r->rs - q->re > max_gap_ref) break;
					if (r->dp_max + q->dp_max < dp_thres) continue;
					const long long score = (long long)((uint64_t)(uint32_t)(r->dp_max + q->dp_max) << 32 | (uint32_t)(r->hash + q->hash));
					if (score > max) { max = score; if (ej.s) max_idx1 = j; else max_idx0 = j; if (ei.s) max_idx1 = i; else max_idx0 = i; }
					if (n_sc < sc_cap) sc[n_sc++] = (uint64_t)score; else *ovf = true;
				}
			} else { if (ei.rev) last1 = i; else last0 = i; }
		}
		if (n_sc > 1) d_sort64(sc, n_sc);
		if (n_sc > 0 && max > 0) {
			int n_sub = 0, mapq_pe;
			AlReg *const ra = PR(pa[max_idx0]), *const rb = PR(pa[max_idx1]);
			ra->flags |= ALR_PROPER; rb->flags |= ALR_PROPER;
#pragma unroll
			for (int s = 0; s < 2; ++s) {
				AlReg *const rs_ = s ? rb : ra;
				if (rs_->id != rs_->parent) {
					AlReg *p = &RG(s)[rs_->parent];
					for (int i = 0; i < NR(s); ++i) if (RG(s)[i].parent == p->id) RG(s)[i].parent = rs_->id;
					p->mapq = 0;
				}
				if (!(rs_->flags & ALR_SAM_PRI)) {
					for (int i = 0; i < NR(s); ++i) RG(s)[i].flags &= ~ALR_SAM_PRI;
					rs_->flags |= ALR_SAM_PRI;
				}
			}
			mapq_pe = ra->mapq > rb->mapq ? (int)ra->mapq : (int)rb->mapq;
			for (int i = 0; i < n_sc; ++i) if ((sc[i] >> 32) + sub_diff >= (uint64_t)max >> 32) ++n_sub;
			if (n_sc > 1) {
				const uint64_t diff = (uint64_t)(max >> 32) - (sc[n_sc - 2] >> 32);
				const int mapq_pe_alt = (int)__fsub_rn(al_fdiv(__fmul_rn(6.02f, (float)diff), (float)match_sc), __fmul_rn(4.343f, al_logf_i(lt, n_sub)));
				mapq_pe = mapq_pe < mapq_pe_alt ? mapq_pe : mapq_pe_alt;
			}
			if ((int)ra->mapq < mapq_pe) ra->mapq = (uint32_t)(int)__fadd_rn(__fadd_rn(__fmul_rn(.2f, (float)ra->mapq), __fmul_rn(.8f, (float)mapq_pe)), .499f) & 0xffu;
			if ((int)rb->mapq < mapq_pe) rb->mapq = (uint32_t)(int)__fadd_rn(__fadd_rn(__fmul_rn(.2f, (float)rb->mapq), __fmul_rn(.8f, (float)mapq_pe)), .499f) & 0xffu;
			if (n_sc == 1) { if (ra->mapq < 2) ra->mapq = 2; if (rb->mapq < 2) rb->mapq = 2; }
			else if ((uint64_t)max >> 32 > sc[n_sc - 2] >> 32) { if (ra->mapq < 1) ra->mapq = 1; if (rb->mapq < 1) rb->mapq = 1; }
		}
#undef PR
	}
	// mm_set_pe_thru
	int n_pri0 = 0, n_pri1 = 0, pri0 = -1, pri1 = -1;
	for (int i = 0; i < n0; ++i) if (regs0[i].id == regs0[i].parent) { ++n_pri0; pri0 = i; }
	for (int i = 0; i < n1; ++i) if (regs1[i].id == regs1[i].parent) { ++n_pri1; pri1 = i; }
	if (n_pri0 == 1 && n_pri1 == 1) {
		AlReg *p = &regs0[pri0], *q = &regs1[pri1];
		const int d1 = p->rs - q->rs, d2 = p->re - q->re;
		if (p->rid == q->rid && (p->flags & ALR_REV) == (q->flags & ALR_REV) && (d1 < 0 ? -d1 : d1) < 3 && (d2 < 0 ? -d2 : d2) < 3
		    && ((p->qs == 0 && ql1 - q->qe == 0) || (q->qs == 0 && ql0 - p->qe == 0))) { p->flags |= ALR_PE_THRU; q->flags |= ALR_PE_THRU; }
	}
#undef RG
#undef NR
}
__device__ __forceinline__ void d_pair(const AlParams &P, int max_gap_ref, const int *qlens, int *n_regs, AlReg *const *regs, PairEnt *pa, uint64_t *sc, int sc_cap, const AlLogTab &lt, bool *tie, bool *ovf)
{
	d_pair2(P, max_gap_ref, qlens[0], qlens[1], n_regs[0], n_regs[1], regs[0], regs[1], pa, sc, sc_cap, lt, tie, ovf);
}

// d_pair2 for exactly one hit per mate, both held in registers (no sort scratch, no score list): the only candidate pair is
// (first, second) in key order, found iff the first is on the "left" side (key bit 0 clear), the second on the "right",
// same strand class and contig, within max_gap_ref and above the score threshold (pe.c:76-150); then the single-pair
// MAPQ rules (pe.c:152-170) and mm_set_pe_thru (pe.c:45-74).
__device__ __forceinline__ void d_pair11(const AlParams &P, int max_gap_ref, const int ql0, const int ql1, AlReg &R0, AlReg &R1)
{
	const int rev0 = (R0.flags & ALR_REV) ? 1 : 0, rev1 = (R1.flags & ALR_REV) ? 1 : 0;
	const uint64_t key0 = (uint64_t)(uint32_t)R0.rid << 32 | (uint64_t)(uint32_t)(R0.rs << 1) | (uint64_t)(0 ^ rev0);
	const uint64_t key1 = (uint64_t)(uint32_t)R1.rid << 32 | (uint64_t)(uint32_t)(R1.rs << 1) | (uint64_t)(1 ^ rev1);
	int dp_thres = R0.dp_max + R1.dp_max - P.pe_bonus; if (dp_thres < 0) dp_thres = 0;
	const bool swap = key1 < key0;                                          // stable insertion sort of two keys
	const uint64_t keyA = swap ? key1 : key0, keyB = swap ? key0 : key1;
	const int revA = swap ? rev1 : rev0, revB = swap ? rev0 : rev1;
	const int32_t A_rid = swap ? R1.rid : R0.rid, A_re = swap ? R1.re : R0.re, B_rid = swap ? R0.rid : R1.rid, B_rs = swap ? R0.rs : R1.rs;
	bool paired = !(keyA & 1) && (keyB & 1) && revA == revB && B_rid == A_rid && B_rs - A_re <= max_gap_ref && R0.dp_max + R1.dp_max >= dp_thres;
	const long long score = (long long)((uint64_t)(uint32_t)(R0.dp_max + R1.dp_max) << 32 | (uint32_t)(R0.hash + R1.hash));
	if (paired && score > 0) {
		R0.flags |= ALR_PROPER; R1.flags |= ALR_PROPER;
		// (each hit is its own parent and already sam_pri; one pair score: n_sub = 1, no alternative)
		const int mapq_pe = R0.mapq > R1.mapq ? (int)R0.mapq : (int)R1.mapq;
		if ((int)R0.mapq < mapq_pe) R0.mapq = (uint32_t)(int)__fadd_rn(__fadd_rn(__fmul_rn(.2f, (float)R0.mapq), __fmul_rn(.8f, (float)mapq_pe)), .499f) & 0xffu;
		if ((int)R1.mapq < mapq_pe) R1.mapq = (uint32_t)(int)__fadd_rn(__fadd_rn(__fmul_rn(.2f, (float)R1.mapq), __fmul_rn(.8f, (float)mapq_pe)), .499f) & 0xffu;
		if (R0.mapq < 2) R0.mapq = 2;
		if (R1.mapq < 2) R1.mapq = 2;
	}
	// mm_set_pe_thru
	const int d1 = R0.rs - R1.rs, d2 = R0.re - R1.re;
	if (R0.rid == R1.rid && (R0.flags & ALR_REV) == (R1.flags & ALR_REV) && (d1 < 0 ? -d1 : d1) < 3 && (d2 < 0 ? -d2 : d2) < 3
	    && ((R0.qs == 0 && ql1 - R1.qe == 0) || (R1.qs == 0 && ql0 - R0.qe == 0))) { R0.flags |= ALR_PE_THRU; R1.flags |= ALR_PE_THRU; }
}

// ---------------------------------------------------------------------------------------------
// Per-fragment workspace layout.  With P_f = sum of n_u over earlier fragments and c_f = 4*n_u + 4 (room for up to three
// z-drop splits per hit; exceeding it is a counted error):
//   regs0      : AlReg   x n_u         at P_f                      (fragment-level hits)
//   mate regs  : AlReg   x c_f each    at 2*(4*P_f + 4*f) + s*c_f  (per-mate hits; spare room for z-drop splits)
//   scratch    : AlAnchor/ u64 / int / AlReg x c_f                 at B_f = 2*P_f + 4*f
//   seg_u      : u64     x n_u each    at 2*P_f + s*n_u
//   seg_a      : anchors of mate 0 then mate 1 inside the fragment's anchor range a_off[f]..
struct RegExt {                    // per-hit state carried from prep to finish, 32 bytes
	int32_t rs, qs, re, qe, rs0, re0, core_score;
	uint32_t job;                  // index of the left job; right job = job + 1
};
struct FragWs {
	AlReg *regs0, *mreg[2], *rtmp; AlAnchor *aux128, *seg_a[2]; uint64_t *aux64, *seg_u[2]; int *auxi; int cap;
};
struct WsBase {
	AlReg *regs0, *mregs, *rtmp; AlAnchor *aux128, *seg_a; uint64_t *aux64, *seg_u; int *auxi;
	const uint64_t *nu_off; const uint32_t *frag_nu; const uint64_t *a_off; uint32_t *reg_cnt /* per read */; uint32_t *seg_na /* per read */;
	RegExt *rext; uint32_t *seg_fast;   // per read: 1 = the read's single hit has its ungapped-core coordinates in rext already (k_regs fast path)
	// Room for the per-mate hits: 4 n + 4 per mate, n = hits kept by chain_post (cap2 / b2_off, set once k_regs_select has run; a
	// fragment it did not handle: n = its chains).  nullptr while the selection kernels run (they do not touch the per-mate arrays).
	const uint32_t *cap2; const uint64_t *b2_off;
};
__device__ __forceinline__ void d_frag_ws(const WsBase &W, uint32_t f, FragWs &o)
{
	const uint64_t Pf = W.nu_off[f]; const uint32_t nu = W.frag_nu[f]; const uint64_t B = 4 * Pf + 4ULL * f; const int c = 4 * (int)nu + 4;
	o.regs0 = W.regs0 + Pf;
	if (W.b2_off) { const int c2 = (int)W.cap2[f]; const uint64_t B2 = W.b2_off[f]; o.cap = c2; o.mreg[0] = W.mregs + 2 * B2; o.mreg[1] = o.mreg[0] + c2; o.rtmp = W.rtmp + B2; }
	else { o.cap = c; o.mreg[0] = nullptr; o.mreg[1] = nullptr; o.rtmp = nullptr; }
	o.aux128 = W.aux128 + B; o.aux64 = W.aux64 + B; o.auxi = W.auxi + 2 * B;
	o.seg_u[0] = W.seg_u + 2 * Pf; o.seg_u[1] = o.seg_u[0] + nu;
	o.seg_a[0] = W.seg_a + W.a_off[f]; o.seg_a[1] = nullptr;
}

// ---------------------------------------------------------------------------------------------
// KA for fragments with many chains (reads inside interspersed repeats: hundreds to thousands of chains, of which
// chain_post keeps the primaries and at most best_n secondaries).  One wavefront per fragment does mm_gen_regs's ordering
// (hit.c:52-88), mm_set_parent (hit.c:109-167) and mm_select_sub / mm_select_sub_multi (hit.c:238-255, pe.c:6-43) in one
// pass over the chains in score order, 64 at a time:
//   * keys (score << 32 | count) ^ hash are sorted in registers (<= 64 chains) or by a bitonic network in LDS;
//   * every lane tests its chain against the primaries found so far (kept in LDS: they are few, a primary must leave half
//     of an earlier one uncovered).  The first chain of the group that no primary masks becomes a primary itself and the
//     lanes behind it are tested again with the longer list; the lanes before it are final -- which is the order the
//     reference's loop takes its decisions in.  uncov_len is the part of the chain's query interval that no overlapping
//     primary covers: computed by an O(k^2) sweep without the reference's interval sort (same integer);
//   * a final lane applies its score / count to the parent's subsc / n_sub (max and a counter: order does not matter),
//     decides mm_select_sub(_multi) against its parent's record and takes the next output slot if it is kept.
// Only the kept hits are written (ws.regs0[0 .. n0), ids and parents renumbered as mm_sync_regs does); regs_n0[f] = n0 tells
// k_regs to start at mm_seg_gen.  Equal sort keys, more than PMAX primaries or more chains than the tile: regs_n0[f] stays
// unset and k_regs runs the reference's sequence on one lane (exact order among equal keys).
#define AL_REGS_PMAX 64               // primaries a fragment can have before k_regs takes it (short chains tiling a pair: dozens).  With AL_REGS_KCAP these two tables set the LDS of the pass (5.5 KB at 64 / 96 against 12.8 KB at 160 / 192: the one-wavefront kernels are then bound by wave slots, not LDS: regs 22.5 -> 20.0 ms on C4, no fragment more for the serial code)
#define AL_REGS_UNSET 0xffffffffu
#define AL_REGS_DONE 0xfffffffeu
#define AL_REGS_BAIL(v) ((v) >= 0xfffffff0u && (v) < AL_REGS_DONE)   // k_regs_select gave up: 0xfffffff1 equal sort keys (> 65535 chains), 0xfffffff2 too many primaries, 0xfffffff3 parent slot reused (in-place compaction of the reference)
#define AL_REGS_KCAP 96               // kept hits whose records k_regs_select holds for the in-place compaction of the reference's selection (primaries + best_n)
struct RegsSelKept { int32_t score[AL_REGS_KCAP], ridrev[AL_REGS_KCAP], rs[AL_REGS_KCAP], re[AL_REGS_KCAP]; uint16_t qs[AL_REGS_KCAP], qe[AL_REGS_KCAP]; };   // (query coordinates are below 2^16: reads of at most 32768 bases)
struct RegsSelShared {
	int32_t qs[AL_REGS_PMAX], qe[AL_REGS_PMAX], score[AL_REGS_PMAX], cnt[AL_REGS_PMAX], as[AL_REGS_PMAX], rs[AL_REGS_PMAX], re[AL_REGS_PMAX], ridrev[AL_REGS_PMAX];
	int32_t subsc[AL_REGS_PMAX], nsub[AL_REGS_PMAX], slot[AL_REGS_PMAX], orig[AL_REGS_PMAX]; uint32_t hash[AL_REGS_PMAX];   // slot: rank among the kept hits; orig: position in score order
};
// PHASE 1 (CAP > 0): the sort only -- keys, order and the per-position tables are left in the fragment's global work area, and
// k_regs_select<-2> (one wavefront, no sort tile: a dozen blocks per CU instead of the one or two the tile allows) makes the pass.
template <int CAP, int PHASE = 0>      // CAP > 0: sort tile in LDS; 0: at most 64 chains, registers; -1: sort keys in the fragment's global work area (any count); -2: keys sorted already (PHASE 1 ran), the pass only
__global__ void __launch_bounds__(CAP == 0 || CAP == 256 || CAP == -2 ? 64 : CAP < 0 ? 1024 : (PHASE == 1 && CAP >= 8192) ? 512 : 256)
k_regs_select(const AlAnchor *__restrict__ chained, const uint64_t *__restrict__ u_all, const uint32_t *__restrict__ uo_all, const uint32_t *__restrict__ frag_first,
              const uint32_t *__restrict__ rd_len, const uint32_t *__restrict__ frag_hash, WsBase W, const uint32_t *__restrict__ list, int n_list,
              AlParams P, uint32_t *__restrict__ regs_n0)
{
	// (measured twice in round 6 with the 8192-chain sort tile's chain numbers in the work area instead of LDS, for two blocks per CU: 3.0 -> 4.0 - 4.3 ms: that class is not short of wavefronts)
	constexpr bool PK1 = false;
	constexpr bool IDX_GLOBAL = CAP <= 0;
	__shared__ uint64_t skey_l[CAP > 0 ? CAP : 1];
	__shared__ uint16_t sidx_l[CAP > 0 && !IDX_GLOBAL ? CAP : 1];
	__shared__ RegsSelShared S; __shared__ RegsSelKept K;
	constexpr int NT = CAP == 0 || CAP == 256 || CAP == -2 ? 64 : CAP < 0 ? 1024 : (PHASE == 1 && CAP >= 8192) ? 512 : 256;   // (the sort-only form of the largest tile: one block per CU, all of its wavefronts)   // (65 ... 256 chains: one wavefront sorts and makes the pass -- four times the blocks per CU of the 256-thread form, whose other three wavefronts only sort)
	                // all threads sort (the sort in global memory, any count: 1024 of them); the first wavefront makes the pass
	const int tid = threadIdx.x, lane = tid & 63;
	if ((int)blockIdx.x >= n_list) return;
	const uint32_t f = list[blockIdx.x];
	const uint32_t r0 = frag_first[f], n_segs = frag_first[f + 1] - r0;
	const int n_u = (int)W.frag_nu[f];
	if ((CAP >= 0 && n_u > (CAP > 0 ? CAP : 64)) || n_u < 2) return;        // regs_n0[f] pre-set to AL_REGS_UNSET
	if (CAP == -2 && regs_n0[f] != AL_REGS_UNSET) return;                   // the sort kernel gave up on it
	FragWs ws; d_frag_ws(W, f, ws);
	uint64_t *const skey = CAP > 0 ? skey_l : ws.aux64;                      // capacity 4 n_u + 4 >= the next power of two
	typedef typename std::conditional<!IDX_GLOBAL, uint16_t, uint32_t>::type IdxT;
	IdxT *const sidx = !IDX_GLOBAL ? (IdxT *)sidx_l : (IdxT *)(ws.auxi + (4 * n_u + 4));
	const int ql0 = (int)rd_len[r0], ql1 = n_segs > 1 ? (int)rd_len[r0 + 1] : 0, qlen = ql0 + ql1;
	const AlAnchor *a = chained + W.a_off[f]; const uint64_t *u = u_all + W.a_off[f] + f;
	const uint32_t *const as_arr = uo_all + W.a_off[f] + f;                  // first anchor of chain c: carried by the chain list (the chaining kernels write every chain at its segment's place)
	const uint32_t fhash = frag_hash[f];
	int max_gap_ref;
	if (P.max_gap_ref > 0) max_gap_ref = P.max_gap_ref;
	else if (P.max_frag_len > 0) { max_gap_ref = P.max_frag_len - qlen; if (max_gap_ref < P.max_gap) max_gap_ref = P.max_gap; }
	else max_gap_ref = P.max_gap;
	int4 *const GR = (int4 *)ws.aux128;                                      // two words of 16 bytes per chain, by chain number (capacity 4 n_u + 4)
	// ---- keys: (u ^ hash of the chain's first anchor) ----
	uint64_t key_r = 0; int idx_r = 0;                                      // n_u <= 64: lane's own entry
	if (CAP != -2) {
		if (CAP == 0) {
			if (lane < n_u) {
				const uint64_t uc = u[lane]; const AlAnchor fa = a[as_arr[lane]];
				key_r = uc ^ (uint32_t)d_hash64((d_hash64(fa.x) + d_hash64(fa.y)) ^ fhash); idx_r = lane;
			}
		} else {                                                             // keys: every thread of the block; four chains per round, the two dependent loads of each round in flight together (clamped indices: no branch around a load)
			for (int c0 = tid; c0 < n_u; c0 += 4 * NT) {
				int cc[4]; uint32_t as4[4]; uint64_t u4[4]; AlAnchor f4[4], l4[4];
#pragma unroll
				for (int q = 0; q < 4; ++q) { cc[q] = c0 + q * NT < n_u ? c0 + q * NT : n_u - 1; as4[q] = as_arr[cc[q]]; u4[q] = u[cc[q]]; }
#pragma unroll
				for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(as4[q]), "+v"(u4[q]));
#pragma unroll
				for (int q = 0; q < 4; ++q) { f4[q] = a[as4[q]]; const uint32_t cn = (uint32_t)u4[q]; l4[q] = a[as4[q] + (cn ? cn - 1u : 0u)]; }
#pragma unroll
				for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(f4[q].x), "+v"(f4[q].y), "+v"(l4[q].x), "+v"(l4[q].y));
#pragma unroll
				for (int q = 0; q < 4; ++q) if (c0 + q * NT < n_u) {
					skey[cc[q]] = u4[q] ^ (uint32_t)d_hash64((d_hash64(f4[q].x) + d_hash64(f4[q].y)) ^ fhash); if (!PK1) sidx[cc[q]] = (IdxT)cc[q];
					// (round 6) the chain's record for the pass -- query / reference interval, contig + strand, count, first anchor -- is made HERE, in chain order, where
					// the chains' anchors are read front to back; the pass fetches the record of the chain at a sorted position (32 bytes out of a table that stays in L2)
					// instead of a second, randomly ordered round of two 16-byte anchor reads per chain after the sort (11 GB per launch of the 257 ... 1024-chain class)
					const AlAnchor fa = f4[q], la = l4[q]; const int cnt = (int)(uint32_t)u4[q];
					const int q_span = (int)(fa.y >> 32 & 0xff), rev = (int)(fa.x >> 63), rid = (int)(fa.x << 1 >> 33);
					int4 g;
					g.z = (int32_t)fa.x + 1 > q_span ? (int32_t)fa.x + 1 - q_span : 0; g.w = (int32_t)la.x + 1;
					if (!rev) { g.x = (int32_t)fa.y + 1 - q_span; g.y = (int32_t)la.y + 1; }
					else { g.x = qlen - ((int32_t)la.y + 1); g.y = qlen - ((int32_t)fa.y + 1 - q_span); }
					GR[2 * cc[q]] = g; GR[2 * cc[q] + 1] = make_int4(rid << 1 | rev, cnt, (int)as4[q], 0);
				}
			}
		}
	}
	bool tie = false;
	if (CAP != 0 && CAP != -2) {   // descending sort of (key, chain)
		// (round 6) LDS tiles of up to 16 keys per thread: the register network of the anchor sorts (al_dev_net.h) on ONE word per chain,
		// (~key & 2^48 - 1) << 16 | chain -- a key is (score << 32 | count) ^ a 32-bit hash, below 2^48 while scores stay below 2^16, which is checked --
		// ascending, i.e. descending by key.  The bitonic steps through LDS with a barrier each were most of these kernels (the 8192-chain tile:
		// 3.9 of 4.6 ms, a block per CU and nothing else on the main stream meanwhile).  Equal keys end up next to each other either way and are
		// put in the reference's order below; a key of 2^48 or more sends the block to the LDS steps.
		constexpr int PERN = CAP > 0 ? CAP / NT : 0;
		bool net_done = false;
		if constexpr (CAP > 0 && PERN >= 1 && PERN <= 16 && PERN * NT == CAP) {
			__shared__ int s_big;
			if (tid == 0) s_big = 0;
			__syncthreads();
			const uint64_t M48 = (1ULL << 48) - 1ULL;
			uint64_t kn[PERN]; bool big = false;
#pragma unroll
			for (int j = 0; j < PERN; ++j) {
				const int e = tid + j * NT;
				uint64_t v = UINT64_MAX;
				if (e < n_u) { const uint64_t key = skey[e]; big = big || (key >> 48) != 0; v = ((~key) & M48) << 16 | (uint64_t)(uint32_t)e; }
				kn[j] = v;
			}
			if (big) s_big = 1;
			__syncthreads();
			if (!s_big && !((P.dbg >> 15) & 1)) {
				d_bt_levels<PERN, NT, 2>(kn, skey, tid);                          // (its exchange tile is the key tile: every thread holds its keys in registers by now)
				__syncthreads();
#pragma unroll
				for (int r = 0; r < PERN; ++r) {
					const int e = tid * PERN + r; const uint64_t v = kn[r];
					skey[e] = v == UINT64_MAX ? 0ULL : ((~(v >> 16)) & M48);
					if (e < n_u) sidx[e] = (IdxT)(v & 0xffffu);                   // (the pads sort last: positions n_u and up, which nothing reads -- and which the work area of a small fragment does not have)
				}
				__syncthreads();
				net_done = true;
			}
		}
		int npow2 = 1; while (npow2 < n_u) npow2 <<= 1;
		if (!net_done) {
		if (PK1) { for (int c = tid; c < n_u; c += NT) sidx[c] = (IdxT)c; }   // (the key phase left them out)
		for (int c = n_u + tid; c < npow2; c += NT) { skey[c] = 0; sidx[c] = (IdxT)~0u; }   // keys are > 0 (a chain's score, in the high word): padding sorts last
		__syncthreads();
		// A thread takes comparators, not elements: pair p of a step works on i = p with a zero inserted at bit log2(j) and on i | j, so every
		// thread is busy, and the loads of all its pairs (LDS tiles: a compile-time count) are in flight before the first comparison -- one LDS
		// round trip per step instead of two per pair (the 8192-key tile: 3.9 of the kernel's 6.7 ms were these steps).
		constexpr int PP = CAP > 0 ? (CAP / 2 + NT - 1) / NT : 1;
		const int half = npow2 >> 1;
		for (int kk = 2; kk <= (((P.dbg >> 15) & 1) ? 0 : npow2); kk <<= 1)   // (AL_DBG bit 15: timing experiment, no sort)
			for (int j = kk >> 1; j > 0; j >>= 1) {
				if (CAP > 0) {
					uint64_t kx[PP], ky[PP]; IdxT ix[PP], iy[PP];
#pragma unroll
					for (int q = 0; q < PP; ++q) {
						const int p_ = tid + q * NT, i = ((p_ & ~(j - 1)) << 1) | (p_ & (j - 1));
						if (p_ < half) { kx[q] = skey[i]; ky[q] = skey[i | j]; ix[q] = sidx[i]; iy[q] = sidx[i | j]; }
					}
#pragma unroll
					for (int q = 0; q < PP; ++q) {
						const int p_ = tid + q * NT, i = ((p_ & ~(j - 1)) << 1) | (p_ & (j - 1));
						if (p_ < half && (kx[q] < ky[q]) == ((i & kk) == 0)) { skey[i] = ky[q]; skey[i | j] = kx[q]; sidx[i] = iy[q]; sidx[i | j] = ix[q]; }
					}
				} else {
					for (int p_ = tid; p_ < half; p_ += NT) {
						const int i = ((p_ & ~(j - 1)) << 1) | (p_ & (j - 1)), ixj = i | j;
						const uint64_t x = skey[i], y = skey[ixj];
						if ((x < y) == ((i & kk) == 0)) { skey[i] = y; skey[ixj] = x; const IdxT t = sidx[i]; sidx[i] = sidx[ixj]; sidx[ixj] = t; }
					}
				}
				__syncthreads();
			}
		}
		__shared__ int s_tie; __shared__ uint16_t s_rs[AL_RS_SCRATCH / 2];
		if (tid == 0) s_tie = 0;
		__syncthreads();
		for (int i = tid; i + 1 < n_u; i += NT) if (skey[i] == skey[i + 1]) s_tie = 1;
		__syncthreads();
		if (s_tie && ((P.dbg >> 17) & 1)) { if (tid == 0) regs_n0[f] = 0xfffffff1u; return; }
		if (s_tie) {
			// Equal keys (a minimizer the sketch emitted twice makes two identical chains): their order is what the reference's unstable
			// radix sort (ksort.h:116-151, more than 64 entries here) leaves.  Rare: the keys go back into chain order and one lane
			// restates that sort on them (ascending), the block reverses the result (hit.c:76).
			if (n_u > 65535) { if (tid == 0) regs_n0[f] = 0xfffffff1u; return; }
			for (int c = tid; c < n_u; c += NT) { const AlAnchor fa = a[as_arr[c]]; skey[c] = u[c] ^ (uint32_t)d_hash64((d_hash64(fa.x) + d_hash64(fa.y)) ^ fhash); sidx[c] = (IdxT)c; }
			__syncthreads();
			if (tid < 64) {                                                       // the first wavefront: counts and small buckets in parallel, the permutation by lane 0
				struct KI { uint64_t k; IdxT i; };
				struct { typedef KI E; uint64_t *k; IdxT *i;
				         __device__ __forceinline__ uint64_t keyof(const KI &e) const { return e.k; }
				         __device__ __forceinline__ uint64_t key(int j) const { return k[j]; }
				         __device__ __forceinline__ KI get(int j) const { return KI{k[j], i[j]}; }
				         __device__ __forceinline__ void set(int j, const KI &e) { k[j] = e.k; i[j] = e.i; } } acc{skey, sidx};
				(void)d_rs_sort_wave(acc, n_u, s_rs, tid);
			}
			__syncthreads();
			for (int i = tid; i < n_u / 2; i += NT) {
				const uint64_t tk = skey[i]; skey[i] = skey[n_u - 1 - i]; skey[n_u - 1 - i] = tk;
				const IdxT ti = sidx[i]; sidx[i] = sidx[n_u - 1 - i]; sidx[n_u - 1 - i] = ti;
			}
			__syncthreads();
		}
		if (PHASE == 1) for (int p = tid; p < n_u; p += NT) { ws.aux64[p] = skey[p]; if (!PK1) ((uint32_t *)(ws.auxi + (4 * n_u + 4)))[p] = (uint32_t)sidx[p]; }   // keys and order for the pass kernel (PK1: the order is there already)
		if (PHASE == 1) return;
		__syncthreads();
		if (tid >= 64) return;
	} else if (CAP == 0) {         // rank sort in registers (descending); equal keys: the stable ascending insertion sort (ksort.h:149), reversed, puts the later chain first
		int rank = 0; const int klo = (int)(uint32_t)key_r, khi = (int)(uint32_t)(key_r >> 32); bool tie16 = false;
		for (int j = 0; j < n_u; ++j) {
			const uint64_t kj = (uint64_t)(uint32_t)__shfl(klo, j) | (uint64_t)(uint32_t)__shfl(khi, j) << 32;
			if (lane < n_u) { rank += (kj > key_r || (kj == key_r && j > lane)) ? 1 : 0; tie16 = tie16 || (kj == key_r && j != lane); }
		}
		if (((P.dbg >> 16) & 1) && __ballot(tie16)) { if (lane == 0) regs_n0[f] = 0xfffffff1u; return; }
		// lane takes the entry whose rank is its lane number
		int src = 0;
		for (int j = 0; j < n_u; ++j) { const int rj = __shfl(rank, j); if (rj == lane) src = j; }
		const int klo2 = __shfl(klo, src), khi2 = __shfl(khi, src), id2 = __shfl(idx_r, src);
		key_r = (uint64_t)(uint32_t)klo2 | (uint64_t)(uint32_t)khi2 << 32; idx_r = id2;
	}
	// ---- one pass in score order ----
	const float mask_level = P.mask_level;
	const int min_diff = P.k * 2, best_n = P.best_n;
	const int max_dist = n_segs == 2 ? ql0 + ql1 + max_gap_ref : 0;
	int k = 0, slot_base = 0, n_2nd = 0; bool overflow = false;
	const unsigned long long below = (1ULL << lane) - 1ULL;
	__shared__ uint32_t s_cov[64];                                           // query positions covered by the primaries so far (fragments of up to 2048 bases)
	const bool use_cov = qlen <= 2048;
	if (qlen > 65535) { if (lane == 0) regs_n0[f] = 0xfffffff3u; return; }       // (the kept-hit records hold query coordinates in 16 bits: a pair of two 32768-base reads goes to the serial code)
	s_cov[lane] = 0;
	__threadfence_block();
	for (int p0 = 0; p0 < n_u && !overflow; p0 += 64) {
		const int p = p0 + lane; const bool v = p < n_u;
		uint64_t key = 0; int c = 0;
		if (v) { if (CAP != 0) { key = skey[p]; c = (int)sidx[p]; } else { key = key_r; c = idx_r; } }
		const int score = (int)(key >> 32); const uint32_t hsh = (uint32_t)key;
		int cnt = 0, as = 0, rs = 0, re = 0, qs = 0, qe = 0, rid = 0, rev = 0;
		if (v && CAP != 0) {
			const int4 g = GR[2 * c], h = GR[2 * c + 1];
			qs = g.x; qe = g.y; rs = g.z; re = g.w; rid = h.x >> 1; rev = h.x & 1; cnt = h.y; as = h.z;
		} else if (v) {
			cnt = (int)(uint32_t)u[c]; as = (int)as_arr[c];
			const AlAnchor fa = a[as], la = a[as + cnt - 1];
			const int q_span = (int)(fa.y >> 32 & 0xff);
			rev = (int)(fa.x >> 63); rid = (int)(fa.x << 1 >> 33);
			rs = (int32_t)fa.x + 1 > q_span ? (int32_t)fa.x + 1 - q_span : 0; re = (int32_t)la.x + 1;
			if (!rev) { qs = (int32_t)fa.y + 1 - q_span; qe = (int32_t)la.y + 1; }
			else { qs = qlen - ((int32_t)la.y + 1); qe = qlen - ((int32_t)fa.y + 1 - q_span); }
		}
		bool pending = v;
		while (__ballot(pending)) {
			int pj = -1;                                                     // masking primary
			if (pending && use_cov) {
				// uncov_len (hit.c:133-141): the part of [qs, qe) outside the union of the clipped overlapping intervals = outside the union
				// of ALL primaries (the others do not reach into [qs, qe)): a bitmap of the query positions the primaries cover, looked at
				// once the first overlapping primary turns up
				int uncov = -1;
				for (int j = 0; j < k; ++j) {
					const int sj = S.qs[j], ej = S.qe[j];
					if (ej <= qs || sj >= qe) continue;
					if (uncov < 0) {
						int covered = 0;
						for (int w = qs >> 5; w <= (qe - 1) >> 5; ++w) {
							const int lo = qs > (w << 5) ? qs - (w << 5) : 0, hi = qe < ((w + 1) << 5) ? qe - (w << 5) : 32;
							const uint32_t m = (hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
							covered += __popc(s_cov[w] & m);
						}
						uncov = (qe - qs) - covered;
					}
					const int mn = ej - sj < qe - qs ? ej - sj : qe - qs, mx = ej - sj > qe - qs ? ej - sj : qe - qs;
					const int ol = qs < sj ? (qe < sj ? 0 : qe < ej ? qe - sj : ej - sj) : (ej < qs ? 0 : ej < qe ? ej - qs : qe - qs);
					if (__fsub_rn(al_fdiv((float)ol, (float)mn), al_fdiv((float)uncov, (float)mx)) > mask_level) { pj = j; break; }
				}
			} else if (pending) {
				int n_cov = 0;
				for (int j = 0; j < k; ++j) { const int sj = S.qs[j], ej = S.qe[j]; if (!(ej <= qs || sj >= qe)) ++n_cov; }
				if (n_cov > 0) {
					int uncov = 0;
					int x = qs;
					for (;;) {
						int best = 0x7fffffff;
						for (int j = 0; j < k; ++j) { int sj = S.qs[j], ej = S.qe[j]; if (ej <= qs || sj >= qe) continue; if (sj < qs) sj = qs; if (ej > qe) ej = qe; if (ej > x && sj < best) best = sj; }
						if (best == 0x7fffffff) break;
						if (best > x) { uncov += best - x; x = best; }
						bool grew = true;
						while (grew) { grew = false; for (int j = 0; j < k; ++j) { int sj = S.qs[j], ej = S.qe[j]; if (ej <= qs || sj >= qe) continue; if (sj < qs) sj = qs; if (ej > qe) ej = qe; if (sj <= x && ej > x) { x = ej; grew = true; } } }
					}
					if (qe > x) uncov += qe - x;
					for (int j = 0; j < k; ++j) {
						const int sj = S.qs[j], ej = S.qe[j];
						if (ej <= qs || sj >= qe) continue;
						const int mn = ej - sj < qe - qs ? ej - sj : qe - qs, mx = ej - sj > qe - qs ? ej - sj : qe - qs;
						const int ol = qs < sj ? (qe < sj ? 0 : qe < ej ? qe - sj : ej - sj) : (ej < qs ? 0 : ej < qe ? ej - qs : qe - qs);
						if (__fsub_rn(al_fdiv((float)ol, (float)mn), al_fdiv((float)uncov, (float)mx)) > mask_level) { pj = j; break; }
					}
				}
			}
			const unsigned long long um = __ballot(pending && pj < 0);
			const int first = um ? __ffsll((long long)um) - 1 : 64;
			const bool fin = pending && lane < first;                        // masked, final
			// mm_select_sub(_multi) compact the hit array in place while they still look a hit's parent up by its OLD index P (hit.c:238-255,
			// pe.c:6-43): once more hits have been kept than P, slot P holds the kept hit of rank P, and the tests are made against THAT
			// record -- unless nothing in front of the parent was dropped (its rank is P: it was copied onto itself).  K[] keeps the fields
			// those tests read for every kept hit, by rank.  k_now (hits kept before this one) is known exactly except while the kept count
			// crosses P inside this group of lanes: then the lanes are settled one after the other.
			const unsigned long long fm = __ballot(fin);
			int P_o = 0; bool moved = false;
			if (fin) {
				atomicMax(&S.subsc[pj], score);
				if (cnt >= S.cnt[pj]) atomicAdd(&S.nsub[pj], 1);
				P_o = S.orig[pj]; moved = S.slot[pj] != P_o;
			}
			auto decide = [&](int psc, int pqs, int pqe, int pridrev, int prs, int pre) -> bool {
				if (!(P.pri_ratio > 0.0f)) return true;                          // selection switched off: everything is kept
				if (n_segs <= 1) {                                              // mm_select_sub
					if ((float)score >= __fmul_rn((float)psc, P.pri_ratio) || score + min_diff >= psc)
						return !(qs == pqs && qe == pqe && (rid << 1 | rev) == pridrev && rs == prs && re == pre);
					return false;
				}
				if (score + min_diff >= psc) return true;                       // mm_select_sub_multi
				const int prev = pridrev & 1, prid = pridrev >> 1;
				if (prev == rev && prid == rid && re - prs < max_dist && pre - rs < max_dist) return (float)score >= __fmul_rn((float)psc, 0.2f);
				const int is_par_both = (n_segs == 2 && pqs < ql0 && pqe > ql0);
				const int is_chi_both = (n_segs == 2 && qs < ql0 && qe > ql0);
				if (is_chi_both || is_chi_both == is_par_both) return (float)score >= __fmul_rn((float)psc, P.pri_ratio);
				return (float)score >= __fmul_rn((float)psc, 0.7f);
			};
			auto decide_at = [&](bool aliased) -> bool {                       // against the slot's present content
				if (aliased) return decide(K.score[P_o], (int)K.qs[P_o], (int)K.qe[P_o], K.ridrev[P_o], K.rs[P_o], K.re[P_o]);
				return decide(S.score[pj], S.qs[pj], S.qe[pj], S.ridrev[pj], S.rs[pj], S.re[pj]);
			};
			auto keep_hit = [&](int rank) {
				AlReg R; d_reg_clear(&R);
				R.id = rank; R.parent = S.slot[pj]; R.score = R.score0 = score; R.hash = hsh; R.cnt = cnt; R.as = as;
				d_reg_set_coor(&R, qlen, a);
				ws.regs0[R.id] = R;
				if (rank < AL_REGS_KCAP) { K.score[rank] = score; K.qs[rank] = (uint16_t)qs; K.qe[rank] = (uint16_t)qe; K.ridrev[rank] = rid << 1 | rev; K.rs[rank] = rs; K.re[rank] = re; }
			};
			const int n_before = (int)__popcll(fm & below);
			const bool full = (P.pri_ratio > 0.0f) && n_2nd >= best_n;           // best_n secondaries have qualified: every further one is dropped whatever its test says
			const bool amb = !full && fin && moved && (P.pri_ratio > 0.0f) && slot_base <= P_o && P_o < slot_base + n_before;   // k_now may or may not have passed P
			if (__ballot(!full && fin && moved && slot_base + n_before > P_o && P_o >= AL_REGS_KCAP)) { if (lane == 0) regs_n0[f] = 0xfffffff3u; return; }   // beyond the kept records: serial code
			if (full) { /* nothing to decide */ }
			else if (!__ballot(amb)) {
				bool qual = false;
				if (fin) qual = decide_at(moved && slot_base > P_o);              // (not ambiguous: P < slot_base <= k_now, or k_now <= P)
				const unsigned long long qm = __ballot(qual);
				const bool kept = qual && (!(P.pri_ratio > 0.0f) || n_2nd + __popcll(qm & below) < best_n);
				const unsigned long long km = __ballot(kept);
				if (kept) keep_hit(slot_base + (int)__popcll(km & below));
				n_2nd += __popcll(qm); slot_base += __popcll(km);
			} else {
				for (unsigned long long rem = fm; rem; rem &= rem - 1) {          // in order; slot_base and n_2nd are the running counts
					const int l = __ffsll((long long)rem) - 1;
					int q = 0, kp = 0;
					if (lane == l) {
						q = decide_at(moved && slot_base > P_o) ? 1 : 0;
						kp = q && n_2nd < best_n ? 1 : 0;
						if (kp) keep_hit(slot_base);
					}
					__threadfence_block();
					n_2nd += __shfl(q, l); slot_base += __shfl(kp, l);
				}
			}
			__threadfence_block();
			if (first < 64) {
				if (k >= AL_REGS_PMAX) overflow = true;
				else if (lane == first) {
					S.qs[k] = qs; S.qe[k] = qe; S.score[k] = score; S.cnt[k] = cnt; S.as[k] = as; S.rs[k] = rs; S.re[k] = re; S.ridrev[k] = rid << 1 | rev;
					S.subsc[k] = 0; S.nsub[k] = 0; S.slot[k] = slot_base; S.orig[k] = p; S.hash[k] = hsh;
					if (slot_base < AL_REGS_KCAP) { K.score[slot_base] = score; K.qs[slot_base] = (uint16_t)qs; K.qe[slot_base] = (uint16_t)qe; K.ridrev[slot_base] = rid << 1 | rev; K.rs[slot_base] = rs; K.re[slot_base] = re; }
					if (use_cov && qe > qs) for (int w = qs >> 5; w <= (qe - 1) >> 5; ++w) {
						const int lo = qs > (w << 5) ? qs - (w << 5) : 0, hi = qe < ((w + 1) << 5) ? qe - (w << 5) : 32;
						s_cov[w] |= (hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
					}
				}
				if (!overflow) { ++k; ++slot_base; }
			}
			__threadfence_block();                                           // one wavefront: the new primary's LDS record before the next round reads it
			pending = pending && lane > first;
			if (overflow) break;
		}
	}
	if (overflow) { if (lane == 0) regs_n0[f] = 0xfffffff2u; return; }
	const int n0 = slot_base;
	for (int j = lane; j < k; j += 64) {                                     // the primaries, with their final subsc / n_sub
		AlReg R; d_reg_clear(&R);
		R.id = S.slot[j]; R.parent = R.id; R.score = R.score0 = S.score[j]; R.hash = S.hash[j]; R.cnt = S.cnt[j]; R.as = S.as[j];
		R.subsc = S.subsc[j]; R.n_sub = S.nsub[j];
		d_reg_set_coor(&R, qlen, a);
		if (n0 != n_u && j == 0) R.flags |= ALR_SAM_PRI;                    // mm_sync_regs -> mm_set_sam_pri runs only when something was dropped
		ws.regs0[R.id] = R;
	}
	if (lane == 0) regs_n0[f] = (uint32_t)n0;
}
template __global__ void k_regs_select<0>(const AlAnchor *, const uint64_t *, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, WsBase, const uint32_t *, int, AlParams, uint32_t *);
template __global__ void k_regs_select<256>(const AlAnchor *, const uint64_t *, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, WsBase, const uint32_t *, int, AlParams, uint32_t *);
template __global__ void k_regs_select<1024>(const AlAnchor *, const uint64_t *, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, WsBase, const uint32_t *, int, AlParams, uint32_t *);
template __global__ void k_regs_select<2048>(const AlAnchor *, const uint64_t *, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, WsBase, const uint32_t *, int, AlParams, uint32_t *);
template __global__ void k_regs_select<4096>(const AlAnchor *, const uint64_t *, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, WsBase, const uint32_t *, int, AlParams, uint32_t *);
template __global__ void k_regs_select<8192>(const AlAnchor *, const uint64_t *, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, WsBase, const uint32_t *, int, AlParams, uint32_t *);
template __global__ void k_regs_select<1024, 1>(const AlAnchor *, const uint64_t *, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, WsBase, const uint32_t *, int, AlParams, uint32_t *);
template __global__ void k_regs_select<8192, 1>(const AlAnchor *, const uint64_t *, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, WsBase, const uint32_t *, int, AlParams, uint32_t *);
template __global__ void k_regs_select<-2>(const AlAnchor *, const uint64_t *, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, WsBase, const uint32_t *, int, AlParams, uint32_t *);
template __global__ void k_regs_select<-1>(const AlAnchor *, const uint64_t *, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, WsBase, const uint32_t *, int, AlParams, uint32_t *);

__global__ void __launch_bounds__(256)
k_regs_cap2(const uint32_t *__restrict__ frag_nu, const uint32_t *__restrict__ regs_n0, int n_frag, uint32_t *__restrict__ cap2)
{
	const int f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f > n_frag) return;
	if (f == n_frag) { cap2[f] = 0; return; }
	const uint32_t n0 = regs_n0 ? regs_n0[f] : AL_REGS_UNSET;
	cap2[f] = 4u * ((n0 == AL_REGS_UNSET || AL_REGS_BAIL(n0) || n0 == AL_REGS_DONE) ? frag_nu[f] : n0) + 4u;
}

static size_t al_regs_heavy_lds(int RC, int AC) { return (size_t)3 * RC * sizeof(AlReg) + ((size_t)2 * AC + 2 * ((size_t)RC + AL_RS_SCRATCH / 16 + 2)) * sizeof(AlAnchor) + (size_t)4 * RC * 8 + ((size_t)2 * (2 * RC + 4) + RC + 1 + RC) * 4 + 8 * 4 + 64; }   // (the scratch of the per-mate code twice: the mates run on a lane each)

// one mate's share of mm_seg_gen's tail (hit.c:395-408 + map.c:401 + align.c:873): its hits from its chains, parents, anchors squeezed
__device__ __forceinline__ bool d_regs_mate(const AlParams &P, const uint32_t hash, const int ql, const int n, const uint64_t *su, AlAnchor *sa, AlReg *mreg,
                                            AlAnchor *aux128, uint64_t *aux64, int *auxi, const uint32_t seg_flag, int &na_squeezed)
{
	const bool tie = d_gen_regs(hash, ql, n, su, sa, mreg, aux128);
	for (int i = 0; i < n; ++i) mreg[i].flags |= ALR_SEG_SPLIT | seg_flag;
	d_set_parent(P.mask_level, n, mreg, P.a * 2 + P.b, aux64, auxi);                              // map.c:401
	na_squeezed = d_squeeze_a(n, mreg, sa, aux64);                                                // align.c:873
	return tie;
}

// What follows chain_post in mm_map_frag for the kept hits regs0[0 .. n0): single end -- the hits become the mate's hits and
// their anchors are squeezed (align.c:873); paired end -- mm_seg_gen (hit.c:356-410): anchors split per mate (y rebased), per-mate
// mm_gen_regs + mm_set_parent + squeeze.  a[] is indexed by regs0[i].as; sa0 receives mate 0's anchors, mate 1's follow at
// sa0 + sna0.  Serial code: run by one lane on global memory (k_regs) or on LDS copies (k_regs_heavy).  Returns the sort-tie flag.
__device__ __forceinline__ bool d_regs_tail(const AlParams &P, const uint32_t hash, const int n_segs, const int ql0, const int ql1, const int n0,
                                            AlReg *regs0, const AlAnchor *a, const uint32_t tot_se, AlReg *mreg0, AlReg *mreg1, uint64_t *su0, uint64_t *su1,
                                            AlAnchor *sa0, AlAnchor *aux128, uint64_t *aux64, int *auxi, uint32_t &cnt0, uint32_t &cnt1, uint32_t &sna0, uint32_t &sna1)
{
	bool tie = false; const int qlen_sum = ql0 + ql1;
	if (n_segs == 1) {
		for (int i = 0; i < n0; ++i) mreg0[i] = regs0[i];
		for (uint32_t i = 0; i < tot_se; ++i) sa0[i] = a[i];
		const int na = d_squeeze_a(n0, mreg0, sa0, aux64);                                            // mm_align_skeleton, align.c:873
		cnt0 = (uint32_t)n0; sna0 = (uint32_t)na;
	} else {
		// mm_seg_gen, hit.c:356-410
		uint32_t na0 = 0, na1 = 0, nus0 = 0, nus1 = 0;
		for (int i = 0; i < n0; ++i) {
			const AlReg *r = &regs0[i];
			uint32_t c1 = 0;
			for (int j = 0; j < r->cnt; ++j) c1 += (uint32_t)((a[r->as + j].y & AL_SEED_SEG_MASK) >> AL_SEED_SEG_SHIFT) & 1u;
			const uint32_t c0 = (uint32_t)r->cnt - c1;
			su0[i] = (uint64_t)(uint32_t)r->score << 32 | c0; su1[i] = (uint64_t)(uint32_t)r->score << 32 | c1;
			na0 += c0; na1 += c1;
		}
		AlAnchor *const sa1 = sa0 + na0;
		for (int i = 0; i < n0; ++i) { if ((int32_t)su0[i] != 0) su0[nus0++] = su0[i]; if ((int32_t)su1[i] != 0) su1[nus1++] = su1[i]; }
		uint32_t w0 = 0, w1 = 0;
		for (int i = 0; i < n0; ++i) {
			const AlReg *r = &regs0[i];
			for (int j = 0; j < r->cnt; ++j) {
				AlAnchor a1 = a[r->as + j]; const bool s1 = ((a1.y & AL_SEED_SEG_MASK) >> AL_SEED_SEG_SHIFT) & 1;
				const int qls = s1 ? ql1 : ql0, acc = s1 ? ql0 : 0;
				a1.y -= (a1.x >> 63) ? (uint64_t)(qlen_sum - (qls + acc)) : (uint64_t)acc;
				if (s1) sa1[w1++] = a1; else sa0[w0++] = a1;
			}
		}
		int na_sq = 0;
		tie = d_regs_mate(P, hash, ql0, (int)nus0, su0, sa0, mreg0, aux128, aux64, auxi, 0u, na_sq) || tie; cnt0 = nus0;
		tie = d_regs_mate(P, hash, ql1, (int)nus1, su1, sa1, mreg1, aux128, aux64, auxi, 1u << 8, na_sq) || tie; cnt1 = nus1; sna1 = (uint32_t)na_sq;
		sna0 = na0;   // seg_a[1] starts na0 anchors after seg_a[0] (kept un-squeezed size for addressing)
	}
	return tie;
}

// The tail of KA (d_regs_tail) for fragments that keep many hits (k_regs_select left 9 .. AL_RH_RC of them): the serial code
// does O(n0^2) small moves and follows every hit's anchors three times -- on global memory that is tens of milliseconds for one
// lane, and the launch waits for it.  Here a wavefront stages the kept hits and their anchors in LDS, lane 0 runs the same code
// on the copies, the wavefront writes the results back.  Fragments that do not fit the tiles stay with k_regs.
template <int RC, int AC, int RCL, int ACL>   // RC kept hits, AC anchors of theirs: the LDS tiles (72 / 1024: 62 KB, two blocks per CU; 200 / 2048: 141 KB for the few larger ones);
                                              // RCL / ACL: tiles of the next smaller instance, which takes what fits them (the instances run side by side)
__global__ void __launch_bounds__(64)
k_regs_heavy(const AlAnchor *__restrict__ chained, const uint32_t *__restrict__ frag_first, const uint32_t *__restrict__ rd_len,
             const uint32_t *__restrict__ frag_hash, WsBase W, const uint32_t *__restrict__ list, int n_list, AlParams P, unsigned long long *counters,
             uint32_t *__restrict__ regs_n0)
{
	extern __shared__ __align__(16) unsigned char s_raw[];
	AlReg *const s_r0 = (AlReg *)s_raw, *const s_m0 = s_r0 + RC, *const s_m1 = s_m0 + RC;
	AlAnchor *const s_src = (AlAnchor *)(s_m1 + RC), *const s_sa = s_src + AC, *const s_aux128 = s_sa + AC, *const s_aux128b = s_aux128 + RC + AL_RS_SCRATCH / 16 + 2;
	uint64_t *const s_aux64 = (uint64_t *)(s_aux128b + RC + AL_RS_SCRATCH / 16 + 2), *const s_aux64b = s_aux64 + RC, *const s_su0 = s_aux64b + RC, *const s_su1 = s_su0 + RC;
	int *const s_auxi = (int *)(s_su1 + RC), *const s_auxib = s_auxi + 2 * RC + 4, *const s_off = s_auxib + 2 * RC + 4, *const s_as = s_off + RC + 1;
	uint32_t *const s_res = (uint32_t *)(s_as + RC);
	const int lane = threadIdx.x;
	if ((int)blockIdx.x >= n_list) return;
	const long long tk0 = clock64();
	const uint32_t f = list[blockIdx.x];
	const uint32_t pre = regs_n0[f];
	if (pre == AL_REGS_UNSET || pre == AL_REGS_DONE || AL_REGS_BAIL(pre) || pre < 9u || pre > (uint32_t)RC) return;
	const int n0 = (int)pre;
	const uint32_t r0 = frag_first[f], n_segs = frag_first[f + 1] - r0;
	FragWs ws; d_frag_ws(W, f, ws);
	const int ql0 = (int)rd_len[r0], ql1 = n_segs > 1 ? (int)rd_len[r0 + 1] : 0;
	const AlAnchor *a = chained + W.a_off[f];
	// stage the kept hits; anchors of hit i at s_off[i] (single end: in ascending order of `as`, which is what squeezing leaves)
	{   // the kept hits' records, four words per lane in flight (a lane past the end reads the last word again)
		const int nw = n0 * (int)(sizeof(AlReg) / 4);
		for (int i0 = 0; i0 < nw; i0 += 256) {
			uint32_t v[4];
#pragma unroll
			for (int u = 0; u < 4; ++u) { const int i = i0 + u * 64 + lane; v[u] = ((const uint32_t *)ws.regs0)[i < nw ? i : nw - 1]; }
#pragma unroll
			for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
			for (int u = 0; u < 4; ++u) { const int i = i0 + u * 64 + lane; if (i < nw) ((uint32_t *)s_r0)[i] = v[u]; }
		}
	}
	__syncthreads();
	if (lane == 0) {
		int tot = 0;
		if (n_segs == 1) {     // order of staging = ascending as (ties impossible: chains are disjoint anchor ranges)
			for (int i = 0; i < n0; ++i) s_auxi[i] = i;
			for (int i = 1; i < n0; ++i) { const int t = s_auxi[i]; int j = i; while (j > 0 && s_r0[s_auxi[j - 1]].as > s_r0[t].as) { s_auxi[j] = s_auxi[j - 1]; --j; } s_auxi[j] = t; }
			for (int k = 0; k < n0; ++k) { const int i = s_auxi[k]; s_off[i] = tot; tot += s_r0[i].cnt; }
		} else for (int i = 0; i < n0; ++i) { s_off[i] = tot; tot += s_r0[i].cnt; }
		s_off[n0] = tot;
	}
	__syncthreads();
	const int tot = s_off[n0];
	if (tot > AC || (n0 <= RCL && tot <= ACL)) return;                       // the larger / the smaller instantiation, or k_regs, takes it
	// (round 6) four hits' anchors per round, their loads without a branch around them (a lane past a hit's count reads the hit's last anchor): one round trip per four
	// hits instead of one per hit (n0 reaches 200)
	for (int i0 = 0; i0 < n0; i0 += 4) {
		AlAnchor v[4]; int as_[4], cnt_[4], o_[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const int i = i0 + u < n0 ? i0 + u : n0 - 1;
			as_[u] = s_r0[i].as; cnt_[u] = i0 + u < n0 ? s_r0[i].cnt : 0; o_[u] = s_off[i];
			const int j = lane < cnt_[u] ? lane : (cnt_[u] > 0 ? cnt_[u] - 1 : 0);
			v[u] = a[as_[u] + j];
		}
#pragma unroll
		for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(v[u].x), "+v"(v[u].y));
#pragma unroll
		for (int u = 0; u < 4; ++u) if (lane < cnt_[u]) s_src[o_[u] + lane] = v[u];
#pragma unroll
		for (int u = 0; u < 4; ++u) for (int j = lane + 64; j < cnt_[u]; j += 64) s_src[o_[u] + j] = a[as_[u] + j];   // (chains of more than 64 anchors)
	}
	__syncthreads();
	if (n_segs != 2) {
		if (lane == 0) {
			for (int i = 0; i < n0; ++i) { s_as[i] = s_r0[i].as; s_r0[i].as = s_off[i]; }
			uint32_t cnt0 = 0, cnt1 = 0, sna0 = 0, sna1 = 0;
			// single end: the staged anchors are already squeezed in `as` order; d_regs_tail copies tot of them and squeezes (a no-op move)
			const bool tie = d_regs_tail(P, frag_hash[f], (int)n_segs, ql0, ql1, n0, s_r0, s_src, (uint32_t)tot, s_m0, s_m1, s_su0, s_su1, s_sa, s_aux128, s_aux64, s_auxi, cnt0, cnt1, sna0, sna1);
			s_res[0] = cnt0; s_res[1] = cnt1; s_res[2] = sna0; s_res[3] = sna1; s_res[4] = tie ? 1u : 0u;
		}
	} else {
		// Paired end, d_regs_tail's sequence with the wavefront's lanes put to use: mm_seg_gen's two passes over the anchors (hit.c:356-394: count
		// per hit and mate, then split with y rebased) are per-anchor work -- a lane per hit for the counts, a lane per anchor for the split -- and
		// the per-mate tails (hits from chains, parents, squeeze) are independent of each other: lane 0 takes mate 0, lane 1 mate 1.
		const int qlen_sum = ql0 + ql1;
		for (int i = lane; i < n0; i += 64) {
			const int o = s_off[i], cnt = s_r0[i].cnt; uint32_t c1 = 0;
			for (int j = 0; j < cnt; ++j) c1 += (uint32_t)((s_src[o + j].y & AL_SEED_SEG_MASK) >> AL_SEED_SEG_SHIFT) & 1u;
			const uint64_t sc = (uint64_t)(uint32_t)s_r0[i].score << 32;
			s_su0[i] = sc | ((uint32_t)cnt - c1); s_su1[i] = sc | c1;
		}
		__syncthreads();
		if (lane == 0) {
			uint32_t na0 = 0, nus0 = 0, nus1 = 0;
			for (int i = 0; i < n0; ++i) na0 += (uint32_t)s_su0[i];
			for (int i = 0; i < n0; ++i) { if ((int32_t)s_su0[i] != 0) s_su0[nus0++] = s_su0[i]; if ((int32_t)s_su1[i] != 0) s_su1[nus1++] = s_su1[i]; }
			s_res[0] = nus0; s_res[1] = nus1; s_res[2] = na0;
		}
		__syncthreads();
		const uint32_t na0 = s_res[2];
		AlAnchor *const sa1 = s_sa + na0;
		{   // the split: anchor t of the staged list (hits in order, their anchors in order) goes behind the earlier anchors of its mate
			uint32_t run1 = 0;
			for (int base = 0; base < tot; base += 64) {
				const int t = base + lane; const bool on = t < tot;
				AlAnchor a1; a1.x = 0; a1.y = 0; bool s1 = false;
				if (on) { a1 = s_src[t]; s1 = ((a1.y & AL_SEED_SEG_MASK) >> AL_SEED_SEG_SHIFT) & 1; }
				const unsigned long long b1 = __ballot(on && s1);
				const uint32_t r1 = run1 + (uint32_t)__popcll(b1 & ((1ULL << lane) - 1ULL));
				if (on) {
					const int qls = s1 ? ql1 : ql0, acc = s1 ? ql0 : 0;
					a1.y -= (a1.x >> 63) ? (uint64_t)(qlen_sum - (qls + acc)) : (uint64_t)acc;
					if (s1) sa1[r1] = a1; else s_sa[(uint32_t)t - r1] = a1;
				}
				run1 += (uint32_t)__popcll(b1);
			}
		}
		__syncthreads();
		bool tie = false; int na_sq = 0;
		// (round 6) mm_gen_regs of the two mates by the whole wavefront.  One lane's insertion sort of the kept hits (radix_sort_128x below 65 elements, ksort.h) was half of this
		// kernel -- 6600 cycles per hit at 15 hits per mate: O(n^2) moves of 16-byte LDS records.  It is a STABLE ascending sort followed by a reversal, so hit e goes to
		// n - 1 - #{j : x_j < x_e or (x_j = x_e and j < e)}: a lane per hit counts that over n broadcast reads.  The keys (a lane per hit, the hits' anchor offsets by a wavefront
		// scan) and the records (a lane per hit: coordinates and fuzzy lengths walk the hit's anchors) likewise; above 64 hits the reference takes its radix passes, whose
		// order only the serial restatement reproduces: lane m for mate m, as before.  mm_set_parent and the squeeze stay one lane per mate.
		const long long tg0 = (P.dbg2 & 16) ? clock64() : 0;
		const uint32_t fh = frag_hash[f];
#pragma unroll 1
		for (int m = 0; m < 2; ++m) {
			const int n = (int)s_res[m]; const uint64_t *su = m ? s_su1 : s_su0; const AlAnchor *sa = m ? sa1 : s_sa; AlAnchor *z = m ? s_aux128b : s_aux128;
			int run = 0;
			for (int base = 0; base < n; base += 64) {                         // keys: hit.c:60-67
				const int i = base + lane; const bool on = i < n;
				const uint64_t ui = on ? su[i] : 0ULL; const int c = (int32_t)(uint32_t)ui;
				int incl = c;
				for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
				const int k = run + incl - c;
				if (on) {
					const AlAnchor ak = sa[k];
					const uint32_t h = (uint32_t)d_hash64((d_hash64(ak.x) + d_hash64(ak.y)) ^ fh);
					AlAnchor e; e.x = ui ^ (uint64_t)h; e.y = (uint64_t)k << 32 | (uint32_t)(int32_t)(uint32_t)ui;
					z[i] = e;
				}
				run += __shfl(incl, 63);
			}
		}
		__syncthreads();
#pragma unroll 1
		for (int m = 0; m < 2; ++m) {                                            // order: descending by key, equal keys as the reference's sort leaves them
			const int n = (int)s_res[m]; AlAnchor *z = m ? s_aux128b : s_aux128;
			if (n > 64 || n <= 0) continue;
			AlAnchor e; e.x = 0; e.y = 0; if (lane < n) e = z[lane];
			int rank = 0;
			for (int j = 0; j < n; ++j) { const uint64_t xj = z[j].x; rank += (xj < e.x || (xj == e.x && j < lane)) ? 1 : 0; }
			__syncthreads();
			if (lane < n) z[n - 1 - rank] = e;
		}
		__syncthreads();
		if (lane < 2) {
			const int n = (int)s_res[lane]; AlAnchor *z = lane ? s_aux128b : s_aux128;
			if (n > 64) {
				tie = d_sort128(z, n, z + n);
				for (int i = 0; i < n >> 1; ++i) { const AlAnchor t = z[i]; z[i] = z[n - 1 - i]; z[n - 1 - i] = t; }
			}
		}
		__syncthreads();
#pragma unroll 1
		for (int m = 0; m < 2; ++m) {                                            // records: hit.c:70-86, then the mate's flags (hit.c:397-399)
			const int n = (int)s_res[m], ql = m ? ql1 : ql0; const AlAnchor *sa = m ? sa1 : s_sa; const AlAnchor *z = m ? s_aux128b : s_aux128; AlReg *mreg = m ? s_m1 : s_m0;
			for (int i = lane; i < n; i += 64) {
				const AlAnchor zi = z[i];
				AlReg R; d_reg_clear(&R);
				R.id = i; R.parent = AL_PARENT_UNSET;
				R.score = R.score0 = (int32_t)(zi.x >> 32);
				R.hash = (uint32_t)zi.x;
				R.cnt = (int32_t)zi.y; R.as = (int32_t)(zi.y >> 32);
				d_reg_set_coor(&R, ql, sa);
				R.flags |= ALR_SEG_SPLIT | (m ? 1u << 8 : 0u);
				mreg[i] = R;
			}
		}
		__syncthreads();
		const long long tg1 = (P.dbg2 & 16) ? clock64() : 0;
		if (lane < 2) {
			const bool m1 = lane == 1; const int n = (int)s_res[m1 ? 1 : 0]; AlReg *mreg = m1 ? s_m1 : s_m0;
			d_set_parent_pq(P.mask_level, n, mreg, P.a * 2 + P.b, m1 ? s_aux64b : s_aux64, m1 ? s_auxib : s_auxi);   // map.c:401 (s_auxi: 2 RC + 4 ints per mate)
			const long long tg2 = (P.dbg2 & 16) ? clock64() : 0;
			na_sq = d_squeeze_a(n, mreg, m1 ? sa1 : s_sa, m1 ? s_aux64b : s_aux64);                                  // align.c:873
			if ((P.dbg2 & 16) && lane == 0) { const long long tg3 = clock64(); atomicAdd(&counters[24], (unsigned long long)(tg1 - tg0)); atomicAdd(&counters[25], (unsigned long long)(tg2 - tg1)); atomicAdd(&counters[26], (unsigned long long)(tg3 - tg2)); atomicAdd(&counters[27], (unsigned long long)n); }
		}
		const unsigned long long tb = __ballot(tie);
		const int na1 = __shfl(na_sq, 1);
		__syncthreads();
		if (lane == 0) { const uint32_t c0 = s_res[0], c1 = s_res[1]; s_res[0] = c0; s_res[1] = c1; s_res[2] = na0; s_res[3] = (uint32_t)na1; s_res[4] = tb ? 1u : 0u; }
	}
	__syncthreads();
	const uint32_t cnt0 = s_res[0], cnt1 = s_res[1], sna0 = s_res[2], sna1 = s_res[3];
	for (int i = lane; i < (int)cnt0 * (int)(sizeof(AlReg) / 4); i += 64) ((uint32_t *)ws.mreg[0])[i] = ((const uint32_t *)s_m0)[i];
	if (n_segs > 1) for (int i = lane; i < (int)cnt1 * (int)(sizeof(AlReg) / 4); i += 64) ((uint32_t *)ws.mreg[1])[i] = ((const uint32_t *)s_m1)[i];
	for (int i = lane; i < tot; i += 64) ws.seg_a[0][i] = s_sa[i];
	if (lane == 0) {
		W.reg_cnt[r0] = cnt0; W.seg_na[r0] = sna0; W.seg_fast[r0] = 0;
		if (n_segs > 1) { W.reg_cnt[r0 + 1] = cnt1; W.seg_na[r0 + 1] = sna1; W.seg_fast[r0 + 1] = 0; }
		if (s_res[4]) atomicAdd(&counters[10], 1ULL);
		if (P.dbg2 & 16) { atomicAdd(&counters[28], (unsigned long long)(clock64() - tk0)); atomicAdd(&counters[29], 1ULL); }
		regs_n0[f] = AL_REGS_DONE;
	}
}

// Which tile size of k_regs_heavy takes a fragment: the smallest (kept hits, their anchors) tile it fits (the tiles are nested).  One lane per
// candidate; the five lists go to five launches of exactly their length -- every instance used to be launched on all candidates, staging each
// fragment's hits to find out that it was not its own, the largest tile at one block per CU.
struct HeavyCls { int rc[5], ac[5]; };
__global__ void __launch_bounds__(256)
k_regs_heavy_classify(WsBase W, const uint32_t *__restrict__ list, int n_list, const uint32_t *__restrict__ regs_n0, HeavyCls T,
                      uint32_t *__restrict__ cls_list /* 5 x n_list */, uint32_t *__restrict__ cls_cnt /* 5 */)
{
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n_list) return;
	const uint32_t f = list[t];
	const uint32_t pre = regs_n0[f];
	if (pre == AL_REGS_UNSET || pre == AL_REGS_DONE || AL_REGS_BAIL(pre) || pre < 9u || pre > (uint32_t)T.rc[4]) return;
	const AlReg *r = W.regs0 + W.nu_off[f];
	int tot = 0;
	for (uint32_t i = 0; i < pre; ++i) tot += r[i].cnt;
	int k = 0;
	while (k < 5 && !((int)pre <= T.rc[k] && tot <= T.ac[k])) ++k;
	if (k < 5) cls_list[(size_t)k * n_list + atomicAdd(&cls_cnt[k], 1u)] = f;
}

extern "C" __global__ void __launch_bounds__(256, AL_LB_REGS)
k_regs(const AlAnchor *__restrict__ chained, const uint64_t *__restrict__ u_all, const uint32_t *__restrict__ uo_all, const uint32_t *__restrict__ frag_first,
       const uint32_t *__restrict__ rd_len, const uint32_t *__restrict__ frag_hash, WsBase W, int n_frag, AlParams P, unsigned long long *counters,
       const uint32_t *__restrict__ regs_n0, const int part /* 0: every fragment; 1: all but k_regs_heavy's candidates (9 ... 200 kept hits), beside those kernels; 2: the candidates they left */)
{
	const int f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= n_frag) return;
	const uint32_t r0 = frag_first[f], n_segs = frag_first[f + 1] - r0, n_u = W.frag_nu[f];
	if (regs_n0) {
		const uint32_t v = regs_n0[f];
		if (v == AL_REGS_DONE) return;                                          // k_regs_heavy wrote this fragment's hits
		const bool cand = v >= 9u && v <= 200u;                                 // (k_regs_heavy turns such a value into AL_REGS_DONE and nothing else: part 1 may run beside it)
		if (part == 1 ? cand : part == 2 ? !cand : false) return;
	}
	W.reg_cnt[r0] = 0; W.seg_na[r0] = 0; W.seg_fast[r0] = 0;
	if (n_segs > 1) { W.reg_cnt[r0 + 1] = 0; W.seg_na[r0 + 1] = 0; W.seg_fast[r0 + 1] = 0; }
	if (n_u == 0) return;
	// NB: no array below is indexed by a run-time value (mate ids become two-way selects, loops over the <= 2 mates are
	// unrolled): a dynamically indexed local array would live in scratch memory, whose lines compete with the data for L2.
	FragWs ws; d_frag_ws(W, f, ws);
	const int ql0 = (int)rd_len[r0], ql1 = n_segs > 1 ? (int)rd_len[r0 + 1] : 0;
	const int qlens[2] = {ql0, ql1}, qlen_sum = ql0 + ql1;
	const AlAnchor *a = chained + W.a_off[f]; const uint64_t *u = u_all + W.a_off[f] + f; const uint32_t *uo = uo_all + W.a_off[f] + f;
	const uint32_t hash = frag_hash[f];
	int max_gap_ref;
	if (P.max_gap_ref > 0) max_gap_ref = P.max_gap_ref;
	else if (P.max_frag_len > 0) { max_gap_ref = P.max_frag_len - qlen_sum; if (max_gap_ref < P.max_gap) max_gap_ref = P.max_gap; }
	else max_gap_ref = P.max_gap;
	if (n_u == 1 && n_segs == 2 && !((P.dbg >> 20) & 1)) {
		// One chain, paired end (the common fragment): the fragment-level hit has no competitor, so chain_post (map.c:249-258)
		// cannot change it and only mm_seg_gen (hit.c:356-410) matters.  Two passes over the chain's anchors (count, then
		// split + coordinates + fuzzy lengths), both per-mate hit records built in registers and stored once; nothing else
		// of the per-fragment workspace is touched.  Same result as the general code below for n_u == 1.
		const uint64_t u0 = u[0]; const int cnt = (int32_t)(uint32_t)u0;
		a += uo[0];                                                             // the chain's anchors, at their segment's place
		uint32_t c1 = 0;
		for (int j = 0; j < cnt; ++j) c1 += (uint32_t)((a[j].y & AL_SEED_SEG_MASK) >> AL_SEED_SEG_SHIFT) & 1u;
		const uint32_t c0 = (uint32_t)cnt - c1;
		AlAnchor *const sa0 = ws.seg_a[0], *const sa1 = sa0 + c0;
		uint32_t w0 = 0, w1 = 0;
		AlAnchor f0{0, 0}, f1{0, 0}, l0{0, 0}, l1{0, 0};                       // first / last (= previous) anchor of each mate, y rebased
		int32_t bl0 = 0, ml0 = 0, bl1 = 0, ml1 = 0;
		// mm_max_stretch (align.c:495-521) of each mate's anchors on the fly: the longest run with equal reference and query
		// steps; its end anchors give the ungapped core k_ext_prep starts from, so that kernel need not re-read the anchors
		struct Stretch { AlAnchor cf, mf, ml; int score, len, max_score, max_len; };
		Stretch S0{{0, 0}, {0, 0}, {0, 0}, 0, 0, -1, 0}, S1 = S0;
		auto st_step = [](Stretch &S, const bool first, const AlAnchor &pv, const AlAnchor &cur, const int tl, const int ql, const int span) {
			if (first) { S.cf = cur; S.score = span; S.len = 1; }
			else if (ql == tl) { S.score += ql < span ? ql : span; ++S.len; }
			else { if (S.score > S.max_score) { S.max_score = S.score; S.max_len = S.len; S.mf = S.cf; S.ml = pv; } S.cf = cur; S.score = span; S.len = 1; }
		};
		for (int j = 0; j < cnt; ++j) {
			AlAnchor a1 = a[j]; const bool s1 = ((a1.y & AL_SEED_SEG_MASK) >> AL_SEED_SEG_SHIFT) & 1;
			const int qls = s1 ? ql1 : ql0, acc = s1 ? ql0 : 0;
			a1.y -= (a1.x >> 63) ? (uint64_t)(qlen_sum - (qls + acc)) : (uint64_t)acc;
			const int span = (int)(a1.y >> 32 & 0xff);
			const bool first = s1 ? w1 == 0 : w0 == 0;
			const AlAnchor pv = s1 ? l1 : l0;
			const int tl = (int32_t)a1.x - (int32_t)pv.x, ql = (int32_t)a1.y - (int32_t)pv.y;      // mm_cal_fuzzy_len, hit.c:8-24
			const int db = first ? span : (tl > ql ? tl : ql), dm = first ? span : (tl > span && ql > span ? span : tl < ql ? tl : ql);
			if (s1) { st_step(S1, first, pv, a1, tl, ql, span); if (first) f1 = a1; l1 = a1; bl1 += db; ml1 += dm; sa1[w1++] = a1; }
			else { st_step(S0, first, pv, a1, tl, ql, span); if (first) f0 = a1; l0 = a1; bl0 += db; ml0 += dm; sa0[w0++] = a1; }
		}
		auto st_done = [&](Stretch &S, const uint32_t c, const AlAnchor &fa, const AlAnchor &la, const uint32_t sid) {
			if (c == 0) return;
			AlAnchor A = fa, B = la;                                              // cnt < 2: the whole (one-anchor) hit
			if (c >= 2) { if (S.score > S.max_score) { S.mf = S.cf; S.ml = la; } A = S.mf; B = S.ml; }
			RegExt x; x.rs0 = x.re0 = x.core_score = 0; x.job = 0;
			x.rs = (int32_t)A.x + 1 - (int32_t)(A.y >> 32 & 0xff); x.qs = (int32_t)A.y + 1 - (int32_t)(A.y >> 32 & 0xff);
			x.re = (int32_t)B.x + 1; x.qe = (int32_t)B.y + 1;
			W.rext[(uint64_t)(ws.mreg[0] - W.mregs) + (uint64_t)sid * ws.cap] = x;
			W.seg_fast[r0 + sid] = 1;
		};
		st_done(S0, c0, f0, l0, 0u); st_done(S1, c1, f1, l1, 1u);
		auto make = [&](const uint32_t sid, const uint32_t c, const AlAnchor &fa, const AlAnchor &la, const int32_t blen, const int32_t mlen, const int qlen) -> AlReg {
			AlReg R; d_reg_clear(&R);                                              // mm_gen_regs for one chain (hit.c:52-88) + mm_reg_set_coor (hit.c:26-41)
			const uint32_t h = (uint32_t)d_hash64((d_hash64(fa.x) + d_hash64(fa.y)) ^ hash);
			const uint64_t zx = ((u0 >> 32 << 32) | c) ^ h;
			R.id = 0; R.parent = 0; R.score = R.score0 = (int32_t)(zx >> 32); R.hash = (uint32_t)zx; R.cnt = (int32_t)c; R.as = 0;
			const int32_t q_span = (int32_t)(fa.y >> 32 & 0xff); const int rev = (int)(fa.x >> 63);
			R.flags = (rev ? ALR_REV : 0) | ALR_SEG_SPLIT | (sid << 8);
			R.rid = (int32_t)(fa.x << 1 >> 33);
			R.rs = (int32_t)fa.x + 1 > q_span ? (int32_t)fa.x + 1 - q_span : 0;
			R.re = (int32_t)la.x + 1;
			if (!rev) { R.qs = (int32_t)fa.y + 1 - q_span; R.qe = (int32_t)la.y + 1; }
			else { R.qs = qlen - ((int32_t)la.y + 1); R.qe = qlen - ((int32_t)fa.y + 1 - q_span); }
			R.mlen = mlen; R.blen = blen;
			return R;
		};
		if (c0) ws.mreg[0][0] = make(0u, c0, f0, l0, bl0, ml0, ql0);
		if (c1) ws.mreg[1][0] = make(1u, c1, f1, l1, bl1, ml1, ql1);
		W.reg_cnt[r0] = c0 ? 1u : 0u; W.reg_cnt[r0 + 1] = c1 ? 1u : 0u;
		W.seg_na[r0] = c0; W.seg_na[r0 + 1] = c1;
		return;
	}
	bool tie = false; int n0 = (int)n_u;
	const uint32_t pre_n0 = regs_n0 ? regs_n0[f] : AL_REGS_UNSET;
	if (pre_n0 != AL_REGS_UNSET && !AL_REGS_BAIL(pre_n0)) n0 = (int)pre_n0;                                                    // k_regs_select did chain_post: ws.regs0[0 .. n0) are the kept hits
	else {
		tie = d_gen_regs(hash, qlen_sum, (int)n_u, u, a, ws.regs0, ws.aux128, uo);
		d_set_parent(P.mask_level, n0, ws.regs0, P.a * 2 + P.b, ws.aux64, ws.auxi);                   // chain_post, map.c:249-258
		if (n_segs <= 1) d_select_sub(P.pri_ratio, P.k * 2, P.best_n, &n0, ws.regs0, ws.auxi);
		else d_select_sub_multi(P.pri_ratio, 0.2f, 0.7f, max_gap_ref, P.k * 2, P.best_n, (int)n_segs, qlens, &n0, ws.regs0, ws.auxi);
	}
	uint32_t cnt0 = 0, cnt1 = 0, sna0 = 0, sna1 = 0; uint32_t tot = 0;
	if (n_segs == 1) for (int i = 0; i < n0; ++i) { const uint32_t e = (uint32_t)(ws.regs0[i].as + ws.regs0[i].cnt); tot = e > tot ? e : tot; }   // single end: the range of a[] the kept hits' anchors lie in
	tie = d_regs_tail(P, hash, (int)n_segs, ql0, ql1, n0, ws.regs0, a, tot, ws.mreg[0], ws.mreg[1], ws.seg_u[0], ws.seg_u[1], ws.seg_a[0], ws.aux128, ws.aux64, ws.auxi, cnt0, cnt1, sna0, sna1) || tie;
	W.reg_cnt[r0] = cnt0; W.seg_na[r0] = sna0;
	if (n_segs > 1) { W.reg_cnt[r0 + 1] = cnt1; W.seg_na[r0 + 1] = sna1; }
	if (tie) atomicAdd(&counters[10], 1ULL);
}

// body of the monolithic kernel: L is the group's state block -- in LDS (k_align) or, for reads beyond the LDS tiles, in HBM (k_align_long)
template <int TMAX, int QMAX>
__device__ __forceinline__ void d_align_frags(GroupLds<TMAX, QMAX> &L, const int g, const int gl,
        const uint32_t *__restrict__ rd_seq, const uint64_t *__restrict__ rd_off, const uint32_t *__restrict__ rd_len,
        const uint32_t *__restrict__ frag_first, const int32_t *__restrict__ frag_rep, const WsBase &W, const AlignShared &G, const AlLogTab &lt,
        uint8_t *__restrict__ gws, size_t gws_stride, size_t p_bytes, size_t cig_words, int n_frag, const AlParams &P,
        const uint32_t *__restrict__ frag_list, int n_list)
{
	GroupWs ws;
	{
		uint8_t *base = gws + ((size_t)blockIdx.x * (blockDim.x / GW) + g) * gws_stride;   // (groups per block from the launch: a short list runs ONE group per wavefront, see launch_mono)
		ws.p = base; ws.cig = (uint32_t *)(base + p_bytes); ws.ezc = ws.cig + cig_words; ws.sc = (uint64_t *)(ws.ezc + cig_words); ws.dbg = G.dbg;
	}
	const long long tK0 = PROF_ON(P) ? clock64() : 0;
	for (int fi = blockIdx.x * (int)(blockDim.x / GW) + g; fi < n_list; fi += gridDim.x * (int)(blockDim.x / GW)) {
		const int f = frag_list ? (int)frag_list[fi] : fi;
		const uint32_t r0 = frag_first[f], n_segs = frag_first[f + 1] - r0, n_u = W.frag_nu[f];
		if (n_u == 0) continue;
		FragWs fw; d_frag_ws(W, (uint32_t)f, fw);
		if (n_segs == 2) fw.seg_a[1] = fw.seg_a[0] + W.seg_na[r0];   // see k_regs
		int qlens[2] = {0, 0}, qlen_sum = 0, n_regs[2] = {0, 0};
		for (uint32_t s = 0; s < n_segs; ++s) { qlens[s] = (int)rd_len[r0 + s]; qlen_sum += qlens[s]; }
		int max_gap_ref;
		if (P.max_gap_ref > 0) max_gap_ref = P.max_gap_ref;
		else if (P.max_frag_len > 0) { max_gap_ref = P.max_frag_len - qlen_sum; if (max_gap_ref < P.max_gap) max_gap_ref = P.max_gap; }
		else max_gap_ref = P.max_gap;
		const int rep_len = frag_rep[f];
		bool tie = false;
		AlReg *mregs[2] = {fw.mreg[0], fw.mreg[1]};      // where each mate's hits currently live (LDS tile or HBM)
		for (uint32_t s = 0; s < n_segs; ++s) {
			const int qlen = qlens[s]; int n = (int)W.reg_cnt[r0 + s];
			// stage the mate's hits, scratch and anchors in LDS when they fit the tiles
			AlReg *regs = fw.mreg[s]; int cap = fw.cap;
			AlReg *rtmp = fw.rtmp; AlAnchor *aux128 = fw.aux128; uint64_t *aux64 = fw.aux64; int *auxi = fw.auxi;
			if (n + 1 <= AL_LREG) {
				const uint32_t *src = (const uint32_t *)fw.mreg[s]; uint32_t *dst = (uint32_t *)L.regs[s];
				for (int i = gl; i < n * (int)(sizeof(AlReg) / 4); i += GW) dst[i] = src[i];
				regs = L.regs[s]; cap = AL_LREG; rtmp = L.rtmp; aux128 = L.aux128; aux64 = L.aux64; auxi = L.auxi;
			}
			const AlAnchor *a = fw.seg_a[s];
			{
				const uint32_t na = (n_segs == 2 && s == 0) ? W.seg_na[r0] : 0xffffffffu;   // mate 0's count is exact; otherwise bound by the hits
				int need = 0; for (int i = 0; i < n; ++i) { const int e = fw.mreg[s][i].as + fw.mreg[s][i].cnt; need = e > need ? e : need; }
				(void)na;
				if (need <= AL_LANC) { for (int i = gl; i < need; i += GW) L.anc[i] = a[i]; a = L.anc; }
			}
			GSYNC();
			if (n > 0 && qlen <= QMAX) {
				const uint32_t *seq = rd_seq + rd_off[r0 + s];
				for (int i = gl; i < qlen; i += GW) {                          // align.c:865-870
					const uint8_t c = (uint8_t)((seq[i >> 3] >> ((i & 7) << 2)) & 0xf);
					L.q0[i] = c; L.q1[qlen - 1 - i] = c < 4 ? 3 - c : 4;
				}
				GSYNC();
				EzD ez; d_ez_reset(ez);
				for (int i = 0; i < n; ++i) {                                  // align.c:875-906
					AlReg r2; r2.cnt = 0;
					const long long tA = PROF_ON(P) ? clock64() : 0;
					if (!(P.dbg & 16)) d_align1(L, gl, ws, P, G, qlen, &regs[i], &r2, a, ez);
					if (PROF_ON(P) && gl == 0) atomicAdd(&ws.dbg[603], (unsigned long long)(clock64() - tA));
					if (r2.cnt > 0) {
						if (n + 1 > cap && regs != fw.mreg[s]) {               // outgrew the LDS tile: continue in HBM
							for (int j = 0; j < n; ++j) fw.mreg[s][j] = regs[j];
							regs = fw.mreg[s]; cap = fw.cap; rtmp = fw.rtmp; aux128 = fw.aux128; aux64 = fw.aux64; auxi = fw.auxi;
						}
						if (n + 1 <= cap) {                                    // mm_insert_reg, align.c:847-855
							for (int j = n - 1; j > i; --j) regs[j + 1] = regs[j];
							regs[i + 1] = r2; ++n;
						} else if (gl == 0) atomicAdd(&G.counters[7], 1ULL << 32);
					}
				}
			} else if (n > 0) { if (gl == 0) atomicAdd(&G.counters[7], 1ULL << 24); n = 0; }
			if (P.dbg & 8) { n_regs[s] = n; mregs[s] = regs; continue; }
			const long long tB = PROF_ON(P) ? clock64() : 0;
			d_filter_regs(P, qlen, &n, regs);                                  // align.c:910-911
			tie = d_hit_sort(&n, regs, aux128, rtmp) || tie;
			d_set_parent(P.mask_level, n, regs, P.a * 2 + P.b, aux64, auxi);   // align_regs tail, map.c:264-268
			d_select_sub(P.pri_ratio, P.k * 2, P.best_n, &n, regs, auxi);
			d_set_sam_pri(n, regs);
			d_set_mapq(n, regs, P.min_chain_score, P.a, rep_len, lt);
			n_regs[s] = n; mregs[s] = regs;
			if (PROF_ON(P) && gl == 0) atomicAdd(&ws.dbg[604], (unsigned long long)(clock64() - tB));
			GSYNC();
		}
		if (n_segs == 2 && P.pe_ori >= 0) {
			bool ovf = false;
			const bool small = n_regs[0] <= AL_LREG && n_regs[1] <= AL_LREG;
			d_pair(P, max_gap_ref, qlens, n_regs, mregs, small ? (PairEnt *)L.rtmp : (PairEnt *)fw.rtmp, small ? L.sc : ws.sc, small ? AL_LREG * AL_LREG : AL_PAIR_SC_CAP, lt, &tie, &ovf);
			if (ovf && gl == 0) atomicAdd(&G.counters[7], 1ULL << 40);
		}
		GSYNC();
		for (uint32_t s = 0; s < n_segs; ++s) {
			// worker_for un-flip (map.c:486-497) is applied on the host where the flip flag lives
			if (mregs[s] != fw.mreg[s]) {                                      // publish the LDS-resident hits
				const uint32_t *src = (const uint32_t *)mregs[s]; uint32_t *dst = (uint32_t *)fw.mreg[s];
				for (int i = gl; i < n_regs[s] * (int)(sizeof(AlReg) / 4); i += GW) dst[i] = src[i];
			}
			W.reg_cnt[r0 + s] = (uint32_t)n_regs[s];
		}
		if (tie && gl == 0) atomicAdd(&G.counters[10], 1ULL);
		GSYNC();
	}
	if (PROF_ON(P) && gl == 0) atomicAdd(&G.dbg[605], (unsigned long long)(clock64() - tK0));
}

template <int TMAX, int QMAX>
__global__ void __launch_bounds__(64)
k_align(const uint32_t *__restrict__ rd_seq, const uint64_t *__restrict__ rd_off, const uint32_t *__restrict__ rd_len,
        const uint32_t *__restrict__ frag_first, const int32_t *__restrict__ frag_rep, WsBase W, AlignShared G, AlLogTab lt,
        uint8_t *__restrict__ gws, size_t gws_stride, size_t p_bytes, size_t cig_words, int n_frag, AlParams P,
        const uint32_t *__restrict__ frag_list, int n_list)
{
	__shared__ GroupLds<TMAX, QMAX> lds[AL_GPB];
	const int g = threadIdx.x / GW, gl = threadIdx.x % GW;
	d_align_frags<TMAX, QMAX>(lds[g], g, gl, rd_seq, rd_off, rd_len, frag_first, frag_rep, W, G, lt, gws, gws_stride, p_bytes, cig_words, n_frag, P, frag_list, n_list);
}

// Reads longer than the LDS tiles (the whole regions AirLift's stage 3 feeds the aligner: extract_fasta_regions_with_bedfile.sh:4-7
// -> align_gaps.sh:14-15; the fork's own test/MT-orang.fa, q-inv.fa): the same code with the group's state block -- sequences, the
// seven int8 DP rows, H, the hit tiles -- in HBM instead of LDS, up to AL_MAX_READ_LEN bases per read.  Correct for any such
// length; slow per read (every DP row is a few dependent HBM round trips), which is acceptable for the few thousand
// sequences of that stage.
#define AL_LONG_QMAX AL_MAX_READ_LEN
#define AL_LONG_TMAX (2 * AL_MAX_READ_LEN + 128)
typedef GroupLds<AL_LONG_TMAX, AL_LONG_QMAX> GroupLong;
__global__ void __launch_bounds__(64)
k_align_long(const uint32_t *__restrict__ rd_seq, const uint64_t *__restrict__ rd_off, const uint32_t *__restrict__ rd_len,
             const uint32_t *__restrict__ frag_first, const int32_t *__restrict__ frag_rep, WsBase W, AlignShared G, AlLogTab lt,
             uint8_t *__restrict__ gws, size_t gws_stride, size_t p_bytes, size_t cig_words, int n_frag, AlParams P,
             const uint32_t *__restrict__ frag_list, int n_list, GroupLong *__restrict__ state)
{
	const int g = threadIdx.x / GW, gl = threadIdx.x % GW;
	d_align_frags<AL_LONG_TMAX, AL_LONG_QMAX>(state[(size_t)blockIdx.x * AL_GPB + g], g, gl, rd_seq, rd_off, rd_len, frag_first, frag_rep, W, G, lt, gws, gws_stride, p_bytes, cig_words, n_frag, P, frag_list, n_list);
}

// Tap for the parity tests: the extension DP alone on caller-supplied (target, query) pairs, one 16-lane group per pair -- what the
// reference's --print-aln-seq tap shows per ksw call (align.c:313-339): sequences as passed to ksw_extd2_sse, flag; out: ez->score, CIGAR.
struct DbgKswJob { uint32_t toff, qoff; int32_t tlen, qlen, flag, pad; };
struct DbgPkLds {                  // what d_ksw_pk asks of its LDS structure (AL_DBG bit 20: the two-cells-per-lane form on every tap call of up to 352 target bases)
	static constexpr int kPtb = 0;
	uint8_t selE[512 + 2 * 352 + 32], selO[512 + 2 * 352 + 32];
	uint32_t __attribute__((aligned(8))) wtab[352];
	uint8_t tbuf[352 + 16];
	uint32_t ezc[AL_LCIG];
	uint8_t ptb[4];
};
__global__ void __launch_bounds__(64)
k_dbg_ksw(const uint8_t *__restrict__ seqs, const DbgKswJob *__restrict__ jobs, int n, AlParams P, uint8_t *__restrict__ gws, size_t gws_stride, size_t p_bytes, size_t cig_words,
          int32_t *__restrict__ out /* per job: score, max, max_q, max_t, mqe, mqe_t, zdropped, reach_end, n_cigar */, uint32_t *__restrict__ cig_out, int cig_cap)
{
	__shared__ GroupLds<1024, 512> lds[AL_GPB];
	__shared__ DbgPkLds pk[AL_GPB];
	const int g = threadIdx.x / GW, gl = threadIdx.x % GW;
	GroupLds<1024, 512> &L = lds[g];
	GroupWs ws;
	{
		uint8_t *base = gws + ((size_t)blockIdx.x * AL_GPB + g) * gws_stride;
		ws.p = base; ws.cig = (uint32_t *)(base + p_bytes); ws.ezc = ws.cig + cig_words; ws.sc = nullptr; ws.dbg = nullptr;
	}
	const int bw = (int)(P.bw * 1.5 + 1.);
	for (int j = blockIdx.x * AL_GPB + g; j < n; j += gridDim.x * AL_GPB) {
		const DbgKswJob jb = jobs[j];
		int32_t *o = out + (size_t)j * 9;
		if (jb.tlen > 1024 || jb.qlen > 512) { if (gl == 0) o[8] = -1; continue; }
		for (int i = gl; i < jb.tlen; i += GW) L.tbuf[i] = seqs[jb.toff + i];
		for (int i = gl; i < jb.qlen; i += GW) L.qbuf[i] = seqs[jb.qoff + i];
		GSYNC();
		EzD ez;
		const int eb = (jb.flag & EZ_EXTZ_ONLY) ? P.end_bonus : -1;                      // align.c: extensions pass opt->end_bonus, the core re-alignment -1
		if (((P.dbg >> 20) & 1) && jb.tlen <= 352) {
			DbgPkLds &K = pk[g];
			const int nb = (jb.tlen + 15) / 16, np = nb <= 4 ? 2 : nb <= 8 ? 4 : nb <= 12 ? 6 : nb <= 16 ? 8 : 11, pad = 32 * np;
			for (int i = gl; i < jb.tlen; i += GW) K.tbuf[i] = L.tbuf[i];
			for (int i = gl; i < jb.qlen + 2 * pad + 16; i += GW) { const int t = i - pad; const uint32_t b0 = t >= 0 && t < jb.qlen ? L.qbuf[jb.qlen - 1 - t] : 0u; K.selE[i] = (uint8_t)(b0 < 4 ? b0 : 0x0du); K.selO[i] = (uint8_t)(b0 < 4 ? 4u + b0 : 0x0du); }
			GSYNC();
			d_ez_reset(ez);
			if (np == 2) d_ksw_pk<2>(K, K.selE + pad, K.selO + pad, gl, ws, jb.qlen, jb.tlen, P, bw, P.zdrop, eb, jb.flag, ez);
			else if (np == 4) d_ksw_pk<4>(K, K.selE + pad, K.selO + pad, gl, ws, jb.qlen, jb.tlen, P, bw, P.zdrop, eb, jb.flag, ez);
			else if (np == 6) d_ksw_pk<6>(K, K.selE + pad, K.selO + pad, gl, ws, jb.qlen, jb.tlen, P, bw, P.zdrop, eb, jb.flag, ez);
			else if (np == 8) d_ksw_pk<8>(K, K.selE + pad, K.selO + pad, gl, ws, jb.qlen, jb.tlen, P, bw, P.zdrop, eb, jb.flag, ez);
			else d_ksw_pk<11>(K, K.selE + pad, K.selO + pad, gl, ws, jb.qlen, jb.tlen, P, bw, P.zdrop, eb, jb.flag, ez);
		} else
		d_ksw_extd2(L, gl, ws, jb.qlen, jb.tlen, P, bw, P.zdrop, eb, jb.flag, ez);
		if (gl == 0) {
			o[0] = ez.score; o[1] = ez.max; o[2] = ez.max_q; o[3] = ez.max_t; o[4] = ez.mqe; o[5] = ez.mqe_t; o[6] = ez.zdropped; o[7] = ez.reach_end; o[8] = ez.n_cigar;
			for (int i = 0; i < ez.n_cigar && i < cig_cap; ++i) cig_out[(size_t)j * cig_cap + i] = ws.cur_ezc[i];
		}
		GSYNC();
	}
}
extern "C" int al_dbg_ksw(al_ctx_t *c, int n, const uint8_t *seqs, size_t n_seq_bytes, const int32_t *jobs6 /* toff, qoff, tlen, qlen, flag, 0 per job */, int32_t *out9, uint32_t *cig_out, int cig_cap)
{
	if (!c || n <= 0) return -1;
	AL_HIP_CHECK(hipSetDevice(c->device));
	hipStream_t s = c->stream;
	const int bw = (int)(c->opt.bw * 1.5 + 1.), ncol = ((std::min(512, bw + 1) + 15) / 16 + 1);
	const size_t p_bytes = ((size_t)(512 + 1024) * ncol * 16 + 63) / 64 * 64, cig_words = 1600, stride = p_bytes + cig_words * 8;
	int nb = (n + AL_GPB - 1) / AL_GPB; if (nb > 512) nb = 512;
	uint8_t *d_seq = nullptr, *d_ws = nullptr; DbgKswJob *d_jobs = nullptr; int32_t *d_out = nullptr; uint32_t *d_cig = nullptr;
	int rc = -1;
	if (hipMalloc((void **)&d_seq, n_seq_bytes + 64) == hipSuccess && hipMalloc((void **)&d_ws, (size_t)nb * AL_GPB * stride) == hipSuccess && hipMalloc((void **)&d_jobs, (size_t)n * sizeof(DbgKswJob)) == hipSuccess &&
	    hipMalloc((void **)&d_out, (size_t)n * 36) == hipSuccess && hipMalloc((void **)&d_cig, (size_t)n * cig_cap * 4 + 16) == hipSuccess &&
	    hipMemcpyAsync(d_seq, seqs, n_seq_bytes, hipMemcpyHostToDevice, s) == hipSuccess && hipMemcpyAsync(d_jobs, jobs6, (size_t)n * sizeof(DbgKswJob), hipMemcpyHostToDevice, s) == hipSuccess) {
		hipLaunchKernelGGL(k_dbg_ksw, dim3(nb), dim3(GW * AL_GPB), 0, s, (const uint8_t *)d_seq, (const DbgKswJob *)d_jobs, n, c->P, d_ws, stride, p_bytes, cig_words, d_out, d_cig, cig_cap);
		if (hipMemcpyAsync(out9, d_out, (size_t)n * 36, hipMemcpyDeviceToHost, s) == hipSuccess && hipMemcpyAsync(cig_out, d_cig, (size_t)n * cig_cap * 4, hipMemcpyDeviceToHost, s) == hipSuccess &&
		    hipStreamSynchronize(s) == hipSuccess) rc = 0;
	}
	(void)hipFree(d_seq); (void)hipFree(d_ws); (void)hipFree(d_jobs); (void)hipFree(d_out); (void)hipFree(d_cig);
	return rc;
}

// =============================================================================================
// Fast path of the extension stage: the same work as k_align, batched by KIND of work so that every wavefront runs
// one code path on similar-sized problems (k_align, with its per-fragment mix of DP sizes, spends most of its cycles
// waiting on divergent groups):
//   k_ext_prep    lane / fragment : max colinear stretch, DP windows, ungapped core score + z-drop test, emits <= 2 DP jobs / hit
//   (hipCUB radix sort of the jobs by (block-count class, rows))
//   k_ext_dp<NB>  16-lane group / job, 4 jobs / wavefront, one launch per block-count class
//   k_ext_finish  lane / fragment : CIGAR assembly, mm_update_extra, filtering, MAPQ, pairing
// Fragments that need something unusual (z-drop re-alignment or split of the core, CIGARs longer than the per-lane
// buffer) are flagged and re-done from scratch by k_align on that short list.
// =============================================================================================
struct ExtJob {                    // 32 bytes
	uint64_t toff;                 // reference base index of target[0] (left jobs walk downwards from it)
	uint32_t read;                 // read index (query source)
	uint32_t qoff;                 // index into qseq0[rev] of query[0] (left jobs walk downwards)
	uint16_t qlen, tlen;
	uint8_t rev, kind, pad0, pad1; // kind 0 = left extension (both sequences reversed), 1 = right extension
	uint32_t pad2;
};
struct ExtOut {                    // 48 bytes
	int32_t max, max_q, max_t, mqe_t;
	uint32_t flags_ncig;           // bit0 reach_end, bit1 zdropped, n_cigar << 8
	uint32_t cig_off;              // arena index when n_cigar > 6
	uint32_t cig[6];
};
#define AL_FCIG 32                 // CIGAR words a finish lane can assemble in registers/scratch

struct ExtShared {
	ExtJob *jobs; ExtOut *outs; RegExt *rext; const uint64_t *job_off; uint32_t *frag_slow; uint32_t *job_key;
	unsigned long long *hist;      // [0..5] jobs per block-count class, [6] empty slots
};

#define AL_LANE_QC 64              // longest query a lane-per-job DP handles
#define AL_NCLS 10                 // job classes: 0..2 lane-per-job (target <= 16/32/64), 3..8 group DP (NB = 1,2,4,8,22,32), 9 LDS-row DP; 10 = empty slot
#define AL_HIST_N 40               // [0..AL_NCLS] jobs per class, [12..19] job cursors of the group-DP kernels, [20] fragments left to the monolithic kernel by prep, [24..24+AL_NCLS] target bases per class, [36..37] jobs of class 7 with <= 12 / 13 ... 16 blocks
__device__ __forceinline__ int d_job_class(int qlen, int tlen, int lane_ok)
{
	const int b = (tlen + 15) / 16;
	if (lane_ok && qlen <= AL_LANE_QC && b <= 2) return b <= 1 ? 0 : 1;   // (class 2, targets <= 64, measured slower than the group DP: LDS-bound)
	return b <= 1 ? 3 : b <= 2 ? 4 : b <= 4 ? 5 : b <= 8 ? 6 : b <= 22 ? 7 : b <= 32 ? 8 : 9;
}

// One fragment's share of k_ext_prep.  WAVE = false: the calling lane walks the fragment's hits one after the other.  WAVE = true: the 64 lanes
// of the calling wavefront take a hit each (fragments with many hits: a lane per fragment ran 6 ms on them while the rest of the batch was done
// in 2).  Everything a hit needs is its own -- window, ungapped core, z-drop test, the two flank jobs at job_off[f] + 2 * (hit's position) --
// except the rule that the first hit that needs the monolithic kernel ends the fragment: hits before it keep their jobs, it and the hits
// after it leave empty slots.  The wavefront form settles that with a ballot per group of 64 hits.
#define AL_PREP_HEAVY 8            // jobs (two per hit) from which a fragment goes to the wavefront form
template <bool WAVE>
__device__ __forceinline__ void d_ext_prep_frag(const int f, const int lane, const uint32_t *__restrict__ rd_seq, const uint64_t *__restrict__ rd_off, const uint32_t *__restrict__ rd_len,
                                const uint32_t *__restrict__ frag_first, const WsBase &W, const AlignShared &G, const ExtShared &E, const AlParams &P, const int tmax, const int qmax,
                                unsigned *s_hist, unsigned *s_tl, unsigned *s_sub, unsigned long long &c_regs, unsigned long long &c_ref, unsigned long long &c_cig)
{
	const int lane_ok = !((P.dbg >> 29) & 1);
	const uint32_t r0 = frag_first[f], n_segs = frag_first[f + 1] - r0;
	FragWs fw; d_frag_ws(W, (uint32_t)f, fw);
	const uint64_t B2 = (uint64_t)(fw.mreg[0] - W.mregs);      // index of mreg[0][0] in the hit array == index into rext
	const uint32_t jb0 = (uint32_t)E.job_off[f]; bool slow = false;
	uint32_t n_done = 0;                                       // hits whose jobs were emitted
	AlReg *const mreg0 = fw.mreg[0], *const mreg1 = fw.mreg[1];
	AlAnchor *const sa0 = fw.seg_a[0], *const sa1 = n_segs == 2 ? fw.seg_a[0] + W.seg_na[r0] : nullptr;
	const int ge1 = P.q + P.e, ge2 = P.q2 + P.e2;
	const bool diag_ok = P.a > 0 && P.b > 0 && P.a + P.b < (ge1 < ge2 ? ge1 : ge2) && (P.zdrop < 0 || P.zdrop >= P.b + (ge1 > ge2 ? ge1 : ge2));
	// one hit: 0 = jobs in x / jl / jr, 1 = needs the monolithic kernel, 2 = the same because its window exceeds the tiles (counted)
	auto hit = [&](const uint32_t s, const int i, const AlReg *regs, const AlAnchor *a, const int qlen, const uint32_t *seq, const uint32_t jb, RegExt &x, ExtJob &jl, ExtJob &jr) -> int {
		const AlReg *r = &regs[i];
		jl.qlen = jl.tlen = 0; jr.qlen = jr.tlen = 0; jl.pad0 = jl.pad1 = jr.pad0 = jr.pad1 = 0; jl.pad2 = jr.pad2 = 0;
		jl.toff = jr.toff = 0; jl.read = jr.read = r0 + s; jl.qoff = jr.qoff = 0; jl.rev = jr.rev = 0; jl.kind = 0; jr.kind = 1;
		x.job = jb; x.rs = x.qs = x.re = x.qe = x.rs0 = x.re0 = x.core_score = 0;
		if (r->cnt > 0) {
			const int32_t rid = r->rid, rev = (r->flags & ALR_REV) ? 1 : 0;   // == contig / strand of a[r->as] (mm_reg_set_coor)
			const int32_t ref_len = (int32_t)G.seq_len[rid]; const uint64_t ref_off = G.seq_off[rid];
			int32_t rs, qs, re, qe;
			if (i == 0 && W.seg_fast[r0 + s]) { const RegExt pre = E.rext[B2 + (uint64_t)s * fw.cap]; rs = pre.rs; qs = pre.qs; re = pre.re; qe = pre.qe; }   // from k_regs
			else {
				int as1, cnt1;
				d_max_stretch(r, a, &as1, &cnt1);
				rs = (int32_t)a[as1].x + 1 - (int32_t)(a[as1].y >> 32 & 0xff);
				qs = (int32_t)a[as1].y + 1 - (int32_t)(a[as1].y >> 32 & 0xff);
				re = (int32_t)a[as1 + cnt1 - 1].x + 1; qe = (int32_t)a[as1 + cnt1 - 1].y + 1;
			}
			int l = qs;                                                   // align.c:613-620
			l += l * P.a + P.end_bonus > P.q ? (l * P.a + P.end_bonus - P.q) / P.e : 0;
			const int32_t rs0 = rs - l > 0 ? rs - l : 0;
			l = qlen - qe;
			l += l * P.a + P.end_bonus > P.q ? (l * P.a + P.end_bonus - P.q) / P.e : 0;
			const int32_t re0 = re + l < ref_len ? re + l : ref_len;
			if (re0 - rs0 > tmax) return 2;
			// ungapped core (align.c:724-731) + mm_test_zdrop with the single 'M' op (align.c:47-89)
			const ReadAcc Q{seq, qlen, rev, qs}; const RefAcc T{G.S4, ref_off + (uint64_t)rs};
			const int len = qe - qs; int sc = 0, zs = 0, zmax = INT32_MIN, zmi = -1, zdrop_max = 0;
			int k = 0;
			{   // eight bases per step; eight unambiguous matches raise both running scores monotonically
				const int msc = P.a < 0 ? -P.a : P.a;
				auto win = [&](const uint32_t qw, const uint32_t tw) {
					if (qw == tw && !(qw & 0x44444444u)) { sc += 8 * P.a; zs += 8 * msc; if (zs >= zmax) zmax = zs; else { const int z = zmax - zs; if (z > zdrop_max) zdrop_max = z; } return; }
#pragma unroll
					for (int b = 0; b < 8; ++b) {
						const int cq = (int)(qw >> (4 * b) & 0xf), ct = (int)(tw >> (4 * b) & 0xf);
						if (cq >= 4 || ct >= 4) sc += P.e2; else sc += cq == ct ? P.a : -P.b;
						zs += d_mat(P, ct, cq);
						if (zs < zmax) { const int z = zmax - zs; if (z > zdrop_max) zdrop_max = z; }   // diff = 0 along the diagonal
						else zmax = zs;
					}
				};
				for (; k + 32 <= len; k += 32) {                                 // (round 6) four windows' loads at a time
					uint32_t qw[4], tw[4]; d_win8x4(Q, k, T, k, qw, tw);
#pragma unroll
					for (int u = 0; u < 4; ++u) win(qw[u], tw[u]);
				}
				for (; k + 8 <= len; k += 8) win(Q.win8(k), T.win8(k));
			}
			for (; k < len; ++k) {
				const int cq = Q(k), ct = T(k);
				if (cq >= 4 || ct >= 4) sc += P.e2; else sc += cq == ct ? P.a : -P.b;
				zs += d_mat(P, ct, cq);
				if (zs < zmax) { const int z = zmax - zs; (void)zmi; if (z > zdrop_max) zdrop_max = z; }   // diff = 0 along the diagonal
				else { zmax = zs; zmi = k; }
			}
			if (zdrop_max > P.zdrop) return 1;                            // needs the second DP pass / split: monolithic path
			x.rs = rs; x.qs = qs; x.re = re; x.qe = qe; x.rs0 = rs0; x.re0 = re0; x.core_score = sc;
			if (qs > 0 && rs > 0) {                                       // left extension job (align.c:690-705)
				jl.qlen = (uint16_t)qs; jl.tlen = (uint16_t)(rs - rs0); jl.rev = (uint8_t)rev;
				jl.qoff = (uint32_t)(qs - 1); jl.toff = ref_off + (uint64_t)(rs - 1);
			}
			if (qe < qlen && re < re0) {                                  // right extension job (align.c:760-771)
				jr.qlen = (uint16_t)(qlen - qe); jr.tlen = (uint16_t)(re0 - re); jr.rev = (uint8_t)rev;
				jr.qoff = (uint32_t)qe; jr.toff = ref_off + (uint64_t)re;
			}
		}
		// Closed form instead of a DP job when the flank matches its target with at most one mismatch (no N, query not longer
		// than the target): the extension then runs along the diagonal, every gapped path loses >= q+e > a+b, and max / max_q /
		// max_t / mqe_t / reach_end / CIGAR follow from the running diagonal score exactly as ksw_extd2 computes them
		// (checked against the reference DP in tests/test_diag_shortcut.py; jobs done this way carry pad0 = 1).
		// (the argument needs one mismatch to cost less than any gap: a + b < min(q + e, q2 + e2); true for the short-read
		// scores 2/8/12,2/24,1 -- with other scores every flank takes the DP kernels)
		// and the z-drop test (ksw2.h:160-176) must be unable to fire: along and after the diagonal the row maximum stays
		// within b + max(q + e, q2 + e2) (+ e2 per unit of diagonal offset, which the test allows for) of the running maximum
		if (!((P.dbg >> 31) & 1) && diag_ok && r->cnt > 0) {
			const int32_t rid2 = r->rid, rev2 = (r->flags & ALR_REV) ? 1 : 0;
			const uint64_t ref_off2 = G.seq_off[rid2];
			const ReadAcc Qa{seq, qlen, rev2, 0}; const RefAcc Ta{G.S4, ref_off2};
			auto shortcut = [&](const int side, const int ql, const int tl) -> bool {
				if (ql == 0 || ql > tl) return false;
				int sc = 0, mx = 0, pos = -1, nmm = 0;
				for (int k = 0; k < ql; ) {
					// eight bases at once while they all match (the order inside the eight does not matter for that: the left flank's window
					// is read forwards): the running score rises with every base, so the maximum and its position are the last of them
					if (k + 8 <= ql) {
						const uint32_t qw = side == 0 ? Qa.win8(x.qs - 8 - k) : Qa.win8(x.qe + k), tw = side == 0 ? Ta.win8(x.rs - 8 - k) : Ta.win8(x.re + k);
						if (qw == tw && !(qw & 0x44444444u)) { sc += 8 * P.a; if (sc > mx) { mx = sc; pos = k + 7; } k += 8; continue; }
					}
					const int kend = k + 8 <= ql ? k + 8 : ql;
					for (; k < kend; ++k) {
						const int cq = side == 0 ? Qa(x.qs - 1 - k) : Qa(x.qe + k);
						const int ct = side == 0 ? Ta(x.rs - 1 - k) : Ta(x.re + k);
						if (cq > 3 || ct > 3) return false;
						if (cq == ct) sc += P.a; else { sc -= P.b; if (++nmm > 1) return false; }
						if (sc > mx) { mx = sc; pos = k; }
					}
				}
				// a target at least w + 1 longer than the query makes the band run off the matrix (st > en at row 2*ql + w - 1,
				// ksw2_extd2_sse.c:135): the reference flags that like a z-drop and ends at the maximum, never at the query end
				const int wband = (int)(P.bw * 1.5 + 1.);
				const bool runoff = tl >= ql + wband + 1;
				const bool reach = !runoff && sc + P.end_bonus > mx;
				const int ncig = (reach || pos >= 0) ? 1 : 0;
				ExtOut o; o.max = mx; o.max_q = pos; o.max_t = pos; o.mqe_t = ql - 1;
				o.flags_ncig = (uint32_t)(reach ? 1 : 0) | (uint32_t)(runoff ? 2 : 0) | (uint32_t)ncig << 8; o.cig_off = 0;
				o.cig[0] = ncig ? (uint32_t)(reach ? ql : pos + 1) << 4 : 0; o.cig[1] = o.cig[2] = o.cig[3] = o.cig[4] = o.cig[5] = 0;
				E.outs[jb + side] = o;                                        // (the slot is this hit's own: harmless if the hit ends up behind the fragment's first slow one)
				return true;
			};
			if (shortcut(0, jl.qlen, jl.tlen)) jl.pad0 = 1;
			if (shortcut(1, jr.qlen, jr.tlen)) jr.pad0 = 1;
		}
		if (r->cnt > 0 && !((P.dbg >> 18) & 1) && (jl.qlen == 0 || jl.pad0) && (jr.qlen == 0 || jr.pad0)) x.job |= 0x80000000u;   // no DP needed: finished below
		return 0;
	};
	auto commit = [&](const uint32_t s, const int i, const uint32_t jb, const RegExt &x, const ExtJob &jl, const ExtJob &jr) {
		E.rext[B2 + (uint64_t)s * fw.cap + i] = x;
		E.jobs[jb] = jl; E.jobs[jb + 1] = jr;
		const int c0 = (jl.qlen && !jl.pad0) ? d_job_class(jl.qlen, jl.tlen, lane_ok) : AL_NCLS, c1 = (jr.qlen && !jr.pad0) ? d_job_class(jr.qlen, jr.tlen, lane_ok) : AL_NCLS;
		// key: class | 16-cell blocks of the target | size -- inside a class the jobs are ordered by block count first, so that the 9 ... 22-block class
		// can be launched as three kernels (12, 16, 22 register blocks: a row costs every instantiated block a skip test and two selects)
		const uint32_t b0 = (uint32_t)std::min(63, (jl.tlen + 15) / 16), b1 = (uint32_t)std::min(63, (jr.tlen + 15) / 16);
		// (round 5) ... then by direction (left extensions have right-aligned gaps, align.c:694-704): a wavefront of the DP kernels holds four jobs of one direction
		E.job_key[jb] = c0 < AL_NCLS ? ((uint32_t)c0 << 20 | b0 << 14 | 0u << 13 | (uint32_t)std::min(0x1fff, jl.qlen + jl.tlen)) : 0xffffffffu;
		E.job_key[jb + 1] = c1 < AL_NCLS ? ((uint32_t)c1 << 20 | b1 << 14 | 1u << 13 | (uint32_t)std::min(0x1fff, jr.qlen + jr.tlen)) : 0xffffffffu;
		atomicAdd(&s_hist[c0], 1u); atomicAdd(&s_hist[c1], 1u);
		if (c0 == 7 && b0 <= 16) atomicAdd(&s_sub[b0 <= 12 ? 0 : 1], 1u);
		if (c1 == 7 && b1 <= 16) atomicAdd(&s_sub[b1 <= 12 ? 0 : 1], 1u);
		if (c0 < AL_NCLS) atomicAdd(&s_tl[c0], (unsigned)jl.tlen); if (c1 < AL_NCLS) atomicAdd(&s_tl[c1], (unsigned)jr.tlen);
	};
	auto do_seg = [&](const uint32_t s, const AlReg *regs, const AlAnchor *a) {
		const int qlen = (int)rd_len[r0 + s], n = (int)W.reg_cnt[r0 + s];
		const uint32_t *seq = rd_seq + rd_off[r0 + s];
		if (n > 0 && qlen > qmax) { if (!WAVE || lane == 0) atomicAdd(&G.counters[7], 1ULL << 0); slow = true; return; }
		if (!WAVE) {
			for (int i = 0; i < n; ++i) {
				RegExt x; ExtJob jl, jr; const uint32_t jb = jb0 + 2u * n_done;
				const int code = hit(s, i, regs, a, qlen, seq, jb, x, jl, jr);
				if (code) { if (code == 2) atomicAdd(&G.counters[7], 1ULL << 8); slow = true; return; }
				commit(s, i, jb, x, jl, jr); ++n_done;
			}
		} else {
			for (int base = 0; base < n; base += 64) {
				const int i = base + lane; const bool on = i < n;
				RegExt x; ExtJob jl, jr; const uint32_t jb = jb0 + 2u * (n_done + (uint32_t)lane);
				const int code = on ? hit(s, i, regs, a, qlen, seq, jb, x, jl, jr) : 0;
				const unsigned long long bad = __ballot(code != 0);
				const int first = bad ? __ffsll((long long)bad) - 1 : 64;
				if (on && lane < first) commit(s, i, jb, x, jl, jr);
				if (bad) { if (lane == first && code == 2) atomicAdd(&G.counters[7], 1ULL << 8); n_done += (uint32_t)first; slow = true; return; }
				n_done += (uint32_t)(n - base < 64 ? n - base : 64);
			}
		}
	};
	do_seg(0, mreg0, sa0);
	if (n_segs == 2 && !slow) do_seg(1, mreg1, sa1);
	// Hits whose two flanks are closed forms (or absent) are finished right here -- what k_ext_finish does for a hit
	// after its DP jobs (mm_align1 tail, align.c:698-788: CIGAR = one M run, coordinates, mm_update_extra) -- so that
	// k_ext_finish neither re-reads their job records nor streams their sequences again.  Done after the scan above
	// because a fragment that turned out "slow" is redone from the untouched hits by the monolithic kernel.
	if (WAVE) __threadfence_block();                                  // (the hits' records written by other lanes of the wavefront)
	auto fin_seg = [&](const uint32_t s, AlReg *regs, const AlAnchor *a) {
		const int qlen = (int)rd_len[r0 + s], n = (int)W.reg_cnt[r0 + s];
		const uint32_t *seq = rd_seq + rd_off[r0 + s];
		for (int i = WAVE ? lane : 0; i < n; i += WAVE ? 64 : 1) {
			RegExt x = E.rext[B2 + (uint64_t)s * fw.cap + i];
			if (!(x.job & 0x80000000u)) continue;
			x.job &= 0x7fffffffu;
			AlReg R = regs[i];
			const int32_t rid = R.rid, rev = (R.flags & ALR_REV) ? 1 : 0;
			const uint64_t ref_off = G.seq_off[rid];
			R.n_cigar = 0; R.dp_score = 0; R.dp_max = 0; R.dp_max2 = 0; R.n_ambi = 0;
			int32_t rs1 = x.rs, qs1 = x.qs, re1 = x.re, qe1 = x.qe;
			R.dp_score = x.core_score;
			if (E.jobs[x.job].qlen) {
				const ExtOut *po = &E.outs[x.job]; const bool reach = po->flags_ncig & 1;
				if (po->flags_ncig >> 8) R.dp_score += po->max;
				rs1 = x.rs - (reach ? po->mqe_t + 1 : po->max_t + 1);
				qs1 = x.qs - (reach ? x.qs : po->max_q + 1);
			}
			if (E.jobs[x.job + 1].qlen) {
				const ExtOut *po = &E.outs[x.job + 1]; const bool reach = po->flags_ncig & 1;
				if (po->flags_ncig >> 8) R.dp_score += po->max;
				re1 = x.re + (reach ? po->mqe_t + 1 : po->max_t + 1);
				qe1 = x.qe + (reach ? qlen - x.qe : po->max_q + 1);
			}
			uint32_t cg1[1] = { (uint32_t)(qe1 - qs1) << 4 };                // left M + core M + right M merge into one run (mm_append_cigar)
			R.n_cigar = 1; R.flags |= ALR_HAS_P;
			R.rs = rs1; R.re = re1;
			if (rev) { R.qs = qlen - qe1; R.qe = qlen - qs1; } else { R.qs = qs1; R.qe = qe1; }
			d_update_extra<true>(P, &R, cg1, ReadAcc{seq, qlen, rev, qs1}, RefAcc{G.S4, ref_off + (uint64_t)rs1});
			R.cig_inl[0] = cg1[0]; R.cig_inl[1] = R.cig_inl[2] = R.cig_inl[3] = 0; R.cigar_off = AL_CIG_INLINE;
			regs[i] = R;
			++c_regs; c_ref += (unsigned long long)(x.re0 - x.rs0); c_cig += 1;
		}
	};
	if (!slow) { fin_seg(0, mreg0, sa0); if (n_segs == 2) fin_seg(1, mreg1, sa1); }
	// unused job slots of this fragment (a slow fragment stops early): mark empty
	for (uint32_t j = jb0 + 2u * n_done + (WAVE ? (uint32_t)lane : 0u); j < (uint32_t)E.job_off[f + 1]; j += WAVE ? 64u : 1u) {
		E.job_key[j] = 0xffffffffu; ExtJob z; z.qlen = z.tlen = 0; z.toff = 0; z.read = 0; z.qoff = 0; z.rev = z.kind = z.pad0 = z.pad1 = 0; z.pad2 = 0; E.jobs[j] = z; atomicAdd(&s_hist[AL_NCLS], 1u);
	}
	if (!WAVE || lane == 0) E.frag_slow[f] = slow ? 1u : 0u;
}

extern "C" __global__ void __launch_bounds__(256, AL_LB_PREP)
k_ext_prep(const uint32_t *__restrict__ rd_seq, const uint64_t *__restrict__ rd_off, const uint32_t *__restrict__ rd_len,
           const uint32_t *__restrict__ frag_first, WsBase W, AlignShared G, ExtShared E, int n_frag, AlParams P, int tmax, int qmax,
           const uint32_t *__restrict__ order /* fragments by number of hits, descending: the lanes of a wavefront walk equally many hits (a lane per fragment runs as long as its wavefront's longest) */,
           int heavy_jobs /* fragments with at least this many job slots are k_ext_prep_wave's (0: none) */)
{
	const int t_ = blockIdx.x * blockDim.x + threadIdx.x;
	const int f = t_ < n_frag ? (order ? (int)order[t_] : t_) : n_frag;
	unsigned long long c_regs = 0, c_ref = 0, c_cig = 0;
	__shared__ unsigned s_hist[AL_NCLS + 1], s_tl[AL_NCLS + 1], s_sub[2];      // jobs / target bases per class of this block (no run-time indexed local arrays: see k_regs); s_sub: jobs of class 7 with <= 12 / 13 ... 16 blocks
	if (threadIdx.x <= AL_NCLS) { s_hist[threadIdx.x] = 0; s_tl[threadIdx.x] = 0; }
	if (threadIdx.x < 2) s_sub[threadIdx.x] = 0;
	__syncthreads();
	if (f < n_frag && W.frag_nu[f] != 0) {
		if (!(heavy_jobs > 0 && (int)(E.job_off[f + 1] - E.job_off[f]) >= heavy_jobs))
			d_ext_prep_frag<false>(f, 0, rd_seq, rd_off, rd_len, frag_first, W, G, E, P, tmax, qmax, s_hist, s_tl, s_sub, c_regs, c_ref, c_cig);
	} else if (f < n_frag) E.frag_slow[f] = 0;
	__syncthreads();
	if (threadIdx.x <= AL_NCLS && s_hist[threadIdx.x]) atomicAdd(&E.hist[threadIdx.x], (unsigned long long)s_hist[threadIdx.x]);   // one atomic per block and class
	if (threadIdx.x <= AL_NCLS && s_tl[threadIdx.x]) atomicAdd(&E.hist[24 + threadIdx.x], (unsigned long long)s_tl[threadIdx.x]);
	if (threadIdx.x < 2 && s_sub[threadIdx.x]) atomicAdd(&E.hist[36 + threadIdx.x], (unsigned long long)s_sub[threadIdx.x]);
	for (int d = 32; d > 0; d >>= 1) { c_regs += __shfl_xor(c_regs, d); c_ref += __shfl_xor(c_ref, d); c_cig += __shfl_xor(c_cig, d); }
	if ((threadIdx.x & 63) == 0) { if (c_regs) atomicAdd(&G.counters[4], c_regs); if (c_ref) atomicAdd(&G.counters[5], c_ref); if (c_cig) atomicAdd(&G.counters[6], c_cig); }
}

// the fragments at the head of the hits-descending order (at least heavy_jobs job slots), a wavefront each
extern "C" __global__ void __launch_bounds__(64)
k_ext_prep_wave(const uint32_t *__restrict__ rd_seq, const uint64_t *__restrict__ rd_off, const uint32_t *__restrict__ rd_len,
                const uint32_t *__restrict__ frag_first, WsBase W, AlignShared G, ExtShared E, int n_frag, AlParams P, int tmax, int qmax,
                const uint32_t *__restrict__ order, int heavy_jobs)
{
	unsigned long long c_regs = 0, c_ref = 0, c_cig = 0;
	__shared__ unsigned s_hist[AL_NCLS + 1], s_tl[AL_NCLS + 1], s_sub[2];
	if (threadIdx.x <= AL_NCLS) { s_hist[threadIdx.x] = 0; s_tl[threadIdx.x] = 0; }
	if (threadIdx.x < 2) s_sub[threadIdx.x] = 0;
	__syncthreads();
	for (int t = blockIdx.x; t < n_frag; t += gridDim.x) {
		const int f = (int)order[t];
		if ((int)(E.job_off[f + 1] - E.job_off[f]) < heavy_jobs) break;     // (descending order: nothing heavy behind it)
		if (W.frag_nu[f] == 0) continue;
		d_ext_prep_frag<true>(f, (int)threadIdx.x, rd_seq, rd_off, rd_len, frag_first, W, G, E, P, tmax, qmax, s_hist, s_tl, s_sub, c_regs, c_ref, c_cig);
	}
	__syncthreads();
	if (threadIdx.x <= AL_NCLS && s_hist[threadIdx.x]) atomicAdd(&E.hist[threadIdx.x], (unsigned long long)s_hist[threadIdx.x]);
	if (threadIdx.x <= AL_NCLS && s_tl[threadIdx.x]) atomicAdd(&E.hist[24 + threadIdx.x], (unsigned long long)s_tl[threadIdx.x]);
	if (threadIdx.x < 2 && s_sub[threadIdx.x]) atomicAdd(&E.hist[36 + threadIdx.x], (unsigned long long)s_sub[threadIdx.x]);
	for (int d = 32; d > 0; d >>= 1) { c_regs += __shfl_xor(c_regs, d); c_ref += __shfl_xor(c_ref, d); c_cig += __shfl_xor(c_cig, d); }
	if (threadIdx.x == 0) { if (c_regs) atomicAdd(&G.counters[4], c_regs); if (c_ref) atomicAdd(&G.counters[5], c_ref); if (c_cig) atomicAdd(&G.counters[6], c_cig); }
}

template <int QMAXJ, int TMAXJ, bool PK = false> struct JobLds {
	static constexpr bool kQrReady = true;   // k_ext_dp stores the reversed, padded query itself
	// traceback tile: jobs of more than 64 target bases have more than 40 rows of at least 32 bytes -- they never fit it, and without it twice
	// as many wavefronts fit a CU (the kernel waits on LDS reads and byte stores with 3.5 waves per SIMD)
	static constexpr int kPtb = TMAXJ <= 64 && !PK ? AL_LPTB : 0;   // (the two-cells-per-lane form has no LDS traceback tile: al_dev_ksw2.h)
	uint8_t sq[QMAXJ + 2 * TMAXJ + 32];      // TMAXJ bytes of front pad, the reversed query, zeros up to qlen + TMAXJ + 16 (al_dev_ksw.h)
	uint8_t selO[PK ? QMAXJ + 2 * TMAXJ + 32 : 1];   // two-cells-per-lane form (al_dev_ksw2.h): sq holds the score permute's selector bytes for a cell in the low half, selO for one in the high half
	uint32_t __attribute__((aligned(8))) wtab[PK ? TMAXJ : 2];   // ... and the score tables of the target bases, two words per lane and superblock
	uint8_t tbuf[TMAXJ + 16];
	uint32_t ezc[AL_LCIG];
	uint8_t ptb[kPtb > 0 ? kPtb : 4];
};

// DP jobs of one block-count class: 4 jobs per wavefront, all running d_ksw_reg<NB>
#ifndef AL_WPE_DP
#define AL_WPE_DP(NB) ((NB) <= 8 ? 4 : (NB) <= 16 ? 3 : 2)
#endif
template <int NB, int QMAXJ, int TMAXJ, bool PK = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(AL_WPE_DP(NB), AL_WPE_DP(NB))))
k_ext_dp(const uint32_t *__restrict__ rd_seq, const uint64_t *__restrict__ rd_off, const uint32_t *__restrict__ rd_len,
         AlignShared G, ExtShared E, const uint32_t *__restrict__ sorted_idx, uint32_t first, uint32_t count,
         uint8_t *__restrict__ gws, size_t gws_stride, size_t p_bytes, size_t cig_words, AlParams P)
{
	__shared__ JobLds<QMAXJ, TMAXJ, PK> lds[4];
	const int g = threadIdx.x / GW, gl = threadIdx.x % GW;
	JobLds<QMAXJ, TMAXJ, PK> &L = lds[g];
	GroupWs ws;
	{ uint8_t *base = gws + ((size_t)blockIdx.x * 4 + g) * gws_stride; ws.p = base; ws.cig = (uint32_t *)(base + p_bytes); ws.ezc = ws.cig + cig_words; ws.sc = nullptr; ws.dbg = G.dbg; ws.cur_cig = nullptr; ws.cur_cig_cap = 0; ws.cur_ezc = nullptr; }
	const int bw = (int)(P.bw * 1.5 + 1.);
	// jobs are handed out largest first from a shared cursor, four neighbours of the size-sorted list per wavefront:
	// the grid is as large as the chip holds resident and no wavefront is left with a long tail
	unsigned long long *cursor = E.hist + 12 + (NB == 1 ? 0 : NB == 2 ? 1 : NB == 4 ? 2 : NB == 8 ? 3 : NB == 22 ? 4 : NB == 32 ? 5 : NB == 12 ? 6 : 7);
	for (;;) {
		unsigned long long base = 0;
		if (threadIdx.x == 0) base = atomicAdd(cursor, 4ULL);
		base = __shfl(base, 0);
		if (base >= count) break;
		if (base + g >= count) continue;
		const uint32_t jj = count - 1 - ((uint32_t)base + g);
		const uint32_t j = sorted_idx[first + jj];
		const ExtJob job = E.jobs[j];
		const int ql = job.qlen, tl = job.tlen;
		const uint32_t *seq = rd_seq + rd_off[job.read]; const int rqlen = (int)rd_len[job.read];
		// the query goes in reversed (ksw2_extd2_sse.c:118: qr[t] = query[qlen - 1 - t]) behind the front pad, zeros after it
		uint8_t *const qr = L.sq + TMAXJ;
		if (job.kind == 0) {                                                  // left: both reversed (mm_seq_rev, align.c:694-695): query[i] = Q(qoff - i)
			const ReadAcc Q{seq, rqlen, job.rev, 0};
			for (int t = gl; t < ql + TMAXJ + 16; t += GW) qr[t] = t < ql ? (uint8_t)Q((int)job.qoff - (ql - 1 - t)) : 0;
			for (int i = gl; i < tl; i += GW) L.tbuf[i] = (uint8_t)d_seq4(G.S4, job.toff - (uint64_t)i);
		} else {
			const ReadAcc Q{seq, rqlen, job.rev, (int)job.qoff};
			for (int t = gl; t < ql + TMAXJ + 16; t += GW) qr[t] = t < ql ? (uint8_t)Q(ql - 1 - t) : 0;
			for (int i = gl; i < tl; i += GW) L.tbuf[i] = (uint8_t)d_seq4(G.S4, job.toff + (uint64_t)i);
		}
		GSYNC();
		EzD ez; d_ez_reset(ez);
		const int flag = job.kind == 0 ? (EZ_EXTZ_ONLY | EZ_RIGHT | EZ_REV_CIGAR) : EZ_EXTZ_ONLY;
		static_assert(TMAXJ == 16 * NB, "the query pad of JobLds is one block row");
		if constexpr (PK) {
			static_assert(NB % 2 == 0, "two blocks per superblock");
			// selector words of the score permute, from the staged bytes (front pad included: whatever it holds, its words are in bounds and unused)
			for (int i = gl; i < ql + 2 * TMAXJ + 16; i += GW) { const uint32_t b0 = L.sq[i]; L.selO[i] = (uint8_t)(b0 < 4 ? 4u + b0 : 0x0du); L.sq[i] = (uint8_t)(b0 < 4 ? b0 : 0x0du); }
			GSYNC();
			// the job list is ordered by direction inside a block count (k_ext_prep's key): nearly every wavefront holds jobs of one direction and takes the form compiled for it
			const unsigned long long w_act = __ballot(1), w_right = __ballot((flag & EZ_RIGHT) != 0);
			if (w_right == 0) d_ksw_pk<NB / 2, JobLds<QMAXJ, TMAXJ, PK>, 0>(L, L.sq + TMAXJ, L.selO + TMAXJ, gl, ws, ql, tl, P, bw, P.zdrop, P.end_bonus, flag, ez);
			else if (w_right == w_act) d_ksw_pk<NB / 2, JobLds<QMAXJ, TMAXJ, PK>, 1>(L, L.sq + TMAXJ, L.selO + TMAXJ, gl, ws, ql, tl, P, bw, P.zdrop, P.end_bonus, flag, ez);
			else d_ksw_pk<NB / 2, JobLds<QMAXJ, TMAXJ, PK>, 2>(L, L.sq + TMAXJ, L.selO + TMAXJ, gl, ws, ql, tl, P, bw, P.zdrop, P.end_bonus, flag, ez);
		} else
		d_ksw_reg<NB>(L, gl, ws, ql, tl, P, bw, P.zdrop, P.end_bonus, flag, ez);
		ExtOut o;
		o.max = ez.max; o.max_q = ez.max_q; o.max_t = ez.max_t; o.mqe_t = ez.mqe_t;
		o.flags_ncig = (uint32_t)(ez.reach_end ? 1 : 0) | (uint32_t)(ez.zdropped ? 2 : 0) | (uint32_t)ez.n_cigar << 8; o.cig_off = 0;
		for (int i = 0; i < 6; ++i) o.cig[i] = i < ez.n_cigar ? ws.cur_ezc[i] : 0;
		if (ez.n_cigar > 6) {
			unsigned long long off = 0;
			if (gl == 0) off = atomicAdd(G.arena_cnt, (unsigned long long)ez.n_cigar);
			off = __shfl(off, 0, GW);
			if (off + ez.n_cigar <= G.arena_cap) { for (int i = gl; i < ez.n_cigar; i += GW) G.arena[off + i] = ws.cur_ezc[i]; o.cig_off = (uint32_t)off; }
			else { if (gl == 0) atomicAdd(&G.counters[9], 1ULL); o.cig_off = 0xffffffffu; }
		}
		if (gl == 0) E.outs[j] = o;
		GSYNC();
	}
}

// DP jobs with short targets (<= TC cells) and short queries: ONE LANE PER JOB, 64 jobs per wavefront.  A 16-lane group
// spends one wave-instruction per 4 jobs with most lanes idle on such tiny problems; here every lane walks its own
// anti-diagonals over the reference's 16-cell blocks (garbage cells included, so results stay bit-identical) with the
// packed cell state in LDS as [cell][lane] (conflict-free) and the traceback bytes in a wave-interleaved HBM scratch.
template <int TC, int QC>
__global__ void __launch_bounds__(64)
k_ext_dp_lane(const uint32_t *__restrict__ rd_seq, const uint64_t *__restrict__ rd_off, const uint32_t *__restrict__ rd_len,
              AlignShared G, ExtShared E, const uint32_t *__restrict__ sorted_idx, uint32_t first, uint32_t count,
              uint8_t *__restrict__ tbws, size_t tb_stride, AlParams P)
{
	__shared__ uint32_t sA[TC * 64], sB[TC * 64];
	__shared__ int32_t sH[TC * 64];
	__shared__ uint8_t sQ[(QC + 32) * 64];
	const int lane = threadIdx.x;
	uint8_t *tb = tbws + (size_t)blockIdx.x * tb_stride;        // [row][cell][lane]
#define LA(t) sA[(t) * 64 + lane]
#define LB(t) sB[(t) * 64 + lane]
#define LH(t) sH[(t) * 64 + lane]
#define LQ(i) sQ[(i) * 64 + lane]
	int q = P.q, e = P.e, q2 = P.q2, e2 = P.e2;
	if (q2 + e2 < q + e) { int t = q; q = q2; q2 = t; t = e; e = e2; e2 = t; }
	const int qe = q + e;
	const int8_t qe_ = (int8_t)(q + e), qe2_ = (int8_t)(q2 + e2);
	const int8_t sc_mch = (int8_t)P.a, sc_mis = (int8_t)(-P.b), sc_amb = (int8_t)(P.sc_ambi > 0 ? -P.sc_ambi : P.sc_ambi);
	const int8_t sc_N = sc_amb == 0 ? (int8_t)(-e2) : sc_amb;
	int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
	const int w = (int)(P.bw * 1.5 + 1.);
	const uint32_t m1 = (uint8_t)(int8_t)(-q - e), m2 = (uint8_t)(int8_t)(-q2 - e2);
	for (uint32_t jj = blockIdx.x * 64 + lane; jj < count; jj += gridDim.x * 64) {
		const uint32_t j = sorted_idx[first + jj];
		const ExtJob job = E.jobs[j];
		const int qlen = job.qlen, tlen = job.tlen;
		const bool right = job.kind == 0;                                   // left extension: KSW_EZ_RIGHT | REV_CIGAR (align.c:697)
		const int rev_cigar = job.kind == 0;
		const uint32_t *seq = rd_seq + rd_off[job.read]; const int rqlen = (int)rd_len[job.read];
		const int tlen_ = (tlen + 15) / 16, qlen_ = (qlen + 15) / 16;
		int n_col_ = qlen < tlen ? qlen : tlen;
		n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
		const int prow = n_col_ * 16;
		// reversed query (qr[t] = query[qlen-1-t], zero padded) and target bytes
		{
			const ReadAcc Q{seq, rqlen, job.rev, job.kind == 0 ? 0 : (int)job.qoff};
			for (int t = 0; t < qlen_ * 16 + 32 && t < QC + 32; ++t) {
				int c = 0;
				if (t < qlen) { const int i = qlen - 1 - t; c = job.kind == 0 ? Q((int)job.qoff - i) : Q(i); }
				LQ(t) = (uint8_t)c;
			}
			for (int t = 0; t < tlen_ * 16; ++t) {
				uint32_t sfv = 0;
				if (t < tlen) sfv = d_seq4(G.S4, job.kind == 0 ? job.toff - (uint64_t)t : job.toff + (uint64_t)t);
				LA(t) = m1 | m1 << 8 | m2 << 16 | m1 << 24;
				LB(t) = m1 | m2 << 8 | 0u << 16 | sfv << 24;
				LH(t) = KSW_NEG_INF;
			}
		}
		EzD ez; d_ez_reset(ez);
		int last_st = -1, last_en = -1, r;
		for (r = 0; r < qlen + tlen - 1; ++r) {
			int st, en;
			d_row_bounds(r, qlen, tlen, w, st, en);
			if (st > en) { ez.zdropped = 1; break; }
			const int st0 = st, en0 = en;
			st = st / 16 * 16; en = (en + 16) / 16 * 16 - 1;
			int8_t x1 = (int8_t)(-q - e), x21 = (int8_t)(-q2 - e2), v1 = (int8_t)(-q - e);
			if (st > 0) {
				if (st - 1 >= last_st && st - 1 <= last_en) { const uint32_t av = LA(st - 1); x1 = (int8_t)av; v1 = (int8_t)(av >> 8); x21 = (int8_t)(av >> 16); }
			} else v1 = r == 0 ? (int8_t)(-q - e) : r < long_thres ? (int8_t)(-e) : r == long_thres ? (int8_t)long_diff : (int8_t)(-e2);
			if (en >= r) {                                                    // :150-153
				const int8_t ub = r == 0 ? (int8_t)(-q - e) : r < long_thres ? (int8_t)(-e) : r == long_thres ? (int8_t)long_diff : (int8_t)(-e2);
				LB(r) = (LB(r) & 0xffff0000u) | m1 | m2 << 8;
				LA(r) = (LA(r) & 0x00ffffffu) | (uint32_t)(uint8_t)ub << 24;
			}
			for (int t = st0; t <= en0; t += 16)                               // score bytes, 16 per store (:158-176)
				for (int i = 0; i < 16; ++i) {
					const int tt = t + i;
					if (tt >= tlen_ * 16) break;
					const uint32_t bv = LB(tt);
					const uint8_t sq = (uint8_t)(bv >> 24), sq2 = LQ(qlen - 1 - r + tt);
					int8_t sc = sq == sq2 ? sc_mch : sc_mis;
					if (sq == 4 || sq2 == 4) sc = sc_N;
					LB(tt) = (bv & 0xff00ffffu) | (uint32_t)(uint8_t)sc << 16;
				}
			uint8_t *pr = tb + ((size_t)r * prow - st) * 64 + lane;
			int8_t xprev = x1, x2prev = x21, vprev = v1;
			for (int t = st; t <= en; ++t) {                                   // :182-306, cell by cell
				const uint32_t av = LA(t), bv = LB(t);
				const int8_t xt1 = xprev, vt1 = vprev, x2t1 = x2prev;
				xprev = (int8_t)av; vprev = (int8_t)(av >> 8); x2prev = (int8_t)(av >> 16);
				const int8_t ut = (int8_t)(av >> 24), yo = (int8_t)bv, y2o = (int8_t)(bv >> 8);
				int8_t z = (int8_t)(bv >> 16);
				int8_t a = (int8_t)(xt1 + vt1), bb = (int8_t)(yo + ut), a2 = (int8_t)(x2t1 + vt1), b2 = (int8_t)(y2o + ut);
				int d;
				d = (a > z || (right && a == z)) ? 1 : 0;   z = z > a ? z : a;
				d = (bb > z || (right && bb == z)) ? 2 : d; z = z > bb ? z : bb;
				d = (a2 > z || (right && a2 == z)) ? 3 : d; z = z > a2 ? z : a2;
				d = (b2 > z || (right && b2 == z)) ? 4 : d; z = z > b2 ? z : b2;
				z = z < sc_mch ? z : sc_mch;
				const int8_t un = (int8_t)(z - vt1), vn = (int8_t)(z - ut);
				int8_t tmp = (int8_t)(z - q); a = (int8_t)(a - tmp); bb = (int8_t)(bb - tmp);
				tmp = (int8_t)(z - q2); a2 = (int8_t)(a2 - tmp); b2 = (int8_t)(b2 - tmp);
				const bool pa = right ? a >= 0 : a > 0, pb = right ? bb >= 0 : bb > 0, pa2 = right ? a2 >= 0 : a2 > 0, pb2 = right ? b2 >= 0 : b2 > 0;
				const int8_t xn = (int8_t)((pa ? a : 0) - qe_), yn = (int8_t)((pb ? bb : 0) - qe_);
				const int8_t x2n = (int8_t)((pa2 ? a2 : 0) - qe2_), y2n = (int8_t)((pb2 ? b2 : 0) - qe2_);
				d |= (pa ? 0x08 : 0) | (pb ? 0x10 : 0) | (pa2 ? 0x20 : 0) | (pb2 ? 0x40 : 0);
				LA(t) = (uint32_t)(uint8_t)xn | (uint32_t)(uint8_t)vn << 8 | (uint32_t)(uint8_t)x2n << 16 | (uint32_t)(uint8_t)un << 24;
				LB(t) = (bv & 0xffff0000u) | (uint32_t)(uint8_t)yn | (uint32_t)(uint8_t)y2n << 8;
				pr[(size_t)t * 64] = (uint8_t)d;
			}
			{   // exact max (:307-349) in the reference's evaluation order
				int max_H, max_t;
				if (r > 0) {
					int HH[4], tt4[4]; const int en1 = st0 + (en0 - st0) / 4 * 4; int t;
					const int u_en0 = (int8_t)(LA(en0) >> 24), v_en0 = (int8_t)(LA(en0) >> 8);
					max_H = en0 > 0 ? LH(en0 - 1) + u_en0 : LH(en0) + v_en0; LH(en0) = max_H;
					max_t = en0;
					for (int i = 0; i < 4; ++i) HH[i] = max_H, tt4[i] = max_t;
					for (t = st0; t < en1; t += 4)
						for (int i = 0; i < 4; ++i) { const int h = LH(t + i) + (int)(int8_t)(LA(t + i) >> 8); LH(t + i) = h; if (h > HH[i]) HH[i] = h, tt4[i] = t; }
					for (int i = 0; i < 4; ++i) if (max_H < HH[i]) max_H = HH[i], max_t = tt4[i] + i;
					for (; t < en0; ++t) { const int h = LH(t) + (int)(int8_t)(LA(t) >> 8); LH(t) = h; if (h > max_H) max_H = h, max_t = t; }
				} else { LH(0) = (int)(int8_t)(LA(0) >> 8) - qe; max_H = LH(0); max_t = 0; }
				if (r - st0 == qlen - 1 && LH(st0) > ez.mqe) { ez.mqe = LH(st0); ez.mqe_t = st0; }
				bool brk = false;
				if (max_H > ez.max) { ez.max = max_H; ez.max_t = max_t; ez.max_q = r - max_t; }
				else if (max_t >= ez.max_t && r - max_t >= ez.max_q) {
					const int tl = max_t - ez.max_t, ql = (r - max_t) - ez.max_q, l = tl > ql ? tl - ql : ql - tl;
					if (P.zdrop >= 0 && ez.max - max_H > P.zdrop + l * e2) { ez.zdropped = 1; brk = true; }
				}
				if (brk) break;
				if (r == qlen + tlen - 2 && en0 == tlen - 1) ez.score = LH(tlen - 1);
			}
			last_st = st; last_en = en;
		}
		// backtrack (ksw2.h:119-151) with the run being extended kept in registers; up to 6 ops inline, longer CIGARs in the arena
		ExtOut o; uint32_t big[AL_FCIG]; int n_c = 0; uint32_t cur = 0xffffffffu; bool ovf = false;
		{
			int i0 = -1, j0 = -1;
			if (!ez.zdropped && ez.mqe + P.end_bonus > ez.max) { ez.reach_end = 1; i0 = ez.mqe_t; j0 = qlen - 1; }
			else if (ez.max_t >= 0 && ez.max_q >= 0) { i0 = ez.max_t; j0 = ez.max_q; }
			int i = i0, jq = j0, state = 0;
#define PUSH(op_, len_) do { const uint32_t op__ = (op_); if (cur != 0xffffffffu && op__ == (cur & 0xf)) cur += (uint32_t)(len_) << 4; else { if (cur != 0xffffffffu) { if (n_c < AL_FCIG) big[n_c] = cur; else ovf = true; ++n_c; } cur = (uint32_t)(len_) << 4 | op__; } } while (0)
			while (i >= 0 && jq >= 0) {
				int force_state = -1, st, en; const int rr = i + jq;
				d_row_bounds(rr, qlen, tlen, w, st, en);
				const int off = st / 16 * 16, off_end = (en + 16) / 16 * 16 - 1;
				if (i < off) force_state = 2;
				if (i > off_end) force_state = 1;
				const uint32_t tmp = force_state < 0 ? tb[((size_t)rr * prow + i - off) * 64 + lane] : 0;
				if (state == 0) state = tmp & 7;
				else if (!(tmp >> (state + 2) & 1)) state = 0;
				if (state == 0) state = tmp & 7;
				if (force_state >= 0) state = force_state;
				if (state == 0) { PUSH(0, 1); --i; --jq; }
				else if (state == 1 || state == 3) { PUSH(2, 1); --i; }
				else { PUSH(1, 1); --jq; }
			}
			if (i0 >= 0) { if (i >= 0) PUSH(2, i + 1); if (jq >= 0) PUSH(1, jq + 1); }
			if (cur != 0xffffffffu) { if (n_c < AL_FCIG) big[n_c] = cur; else ovf = true; ++n_c; }
#undef PUSH
			if (!rev_cigar) for (int k2 = 0; k2 < n_c >> 1 && n_c <= AL_FCIG; ++k2) { const uint32_t t = big[k2]; big[k2] = big[n_c - 1 - k2]; big[n_c - 1 - k2] = t; }
		}
		if (ovf) atomicAdd(&G.counters[7], 1ULL << 48);
		o.max = ez.max; o.max_q = ez.max_q; o.max_t = ez.max_t; o.mqe_t = ez.mqe_t;
		o.flags_ncig = (uint32_t)(ez.reach_end ? 1 : 0) | (uint32_t)(ez.zdropped ? 2 : 0) | (uint32_t)n_c << 8; o.cig_off = 0;
		for (int k2 = 0; k2 < 6; ++k2) o.cig[k2] = k2 < n_c ? big[k2] : 0;
		if (n_c > 6 && !ovf) {
			const unsigned long long aoff = atomicAdd(G.arena_cnt, (unsigned long long)n_c);
			if (aoff + n_c <= G.arena_cap) { for (int k2 = 0; k2 < n_c; ++k2) G.arena[aoff + k2] = big[k2]; o.cig_off = (uint32_t)aoff; }
			else { atomicAdd(&G.counters[9], 1ULL); o.cig_off = 0xffffffffu; }
		}
		E.outs[j] = o;
	}
#undef LA
#undef LB
#undef LH
#undef LQ
}

// same for targets wider than 22 blocks: LDS-row DP
template <int TMAX, int QMAX>
__global__ void __launch_bounds__(64)
k_ext_dp_lds(const uint32_t *__restrict__ rd_seq, const uint64_t *__restrict__ rd_off, const uint32_t *__restrict__ rd_len,
             AlignShared G, ExtShared E, const uint32_t *__restrict__ sorted_idx, uint32_t first, uint32_t count,
             uint8_t *__restrict__ gws, size_t gws_stride, size_t p_bytes, size_t cig_words, AlParams P)
{
	__shared__ GroupLds<TMAX, QMAX> lds[1];
	const int gl = threadIdx.x % GW;
	GroupLds<TMAX, QMAX> &L = lds[0];
	GroupWs ws;
	{ uint8_t *base = gws + (size_t)blockIdx.x * gws_stride; ws.p = base; ws.cig = (uint32_t *)(base + p_bytes); ws.ezc = ws.cig + cig_words; ws.sc = nullptr; ws.dbg = G.dbg; ws.cur_cig = nullptr; ws.cur_cig_cap = 0; ws.cur_ezc = nullptr; }
	const int bw = (int)(P.bw * 1.5 + 1.);
	for (uint32_t jj = blockIdx.x; jj < count; jj += gridDim.x) {
		const uint32_t j = sorted_idx[first + jj];
		const ExtJob job = E.jobs[j];
		const int ql = job.qlen, tl = job.tlen;
		const uint32_t *seq = rd_seq + rd_off[job.read]; const int rqlen = (int)rd_len[job.read];
		if (job.kind == 0) {
			const ReadAcc Q{seq, rqlen, job.rev, 0};
			for (int i = gl; i < ql; i += GW) L.qbuf[i] = (uint8_t)Q((int)job.qoff - i);
			for (int i = gl; i < tl; i += GW) L.tbuf[i] = (uint8_t)d_seq4(G.S4, job.toff - (uint64_t)i);
		} else {
			const ReadAcc Q{seq, rqlen, job.rev, (int)job.qoff};
			for (int i = gl; i < ql; i += GW) L.qbuf[i] = (uint8_t)Q(i);
			for (int i = gl; i < tl; i += GW) L.tbuf[i] = (uint8_t)d_seq4(G.S4, job.toff + (uint64_t)i);
		}
		GSYNC();
		EzD ez; d_ez_reset(ez);
		const int flag = job.kind == 0 ? (EZ_EXTZ_ONLY | EZ_RIGHT | EZ_REV_CIGAR) : EZ_EXTZ_ONLY;
		d_ksw_lds(L, gl, ws, ql, tl, P, bw, P.zdrop, P.end_bonus, flag, ez);
		ExtOut o;
		o.max = ez.max; o.max_q = ez.max_q; o.max_t = ez.max_t; o.mqe_t = ez.mqe_t;
		o.flags_ncig = (uint32_t)(ez.reach_end ? 1 : 0) | (uint32_t)(ez.zdropped ? 2 : 0) | (uint32_t)ez.n_cigar << 8; o.cig_off = 0;
		for (int i = 0; i < 6; ++i) o.cig[i] = i < ez.n_cigar ? ws.cur_ezc[i] : 0;
		if (ez.n_cigar > 6) {
			unsigned long long off = 0;
			if (gl == 0) off = atomicAdd(G.arena_cnt, (unsigned long long)ez.n_cigar);
			off = __shfl(off, 0, GW);
			if (off + ez.n_cigar <= G.arena_cap) { for (int i = gl; i < ez.n_cigar; i += GW) G.arena[off + i] = ws.cur_ezc[i]; o.cig_off = (uint32_t)off; }
			else { if (gl == 0) atomicAdd(&G.counters[9], 1ULL); o.cig_off = 0xffffffffu; }
		}
		if (gl == 0) E.outs[j] = o;
		GSYNC();
	}
}

struct LdsCig { uint32_t *b; __device__ __forceinline__ uint32_t &operator[](uint32_t i) const { return b[i * 256]; } };
template <class CG>
__device__ __forceinline__ void d_fcig_append(AlReg *r, CG cig, int n, const uint32_t *src)
{   // mm_append_cigar (align.c:288-311) into a per-lane buffer
	if (n == 0) return;
	r->flags |= ALR_HAS_P;
	if (r->n_cigar > 0 && (cig[r->n_cigar - 1] & 0xf) == (src[0] & 0xf)) {
		cig[r->n_cigar - 1] += src[0] >> 4 << 4;
		for (int i = 1; i < n; ++i) cig[r->n_cigar + i - 1] = src[i];
		r->n_cigar += n - 1;
	} else { for (int i = 0; i < n; ++i) cig[r->n_cigar + i] = src[i]; r->n_cigar += n; }
}

// fragments k_ext_prep left to the monolithic kernel (oversize, z-drop inside a closed-form flank): known before the DP jobs run
__global__ void __launch_bounds__(256)
k_collect_slow(const uint32_t *__restrict__ frag_slow, int n_frag, uint32_t *__restrict__ list, uint32_t *__restrict__ cnt)
{
	const int f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f < n_frag && frag_slow[f] != 0) list[atomicAdd(cnt, 1u)] = (uint32_t)f;
}

// One fragment's share of k_ext_finish.  WAVE: the wavefront's lanes take a hit each for the per-hit part (CIGAR assembly, mm_update_extra: loops over
// the hit's bases), lane 0 does the bookkeeping between hits (filter, order, parents, selection, MAPQ, pairing) as the lane form does.
// CS: lane stride of the CIGAR assembly buffer (s_cig points at the calling lane's first word).
template <int S> struct LdsCigS { uint32_t *b; __device__ __forceinline__ uint32_t &operator[](uint32_t i) const { return b[i * S]; } };
#define AL_FIN_HEAVY 24            // job slots (two per hit) from which a fragment goes to the wavefront form (batches of at most 400 k fragments)
template <bool WAVE, int CS>
__device__ __forceinline__ void d_ext_finish_frag(const int f, const int lane, uint32_t *s_cig, const uint32_t *__restrict__ rd_seq, const uint64_t *__restrict__ rd_off, const uint32_t *__restrict__ rd_len,
                                  const uint32_t *__restrict__ frag_first, const int32_t *__restrict__ frag_rep, const WsBase &W, const AlignShared &G, const ExtShared &E,
                                  const AlLogTab &lt, uint64_t *__restrict__ sc_ws, const uint64_t *__restrict__ sc_off, const AlParams &P, uint32_t *__restrict__ slow_list, uint32_t *__restrict__ n_slow,
                                  unsigned long long &c_regs, unsigned long long &c_ref, unsigned long long &c_cig)
{
		const uint32_t r0 = frag_first[f], n_segs = frag_first[f + 1] - r0;
		FragWs fw; d_frag_ws(W, (uint32_t)f, fw);
		const uint64_t B2 = (uint64_t)(fw.mreg[0] - W.mregs);
		bool slow = E.frag_slow[f] != 0;
		// pre-check: every hit's CIGAR must fit the per-lane buffer, and none of its DP jobs may have been z-dropped in a
		// way the monolithic path handles differently (nothing to do: z-dropped extensions are handled identically)
		auto precheck = [&](const uint32_t s, const AlReg *regs) {
			const int n = (int)W.reg_cnt[r0 + s];
			bool bad = false;
			for (int i = WAVE ? lane : 0; i < n && !bad && (WAVE || !slow); i += WAVE ? 64 : 1) {
				if (regs[i].cnt == 0 || (regs[i].flags & ALR_HAS_P)) continue;     // empty, or already finished by k_ext_prep
				const RegExt x = E.rext[B2 + (uint64_t)s * fw.cap + i];
				const uint32_t nl = E.jobs[x.job].qlen ? E.outs[x.job].flags_ncig >> 8 : 0, nr = E.jobs[x.job + 1].qlen ? E.outs[x.job + 1].flags_ncig >> 8 : 0;
				if (nl + 1 + nr > AL_FCIG) bad = true;
			}
			if (WAVE ? (__ballot(bad) != 0ULL) : bad) slow = true;
		};
		if (!slow) precheck(0, fw.mreg[0]);
		if (!slow && n_segs == 2) precheck(1, fw.mreg[1]);
		if (slow) { if (!WAVE || lane == 0) slow_list[atomicAdd(n_slow, 1u)] = (uint32_t)f; }
		else {
			// (no local array is indexed by a run-time value and the hit being finished lives in registers: see k_regs)
			const int ql0 = (int)rd_len[r0], ql1 = n_segs > 1 ? (int)rd_len[r0 + 1] : 0, qlen_sum = ql0 + ql1;
			int max_gap_ref;
			if (P.max_gap_ref > 0) max_gap_ref = P.max_gap_ref;
			else if (P.max_frag_len > 0) { max_gap_ref = P.max_frag_len - qlen_sum; if (max_gap_ref < P.max_gap) max_gap_ref = P.max_gap; }
			else max_gap_ref = P.max_gap;
			const int rep_len = frag_rep[f]; bool tie = false;
			const LdsCigS<CS> cig{s_cig};
			auto finish_reg = [&](const int qlen, const uint32_t *seq, const AlReg &Rin, const RegExt x, const AlAnchor *a) -> AlReg {
				AlReg R = Rin;
				const int32_t rid = R.rid, rev = (R.flags & ALR_REV) ? 1 : 0;
				const uint64_t ref_off = G.seq_off[rid];
				R.n_cigar = 0; R.dp_score = 0; R.dp_max = 0; R.dp_max2 = 0; R.n_ambi = 0;
				int32_t rs1 = x.rs, qs1 = x.qs, re1, qe1;
				++c_regs; c_ref += (unsigned long long)(x.re0 - x.rs0);
				if (E.jobs[x.job].qlen) {
					const ExtOut *po = &E.outs[x.job];
					const uint32_t fl = po->flags_ncig; const int nc = (int)(fl >> 8); const bool reach = fl & 1;
					if (nc > 0 && (nc <= 6 || po->cig_off != 0xffffffffu)) { d_fcig_append(&R, cig, nc, nc <= 6 ? po->cig : G.arena + po->cig_off); R.dp_score += po->max; }
					rs1 = x.rs - (reach ? po->mqe_t + 1 : po->max_t + 1);
					qs1 = x.qs - (reach ? x.qs : po->max_q + 1);
				}
				{ const uint32_t m = (uint32_t)(x.qe - x.qs) << 4; d_fcig_append(&R, cig, 1, &m); R.dp_score += x.core_score; }
				re1 = x.re; qe1 = x.qe;
				if (E.jobs[x.job + 1].qlen) {
					const ExtOut *po = &E.outs[x.job + 1];
					const uint32_t fl = po->flags_ncig; const int nc = (int)(fl >> 8); const bool reach = fl & 1;
					if (nc > 0 && (nc <= 6 || po->cig_off != 0xffffffffu)) { d_fcig_append(&R, cig, nc, nc <= 6 ? po->cig : G.arena + po->cig_off); R.dp_score += po->max; }
					re1 = x.re + (reach ? po->mqe_t + 1 : po->max_t + 1);
					qe1 = x.qe + (reach ? qlen - x.qe : po->max_q + 1);
				}
				R.rs = rs1; R.re = re1;
				if (rev) { R.qs = qlen - qe1; R.qe = qlen - qs1; } else { R.qs = qs1; R.qe = qe1; }
				d_update_extra(P, &R, cig, ReadAcc{seq, qlen, rev, qs1}, RefAcc{G.S4, ref_off + (uint64_t)rs1});
				c_cig += R.n_cigar;
				uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
				if (R.n_cigar <= 4) {
					if (R.n_cigar > 0) c0 = cig[0]; if (R.n_cigar > 1) c1 = cig[1]; if (R.n_cigar > 2) c2 = cig[2]; if (R.n_cigar > 3) c3 = cig[3];
					R.cigar_off = AL_CIG_INLINE;
				} else {
					const unsigned long long off = atomicAdd(G.arena_cnt, (unsigned long long)R.n_cigar);
					if (off + R.n_cigar <= G.arena_cap) { for (uint32_t k = 0; k < R.n_cigar; ++k) G.arena[off + k] = cig[k]; R.cigar_off = (uint32_t)off; }
					else { atomicAdd(&G.counters[9], 1ULL); R.cigar_off = 0xffffffffu; }
				}
				R.cig_inl[0] = c0; R.cig_inl[1] = c1; R.cig_inl[2] = c2; R.cig_inl[3] = c3;
				return R;
			};
			auto do_seg = [&](const uint32_t s, const int qlen, AlReg *regs, const AlAnchor *a) -> int {
				int n = (int)W.reg_cnt[r0 + s];
				const uint32_t *seq = rd_seq + rd_off[r0 + s];
				for (int i = WAVE ? lane : 0; i < n; i += WAVE ? 64 : 1) {       // mm_align1 after the DP calls (align.c:698-788); wavefront form: a lane per hit
					if (regs[i].cnt == 0 || (regs[i].flags & ALR_HAS_P)) continue;
					regs[i] = finish_reg(qlen, seq, regs[i], E.rext[B2 + (uint64_t)s * fw.cap + i], a);
				}
				if (WAVE) { __threadfence_block(); if (lane != 0) return 0; }     // the hit-level bookkeeping below is one lane's
				d_filter_regs(P, qlen, &n, regs);                                // align.c:910-911
				tie = d_hit_sort(&n, regs, fw.aux128, fw.rtmp) || tie;
				d_set_parent(P.mask_level, n, regs, P.a * 2 + P.b, fw.aux64, fw.auxi);
				d_select_sub(P.pri_ratio, P.k * 2, P.best_n, &n, regs, fw.auxi);
				d_set_sam_pri(n, regs);
				d_set_mapq(n, regs, P.min_chain_score, P.a, rep_len, lt);
				return n;
			};
			AlReg *const mreg0 = fw.mreg[0], *const mreg1 = fw.mreg[1];
			const AlAnchor *const sa0 = fw.seg_a[0], *const sa1 = n_segs == 2 ? fw.seg_a[0] + W.seg_na[r0] : nullptr;
			if (!WAVE && n_segs == 2 && W.reg_cnt[r0] == 1 && W.reg_cnt[r0 + 1] == 1 && mreg0[0].cnt != 0 && mreg1[0].cnt != 0 && !((P.dbg >> 19) & 1)) {
				// One hit per mate (the common fragment): both records stay in registers from the DP results to the final store;
				// the post-DP bookkeeping of a single hit (filter, parent = self, sam_pri, MAPQ) and the 1 x 1 pairing need no
				// scratch arrays.  Same result as the general code below.
				AlReg R0 = mreg0[0], R1 = mreg1[0];
				if (!(R0.flags & ALR_HAS_P)) R0 = finish_reg(ql0, rd_seq + rd_off[r0], R0, E.rext[B2], sa0);
				if (!(R1.flags & ALR_HAS_P)) R1 = finish_reg(ql1, rd_seq + rd_off[r0 + 1], R1, E.rext[B2 + (uint64_t)fw.cap], sa1);
				auto post1 = [&](AlReg &R, const int qlen) -> int {
					int n1 = 1;
					d_filter_regs(P, qlen, &n1, &R);                              // align.c:910-911
					if (n1 == 0) return 0;
					R.id = 0; R.parent = 0;                                       // mm_set_parent / mm_select_sub with one hit
					d_set_sam_pri(1, &R);
					d_set_mapq(1, &R, P.min_chain_score, P.a, rep_len, lt);
					return 1;
				};
				const int k0 = post1(R0, ql0), k1 = post1(R1, ql1);
				if (P.pe_ori >= 0 && k0 && k1) {
					if (sc_off[f + 1] - sc_off[f] < 1) atomicAdd(&G.counters[7], 1ULL << 56);
					d_pair11(P, max_gap_ref, ql0, ql1, R0, R1);
				}
				if (k0) mreg0[0] = R0;
				if (k1) mreg1[0] = R1;
				W.reg_cnt[r0] = (uint32_t)k0; W.reg_cnt[r0 + 1] = (uint32_t)k1;
			} else {
			const int nr0 = do_seg(0, ql0, mreg0, sa0);
			const int nr1 = n_segs == 2 ? do_seg(1, ql1, mreg1, sa1) : 0;
			if (WAVE && lane != 0) return;
			if (n_segs == 2 && P.pe_ori >= 0) {
				bool ovf = false;
				// pair scores: at most n0*n1 entries (bounded by the hit counts after k_regs; see k_ext_counts)
				d_pair2(P, max_gap_ref, ql0, ql1, nr0, nr1, mreg0, mreg1, (PairEnt *)fw.rtmp, sc_ws + sc_off[f], (int)(sc_off[f + 1] - sc_off[f]), lt, &tie, &ovf);
				if (ovf) { atomicAdd(&G.counters[7], 1ULL << 56); }
			}
			W.reg_cnt[r0] = (uint32_t)nr0;
			if (n_segs == 2) W.reg_cnt[r0 + 1] = (uint32_t)nr1;
			}
			if (tie) atomicAdd(&G.counters[10], 1ULL);
		}
}

extern "C" __global__ void __launch_bounds__(256, AL_LB_FIN)
k_ext_finish(const uint32_t *__restrict__ rd_seq, const uint64_t *__restrict__ rd_off, const uint32_t *__restrict__ rd_len,
             const uint32_t *__restrict__ frag_first, const int32_t *__restrict__ frag_rep, WsBase W, AlignShared G, ExtShared E,
             AlLogTab lt, uint64_t *__restrict__ sc_ws, const uint64_t *__restrict__ sc_off, int n_frag, AlParams P, uint32_t *__restrict__ slow_list, uint32_t *__restrict__ n_slow,
             int early_done /* fragments k_ext_prep marked slow are already with the monolithic kernel (side stream) */, const uint32_t *__restrict__ order /* as k_ext_prep */,
             int heavy_jobs /* fragments with at least this many job slots are k_ext_finish_wave's (0: none) */)
{
	const int t_ = blockIdx.x * blockDim.x + threadIdx.x;
	const int f = t_ < n_frag ? (order ? (int)order[t_] : t_) : n_frag;
	__shared__ uint32_t s_cig[AL_FCIG * 256];                                  // per-lane CIGAR assembly buffer, [word][lane]
	unsigned long long c_regs = 0, c_ref = 0, c_cig = 0;
	if (f < n_frag && W.frag_nu[f] != 0 && !(early_done && E.frag_slow[f] != 0) && !(heavy_jobs > 0 && (int)(E.job_off[f + 1] - E.job_off[f]) >= heavy_jobs))
		d_ext_finish_frag<false, 256>(f, 0, s_cig + threadIdx.x, rd_seq, rd_off, rd_len, frag_first, frag_rep, W, G, E, lt, sc_ws, sc_off, P, slow_list, n_slow, c_regs, c_ref, c_cig);
	for (int d = 32; d > 0; d >>= 1) { c_regs += __shfl_xor(c_regs, d); c_ref += __shfl_xor(c_ref, d); c_cig += __shfl_xor(c_cig, d); }
	if ((threadIdx.x & 63) == 0) { if (c_regs) atomicAdd(&G.counters[4], c_regs); if (c_ref) atomicAdd(&G.counters[5], c_ref); if (c_cig) atomicAdd(&G.counters[6], c_cig); }
}

// the fragments at the head of the hits-descending order, a wavefront each
extern "C" __global__ void __launch_bounds__(64)
k_ext_finish_wave(const uint32_t *__restrict__ rd_seq, const uint64_t *__restrict__ rd_off, const uint32_t *__restrict__ rd_len,
                  const uint32_t *__restrict__ frag_first, const int32_t *__restrict__ frag_rep, WsBase W, AlignShared G, ExtShared E,
                  AlLogTab lt, uint64_t *__restrict__ sc_ws, const uint64_t *__restrict__ sc_off, int n_frag, AlParams P, uint32_t *__restrict__ slow_list, uint32_t *__restrict__ n_slow,
                  int early_done, const uint32_t *__restrict__ order, int heavy_jobs)
{
	__shared__ uint32_t s_cig[AL_FCIG * 64];
	unsigned long long c_regs = 0, c_ref = 0, c_cig = 0;
	for (int t = blockIdx.x; t < n_frag; t += gridDim.x) {
		const int f = (int)order[t];
		if ((int)(E.job_off[f + 1] - E.job_off[f]) < heavy_jobs) break;     // (descending order: nothing heavy behind it)
		if (W.frag_nu[f] == 0 || (early_done && E.frag_slow[f] != 0)) continue;
		d_ext_finish_frag<true, 64>(f, (int)threadIdx.x, s_cig + threadIdx.x, rd_seq, rd_off, rd_len, frag_first, frag_rep, W, G, E, lt, sc_ws, sc_off, P, slow_list, n_slow, c_regs, c_ref, c_cig);
		__syncthreads();
	}
	for (int d = 32; d > 0; d >>= 1) { c_regs += __shfl_xor(c_regs, d); c_ref += __shfl_xor(c_ref, d); c_cig += __shfl_xor(c_cig, d); }
	if (threadIdx.x == 0) { if (c_regs) atomicAdd(&G.counters[4], c_regs); if (c_ref) atomicAdd(&G.counters[5], c_ref); if (c_cig) atomicAdd(&G.counters[6], c_cig); }
}

extern "C" __global__ void __launch_bounds__(256)
k_ext_counts(const uint32_t *__restrict__ frag_first, WsBase W, uint32_t *__restrict__ n_jobs, uint32_t *__restrict__ n_sc, int n_frag)
{
	const int f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f > n_frag) return;
	if (f == n_frag) { n_jobs[f] = 0; n_sc[f] = 0; return; }
	const uint32_t r0 = frag_first[f], n_segs = frag_first[f + 1] - r0;
	uint32_t c0 = 0, c1 = 0;
	if (W.frag_nu[f] != 0) { c0 = W.reg_cnt[r0]; if (n_segs == 2) c1 = W.reg_cnt[r0 + 1]; }
	n_jobs[f] = 2 * (c0 + c1); n_sc[f] = c0 * c1;
}
extern "C" __global__ void __launch_bounds__(256) k_iota(uint32_t *a, uint32_t n) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) a[i] = i; }

// compaction of the per-mate hit arrays into one dense output array
extern "C" __global__ void __launch_bounds__(256)
k_compact(const uint32_t *__restrict__ frag_first, WsBase W, const uint64_t *__restrict__ out_off, AlReg *__restrict__ out, int n_frag)
{
	const int f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= n_frag) return;
	const uint32_t r0 = frag_first[f], n_segs = frag_first[f + 1] - r0;
	if (W.frag_nu[f] == 0) return;
	FragWs fw; d_frag_ws(W, (uint32_t)f, fw);
	for (uint32_t s = 0; s < n_segs; ++s) {
		const uint32_t n = W.reg_cnt[r0 + s]; AlReg *o = out + out_off[r0 + s];
		for (uint32_t i = 0; i < n; ++i) o[i] = fw.mreg[s][i];
	}
}

// ---------------------------------------------------------------------------------------------
// host side of the stage
struct CastU64b { __host__ __device__ uint64_t operator()(const uint32_t &v) const { return (uint64_t)v; } };
static int scan32(al_ctx_t *c, const uint32_t *in, uint64_t *out, int n)
{
	auto it = rocprim::make_transform_iterator((const uint32_t *)in, CastU64b());
	size_t bytes = 0;
	AL_HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, it, out, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>(), c->stream));
	if (c->scan_tmp.ensure(bytes + 16)) return -1;
	AL_HIP_CHECK(rocprim::exclusive_scan(c->scan_tmp.p, bytes, it, out, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>(), c->stream));
	return 0;
}

__global__ void k_lower_bounds(const uint32_t *, uint32_t, LbThr, uint32_t *);   // al_kernels_seed.hip
struct AlignState {            // lives in al_ctx_s::align_state (opaque there)
	DevBuf<AlReg> regs0, mregs, rtmp, out;
	DevBuf<AlAnchor> aux128, seg_a;
	DevBuf<uint64_t> aux64, seg_u, nu_off, out_off;
	DevBuf<int32_t> auxi;
	DevBuf<uint32_t> reg_cnt, seg_na, arena, seg_fast, regs_n0, cap2; DevBuf<uint64_t> b2_off;
	uint64_t arena_scale = 1;       // doubled by al_align_grow_arena() when a batch's long CIGARs overflowed the arena
	DevBuf<uint8_t> gws, gws2, long_state;
	DevBuf<float> logtab;
	DevBuf<unsigned long long> dbgbuf, hist;
	DevBuf<ExtJob> jobs; DevBuf<ExtOut> outs; DevBuf<RegExt> rext;
	DevBuf<uint64_t> job_off, sc_off, sc_ws; DevBuf<uint32_t> n_jobs, n_sc, job_key, job_key2, job_idx, job_idx2, frag_slow, slow_list, early_list;
	DevBuf<uint8_t> sort_tmp;
	DevBuf<uint32_t> heavy_list;                 // k_regs_heavy: the candidates by tile size, five lists
	DevBuf<uint32_t> ford_key, ford_idx, ford;   // fragments ordered by their number of hits (k_ext_prep / k_ext_finish)
	int logtab_a = -1, logtab_n = 0;
	uint64_t out_total = 0;
};
static std::map<al_ctx_t *, AlignState *> g_states;
static std::mutex g_states_mtx;
static AlignState *get_state(al_ctx_t *c)
{
	std::lock_guard<std::mutex> lk(g_states_mtx);
	auto it = g_states.find(c);
	if (it != g_states.end()) return it->second;
	AlignState *s = new AlignState(); g_states[c] = s; return s;
}
void al_align_state_free(al_ctx_t *c)
{
	std::lock_guard<std::mutex> lk(g_states_mtx);
	auto it = g_states.find(c);
	if (it == g_states.end()) return;
	AlignState *s = it->second;
	s->regs0.release(); s->mregs.release(); s->rtmp.release(); s->out.release(); s->aux128.release(); s->seg_a.release(); s->aux64.release(); s->seg_u.release();
	s->nu_off.release(); s->out_off.release(); s->auxi.release(); s->reg_cnt.release(); s->seg_na.release(); s->arena.release(); s->gws.release(); s->long_state.release(); s->logtab.release(); s->dbgbuf.release(); s->hist.release(); s->jobs.release(); s->outs.release(); s->rext.release(); s->seg_fast.release(); s->regs_n0.release(); s->cap2.release(); s->b2_off.release();
	s->job_off.release(); s->sc_off.release(); s->sc_ws.release(); s->n_jobs.release(); s->n_sc.release(); s->job_key.release(); s->job_key2.release(); s->job_idx.release(); s->job_idx2.release(); s->frag_slow.release(); s->slow_list.release(); s->early_list.release(); s->gws2.release(); s->sort_tmp.release(); s->ford_key.release(); s->ford_idx.release(); s->ford.release(); s->heavy_list.release();
	delete s; g_states.erase(it);
}

int al_run_align_stage(al_ctx_t *c)
{
	hipStream_t s = c->stream;
	AlignState *A = get_state(c);
	const int nf = c->n_frag, nr = c->n_reads;
	if (nf == 0) { for (int i = ST_REGS; i < ST_COMPACT; ++i) AL_HIP_CHECK(hipEventRecord(c->ev[i + 1], s)); return 0; }
	// logf table from the HOST libm (the reference's logf is glibc's): logf((float)k / a) and logf((float)k)
	const int Lmax0 = c->max_rd_len;
	const int log_n = std::max(AL_LOGTAB_N, (std::max(c->opt.a, 1) * 2 * Lmax0 + 1024 + 4095) / 4096 * 4096);   // dp_max <= match_sc * read length; room to spare
	if (A->logtab_a != c->opt.a || A->logtab_n < log_n) {
		std::vector<float> h(2 * (size_t)log_n);
		for (int k = 0; k < log_n; ++k) { h[k] = logf((float)k / c->opt.a); h[log_n + k] = logf((float)k); }
		if (A->logtab.ensure(2 * (size_t)log_n)) return -1;
		AL_HIP_CHECK(hipMemcpyAsync(A->logtab.p, h.data(), h.size() * 4, hipMemcpyHostToDevice, s));
		AL_HIP_CHECK(hipStreamSynchronize(s));
		A->logtab_a = c->opt.a; A->logtab_n = log_n;
	}
	// workspace sizes from the number of chains
	if (A->nu_off.ensure(nf + 2)) return -1;
	AL_HIP_CHECK(hipMemsetAsync(c->frag_nu.p + nf, 0, 4, s));
	if (scan32(c, c->frag_nu.p, A->nu_off.p, nf)) return -1;
	uint64_t nu_total = 0;
	AL_HIP_CHECK(hipMemcpyAsync(&nu_total, A->nu_off.p + nf, 8, hipMemcpyDeviceToHost, s));
	AL_HIP_CHECK(hipStreamSynchronize(s));
	const uint64_t Btot = 4 * nu_total + 4ULL * nf + 8;
	if ((!c->no_taps && c->seg_a.ensure(c->n_anchor_total + 1)) || A->regs0.ensure(nu_total + 1) || A->aux128.ensure(Btot) || A->aux64.ensure(Btot) || A->auxi.ensure(2 * Btot) ||
	    A->seg_u.ensure(2 * nu_total + 2) || A->reg_cnt.ensure(nr + 1) || A->seg_na.ensure(nr + 1) || A->out_off.ensure(nr + 2) || A->seg_fast.ensure(nr + 1) || A->cap2.ensure(nf + 2) || A->b2_off.ensure(nf + 2)) return -1;
	{   // AL_TEST_SCRUB=<byte> (tests): the stage's work areas -- and the chaining scratch it reuses -- filled with that byte first: a result
		// that depends on what an earlier stage or batch left there shows up as a difference between two byte values
		static const char *scrub = getenv("AL_TEST_SCRUB");
		if (scrub) {
			const int v = atoi(scrub);
			AL_HIP_CHECK(hipMemsetAsync(A->regs0.p, v, A->regs0.cap * sizeof(AlReg), s)); AL_HIP_CHECK(hipMemsetAsync(A->aux128.p, v, A->aux128.cap * sizeof(AlAnchor), s));
			AL_HIP_CHECK(hipMemsetAsync(A->aux64.p, v, A->aux64.cap * 8, s)); AL_HIP_CHECK(hipMemsetAsync(A->auxi.p, v, A->auxi.cap * 4, s)); AL_HIP_CHECK(hipMemsetAsync(A->seg_u.p, v, A->seg_u.cap * 8, s));
			if (!c->no_taps) AL_HIP_CHECK(hipMemsetAsync(c->seg_a.p, v, c->seg_a.cap * sizeof(AlAnchor), s));
			AL_HIP_CHECK(hipMemsetAsync(A->reg_cnt.p, v, A->reg_cnt.cap * 4, s)); AL_HIP_CHECK(hipMemsetAsync(A->seg_na.p, v, A->seg_na.cap * 4, s)); AL_HIP_CHECK(hipMemsetAsync(A->seg_fast.p, v, A->seg_fast.cap * 4, s));
		}
	}
	WsBase W;
	W.regs0 = A->regs0.p; W.aux128 = A->aux128.p; W.seg_a = c->no_taps ? c->anchors.p : c->seg_a.p; W.aux64 = A->aux64.p; W.seg_u = A->seg_u.p; W.auxi = A->auxi.p;
	W.nu_off = A->nu_off.p; W.frag_nu = c->frag_nu.p; W.a_off = c->a_off.p; W.reg_cnt = A->reg_cnt.p; W.seg_na = A->seg_na.p; W.seg_fast = A->seg_fast.p; W.cap2 = nullptr; W.b2_off = nullptr; W.mregs = nullptr; W.rtmp = nullptr; W.rext = nullptr;
	// fragments with many chains: chain_post by a wavefront each (k_regs_select), by chain-count class
	uint32_t *regs_n0 = nullptr; uint32_t heavy_from = 0, heavy_n = 0; uint64_t Btot2 = 0;
	if (!((c->P.dbg >> 19) & 1)) {
		if (A->regs_n0.ensure(nf + 1) || c->chain_key.ensure(nf + 1) || c->chain_idx.ensure(nf + 1) || c->chain_idx2.ensure(nf + 1) || c->lb_buf.ensure(16)) return -1;
		regs_n0 = A->regs_n0.p;
		AL_HIP_CHECK(hipMemsetAsync(regs_n0, 0xff, (size_t)nf * 4, s));
		hipLaunchKernelGGL(k_iota, dim3((nf + 255) / 256), dim3(256), 0, s, c->chain_idx.p, (uint32_t)nf);
		size_t bytes = 0;
		AL_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t *)c->frag_nu.p, c->chain_key.p, (const uint32_t *)c->chain_idx.p, c->chain_idx2.p, nf, 0, 32, s));
		if (c->scan_tmp.ensure(bytes + 16)) return -1;
		AL_HIP_CHECK(rocprim::radix_sort_pairs(c->scan_tmp.p, bytes, (const uint32_t *)c->frag_nu.p, c->chain_key.p, (const uint32_t *)c->chain_idx.p, c->chain_idx2.p, nf, 0, 32, s));
		uint32_t init[7] = {(uint32_t)nf, (uint32_t)nf, (uint32_t)nf, (uint32_t)nf, (uint32_t)nf, (uint32_t)nf, (uint32_t)nf}, lb[7];
		AL_HIP_CHECK(hipMemcpyAsync(c->lb_buf.p, init, 28, hipMemcpyHostToDevice, s));
		LbThr T; T.n = 7; T.v[0] = 5; T.v[1] = 65; T.v[2] = 1025; T.v[3] = 8193; T.v[4] = 2049; T.v[5] = 4097; T.v[6] = 257; for (int i = 7; i < 16; ++i) T.v[i] = 0xffffffffu;
		hipLaunchKernelGGL(k_lower_bounds, dim3((nf + 255) / 256), dim3(256), 0, s, (const uint32_t *)c->chain_key.p, (uint32_t)nf, T, c->lb_buf.p);
		AL_HIP_CHECK(hipMemcpyAsync(lb, c->lb_buf.p, 28, hipMemcpyDeviceToHost, s));
		AL_HIP_CHECK(hipStreamSynchronize(s));
		const uint32_t *ord = c->chain_idx2.p;
		T.n = 1; T.v[0] = 9;      // fragments with at least nine chains may keep at least nine hits: candidates of k_regs_heavy
		{ uint32_t i9 = (uint32_t)nf; AL_HIP_CHECK(hipMemcpyAsync(c->lb_buf.p, &i9, 4, hipMemcpyHostToDevice, s)); hipLaunchKernelGGL(k_lower_bounds, dim3((nf + 255) / 256), dim3(256), 0, s, (const uint32_t *)c->chain_key.p, (uint32_t)nf, T, c->lb_buf.p);
		  AL_HIP_CHECK(hipMemcpyAsync(&i9, c->lb_buf.p, 4, hipMemcpyDeviceToHost, s)); AL_HIP_CHECK(hipStreamSynchronize(s)); heavy_from = i9; heavy_n = (uint32_t)nf - i9; }
		// The classes are disjoint sets of fragments.  Two fill the chip (65 ... 256 and 257 ... 1024 chains: hundreds of thousands of blocks); the
		// others are a few hundred to a few thousand blocks that sort for a millisecond: on the side streams, beside the two.  4097 ... 8192 chains: the sort tile is 80 KB; its sort
		// leaves keys and order in the work area for a one-wavefront pass (k_regs_select<-2>).
		static const int split = getenv("AL_REGS_SPLIT") ? atoi(getenv("AL_REGS_SPLIT")) : 1;   // bit 0: 257 ... 1024 chains sorted and passed over by two kernels as well (4.6 + 2.2 ms against 7.7 in one: the pass alone fits twelve blocks a CU)
#define LSEL(CAPV, PH, NT, A, B, ST) do { if ((B) > (A)) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_regs_select<CAPV, PH>), dim3((B) - (A)), dim3(NT), 0, ST, c->chained.p, c->u.p, c->uo.p, c->frag_first.p, c->rd_len.p, c->frag_hash.p, W, ord + (A), (int)((B) - (A)), c->P, regs_n0); } while (0)
		AL_HIP_CHECK(hipEventRecord(c->ev_fj[0], s));
		for (int i = 0; i < 3; ++i) AL_HIP_CHECK(hipStreamWaitEvent(c->aux[i], c->ev_fj[0], 0));
		// (round 6) the 4097 ... 8192 class on a side stream BESIDE the chip-filling classes (its sort, then its one-wavefront pass k_regs_select<-2>), launched first:
		// 6.5 ms beside the others instead of 4.3 ms before them (regs 16.5 -> 15.7 ms; round 4 had measured 20 ms beside them, with the sort and the pass in one kernel)
		LSEL(8192, 1, 512, lb[5], lb[3], c->aux[0]);
		LSEL(-2, 0, 64, lb[5], lb[3], c->aux[0]);
		LSEL(4096, 0, 256, lb[4], lb[5], c->aux[1]);
		LSEL(-1, 0, 1024, lb[3], (uint32_t)nf, c->aux[1]);
		LSEL(2048, 0, 256, lb[2], lb[4], c->aux[2]);
		if (split & 1) { LSEL(1024, 1, 256, lb[6], lb[2], s); LSEL(-2, 0, 64, lb[6], lb[2], s); } else LSEL(1024, 0, 256, lb[6], lb[2], s);
		LSEL(256, 0, 64, lb[1], lb[6], s);
		LSEL(0, 0, 64, lb[0], lb[1], s);
#undef LSEL
		for (int i = 0; i < 3; ++i) { AL_HIP_CHECK(hipEventRecord(c->ev_aux[i], c->aux[i])); AL_HIP_CHECK(hipStreamWaitEvent(s, c->ev_aux[i], 0)); }
	}
	uint32_t hv_cnt[5] = {0, 0, 0, 0, 0};
	const HeavyCls HT{{12, 24, 48, 72, 200}, {256, 512, 768, 1024, 2048}};
	if (regs_n0 && heavy_n > 0) {
		if (A->heavy_list.ensure((size_t)5 * heavy_n + 8)) return -1;
		uint32_t *const cnt_d = (uint32_t *)(c->counters.p + 20);               // (counters[20..22]: five 32-bit counts)
		AL_HIP_CHECK(hipMemsetAsync(cnt_d, 0, 24, s));
		hipLaunchKernelGGL(k_regs_heavy_classify, dim3((heavy_n + 255) / 256), dim3(256), 0, s, W, (const uint32_t *)c->chain_idx2.p + heavy_from, (int)heavy_n, (const uint32_t *)regs_n0, HT, A->heavy_list.p, cnt_d);
		AL_HIP_CHECK(hipMemcpyAsync(hv_cnt, cnt_d, 20, hipMemcpyDeviceToHost, s));   // (read with the total below: one synchronisation)
	}
	{   // room for the per-mate hits, from what chain_post kept
		hipLaunchKernelGGL(k_regs_cap2, dim3((nf + 256) / 256), dim3(256), 0, s, (const uint32_t *)c->frag_nu.p, (const uint32_t *)regs_n0, nf, A->cap2.p);
		if (scan32(c, A->cap2.p, A->b2_off.p, nf)) return -1;
		uint64_t b2_total = 0;
		AL_HIP_CHECK(hipMemcpyAsync(&b2_total, A->b2_off.p + nf, 8, hipMemcpyDeviceToHost, s));
		AL_HIP_CHECK(hipStreamSynchronize(s));
		Btot2 = b2_total + 8;
		if (A->mregs.ensure(2 * Btot2) || A->rtmp.ensure(Btot2) || A->rext.ensure(2 * Btot2 + 1)) return -1;
		{ static const char *scrub = getenv("AL_TEST_SCRUB");
		  if (scrub) { const int v = atoi(scrub); AL_HIP_CHECK(hipMemsetAsync(A->mregs.p, v, A->mregs.cap * sizeof(AlReg), s)); AL_HIP_CHECK(hipMemsetAsync(A->rtmp.p, v, A->rtmp.cap * sizeof(AlReg), s)); AL_HIP_CHECK(hipMemsetAsync(A->rext.p, v, A->rext.cap * sizeof(RegExt), s)); } }
		W.mregs = A->mregs.p; W.rtmp = A->rtmp.p; W.rext = A->rext.p; W.cap2 = A->cap2.p; W.b2_off = A->b2_off.p;
	}
	int regs_part = 0;
	if (regs_n0 && heavy_n > 0) {
		// Five tile sizes; every block takes its fragment only if the kept hits fit its tiles and not the next smaller instance's, so the
		// instances work on disjoint fragments and run side by side.  The code is one lane on LDS copies: what sets the rate is how many
		// wavefronts a CU holds, i.e. the tile (17 KB: 9 per CU ... 141 KB: one).
		const size_t lds_a = al_regs_heavy_lds(12, 256), lds_t = al_regs_heavy_lds(24, 512), lds_b = al_regs_heavy_lds(48, 768), lds_s = al_regs_heavy_lds(72, 1024), lds_l = al_regs_heavy_lds(200, 2048);
		if (!c->attr_regs_heavy) {
			AL_HIP_CHECK(hipFuncSetAttribute((const void *)k_regs_heavy<200, 2048, 72, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_l));
			AL_HIP_CHECK(hipFuncSetAttribute((const void *)k_regs_heavy<72, 1024, 48, 768>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
			c->attr_regs_heavy = true;
		}
		hipStream_t sd = c->side;
		AL_HIP_CHECK(hipEventRecord(c->ev_fj[0], s)); AL_HIP_CHECK(hipStreamWaitEvent(sd, c->ev_fj[0], 0));
#define LHV(K, RC, AC, RCL, ACL, LDS, ST) do { if (hv_cnt[K] > 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_regs_heavy<RC, AC, RCL, ACL>), dim3(hv_cnt[K]), dim3(64), LDS, ST, c->chained.p, c->frag_first.p, c->rd_len.p, c->frag_hash.p, W, (const uint32_t *)A->heavy_list.p + (size_t)(K) * heavy_n, (int)hv_cnt[K], c->P, c->counters.p, regs_n0); } while (0)
		for (int i = 0; i < 3; ++i) AL_HIP_CHECK(hipStreamWaitEvent(c->aux[i], c->ev_fj[0], 0));
		LHV(3, 72, 1024, 48, 768, lds_s, sd); LHV(2, 48, 768, 24, 512, lds_b, c->aux[0]); LHV(4, 200, 2048, 72, 1024, lds_l, c->aux[1]); LHV(0, 12, 256, 0, 0, lds_a, c->aux[2]);
		// (round 6) every fragment that is not one of these kernels' candidates: k_regs beside them, behind the shortest class (1.8 ms that used to follow the longest, 4 ms)
		hipLaunchKernelGGL(k_regs, dim3((nf + 255) / 256), dim3(256), 0, c->aux[2], c->chained.p, c->u.p, c->uo.p, c->frag_first.p, c->rd_len.p, c->frag_hash.p, W, nf, c->P, c->counters.p, (const uint32_t *)regs_n0, 1);
		regs_part = 2;
		AL_HIP_CHECK(hipEventRecord(c->ev_fj[1], sd));
		LHV(1, 24, 512, 12, 256, lds_t, s);                                       // (the fifth tile size on the main stream, beside the other four)
		for (int i = 0; i < 3; ++i) { AL_HIP_CHECK(hipEventRecord(c->ev_aux[i], c->aux[i])); AL_HIP_CHECK(hipStreamWaitEvent(s, c->ev_aux[i], 0)); }
#undef LHV
		AL_HIP_CHECK(hipStreamWaitEvent(s, c->ev_fj[1], 0));
	}
	hipLaunchKernelGGL(k_regs, dim3((nf + 255) / 256), dim3(256), 0, s, c->chained.p, c->u.p, c->uo.p, c->frag_first.p, c->rd_len.p, c->frag_hash.p, W, nf, c->P, c->counters.p, (const uint32_t *)regs_n0, regs_part);
	if (getenv("AL_DBG_FRAG")) {   // debugging aid: the chain_post result of one fragment (the kept hits at the front of its regs0 range)
		const int f = atoi(getenv("AL_DBG_FRAG"));
		if (f >= 0 && f < nf) {
			uint64_t o[2]; uint32_t n0v = 0, nu = 0;
			AL_HIP_CHECK(hipMemcpy(o, A->nu_off.p + f, 16, hipMemcpyDeviceToHost)); AL_HIP_CHECK(hipMemcpy(&nu, c->frag_nu.p + f, 4, hipMemcpyDeviceToHost));
			if (regs_n0) AL_HIP_CHECK(hipMemcpy(&n0v, regs_n0 + f, 4, hipMemcpyDeviceToHost));
			const int show = (int)std::min<uint64_t>(o[1] - o[0], 40);
			std::vector<AlReg> h(show);
			AL_HIP_CHECK(hipMemcpy(h.data(), A->regs0.p + o[0], (size_t)show * sizeof(AlReg), hipMemcpyDeviceToHost));
			fprintf(stderr, "[airlift] dbg frag %d: %u chains, regs_n0 %x\n", f, nu, n0v);
			for (int i = 0; i < show; ++i) fprintf(stderr, "  [%d] id %d parent %d score %d cnt %d as %d hash %08x qs %d qe %d rs %d re %d subsc %d n_sub %d flags %x\n", i, h[i].id, h[i].parent, h[i].score, h[i].cnt, h[i].as, h[i].hash, h[i].qs, h[i].qe, h[i].rs, h[i].re, h[i].subsc, h[i].n_sub, h[i].flags);
			// the per-mate hit lists chain_post / seg_gen left for the extension stage
			uint64_t b2 = 0; uint32_t cap2 = 0, r0 = 0, rc[2] = {0, 0};
			AL_HIP_CHECK(hipMemcpy(&b2, A->b2_off.p + f, 8, hipMemcpyDeviceToHost)); AL_HIP_CHECK(hipMemcpy(&cap2, A->cap2.p + f, 4, hipMemcpyDeviceToHost)); AL_HIP_CHECK(hipMemcpy(&r0, c->frag_first.p + f, 4, hipMemcpyDeviceToHost));
			AL_HIP_CHECK(hipMemcpy(rc, A->reg_cnt.p + r0, 8, hipMemcpyDeviceToHost));
			for (int m = 0; m < 2; ++m) {
				const int nn = (int)std::min<uint32_t>(rc[m], 30u); std::vector<AlReg> g(nn + 1);
				if (nn) AL_HIP_CHECK(hipMemcpy(g.data(), A->mregs.p + 2 * b2 + (uint64_t)m * cap2, (size_t)nn * sizeof(AlReg), hipMemcpyDeviceToHost));
				for (int i = 0; i < nn; ++i) fprintf(stderr, "  mate %d [%d] id %d parent %d score %d cnt %d as %d hash %08x qs %d qe %d rs %d re %d subsc %d n_sub %d\n", m, i, g[i].id, g[i].parent, g[i].score, g[i].cnt, g[i].as, g[i].hash, g[i].qs, g[i].qe, g[i].rs, g[i].re, g[i].subsc, g[i].n_sub);
			}
		}
	}
	if (regs_n0 && getenv("AL_TRACE")) {   // which fragments were left to the one-lane code?
		std::vector<uint32_t> h0(nf), hu(nf);
		AL_HIP_CHECK(hipMemcpyAsync(h0.data(), regs_n0, (size_t)nf * 4, hipMemcpyDeviceToHost, s)); AL_HIP_CHECK(hipMemcpyAsync(hu.data(), c->frag_nu.p, (size_t)nf * 4, hipMemcpyDeviceToHost, s));
		AL_HIP_CHECK(hipStreamSynchronize(s));
		uint64_t n_unset = 0, n_set = 0, n_done = 0; uint32_t mx_unset = 0, mx_set = 0, mx_set_nu = 0;
		for (int i = 0; i < nf; ++i) {
			if (h0[i] == AL_REGS_DONE) ++n_done;
			else if (h0[i] == AL_REGS_UNSET || AL_REGS_BAIL(h0[i])) { if (hu[i] >= 5) { ++n_unset; mx_unset = std::max(mx_unset, hu[i]); if (hu[i] > 64) fprintf(stderr, "[airlift] trace: regs: fragment %d with %u chains not selected, code %x\n", i, hu[i], h0[i]); } }
			else { ++n_set; if (h0[i] > mx_set) { mx_set = h0[i]; mx_set_nu = hu[i]; } }
		}
		fprintf(stderr, "[airlift] trace: regs: %llu fragments finished by k_regs_heavy; left to k_regs: %llu after selection (most kept hits %u, of %u chains), %llu without selection (>= 5 chains; most chains %u)\n",
		        (unsigned long long)n_done, (unsigned long long)n_set, mx_set, mx_set_nu, (unsigned long long)n_unset, mx_unset);
	}
	AL_HIP_CHECK(hipEventRecord(c->ev[ST_REGS + 1], s));
	// extension stage geometry
	const int Lmax = c->max_rd_len;
	const al_mapopt_t &o = c->opt;
	int ext = Lmax; { int l = Lmax; ext = l + (l * o.a + o.end_bonus > o.q ? (l * o.a + o.end_bonus - o.q) / o.e : 0); }
	const int tbound = std::max(2 * Lmax + 16, ext + 16);
	const int bw = (int)(o.bw * 1.5 + 1.);
	const int ncol = ((std::min(Lmax, bw + 1) + 15) / 16 + 1);
	const size_t p_bytes = ((size_t)(Lmax + tbound) * ncol * 16 + 63) / 64 * 64;
	const size_t cig_words = ((size_t)(Lmax + tbound) + 16 + 15) / 16 * 16;
	const size_t stride = p_bytes + cig_words * 8 + AL_PAIR_SC_CAP * 8;
	int nb = (nf + AL_GPB - 1) / AL_GPB; const int nb_max = Lmax > 512 ? 16 : 256 * 16; if (nb > nb_max) nb = nb_max;   // reads beyond the LDS tiles: few groups (megabytes of traceback each)
	const uint64_t arena_cap = ((uint64_t)nr * 12 + 4096 + (uint64_t)c->n_bases / 8) * A->arena_scale;
	if (A->arena.ensure(arena_cap)) return -1;
	AlignShared G; G.S4 = c->di.S4; G.seq_off = c->di.seq_off; G.seq_len = c->di.seq_len; G.arena = A->arena.p; G.arena_cnt = c->counters.p + 11; G.arena_cap = arena_cap; G.counters = c->counters.p;
	if (A->dbgbuf.ensure(640)) return -1;
	AL_HIP_CHECK(hipMemsetAsync(A->dbgbuf.p, 0, 640 * 8, s));
	G.dbg = A->dbgbuf.p;
	AlLogTab lt; lt.t = A->logtab.p; lt.n = A->logtab_n; lt.miss = c->counters.p + 8;
	const int tmax = (Lmax <= 160 && tbound <= 336) ? 336 : (Lmax <= 256 && tbound <= 512) ? 512 : (Lmax <= 512 && tbound <= 1024) ? 1024 : 0;
	const int qmax = tmax == 336 ? 160 : tmax == 512 ? 256 : 512;
	const bool long_mode = tmax == 0;              // reads beyond the LDS tiles: the whole batch through k_align_long (state blocks in HBM)
	if (long_mode && (Lmax > AL_LONG_QMAX || tbound > AL_LONG_TMAX)) { fprintf(stderr, "[airlift] reads longer than %d bp (or an extension window longer than %d bp: read length + (read length * A + end bonus - O) / E + 16) are not supported by the device extension kernels (max read length in batch: %d, window %d)\n", AL_LONG_QMAX, AL_LONG_TMAX, Lmax, tbound); return -3; }
	if (long_mode) {   // few groups: each needs a state block of ~1.3 MB and a traceback area of (Lmax + window) * band columns bytes
		if (nb > 16) nb = 16;
		if (A->gws.ensure((size_t)nb * AL_GPB * stride + 64) || A->long_state.ensure((size_t)nb * AL_GPB * sizeof(GroupLong) + 64)) return -1;
	}
	auto launch_mono = [&](const uint32_t *list, int n_list, hipStream_t st, uint8_t *wsp, int nb) -> int {      // monolithic kernel (whole batch, or the slow-path list)
		int nbm = (n_list + AL_GPB - 1) / AL_GPB; if (nbm > nb) nbm = nb; if (nbm < 1) nbm = 1;
		// A short list (the z-drop fragments of a batch: tens) runs ONE fragment per wavefront: the four groups of a wavefront each walk their own fragment's
		// hits and DP calls, i.e. four divergent code paths executed one after the other -- and the stage waits for the slowest fragment (C5: 70 fragments,
		// 111 ms).  Sixteen lanes of a wavefront alone run it up to four times as fast; the chip has room for a few hundred such wavefronts.
		const bool solo = !long_mode && n_list > 0 && n_list <= nb * AL_GPB && n_list <= 512;
		const dim3 blk(solo ? GW : GW * AL_GPB);
		if (solo) nbm = n_list;
		if (long_mode) hipLaunchKernelGGL(k_align_long, dim3(nbm), dim3(GW * AL_GPB), 0, st, c->rd_seq.p, c->rd_off.p, c->rd_len.p, c->frag_first.p, c->frag_rep.p, W, G, lt, wsp, stride, p_bytes, cig_words, nf, c->P, list, n_list, (GroupLong *)A->long_state.p);
		else if (tmax == 336) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_align<336, 160>), dim3(nbm), blk, 0, st, c->rd_seq.p, c->rd_off.p, c->rd_len.p, c->frag_first.p, c->frag_rep.p, W, G, lt, wsp, stride, p_bytes, cig_words, nf, c->P, list, n_list);
		else if (tmax == 512) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_align<512, 256>), dim3(nbm), blk, 0, st, c->rd_seq.p, c->rd_off.p, c->rd_len.p, c->frag_first.p, c->frag_rep.p, W, G, lt, wsp, stride, p_bytes, cig_words, nf, c->P, list, n_list);
		else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_align<1024, 512>), dim3(nbm), blk, 0, st, c->rd_seq.p, c->rd_off.p, c->rd_len.p, c->frag_first.p, c->frag_rep.p, W, G, lt, wsp, stride, p_bytes, cig_words, nf, c->P, list, n_list);
		return 0;
	};
	if (((c->P.dbg >> 26) & 1) || long_mode) { if (A->gws.ensure((size_t)nb * AL_GPB * stride + 64)) return -1; if (launch_mono(nullptr, nf, s, A->gws.p, nb)) return -1; for (int i = ST_EXT_PREP; i <= ST_EXT_FINISH; ++i) AL_HIP_CHECK(hipEventRecord(c->ev[i + 1], s)); }   // AL_DBG bit 26: whole batch through the monolithic kernel
	else {
		// ---- fast path: prep -> size-sorted DP job queue -> finish -> (slow list) monolithic
		if (A->n_jobs.ensure(nf + 2) || A->n_sc.ensure(nf + 2) || A->job_off.ensure(nf + 2) || A->sc_off.ensure(nf + 2) || A->frag_slow.ensure(nf + 1) || A->slow_list.ensure(nf + 1) || A->hist.ensure(AL_HIST_N)) return -1;
		hipLaunchKernelGGL(k_ext_counts, dim3((nf + 256) / 256), dim3(256), 0, s, c->frag_first.p, W, A->n_jobs.p, A->n_sc.p, nf);
		// fragments by number of hits, descending: k_ext_prep / k_ext_finish give a lane to a fragment and a wavefront runs as long as its longest lane
		// (a read pair inside repeats keeps a primary and twenty secondaries per mate, most pairs one hit per mate)
		const uint32_t *frag_ord = nullptr;
		if (!((c->P.dbg >> 18) & 1)) {
			if (A->ford_key.ensure(nf + 1) || A->ford_idx.ensure(nf + 1) || A->ford.ensure(nf + 1)) return -1;
			hipLaunchKernelGGL(k_iota, dim3((nf + 255) / 256), dim3(256), 0, s, A->ford_idx.p, (uint32_t)nf);
			size_t bytes = 0;
			AL_HIP_CHECK(rocprim::radix_sort_pairs_desc(nullptr, bytes, (const uint32_t *)A->n_jobs.p, A->ford_key.p, (const uint32_t *)A->ford_idx.p, A->ford.p, nf, 0, 16, s));
			if (A->sort_tmp.ensure(bytes + 16)) return -1;
			AL_HIP_CHECK(rocprim::radix_sort_pairs_desc(A->sort_tmp.p, bytes, (const uint32_t *)A->n_jobs.p, A->ford_key.p, (const uint32_t *)A->ford_idx.p, A->ford.p, nf, 0, 16, s));
			frag_ord = A->ford.p;
		}
		if (scan32(c, A->n_jobs.p, A->job_off.p, nf) || scan32(c, A->n_sc.p, A->sc_off.p, nf)) return -1;
		uint64_t tot[2] = {0, 0};
		AL_HIP_CHECK(hipMemcpyAsync(&tot[0], A->job_off.p + nf, 8, hipMemcpyDeviceToHost, s));
		AL_HIP_CHECK(hipMemcpyAsync(&tot[1], A->sc_off.p + nf, 8, hipMemcpyDeviceToHost, s));
		AL_HIP_CHECK(hipStreamSynchronize(s));
		const uint32_t nj = (uint32_t)tot[0];
		if (A->jobs.ensure(nj + 1) || A->outs.ensure(nj + 1) || A->job_key.ensure(nj + 1) || A->job_key2.ensure(nj + 1) || A->job_idx.ensure(nj + 1) || A->job_idx2.ensure(nj + 1) ||
		    A->sc_ws.ensure(tot[1] + 1)) return -1;
		AL_HIP_CHECK(hipMemsetAsync(A->hist.p, 0, AL_HIST_N * 8, s));
		ExtShared E; E.jobs = A->jobs.p; E.outs = A->outs.p; E.rext = A->rext.p; E.job_off = A->job_off.p; E.frag_slow = A->frag_slow.p; E.job_key = A->job_key.p; E.hist = A->hist.p;
		// fragments with a dozen hits or more (the head of the hits-descending order): a wavefront each, a lane per hit, on the side stream beside the rest
		static const int heavy_env = getenv("AL_PREP_HEAVY") ? atoi(getenv("AL_PREP_HEAVY")) : AL_PREP_HEAVY;      // (0: every fragment on a lane)
		const int heavy_jobs = frag_ord ? heavy_env : 0;
		if (heavy_jobs > 0) {
			AL_HIP_CHECK(hipEventRecord(c->ev_fj[0], s)); AL_HIP_CHECK(hipStreamWaitEvent(c->side, c->ev_fj[0], 0));
			hipLaunchKernelGGL(k_ext_prep_wave, dim3(std::min(nf, 16384)), dim3(64), 0, c->side, c->rd_seq.p, c->rd_off.p, c->rd_len.p, c->frag_first.p, W, G, E, nf, c->P, tmax, qmax, frag_ord, heavy_jobs);
			AL_HIP_CHECK(hipEventRecord(c->ev_fj[1], c->side));
		}
		hipLaunchKernelGGL(k_ext_prep, dim3((nf + 255) / 256), dim3(256), 0, s, c->rd_seq.p, c->rd_off.p, c->rd_len.p, c->frag_first.p, W, G, E, nf, c->P, tmax, qmax, frag_ord, heavy_jobs);
		if (heavy_jobs > 0) AL_HIP_CHECK(hipStreamWaitEvent(s, c->ev_fj[1], 0));
		// fragments k_ext_prep left to the monolithic kernel are known now: a handful of them, milliseconds each on one 16-lane group --
		// they go to the side stream at once and run beside the DP jobs
		uint32_t *const n_early_d = (uint32_t *)(A->hist.p + 20); uint32_t n_early = 0;
		if (A->early_list.ensure(nf + 1)) return -1;
		hipLaunchKernelGGL(k_collect_slow, dim3((nf + 255) / 256), dim3(256), 0, s, (const uint32_t *)A->frag_slow.p, nf, A->early_list.p, n_early_d);
		AL_HIP_CHECK(hipEventRecord(c->ev[ST_EXT_PREP + 1], s));
		unsigned long long hist[16] = {0};
		for (int i = 0; i < 10; ++i) c->stat_dp_jobs[i] = c->stat_dp_tbases[i] = 0;
		auto early_mono = [&]() -> int {                                         // (after a synchronisation point of the main stream: n_early is on the host)
			if (n_early == 0) return 0;
			int nbs = ((int)n_early + AL_GPB - 1) / AL_GPB; if (nbs > 1024) nbs = 1024;
			if (A->gws2.ensure((size_t)nbs * AL_GPB * stride + 64)) return -1;
			AL_HIP_CHECK(hipEventRecord(c->ev_fj[0], s)); AL_HIP_CHECK(hipStreamWaitEvent(c->side, c->ev_fj[0], 0));
			if (launch_mono(A->early_list.p, (int)n_early, c->side, A->gws2.p, nbs)) return -1;
			AL_HIP_CHECK(hipEventRecord(c->ev_fj[1], c->side));
			return 0;
		};
		AL_HIP_CHECK(hipMemcpyAsync(&n_early, n_early_d, 4, hipMemcpyDeviceToHost, s));
		if (nj == 0) { AL_HIP_CHECK(hipStreamSynchronize(s)); if (early_mono()) return -1; }
		if (nj > 0) {
			hipLaunchKernelGGL(k_iota, dim3((nj + 255) / 256), dim3(256), 0, s, A->job_idx.p, nj);
			size_t bytes = 0;
			AL_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, A->job_key.p, A->job_key2.p, A->job_idx.p, A->job_idx2.p, (int)nj, 0, 24, s));
			if (A->sort_tmp.ensure(bytes + 16)) return -1;
			AL_HIP_CHECK(rocprim::radix_sort_pairs(A->sort_tmp.p, bytes, A->job_key.p, A->job_key2.p, A->job_idx.p, A->job_idx2.p, (int)nj, 0, 24, s));
			AL_HIP_CHECK(hipEventRecord(c->ev[ST_EXT_SORT + 1], s));
			unsigned long long sub7[2] = {0, 0};
			AL_HIP_CHECK(hipMemcpyAsync(hist, A->hist.p, (AL_NCLS + 1) * 8, hipMemcpyDeviceToHost, s));
			AL_HIP_CHECK(hipMemcpyAsync(sub7, A->hist.p + 36, 16, hipMemcpyDeviceToHost, s));
			AL_HIP_CHECK(hipMemcpyAsync(c->stat_dp_tbases, A->hist.p + 24, AL_NCLS * 8, hipMemcpyDeviceToHost, s));
			AL_HIP_CHECK(hipStreamSynchronize(s));
			for (int i = 0; i < AL_NCLS; ++i) c->stat_dp_jobs[i] = hist[i];
			if (early_mono()) return -1;
			if (getenv("AL_TRACE")) fprintf(stderr, "[airlift] trace: prep + job sort done\n");
			if ((c->P.dbg >> 30) & 1) { fprintf(stderr, "[airlift] DP jobs per class (lane16 lane32 lane64 g1 g2 g4 g8 g22 g32 lds | empty):"); for (int i = 0; i <= AL_NCLS; ++i) fprintf(stderr, " %llu", hist[i]); fprintf(stderr, "\n"); }
			// one launch per job class over its slice of the sorted job list
			static const int NBs[6] = {1, 2, 4, 8, 22, 32};
			uint32_t first = 0;
			// (round 5) A small batch's classes do not fill the chip (C2: 32 000 jobs of 9 ... 22 blocks over three kernels, 8 blocks per CU) and every class ends in the
			// tail of its longest job (the 22-block kernel: 0.85 ms whatever the batch): below AL_DP_CONC jobs (default 700 000; 0: never) the class kernels run side by
			// side on four streams, each on a workspace range of its own; the per-class intervals of the stage table then hold launch order only.
			static const long long conc_thr = getenv("AL_DP_CONC") ? atoll(getenv("AL_DP_CONC")) : 700000;
			long long n_real = 0; for (int cls = 0; cls < AL_NCLS; ++cls) n_real += (long long)hist[cls];                   // (nj counts two slots per hit, most of them empty)
			const bool dp_conc = n_real < conc_thr;
			size_t ws_off[AL_NCLS + 1]; ws_off[0] = 0;
			{   // one workspace range for every class of this batch: sized for the largest now, not grown class by class (a regrow frees -- and waits for -- what the running class uses)
				size_t need = 0;
				for (int cls = 0; cls < AL_NCLS; ++cls) {
					const uint32_t cnt = (uint32_t)hist[cls];
					size_t b = 0;
					if (cnt == 0) b = 0;
					else if (cls < 3) { const int TC = 16 << cls; int nw = (int)((cnt + 63) / 64); if (nw > 1024) nw = 1024; b = (size_t)nw * ((size_t)(AL_LANE_QC + TC) * (size_t)(TC + 16) * 64); }
					else if (cls < 9) {
						const int NB = NBs[cls - 3];
						const size_t pb = (((size_t)(Lmax + 16 * NB) * (size_t)(NB + 1) * 16) + 63) / 64 * 64, cw = ((size_t)(Lmax + 16 * NB) + 31) / 16 * 16, st2 = pb + cw * 8;
						const int cap = NB <= 4 ? (getenv("AL_CAP4") ? atoi(getenv("AL_CAP4")) : 4096) : NB <= 8 ? (getenv("AL_CAP8") ? atoi(getenv("AL_CAP8")) : 4096) : NB <= 22 ? (getenv("AL_CAP22") ? atoi(getenv("AL_CAP22")) : 3072) : 2048;
						const int cap_eff = dp_conc ? (NB <= 4 ? cap / 2 : NB <= 8 ? cap * 5 / 8 : NB <= 22 ? cap * 5 / 6 : cap) : cap;   // (side by side the ranges add up: slightly fewer blocks each keep the sum near what the largest range was)
						int nbj = (int)((cnt + 3) / 4); if (nbj > cap_eff) nbj = cap_eff;
						b = (size_t)nbj * 4 * st2;
						if (cls == 7) { const unsigned long long c12 = std::min<unsigned long long>(sub7[0], cnt), c16 = std::min<unsigned long long>(sub7[1], cnt - c12), c22 = cnt - c12 - c16;
						                if (c22 > 0 && c22 <= 8192) b += (size_t)std::min<unsigned long long>((c22 + 3) / 4, (unsigned long long)nbj) * 4 * st2; }   // (a thin 22-block kernel beside the other two of its class: its own range behind theirs)
					} else { int nbj = (int)cnt; if (nbj > 2048) nbj = 2048; b = (size_t)nbj * stride; }
					b = (b + 255) / 256 * 256;
					ws_off[cls + 1] = dp_conc ? ws_off[cls] + b : 0;
					if (dp_conc) need += b; else if (b > need) need = b;
				}
				if (need && A->gws.ensure(need + 64)) return -1;
			}
			hipStream_t dps[4] = {s, c->aux[0], c->aux[1], c->aux[2]}; int dpk = 0;
			if (dp_conc) { AL_HIP_CHECK(hipEventRecord(c->ev_fj[0], s)); for (int i = 0; i < 3; ++i) AL_HIP_CHECK(hipStreamWaitEvent(c->aux[i], c->ev_fj[0], 0)); }
			hipStream_t const s_main = s;
			bool g12_done = false, thin22_join = false; hipStream_t thin22_stream = nullptr;
			// two cells per lane (al_dev_ksw2.h) where its arithmetic holds: the permute's constant 0xff is the score of an N, scores within +-16
			// (int16 H of the 352 x 512 tile); AL_DP_PK=0: the one-cell form everywhere (tests, A/B)
			static const int pk_env = getenv("AL_DP_PK") ? atoi(getenv("AL_DP_PK")) : 1;
			const int sc_amb_ = c->P.sc_ambi > 0 ? -c->P.sc_ambi : c->P.sc_ambi, sc_N_ = sc_amb_ == 0 ? -std::min(c->P.e2, c->P.e) : sc_amb_;
			const bool dp_pk = pk_env != 0 && sc_N_ == -1 && c->P.a > 0 && c->P.a <= 16 && c->P.b >= 0 && c->P.b <= 16 && c->P.q + c->P.e <= 64 && c->P.q2 + c->P.e2 <= 64 && c->P.q >= 0 && c->P.e >= 0 && c->P.q2 >= 0 && c->P.e2 >= 0;
			for (int cls = 0; cls < AL_NCLS; ++cls) {
				const uint32_t cnt = (uint32_t)hist[cls];
				if (cls == 3) AL_HIP_CHECK(hipEventRecord(c->ev[ST_EXT_DP_LANE + 1], s));
				if (cls == 6) AL_HIP_CHECK(hipEventRecord(c->ev[ST_EXT_DP_G4 + 1], s));
				if (cls == 7) AL_HIP_CHECK(hipEventRecord(c->ev[ST_EXT_DP_G8 + 1], s));
				if (cls == 8 && !g12_done) { AL_HIP_CHECK(hipEventRecord(c->ev[ST_EXT_DP_G12 + 1], s)); AL_HIP_CHECK(hipEventRecord(c->ev[ST_EXT_DP_G16 + 1], s)); g12_done = true; }
				if (cnt == 0) continue;
				unsigned char *const gbase = A->gws.p + ws_off[cls];
				hipStream_t s = dp_conc ? dps[dpk++ & 3] : s_main;                 // (shadows the stage's stream inside the class)
				if (cls < 3) {                                                    // lane-per-job
					const int TC = 16 << cls;
					const size_t tbs = (size_t)(AL_LANE_QC + TC) * (size_t)(TC + 16) * 64;
					int nw = (int)((cnt + 63) / 64); if (nw > 1024) nw = 1024;
					if (cls == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ext_dp_lane<16, AL_LANE_QC>), dim3(nw), dim3(64), 0, s, c->rd_seq.p, c->rd_off.p, c->rd_len.p, G, E, A->job_idx2.p, first, cnt, gbase, tbs, c->P);
					else if (cls == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ext_dp_lane<32, AL_LANE_QC>), dim3(nw), dim3(64), 0, s, c->rd_seq.p, c->rd_off.p, c->rd_len.p, G, E, A->job_idx2.p, first, cnt, gbase, tbs, c->P);
					else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ext_dp_lane<64, AL_LANE_QC>), dim3(nw), dim3(64), 0, s, c->rd_seq.p, c->rd_off.p, c->rd_len.p, G, E, A->job_idx2.p, first, cnt, gbase, tbs, c->P);
				} else if (cls < 9) {
					const int NB = NBs[cls - 3];
					const size_t pb = (((size_t)(Lmax + 16 * NB) * (size_t)(NB + 1) * 16) + 63) / 64 * 64, cw = ((size_t)(Lmax + 16 * NB) + 31) / 16 * 16, st2 = pb + cw * 8;
					int nbj = (int)((cnt + 3) / 4); { static const int caps[3] = { getenv("AL_CAP4") ? atoi(getenv("AL_CAP4")) : 4096, getenv("AL_CAP8") ? atoi(getenv("AL_CAP8")) : 4096, getenv("AL_CAP22") ? atoi(getenv("AL_CAP22")) : 3072 };
					  const int cap = NB <= 4 ? caps[0] : NB <= 8 ? caps[1] : NB <= 22 ? caps[2] : 2048; const int cap_eff = dp_conc ? (NB <= 4 ? cap / 2 : NB <= 8 ? cap * 5 / 8 : NB <= 22 ? cap * 5 / 6 : cap) : cap; if (nbj > cap_eff) nbj = cap_eff; }
#define LAUNCH_DP(NBV) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ext_dp<NBV, 512, NBV * 16>), dim3(nbj), dim3(64), 0, s, c->rd_seq.p, c->rd_off.p, c->rd_len.p, G, E, A->job_idx2.p, first, cnt, gbase, st2, pb, cw, c->P)
// (queries of up to 256 bases -- every short-read set -- get the instance with the smaller query arrays: 12 instead of 14 KB of LDS per block at 16 blocks, a third wavefront per SIMD)
#define LAUNCH_DPK(NBV) do { if (Lmax <= 256) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ext_dp<NBV, 256, NBV * 16, true>), dim3(nbj), dim3(64), 0, s, c->rd_seq.p, c->rd_off.p, c->rd_len.p, G, E, A->job_idx2.p, first, cnt, gbase, st2, pb, cw, c->P); \
	                         else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ext_dp<NBV, 512, NBV * 16, true>), dim3(nbj), dim3(64), 0, s, c->rd_seq.p, c->rd_off.p, c->rd_len.p, G, E, A->job_idx2.p, first, cnt, gbase, st2, pb, cw, c->P); } while (0)
					if (NB == 1) LAUNCH_DP(1); else if (NB == 2) LAUNCH_DP(2); else if (NB == 4) LAUNCH_DP(4); else if (NB == 8) { if (dp_pk) LAUNCH_DPK(8); else LAUNCH_DP(8); } else if (NB == 32) { static const bool pk32 = !(getenv("AL_DP_PK32") && atoi(getenv("AL_DP_PK32")) == 0); if (dp_pk && pk32) LAUNCH_DPK(32); else LAUNCH_DP(32); }
					else {   // 9 ... 22 blocks: the sorted slice holds the jobs of <= 12 blocks first, then 13 ... 16, then the rest
						static const bool split = !getenv("AL_DP_NO_SPLIT");
						const uint32_t c12 = split ? (uint32_t)std::min<unsigned long long>(sub7[0], cnt) : 0u, c16 = split ? (uint32_t)std::min<unsigned long long>(sub7[1], cnt - c12) : 0u, c22 = cnt - c12 - c16;
						const uint32_t first0 = first, cnt0 = cnt;
#define LAUNCH_DPS(NBV, F, N) do { if ((N) > 0) { int nb2 = (int)(((N) + 3) / 4); if (nb2 > nbj) nb2 = nbj; \
							if (dp_pk && Lmax <= 256) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ext_dp<NBV, 256, NBV * 16, true>), dim3(nb2), dim3(64), 0, s, c->rd_seq.p, c->rd_off.p, c->rd_len.p, G, E, A->job_idx2.p, (F), (N), gbase + (size_t)gw_used * 4 * st2, st2, pb, cw, c->P); \
							else if (dp_pk) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ext_dp<NBV, 512, NBV * 16, true>), dim3(nb2), dim3(64), 0, s, c->rd_seq.p, c->rd_off.p, c->rd_len.p, G, E, A->job_idx2.p, (F), (N), gbase + (size_t)gw_used * 4 * st2, st2, pb, cw, c->P); \
							else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ext_dp<NBV, 512, NBV * 16>), dim3(nb2), dim3(64), 0, s, c->rd_seq.p, c->rd_off.p, c->rd_len.p, G, E, A->job_idx2.p, (F), (N), gbase + (size_t)gw_used * 4 * st2, st2, pb, cw, c->P); gw_used += nb2; } } while (0)
						// (the three kernels run one after the other on this stream: they may share the workspace range)
						int gw_used = 0;
						{
							// few 22-block jobs (C4: a few hundred): the kernel is the tail of its longest job, 0.85 ms alone at the end of the stage -- it starts first, on a side stream, beside the other two.
							// (Side by side with the other classes the three kernels of this one still run one after the other and share a range: a range each cost 6.8 GB per context.)
							hipStream_t const s_cls = s;
							const bool thin22 = c22 > 0 && c22 <= 8192u && split;
							if (thin22) { hipStream_t const s22 = s_cls == c->aux[1] ? c->aux[2] : c->aux[1]; AL_HIP_CHECK(hipEventRecord(c->ev_fj[0], s_cls)); AL_HIP_CHECK(hipStreamWaitEvent(s22, c->ev_fj[0], 0)); s = s22; gw_used = nbj; LAUNCH_DPS(22, first0 + c12 + c16, c22); s = s_cls; gw_used = 0; thin22_join = true; thin22_stream = s22; }
							LAUNCH_DPS(12, first0, c12); if (!dp_conc) AL_HIP_CHECK(hipEventRecord(c->ev[ST_EXT_DP_G12 + 1], s_main)); gw_used = 0; LAUNCH_DPS(16, first0 + c12, c16); if (!dp_conc) AL_HIP_CHECK(hipEventRecord(c->ev[ST_EXT_DP_G16 + 1], s_main)); gw_used = 0;
							if (!thin22) LAUNCH_DPS(22, first0 + c12 + c16, c22);
							if (dp_conc) { AL_HIP_CHECK(hipEventRecord(c->ev[ST_EXT_DP_G12 + 1], s_main)); AL_HIP_CHECK(hipEventRecord(c->ev[ST_EXT_DP_G16 + 1], s_main)); }
						}
						g12_done = true;
						(void)cnt0;
#undef LAUNCH_DPS
					}
#undef LAUNCH_DP
#undef LAUNCH_DPK
				} else {
					int nbj = (int)cnt; if (nbj > 2048) nbj = 2048;
					if (tmax <= 512) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ext_dp_lds<512, 256>), dim3(nbj), dim3(GW), 0, s, c->rd_seq.p, c->rd_off.p, c->rd_len.p, G, E, A->job_idx2.p, first, cnt, gbase, stride, p_bytes, cig_words, c->P);
					else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ext_dp_lds<1024, 512>), dim3(nbj), dim3(GW), 0, s, c->rd_seq.p, c->rd_off.p, c->rd_len.p, G, E, A->job_idx2.p, first, cnt, gbase, stride, p_bytes, cig_words, c->P);
				}
				if (getenv("AL_TRACE")) { const hipError_t e = hipStreamSynchronize(s); fprintf(stderr, "[airlift] trace: DP class %d (%u jobs) -> %s\n", cls, cnt, hipGetErrorName(e)); }
				first += cnt;
			}
			if (dp_conc) for (int i = 0; i < 3; ++i) { AL_HIP_CHECK(hipEventRecord(c->ev_aux[i], c->aux[i])); AL_HIP_CHECK(hipStreamWaitEvent(s, c->ev_aux[i], 0)); }
			if (thin22_join && !dp_conc) { AL_HIP_CHECK(hipEventRecord(c->ev_aux[1], thin22_stream)); AL_HIP_CHECK(hipStreamWaitEvent(s, c->ev_aux[1], 0)); }   // (side by side: the join above covers every aux stream)
		}
		else { AL_HIP_CHECK(hipEventRecord(c->ev[ST_EXT_SORT + 1], s)); for (int i = ST_EXT_DP_LANE; i < ST_EXT_DP_G22; ++i) AL_HIP_CHECK(hipEventRecord(c->ev[i + 1], s)); }   // (no jobs: empty intervals)
		AL_HIP_CHECK(hipEventRecord(c->ev[ST_EXT_DP_G22 + 1], s));
		uint32_t *n_slow_d = (uint32_t *)(c->counters.p + 14);
		{   // Fragments with a dozen hits or more: a wavefront each for the per-hit part (its own stream), the rest a lane each -- in SMALL batches only.
			// Such a fragment's bookkeeping between hits (order, parents, pairing of ~20 x 20 hits on global memory) is a millisecond of one lane
			// whichever form runs it; sixty-four of them to a wavefront is how a large batch gets through its tens of thousands (1 M pairs: 5.1 ms
			// against 6.5 ... 8.4 with the wavefront form), while a 262 144-pair batch has few enough for the per-hit part to matter (3.6 -> 2.2 ms).
			static const int fin_env = getenv("AL_FIN_HEAVY") ? atoi(getenv("AL_FIN_HEAVY")) : -1;                    // (0: every fragment on a lane)
			const int fin_heavy = !frag_ord ? 0 : fin_env >= 0 ? fin_env : nf <= 400000 ? AL_FIN_HEAVY : 0;
			if (fin_heavy > 0) {
				AL_HIP_CHECK(hipEventRecord(c->ev_fj[0], s)); AL_HIP_CHECK(hipStreamWaitEvent(c->aux[0], c->ev_fj[0], 0));   // (not the side stream: the monolithic kernel may still be running there)
				hipLaunchKernelGGL(k_ext_finish_wave, dim3(std::min(nf, 16384)), dim3(64), 0, c->aux[0], c->rd_seq.p, c->rd_off.p, c->rd_len.p, c->frag_first.p, c->frag_rep.p, W, G, E, lt, A->sc_ws.p, A->sc_off.p, nf, c->P, A->slow_list.p, n_slow_d, 1, frag_ord, fin_heavy);
				AL_HIP_CHECK(hipEventRecord(c->ev_aux[0], c->aux[0]));
			}
			hipLaunchKernelGGL(k_ext_finish, dim3((nf + 255) / 256), dim3(256), 0, s, c->rd_seq.p, c->rd_off.p, c->rd_len.p, c->frag_first.p, c->frag_rep.p, W, G, E, lt, A->sc_ws.p, A->sc_off.p, nf, c->P, A->slow_list.p, n_slow_d, 1, frag_ord, fin_heavy);
			if (fin_heavy > 0) AL_HIP_CHECK(hipStreamWaitEvent(s, c->ev_aux[0], 0));
		}
		uint32_t n_slow = 0;
		AL_HIP_CHECK(hipMemcpyAsync(&n_slow, n_slow_d, 4, hipMemcpyDeviceToHost, s));
		AL_HIP_CHECK(hipStreamSynchronize(s));
		if (n_slow > 0) {
			int nbs = ((int)n_slow + AL_GPB - 1) / AL_GPB; if (nbs > 1024) nbs = 1024;
			if (A->gws.ensure((size_t)nbs * AL_GPB * stride + 64)) return -1;
			if (launch_mono(A->slow_list.p, (int)n_slow, s, A->gws.p, nbs)) return -1;
		}
		if (n_early > 0) AL_HIP_CHECK(hipStreamWaitEvent(s, c->ev_fj[1], 0));
		c->stat_n_slow = n_slow + n_early;
		if (getenv("AL_TRACE")) fprintf(stderr, "[airlift] trace: ext: of %d fragments the monolithic kernel takes %u known after prep (oversize, z-drop in a closed-form flank; side stream) and %u after the DP (CIGAR above the fast path's buffer)\n", nf, n_early, n_slow);
		AL_HIP_CHECK(hipEventRecord(c->ev[ST_EXT_FINISH + 1], s));
	}
	AL_HIP_CHECK(hipGetLastError());
	if ((c->P.dbg >> 21) & 1) {
		unsigned long long h[8]; AL_HIP_CHECK(hipMemcpyAsync(h, A->dbgbuf.p + 600, 64, hipMemcpyDeviceToHost, s)); AL_HIP_CHECK(hipStreamSynchronize(s));
		fprintf(stderr, "[airlift] K5 profile (group-cycles): dp_init=%llu dp_rows=%llu backtrack=%llu align1_total=%llu post=%llu group_total=%llu n_dp=%llu n_rows=%llu\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
	}
	if ((c->P.dbg >> 20) & 1) {
		std::vector<unsigned long long> h(1 + 32 * 16);
		AL_HIP_CHECK(hipMemcpyAsync(h.data(), A->dbgbuf.p, h.size() * 8, hipMemcpyDeviceToHost, s)); AL_HIP_CHECK(hipStreamSynchronize(s));
		fprintf(stderr, "[airlift] DP differential check: %llu mismatching calls\n", h[0]);
		for (unsigned long long k = 0; k < h[0] && k < 32; ++k) { const unsigned long long *o = h.data() + 1 + k * 16; fprintf(stderr, "  qlen=%llu tlen=%llu flag=%llu max %d/%d max_t %d/%d max_q %d/%d mqe %d/%d mqe_t %d/%d score %d/%d zd %llu\n", o[0], o[1], o[2], (int)o[3], (int)o[4], (int)o[5], (int)o[6], (int)o[7], (int)o[8], (int)o[9], (int)o[10], (int)o[11], (int)o[12], (int)o[13], (int)o[14], o[15]); }
	}
	// dense output
	AL_HIP_CHECK(hipMemsetAsync(A->reg_cnt.p + nr, 0, 4, s));
	if (scan32(c, A->reg_cnt.p, A->out_off.p, nr)) return -1;
	uint64_t out_total = 0;
	AL_HIP_CHECK(hipMemcpyAsync(&out_total, A->out_off.p + nr, 8, hipMemcpyDeviceToHost, s));
	AL_HIP_CHECK(hipStreamSynchronize(s));
	if (A->out.ensure(out_total + 1)) return -1;
	A->out_total = out_total;
	hipLaunchKernelGGL(k_compact, dim3((nf + 255) / 256), dim3(256), 0, s, c->frag_first.p, W, A->out_off.p, A->out.p, nf);
	AL_HIP_CHECK(hipGetLastError());
	return 0;
}

// the CIGAR arena of the last al_run_align_stage() was too small (counters[9] != 0): twice the size for the re-run
void al_align_grow_arena(al_ctx_t *c) { get_state(c)->arena_scale *= 2; }

int al_fetch_raw(al_ctx_t *c, AlRawResult &R)
{   // device -> host copies of the last al_batch_run: per-read record offsets, records, CIGAR arena, repeat lengths
	AlignState *A = get_state(c);
	const int nf = c->n_frag, nr = c->n_reads;
	AL_HIP_CHECK(hipSetDevice(c->device));
	if (c->dev_batch) { fprintf(stderr, "[airlift] al_fetch_raw: the batch was built on the device (stream driver); its records are read through al_stream_sam\n"); return -1; }
	unsigned long long h[16]; AL_HIP_CHECK(hipMemcpy(h, c->counters.p, sizeof(h), hipMemcpyDeviceToHost));
	if (h[7] || h[8] || h[9]) { fprintf(stderr, "[airlift] device pipeline error: limit=0x%llx (byte k = site k: 0 prep qlen, 1 prep window, 2 dp window, 3 qlen, 4 split capacity, 5 pair scores, 6 lane cigar, 7 finish pair scores) logf_miss=%llu cigar_arena_overflow=%llu\n", h[7], h[8], h[9]); return -4; }
	if (R.off.resize(nr + 1) || R.out.resize(A->out_total) || R.rep.resize(nf)) return -1;
	R.flip = c->h_flip; R.rd_len.assign(c->h_rd_len.begin(), c->h_rd_len.begin() + nr);
	if (nf == 0) return 0;
	AL_HIP_CHECK(hipMemcpy(R.off.data(), A->out_off.p, (size_t)(nr + 1) * 8, hipMemcpyDeviceToHost));
	if (A->out_total) AL_HIP_CHECK(hipMemcpy(R.out.data(), A->out.p, A->out_total * sizeof(AlReg), hipMemcpyDeviceToHost));
	AL_HIP_CHECK(hipMemcpy(R.rep.data(), c->frag_rep.p, (size_t)nf * 4, hipMemcpyDeviceToHost));
	const uint64_t n_arena = h[11];
	if (R.arena.resize(n_arena)) return -1;
	if (n_arena) AL_HIP_CHECK(hipMemcpy(R.arena.data(), A->arena.p, n_arena * 4, hipMemcpyDeviceToHost));
	return 0;
}

// device pointers of the last al_batch_run's result block (records per read, CIGAR arena) for the device SAM writer
#include "al_stream.h"
int al_align_result(al_ctx_t *c, AlDevResult *r)
{
	AlignState *A = get_state(c);
	if (!c->ran) return -1;
	AL_HIP_CHECK(hipSetDevice(c->device));
	unsigned long long h[16]; AL_HIP_CHECK(hipMemcpyAsync(h, c->counters.p, sizeof(h), hipMemcpyDeviceToHost, c->stream)); AL_HIP_CHECK(hipStreamSynchronize(c->stream));
	if (h[7] || h[8] || h[9]) { fprintf(stderr, "[airlift] device pipeline error: limit=0x%llx logf_miss=%llu cigar_arena_overflow=%llu\n", h[7], h[8], h[9]); return -4; }
	r->out = A->out.p; r->out_off = A->out_off.p; r->arena = A->arena.p; r->out_total = A->out_total;
	return 0;
}

void al_reg_from_raw(const AlRawResult &R, int read, int k, al_reg1_t &q)
{   // AlReg -> al_reg1_t; q.cigar points into R (inline words or arena): callers that hand ownership out must copy it
	const AlReg &r = R.out[R.off[read] + k];
	q.id = r.id; q.cnt = r.cnt; q.rid = r.rid; q.score = r.score; q.qs = r.qs; q.qe = r.qe; q.rs = r.rs; q.re = r.re;
	q.parent = r.parent; q.subsc = r.subsc; q.mlen = r.mlen; q.blen = r.blen; q.n_sub = r.n_sub; q.score0 = r.score0;
	q.mapq = r.mapq & 0xff; q.split = r.flags & 3; q.rev = (r.flags & ALR_REV) ? 1 : 0; q.inv = 0; q.sam_pri = (r.flags & ALR_SAM_PRI) ? 1 : 0;
	q.proper_frag = (r.flags & ALR_PROPER) ? 1 : 0; q.pe_thru = (r.flags & ALR_PE_THRU) ? 1 : 0; q.seg_split = (r.flags & ALR_SEG_SPLIT) ? 1 : 0;
	q.seg_id = (r.flags >> 8) & 0xff; q.split_inv = 0; q.dummy = 0; q.hash = r.hash; q.dp_score = r.dp_score; q.dp_max = r.dp_max; q.dp_max2 = r.dp_max2; q.n_ambi = r.n_ambi;
	q.n_cigar = (r.flags & ALR_HAS_P) ? r.n_cigar : 0;
	q.cigar = q.n_cigar ? const_cast<uint32_t *>(r.cigar_off == AL_CIG_INLINE ? r.cig_inl : R.arena.data() + r.cigar_off) : nullptr;
	if (R.flip[read]) { const int qlen = (int)R.rd_len[read], t = q.qs; q.qs = qlen - q.qe; q.qe = qlen - t; q.rev = !q.rev; }   // map.c:486-497
}

int al_fetch_align(al_ctx_t *c, int *n_regs, al_reg1_t **regs, int *rep_len)
{
	AlRawResult R;
	const int rc = al_fetch_raw(c, R);
	if (rc) return rc;
	const int nf = c->n_frag, nr = c->n_reads;
	for (int f = 0; f < nf; ++f) if (rep_len) rep_len[f] = R.rep[f];
	for (int i = 0; i < nr; ++i) {
		const int n = (int)(R.off[i + 1] - R.off[i]);
		n_regs[i] = n; regs[i] = nullptr;
		if (n == 0) continue;
		al_reg1_t *o = (al_reg1_t *)calloc(n, sizeof(al_reg1_t));
		for (int k = 0; k < n; ++k) {
			al_reg_from_raw(R, i, k, o[k]);
			if (o[k].n_cigar) { uint32_t *cg = (uint32_t *)malloc((size_t)o[k].n_cigar * 4); memcpy(cg, o[k].cigar, (size_t)o[k].n_cigar * 4); o[k].cigar = cg; }
		}
		regs[i] = o;
	}
	return 0;
}
