// al_dev_ksw.h -- register-resident ("systolic") form of the extension DP for one 16-lane group.
//
// Same arithmetic as d_ksw_lds() in al_kernels_align.hip (= ksw_extd2_sse, ksw2_extd2_sse.c:26-393) but the seven
// int8 state rows and the int32 H row live in VGPRs: lane l of the group owns cells t = 16*b + l for b < NB, packed as
//   A[b] = x | v<<8 | x2<<16 | u<<24        B[b] = y | y2<<8 | s<<16 | sf<<24        H[b]
// The left-neighbour values (x,v,x2)[t-1] arrive with one DPP row_shr:1 on A (lane 0 takes the carry of the previous
// 16-cell block, exactly the x1_/v1_/x21_ registers of the SSE code); the row maximum is a 4-step DPP butterfly on a
// 64-bit key that encodes the reference's evaluation order.  No LDS traffic and no barrier inside a row except the
// query byte (one ds_read_u8 per active block) and the traceback byte store.
// Instruction diet of the block body (it runs at the VALU issue rate, so every instruction is time):
//   * the substitution score is one v_perm_b32: W[b] holds the lane's target base scored against query bases 0..3, one per byte
//     (all sc_N for a target N), the query byte is the selector (4 = N picks sc_N from the other source);
//   * the query row is padded by 16*NB bytes on both sides, so the query byte of block b is a constant 16*b from the lane's
//     row pointer -- no index clamps;
//   * inside the row a lane only tracks its best cell as H*64 | end-cell flag | (31 - block): the position key of the
//     reference's evaluation order (:307-349) is rebuilt once per row from the winning block, not once per block.
#pragma once

#define DPP_ROW_SHR1    0x111
#define DPP_QUAD_XOR1   0xB1
#define DPP_QUAD_XOR2   0x4E
#define DPP_HALF_MIRROR 0x141
#define DPP_ROW_MIRROR  0x140

#define DPP_ROW_BCAST15 0x15F    // row_newbcast:15 -- lane 15 of every 16-lane row to all lanes of the row

__device__ __forceinline__ int d_dpp_shr1(int carry, int v) { return __builtin_amdgcn_update_dpp(carry, v, DPP_ROW_SHR1, 0xf, 0xf, false); }
__device__ __forceinline__ int d_dpp_last(int v) { return __builtin_amdgcn_update_dpp(0, v, DPP_ROW_BCAST15, 0xf, 0xf, false); }   // (a VALU move: no LDS round trip as with ds_bpermute)

template <int NB, class LT>
__device__ __forceinline__ void d_ksw_reg(LT &L, const int gl, GroupWs &ws, int qlen, int tlen, const AlParams &P,
                          int w, int zdrop, int end_bonus, int flag, EzD &ez, bool do_bt = true)
{
	int q = P.q, e = P.e, q2 = P.q2, e2 = P.e2;
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
	int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
	const bool right = (flag & EZ_RIGHT) != 0;
	const long long tP0 = PROF_ON(P) ? clock64() : 0;
	// reversed query, zero padded (ksw2_extd2_sse.c:118), with 16*NB bytes in front (never used by a cell that takes its score) and
	// zeros up to qlen + 16*NB + 16 behind: every block of every row reads in bounds without a clamp
	constexpr int QPAD = 16 * NB;
	uint8_t *qr = L.sq + QPAD;
	if constexpr (!LT::kQrReady) for (int t = gl; t < qlen + QPAD + 16; t += GW) qr[t] = t < qlen ? L.qbuf[qlen - 1 - t] : 0;
	(void)qlen_;
	uint32_t A[NB], B[NB], W[NB]; int32_t H[NB];
	uint32_t nreg = (uint32_t)(uint8_t)sc_N;                  // source 0 of the score permute: byte 0 = score against a query N
	{
		const uint32_t m1 = (uint8_t)(int8_t)(-q - e), m2 = (uint8_t)(int8_t)(-q2 - e2);
		const uint32_t misrep = 0x01010101u * (uint32_t)(uint8_t)sc_mis, nrep = 0x01010101u * (uint32_t)(uint8_t)sc_N;
#pragma unroll
		for (int b = 0; b < NB; ++b) {
			const int t = 16 * b + gl;
			const uint32_t tb = t < tlen ? L.tbuf[t] : 0;
			A[b] = m1 | m1 << 8 | m2 << 16 | m1 << 24;
			B[b] = m1 | m2 << 8;
			W[b] = tb >= 4 ? nrep : ((misrep & ~(0xffu << (8 * tb))) | (uint32_t)(uint8_t)sc_mch << (8 * tb));
			H[b] = KSW_NEG_INF;
		}
	}
	asm volatile("" : "+v"(nreg));
	GSYNC();
	const size_t prow = (size_t)n_col_ * 16;
	uint8_t *ptb = ws.p;                                  // traceback bytes: the LDS tile when the job's rows fit (a structure without a tile: always the HBM scratch)
	if constexpr (LT::kPtb > 0) { if ((size_t)(qlen + tlen - 1) * prow <= (size_t)LT::kPtb) ptb = L.ptb; }
	int last_st = -1, last_en = -1, r;
	const long long tP1 = PROF_ON(P) ? clock64() : 0;
	const int n_rows_dbg = ((P.dbg >> 22) & 1) ? 0 : qlen + tlen - 1;
	for (r = 0; r < n_rows_dbg; ++r) {
		int st, en;
		d_row_bounds(r, qlen, tlen, w, st, en);
		if (st > en) { ez.zdropped = 1; break; }
		const int st0 = st, en0 = en;
		st = st / 16 * 16; en = (en + 16) / 16 * 16 - 1;
		const int st_ = st >> 4, en_ = en >> 4;
		const int cover_end = st0 + ((en0 - st0) >> 4) * 16 + 15;       // last byte written by the 16-byte score stores (:158-176)
		// ---- boundary conditions (:141-157)
		uint32_t carry;
		{
			int8_t x1 = (int8_t)(-q - e), x21 = (int8_t)(-q2 - e2), v1 = (int8_t)(-q - e);
			if (st > 0) {
				if (st - 1 >= last_st && st - 1 <= last_en) {
					uint32_t av = 0;
#pragma unroll
					for (int b = 0; b < NB; ++b) if (b == st_ - 1) av = (uint32_t)d_dpp_last((int)A[b]);
					x1 = (int8_t)av; v1 = (int8_t)(av >> 8); x21 = (int8_t)(av >> 16);
				}
			} else v1 = r == 0 ? (int8_t)(-q - e) : r < long_thres ? (int8_t)(-e) : r == long_thres ? (int8_t)long_diff : (int8_t)(-e2);
			carry = (uint32_t)(uint8_t)x1 | (uint32_t)(uint8_t)v1 << 8 | (uint32_t)(uint8_t)x21 << 16;
		}
		const int8_t ubound = r == 0 ? (int8_t)(-q - e) : r < long_thres ? (int8_t)(-e) : r == long_thres ? (int8_t)long_diff : (int8_t)(-e2);
		uint8_t *const prl = ptb + (size_t)r * prow - st + gl;              // this lane's byte of block 0; block b is a constant 16*b further
		const uint8_t *const qrow = qr + (qlen - 1 - r) + gl;               // this lane's query byte of block 0; block b is a constant 16*b further
		const int be = en0 >> 4;
		int hprev15 = 0;                                                 // H[r-1][en0-1] when en0 is the first lane of its block (that block may be outside [st_,en_])
		if (NB > 1 && r > 0 && (en0 & 15) == 0) {                           // (one row in sixteen per group: skipped when no group of the wavefront needs it)
			int hsel = 0;
#pragma unroll
			for (int b = 0; b < NB; ++b) hsel = (b + 1 == be) ? H[b] : hsel;
			hprev15 = d_dpp_last(hsel);
		}
		const int en1 = st0 + (en0 - st0) / 4 * 4;
		uint32_t ybits = (uint32_t)(uint8_t)(int8_t)(-q - e) | (uint32_t)(uint8_t)(int8_t)(-q2 - e2) << 8, ub0 = (uint32_t)(uint8_t)ubound;
		asm volatile("" : "+v"(ybits), "+v"(ub0));                         // (in VGPRs: one v_perm_b32 each below, no second literal on the constant bus)
		const bool enr = en >= r;
		const int tend = tlen_ * 16;
		const bool store_p = !((P.dbg >> 23) & 1);
		int lkey = (int)0x80000000;                                      // this lane's best cell: H*64 | (end cell or row 0) << 5 | 31 - block
		// The block body is straight-line code: every condition below is uniform inside a 16-lane group but differs between
		// the four groups of a wavefront, so each `if` would cost an exec-mask round trip per block.  Only two branches
		// remain: skipping a block no group-lane needs, and the timing-experiment switch around the traceback store.
#pragma unroll
		for (int b = 0; b < NB; ++b) {
			const bool act = NB == 1 ? true : (b >= st_ && b <= en_);
			if (NB >= 4 && !act) continue;
			const int t = 16 * b + gl;
			uint32_t a_old = A[b], b_old = B[b];
			const bool isr = act && enr && t == r;                           // y[r], y2[r], u[r] (:150-153)
			{
				const uint32_t by = __builtin_amdgcn_perm(b_old, ybits, 0x07060100u);   // bytes 0,1 <- y, y2
				const uint32_t au = __builtin_amdgcn_perm(ub0, a_old, 0x04020100u);     // byte 3 <- u
				b_old = isr ? by : b_old; a_old = isr ? au : a_old;
			}
			{                                                                // score bytes (:158-176): score of (target base of this lane, query byte) by one byte permute
				const bool son = act && t >= st0 && t <= cover_end && t < tend;
				const uint32_t sq2 = qrow[16 * b];
				const uint32_t scw = __builtin_amdgcn_perm(nreg, W[b], sq2);
				const uint32_t bs = __builtin_amdgcn_perm(b_old, scw, 0x07000504u);     // byte 2 <- score
				b_old = son ? bs : b_old;
			}
			const uint32_t left = (uint32_t)d_dpp_shr1((int)carry, (int)a_old);
			if (NB > 1) { const uint32_t cn = (uint32_t)d_dpp_last((int)a_old); carry = act ? cn : carry; }
			const int xt1 = (int8_t)left, vt1 = (int8_t)(left >> 8), x2t1 = (int8_t)(left >> 16);
			const int ut = (int8_t)(a_old >> 24), yo = (int8_t)b_old, y2o = (int8_t)(b_old >> 8);
			int z = (int8_t)(b_old >> 16);
			// int8 lanes of the reference, held sign-extended in 32-bit registers: every sum below is re-wrapped to int8
			int a = (int8_t)(xt1 + vt1), bb = (int8_t)(yo + ut), a2 = (int8_t)(x2t1 + vt1), b2 = (int8_t)(y2o + ut);
			// left-aligned gaps take a strict '>' (ksw2_extd2_sse.c:206-214), right-aligned '>=' (:252-260)
			// The four gap candidates among themselves first, the substitution score (it comes from the LDS read above) last.  Same result as the
			// reference's chain score -> a -> b -> a2 -> b2: with '>' d is the FIRST position of the maximum of the five, with '>=' the LAST one, and
			// the score sits at position 0 either way.
			const int ge = right ? 1 : 0;
			int d = 1, zc = a;
			d = (bb + ge > zc) ? 2 : d; zc = zc > bb ? zc : bb;
			d = (a2 + ge > zc) ? 3 : d; zc = zc > a2 ? zc : a2;
			d = (b2 + ge > zc) ? 4 : d; zc = zc > b2 ? zc : b2;
			d = (zc + ge > z) ? d : 0;  z = z > zc ? z : zc;
			z = z < (int)sc_mch ? z : (int)sc_mch;
			const int un = (int8_t)(z - vt1), vn = (int8_t)(z - ut);
			int tmp = (int8_t)(z - q); a = (int8_t)(a - tmp); bb = (int8_t)(bb - tmp);
			tmp = (int8_t)(z - q2); a2 = (int8_t)(a2 - tmp); b2 = (int8_t)(b2 - tmp);
			const bool pa = a + ge > 0, pb = bb + ge > 0, pa2 = a2 + ge > 0, pb2 = b2 + ge > 0;
			const int xn = (int8_t)((pa ? a : 0) - (int)qe_), yn = (int8_t)((pb ? bb : 0) - (int)qe_);
			const int x2n = (int8_t)((pa2 ? a2 : 0) - (int)qe2_), y2n = (int8_t)((pb2 ? b2 : 0) - (int)qe2_);
			d |= (pa ? 0x08 : 0) | (pb ? 0x10 : 0) | (pa2 ? 0x20 : 0) | (pb2 ? 0x40 : 0);
			const uint32_t An = ((uint32_t)xn & 0xffu) | ((uint32_t)vn & 0xffu) << 8 | ((uint32_t)x2n & 0xffu) << 16 | (uint32_t)un << 24;
			const uint32_t Bn = (b_old & 0xffff0000u) | ((uint32_t)yn & 0xffu) | ((uint32_t)y2n & 0xffu) << 8;
			A[b] = act ? An : A[b]; B[b] = act ? Bn : B[b];
			if (store_p) { if (NB >= 4 || act) prl[16 * b] = (uint8_t)d; }
			// ---- exact max (:307-349): H row update and this lane's candidate
			{
				const int hold = H[b];
				const int hl = d_dpp_shr1(hprev15, hold);                       // H[r-1][t-1]; consumed by the lane t == en0 only
				const bool inr = act && r > 0 && t >= st0 && t <= en0;
				const bool r0c = act && r == 0 && t == 0;
				const bool isen = t == en0;
				const int h_v = hold + vn, h_u = hl + un, h_0 = vn - qe;
				int h = (isen && en0 > 0) ? h_u : h_v;
				h = r0c ? h_0 : h;
				const bool upd = inr || r0c;
				H[b] = upd ? h : hold;
				// equal H inside a lane: the end cell (evaluated first by the reference) wins, then the lower block (its cells come earlier in the
				// reference's order: (t - st0) & 3 is the same for all of a lane's cells, the tail cells t >= en1 sit in the last blocks)
				{ const int k2 = h * 64 + ((isen || r0c) ? 63 - b : 31 - b); lkey = (upd && k2 > lkey) ? k2 : lkey; }
			}
		}
		int key32;                                                       // this lane's best as (H<<16 | 0xffff-ord), ord = rank in the reference's evaluation order
		{
			const int cb = lkey & 63, tt = 16 * (31 - (cb & 31)) + gl, dt = tt - st0;
			const int ord_a = 1 + (dt & 3) * 4096 + (dt >> 2), ord_b = 1 + 4 * 4096 + (tt - en1);
			int ord = tt < en1 ? ord_a : ord_b;
			ord = cb >= 32 ? 0 : ord;
			key32 = lkey == (int)0x80000000 ? lkey : (lkey >> 6) * 65536 + (0xffff - ord);
		}
		int max_H, max_t;
		{
			int ord;
			{   // |H| < 2^15 on this path (targets <= 352, queries <= 512 bases): H and the scan position share one 32-bit key
				int k = key32;
				{ const int o = __builtin_amdgcn_update_dpp(k, k, DPP_QUAD_XOR1, 0xf, 0xf, false); k = o > k ? o : k; }
				{ const int o = __builtin_amdgcn_update_dpp(k, k, DPP_QUAD_XOR2, 0xf, 0xf, false); k = o > k ? o : k; }
				{ const int o = __builtin_amdgcn_update_dpp(k, k, DPP_HALF_MIRROR, 0xf, 0xf, false); k = o > k ? o : k; }
				{ const int o = __builtin_amdgcn_update_dpp(k, k, DPP_ROW_MIRROR, 0xf, 0xf, false); k = o > k ? o : k; }
				max_H = k >> 16; ord = 0xffff - (k & 0xffff);
			}
			max_t = r == 0 ? 0 : ord == 0 ? en0 : ord < 1 + 4 * 4096 ? st0 + ((ord - 1) & 4095) * 4 + ((ord - 1) >> 12) : en1 + (ord - 1 - 4 * 4096);
		}
		if (r - st0 == qlen - 1) {                                           // :353-354
			int hs = 0;
#pragma unroll
			for (int b = 0; b < NB; ++b) if (b == (st0 >> 4)) hs = __shfl(H[b], st0 & 15, GW);
			if (hs > ez.mqe) { ez.mqe = hs; ez.mqe_t = st0; }
		}
		bool brk = false;                                                    // ksw_apply_zdrop, ksw2.h:160-176
		if (max_H > ez.max) { ez.max = max_H; ez.max_t = max_t; ez.max_q = r - max_t; }
		else if (max_t >= ez.max_t && r - max_t >= ez.max_q) {
			const int tl = max_t - ez.max_t, ql = (r - max_t) - ez.max_q, l = tl > ql ? tl - ql : ql - tl;
			if (zdrop >= 0 && ez.max - max_H > zdrop + l * e2) { ez.zdropped = 1; brk = true; }
		}
		if (brk) break;
		if (r == qlen + tlen - 2 && en0 == tlen - 1) {
			int hs = 0;
#pragma unroll
			for (int b = 0; b < NB; ++b) if (b == ((tlen - 1) >> 4)) hs = __shfl(H[b], (tlen - 1) & 15, GW);
			ez.score = hs;
		}
		last_st = st; last_en = en;
	}
	GSYNC();
	const long long tP2 = PROF_ON(P) ? clock64() : 0;
	if (do_bt && !((P.dbg >> 25) & 1)) {
		const int rev_cigar = !!(flag & EZ_REV_CIGAR);
		CigW cw{L.ezc, 0, AL_LCIG, ws.ezc, 0xffffffffu};
		if (!ez.zdropped && !(flag & EZ_EXTZ_ONLY)) d_backtrack(ptb, n_col_ * 16, qlen, tlen, w, rev_cigar, tlen - 1, qlen - 1, cw);
		else if (!ez.zdropped && (flag & EZ_EXTZ_ONLY) && ez.mqe + end_bonus > ez.max) { ez.reach_end = 1; d_backtrack(ptb, n_col_ * 16, qlen, tlen, w, rev_cigar, ez.mqe_t, qlen - 1, cw); }
		else if (ez.max_t >= 0 && ez.max_q >= 0) d_backtrack(ptb, n_col_ * 16, qlen, tlen, w, rev_cigar, ez.max_t, ez.max_q, cw);
		ez.n_cigar = cw.n; ws.cur_ezc = cw.c;
	}
	GSYNC();
	if (PROF_ON(P) && gl == 0) { const long long tP3 = clock64(); atomicAdd(&ws.dbg[600], (unsigned long long)(tP1 - tP0)); atomicAdd(&ws.dbg[601], (unsigned long long)(tP2 - tP1)); atomicAdd(&ws.dbg[602], (unsigned long long)(tP3 - tP2)); atomicAdd(&ws.dbg[606], 1ULL); atomicAdd(&ws.dbg[607], (unsigned long long)(qlen + tlen - 1)); }
}
