// al_dev_ksw2.h -- the extension DP of al_dev_ksw.h with TWO cells per lane (round 5).
//
// d_ksw_reg runs at the VALU issue rate with one cell per lane and ~98 instructions per 16-cell block step, most of them unpacking and
// re-wrapping the int8 lanes of ksw_extd2_sse (ksw2_extd2_sse.c:26-393) in 32-bit registers.  Here a lane of the 16-lane group owns the two
// NEIGHBOURING cells t0 = 32c + 2l and t1 = t0 + 1 of every 32-cell superblock c, each state row is one register per superblock with a cell in
// either 16-bit half, and an int8 value v is held as v << 8:
//   * the reference's wrapping int8 adds / subtracts are v_pk_add_u16 / v_pk_sub_u16 (the low byte stays 0, the wrap happens at bit 15 of the
//     half = bit 7 of the value), its signed compares and max / min are v_pk_max_i16 / v_pk_min_i16 on the same halves;
//   * the five-way choice z = max(s, a, b, a2, b2) with the traceback state d (:206-214 left-aligned: first maximum, :252-260 right-aligned: last)
//     is four v_pk_max_i16 on candidates that carry a tag in their free low byte (7 - position or position): the winning tag is d;
//   * "a > 0" / "a >= 0" of the gap-open tests are one v_pk_max_i16 + v_pk_min_u16 on a | ge;
//   * both cells of a lane lie in the same 16-cell block of the reference, so everything that is per block there (the block range [st_, en_],
//     the stale cells outside [st0, en0]) is per lane here, and the left neighbour is the lane's own other half or one DPP row_shr:1;
//   * H is held as int16 (|H| < 2^14 on this path: targets <= 352, queries <= 512, |score step| <= 16) -- two cells per v_pk_add_u16;
//   * the query row is kept as the selector bytes of the score permute (one array per half), the target's score tables wait in LDS as well.
// The traceback bytes go to the same addresses as in d_ksw_reg (two cells = one 16-bit store), so d_backtrack is shared.
// Preconditions (checked by the caller, d_ksw_reg otherwise): sc_N == -1 (the permute's constant 0xff stands for it), |a|, |b| <= 16.
#pragma once

#define DPP_ROW_BCAST7 0x157     // row_newbcast:7

typedef short al_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short al_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ al_s2 pk_s(uint32_t v) { return __builtin_bit_cast(al_s2, v); }
__device__ __forceinline__ uint32_t pk_u(al_s2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return pk_u(pk_s(a) + pk_s(b)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return pk_u(pk_s(a) - pk_s(b)); }
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { return pk_u(__builtin_elementwise_max(pk_s(a), pk_s(b))); }
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) { return pk_u(__builtin_elementwise_min(pk_s(a), pk_s(b))); }
__device__ __forceinline__ uint32_t pk_sar8(uint32_t a) { return pk_u(pk_s(a) >> (al_s2)(8)); }
// (the two below in assembly: the compiler turns min(x, 1) on halves into compares and selects)
__device__ __forceinline__ uint32_t pk_minu1(uint32_t a) { uint32_t d; asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(d) : "v"(a)); return d; }       // per half: a != 0
__device__ __forceinline__ uint32_t pk_subsat(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b)); return d; }   // per half: max(a - b, 0), unsigned
__device__ __forceinline__ uint32_t pk_dec(uint32_t a) { uint32_t d; asm("v_pk_sub_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(d) : "v"(a)); return d; }          // per half: a - 1
__device__ __forceinline__ uint32_t pk_mask_eq(uint32_t t, uint32_t v) { return pk_dec(pk_minu1(t ^ v)); }                                    // per half: 0xffff where t == v
__device__ __forceinline__ uint32_t pk_mask_in(uint32_t t, uint32_t lo, uint32_t width) { return pk_dec(pk_minu1(pk_subsat(pk_sub(t, lo), width))); }   // per half: 0xffff where lo <= t <= lo + width (unsigned)
__device__ __forceinline__ uint32_t d_bfi(uint32_t mask, uint32_t if_set, uint32_t if_clear) { return (if_set & mask) | (if_clear & ~mask); }   // v_bfi_b32
__device__ __forceinline__ uint32_t pk_splat(int v) { return (uint32_t)(v & 0xffff) * 0x00010001u; }
__device__ __forceinline__ uint32_t pk_splat8(int v) { return (uint32_t)((v & 0xff) << 8) * 0x00010001u; }   // int8 value in the high byte of both halves

// NP superblocks of 32 cells; LT as for d_ksw_reg; selE / selO: the reversed query as selector bytes of the score permute for a cell in the low
// half (E(base) = base) and in the high half (O(base) = 4 + base), E(N) = O(N) = 0x0d (the permute's constant 0xff = sc_N); 32 * NP bytes of front
// pad before byte 0 and E(0) / O(0) (base 0, as the reference's zero padding) up to qlen + 32 * NP + 1 behind the query.
// DIR: 0 = left-aligned gaps, 1 = right-aligned (EZ_RIGHT), 2 = by the flag (a wavefront whose groups differ).  The tags of the five-way choice RIDE IN THE STATE: x, y, x2, y2
// carry theirs in the low byte of either half from the moment they are made (the constant subtracted there is q + e less the tag), u and v have zero low bytes,
// so the sums a = x + v ... come out tagged and the four ors per cell pair are gone; with DIR known the gap-stays-open test needs no or either.
template <int NP, class LT, int DIR = 2>
__device__ __forceinline__ void d_ksw_pk(LT &L, const uint8_t *__restrict__ selE, const uint8_t *__restrict__ selO, const int gl, GroupWs &ws, int qlen, int tlen, const AlParams &P,
                                         int w, int zdrop, int end_bonus, int flag, EzD &ez, bool do_bt = true)
{
	int q = P.q, e = P.e, q2 = P.q2, e2 = P.e2;
	if (q2 + e2 < q + e) { int t = q; q = q2; q2 = t; t = e; e = e2; e2 = t; }
	const int qe = q + e;
	const int sc_mch = P.a, sc_mis = -P.b;
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	const int tlen_ = (tlen + 15) / 16;
	int n_col_ = qlen < tlen ? qlen : tlen;
	n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
	int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
	const bool right = DIR == 2 ? (flag & EZ_RIGHT) != 0 : DIR == 1;
	const uint32_t M1 = pk_splat8(-q - e), M2 = pk_splat8(-q2 - e2), QE = pk_splat8(q + e), QE2 = pk_splat8(q2 + e2), QQ = pk_splat8(q), QQ2 = pk_splat8(q2), MCH = pk_splat8(sc_mch);
	// tags of the five candidates in the low byte of a half: first maximum wins (left-aligned) <=> larger tag for the earlier position
	const uint32_t TG0 = right ? 0x00000000u : 0x00070007u, TG1 = right ? 0x00010001u : 0x00060006u, TG2 = right ? 0x00020002u : 0x00050005u,
	               TG3 = right ? 0x00030003u : 0x00040004u, TG4 = right ? 0x00040004u : 0x00030003u, TGX = right ? 0u : 0x00070007u;
	const uint32_t M1X = M1 | TG1, M1Y = M1 | TG2, M2X = M2 | TG3, M2Y = M2 | TG4;                    // boundary values of x, y, x2, y2 with their tags
	const uint32_t QE_X = pk_sub(QE, TG1), QE_Y = pk_sub(QE, TG2), QE2_X = pk_sub(QE2, TG3), QE2_Y = pk_sub(QE2, TG4);
	const uint32_t FM = right ? 0xffffffffu : 0xff00ff00u;                                          // (DIR == 2) what of max(a - (z - q), 0) says "the gap stays open": >= 0 (the tag counts) or > 0
	// ---- state: cells t0 = 32c + 2gl (low half) and t0 + 1 (high half)
	uint32_t X[NP], V[NP], X2[NP], U[NP], Y[NP], Y2[NP], S[NP], H[NP];
	// the score tables of the lane's two target bases (byte k = score against query base k; all 0xff = sc_N for a target N) wait in LDS: a 64-bit read
	// per superblock step instead of two registers per superblock held for the whole job
	uint2 *const wt = reinterpret_cast<uint2 *>(L.wtab) + gl;
	{
		const uint32_t misrep = 0x01010101u * (uint32_t)(uint8_t)(int8_t)sc_mis;
#pragma unroll
		for (int c = 0; c < NP; ++c) {
			const int t0 = 32 * c + 2 * gl;
			const uint32_t tb0 = t0 < tlen ? L.tbuf[t0] : 0, tb1 = t0 + 1 < tlen ? L.tbuf[t0 + 1] : 0;
			X[c] = M1X; V[c] = M1; X2[c] = M2X; U[c] = M1; Y[c] = M1Y; Y2[c] = M2Y; S[c] = TG0; H[c] = 0;
			wt[16 * c] = make_uint2(tb0 >= 4 ? 0xffffffffu : ((misrep & ~(0xffu << (8 * tb0))) | (uint32_t)(uint8_t)(int8_t)sc_mch << (8 * tb0)),
			                        tb1 >= 4 ? 0xffffffffu : ((misrep & ~(0xffu << (8 * tb1))) | (uint32_t)(uint8_t)(int8_t)sc_mch << (8 * tb1)));
		}
	}
	const uint32_t TT = (uint32_t)(2 * gl) | (uint32_t)(2 * gl + 1) << 16;     // (t0, t1) of superblock 0; superblock c: + 32 per half
	const int myb = gl >> 3;                                                   // this lane's block inside a superblock
	GSYNC();
	const size_t prow = (size_t)n_col_ * 16;
	// (round 6) the traceback bytes of this form always go to the group's HBM workspace: with the LDS tile of the <= 4-block jobs (JobLds::kPtb) it gave
	// CIGARs that differ from the reference's, the cause was never found, and the code path is gone -- that class runs d_ksw_reg (k_ext_dp<4, ., ., false>)
	static_assert(LT::kPtb == 0, "d_ksw_pk keeps its traceback in the HBM workspace: instantiate it with an LDS structure without a traceback tile");
	uint8_t *const ptb = ws.p;
	int last_st = -1, last_en = -1, r;
	const int n_rows = qlen + tlen - 1;
	for (r = 0; r < n_rows; ++r) {
		int st, en;
		d_row_bounds(r, qlen, tlen, w, st, en);
		if (st > en) { ez.zdropped = 1; break; }
		const int st0 = st, en0 = en;
		st = st / 16 * 16; en = (en + 16) / 16 * 16 - 1;
		const int st_ = st >> 4, en_ = en >> 4;
		const int cover_end = st0 + ((en0 - st0) >> 4) * 16 + 15;
		const int tend = tlen_ * 16;
		// ---- boundary conditions (:141-157): x, v, x2 left of the first cell of block st_
		uint32_t cX = M1X, cV = M1, cX2 = M2X;
		if (st > 0) {
			if (st - 1 >= last_st && st - 1 <= last_en) {                      // (only in a row whose first block moved)
				const int cb = (st_ - 1) >> 1;                                 // cell st - 1 = the last cell of block st_ - 1: high half of lane 7 or 15 of superblock cb
				uint32_t sx = 0, sv = 0, sx2 = 0;
#pragma unroll
				for (int c = 0; c < NP; ++c) { const bool is = c == cb; sx = is ? X[c] : sx; sv = is ? V[c] : sv; sx2 = is ? X2[c] : sx2; }
				const bool odd = ((st_ - 1) & 1) != 0;
				const uint32_t x15 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sx, DPP_ROW_BCAST15, 0xf, 0xf, false), x7 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sx, DPP_ROW_BCAST7, 0xf, 0xf, false);
				const uint32_t v15 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sv, DPP_ROW_BCAST15, 0xf, 0xf, false), v7 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sv, DPP_ROW_BCAST7, 0xf, 0xf, false);
				const uint32_t y15 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sx2, DPP_ROW_BCAST15, 0xf, 0xf, false), y7 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sx2, DPP_ROW_BCAST7, 0xf, 0xf, false);
				cX = odd ? x15 : x7; cV = odd ? v15 : v7; cX2 = odd ? y15 : y7;   // (the wanted value sits in the HIGH half: that is where the alignbit below takes it from)
			}
		} else {
			const int v1 = r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
			cV = pk_splat8(v1);
		}
		const int ubound = r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
		const uint32_t UB = pk_splat8(ubound);
		// per-row splats for the cell masks
		const uint32_t R_ = pk_splat(en >= r ? r : 0xffff);                     // y[r], y2[r], u[r] (:150-153) only while r is inside the blocks
		const uint32_t ST0 = pk_splat(st0), SONW = pk_splat((cover_end < tend - 1 ? cover_end : tend - 1) - st0), INW = pk_splat(en0 - st0);
		const uint32_t EN0 = pk_splat(en0 > 0 ? en0 : 0xffff);
		uint16_t *const prl = reinterpret_cast<uint16_t *>(ptb + (size_t)r * prow - st + 2 * gl);     // this lane's two bytes of superblock 0; superblock c is 32 bytes further
		const uint8_t *const qrE = selE + (qlen - 1 - r) + 2 * gl, *const qrO = selO + (qlen - r) + 2 * gl;   // selector bytes of this lane's two cells in superblock 0; superblock c is 32 bytes further
		// H[r-1][en0-1] when en0 is the first cell of a superblock (the only case in which it is another superblock's cell)
		uint32_t hprev = 0;
		if (NP > 1 && r > 0 && (en0 & 31) == 0) {
			uint32_t hs = 0; const int cb = (en0 >> 5) - 1;
#pragma unroll
			for (int c = 0; c < NP; ++c) hs = c == cb ? H[c] : hs;
			hprev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hs, DPP_ROW_BCAST15, 0xf, 0xf, false);
		}
		uint32_t bkey = 0x80008000u, bcc = 0;                                   // per half: best 2 H + (end cell) of this lane's cells, and its superblock
		// the cell recurrence (:177-265) on two cells: left neighbours (xt1, vt1, x2t1), the cells' own u, y, y2 and score -> new state and traceback byte
		auto cell = [&](const uint32_t xt1, const uint32_t vt1, const uint32_t x2t1, const uint32_t uo, const uint32_t yo, const uint32_t y2o, const uint32_t so,
		                uint32_t &xn, uint32_t &vn, uint32_t &x2n, uint32_t &un, uint32_t &yn, uint32_t &y2n, uint32_t &d) {
			const uint32_t a = pk_add(xt1, vt1), bb = pk_add(yo, uo), a2 = pk_add(x2t1, vt1), b2 = pk_add(y2o, uo);   // (tagged: x, y, x2, y2 are; so is so)
			uint32_t zk = pk_max(so, a);
			zk = pk_max(zk, bb); zk = pk_max(zk, a2); zk = pk_max(zk, b2);
			d = (zk & 0x00070007u) ^ TGX;
			const uint32_t z = pk_min(zk & 0xff00ff00u, MCH);
			un = pk_sub(z, vt1); vn = pk_sub(z, uo);
			const uint32_t tq = pk_sub(z, QQ), tq2 = pk_sub(z, QQ2);
			const uint32_t aa = pk_sub(a, tq), ab = pk_sub(bb, tq), aa2 = pk_sub(a2, tq2), ab2 = pk_sub(b2, tq2);
			const uint32_t pa = pk_max(aa, 0u), pb = pk_max(ab, 0u), pa2 = pk_max(aa2, 0u), pb2 = pk_max(ab2, 0u);   // != 0 where the value is >= 0 (low byte = its tag, never 0 there); high byte != 0 where it is > 0
			const uint32_t ma = pa & 0xff00ff00u, mb = pb & 0xff00ff00u, ma2 = pa2 & 0xff00ff00u, mb2 = pb2 & 0xff00ff00u;
			xn = pk_sub(ma, QE_X); yn = pk_sub(mb, QE_Y); x2n = pk_sub(ma2, QE2_X); y2n = pk_sub(mb2, QE2_Y);
			if (DIR == 0) d |= pk_minu1(ma) << 3 | pk_minu1(mb) << 4 | pk_minu1(ma2) << 5 | pk_minu1(mb2) << 6;                // the gap stays open: > 0
			else if (DIR == 1) d |= pk_minu1(pa) << 3 | pk_minu1(pb) << 4 | pk_minu1(pa2) << 5 | pk_minu1(pb2) << 6;           // >= 0
			else d |= pk_minu1(pa & FM) << 3 | pk_minu1(pb & FM) << 4 | pk_minu1(pa2 & FM) << 5 | pk_minu1(pb2 & FM) << 6;
		};
#pragma unroll
		for (int c = 0; c < NP; ++c) {
			if (NP >= 3) { if (!(2 * c + 1 >= st_ && 2 * c <= en_)) { continue; } }   // no active lane of the group in this superblock
			// Interior superblock: both blocks active, every cell strictly inside [st0, en0) (so none is the end cell, none is cell r, all take their score
			// and their H) -- no masks, no selects.  Taken when it holds for every group of the wavefront that is still in this superblock.
			const bool interior = 2 * c >= st_ && 2 * c + 1 <= en_ && 32 * c >= st0 && 32 * c + 31 < en0;
			if (__ballot(!interior) == 0) {
				const uint32_t xo = X[c], vo = V[c], x2o = X2[c];
				const uint32_t sel = __builtin_amdgcn_perm((uint32_t)qrO[32 * c], (uint32_t)qrE[32 * c], 0x040c000cu);
				const uint2 wv = wt[16 * c];
				const uint32_t so = (__builtin_amdgcn_perm(wv.y, wv.x, sel) & 0xff00ff00u) | TG0;
				const uint32_t xl = (uint32_t)d_dpp_shr1((int)cX, (int)xo), vl = (uint32_t)d_dpp_shr1((int)cV, (int)vo), x2l = (uint32_t)d_dpp_shr1((int)cX2, (int)x2o);
				const uint32_t xt1 = __builtin_amdgcn_alignbit(xo, xl, 16), vt1 = __builtin_amdgcn_alignbit(vo, vl, 16), x2t1 = __builtin_amdgcn_alignbit(x2o, x2l, 16);
				cX = (uint32_t)__builtin_amdgcn_mov_dpp((int)xo, DPP_ROW_BCAST15, 0xf, 0xf, true); cV = (uint32_t)__builtin_amdgcn_mov_dpp((int)vo, DPP_ROW_BCAST15, 0xf, 0xf, true); cX2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)x2o, DPP_ROW_BCAST15, 0xf, 0xf, true);
				uint32_t xn, vn, x2n, un, yn, y2n, d;
				cell(xt1, vt1, x2t1, U[c], Y[c], Y2[c], so, xn, vn, x2n, un, yn, y2n, d);
				X[c] = xn; V[c] = vn; X2[c] = x2n; U[c] = un; Y[c] = yn; Y2[c] = y2n; S[c] = so;
				prl[16 * c] = (uint16_t)__builtin_amdgcn_perm(0u, d, 0x0c0c0200u);
				const uint32_t hn = pk_add(H[c], pk_sar8(vn));
				H[c] = hn;
				const uint32_t nk = pk_max(bkey, pk_add(hn, hn));
				const uint32_t ch = pk_minu1(nk ^ bkey);
				bcc = pk_u(pk_s(bcc) + pk_s(ch) * (pk_s(pk_splat(c)) - pk_s(bcc)));
				bkey = nk;
				continue;
			}
			const int b = 2 * c + myb;
			const bool act = b >= st_ && b <= en_;
			const uint32_t T = TT + (uint32_t)c * 0x00200020u;
			uint32_t xo = X[c], vo = V[c], x2o = X2[c], uo = U[c], yo = Y[c], y2o = Y2[c], so = S[c];
			{   // y[r], y2[r], u[r]
				const uint32_t m = pk_mask_eq(T, R_);
				yo = d_bfi(m, M1Y, yo); y2o = d_bfi(m, M2Y, y2o); uo = d_bfi(m, UB, uo);
			}
			{   // score bytes (:158-176) of the cells [st0, cover_end] below the padded target length
				const uint32_t sel = __builtin_amdgcn_perm((uint32_t)qrO[32 * c], (uint32_t)qrE[32 * c], 0x040c000cu);   // E << 8 | O << 24
				const uint2 wv = wt[16 * c];
				const uint32_t sc = __builtin_amdgcn_perm(wv.y, wv.x, sel);
				so = d_bfi(pk_mask_in(T, ST0, SONW) & 0xff00ff00u, sc, so);          // (the permute leaves a score byte in the low byte of either half as well: only the high bytes are taken)
			}
			// left neighbours: the low half's is the high half of the lane below (or the carry), the high half's is the lane's own low half
			const bool inj = gl == 8 && st_ == 2 * c + 1;                       // the first cell of the first block takes the boundary values, not lane 7's
			uint32_t xl = (uint32_t)d_dpp_shr1((int)cX, (int)xo), vl = (uint32_t)d_dpp_shr1((int)cV, (int)vo), x2l = (uint32_t)d_dpp_shr1((int)cX2, (int)x2o);
			xl = inj ? cX : xl; vl = inj ? cV : vl; x2l = inj ? cX2 : x2l;
			const uint32_t xt1 = __builtin_amdgcn_alignbit(xo, xl, 16), vt1 = __builtin_amdgcn_alignbit(vo, vl, 16), x2t1 = __builtin_amdgcn_alignbit(x2o, x2l, 16);
			{   // carry into the next superblock: lane 15's cells, if its block is active
				const bool a15 = 2 * c + 1 >= st_ && 2 * c + 1 <= en_;
				const uint32_t nx = (uint32_t)__builtin_amdgcn_mov_dpp((int)xo, DPP_ROW_BCAST15, 0xf, 0xf, true), nv = (uint32_t)__builtin_amdgcn_mov_dpp((int)vo, DPP_ROW_BCAST15, 0xf, 0xf, true), nx2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)x2o, DPP_ROW_BCAST15, 0xf, 0xf, true);
				cX = a15 ? nx : cX; cV = a15 ? nv : cV; cX2 = a15 ? nx2 : cX2;
			}
			uint32_t xn, vn, x2n, un, yn, y2n, d;
			cell(xt1, vt1, x2t1, uo, yo, y2o, so, xn, vn, x2n, un, yn, y2n, d);
			X[c] = act ? xn : X[c]; V[c] = act ? vn : V[c]; X2[c] = act ? x2n : X2[c]; U[c] = act ? un : U[c];
			Y[c] = act ? yn : Y[c]; Y2[c] = act ? y2n : Y2[c]; S[c] = act ? so : S[c];
			if (act) prl[16 * c] = (uint16_t)__builtin_amdgcn_perm(0u, d, 0x0c0c0200u);
			// ---- exact max (:307-349)
			{
				const uint32_t hold = H[c];
				const uint32_t hsh = (uint32_t)d_dpp_shr1((int)hprev, (int)hold);
				const uint32_t hl = __builtin_amdgcn_alignbit(hold, hsh, 16);       // H[r-1][t-1]
				const uint32_t men = pk_mask_eq(T, EN0);
				uint32_t h = d_bfi(men, pk_add(hl, pk_sar8(un)), pk_add(hold, pk_sar8(vn)));
				uint32_t upd = (r > 0 && act) ? pk_mask_in(T, ST0, INW) : 0u;
				if (c == 0 && r == 0) { const uint32_t m0 = (gl == 0 && act) ? 0x0000ffffu : 0u; h = d_bfi(m0, pk_sub(pk_sar8(vn), pk_splat(qe)), h); upd |= m0; }   // H[0][0] = v - qe
				const uint32_t hn = d_bfi(upd, h, hold);
				H[c] = hn;
				// candidate key 2 H + (end cell or cell (0, 0)): a strictly greater key takes over -- among equal H the end cell (evaluated first by the
				// reference) wins, then the earlier superblock (its cell comes earlier in the reference's order, see d_ksw_reg)
				const uint32_t endbit = (men | (r == 0 ? 0x0000ffffu : 0u)) & 0x00010001u;
				const uint32_t key = d_bfi(upd, pk_add(pk_add(hn, hn), endbit), 0x80008000u);
				const uint32_t nk = pk_max(bkey, key);
				const uint32_t ch = pk_minu1(nk ^ bkey);                             // 1 where the best changed
				bcc = pk_u(pk_s(bcc) + pk_s(ch) * (pk_s(pk_splat(c)) - pk_s(bcc)));
				bkey = nk;
			}
		}
		// ---- this lane's best as (H << 16 | 0xffff - ord), ord = rank in the reference's evaluation order; then the group's
		const int en1 = st0 + (en0 - st0) / 4 * 4;
		int key32 = (int)0x80000000;
#pragma unroll
		for (int hh = 0; hh < 2; ++hh) {
			const int k16 = (int)(int16_t)(hh ? bkey >> 16 : bkey & 0xffffu), cc = (int)(hh ? bcc >> 16 : bcc & 0xffffu);
			const int tt = 32 * cc + 2 * gl + hh, dt = tt - st0;
			const int ord_a = 1 + (dt & 3) * 4096 + (dt >> 2), ord_b = 1 + 4 * 4096 + (tt - en1);
			int ord = tt < en1 ? ord_a : ord_b;
			ord = (k16 & 1) ? 0 : ord;
			const int k = k16 == -32768 ? (int)0x80000000 : (k16 >> 1) * 65536 + (0xffff - ord);
			key32 = k > key32 ? k : key32;
		}
		int max_H, max_t;
		{
			int ord, k = key32;
			{ const int o = __builtin_amdgcn_update_dpp(k, k, DPP_QUAD_XOR1, 0xf, 0xf, false); k = o > k ? o : k; }
			{ const int o = __builtin_amdgcn_update_dpp(k, k, DPP_QUAD_XOR2, 0xf, 0xf, false); k = o > k ? o : k; }
			{ const int o = __builtin_amdgcn_update_dpp(k, k, DPP_HALF_MIRROR, 0xf, 0xf, false); k = o > k ? o : k; }
			{ const int o = __builtin_amdgcn_update_dpp(k, k, DPP_ROW_MIRROR, 0xf, 0xf, false); k = o > k ? o : k; }
			max_H = k >> 16; ord = 0xffff - (k & 0xffff);
			max_t = r == 0 ? 0 : ord == 0 ? en0 : ord < 1 + 4 * 4096 ? st0 + ((ord - 1) & 4095) * 4 + ((ord - 1) >> 12) : en1 + (ord - 1 - 4 * 4096);
		}
		auto h_of = [&](int t) -> int {                                          // H of cell t (group-uniform t)
			uint32_t hs = 0;
#pragma unroll
			for (int c = 0; c < NP; ++c) hs = c == (t >> 5) ? H[c] : hs;
			const uint32_t v = (uint32_t)__shfl((int)hs, (t & 31) >> 1, GW);
			return (int)(int16_t)((t & 1) ? v >> 16 : v & 0xffffu);
		};
		if (r - st0 == qlen - 1) { const int hs = h_of(st0); if (hs > ez.mqe) { ez.mqe = hs; ez.mqe_t = st0; } }   // :353-354
		bool brk = false;                                                    // ksw_apply_zdrop, ksw2.h:160-176
		if (max_H > ez.max) { ez.max = max_H; ez.max_t = max_t; ez.max_q = r - max_t; }
		else if (max_t >= ez.max_t && r - max_t >= ez.max_q) {
			const int tl = max_t - ez.max_t, ql = (r - max_t) - ez.max_q, l = tl > ql ? tl - ql : ql - tl;
			if (zdrop >= 0 && ez.max - max_H > zdrop + l * e2) { ez.zdropped = 1; brk = true; }
		}
		if (brk) break;
		if (r == qlen + tlen - 2 && en0 == tlen - 1) ez.score = h_of(tlen - 1);
		last_st = st; last_en = en;
	}
	GSYNC();
	if (do_bt) {
		const int rev_cigar = !!(flag & EZ_REV_CIGAR);
		CigW cw{L.ezc, 0, AL_LCIG, ws.ezc, 0xffffffffu};
		if (!ez.zdropped && !(flag & EZ_EXTZ_ONLY)) d_backtrack(ptb, n_col_ * 16, qlen, tlen, w, rev_cigar, tlen - 1, qlen - 1, cw);
		else if (!ez.zdropped && (flag & EZ_EXTZ_ONLY) && ez.mqe + end_bonus > ez.max) { ez.reach_end = 1; d_backtrack(ptb, n_col_ * 16, qlen, tlen, w, rev_cigar, ez.mqe_t, qlen - 1, cw); }
		else if (ez.max_t >= 0 && ez.max_q >= 0) d_backtrack(ptb, n_col_ * 16, qlen, tlen, w, rev_cigar, ez.max_t, ez.max_q, cw);
		ez.n_cigar = cw.n; ws.cur_ezc = cw.c;
	}
	GSYNC();
}
