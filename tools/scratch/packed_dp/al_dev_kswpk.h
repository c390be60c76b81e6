// al_dev_kswpk.h -- the extension DP (ksw_extd2_sse, ksw2_extd2_sse.c:26-393) with TWO cells per lane and instruction.
//
// d_ksw_reg (al_dev_ksw.h) keeps the reference's seven int8 rows byte-packed and spends one 32-bit VALU instruction per cell and
// operation (~130 instructions per 16-cell block and row).  Here every row lives in packed 16-bit halves (v_pk_add_u16, v_pk_max_i16 ...):
// lane l of the 16-lane group owns cell 16 b + l of TWO blocks per register, the blocks b and b + D of a group of 2 D blocks.  The
// anti-diagonal's window is about a dozen blocks wide and slides one block at a time, so with D = 6 both halves of most registers are
// inside it: a row costs half as many block steps.
//   * An int8 value v of the reference is kept as v << 8 in its 16-bit half: 16-bit adds and subtracts then wrap exactly where the
//     reference's _mm_add_epi8 / _mm_sub_epi8 wrap, signed max / min / compares are unchanged, and the low byte is free --
//   * -- for the priority of a candidate: max over (candidate | priority) gives z and the reference's choice among equal candidates at
//     once (left-aligned gaps: the first maximum wins, :206-214; right-aligned: the last, :252-260), without a compare or a select.
//   * Blocks strictly inside the row's range -- for all four jobs of the wavefront -- run a body without any per-cell condition (score
//     store, y / u reset at t == r, range of the H update, scan order of the tail); the others run the same body with the conditions
//     as packed masks.  Registers whose two blocks are both outside the window are skipped.
//   * The query base a cell needs moves one cell to the right per row: it is shifted along with the x / v / x2 neighbours (DPP) instead
//     of being fetched from LDS per cell.
// Traceback bytes, row maxima, z-drop and the backtrack are d_ksw_reg's (same bytes, same d_backtrack).
#pragma once
#include <type_traits>
#include <utility>

template <class F, int... I> __device__ __forceinline__ void al_static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void al_static_for(F &&f) { al_static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{}); }

typedef short pk_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short pk_u2 __attribute__((ext_vector_type(2)));
#define PKS(v) __builtin_bit_cast(pk_s2, (uint32_t)(v))
#define PKU(v) __builtin_bit_cast(pk_u2, (uint32_t)(v))
#define PKR(v) __builtin_bit_cast(uint32_t, (v))
// The packed instructions are written out: left to itself the compiler recognises the mask idioms below (sign of a difference, clamp to
// 0 / 1) as comparisons and turns them back into one compare + select + byte permute per half -- three times the instructions.
#define PK_ASM2(NAME, INSN) __device__ __forceinline__ uint32_t NAME(uint32_t a, uint32_t b) { uint32_t d; asm(INSN " %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
PK_ASM2(pk_add, "v_pk_add_u16") PK_ASM2(pk_sub, "v_pk_sub_i16") PK_ASM2(pk_max, "v_pk_max_i16") PK_ASM2(pk_min, "v_pk_min_i16") PK_ASM2(pk_maxu, "v_pk_max_u16") PK_ASM2(pk_minu, "v_pk_min_u16")
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t m, uint32_t c) { uint32_t d; asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(m), "v"(c)); return d; }
__device__ __forceinline__ uint32_t pk_sgn(uint32_t a) { uint32_t d; asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(d) : "v"(a)); return d; }   // 0xffff where the half is negative
__device__ __forceinline__ uint32_t pk_asr3(uint32_t a) { uint32_t d; asm("v_pk_ashrrev_i16 %0, 3, %1 op_sel_hi:[0,1]" : "=v"(d) : "v"(a)); return d; }
__device__ __forceinline__ uint32_t pk_lsr2(uint32_t a) { uint32_t d; asm("v_pk_lshrrev_b16 %0, 2, %1 op_sel_hi:[0,1]" : "=v"(d) : "v"(a)); return d; }
__device__ __forceinline__ uint32_t pk_max0(uint32_t a) { uint32_t d; asm("v_pk_max_i16 %0, %1, 0" : "=v"(d) : "v"(a)); return d; }                                  // max(a, 0)
__device__ __forceinline__ uint32_t pk_neg(uint32_t a) { uint32_t d; asm("v_pk_sub_i16 %0, 0, %1" : "=v"(d) : "v"(a)); return d; }                                   // 0 - a
__device__ __forceinline__ uint32_t pk_subs(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b)); return d; }        // saturating a - b
__device__ __forceinline__ uint32_t pk_min1u(uint32_t a) { uint32_t d; asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(d) : "v"(a)); return d; }                 // min(a, 1), unsigned
__device__ __forceinline__ uint32_t pk_asr8(uint32_t a) { uint32_t d; asm("v_pk_ashrrev_i16 %0, 8, %1 op_sel_hi:[0,1]" : "=v"(d) : "v"(a)); return d; }
__device__ __forceinline__ uint32_t pk_shl8(uint32_t a) { uint32_t d; asm("v_pk_lshlrev_b16 %0, 8, %1 op_sel_hi:[0,1]" : "=v"(d) : "v"(a)); return d; }
__device__ __forceinline__ uint32_t pk_dup(int v) { return ((uint32_t)v & 0xffffu) * 0x10001u; }
__device__ __forceinline__ uint32_t pk_lt(uint32_t a, uint32_t b) { return pk_sgn(pk_sub(a, b)); }                           // 0xffff where a < b (small values)
__device__ __forceinline__ uint32_t pk_ne(uint32_t a, uint32_t b, uint32_t zero) { const uint32_t d = pk_sub(a, b); return pk_sgn(d | pk_sub(zero, d)); }
__device__ __forceinline__ uint32_t pk_bfi(uint32_t m, uint32_t a, uint32_t b) { return (m & a) | (~m & b); }                   // m ? a : b, bit by bit
#define PK_ROW_ROR1 0x121
__device__ __forceinline__ uint32_t pk_ror1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, PK_ROW_ROR1, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t pk_shr1(uint32_t carry, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)v, 0x111 /* row_shr:1 */, 0xf, 0xf, false); }

__device__ __forceinline__ uint32_t pk_ror1u(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, PK_ROW_ROR1, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t pk_scaled(int v) { return pk_dup((v & 0xff) << 8); }      // an int8 constant in both halves, scaled

// D blocks per half, NG groups of 2 D blocks: NR = D * NG registers per row, targets up to 32 D NG bases.  Returns false (before any work)
// when the job does not fit the registers or the scoring does not fit int8; the caller then leaves the job to the byte-packed kernel.
template <int D, int NG, class LT>
__device__ __forceinline__ bool d_ksw_pk(LT &L, const int gl, GroupWs &ws, int qlen, int tlen, const AlParams &P,
                                         int w, int zdrop, int end_bonus, int flag, EzD &ez)
{
	constexpr int NR = D * NG;
	int q = P.q, e = P.e, q2 = P.q2, e2 = P.e2;
	if (q2 + e2 < q + e) { int t = q; q = q2; q2 = t; t = e; e = e2; e2 = t; }
	const int qe = q + e;
	const int sc_mch = (int8_t)P.a, sc_mis = (int8_t)(-P.b), sc_amb = (int8_t)(P.sc_ambi > 0 ? -P.sc_ambi : P.sc_ambi);
	const int sc_N = sc_amb == 0 ? (int8_t)(-e2) : sc_amb;
	if (q2 + e2 > 127 || qe > 127 || q < 0 || q2 < 0 || e < 0 || e2 < 0) return false;
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	const int tlen_ = (tlen + 15) / 16;
	if (tlen_ > 2 * NR || qlen < 1 || tlen < 1) return false;
	int n_col_ = qlen < tlen ? qlen : tlen;
	n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
	int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
	const bool right = (flag & EZ_RIGHT) != 0;
	const uint32_t DX = right ? 0u : 0x00070007u;
	// priorities of the five candidates s, a, b, a2, b2 in the low three bits (left-aligned: the first maximum wins; right-aligned: the last, s only if alone)
	const uint32_t C0 = pk_dup(right ? 0 : 7), C1 = pk_dup(right ? 1 : 6), C2 = pk_dup(right ? 2 : 5), C3 = pk_dup(right ? 3 : 4), C4 = pk_dup(right ? 4 : 3);
	const uint32_t TH = pk_dup(right ? -1 : 0);                           // "continue the gap" flags: a > 0 (left-aligned, :216-231), a >= 0 (right-aligned, :262-277)
	uint32_t QEu = pk_dup(qe);
	const uint32_t GL2 = pk_dup(gl);
	// wavefront-uniform constants the packed instructions take as vector operands: held in vector registers once, not copied per use
#define PK_VREG(x) asm volatile("" : "+v"(x))
	uint32_t MCH = pk_scaled(sc_mch), DMIS = pk_dup((sc_mis - sc_mch) * 256), SCN = pk_scaled(sc_N), Qs = pk_scaled(q), Q2s = pk_scaled(q2), QEs = pk_scaled(qe), QE2s = pk_scaled(q2 + e2);
	PK_VREG(MCH); PK_VREG(DMIS); PK_VREG(SCN); PK_VREG(Qs); PK_VREG(Q2s); PK_VREG(QEs); PK_VREG(QE2s); PK_VREG(QEu);
#undef PK_VREG
	const bool lane0 = gl == 0;
	uint32_t X[NR], V[NR], X2[NR], U[NR], Y[NR], Y2[NR], S[NR], TQ[NR], Hh[NR];
#define BLK_LO(rho) (((rho) / D) * 2 * D + ((rho) % D))
#define BLK_HI(rho) (BLK_LO(rho) + D)
	{
		const uint32_t m1 = pk_scaled(-q - e), m2 = pk_scaled(-q2 - e2);
		al_static_for<NR>([&](auto rc) __attribute__((always_inline)) {
			constexpr int rho = decltype(rc)::value;
			const int tl = 16 * BLK_LO(rho) + gl, th = 16 * BLK_HI(rho) + gl;
			X[rho] = V[rho] = U[rho] = Y[rho] = m1; X2[rho] = Y2[rho] = m2; S[rho] = 0; Hh[rho] = 0;
			TQ[rho] = (tl < tlen ? (uint32_t)L.tbuf[tl] : 0u) | (th < tlen ? (uint32_t)L.tbuf[th] : 0u) << 16;
		});
	}
	GSYNC();
	const size_t prow = (size_t)n_col_ * 16;
	uint8_t *const ptb = (size_t)(qlen + tlen - 1) * prow <= AL_LPTB ? L.ptb : ws.p;
	int last_st = -1, last_en = -1, r;
	// value of block b (group-uniform, runtime) out of a packed register array, without branches: sign-extended half
#define PK_PICK(ARR, bsel, out) do { uint32_t acc__ = 0; al_static_for<NR>([&](auto rc__) __attribute__((always_inline)) { constexpr int r__ = decltype(rc__)::value; \
		acc__ |= (ARR[r__] & ((bsel) == BLK_LO(r__) ? 0xffffu : 0u)) | ((ARR[r__] >> 16) & ((bsel) == BLK_HI(r__) ? 0xffffu : 0u)); }); (out) = (int)(int16_t)acc__; } while (0)
	for (r = 0; r < qlen + tlen - 1; ++r) {
		int st, en;
		d_row_bounds(r, qlen, tlen, w, st, en);
		if (st > en) { ez.zdropped = 1; break; }
		const int st0 = st, en0 = en;
		st = st / 16 * 16; en = (en + 16) / 16 * 16 - 1;
		const int st_ = st >> 4, en_ = en >> 4;
		const int cover_end = st0 + ((en0 - st0) >> 4) * 16 + 15;
		const int en1 = st0 + (en0 - st0) / 4 * 4;
		const int rowv = r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2;   // v1 at st == 0 and u[r] (:141-153)
		// ---- the left neighbour of the row's first block (:141-149): x8[st-1] ... of the last row if that cell was computed then, constants otherwise
		uint32_t BWX = pk_scaled(-q - e), BWV = pk_scaled(st > 0 ? -q - e : rowv), BWX2 = pk_scaled(-q2 - e2);
		{
			const bool inr1 = st > 0 && st - 1 >= last_st && st - 1 <= last_en;
			if (__any(inr1)) {
				int sx, sv, sx2; const int bp = st_ - 1;
				PK_PICK(X, bp, sx); PK_PICK(V, bp, sv); PK_PICK(X2, bp, sx2);
				const uint32_t pk3 = ((uint32_t)sx >> 8 & 0xffu) | ((uint32_t)sv & 0xff00u) | ((uint32_t)sx2 & 0xff00u) << 8;   // the values are int8: one cross-lane read for the three
				const uint32_t av = (uint32_t)__shfl((int)pk3, GW - 1, GW);
				if (inr1) { BWX = pk_dup((int)(av & 0xff) << 8); BWV = pk_dup((int)(av & 0xff00)); BWX2 = pk_dup((int)(av >> 8 & 0xff00)); }
			}
		}
		uint8_t *const prl = ptb + (size_t)r * prow - st + gl;
		const int be = en0 >> 4;
		int hprev15 = 0;                                                 // H[r-1][en0-1] when en0 is the first lane of its block
		{
			const bool need = (en0 & 15) == 0 && r > 0 && be > 0;
			if (__any(need)) { int hs; PK_PICK(Hh, be - 1, hs); hs = __shfl(hs, GW - 1, GW); hprev15 = need ? hs : 0; }
		}
		const bool enr = en >= r;
		const int ib_lo = st_ + 1, ib_hi = (en1 >> 4) - 1;                  // blocks strictly inside the range (none if ib_hi < ib_lo)
		const int cc = gl - st0;
		const uint32_t CKp = pk_dup(0xffff - (1 + ((cc & 3) << 12) + (cc >> 2)));       // 0xffff - ord of the four-lane scan for the cell of block 0; block b: - 4 b
		const uint32_t qc_in = r < qlen ? (uint32_t)L.qbuf[r] << 4 : 0u;    // the query base that enters at cell 0 (cell t pairs with query[r - t])
		const bool want_mqe = r - st0 == qlen - 1;
		int key32 = (int)0x80000000;
		uint32_t hcap = 0;                                                  // H of the cell t == st0 (mqe, :353-354)
		uint32_t svX = 0, svV = 0, svX2 = 0, svQ = 0;                       // the previous register's old values rotated one lane up: lane 0 holds its lane 15
		uint32_t svX_g = 0, svV_g = 0, svX2_g = 0, svQ_g = 0;               // ... of the last register of the previous group
		al_static_for<NR>([&](auto rc) __attribute__((always_inline)) {
			constexpr int rho = decltype(rc)::value;
			constexpr int bl = BLK_LO(rho), bh = BLK_HI(rho);
			constexpr uint32_t BLH = (uint32_t)bl | (uint32_t)bh << 16;
			// ---- query bases: every register, every row (a block may open later with its bases in place)
			const uint32_t tqo = TQ[rho];
			uint32_t cQ;
			if constexpr (rho % D == 0) {
				const uint32_t hi_src = pk_ror1u(TQ[rho + D - 1]);           // block bh - 1 = low half of the group's last register (not yet shifted)
				cQ = rho == 0 ? __builtin_amdgcn_perm(hi_src, qc_in, 0x05040100u) : __builtin_amdgcn_perm(hi_src, svQ_g, 0x05040302u);
			} else cQ = svQ;
			const uint32_t q_rot = pk_ror1u(tqo);
			const uint32_t tq = pk_bfi(0x00700070u, lane0 ? cQ : q_rot, tqo);
			TQ[rho] = tq;
			svQ = q_rot; if constexpr (rho % D == D - 1) svQ_g = q_rot;
			// ---- is there work in this register?
			const bool act_lo = bl >= st_ && bl <= en_, act_hi = bh >= st_ && bh <= en_;
			if (!__any(act_lo || act_hi)) { if constexpr (rho % D == D - 1) { svX_g = 0; svV_g = 0; svX2_g = 0; } return; }
			const bool interior = bl >= ib_lo && bh <= ib_hi;               // (bl < bh: both blocks inside)
			const uint32_t xo = X[rho], vo = V[rho], x2o = X2[rho];
			uint32_t cX, cV, cX2;
			if constexpr (rho % D == 0) {
				const uint32_t hX = pk_ror1u(X[rho + D - 1]), hV = pk_ror1u(V[rho + D - 1]), hX2 = pk_ror1u(X2[rho + D - 1]);   // (old values: the group's last register comes after this one)
				cX = __builtin_amdgcn_perm(hX, svX_g, 0x05040302u); cV = __builtin_amdgcn_perm(hV, svV_g, 0x05040302u); cX2 = __builtin_amdgcn_perm(hX2, svX2_g, 0x05040302u);
			} else { cX = svX; cV = svV; cX2 = svX2; }
			uint32_t rX, rV, rX2;
			rX = pk_ror1u(xo); rV = pk_ror1u(vo); rX2 = pk_ror1u(x2o);
			if constexpr (rho % D == D - 1) { svX_g = rX; svV_g = rV; svX2_g = rX2; }
			svX = rX; svV = rV; svX2 = rX2;
			auto body = [&](auto gen_c) __attribute__((always_inline)) {
				constexpr bool GEN = decltype(gen_c)::value;
				uint32_t T = 0, inact = 0, son_fail = 0, upd_fail = 0, isen_ne = 0xffffffffu, r0c_ne = 0xffffffffu;
				uint32_t uo = U[rho], yo = Y[rho], y2o = Y2[rho];
				if (GEN) {
					T = pk_add(GL2, 16u * BLH);
					const uint32_t STA = pk_dup(st), ENA = pk_dup(en), ST0p = pk_dup(st0), EN0p = pk_dup(en0);
					inact = pk_lt(T, STA) | pk_lt(ENA, T);
					const uint32_t lt_st0 = pk_lt(T, ST0p);
					son_fail = inact | lt_st0 | pk_lt(pk_dup(cover_end), T);
					const uint32_t isr_fail = inact | pk_ne(T, pk_dup(enr ? r : -1), 0u);    // y[r], y2[r], u[r] (:150-153)
					yo = pk_bfi(isr_fail, yo, pk_scaled(-q - e)); y2o = pk_bfi(isr_fail, y2o, pk_scaled(-q2 - e2)); uo = pk_bfi(isr_fail, uo, pk_scaled(rowv));
					isen_ne = pk_ne(T, EN0p, 0u); r0c_ne = pk_ne(T, pk_dup(r == 0 ? 0 : -1), 0u) | inact;
					upd_fail = (inact | lt_st0 | pk_lt(EN0p, T) | (r > 0 ? 0u : 0xffffffffu)) & r0c_ne;
					const uint32_t eq_st = ~pk_ne(BLH, pk_dup(st_), 0u);          // the row's first block takes the boundary values as its left neighbour
					cX = pk_bfi(eq_st, BWX, cX); cV = pk_bfi(eq_st, BWV, cV); cX2 = pk_bfi(eq_st, BWX2, cX2);
				}
				const uint32_t xl = lane0 ? cX : rX, vl = lane0 ? cV : rV, x2l = lane0 ? cX2 : rX2;
				// score bytes (:158-176)
				const uint32_t tb = tq & 0x00070007u, qb = (tq >> 4) & 0x00070007u;
				uint32_t sc = pk_mad(pk_min1u(tb ^ qb), DMIS, MCH);
				sc = pk_bfi(pk_neg(pk_lsr2(pk_maxu(tb, qb))), SCN, sc);
				const uint32_t s_new = GEN ? pk_bfi(son_fail, S[rho], sc) : sc;
				// the cell (:180-262): int8 value << 8, the candidate's priority in the low bits
				uint32_t a = pk_add(xl, vl), bb = pk_add(yo, uo), a2 = pk_add(x2l, vl), b2 = pk_add(y2o, uo);
				const uint32_t m = pk_max(pk_max(s_new | C0, a | C1), pk_max(bb | C2, pk_max(a2 | C3, b2 | C4)));
				uint32_t dd = (m & 0x00070007u) ^ DX;
				const uint32_t z = pk_min(m & 0xff00ff00u, MCH);
				const uint32_t un = pk_sub(z, vl), vn = pk_sub(z, uo);
				uint32_t tmp = pk_sub(z, Qs); a = pk_sub(a, tmp); bb = pk_sub(bb, tmp);
				tmp = pk_sub(z, Q2s); a2 = pk_sub(a2, tmp); b2 = pk_sub(b2, tmp);
				const uint32_t xn = pk_sub(pk_max0(a), QEs), yn = pk_sub(pk_max0(bb), QEs), x2n = pk_sub(pk_max0(a2), QE2s), y2n = pk_sub(pk_max0(b2), QE2s);
				// the sign of TH - cand (saturating) is "cand continues": moved to the flag's bit of both halves
				dd = pk_bfi(0x00080008u, pk_subs(TH, a) >> 12, dd); dd = pk_bfi(0x00100010u, pk_subs(TH, bb) >> 11, dd);
				dd = pk_bfi(0x00200020u, pk_subs(TH, a2) >> 10, dd); dd = pk_bfi(0x00400040u, pk_subs(TH, b2) >> 9, dd);
				if (GEN) {
					X[rho] = pk_bfi(inact, xo, xn); V[rho] = pk_bfi(inact, vo, vn); X2[rho] = pk_bfi(inact, x2o, x2n);
					U[rho] = pk_bfi(inact, U[rho], un); Y[rho] = pk_bfi(inact, Y[rho], yn); Y2[rho] = pk_bfi(inact, Y2[rho], y2n); S[rho] = pk_bfi(inact, S[rho], s_new);
					if (!(inact & 0xffffu)) prl[16 * bl] = (uint8_t)dd;
					if (!(inact >> 16)) prl[16 * bh] = (uint8_t)(dd >> 16);
				} else {
					X[rho] = xn; V[rho] = vn; X2[rho] = x2n; U[rho] = un; Y[rho] = yn; Y2[rho] = y2n; S[rho] = s_new;
					prl[16 * bl] = (uint8_t)dd; prl[16 * bh] = (uint8_t)(dd >> 16);
				}
				// exact max (:307-349): H row and this lane's candidates
				const uint32_t ho = Hh[rho];
				const uint32_t vn8 = pk_asr8(vn);
				const uint32_t hv_ = pk_add(ho, vn8);
				if (GEN) {
					const uint32_t ho_rot = pk_ror1u(ho);
					const uint32_t hl = lane0 ? pk_dup(hprev15) : ho_rot;            // H[r-1][t-1]: consumed by the cell t == en0 only
					uint32_t h = pk_bfi(isen_ne | (en0 > 0 ? 0u : 0xffffffffu), hv_, pk_add(hl, pk_asr8(un)));
					h = pk_bfi(r0c_ne, h, pk_sub(vn8, QEu));
					const uint32_t hn = pk_bfi(upd_fail, ho, h);
					Hh[rho] = hn;
					hcap |= hn & ~pk_ne(T, pk_dup(st0), 0u) & ~inact;
					uint32_t low = pk_bfi(pk_lt(T, pk_dup(en1)), pk_sub(CKp, 4u * BLH), pk_sub(pk_dup(0xffff - 16385 + en1), T));
					low = pk_bfi(isen_ne & r0c_ne, low, 0xffffffffu);
					const uint32_t klo = (h << 16) | (low & 0xffffu), khi = (h & 0xffff0000u) | (low >> 16);
					const uint32_t flo = (uint32_t)((int32_t)(upd_fail << 16) >> 31), fhi = (uint32_t)((int32_t)upd_fail >> 31);   // the halves' masks, 32 bits wide
					const int k1 = (int)((klo & ~flo) | (flo & 0x80000000u)), k2 = (int)((khi & ~fhi) | (fhi & 0x80000000u));
					key32 = key32 > k1 ? key32 : k1; key32 = key32 > k2 ? key32 : k2;
				} else {
					Hh[rho] = hv_;
					const uint32_t low = pk_sub(CKp, 4u * BLH);
					const int klo = (int)__builtin_amdgcn_perm(hv_, low, 0x05040100u), khi = (int)__builtin_amdgcn_perm(hv_, low, 0x07060302u);
					key32 = key32 > klo ? key32 : klo; key32 = key32 > khi ? key32 : khi;
				}
			};
			if (__all(interior)) body(std::false_type{}); else body(std::true_type{});
		});
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
		{   // :353-354: H of the cell st0 when it is the query's last row (the lane t == st0 caught it above, in the half its block lives in)
			const int hs = __shfl((int)(int16_t)(hcap | hcap >> 16), st0 & 15, GW);
			if (want_mqe && hs > ez.mqe) { ez.mqe = hs; ez.mqe_t = st0; }
		}
		bool brk = false;                                                    // ksw_apply_zdrop, ksw2.h:160-176
		if (max_H > ez.max) { ez.max = max_H; ez.max_t = max_t; ez.max_q = r - max_t; }
		else if (max_t >= ez.max_t && r - max_t >= ez.max_q) {
			const int tl = max_t - ez.max_t, ql = (r - max_t) - ez.max_q, l = tl > ql ? tl - ql : ql - tl;
			if (zdrop >= 0 && ez.max - max_H > zdrop + l * e2) { ez.zdropped = 1; brk = true; }
		}
		if (brk) break;
		if (r == qlen + tlen - 2 && en0 == tlen - 1) { int hs; PK_PICK(Hh, (tlen - 1) >> 4, hs); ez.score = __shfl(hs, (tlen - 1) & 15, GW); }
		last_st = st; last_en = en;
	}
	GSYNC();
	{
		const int rev_cigar = !!(flag & EZ_REV_CIGAR);
		CigW cw{L.ezc, 0, AL_LCIG, ws.ezc, 0xffffffffu};
		if (!ez.zdropped && !(flag & EZ_EXTZ_ONLY)) d_backtrack(ptb, n_col_ * 16, qlen, tlen, w, rev_cigar, tlen - 1, qlen - 1, cw);
		else if (!ez.zdropped && (flag & EZ_EXTZ_ONLY) && ez.mqe + end_bonus > ez.max) { ez.reach_end = 1; d_backtrack(ptb, n_col_ * 16, qlen, tlen, w, rev_cigar, ez.mqe_t, qlen - 1, cw); }
		else if (ez.max_t >= 0 && ez.max_q >= 0) d_backtrack(ptb, n_col_ * 16, qlen, tlen, w, rev_cigar, ez.max_t, ez.max_q, cw);
		ez.n_cigar = cw.n; ws.cur_ezc = cw.c;
	}
	GSYNC();
	return true;
#undef PK_PICK
#undef BLK_LO
#undef BLK_HI
}
