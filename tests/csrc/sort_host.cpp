// Host build of the device sort restatement (airlift_amd/csrc/al_dev_sort.h) for tests/test_dev_sort_cpu.py.
#define AL_SORT_HOST
#include "al_dev_sort.h"

struct A128 { uint64_t x, y; };
struct Acc128 {
	typedef A128 E; A128 *a;
	uint64_t key(int i) const { return a[i].x; }
	uint64_t keyof(const A128 &e) const { return e.x; }
	A128 get(int i) const { return a[i]; }
	void set(int i, const A128 &e) { a[i] = e; }
};
// permutation sort through a key table, as the chain kernel orders chains (elements are ids, keys are looked up)
struct AccPerm {
	typedef int32_t E; int32_t *t; const uint64_t *keys;
	uint64_t keyof(const int32_t &c) const { return keys[c]; }
	uint64_t key(int i) const { return keys[t[i]]; }
	int32_t get(int i) const { return t[i]; }
	void set(int i, const int32_t &c) { t[i] = c; }
};
extern "C" int t_sort128(A128 *a, int n)
{
	uint16_t scr[AL_RS_SCRATCH / 2];
	Acc128 acc{a};
	return d_rs_sort(acc, n, scr) ? 1 : 0;
}
extern "C" int t_sort_perm(int32_t *t, const uint64_t *keys, int n)
{
	uint16_t scr[AL_RS_SCRATCH / 2];
	AccPerm acc{t, keys};
	return d_rs_sort(acc, n, scr) ? 1 : 0;
}
