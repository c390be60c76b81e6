/* TEST INFRASTRUCTURE ONLY -- driver for the reference build "O-full" (see oracle/Makefile).
 *
 * This file is OUR code.  It links against the reference's minimap2 fork (compiled in place from
 * /root/reference, map.c with the ALSER early return removed) through its public C API
 * (minimap.h:180-348) and plays the role of the fork's main(): `-ax sr` mapping of one or two
 * FASTQ/FASTA files to SAM on stdout.  It exists so that goldens can be generated and the
 * restatement in al_oracle.c can be pinned against the real reference.
 *
 * usage: mm2ref [-t N[,N2,...]] [-R rgline] [-K minibatch] [--seeds] [--alnseq] [--hit-only] ref.fa r1.fq [r2.fq]
 *   -t a,b,c : thread sweep for the CPU baseline of bench.py: the input is mapped once per value (stdout to /dev/null for all
 *              but the last) and "[mm2ref] threads=N K=B index_s=.. map_s=.." is printed on stderr for each, so that the index
 *              build is paid once and is not part of the mapping time; a value N@B maps with a mini-batch of B bases
 *              (options.c:122 sets 50M for `sr`; the fork's -K, main.c:185; suffixes k / M / G), the others with the preset's
 *   --save-index FILE : write the index (mm_idx_dump, index.c:438) after building it; a later run given FILE instead of ref.fa
 *              loads it (mm_idx_reader_open recognises index files), so that several test cases share one index build
 *   --max-occ F : print "[mm2ref] max_occ f=F value=V" (mm_idx_cal_max_occ, index.c:164) for the index and exit without mapping
 *   --seeds  : mm_dbg_flag |= MM_DBG_PRINT_SEED  (RS/SD/CN lines on stderr, map.c:333-338,381-385)
 *   --alnseq : mm_dbg_flag |= MM_DBG_PRINT_ALN_SEQ (align.c:315-338)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>
#include <sys/time.h>
#include "minimap.h"
#include "mmpriv.h"

static double wall(void) { struct timeval tv; gettimeofday(&tv, 0); return tv.tv_sec + 1e-6 * tv.tv_usec; }

int main(int argc, char **argv)
{
	int sweep[16], n_sweep = 0, si;
	long long sweep_k[16], k_default;
	double t_idx0, t_idx;
	mm_idxopt_t io; mm_mapopt_t mo;
	int i, n_threads = 1, nfn = 0;
	const char *rg = 0, *fn[4], *save_idx = 0;
	float max_occ_f = 0.f;
	mm_idx_reader_t *r; mm_idx_t *mi;
	mm_verbose = 1;
	mm_set_opt(0, &io, &mo);
	if (mm_set_opt("sr", &io, &mo) < 0) return 2;
	mo.flag |= MM_F_OUT_SAM | MM_F_CIGAR;                 /* -a, main.c:162 */
	for (i = 1; i < argc; ++i) {
		if (!strcmp(argv[i], "-t") && i + 1 < argc) {
			char *e = argv[++i]; n_sweep = 0;
			while (*e && n_sweep < 16) {
				sweep[n_sweep] = (int)strtol(e, &e, 10); sweep_k[n_sweep] = 0;
				if (*e == '@') { double x = strtod(e + 1, &e); if (*e == 'G' || *e == 'g') x *= 1e9, ++e; else if (*e == 'M' || *e == 'm') x *= 1e6, ++e; else if (*e == 'K' || *e == 'k') x *= 1e3, ++e; sweep_k[n_sweep] = (long long)(x + .499); }
				++n_sweep;
				if (*e == ',') ++e; else break;
			}
			n_threads = 1; for (si = 0; si < n_sweep; ++si) if (sweep[si] > n_threads) n_threads = sweep[si];   /* index build: the largest */
		}
		else if (!strcmp(argv[i], "-R") && i + 1 < argc) rg = argv[++i];
		else if (!strcmp(argv[i], "-K") && i + 1 < argc) mo.mini_batch_size = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--save-index") && i + 1 < argc) save_idx = argv[++i];
		else if (!strcmp(argv[i], "--max-occ") && i + 1 < argc) max_occ_f = atof(argv[++i]);
		else if (!strcmp(argv[i], "--seeds")) mm_dbg_flag |= MM_DBG_PRINT_SEED;
		else if (!strcmp(argv[i], "--alnseq")) mm_dbg_flag |= MM_DBG_PRINT_ALN_SEQ;
		else if (!strcmp(argv[i], "--qname")) mm_dbg_flag |= MM_DBG_PRINT_QNAME;
		else if (!strcmp(argv[i], "--hit-only")) mo.flag |= MM_F_SAM_HIT_ONLY;
		/* numeric options with the letters and meaning of main.c:144-215 (parity tests at non-default settings) */
		else if (!strcmp(argv[i], "-k") && i + 1 < argc) io.k = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-w") && i + 1 < argc) io.w = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-g") && i + 1 < argc) mo.max_gap = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-F") && i + 1 < argc) mo.max_frag_len = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-r") && i + 1 < argc) mo.bw = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-N") && i + 1 < argc) mo.best_n = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-p") && i + 1 < argc) mo.pri_ratio = atof(argv[++i]);
		else if (!strcmp(argv[i], "-M") && i + 1 < argc) mo.mask_level = atof(argv[++i]);
		else if (!strcmp(argv[i], "-n") && i + 1 < argc) mo.min_cnt = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-m") && i + 1 < argc) mo.min_chain_score = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-A") && i + 1 < argc) mo.a = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-B") && i + 1 < argc) mo.b = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-s") && i + 1 < argc) mo.min_dp_max = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-O") && i + 1 < argc) { char *e; mo.q = mo.q2 = strtol(argv[++i], &e, 10); if (*e == ',') mo.q2 = strtol(e + 1, &e, 10); }   /* main.c:226-230 */
		else if (!strcmp(argv[i], "-E") && i + 1 < argc) { char *e; mo.e = mo.e2 = strtol(argv[++i], &e, 10); if (*e == ',') mo.e2 = strtol(e + 1, &e, 10); }
		else if (!strcmp(argv[i], "-z") && i + 1 < argc) { char *e; mo.zdrop = mo.zdrop_inv = strtol(argv[++i], &e, 10); if (*e == ',') mo.zdrop_inv = strtol(e + 1, &e, 10); }
		else if (!strcmp(argv[i], "--end-bonus") && i + 1 < argc) mo.end_bonus = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--max-chain-skip") && i + 1 < argc) mo.max_chain_skip = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--score-N") && i + 1 < argc) mo.sc_ambi = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--seed") && i + 1 < argc) mo.seed = atoi(argv[++i]);
		else if (nfn < 4) fn[nfn++] = argv[i];
	}
	if (nfn < 2 && !(nfn == 1 && max_occ_f > 0.f)) { fprintf(stderr, "usage: mm2ref [opts] ref.fa r1.fq [r2.fq]\n"); return 2; }
	if (mm_check_opt(&io, &mo) < 0) return 2;
	r = mm_idx_reader_open(fn[0], &io, 0);
	if (r == 0) { fprintf(stderr, "mm2ref: cannot open %s\n", fn[0]); return 1; }
	if (n_sweep == 0) { sweep_k[n_sweep] = 0; sweep[n_sweep++] = n_threads; }
	k_default = mo.mini_batch_size;
	t_idx0 = wall();
	while ((mi = mm_idx_reader_read(r, n_threads)) != 0) {
		t_idx = wall() - t_idx0;
		if (save_idx) { FILE *fp = fopen(save_idx, "wb"); if (fp) { mm_idx_dump(fp, mi); fclose(fp); } else { perror(save_idx); return 1; } }
		if (max_occ_f > 0.f) { fprintf(stderr, "[mm2ref] max_occ f=%g value=%d\n", max_occ_f, mm_idx_cal_max_occ(mi, max_occ_f)); mm_idx_destroy(mi); continue; }
		mm_mapopt_update(&mo, mi);
		for (si = 0; si < n_sweep; ++si) {
			int saved = -1; double t0;
			if (si + 1 < n_sweep) {            /* not the last sweep point: same work, output discarded */
				int dn; fflush(stdout); saved = dup(1); dn = open("/dev/null", O_WRONLY); dup2(dn, 1); close(dn);
			}
			mo.mini_batch_size = sweep_k[si] > 0 ? sweep_k[si] : k_default;
			t0 = wall();
			mm_write_sam_hdr(mi, rg, 0, 0, 0);
			if (nfn == 2 && !(mo.flag & MM_F_FRAG_MODE)) mm_map_file(mi, fn[1], &mo, sweep[si]);
			else mm_map_file_frag(mi, nfn - 1, &fn[1], &mo, sweep[si]);
			fflush(stdout);
			if (n_sweep > 1 || getenv("MM2REF_TIMING")) fprintf(stderr, "[mm2ref] threads=%d K=%lld index_s=%.3f map_s=%.3f\n", sweep[si], (long long)mo.mini_batch_size, t_idx, wall() - t0);
			if (saved >= 0) { dup2(saved, 1); close(saved); }
		}
		mm_idx_destroy(mi);
		t_idx0 = wall();
	}
	mm_idx_reader_close(r);
	fflush(stdout);
	return 0;
}
