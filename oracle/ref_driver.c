/* TEST INFRASTRUCTURE ONLY -- driver for the reference build "O-full" (see oracle/Makefile).
 *
 * This file is OUR code.  It links against the reference's minimap2 fork (compiled in place from
 * /root/reference, map.c with the ALSER early return removed) through its public C API
 * (minimap.h:180-348) and plays the role of the fork's main(): `-ax sr` mapping of one or two
 * FASTQ/FASTA files to SAM on stdout.  It exists so that goldens can be generated and the
 * restatement in al_oracle.c can be pinned against the real reference.
 *
 * usage: mm2ref [-t N] [-R rgline] [-K minibatch] [--seeds] [--alnseq] [--hit-only] ref.fa r1.fq [r2.fq]
 *   --seeds  : mm_dbg_flag |= MM_DBG_PRINT_SEED  (RS/SD/CN lines on stderr, map.c:333-338,381-385)
 *   --alnseq : mm_dbg_flag |= MM_DBG_PRINT_ALN_SEQ (align.c:315-338)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "minimap.h"
#include "mmpriv.h"

int main(int argc, char **argv)
{
	mm_idxopt_t io; mm_mapopt_t mo;
	int i, n_threads = 1, nfn = 0;
	const char *rg = 0, *fn[4];
	mm_idx_reader_t *r; mm_idx_t *mi;
	mm_verbose = 1;
	mm_set_opt(0, &io, &mo);
	if (mm_set_opt("sr", &io, &mo) < 0) return 2;
	mo.flag |= MM_F_OUT_SAM | MM_F_CIGAR;                 /* -a, main.c:162 */
	for (i = 1; i < argc; ++i) {
		if (!strcmp(argv[i], "-t") && i + 1 < argc) n_threads = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-R") && i + 1 < argc) rg = argv[++i];
		else if (!strcmp(argv[i], "-K") && i + 1 < argc) mo.mini_batch_size = atoi(argv[++i]);
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
	if (nfn < 2) { fprintf(stderr, "usage: mm2ref [opts] ref.fa r1.fq [r2.fq]\n"); return 2; }
	if (mm_check_opt(&io, &mo) < 0) return 2;
	r = mm_idx_reader_open(fn[0], &io, 0);
	if (r == 0) { fprintf(stderr, "mm2ref: cannot open %s\n", fn[0]); return 1; }
	while ((mi = mm_idx_reader_read(r, n_threads)) != 0) {
		mm_write_sam_hdr(mi, rg, 0, 0, 0);
		mm_mapopt_update(&mo, mi);
		if (nfn == 2 && !(mo.flag & MM_F_FRAG_MODE)) mm_map_file(mi, fn[1], &mo, n_threads);
		else mm_map_file_frag(mi, nfn - 1, &fn[1], &mo, n_threads);
		mm_idx_destroy(mi);
	}
	mm_idx_reader_close(r);
	fflush(stdout);
	return 0;
}
