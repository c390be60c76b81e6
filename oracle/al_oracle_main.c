/* TEST INFRASTRUCTURE ONLY -- CLI around al_oracle.c (see al_oracle.h).
 * usage: al_oracle [-t N] [-R rgline] [--seeds] [--alnseq] [--hit-only] [--count] [--stats] [--no-sam] ref.fa r1.fq [r2.fq] */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include "al_oracle.h"

int main(int argc, char **argv)
{
	oopt_t opt; oidx_t *mi; const char *fn[4], *rg = 0; int i, nfn = 0, nt = 1, count = 0, stats = 0, nosam = 0; char rg_id[256]; ostat_t st;
	oopt_sr(&opt);
	memset(&st, 0, sizeof(st));
	for (i = 1; i < argc; ++i) {
		if (!strcmp(argv[i], "-t") && i + 1 < argc) nt = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-R") && i + 1 < argc) rg = argv[++i];
		else if (!strcmp(argv[i], "--seeds")) opt.dbg_seeds = 1;
		else if (!strcmp(argv[i], "--alnseq")) opt.dbg_aln = 1;
		else if (!strcmp(argv[i], "--hit-only")) opt.sam_hit_only = 1;
		else if (!strcmp(argv[i], "--count")) count = 1;
		else if (!strcmp(argv[i], "--stats")) stats = 1;
		else if (!strcmp(argv[i], "--no-sam")) nosam = 1;
		else if (nfn < 4) fn[nfn++] = argv[i];
	}
	if (nfn < 2) { fprintf(stderr, "usage: al_oracle [opts] ref.fa r1.fq [r2.fq]\n"); return 2; }
	mi = oidx_build_file(fn[0], opt.k, opt.w);
	if (!mi) { fprintf(stderr, "al_oracle: cannot open %s\n", fn[0]); return 1; }
	if (count) {  /* a8: the as-shipped observable (main.c:384-391,417): sum over reads of the last file */
		gzFile fp = gzopen(fn[nfn-1], "r"); char *line = (char*)malloc(1<<20); long tot = 0; int ln = 0, fq = -1;
		while (gzgets(fp, line, 1<<20)) {
			int l = strlen(line); while (l && (line[l-1] == '\n' || line[l-1] == '\r')) line[--l] = 0;
			if (fq < 0) fq = line[0] == '@';
			if (fq) { if (ln % 4 == 1) tot += o_alser_count(mi, &opt, l, line); }
			else if (line[0] != '>') tot += o_alser_count(mi, &opt, l, line);   /* single-line FASTA only */
			++ln;
		}
		gzclose(fp); free(line);
		printf("%ld\n", tot);
		return 0;
	}
	if (!nosam) o_write_sam_hdr(stdout, mi, rg, rg_id); else rg_id[0] = 0;
	o_map_files(mi, &opt, fn[1], nfn > 2? fn[2] : 0, nosam? 0 : stdout, rg_id, nt, &st);
	if (stats) fprintf(stderr, "STAT\treads=%lu\tmini=%lu\tanchors=%lu\tregs=%lu\trefbases=%lu\tcigar=%lu\tksw=%lu\n",
		(unsigned long)st.n_reads, (unsigned long)st.n_mini, (unsigned long)st.n_anchor, (unsigned long)st.n_regs_aln, (unsigned long)st.n_refbases, (unsigned long)st.n_cigar, (unsigned long)st.n_ksw);
	oidx_destroy(mi);
	return 0;
}
