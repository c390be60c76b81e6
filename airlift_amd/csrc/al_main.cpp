// airlift-align -- drop-in command line for the aligner invocations in AirLift's stage scripts.
//
// Accepted argv shapes:
//   B1/B2  airlift-align mem [-R RG] [-t N] REF.fa R_1.fastq [R_2.fastq]        (src/0-align_reads.sh:13, 0-align_singletons.sh:12)
//   B3     airlift-align aln [-n X] [-t N] REF.fa GAPS.fa > X.sai               (src/3-align_gaps/align_gaps.sh:14; writes a stub .sai)
//          airlift-align samse REF.fa X.sai GAPS.fa                              (align_gaps.sh:15; does the actual single-end mapping)
//          airlift-align index REF.fa                                             (accepted, no-op: the index is built on the GPU per run)
//   mm2    airlift-align -ax sr [-t N] [-R RG] [-K NUM] [--sam-hit-only] REF.fa R1 [R2]   (fork README usage; main.c:113-273)
//   8e     airlift-align ... --devices 0-7 [-o out.sam]      reads sharded over several GPUs of the node (index built once, copied
//                                                                   device to device; with a regular output file every lane pwrite()s its block)
//   N3     airlift-align ... --bam | --sorted-bam [-l LEVEL]       BAM on stdout; sorted = mapped records in coordinate order,
//                                                                   i.e. the result of `| samtools view -h -F4 | samtools sort -l5`
//   N2     airlift-align tokens --read-size R --skip S [-t N] REF.fa GAPS.fa      gaps_to_fasta.py GAPS.fa R tokens.fa S ; samse REF x tokens.fa
//   a8     airlift-align -ax sr --count-candidates REF.fa READS     the as-shipped fork's observable: seed-cluster count on stderr
// SAM goes to stdout; exit status 0 on success, non-zero on failure (so the caller's pipe fails).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <string>
#include <vector>
#include "../../include/airlift.h"

// numbers with k/m/g suffixes, as the fork's command line reads -K, -r, -g, -F (mm_parse_num, main.c:87-96)
static long long parse_num(const char *str, const char *opt)
{
	char *p; double x = strtod(str, &p);
	if (*p == 'G' || *p == 'g') x *= 1e9, ++p;
	else if (*p == 'M' || *p == 'm') x *= 1e6, ++p;
	else if (*p == 'K' || *p == 'k') x *= 1e3, ++p;
	if (*p) fprintf(stderr, "[WARNING] airlift-align: trailing characters in the value of %s: '%s'\n", opt, str);
	return (long long)(x + .499);
}

static int usage()
{
	fprintf(stderr, "Usage: airlift-align mem [-R RG] [-t N] ref.fa reads_1.fq [reads_2.fq]\n"
	                "       airlift-align aln [-n X] [-t N] ref.fa reads.fa > x.sai ; airlift-align samse ref.fa x.sai reads.fa\n"
	                "       airlift-align -ax sr [-t N] [-R RG] ref.fa reads_1.fq [reads_2.fq]\n");
	return 1;
}

int main(int argc, char **argv)
{
	al_idxopt_t io; al_mapopt_t mo;
	struct timespec tsm; clock_gettime(CLOCK_MONOTONIC, &tsm);
	std::vector<const char *> pos; const char *rg = nullptr; int n_threads = 3, device = -1; std::vector<int> devices; bool count_only = false; int bam_mode = 0, bam_level = 5;
	enum { MODE_MEM, MODE_ALN, MODE_SAMSE, MODE_MM2, MODE_TOKENS } mode = MODE_MM2; int tok_size = 0, tok_skip = 1;
	bool prefilter = false; int pf[4] = {3, 3, 5, 3};          // adjacency e, GreedySnake e, k-mer size, rounds
	const char *dump_fn = nullptr;
	int i = 1; bool k_given = false; int rank = -1, world = 0; const char *rendezvous = nullptr, *out_path = nullptr;
	if (argc < 2) return usage();
	al_set_opt(0, &io, &mo);
	al_set_opt("sr", &io, &mo);
	mo.flag |= AL_F_OUT_SAM | AL_F_CIGAR;
	if (!strcmp(argv[1], "extract-reads")) {       // N1: airlift-align extract-reads [--noprune] READS.bam REGIONS.bed READSIZE > rows   (extract_reads.sh BINDIR BAM BED READSIZE)
		int prune = 1; std::vector<const char *> p;
		for (int j = 2; j < argc; ++j) { if (!strcmp(argv[j], "--noprune")) prune = 0; else p.push_back(argv[j]); }
		if (p.size() < 2 || (prune && p.size() < 3)) { fprintf(stderr, "Usage: airlift-align extract-reads [--noprune] reads.bam regions.bed [readsize]\n"); return 1; }
		const int64_t n = al_extract_reads(p[0], p[1], p.size() > 2 ? atoi(p[2]) : 0, prune, stdout);
		return n < 0 || fflush(stdout) == EOF ? 1 : 0;
	}
	if (!strcmp(argv[1], "extract-sequence")) {    // N1: airlift-align extract-sequence FQ1 FQ2 ROWS OUTDIR   (extract_sequence.sh BINDIR FQ1 FQ2 BED THREAD OUTPUT)
		if (argc < 6) { fprintf(stderr, "Usage: airlift-align extract-sequence reads_1.fq reads_2.fq rows.bed outdir\n"); return 1; }
		int64_t np = 0, ns = 0;
		const int rc = al_extract_sequence(argv[2], argv[3], argv[4], argv[5], &np, &ns);
		if (rc == 0) fprintf(stderr, "[airlift] extract-sequence: %lld pairs, %lld singletons\n", (long long)np, (long long)ns);
		return rc == 0 ? 0 : 1;
	}
	if (!strcmp(argv[1], "remap")) {
		// N1 fused with the re-alignment: airlift-align remap [-t N] [-R RG] [--noprune] [--readsize R] -o PAIRS.sam [--singletons SINGLE.sam] REF.fa READS.bam REGIONS.bed FQ1 FQ2
		// = extract_reads.sh + extract_sequence.sh + 0-align_reads.sh + 0-align_singletons.sh (run_pipeline.sh:58,103-113) without the rows file, the
		// three FASTQ files and the tool processes in between: the selected reads stay in memory files that the stream driver hands to the GPU.
		int prune = 1, read_size = 0, n_threads = 3; const char *rg = nullptr, *out_p = nullptr, *out_s = nullptr; std::vector<const char *> p;
		for (int j = 2; j < argc; ++j) {
			if (!strcmp(argv[j], "--noprune")) prune = 0;
			else if (!strcmp(argv[j], "--readsize") && j + 1 < argc) read_size = atoi(argv[++j]);
			else if (!strcmp(argv[j], "-t") && j + 1 < argc) n_threads = atoi(argv[++j]);
			else if (!strcmp(argv[j], "-R") && j + 1 < argc) rg = argv[++j];
			else if (!strcmp(argv[j], "-o") && j + 1 < argc) out_p = argv[++j];
			else if (!strcmp(argv[j], "--singletons") && j + 1 < argc) out_s = argv[++j];
			else p.push_back(argv[j]);
		}
		if (p.size() != 5 || !out_p || (prune && read_size <= 0)) { fprintf(stderr, "Usage: airlift-align remap [-t N] [-R RG] [--noprune | --readsize R] -o pairs.sam [--singletons single.sam] ref.fa reads.bam regions.bed reads_1.fq reads_2.fq\n"); return 1; }
		if (!getenv("AL_PG_PLAIN")) al_set_program_line(AL_MM_VERSION, argc, argv);
		if (!k_given) setenv("AL_AUTO_BATCH", "1", 0);
		setenv("GPU_MAX_HW_QUEUES", "16", 0);
		int fds[3]; int64_t np = 0, ns = 0;
		if (al_extract_to_memory(p[1], p[2], read_size, prune, p[3], p[4], fds, &np, &ns) != 0) return 1;
		fprintf(stderr, "[airlift] remap: %lld pairs, %lld singletons selected\n", (long long)np, (long long)ns);
		if (al_check_opt(&io, &mo) < 0) return 1;
		al_idx_t *mi = al_idx_build_device(p[0], &io, -1);
		if (!mi) return 1;
		int rc = 0;
		const std::string f1 = "/proc/self/fd/" + std::to_string(fds[0]), f2 = "/proc/self/fd/" + std::to_string(fds[1]), f3 = "/proc/self/fd/" + std::to_string(fds[2]);
		{ FILE *o = fopen(out_p, "wb"); const char *fn[2] = {f1.c_str(), f2.c_str()};
		  if (!o) { perror(out_p); rc = 1; } else { if (al_map_file_frag(mi, 2, fn, &mo, n_threads, o, rg, -1) != 0) rc = 1; if (fclose(o) == EOF) rc = 1; } }
		if (out_s && rc == 0) { FILE *o = fopen(out_s, "wb"); const char *fn[1] = {f3.c_str()};
		  if (!o) { perror(out_s); rc = 1; } else { if (al_map_file_frag(mi, 1, fn, &mo, n_threads, o, rg, -1) != 0) rc = 1; if (fclose(o) == EOF) rc = 1; } }
		for (int j = 0; j < 3; ++j) if (fds[j] >= 0) close(fds[j]);
		al_idx_destroy(mi);
		return rc;
	}
	// @PG as main.c:369 writes it (VN = the fork's MM_VERSION, main.c:16: the version whose records this path reproduces;
	// CL = this process's argv).  AL_PG_PLAIN=1 leaves the bare line (the tests' goldens come from a driver without argv).
	if (!getenv("AL_PG_PLAIN")) al_set_program_line(AL_MM_VERSION, argc, argv);
	if (!strcmp(argv[1], "mem")) mode = MODE_MEM, i = 2;
	else if (!strcmp(argv[1], "aln")) mode = MODE_ALN, i = 2;
	else if (!strcmp(argv[1], "samse")) mode = MODE_SAMSE, i = 2;
	else if (!strcmp(argv[1], "tokens")) mode = MODE_TOKENS, i = 2;
	else if (!strcmp(argv[1], "index")) return 0;        // `bwa index REF`: nothing to precompute, the minimizer index is built on the GPU at start-up
	for (; i < argc; ++i) {
		const char *a = argv[i];
		if (a[0] != '-' || !strcmp(a, "-")) { pos.push_back(a); continue; }
		if (!strcmp(a, "-R") && i + 1 < argc) rg = argv[++i];
		else if (!strcmp(a, "-t") && i + 1 < argc) n_threads = atoi(argv[++i]);
		else if (!strcmp(a, "-x") && i + 1 < argc) { if (al_set_opt(argv[++i], &io, &mo) < 0) { fprintf(stderr, "[ERROR] unknown preset '%s'\n", argv[i]); return 1; } }
		else if (!strcmp(a, "-ax") && i + 1 < argc) { if (al_set_opt(argv[++i], &io, &mo) < 0) { fprintf(stderr, "[ERROR] unknown preset '%s'\n", argv[i]); return 1; } }
		else if (!strcmp(a, "-a")) mo.flag |= AL_F_OUT_SAM | AL_F_CIGAR;
		else if (!strcmp(a, "-d") && i + 1 < argc) dump_fn = argv[++i];                       // main.c:150: dump the index to FILE (mm_idx_dump's format)
		else if (!strcmp(a, "-k") && i + 1 < argc) io.k = atoi(argv[++i]);
		else if (!strcmp(a, "-w") && i + 1 < argc) io.w = atoi(argv[++i]);
		else if (!strcmp(a, "-K") && i + 1 < argc) { mo.mini_batch_size = (int)parse_num(argv[++i], "-K"); k_given = true; }
		else if (!strcmp(a, "-n") && i + 1 < argc) { if (mode == MODE_ALN) ++i; else mo.min_cnt = atoi(argv[++i]); }   // bwa aln -n X: accepted, no analogue; minimap2 -n: main.c:165
		// numeric options of the fork's command line (main.c:144-230), same letters and meaning
		else if (!strcmp(a, "-g") && i + 1 < argc) mo.max_gap = (int)parse_num(argv[++i], "-g");
		else if (!strcmp(a, "-F") && i + 1 < argc) mo.max_frag_len = (int)parse_num(argv[++i], "-F");
		else if (!strcmp(a, "-r") && i + 1 < argc) mo.bw = (int)parse_num(argv[++i], "-r");
		else if (!strcmp(a, "-N") && i + 1 < argc) mo.best_n = atoi(argv[++i]);
		else if (!strcmp(a, "-p") && i + 1 < argc) mo.pri_ratio = (float)atof(argv[++i]);
		else if (!strcmp(a, "-M") && i + 1 < argc) mo.mask_level = (float)atof(argv[++i]);
		else if (!strcmp(a, "-m") && i + 1 < argc) mo.min_chain_score = atoi(argv[++i]);
		else if (!strcmp(a, "-A") && i + 1 < argc) mo.a = atoi(argv[++i]);
		else if (!strcmp(a, "-B") && i + 1 < argc) mo.b = atoi(argv[++i]);
		else if (!strcmp(a, "-s") && i + 1 < argc) mo.min_dp_max = atoi(argv[++i]);
		else if (!strcmp(a, "-O") && i + 1 < argc) { char *e; mo.q = mo.q2 = (int)strtol(argv[++i], &e, 10); if (*e == ',') mo.q2 = (int)strtol(e + 1, &e, 10); }
		else if (!strcmp(a, "-E") && i + 1 < argc) { char *e; mo.e = mo.e2 = (int)strtol(argv[++i], &e, 10); if (*e == ',') mo.e2 = (int)strtol(e + 1, &e, 10); }
		else if (!strcmp(a, "-z") && i + 1 < argc) { char *e; mo.zdrop = mo.zdrop_inv = (int)strtol(argv[++i], &e, 10); if (*e == ',') mo.zdrop_inv = (int)strtol(e + 1, &e, 10); }
		else if (!strcmp(a, "--end-bonus") && i + 1 < argc) mo.end_bonus = atoi(argv[++i]);
		else if (!strcmp(a, "--max-chain-skip") && i + 1 < argc) mo.max_chain_skip = atoi(argv[++i]);
		else if (!strcmp(a, "--score-N") && i + 1 < argc) mo.sc_ambi = atoi(argv[++i]);
		else if (!strcmp(a, "--seed") && i + 1 < argc) mo.seed = atoi(argv[++i]);
		else if (!strcmp(a, "--sam-hit-only")) mo.flag |= AL_F_SAM_HIT_ONLY;
		else if (!strcmp(a, "--count-candidates")) count_only = true;
		else if (!strcmp(a, "--prefilter") && i + 1 < argc) {   // N4: --prefilter ADJ_E[,SNAKE_E[,SNAKE_K[,SNAKE_ITER]]] with --count-candidates
			prefilter = true; char *e; const char *v = argv[++i];
			pf[0] = (int)strtol(v, &e, 10); if (*e == ',') { pf[1] = (int)strtol(e + 1, &e, 10); if (*e == ',') { pf[2] = (int)strtol(e + 1, &e, 10); if (*e == ',') pf[3] = (int)strtol(e + 1, &e, 10); } } else pf[1] = pf[0];
		}
		else if (!strcmp(a, "--read-size") && i + 1 < argc) tok_size = atoi(argv[++i]);
		else if (!strcmp(a, "--skip") && i + 1 < argc) tok_skip = atoi(argv[++i]);
		else if (!strcmp(a, "--bam")) bam_mode = 1;
		else if (!strcmp(a, "--sorted-bam")) bam_mode = 2;
		else if (!strcmp(a, "-l") && i + 1 < argc) bam_level = atoi(argv[++i]);
		else if (!strcmp(a, "--sort-mem") && i + 1 < argc) setenv("AL_SORT_MEM", std::to_string(parse_num(argv[++i], "--sort-mem")).c_str(), 1);   // --sorted-bam: bytes held before a sorted run is spilled (samtools sort -m)
		else if (!strcmp(a, "--device") && i + 1 < argc) device = atoi(argv[++i]);
		else if (!strcmp(a, "--rank") && i + 1 < argc) rank = atoi(argv[++i]);             // one process per GPU: --rank r --world R -o OUT (al_map_file_frag_ranked)
		else if (!strcmp(a, "--world") && i + 1 < argc) world = atoi(argv[++i]);
		else if (!strcmp(a, "--ranked")) {                                                 // ... with rank and world from the launcher's environment (torchrun: RANK, WORLD_SIZE; the GPU is LOCAL_RANK)
			if (!getenv("RANK") || !getenv("WORLD_SIZE")) { fprintf(stderr, "[ERROR] --ranked needs RANK and WORLD_SIZE in the environment\n"); return 1; }
			rank = atoi(getenv("RANK")); world = atoi(getenv("WORLD_SIZE"));
		}
		else if (!strcmp(a, "--rendezvous") && i + 1 < argc) rendezvous = argv[++i];
		else if (!strcmp(a, "--devices") && i + 1 < argc) {   // "0-7", "0,1,2", "0,0" (two lanes on one GPU): reads of every mini-batch sharded over the lanes
			const char *p = argv[++i];
			while (*p) {
				char *e; const long v0 = strtol(p, &e, 10); long v1 = v0;
				if (e == p) { fprintf(stderr, "[ERROR] --devices expects a list like 0-7 or 0,1,2\n"); return 1; }
				if (*e == '-') { p = e + 1; v1 = strtol(p, &e, 10); }
				for (long v = v0; v <= v1 && devices.size() < 64; ++v) devices.push_back((int)v);
				p = *e == ',' ? e + 1 : e;
				if (*e && *e != ',') { fprintf(stderr, "[ERROR] --devices expects a list like 0-7 or 0,1,2\n"); return 1; }
			}
		}
		else if (!strcmp(a, "-o") && i + 1 < argc) out_path = argv[++i];                   // opened below, once it is known whether this is one of several processes (which open the merged file themselves)
		else if (!strcmp(a, "--version")) { puts(al_version()); return 0; }
		else { fprintf(stderr, "[WARNING] airlift-align: option '%s' ignored\n", a); }
	}
	// one process per GPU?  (--world, or --rank / --ranked with the launcher's WORLD_SIZE.)  Anything else -- also WORLD_SIZE = 1, or WORLD_SIZE set
	// without --rank / --ranked -- is a single process and writes -o FILE itself (main.c:183-190).
	if (world <= 0 && rank >= 0 && getenv("WORLD_SIZE")) world = atoi(getenv("WORLD_SIZE"));
	if (world <= 1 && out_path && strcmp(out_path, "-") != 0 && !freopen(out_path, "wb", stdout)) { fprintf(stderr, "[ERROR] failed to write the output to file '%s'\n", out_path); return 1; }
	// -K given: an upper bound of the bases per device batch.  Not given: the stream driver (plain FASTQ in, SAM out) sizes its batches
	// from the free device memory (al_stream_pipe.cpp); the host driver keeps the preset's 50 Mbases, as the reference.
	if (!k_given) setenv("AL_AUTO_BATCH", "1", 0);
	setenv("GPU_MAX_HW_QUEUES", "16", 0);       // (HIP reads it when the runtime starts) the contexts' and slots' streams on separate hardware queues: the default of 4 makes streams that share a queue wait for each other
	if (al_check_opt(&io, &mo) < 0) return 1;
	const char *ref = nullptr; std::vector<const char *> reads;
	if (mode == MODE_ALN) {          // the real work happens in samse; emit a small marker so `> x.sai` is non-empty
		if (pos.size() < 2) return usage();
		fputs("AIRLIFT-SAI-STUB\n", stdout);
		return 0;
	} else if (mode == MODE_SAMSE) {
		if (pos.size() < 3) return usage();
		ref = pos[0]; reads.push_back(pos[2]);
	} else {
		if ((pos.size() < 2 && !(dump_fn && pos.size() == 1)) || pos.size() > 3) return usage();   // (-d FILE ref.fa: index only, main.c:374-376)
		ref = pos[0]; for (size_t j = 1; j < pos.size(); ++j) reads.push_back(pos[j]);
	}
	struct timespec ts0, ts1; clock_gettime(CLOCK_MONOTONIC, &ts0);
	if (!devices.empty()) device = devices[0];
	// plain FASTQ files in, SAM out, one GPU: the run's device memory is obtained by a background thread while the reference is loaded and indexed
	const bool ref_is_idx = al_idx_is_idx(ref) > 0;                              // index.c:585-600: a prebuilt index (the fork's -d file, or this program's) instead of a FASTA
	if (devices.size() <= 1 && world <= 1 && !bam_mode && !count_only && mode != MODE_TOKENS && !getenv("AL_HOST_INDEX") && !getenv("AL_HOST_IO") && !ref_is_idx && !reads.empty()) {
		const int64_t rb = al_device_reserve_for_run(device, ref, (int)reads.size(), reads.data());
		if (rb > 0 && getenv("AL_TIMING")) fprintf(stderr, "[airlift] device memory reserve of %.1f GB started\n", rb / 1e9);
	}
	al_idx_t *mi = ref_is_idx ? al_idx_load(ref) : getenv("AL_HOST_INDEX") ? al_idx_build(ref, &io, n_threads) : al_idx_build_device(ref, &io, device);
	if (mi && ref_is_idx && (al_idx_k(mi) != io.k || al_idx_w(mi) != io.w)) fprintf(stderr, "[WARNING]\033[1;31m Indexing parameters (-k, -w or -H) overridden by parameters used in the prebuilt index.\033[0m\n");   // main.c:378-380
	clock_gettime(CLOCK_MONOTONIC, &ts1);
	if (getenv("AL_TIMING")) fprintf(stderr, "[airlift] index build %.3f s\n", (ts1.tv_sec - ts0.tv_sec) + 1e-9 * (ts1.tv_nsec - ts0.tv_nsec));
	if (!mi) { fprintf(stderr, "[ERROR] failed to open file '%s'\n", ref); return 1; }
	if (dump_fn && !ref_is_idx && al_idx_dump(dump_fn, mi) != 0) { al_idx_destroy(mi); return 1; }
	if (reads.empty()) { al_idx_destroy(mi); fflush(stderr); _exit(0); }       // (index only)
	if (mode == MODE_TOKENS) {   // gaps_to_fasta.py + single-end alignment of the tokens in one step (tokens are cut on the GPU)
		const int rc2 = al_map_tokens_file(mi, reads[0], tok_size, tok_skip, &mo, n_threads, stdout, rg, device);
		al_idx_destroy(mi);
		if (fflush(stdout) == EOF) return 1;
		fflush(stderr);
		_exit(rc2 == 0 ? 0 : 1);
	}
	if (count_only) {   // what the as-shipped fork prints instead of alignments (main.c:384-391, 417)
		int64_t total = 0, o4[4] = {0, 0, 0, 0};
		const int rc2 = prefilter ? al_count_candidates_file_filtered(mi, reads[0], &mo, n_threads, device, pf[0], pf[1], pf[2] > 0 ? pf[2] : 5, pf[3], o4)
		                          : al_count_candidates_file(mi, reads[0], &mo, n_threads, device, &total);
		al_idx_destroy(mi);
		if (rc2 != 0) return 1;
		if (prefilter) total = o4[0];
		fprintf(stderr, "\nTotal No. of Mappings before alignment (verification): %d\n", (int)total);
		if (prefilter) fprintf(stderr, "Candidates kept by the adjacency filter (at most %d absent seeds): %lld\nCandidates kept by GreedySnake (e = %d, k-mer %d, %d rounds): %lld\nCandidates kept by both: %lld\n", pf[0], (long long)o4[1], pf[1], pf[2], pf[3], (long long)o4[2], (long long)o4[3]);
		fflush(stderr);
		_exit(0);
	}
	if (world > 1) {   // one process per GPU
		if (rank < 0 && getenv("RANK")) rank = atoi(getenv("RANK"));
		if (rank < 0 || rank >= world || !out_path || bam_mode == 2) { fprintf(stderr, "[ERROR] a multi-process run needs --rank r (or RANK) below --world, -o FILE after --world, and SAM or unsorted BAM output\n"); return 1; }
		const int rc2 = bam_mode ? al_map_file_frag_ranked_bam(mi, (int)reads.size(), reads.data(), &mo, n_threads, out_path, rg, device, rank, world, rendezvous, 0.0, bam_level)
		                         : al_map_file_frag_ranked(mi, (int)reads.size(), reads.data(), &mo, n_threads, out_path, rg, device, rank, world, rendezvous, 0.0);
		al_idx_destroy(mi);
		fflush(stderr);
		_exit(rc2 == 0 ? 0 : 1);
	}
	int rc = devices.size() > 1 ? al_map_file_frag_multi(mi, (int)reads.size(), reads.data(), &mo, n_threads, stdout, rg, devices.data(), (int)devices.size(), bam_mode, bam_level)
	       : bam_mode ? al_map_file_frag_bam(mi, (int)reads.size(), reads.data(), &mo, n_threads, stdout, rg, device, bam_mode == 2, bam_level)
	                  : al_map_file_frag(mi, (int)reads.size(), reads.data(), &mo, n_threads, stdout, rg, device);
	clock_gettime(CLOCK_MONOTONIC, &ts0);
	al_idx_destroy(mi);
	clock_gettime(CLOCK_MONOTONIC, &ts1);
	if (getenv("AL_TIMING")) fprintf(stderr, "[airlift] index release %.3f s\n", (ts1.tv_sec - ts0.tv_sec) + 1e-9 * (ts1.tv_nsec - ts0.tv_nsec));
	if (fflush(stdout) == EOF) { perror("[ERROR] failed to write the results"); return 1; }
	clock_gettime(CLOCK_MONOTONIC, &ts1);
	if (getenv("AL_TIMING")) { fprintf(stderr, "[airlift] main() %.3f s\n", (ts1.tv_sec - tsm.tv_sec) + 1e-9 * (ts1.tv_nsec - tsm.tv_nsec)); al_device_reserve_report(stderr); }
	// results are flushed and every device object is released: skip the HIP runtime's static teardown (0.3 s)
	fflush(stderr);
	if (getenv("AL_NO_FAST_EXIT")) return rc == 0 ? 0 : 1;          // (profilers flush their traces from exit handlers)
	_exit(rc == 0 ? 0 : 1);
}
