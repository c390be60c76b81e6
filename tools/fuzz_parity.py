#!/usr/bin/env python3
"""Randomised parity sweep: airlift-align (HIP path) against the reference build oracle/_ref/mm2ref on small synthetic
workloads of assorted shapes (read length, single/paired, error and indel rates, N content, repeats, insert size, options).
Usage: tools/fuzz_parity.py [n_cases] [seed]   -> prints one line per case, exits 1 on the first difference."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_synth as g

CLI = os.path.join(ROOT, "airlift_amd", "bin", "airlift-align")
REF = os.path.join(ROOT, "oracle", "_ref", "mm2ref")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
only = os.environ.get('FUZZ_OPTS')
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
OPTS = [[], [], [], ["-r", "40"], ["-g", "60", "-F", "500"], ["-N", "3"], ["-n", "3", "-m", "30"], ["-z", "30,20"], ["--end-bonus", "0"], ["-s", "20"], ["-k", "15", "-w", "5"],
        ["-A", "1", "-B", "2", "-O", "2,12", "-E", "2,1"], ["-z", "33"], ["-z", "34,30"], ["-z", "60"], ["--end-bonus", "30"], ["-B", "12"], ["-B", "3"], ["-O", "5,30", "-E", "3,1"],
        ["-A", "3", "-B", "5", "-O", "9,20", "-E", "3,2", "-z", "45"], ["-r", "10"], ["-r", "200"], ["-p", "0.2", "-N", "50"], ["-M", "0.9"], ["--max-chain-skip", "1"], ["-n", "1", "-m", "10", "-s", "10"],
        ["-F", "300"], ["-g", "20"], ["--score-N", "0"], ["--score-N", "5"]]
bad = 0
for case in range(n_cases):
    d = tempfile.mkdtemp(prefix="al_fuzz_")
    rl = int(rng.choice([30, 50, 76, 100, 125, 150, 151, 200, 250, 300, 400, 500]))
    se = bool(rng.random() < 0.3)
    n_ctg = int(rng.integers(1, 6)); tot = int(rng.integers(60_000, 600_000)) * int(os.environ.get('FUZZ_REFSCALE', '1'))
    ref = g.make_reference(seed=int(rng.integers(1 << 30)), n_contigs=n_ctg, total_len=tot, n_dups=int(rng.integers(0, 60)), dup_len=(200, 3000),
                           dup_div=float(rng.choice([0.0, 0.01, 0.05])), n_frac=float(rng.choice([0, 0, 0.01])), tandem=int(rng.integers(0, 8)))
    if min(len(c) for _, c in ref) < 2 * rl + 1100:
        continue
    if rng.random() < float(os.environ.get("FUZZ_REPEAT", "0.15")):   # a high-copy element: minimizers above mid_occ, re-chain pass, many chains per read
        ulen = int(rng.integers(80, 400)); unit = rng.integers(0, 4, size=ulen, dtype=np.uint8)
        copies = int(rng.integers(300, 1600)); div = float(rng.choice([0.0, 0.01, 0.04]))
        for _, c in ref:
            for _ in range(copies // len(ref) + 1):
                if len(c) <= ulen + 2:
                    break
                p0 = int(rng.integers(0, len(c) - ulen)); u = unit.copy(); msk = rng.random(ulen) < div
                u[msk] = (u[msk] + rng.integers(1, 4, size=int(msk.sum()), dtype=np.uint8)) & 3
                c[p0:p0 + ulen] = u
    ins = int(rng.choice([rl + 5, int(1.3 * rl), 2 * rl + 100, 3 * rl]))
    n = int(rng.integers(500, 4000)) * int(os.environ.get('FUZZ_SCALE', '1'))
    r1, r2 = g.simulate_pairs(ref, n, rl, seed=int(rng.integers(1 << 30)), ins_mean=ins, ins_sd=max(1, ins // 10), ins_lo=rl, ins_hi=max(1000, 4 * rl),
                              sub_rate=float(rng.choice([0, 0.002, 0.01, 0.03])), del_frac=float(rng.choice([0, 0.05, 0.3])), del_len=int(rng.integers(1, 12)),
                              ins_frac=float(rng.choice([0, 0.05, 0.3])), n_rate=float(rng.choice([0, 0, 0.002])), single_end=se)
    g.write_fasta(os.path.join(d, "ref.fa"), ref)
    g.write_fastq(os.path.join(d, "r_1.fq"), r1, "realigned_")
    files = ["r_1.fq"]
    if not se:
        g.write_fastq(os.path.join(d, "r_2.fq"), r2, "realigned_"); files.append("r_2.fq")
    opts = list(OPTS[int(rng.integers(len(OPTS)))])
    if os.environ.get('FUZZ_ONLY') and case != int(os.environ['FUZZ_ONLY']):
        subprocess.run(['rm', '-rf', d]); continue
    if only is not None:
        opts = only.split()
    e = subprocess.run([REF, "-t", "8"] + opts + ["ref.fa"] + files, cwd=d, capture_output=True)
    o = subprocess.run([CLI, "-ax", "sr"] + opts + ["ref.fa"] + files, cwd=d, capture_output=True, env=dict(os.environ, AL_PG_PLAIN="1"))
    if os.environ.get('FUZZ_DBGS'):
        for dbg in os.environ['FUZZ_DBGS'].split():
            o2 = subprocess.run([CLI, "-ax", "sr"] + opts + ["ref.fa"] + files, cwd=d, capture_output=True, env=dict(os.environ, AL_DBG=dbg))
            print("   AL_DBG=%s rc=%d same=%s %s" % (dbg, o2.returncode, o2.stdout == e.stdout, o2.stderr.decode()[-900:].replace("\n", " | ")))
    if o.returncode != 0 and b"not supported" in o.stderr:        # a loud refusal (window beyond the kernels' limit) is not a parity failure
        print("skip case %d: %s" % (case, o.stderr.decode().strip()[-120:])); subprocess.run(["rm", "-rf", d]); continue
    same = e.returncode == 0 and o.returncode == 0 and e.stdout == o.stdout
    desc = "case %d: L=%d %s n=%d ins=%d ref=%dx%d opts=%s" % (case, rl, "SE" if se else "PE", n, ins, n_ctg, tot // n_ctg, " ".join(opts))
    if same:
        print("ok   " + desc + (" kept in " + d if os.environ.get('FUZZ_KEEP') else ""))
        if not os.environ.get('FUZZ_KEEP'): subprocess.run(["rm", "-rf", d])
    else:
        bad += 1
        el, ol = e.stdout.split(b"\n"), o.stdout.split(b"\n")
        nd = sum(1 for a, b in zip(el, ol) if a != b)
        for a, b in [(a, b) for a, b in zip(el, ol) if a != b][:2]:
            fa, fb = a.split(b"\t"), b.split(b"\t")
            print("   exp " + b"\t".join(fa[:9] + fa[11:]).decode()); print("   got " + b"\t".join(fb[:9] + fb[11:]).decode())
        print("DIFF " + desc + "  (%d lines differ, rc %d/%d) kept in %s : %s" % (nd, e.returncode, o.returncode, d, o.stderr.decode()[-700:].replace("\n", " | ")))
        if bad >= 3:
            break
sys.exit(1 if bad else 0)
