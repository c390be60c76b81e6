"""-m gpu test of `airlift-align remap` (SURVEY.md N1 fused with the re-alignment): BAM + BED + the two FASTQ files in, the SAM of the
re-aligned pairs and of the singletons out, with nothing in between on disk.  It must equal the step-by-step flow of run_pipeline.sh:58,
103-113 done with this repository's own commands -- extract-reads, extract-sequence (files), then the aligner on those files -- byte
for byte.  (The extraction rules themselves are checked on the CPU against oracle/n1_oracle.py, tests/test_extract_cpu.py.)"""
import json
import os
import subprocess

import pytest

from test_extract_cpu import write_bam

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "airlift_amd", "bin", "airlift-align")


def _cigar(s):
    out, n = [], ""
    for ch in s:
        if ch.isdigit():
            n += ch
        else:
            out.append((int(n), ch)); n = ""
    return out


def test_remap_equals_the_step_by_step_flow(golden_unpacked, tmp_path):
    d = golden_unpacked["g1_mt150pe"]
    m = json.load(open(os.path.join(d, "meta.json")))
    # a BAM of the golden alignments (what AirLift holds for the old reference), primary records only
    refs, recs = [], []
    for line in open(os.path.join(d, m["sam"]) if "sam" in m else os.path.join(d, "expected.sam")):
        f = line.rstrip("\n").split("\t")
        if line.startswith("@SQ"):
            refs.append((f[1][3:], int(f[2][3:])))
        if line.startswith("@") or int(f[1]) & 0x900 or f[2] == "*":
            continue
        recs.append(([r[0] for r in refs].index(f[2]), int(f[3]) - 1, int(f[4]), int(f[1]), _cigar(f[5]) if f[5] != "*" else [], f[0], len(f[9])))
    assert len(recs) > 1000
    recs.sort(key=lambda r: (r[0], r[1]))
    bam = str(tmp_path / "old.bam"); write_bam(bam, refs, recs)
    bed = str(tmp_path / "regions.bed")
    L = refs[0][1]
    open(bed, "w").write("".join("%s\t%d\t%d\n" % (refs[0][0], b, e) for b, e in ((200, 3000), (2500, 6000), (9000, L - 100))))
    fq = [os.path.join(d, r) for r in m["reads"]]
    env = dict(os.environ, AL_PG_PLAIN="1", AL_TIMING="1")
    rg = ["-R", "@RG\\tID:x\\tSM:y"]
    # step by step: rows -> three FASTQ files -> two aligner runs
    rows = subprocess.run([CLI, "extract-reads", "--noprune", bam, bed], capture_output=True, check=True).stdout
    assert rows.count(b"\n") > 500
    open(tmp_path / "rows.bed", "wb").write(rows)
    subprocess.run([CLI, "extract-sequence", fq[0], fq[1], str(tmp_path / "rows.bed"), str(tmp_path)], capture_output=True, check=True)
    ref = os.path.join(d, m["ref"])
    exp_p = subprocess.run([CLI, "-ax", "sr"] + rg + [ref, str(tmp_path / "reads_1.fastq"), str(tmp_path / "reads_2.fastq")], capture_output=True, check=True, env=env).stdout
    exp_s = subprocess.run([CLI, "-ax", "sr"] + rg + [ref, str(tmp_path / "singletons.fastq")], capture_output=True, check=True, env=env).stdout
    # fused
    r = subprocess.run([CLI, "remap", "--noprune"] + rg + ["-o", str(tmp_path / "p.sam"), "--singletons", str(tmp_path / "s.sam"), ref, bam, bed, fq[0], fq[1]], capture_output=True, env=env)
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    assert open(tmp_path / "p.sam", "rb").read() == exp_p and exp_p.count(b"\n") > 500
    assert open(tmp_path / "s.sam", "rb").read() == exp_s
    assert b"pairs," in r.stderr and r.stderr.count(b"stream pipeline: 1 lane") == 2      # both memory files went through the stream driver
    # pruned selection (MAPQ <= 10 or CIGAR != 150M) goes through the same path
    r = subprocess.run([CLI, "remap", "--readsize", "150", "-o", str(tmp_path / "p2.sam"), ref, bam, bed, fq[0], fq[1]], capture_output=True, env=env)
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    rows2 = subprocess.run([CLI, "extract-reads", bam, bed, "150"], capture_output=True, check=True).stdout
    open(tmp_path / "rows2.bed", "wb").write(rows2)
    os.makedirs(tmp_path / "o2"); subprocess.run([CLI, "extract-sequence", fq[0], fq[1], str(tmp_path / "rows2.bed"), str(tmp_path / "o2")], capture_output=True, check=True)
    exp2 = subprocess.run([CLI, "-ax", "sr", ref, str(tmp_path / "o2" / "reads_1.fastq"), str(tmp_path / "o2" / "reads_2.fastq")], capture_output=True, check=True, env=env).stdout
    assert open(tmp_path / "p2.sam", "rb").read() == exp2
