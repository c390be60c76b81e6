#!/usr/bin/env python3
"""Instruction mix of the extension-DP kernels' row loop, from the compiler's assembly, and the issue ceiling that follows from it.

bench.py's `roofline.valu` rows used to price every wave-instruction at 4 cycles per SIMD; profiles/r05_valu_issue.txt (tools/micro/valu_issue.hip)
measures 2.3 - 2.5 cycles for plain 32-bit VALU instructions, 4.2 - 4.6 for v_pk_*_{i,u}16 and for DPP-modified moves.  This tool compiles
al_kernels_align.hip to assembly, takes for every k_ext_dp<...> kernel the loop that holds most of its packed instructions (the row loop: every
superblock step is unrolled inside it), counts its VALU instructions by kind and writes profiles/<tag>_valu_mix.json:
  cycles_per_valu_inst = sum(count_kind * cycles_kind) / sum(count_kind),  ceiling = 1024 SIMDs * 2.4 GHz / cycles_per_valu_inst.
usage: tools/valu_mix.py [tag]   (no GPU needed)"""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "airlift_amd", "csrc")
# cycles per wave-instruction per SIMD with 2 - 4 wavefronts per SIMD resident (profiles/r05_valu_issue.txt: rows "waves/SIMD 2" and "4" averaged)
CYC = {"plain": 2.5, "packed16": 4.6, "dpp": 4.75, "f64_or_trans": 4.6}
def kind(ins, ops):
    if "row_" in ops or "quad_perm" in ops or "wave_" in ops or "_dpp" in ins: return "dpp"
    if ins.startswith("v_pk_"): return "packed16"
    if "_f64" in ins or ins.startswith(("v_rcp", "v_sqrt", "v_log", "v_exp", "v_sin", "v_cos")): return "f64_or_trans"
    return "plain"
def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    with tempfile.TemporaryDirectory() as td:
        asm = os.environ.get("AL_ALIGN_ASM") or os.path.join(td, "align.s")
        if not os.path.exists(asm): subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result", "-mllvm", "-two-entry-phi-node-folding-threshold=200",
                               "-S", "--cuda-device-only", "-I", CS, "-o", asm, os.path.join(CS, "al_kernels_align.hip")], stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    out = {"source": "static count over the row loop of each kernel in the assembly of al_kernels_align.hip (tools/valu_mix.py)", "cycles_per_kind": CYC,
           "cycles_source": "profiles/r05_valu_issue.txt (tools/micro/valu_issue.hip), 2 - 4 wavefronts per SIMD", "kernels": {}}
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w*k_ext_dp\w*):", lines[i])
        if not m: i += 1; continue
        name = m.group(1); j = i + 1; body = []
        while j < len(lines) and "s_endpgm" not in lines[j]: body.append(lines[j]); j += 1
        i = j
        label_at = {}
        for n, l in enumerate(body):
            lm = re.match(r"^(\.LBB\w+):", l)
            if lm: label_at[lm.group(1)] = n
        best = None
        for n, l in enumerate(body):                      # back edges: a branch to a label above it
            bm = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\w+)", l) or re.match(r"\s+s_branch\s+(\.LBB\w+)", l)
            if not bm or bm.group(1) not in label_at or label_at[bm.group(1)] >= n: continue
            seg = body[label_at[bm.group(1)]:n + 1]
            pk = sum(1 for s in seg if re.match(r"\s+v_pk_", s)); nv = sum(1 for s in seg if re.match(r"\s+v_", s))
            key = (pk, -len(seg)) if pk else (0, nv)          # the tightest loop holding the packed instructions; a kernel without any (one cell per lane): its largest loop
            if best is None or key > best[0]: best = (key, seg)
        if best is None: continue
        cnt = {k: 0 for k in CYC}; lds = vmem = salu = 0
        for s in best[1]:
            im = re.match(r"\s+([a-z_0-9]+)\s*(.*)", s)
            if not im: continue
            ins, ops = im.group(1), im.group(2)
            if ins.startswith("v_"): cnt[kind(ins, ops)] += 1
            elif ins.startswith("ds_"): lds += 1
            elif ins.startswith(("global_", "flat_", "buffer_", "scratch_")): vmem += 1
            elif ins.startswith("s_"): salu += 1
        nv = sum(cnt.values())
        if nv == 0: continue
        cyc = sum(cnt[k] * CYC[k] for k in cnt) / nv
        tm = re.search(r"(k_ext_dp\w*?)I((?:L[ib]\d+E)+)E", name)       # template arguments from the mangled name: Li16E -> 16, Lb1E -> true
        if not tm: continue
        dem = tm.group(1) + "<" + ", ".join((("true" if v == "1" else "false") if t == "b" else v) for t, v in re.findall(r"L([ib])(\d+)E", tm.group(2))) + ">"
        out["kernels"][dem] = {"valu_in_row_loop": nv, "by_kind": cnt, "lds_insts": lds, "vmem_insts": vmem, "salu_insts": salu,
                               "cycles_per_valu_inst": cyc, "issue_ceiling_wave_insts_per_s": 1024 * 2.4e9 / cyc}
    p = os.path.join(ROOT, "profiles", "%s_valu_mix.json" % tag)
    json.dump(out, open(p, "w"), indent=1)
    for k, v in out["kernels"].items(): print("%-44s VALU %4d %s  %.2f cycles -> %.3g /s" % (k, v["valu_in_row_loop"], v["by_kind"], v["cycles_per_valu_inst"], v["issue_ceiling_wave_insts_per_s"]))
if __name__ == "__main__": main()
