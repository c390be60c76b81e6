"""ctypes mirror of include/airlift.h (names and argument meaning follow the C-ABI, which in turn
follows the reference's minimap.h for this path)."""
import ctypes as C
import gzip
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


class AirliftError(RuntimeError):
    pass


def lib_path():
    # AIRLIFT_LIB: another build of the same library (A/B timing of kernel variants on one GPU box)
    return os.environ.get("AIRLIFT_LIB") or os.path.join(HERE, "lib", "libairlift.so")


def build(verbose=False):
    """Compile every HIP source for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-s", "-j4", "-C", os.path.join(HERE, "csrc")], capture_output=not verbose)
    if r.returncode != 0:
        raise AirliftError("build failed:\n" + (r.stderr.decode() if r.stderr else ""))
    return lib_path()


class IdxOpt(C.Structure):
    _fields_ = [("k", C.c_short), ("w", C.c_short), ("flag", C.c_short), ("bucket_bits", C.c_short),
                ("mini_batch_size", C.c_int), ("batch_size", C.c_uint64)]


class MapOpt(C.Structure):
    _fields_ = [("flag", C.c_int64), ("seed", C.c_int), ("bw", C.c_int), ("max_gap", C.c_int), ("max_gap_ref", C.c_int),
                ("max_frag_len", C.c_int), ("max_chain_skip", C.c_int), ("max_chain_iter", C.c_int), ("min_cnt", C.c_int),
                ("min_chain_score", C.c_int), ("mask_level", C.c_float), ("pri_ratio", C.c_float), ("best_n", C.c_int),
                ("a", C.c_int), ("b", C.c_int), ("q", C.c_int), ("e", C.c_int), ("q2", C.c_int), ("e2", C.c_int),
                ("sc_ambi", C.c_int), ("zdrop", C.c_int), ("zdrop_inv", C.c_int), ("end_bonus", C.c_int), ("min_dp_max", C.c_int),
                ("max_clip_ratio", C.c_float), ("pe_ori", C.c_int), ("pe_bonus", C.c_int), ("mid_occ", C.c_int32),
                ("max_occ", C.c_int32), ("mini_batch_size", C.c_int)]


class Reg(C.Structure):
    _fields_ = [("id", C.c_int32), ("cnt", C.c_int32), ("rid", C.c_int32), ("score", C.c_int32),
                ("qs", C.c_int32), ("qe", C.c_int32), ("rs", C.c_int32), ("re", C.c_int32),
                ("parent", C.c_int32), ("subsc", C.c_int32), ("mlen", C.c_int32), ("blen", C.c_int32),
                ("n_sub", C.c_int32), ("score0", C.c_int32),
                ("mapq", C.c_uint32, 8), ("split", C.c_uint32, 2), ("rev", C.c_uint32, 1), ("inv", C.c_uint32, 1),
                ("sam_pri", C.c_uint32, 1), ("proper_frag", C.c_uint32, 1), ("pe_thru", C.c_uint32, 1), ("seg_split", C.c_uint32, 1),
                ("seg_id", C.c_uint32, 8), ("split_inv", C.c_uint32, 1), ("dummy", C.c_uint32, 7),
                ("hash", C.c_uint32), ("dp_score", C.c_int32), ("dp_max", C.c_int32), ("dp_max2", C.c_int32),
                ("n_ambi", C.c_uint32), ("n_cigar", C.c_uint32), ("cigar", C.POINTER(C.c_uint32))]


class BatchStat(C.Structure):
    _fields_ = [("n_frag", C.c_uint64), ("n_reads", C.c_uint64), ("n_bases", C.c_uint64), ("n_mini", C.c_uint64),
                ("n_anchor", C.c_uint64), ("n_chain", C.c_uint64), ("n_regs_aln", C.c_uint64), ("n_refbases", C.c_uint64),
                ("n_cigar", C.c_uint64), ("n_rechain", C.c_uint64), ("n_heap_fallback", C.c_uint64), ("n_sort_tie_flag", C.c_uint64),
                ("bytes_in", C.c_uint64), ("bytes_out", C.c_uint64), ("algorithmic_bytes", C.c_double),
                ("ms_total", C.c_float), ("ms_kernel", C.c_float * 40), ("n_stage", C.c_int), ("ms_side_stream", C.c_float), ("n_chain_fallback", C.c_uint64),
                ("dp_jobs", C.c_uint64 * 10), ("dp_target_bases", C.c_uint64 * 10)]


_lib = None


def load():
    """dlopen libairlift.so; fails loudly if it is missing (no fallback of any kind)."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise AirliftError("libairlift.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc)")
    L = C.CDLL(p)
    vp, ci, cs = C.c_void_p, C.c_int, C.c_char_p
    L.al_set_opt.argtypes = [cs, C.POINTER(IdxOpt), C.POINTER(MapOpt)]; L.al_set_opt.restype = ci
    L.al_check_opt.argtypes = [C.POINTER(IdxOpt), C.POINTER(MapOpt)]; L.al_check_opt.restype = ci
    L.al_idx_build.argtypes = [cs, C.POINTER(IdxOpt), ci]; L.al_idx_build.restype = vp
    L.al_idx_build_device.argtypes = [cs, C.POINTER(IdxOpt), ci]; L.al_idx_build_device.restype = vp
    L.al_idx_export_pos.argtypes = [vp, vp, C.c_int64]; L.al_idx_export_pos.restype = C.c_int64
    L.al_ctx_set_threads.argtypes = [vp, ci]; L.al_ctx_set_threads.restype = None
    L.al_batch_count_candidates.argtypes = [vp, C.POINTER(C.c_int64)]; L.al_batch_count_candidates.restype = ci
    L.al_count_candidates_file.argtypes = [vp, cs, C.POINTER(MapOpt), ci, ci, C.POINTER(C.c_int64)]; L.al_count_candidates_file.restype = ci
    L.al_idx_str.argtypes = [ci, ci, ci, C.POINTER(cs), C.POINTER(cs)]; L.al_idx_str.restype = vp
    L.al_idx_destroy.argtypes = [vp]; L.al_idx_destroy.restype = None
    L.al_idx_n_seq.argtypes = [vp]; L.al_idx_n_seq.restype = C.c_uint32
    L.al_idx_seq_name.argtypes = [vp, C.c_uint32]; L.al_idx_seq_name.restype = cs
    L.al_idx_seq_len.argtypes = [vp, C.c_uint32]; L.al_idx_seq_len.restype = C.c_uint32
    L.al_idx_stat.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]; L.al_idx_stat.restype = None
    L.al_ctx_init.argtypes = [vp, C.POINTER(MapOpt), ci]; L.al_ctx_init.restype = vp
    L.al_ctx_destroy.argtypes = [vp]; L.al_ctx_destroy.restype = None
    L.al_batch_upload.argtypes = [vp, ci, C.POINTER(ci), C.POINTER(ci), C.POINTER(cs), C.POINTER(cs)]; L.al_batch_upload.restype = ci
    L.al_batch_run.argtypes = [vp]; L.al_batch_run.restype = ci
    L.al_batch_fetch.argtypes = [vp, C.POINTER(ci), C.POINTER(C.POINTER(Reg)), C.POINTER(ci)]; L.al_batch_fetch.restype = ci
    L.al_map_batch.argtypes = [vp, ci, C.POINTER(ci), C.POINTER(ci), C.POINTER(cs), C.POINTER(cs), C.POINTER(ci), C.POINTER(C.POINTER(Reg)), C.POINTER(ci)]
    L.al_map_batch.restype = ci
    L.al_map_frag.argtypes = [vp, ci, C.POINTER(ci), C.POINTER(cs), C.POINTER(ci), C.POINTER(C.POINTER(Reg)), vp, C.POINTER(MapOpt), cs]; L.al_map_frag.restype = None
    L.al_map_file_frag.argtypes = [vp, ci, C.POINTER(cs), C.POINTER(MapOpt), ci, vp, cs, ci]; L.al_map_file_frag.restype = ci
    L.al_map_file_frag_bam.argtypes = [vp, ci, C.POINTER(cs), C.POINTER(MapOpt), ci, vp, cs, ci, ci, ci]; L.al_map_file_frag_bam.restype = ci
    L.al_map_tokens_file.argtypes = [vp, cs, ci, ci, C.POINTER(MapOpt), ci, vp, cs, ci]; L.al_map_tokens_file.restype = ci
    L.al_winsrc_create.argtypes = [vp, cs, C.c_uint64]; L.al_winsrc_create.restype = vp
    L.al_winsrc_destroy.argtypes = [vp]; L.al_winsrc_destroy.restype = None
    L.al_batch_upload_windows.argtypes = [vp, vp, ci, C.POINTER(C.c_uint64), ci, C.POINTER(cs)]; L.al_batch_upload_windows.restype = ci
    L.al_batch_stat.argtypes = [vp, C.POINTER(BatchStat)]; L.al_batch_stat.restype = None
    L.al_stage_name.argtypes = [ci]; L.al_stage_name.restype = cs
    L.al_stage_kernel.argtypes = [ci]; L.al_stage_kernel.restype = cs
    L.al_dbg_copy.argtypes = [vp, cs, vp, C.c_int64]; L.al_dbg_copy.restype = C.c_int64
    L.al_dbg_alser_count.argtypes = [vp, C.POINTER(C.c_int64)]; L.al_dbg_alser_count.restype = ci
    L.al_write_sam.argtypes = [C.c_char_p, C.c_size_t, vp, cs, ci, cs, cs, ci, ci, ci, C.POINTER(ci), C.POINTER(C.POINTER(Reg)), cs, ci]; L.al_write_sam.restype = ci
    L.al_dbg_ksw.argtypes = [vp, ci, vp, C.c_size_t, vp, vp, vp, ci]; L.al_dbg_ksw.restype = ci
    L.al_version.restype = cs
    _lib = L
    return L


def read_fastx(path):
    """Minimal FASTA/FASTQ(.gz) reader for tests: returns (names, seqs, quals) as lists of bytes."""
    op = gzip.open if path.endswith(".gz") else open
    names, seqs, quals = [], [], []
    with op(path, "rb") as f:
        data = f.read().split(b"\n")
    i = 0
    while i < len(data):
        l = data[i]
        if not l:
            i += 1; continue
        if l[:1] == b"@":
            names.append(l[1:].split()[0]); seqs.append(data[i + 1].strip()); quals.append(data[i + 3].strip()); i += 4
        elif l[:1] == b">":
            nm = l[1:].split()[0]; i += 1; s = []
            while i < len(data) and data[i][:1] != b">":
                s.append(data[i].strip()); i += 1
            names.append(nm); seqs.append(b"".join(s)); quals.append(None)
        else:
            i += 1
    return names, seqs, quals


class Index:
    """al_idx_t (replaces mm_idx_t)."""

    def __init__(self, fasta=None, seqs=None, names=None, preset="sr", n_threads=4, on_device=None, k=None, w=None):
        L = load()
        self.io, self.mo = IdxOpt(), MapOpt()
        L.al_set_opt(None, C.byref(self.io), C.byref(self.mo))
        if L.al_set_opt(preset.encode(), C.byref(self.io), C.byref(self.mo)) != 0:
            raise AirliftError("unknown preset " + preset)
        if k is not None:
            self.io.k = k
        if w is not None:
            self.io.w = w
        self.mo.flag |= 0x004 | 0x008
        if fasta is not None and on_device is not None:     # sketch + sort + table built by kernels on that GPU
            self.h = L.al_idx_build_device(fasta.encode(), C.byref(self.io), on_device)
        elif fasta is not None:
            self.h = L.al_idx_build(fasta.encode(), C.byref(self.io), n_threads)
        else:
            n = len(seqs)
            sa = (C.c_char_p * n)(*seqs); na = (C.c_char_p * n)(*names)
            self.h = L.al_idx_str(self.io.w, self.io.k, n, sa, na)
        if not self.h:
            raise AirliftError("index build failed")

    @property
    def names(self):
        L = load()
        return [L.al_idx_seq_name(self.h, i).decode() for i in range(L.al_idx_n_seq(self.h))]

    def stat(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        load().al_idx_stat(self.h, C.byref(a), C.byref(b), C.byref(c))
        return {"n_keys": a.value, "n_pos": b.value, "n_bases": c.value}

    def cal_max_occ(self, f):
        """mm_idx_cal_max_occ (index.c:164-185)."""
        fn = load().al_idx_cal_max_occ
        fn.restype = C.c_int32; fn.argtypes = [C.c_void_p, C.c_float]
        v = fn(self.h, f)
        if v < 0:
            raise AirliftError("al_idx_cal_max_occ failed")
        return v

    def positions(self):
        """The occurrence array (grouped by minimizer hash ascending, positions ascending inside a group)."""
        n = load().al_idx_export_pos(self.h, None, 0)
        out = np.zeros(max(n, 0), dtype=np.uint64)
        if n > 0 and load().al_idx_export_pos(self.h, out.ctypes.data_as(C.c_void_p), n) != n:
            raise AirliftError("al_idx_export_pos failed")
        return out

    def close(self):
        if self.h:
            load().al_idx_destroy(self.h); self.h = None


class Context:
    """al_ctx_t (replaces mm_tbuf_t): one per host thread / GPU."""

    def __init__(self, index, device=-1):
        self.idx = index
        self.h = load().al_ctx_init(index.h, C.byref(index.mo), device)
        if not self.h:
            raise AirliftError("al_ctx_init failed: no usable HIP device (there is no CPU fallback)")
        self._keep = None

    def upload(self, n_segs, seqs, names):
        L = load()
        nf, nr = len(n_segs), len(seqs)
        a_ns = (C.c_int * nf)(*n_segs); a_ql = (C.c_int * nr)(*[len(s) for s in seqs])
        a_sq = (C.c_char_p * nr)(*seqs); a_nm = (C.c_char_p * nr)(*names)
        self._keep = (a_ns, a_ql, a_sq, a_nm); self.n_frag, self.n_reads = nf, nr
        rc = L.al_batch_upload(self.h, nf, a_ns, a_ql, a_sq, a_nm)
        if rc != 0:
            raise AirliftError("al_batch_upload failed: %d" % rc)

    def run(self):
        rc = load().al_batch_run(self.h)
        if rc != 0:
            raise AirliftError("al_batch_run failed: %d" % rc)

    def stat(self):
        st = BatchStat(); load().al_batch_stat(self.h, C.byref(st)); return st

    def fetch(self):
        """Returns (n_regs[list], regs[list of list of dict], rep_len[list])."""
        L = load()
        n_regs = (C.c_int * self.n_reads)(); regs = (C.POINTER(Reg) * self.n_reads)(); rep = (C.c_int * self.n_frag)()
        rc = L.al_batch_fetch(self.h, n_regs, regs, rep)
        if rc != 0:
            raise AirliftError("al_batch_fetch failed: %d" % rc)
        return n_regs, regs, rep

    def tap(self, name, dtype, count):
        arr = np.zeros(count, dtype=dtype)
        n = load().al_dbg_copy(self.h, name.encode(), arr.ctypes.data_as(C.c_void_p), arr.nbytes)
        if n < 0:
            raise AirliftError("tap %s failed" % name)
        return arr[: n // arr.itemsize]

    def alser_count(self):
        v = C.c_int64()
        if load().al_dbg_alser_count(self.h, C.byref(v)) != 0:
            raise AirliftError("alser count failed")
        return v.value

    def close(self):
        if self.h:
            load().al_ctx_destroy(self.h); self.h = None
