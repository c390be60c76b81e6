"""TEST INFRASTRUCTURE (oracle/): CPU restatement of the candidate enumeration and the adjacency filter of SURVEY.md N4.

Only tests/ may import this; nothing in airlift_amd/ or include/ does.

* candidates(): the clusters the minimap2 fork's ALSER loop counts (src/minimap2-master_remapping/map.c:299-312): scan the sorted anchors of a
  read, close a cluster where (int32)a[i].x - (int32)a[i-1].x > qlen, count it if it has >= min_cnt - 1 steps; the last cluster is never counted.
* adjacency(): the mrFAST fork's adjacency rule (src/mrfast-master_remapping/MrFAST.c:1741-1764) applied to the read's minimizer seeds: the
  candidate's diagonal predicts where every other seed of the read must occur; `searchKey` = membership of that position word in the seed's sorted
  occurrence list; more than adj_e absent seeds reject.  (The fork applies the rule to its 12-mers; on minimizer seeds it is a re-interpretation --
  parity of this half is against THIS restatement, the GreedySnake half is pinned to the reference's own GreedySnake.c, oracle/_ref/libgreedysnake.so.)
"""
import numpy as np


def candidates(anchors, qlen, min_cnt=2):
    """anchors: (n, 2) uint64 array (x, y) of one read, sorted as the device / reference leaves them -> list of cluster start indices."""
    xs = (anchors[:, 0] & np.uint64(0xffffffff)).astype(np.int64)
    xs = np.where(xs >= 2 ** 31, xs - 2 ** 32, xs)
    out, seed_num, cs = [], 0, 0
    for i in range(1, len(anchors)):
        if xs[i] - xs[i - 1] > qlen:                                    # map.c:301-308
            if seed_num >= min_cnt - 1:
                out.append(cs)
            seed_num, cs = 0, i
        else:
            seed_num += 1
    return out


def candidate_location(anchor_x, anchor_y):
    """(rev, rid, ref_start) of the candidate whose first anchor is (x, y): both coordinates are k-mer end positions."""
    ax, ay = int(anchor_x), int(anchor_y)
    rev = ax >> 63
    rid = (ax << 1 & (2 ** 64 - 1)) >> 33
    return rev, rid, (ax & 0xffffffff) - (ay & 0xffffffff)


def adjacency(seeds, rev, rid, ref_start, qlen, k, ref_len, adj_e):
    """seeds: [(q_pos << 1 | strand, sorted occurrence words rid << 32 | pos << 1 | strand)] of the read (collect_matches, map.c:90-123)."""
    diff = 0
    for qp, lst in seeds:
        qend, qs = qp >> 1, qp & 1
        rp = ref_start + (qlen - (qend + 1 - k) - 1) if rev else ref_start + qend     # map.c:183: the k-mer on the reverse-complemented read
        word = rid << 32 | rp << 1 | ((1 - qs) if rev else qs)
        if not (0 <= rp < ref_len and word in lst):
            diff += 1
    return diff <= adj_e
