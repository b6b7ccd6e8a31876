"""ORACLE (test infrastructure) for SURVEY §8 row A16 — breakage-score-guided traversal.  The reference does not have this
mode (README.md:83: "out of scope"); the specification is this project's own and this file is its plain CPU
restatement, the thing libgasm's gasm_batch_guided is checked against.  PARITY UNPINNED by the reference.

Specification.  Input: one segment's contigs (sorted, distinct; as get_contigs gives them), its reads, the normalised
breakage table, k, the break k-mer size (8) and the fixed-point shift of the batch.
  score(c)  = fx(c) / len(c), an exact rational, where fx(c) = sum over the reads that occur in c (first occurrence,
              lib/DeNovoAssembler.cpp:360) of round(prob(window of the hit, :366-386) * 2^shift)   [the batch scorer's sum]
  b follows a  <=>  a != b and the last k-1 bases of a are the first k-1 bases of b
  repeat: seed = the unused contig with the highest score (ties: the smaller index); path = [seed];
          extend to the right with the best-scoring unused contig that follows the path's last contig, until there is none;
          then to the left with the best-scoring unused contig that the path's first contig follows, until there is none.
  A path reads as its contigs with the k-1 shared bases written once.  Output: the paths by descending length, ties by
  the index of their first contig."""
from fractions import Fraction


def break_window_prob(path, pos, kmer, table):
    """lib/DeNovoAssembler.cpp:366-386 (+ a window cut short by the end of the path to a length the table does not hold: 0)"""
    start = max(0, pos - kmer // 2)
    width = 8
    if start == 0 and pos in (1, 2, 3):
        width = 2 * pos
    return table.get(path[start:start + width], 0.0)


def fixed_sums(contigs, reads, table, kmer, shift):
    out = []
    for c in contigs:
        fx = 0
        for r in reads:
            p = c.find(r)
            if p >= 0:
                fx += int(round(break_window_prob(c, p, kmer, table) * 2.0 ** shift))
        out.append(fx)
    return out


def guided_paths(contigs, fx, k):
    n = len(contigs)
    score = [Fraction(fx[i], len(contigs[i])) for i in range(n)]
    used = [False] * n
    k1 = k - 1

    def best(cands):
        b = None
        for j in cands:
            if b is None or score[j] > score[b]:        # ascending j: ties keep the smaller index
                b = j
        return b

    paths = []
    while True:
        seed = best([j for j in range(n) if not used[j]])
        if seed is None:
            break
        used[seed] = True
        path = [seed]
        while True:
            cur = path[-1]
            nx = best([j for j in range(n) if not used[j] and contigs[j][:k1] == contigs[cur][-k1:]])
            if nx is None:
                break
            used[nx] = True
            path.append(nx)
        while True:
            cur = path[0]
            pv = best([j for j in range(n) if not used[j] and contigs[j][-k1:] == contigs[cur][:k1]])
            if pv is None:
                break
            used[pv] = True
            path.insert(0, pv)
        paths.append(path)
    seqs = []
    for p in paths:
        s = contigs[p[0]]
        for j in p[1:]:
            s += contigs[j][k1:]
        seqs.append((len(s), p[0], s))
    seqs.sort(key=lambda t: (-t[0], t[1]))
    return [s for _, _, s in seqs]
