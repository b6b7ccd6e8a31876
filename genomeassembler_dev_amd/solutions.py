"""The solutions table of one experiment: what `score_solutions()` and `save_results()` of the reference's R6 class leave
on disk (lib/DeNovoAssembler.R:318-479, 268-313) — scores under the true and under the uniform ("random") breakage table,
the KS statistic of every solution's path_freq against the genome's window probabilities, the fraction of the genome
the solutions cover, joined and written as SolutionsTable_….csv with the reference's column set.  The numbers come from
libgasm (calc_breakscore with the KS flag, coverage_percent); this module only orders, joins and formats them."""
import numpy as np

from . import api, qtable

KEYS = ("sequence", "sequence_len", "kmer_breaks", "contig_frac_len", "lev_dist_vs_true")       # join keys, DeNovoAssembler.R:465-469
# columns of calc_breakscore's list in order (lib/DeNovoAssembler.cpp:467-476) without path_freq, + the two added in R
_BASE = ("sequence", "sequence_len", "bp_score", "bp_score_norm_by_break_freqs", "bp_score_norm_by_len", "kmer_breaks",
         "lev_dist_vs_true", "stat_test_KS", "contig_frac_len")
COLUMNS = tuple(c if c in KEYS else c + "_true" for c in _BASE) + tuple(c + "_random" for c in _BASE if c not in KEYS)


def score_solutions_one(paths, reads, true_solution, kmer, bp_kmer, bp_prob, seq_len=None, ctx=None):
    """One pass of the lapply in score_solutions() (DeNovoAssembler.R:325-459), own-assembler mode: calc_breakscore, rows
    ordered by descending bp_score (stable, as data.table's setorder), path_freq_startpos = 0 for every row, KS per row,
    contig_frac_len from the ranges [0, sequence_len]."""
    seq_len = len(true_solution) if seq_len is None else seq_len
    r = api.calc_breakscore(paths, reads, true_solution, kmer, bp_kmer, bp_prob, variant="own", with_lev=True, with_freq=False,
                            with_ks=True, ctx=ctx)
    order = np.argsort(-np.asarray(r["bp_score"]), kind="stable")
    out = {c: (np.asarray(r[c])[order] if c != "sequence" else [r[c][i] for i in order]) for c in _BASE if c in r}
    n = len(order)
    if n:
        cov = api.coverage_percent(np.zeros(n, dtype=np.int64), out["sequence_len"], seq_len, ctx=ctx)
    else:
        cov = 0.0                                                            # DeNovoAssembler.R:446-450
    out["contig_frac_len"] = np.full(n, cov, dtype=np.float64)
    return out


def join_true_random(true_res, random_res):
    """dplyr::inner_join(x = true, y = random, by = KEYS, suffix = c("_true", "_random")) (DeNovoAssembler.R:461-472): x's
    row order; columns = x's (clashing non-keys suffixed) then y's clashing non-keys."""
    def key(res, i):
        return (res["sequence"][i], int(res["sequence_len"][i]), int(res["kmer_breaks"][i]), float(res["contig_frac_len"][i]),
                int(res["lev_dist_vs_true"][i]))
    idx = {}
    for j in range(len(random_res["sequence"])):
        idx.setdefault(key(random_res, j), []).append(j)
    rows = [(i, j) for i in range(len(true_res["sequence"])) for j in idx.get(key(true_res, i), [])]
    table = {}
    for c in _BASE:
        src = true_res[c]
        name = c if c in KEYS else c + "_true"
        table[name] = [src[i] for i, _ in rows]
    for c in _BASE:
        if c not in KEYS:
            table[c + "_random"] = [random_res[c][j] for _, j in rows]
    return table


def score_solutions(paths, reads, true_solution, kmer=8, ctx=None):
    """both passes (true table, then 1/N for every row: DeNovoAssembler.R:326-333) and the join"""
    keys = qtable.keys()
    t = score_solutions_one(paths, reads, true_solution, kmer, keys, qtable.load_normalised(), ctx=ctx)
    u = score_solutions_one(paths, reads, true_solution, kmer, keys, qtable.uniform(), ctx=ctx)
    return join_true_random(t, u)


def _fmt(v):
    if isinstance(v, str):
        return v
    if isinstance(v, (int, np.integer)):
        return str(int(v))
    v = float(v)
    if np.isnan(v):
        return "NA"                       # data.table::fwrite writes NA
    s = "%.15g" % v                       # fwrite: up to 15 significant digits
    return s


def write_solutions_csv(path, table):
    """data.table::fwrite(self$results, file) with its defaults: header, comma, NA for missing"""
    cols = [c for c in COLUMNS if c in table]
    n = len(table[cols[0]]) if cols else 0
    with open(path, "w") as f:
        f.write(",".join(cols) + "\n")
        for i in range(n):
            f.write(",".join(_fmt(table[c][i]) for c in cols) + "\n")
    return path


def solutions_filename(seq_len, seed, read_len, dbg_kmer, kmer, industry=False):
    """DeNovoAssembler.R:296-306"""
    return (f"SolutionsTable_SeqLen-{seq_len}_SeqSeed-{seed}_ReadLen-{read_len}_DBGKmer-{dbg_kmer}_kmer-{kmer}"
            f"_IndustryModel-{'TRUE' if industry else 'FALSE'}.csv")
