"""Pins the CPU oracle to the only reference-produced output available: the toy known-answer vector of
SURVEY.md §8(c) (tests/golden/kat_survey_toy.json)."""
import json
import os

import numpy as np

from oracle import orc

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat_survey_toy.json")))


def _reads():
    g = G["genome"]
    return [g[s:s + G["read_len"]] for s in G["read_starts"]]


def test_kat_graph_and_shuffle():
    km = orc.kmers_from_reads(_reads(), G["dbg_kmer"])
    assert len(km) == G["n_kmers"]
    r = orc.get_contigs(km, G["dbg_kmer"], G["seed"])
    assert r["contigs"] == G["contigs"]
    assert r["perm"].shape == (10000, 4)
    assert [r["contigs"][i] for i in r["perm"][0]] == G["shuffle_row0"]


def test_kat_scaffolds_and_scores(qtable):
    keys, prob = qtable
    km = orc.kmers_from_reads(_reads(), G["dbg_kmer"])
    r = orc.get_contigs(km, G["dbg_kmer"], G["seed"])
    sc = orc.assemble_contigs(r["contigs"], r["perm"], G["dbg_kmer"])
    assert [len(s) for s in sc] == G["scaffold_lens"]
    assert (sc[0] == G["genome"]) == G["scaffold0_is_genome"]
    b = orc.calc_breakscore(sc, _reads(), G["genome"], G["break_kmer"], keys, prob)
    assert b["kmer_breaks"].tolist() == G["kmer_breaks"]
    assert b["lev_dist_vs_true"].tolist() == G["lev_dist_vs_true"]
    # FP: the reference sums in gtl hash order; tolerance from the north star is 1e-9 absolute
    assert abs(b["bp_score"][0] - G["bp_score_0"]) < 1e-15
    assert abs(b["bp_score_norm_by_break_freqs"][0] - G["bp_score_norm_by_break_freqs_0"]) < 1e-15
    assert abs(b["bp_score_norm_by_len"][0] - G["bp_score_norm_by_len_0"]) < 1e-15
    assert [len(b["path_freq"]), len(b["path_freq"][0])] == G["path_freq_shape"]
    assert int((b["path_freq"][0] > 0).sum()) == G["path_freq_nonzeros_0"]


def test_table_normalisation(qtable):
    _, prob = qtable
    raw = np.fromfile(os.path.join(os.path.dirname(__file__), "..", "genomeassembler_dev_amd", "data",
                                   "querytable_raw_f64.bin"), dtype="<f8")
    assert abs(float(np.sum(raw.astype(np.longdouble))) - G["table_sum"]) < 1e-9
    assert abs(prob.sum() - 1.0) < 1e-12
    assert prob.size == 69904 and (prob > 0).all()
