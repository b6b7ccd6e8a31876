"""CPU check of the list-ranking algorithm as the kernels restate it (tools/rank_model.py: k_rank_rulers with tags,
k_rank_lds with terminal / dead / hopping entries, k_link_jump) against a direct walk: chains, isolated cycles, dropped
edges, rulers at every 2nd / 4th / 8th edge.  The kernels themselves are compared with the oracle in the -m gpu tests."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import rank_model  # noqa: E402


def test_list_ranking_model_against_direct_walk():
    for seed in range(40):
        assert rank_model.check(seed) == [], f"seed {seed}"
