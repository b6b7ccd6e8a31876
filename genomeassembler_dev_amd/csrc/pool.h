// pool.h — one rank's state of a pooled (multi-GPU) build; shared by pool.hip (the stage-by-stage gasm_pool_* entry points) and
// exchange.hip (the whole step with the all-to-alls inside the library).
#pragma once
#include "pipeline.h"

struct gasm_pool {
    gasm_ctx* ctx = nullptr;
    DevReads rd;            // this rank's reads of all segments
    BuildState bs;          // current runs (d_keys / d_mult / d_bstart / d_bucket_d), later the rank's graph
    u32 n_runs = 0;         // buckets the current runs cover
    std::vector<u32> h_len; // their lengths
    DevReads own;           // the reads of the rank's own segments (gasm_pool_set_reads)
    u32 n_local = 0;
    bool graphed = false, reads_set = false;
    DevPaths dp;
    ScoreTable tb;
    ScoreState ss;
    bool table_set = false, paths_ready = false;
    bool scored = false;    // a scoring is queued behind the current graph (queued again if the graph had to be repeated)
    int score_kmer = 0;
    std::vector<double> table_copy;
    DBuf d_list, d_off, d_roff, d_rlen;
};

// the graph / contigs of the pool's current runs = the buckets of n_local segments, D distinct k-mers in all, at most maxD in one
// segment (gasm_pool_graph reads both off its host copy of the run lengths; exchange.hip gets them with its plan's report)
int pool_graph_launch(gasm_pool* p, u32 n_local_segments, u64 D, u64 maxD);
// breakage scoring of the rank's contigs, queued behind the graph without waiting for its report when every read holds a k-mer
int pool_score_launch(gasm_pool* p, int kmer, const double* table, bool wait_for_build);
// read the graph's report (every fetch does); a graph that had to be repeated takes its scoring with it
int pool_finish(gasm_pool* p);
