// scaffolds.h — device-resident result of assemble_contigs (scaffolds.hip)
#pragma once
#include "pipeline.h"

struct gasm_scaffolds {
    gasm_ctx* ctx = nullptr;
    DBuf d_words;               // 2-bit, scaffold after scaffold without gaps, + 4 padding words
    std::vector<u64> h_off;     // n + 1 base offsets, final (reference) order: longest first
    u32 n = 0;
    u64 rows_total = 0, rows_on_host = 0;   // permutations merged in all / by the host routine (k_asm_merge hands back rows it cannot decide)
};

int scaffolds_from_signatures(gasm_ctx* ctx, const std::vector<std::string>& contigs, const std::vector<std::string>& sigs, gasm_scaffolds** out);
int assemble_signatures_device(gasm_ctx* ctx, const std::vector<std::string>& contigs, const u32* perm, u64 rows, u64 row_len, int k,
                               std::vector<std::string>& sigs, bool* used, u64* rows_on_host);
int scaffolds_fetch(const gasm_scaffolds* sc, std::vector<char>& data, std::vector<u64>& off);
int scaffolds_as_paths(const gasm_scaffolds* sc, DevPaths& dp);

// Row A16: guided scaffolds of a built and scored batch (scaffolds.hip, k_guided_chain)
struct GuidedState {
    bool valid = false;
    DevPaths dp;                        // the guided scaffolds as paths (segment-major)
    ScoreState ss;                      // their scores (general scorer)
    std::vector<u64> h_seg_off;         // n_segments + 1 scaffold indices
    std::vector<char> h_text;           // filled by fetch
    std::vector<u64> h_text_off;
    void release() { dp.release(); ss.release(); valid = false; }
};
int guided_build(gasm_ctx* ctx, DevReads& rd, BuildState& bs, DevPaths& contig_paths, ScoreState& contig_scores, const ScoreTable& tb, int kmer,
                 GuidedState& out);
int guided_fetch_text(gasm_ctx* ctx, GuidedState& g);
