// align.h -- the reference's tile-aligner surface (reference align.h:18-40),
// kept so a darwin.cpp-shaped caller compiles unchanged.  Behind it there is
// no CPU aligner: AlignWithBT and Align_Batch run the HIP engine through the
// C-ABI of include/gact_hip.h (gact_shim.cpp).
#ifndef DARWIN_HIP_ALIGN_H
#define DARWIN_HIP_ALIGN_H

#include <iostream>
#include <queue>
#include <string>
#include <vector>

#define INF (1 << 30)
#define MAX_TILE_SIZE 2049

typedef int AlnOp;
enum AlnOperands { ZERO_OP, DELETE_OP, INSERT_OP, MATCH_OP };
enum states { Z, D, I, M };

// Same arguments and the same returned queue as reference align.cpp:60-233:
//   first  -> [max_score, max_i, max_j, states...],  !first -> [pos_score, states...]
// ref_len, query_len < 2049 as in the reference (align.cpp:66); beyond 512 a slower kernel takes the tile (csrc/gact_big.hpp).
std::queue<int> AlignWithBT(char *ref_seq, long long int ref_len,
                            char *query_seq, long long int query_len,
                            int match_score, int mismatch_score, int gap_open, int gap_extend,
                            int query_pos, int ref_pos, bool reverse, bool first, int early_terminate);

// reference align.cpp:17-54; ref_lens[j] == -1 yields an empty queue
std::vector<std::queue<int> > Align_Batch(std::vector<std::string> ref_seqs,
                                          std::vector<std::string> query_seqs,
                                          std::vector<int> ref_lens, std::vector<int> query_lens,
                                          int match_score, int mismatch_score, int gap_open, int gap_extend,
                                          std::vector<int> ref_poss_b, std::vector<int> query_poss_b,
                                          std::vector<char> reverses, std::vector<char> firsts,
                                          int early_terminate);

#endif
