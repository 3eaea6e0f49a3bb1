// gact.h -- the reference's GACT surface (reference gact.h:25-98) over the
// MI355X engine.  Same names, argument order and meaning, so darwin.cpp's call
// sites (darwin.cpp:240-246, 279-285, 429-433, 611, 642) compile unchanged.
// Both the "CPU build" entry (GACT) and the "GPU build" entries (GPU_init,
// GACT_Batch, Align_Batch_GPU, GPU_close) are always available; there is no
// -D GPU switch and no CPU aligner behind any of them.
#ifndef DARWIN_HIP_GACT_H
#define DARWIN_HIP_GACT_H

#include <cstdint>
#include <fstream>
#include <queue>
#include <string>
#include <vector>

// defined by the driver (darwin.cpp:41-43,65-69), read here like gact.cpp does
extern int NUM_BLOCKS;
extern int THREADS_PER_BLOCK;
extern int BATCH_SIZE;
extern int tile_size;
extern int tile_overlap;
extern int first_tile_score_threshold;

// one seed hit waiting to be extended (reference gact.h:35-46).  Only
// ref_id, query_id, ref_pos and query_pos are inputs; the chain state the
// reference keeps in the other fields lives on the device here.
typedef struct {
    int ref_id;
    int query_id;
    int ref_pos;
    int query_pos;
    int ref_bpos;
    int query_bpos;
    int score;
    int first_tile_score;
    char first;
    char reverse;
} GACT_call;

// Per feeder thread handle (reference gact.h:51-67 is a bag of 14 device
// pointers + a stream).  darwin.cpp only copies it (darwin.cpp:610-611,625),
// so it is reduced to what the engine needs: which engine, which slot.
typedef struct {
    void *engine;
    int slot;
    int reserved;
} GPU_storage;

// reference gact.cpp:48-228
void GACT(char *ref_str, char *query_str, int ref_length, int query_length,
          int tile_size, int tile_overlap, int ref_pos, int query_pos, int first_tile_score_threshold,
          int ref_id, int query_id, bool complement,
          int match_score, int mismatch_score, int gap_open, int gap_extend,
          std::ofstream &fout);

// reference gact.cpp:231-560
void GACT_Batch(std::vector<GACT_call> calls, int num_calls, bool complement, int offset, GPU_storage *s,
                int match_score, int mismatch_score, int gap_open, int gap_extend, std::ofstream &fout);

// reference cuda_host.cu:193-237 / 239-258
void GPU_init(int tile_size, int tile_overlap, int gap_open, int gap_extend, int match, int mismatch,
              int early_terminate, std::vector<GPU_storage> *s, int num_threads);
void GPU_close(std::vector<GPU_storage> *s, int num_threads);

// reference cuda_host.cu:23-190.  Returns a malloc'd int[BATCH][2*tile_size]
// the caller owns (the reference never frees it, gact.cpp:418): per tile
// [0] score [1] ref steps [2] query steps [3] max_i [4] max_j [5..] states, -1.
// reverses[t] == 1 means "towards position 0" as in gact.cpp:397-402.
int *Align_Batch_GPU(std::vector<std::string> ref_seqs, std::vector<std::string> query_seqs,
                     std::vector<int> ref_lens, std::vector<int> query_lens,
                     int *sub_mat, int gap_open, int gap_extend,
                     std::vector<int> ref_poss, std::vector<int> query_poss,
                     std::vector<char> reverses, std::vector<char> firsts,
                     int early_terminate, int tile_size, GPU_storage *s,
                     int num_blocks, int threads_per_block);

#endif
