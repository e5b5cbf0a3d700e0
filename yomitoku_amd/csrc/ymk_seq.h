// Declarations of the token-sequence kernels (ymk_seq.hip).
#pragma once
#include "ymk_common.h"

namespace ymk {

void layernorm(hipStream_t s, const float* x, int ldx, int in_rows_mod, const float* g, const float* b, float eps,
               float* y, int ldy, int M, int D);
void add_pos_embed(hipStream_t s, float* x, const float* pos, int B, int gh, int gw, int full_gw, int D);

// Ragged batches: device tables of per-sample row offsets and lengths.  A non-null (koff, klen) pair replaces the key /
// value batch stride and Lk; a non-null (qoff, qlen) pair the query / output batch stride and Lq (then Lq / Lk passed
// to the launcher are the maxima over the batch: they size the grid).
struct SeqTab {
  const int *qoff = nullptr, *qlen = nullptr, *koff = nullptr, *klen = nullptr;
};

// O[b, q, h*hd:(h+1)*hd] = softmax(scale * Q K^T) V   (no mask; keys 0..Lk-1)
void flash_attention(hipStream_t s, const float* q, const float* k, const float* v, float* o, int B, int H, int Lq, int Lk,
                     int hd, int ldq, int ldk, int ldv, int ldo, long bsq, long bsk, long bsv, long bso, float scale,
                     const SeqTab* tab = nullptr, const unsigned* amax_q = nullptr, const unsigned* amax_k = nullptr,
                     const unsigned* amax_v = nullptr);
// (amax_q / amax_k / amax_v: max|x| records - ymk_common.h - bounding q, k and v: with all three, and the fp16 split in force for the
// calling thread's convolutions, both products of the attention run as fp16-split MFMAs too; otherwise exact fp32)
// same contract plus boolean masks (non-zero = blocked), for few queries / short key lists
void small_attention(hipStream_t s, const float* q, const float* k, const float* v, float* o, int B, int H, int Lq, int Lk,
                     int hd, int ldq, int ldk, int ldv, int ldo, long bsq, long bsk, long bsv, long bso, float scale,
                     const unsigned char* mask_qk, int ld_mask, const unsigned char* kpm, int ld_kpm,
                     const SeqTab* tab = nullptr);
// decode state at the start of a forward: tok[b][:] = pad, tok[b][0] = bos, state[b] = {0, 0, -1, 0}
void init_decode(hipStream_t s, int* tok, int ld_tok, int* state, int bos_id, int pad_id, int B);

void ctx_embed_ln(hipStream_t s, const int* tok, int ld_tok, int pos0, int npos, const float* emb, const float* posq,
                  const float* g, const float* be, float eps, float* out, int out_rows, int D, int B);
void greedy_step(hipStream_t s, const float* logits, long ld_b, int C, int step, int num_steps, int* tok, int* raw,
                 int ld_tok, int* state, int eos_id, int rep_on, int period_max, int min_run_p1, int min_repeats,
                 int* not_done, const int* prev_not_done, int* arrived, int* host_flag, int B, const int* gid = nullptr,
                 int* gopen = nullptr, int ng = 1, int partials = 0);  // partials: `logits` holds C (max, column) pairs per row
// host_flag of greedy_step may be null: publish_open then reports the step from a launch of its own (what the recogniser does)
void publish_open(hipStream_t s, const int* not_done, const int* prev_not_done, int* host_flag);
void refine_prep(hipStream_t s, const int* raw, int ld_tok, int S, int bos_id, int eos_id, int* tok2, unsigned char* kpm,
                 int B, const int* gid = nullptr, const int* gsteps = nullptr);
void row_argmax(hipStream_t s, const float* logits, int rows, int C, int* out);
void rep_cut(hipStream_t s, float* logits, long ld_b, int C, int S, const int* state, int eos_id, int B);
void row_maxprob(hipStream_t s, const float* logits, int rows, int C, int* ids, float* probs);
void fill_i32(hipStream_t s, int* p, int v, size_t n);

}  // namespace ymk
