// Fused PARSeq decoder step (ymk_decstep.hip).
#pragma once
#include "ymk_common.h"

namespace ymk {

// device pointers; *_t matrices are transposed nn.Linear weights: [in][out]
struct DecStepW {
  const float *emb, *posq, *qsa;
  const float *ncg, *ncb, *n1g, *n1b, *n2g, *n2b, *dng, *dnb;
  const float *Wkv_t, *bkv, *Wo1_t, *bo1, *Wq_t, *bq, *Wo2_t, *bo2, *W1_t, *b1, *W2_t, *b2;
  int D, H, F;
};

bool parseq_dec_step_supported(int D, int H, int F, int L, int NS);
// gid / gopen (grouped forward): row b belongs to mini-batch gid[b]; gopen[step][g] = rows of g still lacking an <eos>
// after that step.  A block whose mini-batch closed at an earlier step does nothing.
// L: encoder-memory rows per sample, or - with mem_off / mem_len (device, per sample: first row, row count) - the
// longest sample of a ragged batch
void parseq_dec_step(hipStream_t s, const DecStepW& W, const int* tok, int ld_tok, int step, float* skv, int NS,
                     const float* memkv, int L, const int* mem_off, const int* mem_len, float* out, const int* prev_not_done,
                     int B, const int* gid = nullptr, const int* gopen = nullptr, int ng = 1);

}  // namespace ymk
