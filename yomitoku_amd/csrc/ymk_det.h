// Declarations of the detection-transformer kernels (ymk_det.hip).
#pragma once
#include "ymk_common.h"

namespace ymk {

// geometry of the multi-scale token memory (3 levels), level-major rows
struct DetGeom {
  int B;          // images
  int h[3], w[3]; // level grids
  int hw[3];      // h*w
  int off[3];     // token offset of each level inside one image's 0..ntok-1 numbering
  int ntok;       // tokens per image (8400 at 640x640, 18900 at 960x960)
};

void add_bcast(hipStream_t s, const float* a, const float* b, int rows_mod, float* out, int M, int D);
void mask_rows(hipStream_t s, const float* in, const float* valid, float* out, const DetGeom& g, int D);
// keys_scratch: B * ntok words of device scratch
void topk_tokens(hipStream_t s, const float* logits, int nc, const DetGeom& g, int K, unsigned* keys_scratch, int* out_idx);
void gather_queries(hipStream_t s, const float* om, const float* bbox, const float* anchors, const int* idx,
                    const DetGeom& g, int K, int D, float* content, float* ref);
void refine_boxes(hipStream_t s, const float* delta, const float* ref, float* out, size_t n);
void deform_sample(hipStream_t s, const float* offs, const float* attw, const float* ref, const float* value, int ldv,
                   const DetGeom& g, int K, float* out);

}  // namespace ymk
