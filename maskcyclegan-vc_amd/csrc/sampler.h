// On-device minibatch sampler (sampler_kernels.hip): utterance bank -> (real_A, mask_A, real_B, mask_B).
#pragma once
#include <hip/hip_runtime.h>

#define MCVC_SAMPLER_BINS 80

struct DrawArgs {
    const float* bank[2];          // side 0 = speaker A, 1 = speaker B: [80][ld] fp32, utterances side by side along the frame axis
    const int* offs[2];            // [n+1] first frame of utterance u in the bank; every utterance has >= T frames (checked on the host)
    long long ld[2];               // total frames of the bank (row pitch)
    int n[2];
    int B, T, max_mask_len;
    unsigned long long seed, step;
    float* real[2];                // [B][80][T]
    float* mask[2];                // [B][80][T]
    int* draws;                    // nullable [B][2][4] = (utterance, crop lo, mask size, mask start)
};

int mcvc_draw_batch_launch(const DrawArgs& a, hipStream_t s);
