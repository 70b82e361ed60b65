// Per-element arithmetic of the point-wise flows (Sigmoid / Logit sigmoid.py:9-56, ELU / LeakyReLU activations.py:11-101),
// shared by the stand-alone sx_pointwise kernels and the SX_STEP_POINTWISE step of the fused flow kernel.
#pragma once
#include "sx_common.h"

#define PW_TINY 1.17549435e-38f          // torch.finfo(float32).tiny
#define PW_ONE_MINUS_EPS 0.99999988f     // 1 - torch.finfo(float32).eps

// exp(v) on v_exp_f32 with the product v*log2(e) carried to double-float accuracy (error ~1e-7 relative for |v| < 88)
__device__ __forceinline__ float pw_exp(float v) {
    const float t = v * 1.44269504088896341f;
    const float r = __builtin_fmaf(v, 1.44269504088896341f, -t) + v * 1.92596299e-8f;   // low part of v*log2(e)
    return __builtin_amdgcn_exp2f(t) * (1.f + r * 0.69314718055994531f);
}
__device__ __forceinline__ float pw_log(float v) { return __builtin_amdgcn_logf(v) * 0.69314718055994531f; }
// softplus(-v) + softplus(v) = |v| + 2 log(1 + exp(-|v|)): one exp, one log (sigmoid.py:44 evaluates two softplus)
__device__ __forceinline__ float pw_two_softplus(float v, float &t) {
    const float a = fabsf(v);
    t = pw_exp(-a);
    return a + 2.f * pw_log(1.f + t);
}

__device__ __forceinline__ void pw_eval(int kind, float param, float log_slope, float x, float &out, float &ld) {
    switch (kind) {
        case SX_PW_SIGMOID: {                                          // sigmoid.py:18-23, 41-44
            float t;
            ld = -pw_two_softplus(x, t);
            const float r = __builtin_amdgcn_rcpf(1.f + t);            // sigmoid(|x|); sigmoid(-|x|) = t * r
            out = fminf(fmaxf(x >= 0.f ? r : t * r, PW_TINY), PW_ONE_MINUS_EPS);
            break;
        }
        case SX_PW_LOGIT: {                                            // sigmoid.py:25-31; minus the forward log-derivative at out
            const float y = fminf(fmaxf(x, PW_TINY), PW_ONE_MINUS_EPS);
            out = logf(y) - log1pf(-y);
            float t;
            ld = pw_two_softplus(out, t);
            break;
        }
        case SX_PW_ELU: {                                              // activations.py:22-27, 57-63
            out = x > 0.f ? x : expm1f(x);
            ld = -fmaxf(-x, 0.f);
            break;
        }
        case SX_PW_ELU_INV: {                                          // activations.py:29-37
            const float lt = log1pf(x);                                // NaN below -1, like torch.min(log1p(y), 0)
            out = fmaxf(x, 0.f) + ((lt < 0.f || lt != lt) ? lt : 0.f);
            ld = (out != out) ? out : fmaxf(-out, 0.f);
            break;
        }
        case SX_PW_LEAKY_RELU: {                                       // activations.py:80-86, param = slope, 94-101
            out = fmaxf(0.f, x) + param * fminf(0.f, x);
            ld = x >= 0.f ? 0.f : log_slope;                           // math.log(negative_slope), computed on the host
            break;
        }
        default: {                                                     // SX_PW_LEAKY_RELU_INV: param = 1 / slope
            out = fmaxf(0.f, x) + param * fminf(0.f, x);
            ld = out >= 0.f ? 0.f : -log_slope;
            break;
        }
    }
}

