// Deterministic fp32 exp / sigmoid: the SAME operation sequence as oracle/np_ops.py det_expf
// (Cephes expf polynomial, every step one IEEE-754 binary32 operation, no fused multiply-add).
// Include only from translation units compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ float myolo_expf(float x)
{
    x = fminf(fmaxf(x, -86.0f), 88.0f);
    const float n = rintf(x * 1.44269504f);
    float r = x - n * 0.693359375f;
    r = r - n * -2.12194440e-4f;
    float p = 1.9875691500e-4f;
    p = p * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    const float y = (p * (r * r) + r) + 1.0f;
    const float scale = __int_as_float(((int)n + 127) << 23);
    return y * scale;
}

__device__ __forceinline__ float myolo_sigmoidf(float x)
{
    return 1.0f / (1.0f + myolo_expf(-x));
}
