// Development probe (run on the GPU box): operand byte layout and scale semantics of
// v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 e4m3 operands.  hipcc --offload-arch=gfx950 -O2 -o probe mfma_fp8_probe.hip && ./probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// OCP e4m3 encode of small values (exact for the test set)
static uint8_t enc(float v)
{
    if (v == 0.f) return 0;
    uint8_t s = v < 0 ? 0x80 : 0;
    float a = fabsf(v);
    int e = (int)floorf(log2f(a));
    float m = a / exp2f((float)e) - 1.0f;        // [0,1)
    int mi = (int)lrintf(m * 8.f);
    if (mi == 8) { mi = 0; e++; }
    int be = e + 7;
    if (be <= 0) { mi = (int)lrintf(a / exp2f(-6.f) * 8.f); return s | (uint8_t)mi; }
    return s | (uint8_t)(be << 3) | (uint8_t)mi;
}

__global__ void k(const uint8_t *A, const uint8_t *B, float *D, int layout, int sa, int sb, int opa, int opb)
{
    const int l = threadIdx.x, li = l & 31, hi = l >> 5;
    union { v8i v; uint8_t b[32]; } a, b;
    for (int p = 0; p < 32; p++) {
        int kk = layout == 0 ? 32 * hi + p : 16 * hi + (p & 15) + 32 * (p >> 4);
        a.b[p] = A[li * 64 + kk];          // A[i][k]
        b.b[p] = B[li * 64 + kk];          // B given as [n][k]
    }
    f32x16 c;
    for (int r = 0; r < 16; r++) c[r] = 0.f;
    if (opa == 0 && opb == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a.v, b.v, c, 0, 0, 0, sa, 0, sb);
    else if (opa == 1) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a.v, b.v, c, 0, 0, 1, sa, 0, sb);
    else if (opa == 2) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a.v, b.v, c, 0, 0, 2, sa, 0, sb);
    else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a.v, b.v, c, 0, 0, 3, sa, 0, sb);
    for (int r = 0; r < 16; r++) D[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + li] = c[r];     // row = A row i, col = B row n
}

int main()
{
    uint8_t hA[32 * 64], hB[32 * 64];
    float fA[32 * 64], fB[32 * 64], ref[32 * 32], hD[32 * 32];
    srand(1);
    const float vals[] = {0.f, 0.5f, 1.f, -1.f, 2.f, -0.25f, 1.5f, -3.f};
    for (int i = 0; i < 32 * 64; i++) {
        fA[i] = vals[rand() % 8]; fB[i] = vals[rand() % 8];
        hA[i] = enc(fA[i]); hB[i] = enc(fB[i]);
    }
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) {
            float s = 0;
            for (int kq = 0; kq < 64; kq++) s += fA[i * 64 + kq] * fB[j * 64 + kq];
            ref[i * 32 + j] = s;
        }
    uint8_t *dA, *dB; float *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    for (int layout = 0; layout < 2; layout++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, layout, 0x7f7f7f7f, 0x7f7f7f7f, 0, 0);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        double e = 0; for (int i = 0; i < 1024; i++) e = fmax(e, fabs(hD[i] - ref[i]));
        printf("layout %d scale 127/127: max err %.4f (D[0]=%.3f ref %.3f)\n", layout, e, hD[0], ref[0]);
    }
    // scale semantics on layout 0: byte 0 of scale_a = 128 (x2) vs other bytes
    const int sas[] = {0x7f7f7f80, 0x7f7f807f, 0x7f807f7f, (int)0x807f7f7f};
    for (int t = 0; t < 4; t++)
        for (int op = 0; op < 4; op++) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, 0, sas[t], 0x7f7f7f7f, op, 0);
            hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
            double r1 = 0, r2 = 0; for (int i = 0; i < 1024; i++) { r1 = fmax(r1, fabs(hD[i] - ref[i])); r2 = fmax(r2, fabs(hD[i] - 2 * ref[i])); }
            printf("scale_a byte%d=128 opsel_a=%d: err vs 1x %.3f  vs 2x %.3f\n", t, op, r1, r2);
        }
    return 0;
}
