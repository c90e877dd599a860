// Operand / result layout of v_mfma_f32_16x16x4_f32 on gfx950, checked against a host product:
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma16_layout.hip -o /tmp/mfma16 && /tmp/mfma16
// hypothesis: A lane l = A[i = l & 15][k = l >> 4], B lane l = B[k = l >> 4][j = l & 15], D reg r of lane l = D[4 (l >> 4) + r][l & 15]
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* D) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
}
int main() {
    float hA[64], hB[64], hD[256], ref[256];
    for (int i = 0; i < 64; ++i) { hA[i] = (float)((i * 7) % 11) - 5.f; hB[i] = (float)((i * 5) % 13) - 6.f; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int kk = 0; kk < 4; ++kk) s += hA[i * 4 + kk] * hB[kk * 16 + j]; ref[i * 16 + j] = s; }
    float *dA, *dB, *dD;
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 1024);
    hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
    printf("mismatches: %d\n", bad);
    return bad != 0;
}
