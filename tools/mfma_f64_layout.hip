// Operand layout probe of v_mfma_f64_16x16x4_f64 on gfx950 (round 4):
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_f64_layout.hip -o /tmp/mfma_probe && /tmp/mfma_probe
// D(16 x 16) = A(16 x 4) B(4 x 16); prints which (lane, register) holds which element.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void probe(const double* A, const double* B, double* Dout, int amode, int bmode) {
    const int l = threadIdx.x;
    // candidate layouts: A[i][k] at lane i + 16 k (amode 0) or lane 4 i + k (amode 1)
    const int ai = amode == 0 ? (l & 15) : (l >> 2), ak = amode == 0 ? (l >> 4) : (l & 3);
    const int bj = bmode == 0 ? (l & 15) : (l >> 2), bk = bmode == 0 ? (l >> 4) : (l & 3);
    const double a = A[ai * 4 + ak], b = B[bk * 16 + bj];
    double4_t c = {0.0, 0.0, 0.0, 0.0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) Dout[l * 4 + r] = c[r];
}
int main() {
    double hA[64], hB[64], hD[256], ref[256];
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) hA[i * 4 + k] = 1.0 + i + 0.01 * k;
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) hB[k * 16 + j] = 2.0 + 0.5 * j + 0.001 * k * k;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double s = 0; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
    double *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    for (int amode = 0; amode < 2; ++amode) for (int bmode = 0; bmode < 2; ++bmode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, amode, bmode);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        // which D layout matches?  (a) row = 4 (l >> 4) + r, col = l & 15   (b) row = (l >> 4) + 4 r, col = l & 15
        int oka = 1, okb = 1;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            const double v = hD[l * 4 + r];
            if (fabs(v - ref[(4 * (l >> 4) + r) * 16 + (l & 15)]) > 1e-9) oka = 0;
            if (fabs(v - ref[((l >> 4) + 4 * r) * 16 + (l & 15)]) > 1e-9) okb = 0;
        }
        printf("A lane = %s, B lane = %s: D row = 4 (lane >> 4) + reg: %s;  D row = (lane >> 4) + 4 reg: %s\n",
               amode == 0 ? "i + 16 k" : "4 i + k", bmode == 0 ? "j + 16 k" : "4 j + k", oka ? "MATCH" : "no", okb ? "MATCH" : "no");
    }
    return 0;
}
