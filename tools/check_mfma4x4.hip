// check_mfma4x4.hip -- operand / result layout of v_mfma_f64_4x4x4_4b_f64 on gfx950 and whether a Gram tile built from
// ten 4x4 block pairs equals the v_mfma_f64_16x16x4_f64 tile bit for bit.  One wave.
// Layout (found with tools/probe_mfma4x4.hip, verified by this program): lane l: k = l / 16, block b = (l % 16) / 4;
// A operand A_b[i][k] with i = l % 4; B operand B_b[k][j] with j = l % 4; result D_b[i][j] in lane 16 i + 4 b + j.
// I.e. the operand of lane (c = l % 16, k = l / 16) is element (column c, row k) exactly like v_mfma_f64_16x16x4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(const double *A, const double *B, double *D)   // A, B: [4 blocks][4][4] row-major, D likewise
{
    const int l = threadIdx.x, b = (l % 16) / 4, x = l % 4, y = l / 16;
    const double a = A[16 * b + 4 * x + y];        // A_b[i = x][k = y]
    const double bb = B[16 * b + 4 * y + x];       // B_b[k = y][j = x]
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bb, 0.0, 0, 0, 0);
    D[16 * b + 4 * y + x] = d;                     // D_b[i = y][j = x]
}

// J: [rows][16] row-major, rows a multiple of 4.  G16: 16x16 tile through the 16x16x4 instruction; G4: the same entries
// through 4x4x4 block pairs q = 0, 1, 2 (A = natural vector N, B = N rotated by q column groups)
__global__ void k_gram(const double *J, int rows, double *G16, double *G4)
{
    const int l = threadIdx.x;
    {
        const int col = l & 15, kq = l >> 4;
        d4 acc = { 0, 0, 0, 0 };
        for (int t = 0; t < rows / 4; ++t) { const double a = J[16 * (4 * t + kq) + col]; acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0); }
        for (int r = 0; r < 4; ++r) G16[16 * (kq + 4 * r) + col] = acc[r];
    }
    {
        const int b = (l % 16) / 4, x = l % 4, y = l / 16;
        double acc[3] = { 0, 0, 0 };
        for (int t = 0; t < rows / 4; ++t) {
            const double n = J[16 * (4 * t + y) + 4 * b + x];
            for (int q = 0; q < 3; ++q) {
                const double r = J[16 * (4 * t + y) + 4 * ((b + q) % 4) + x];
                acc[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(n, r, acc[q], 0, 0, 0);
            }
        }
        // D_q,b[i = y][j = x] = G[4 b + y][4 ((b + q) % 4) + x]
        for (int q = 0; q < 3; ++q) G4[16 * (4 * b + y) + 4 * ((b + q) % 4) + x] = acc[q];
    }
}

int main()
{
    std::vector<double> A(64), B(64), D(64), ref(64);
    for (int i = 0; i < 64; ++i) { A[i] = 1.0 + 0.37 * i; B[i] = 2.0 - 0.11 * i * i; }
    for (int b = 0; b < 4; ++b) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        double s = 0; for (int k = 0; k < 4; ++k) s += A[16 * b + 4 * i + k] * B[16 * b + 4 * k + j];
        ref[16 * b + 4 * i + j] = s;
    }
    double *dA, *dB, *dD;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 512);
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 512, hipMemcpyDeviceToHost);
    double worst = 0; for (int i = 0; i < 64; ++i) worst = fmax(worst, fabs(D[i] - ref[i]) / fabs(ref[i]));
    printf("layout check: max relative difference %.3e %s\n", worst, worst < 1e-13 ? "(layout as assumed)" : "(LAYOUT DIFFERS)");

    const int rows = 56;
    std::vector<double> J(16 * rows), G16(256), G4(256, 0.0);
    unsigned s = 12345;
    for (auto &v : J) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 20001 - 10000) * 1.37e-3; }
    for (int r = 0; r < rows; ++r) J[16 * r + 15] = 0.0;
    double *dJ, *dG16, *dG4;
    hipMalloc(&dJ, sizeof(double) * J.size()); hipMalloc(&dG16, 2048); hipMalloc(&dG4, 2048);
    hipMemcpy(dJ, J.data(), sizeof(double) * J.size(), hipMemcpyHostToDevice);
    hipMemset(dG4, 0, 2048);
    hipLaunchKernelGGL(k_gram, dim3(1), dim3(64), 0, 0, dJ, rows, dG16, dG4);
    hipMemcpy(G16.data(), dG16, 2048, hipMemcpyDeviceToHost); hipMemcpy(G4.data(), dG4, 2048, hipMemcpyDeviceToHost);
    int same = 0, covered = 0; double wd = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        const int q = ((j / 4) - (i / 4) + 4) % 4;
        if (q == 3) continue;                      // pair (I, I + 3) is held as (I + 3, I): the transposed entry
        ++covered;
        if (memcmp(&G16[16 * i + j], &G4[16 * i + j], 8) == 0) ++same;
        wd = fmax(wd, fabs(G16[16 * i + j] - G4[16 * i + j]) / fmax(fabs(G16[16 * i + j]), 1e-300));
    }
    printf("gram: %d of %d covered entries bit-identical to the 16x16x4 tile, max relative difference %.3e\n", same, covered, wd);
    return 0;
}
