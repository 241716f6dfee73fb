// probe_mfma4x4.hip -- discovers the operand / result lane layout of v_mfma_f64_4x4x4_4b_f64 by unit-vector probing
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_probe(unsigned long long *mask)
{
    const int l = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = l == la ? 1.0 : 0.0, b = l == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            const unsigned long long m = __ballot(d != 0.0);
            if (l == 0) mask[64 * la + lb] = m;
        }
}
int main()
{
    unsigned long long *dm; hipMalloc(&dm, 8 * 4096);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dm);
    std::vector<unsigned long long> m(4096);
    hipMemcpy(m.data(), dm, 8 * 4096, hipMemcpyDeviceToHost);
    for (int out = 0; out < 64; ++out) {
        printf("D lane %2d <-", out);
        for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) if (m[64 * la + lb] >> out & 1) printf(" (A%d,B%d)", la, lb);
        printf("\n");
    }
    return 0;
}
