// ubench_stream.hip -- how fast can the record region of the Schur kernels be read at THEIR occupancy, and what does their
// access pattern cost?  k_schur_gram reads 816 B per view (84-double W record + 18-double E record) with one 16-byte load per
// lane and instruction at a stride of 48 bytes inside a record (lane = (column, board): three instructions cover a column),
// everything requested up front (48 loads per wave in flight), two 256-thread workgroups per CU.  Kernels, all over the same
// buffer, same workgroup count, same LDS footprint (occupancy 2 per CU), each lane accumulating what it loads:
//   flat     : lane i of a workgroup reads bytes [16 i, 16 i + 16) of consecutive 4 KB blocks of its chunk -- every wave
//              instruction 1024 consecutive bytes; LOADS per thread in flight before the first use
//   columns  : the W pattern of k_schur_gram (lane (a, kq): three 16-byte loads 16 bytes apart at 48-byte stride between lanes)
// Usage: ubench_stream [MB] [workgroups]      build: hipcc --offload-arch=gfx950 -O3 ubench_stream.hip -o ubench_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));

template <int LOADS>
__global__ __launch_bounds__(256, 2) void k_flat(const d2 *buf, size_t per_wg_d2, double *out)
{
    extern __shared__ double lds[];
    const d2 *p = buf + (size_t)blockIdx.x * per_wg_d2;
    double s = 0.0;
    for (size_t base = 0; base < per_wg_d2; base += (size_t)256 * LOADS) {
        d2 v[LOADS];
#pragma unroll
        for (int j = 0; j < LOADS; ++j) { const size_t e = base + threadIdx.x + 256 * j; v[j] = p[e < per_wg_d2 ? e : per_wg_d2 - 1]; }
#pragma unroll
        for (int j = 0; j < LOADS; ++j) s += v[j][0] + v[j][1];
    }
    if (s == 12345.678) lds[threadIdx.x] = s;
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

// W records of 84 doubles; a workgroup owns `views` consecutive records; wave w, group g: lane (a = lane & 15, kq = lane >> 4)
// reads column a (6 doubles = 3 x 16 B) of record 16 w + 4 g + kq (+ 64 per pass)
template <int GROUPS>
__global__ __launch_bounds__(256, 2) void k_columns(const double *buf, int views, double *out)
{
    extern __shared__ double lds[];
    const double *p = buf + (size_t)blockIdx.x * views * 84;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, a = lane & 15, kq = lane >> 4;
    double s = 0.0;
    for (int base = 0; base < views; base += 16 * GROUPS) {
        d2 v[GROUPS][3];
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            const int r = base + 4 * GROUPS * wave + 4 * g + kq;
            const double *q = p + (size_t)84 * (r < views ? r : views - 1) + 6 * (a < 14 ? a : 13);
#pragma unroll
            for (int k = 0; k < 3; ++k) v[g][k] = *reinterpret_cast<const d2 *>(q + 2 * k);
        }
#pragma unroll
        for (int g = 0; g < GROUPS; ++g)
#pragma unroll
            for (int k = 0; k < 3; ++k) s += v[g][k][0] + v[g][k][1];
    }
    if (s == 12345.678) lds[threadIdx.x] = s;
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
static double time_us(F launch, int reps)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); launch();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return 1e3 * ms / reps;
}

int main(int argc, char **argv)
{
    const size_t mb = argc > 1 ? (size_t)atol(argv[1]) : 180;
    const int lds_bytes = 75 * 1024;
    double *buf, *out, *other;
    const size_t bytes = mb << 20;
    (void)hipMalloc(&buf, bytes); (void)hipMalloc(&other, (size_t)512 << 20); (void)hipMalloc(&out, sizeof(double) * 256 * 8192);
    (void)hipMemset(buf, 0, bytes); (void)hipMemset(other, 0, (size_t)512 << 20);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_flat<6>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_flat<12>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_flat<24>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_columns<4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_columns<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    // between the timed launches 512 MB of other data are streamed, so that the buffer comes from HBM, not from the 256 MB cache
    auto flush = [&]() { k_flat<12><<<2048, 256, lds_bytes>>>(reinterpret_cast<const d2 *>(other), ((size_t)512 << 20) / 16 / 2048, out); };
    for (int wgs : { 512, 1024, 1536, 2560 }) {
        const size_t per_wg_d2 = bytes / 16 / wgs;
        const int views = (int)(bytes / 8 / 84 / wgs);
        struct { const char *name; double us, us_cold; } r[5];
        int k = 0;
        auto run = [&](const char *name, auto call) {
            r[k].name = name;
            r[k].us = time_us(call, 20);
            const double t0 = time_us([&]() { flush(); }, 5);
            r[k].us_cold = time_us([&]() { flush(); call(); }, 5) - t0;
            ++k;
        };
        const d2 *b2 = reinterpret_cast<const d2 *>(buf);
        run("flat, 6 x 16 B in flight", [&]() { k_flat<6><<<wgs, 256, lds_bytes>>>(b2, per_wg_d2, out); });
        run("flat, 12 x 16 B in flight", [&]() { k_flat<12><<<wgs, 256, lds_bytes>>>(b2, per_wg_d2, out); });
        run("flat, 24 x 16 B in flight", [&]() { k_flat<24><<<wgs, 256, lds_bytes>>>(b2, per_wg_d2, out); });
        run("columns (k_schur_gram's W pattern), 4 groups", [&]() { k_columns<4><<<wgs, 256, lds_bytes>>>(buf, views, out); });
        run("columns, 2 groups", [&]() { k_columns<2><<<wgs, 256, lds_bytes>>>(buf, views, out); });
        for (int i = 0; i < k; ++i)
            printf("%4zu MB, %4d workgroups (2 per CU)  %-46s %7.1f us  %6.2f TB/s   behind 512 MB of other traffic: %7.1f us  %6.2f TB/s\n", mb, wgs, r[i].name, r[i].us,
                   bytes / r[i].us * 1e-6, r[i].us_cold, bytes / r[i].us_cold * 1e-6);
    }
    return 0;
}
