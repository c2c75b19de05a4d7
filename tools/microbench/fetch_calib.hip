// fetch_calib.hip -- known-byte read kernels to calibrate rocprofv3's FETCH_SIZE on gfx950 (MI355X).
//
// The guide's gfx950 note (MI355X_MICROARCH.md, HBM) says FETCH_SIZE reports half the bytes of a 16 B/lane streaming read
// and leaves other widths uncalibrated.  The rollout kernels read their tables 16 B per lane (two adjacent columns,
// global_load_dwordx4) or 8 B per lane (one column), every line exactly once per pass and far beyond the 4 MiB L2, so
// these kernels do the same on a buffer whose size is printed: FETCH_SIZE (KiB) x 1024 / bytes is the factor to apply.
//   read8_stream / read16_stream : 1 GiB read once (beyond the 256 MiB Infinity Cache: HBM)
//   read8_mall  / read16_mall    : a 64 MiB buffer read 16 times (Infinity-Cache hits after the first pass, L2 misses)
// Build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip ; run under `rocprofv3 --pmc FETCH_SIZE`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// TAG only gives the stream / mall dispatches distinct kernel names in the profiler output
template <typename T, int TAG>
__global__ __launch_bounds__(256) void read_kernel(const T* __restrict__ src, size_t n, int passes, double* out) {
    double acc = 0.0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (int p = 0; p < passes; ++p)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            const T v = src[i];
            if constexpr (sizeof(T) == 16) acc += v.x + v.y; else acc += v;
        }
    if (acc == 12345.678) out[0] = acc;       // keeps the loads alive, never true for the zero buffer
}


template <typename T, int TAG>
static int run(const char* name, const void* buf, size_t bytes, int passes, double* out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((read_kernel<T, TAG>), dim3(256 * 8), dim3(256), 0, 0, (const T*)buf, bytes / sizeof(T), passes, out);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-14s %zu B/lane  buffer %zu bytes x %d passes = %zu bytes requested  %.3f ms  %.0f GB/s\n", name, sizeof(T), bytes,
           passes, bytes * passes, ms, bytes * (double)passes / ms / 1e6);
    return 0;
}

int main() {
    const size_t big = (size_t)1 << 30, small = (size_t)64 << 20;
    void* buf = nullptr;
    double* out = nullptr;
    CK(hipMalloc(&buf, big));
    CK(hipMalloc(&out, 8));
    CK(hipMemset(buf, 0, big));
    CK(hipDeviceSynchronize());
    // dispatch order = the order of the rows in the PMC summary: 8 B stream, 16 B stream, 8 B mall, 16 B mall
    if (run<double, 0>("read8_stream", buf, big, 1, out)) return 1;
    if (run<double2, 0>("read16_stream", buf, big, 1, out)) return 1;
    if (run<double, 1>("read8_mall", buf, small, 16, out)) return 1;
    if (run<double2, 1>("read16_mall", buf, small, 16, out)) return 1;
    CK(hipFree(buf)); CK(hipFree(out));
    return 0;
}
