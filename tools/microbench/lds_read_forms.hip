// lds_read_forms.hip -- what the LDS read FORM costs on gfx950 for the two access patterns of the pairwise kernels
// (tools/microbench; built by __graft_entry__.build(), run on the GPU box: profiles/r04_lds_read_forms.txt).
//
//   pattern "record": every lane of a wavefront reads the same row record (a broadcast), 5 doubles at a 40-byte stride
//     (rollout_kernel.h, DP = 3) -- as the compiler emits it (pairs of 8-byte reads merged into ds_read2_b64), as five
//     separate ds_read_b64, and as three ds_read_b128 of a record padded to 6 doubles (16-byte aligned);
//   pattern "mfma_a": lane l reads component 4 q + (l >> 4), q = 0..3, of row (l & 15) of a stage with an 18-double row
//     stride (the A operand of v_mfma_f64_16x16x4_f64 in rollout_stream_kernel.h) -- merged (ds_read2_b64, whose two
//     accesses are banked mod 32 over 16-lane groups: rows r and r + 8 collide) and as four separate ds_read_b64.
// 1024 threads per workgroup, one workgroup per CU, no arithmetic besides one add per value read.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef const __attribute__((address_space(3))) double* lds_cptr;
typedef double double2_t __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) double2_t* lds_c2ptr;

__device__ inline double lds_b64(const double* p) {
    lds_cptr q = (lds_cptr)p;
    asm volatile("" : "+v"(q));            // an opaque base register: SILoadStoreOptimizer cannot pair this read with another
    return *q;
}

template <int FORM>
__global__ __launch_bounds__(1024) void record_kernel(double* out, int iters, int rows) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int RS = FORM == 2 ? 6 : 5;
    for (int i = threadIdx.x; i < rows * RS; i += 1024) smem[i] = 1e-3 * i;
    __syncthreads();
    double acc[5] = {0, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        const double* rec = smem;
        for (int r = 0; r < rows; r += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const double* p = rec + u * RS;
                if (FORM == 0) {
#pragma unroll
                    for (int k = 0; k < 5; ++k) acc[k] += p[k];
                } else if (FORM == 1) {
#pragma unroll
                    for (int k = 0; k < 5; ++k) acc[k] += lds_b64(p + k);
                } else {
                    const double2_t* p2 = reinterpret_cast<const double2_t*>(__builtin_assume_aligned(p, 16));
                    const double2_t a = p2[0], b = p2[1], c = p2[2];
                    acc[0] += a.x; acc[1] += a.y; acc[2] += b.x; acc[3] += b.y; acc[4] += c.x;
                }
            }
            rec += 2 * RS;
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + acc[4];
}

template <int FORM>
__global__ __launch_bounds__(1024) void mfma_a_kernel(double* out, int iters, int tiles) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int RS = 18;
    for (int i = threadIdx.x; i < tiles * 16 * RS; i += 1024) smem[i] = 1e-3 * i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    double acc[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        for (int t = 0; t < tiles; ++t) {
            const double* a0p = smem + (size_t)(16 * t + (lane & 15)) * RS + 2 + (lane >> 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += (FORM == 0) ? a0p[4 * q] : lds_b64(a0p + 4 * q);
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <typename F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5.0;
}

int main() {
    int ncu = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) ncu = prop.multiProcessorCount;
    double* out;
    hipMalloc(&out, (size_t)ncu * 1024 * sizeof(double));
    const int iters = 200, rows = 256, tiles = 32;
    const double ghz = 2.4;
    const char* rn[3] = {"compiler's form (ds_read2_b64 pairs + ds_read_b64)", "five separate ds_read_b64", "three ds_read_b128 (record padded to 6 doubles)"};
    printf("# pattern record: 5-double row record broadcast to all lanes, 16 wavefronts per CU, %d CUs\n", ncu);
    for (int f = 0; f < 3; ++f) {
        double ms = 0;
        const size_t lds = (size_t)rows * 6 * 8;
        if (f == 0) ms = time_ms([&] { hipLaunchKernelGGL(record_kernel<0>, dim3(ncu), dim3(1024), lds, 0, out, iters, rows); });
        if (f == 1) ms = time_ms([&] { hipLaunchKernelGGL(record_kernel<1>, dim3(ncu), dim3(1024), lds, 0, out, iters, rows); });
        if (f == 2) ms = time_ms([&] { hipLaunchKernelGGL(record_kernel<2>, dim3(ncu), dim3(1024), lds, 0, out, iters, rows); });
        const double wave_rows = 16.0 * iters * rows;                // per CU
        printf("record  %-60s %8.3f ms  = %6.1f cycles per (wavefront, row) at %.1f GHz\n", rn[f], ms, ms * 1e-3 * ghz * 1e9 / wave_rows, ghz);
    }
    const char* mn[2] = {"compiler's form (ds_read2_b64)", "four separate ds_read_b64"};
    printf("# pattern mfma_a: A operand of the f64 16x16x4 matrix instruction from an 18-double-stride stage, 16 wavefronts per CU\n");
    for (int f = 0; f < 2; ++f) {
        double ms = 0;
        const size_t lds = (size_t)tiles * 16 * 18 * 8;
        if (f == 0) ms = time_ms([&] { hipLaunchKernelGGL(mfma_a_kernel<0>, dim3(ncu), dim3(1024), lds, 0, out, iters * 8, tiles); });
        if (f == 1) ms = time_ms([&] { hipLaunchKernelGGL(mfma_a_kernel<1>, dim3(ncu), dim3(1024), lds, 0, out, iters * 8, tiles); });
        const double wave_tiles = 16.0 * iters * 8 * tiles;
        printf("mfma_a  %-60s %8.3f ms  = %6.1f cycles per (wavefront, 16-row tile: 4 values per lane) at %.1f GHz\n", mn[f], ms,
               ms * 1e-3 * ghz * 1e9 / wave_tiles, ghz);
    }
    hipFree(out);
    return 0;
}
