// Issue-rate probe for gfx950: v_mfma_f64_16x16x4_f64 vs v_fma_f64, independent and dependent chains,
// 1..4 wavefronts per SIMD.  Prints cycles per instruction per SIMD (s_memtime based) and chip TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ void mfma_kernel(double* out, int iters, long long* cyc) {
    d4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = {0.0, 0.0, 0.0, 0.0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    __syncthreads();
    long long t1 = __builtin_readcyclecounter();
    double s = 0.0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int CHAINS>
__global__ void fma_kernel(double* out, int iters, long long* cyc) {
    double acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = c;
    double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-4;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = fma(acc[c], a, b);
    }
    __syncthreads();
    long long t1 = __builtin_readcyclecounter();
    double s = 0.0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <typename K>
void run(const char* name, K kern, int chains, int threads, int blocks, double flops_per_inst) {
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * threads * blocks);
    hipMalloc(&cyc, sizeof(long long));
    const int iters = 20000;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, 100, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, sizeof c, hipMemcpyDeviceToHost);
    const int waves_per_simd = threads / 64 / 4 > 0 ? threads / 64 / 4 : 1;
    const double inst_per_simd = (double)iters * chains * (threads / 64 >= 4 ? waves_per_simd : 1);
    const double total_flops = (double)iters * chains * (threads / 64) * blocks * flops_per_inst;
    printf("%-28s chains %d threads %4d blocks %4d: %7.2f counter ticks / inst / SIMD, %8.3f ms, %7.2f TFLOP/s\n", name, chains, threads,
           blocks, (double)c / inst_per_simd, ms, total_flops / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int threads : {64, 256, 512, 1024}) {
        run("mfma_f64_16x16x4 dependent", mfma_kernel<1>, 1, threads, 256, 2048.0);
        run("mfma_f64_16x16x4 4 chains", mfma_kernel<4>, 4, threads, 256, 2048.0);
    }
    for (int threads : {256, 1024}) {
        run("v_fma_f64 dependent", fma_kernel<1>, 1, threads, 256, 128.0);
        run("v_fma_f64 8 chains", fma_kernel<8>, 8, threads, 256, 128.0);
    }
    run("mfma 4 chains, 2 WG/CU", mfma_kernel<4>, 4, 512, 512, 2048.0);
    return 0;
}
