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

// Do the fp64 matrix pipe and the fp64 vector ALU run side by side?  Every wave issues, per loop iteration, ONE
// v_mfma_f64_16x16x4_f64 (64 cycles of matrix pipe) and NF independent v_fma_f64 (4 cycles of vector ALU each).
// Overlap: max(64, 4 NF) cycles per iteration per SIMD; one shared fp64 datapath: 64 + 4 NF.
template <int NF>
__global__ void mix_kernel(double* out, int iters, long long* cyc) {
    d4 macc[2];
    macc[0] = {0.0, 0.0, 0.0, 0.0};
    macc[1] = {0.0, 0.0, 0.0, 0.0};
    double facc[NF > 0 ? NF : 1];
    for (int c = 0; c < NF; ++c) facc[c] = c;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4, fa = 1.0 + threadIdx.x * 1e-9;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i += 2) {
        macc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, macc[0], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < NF; ++c) facc[c] = fma(facc[c], fa, b);
        macc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, macc[1], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < NF; ++c) facc[c] = fma(facc[c], fa, b);
    }
    __syncthreads();
    long long t1 = __builtin_readcyclecounter();
    double s = macc[0][0] + macc[0][1] + macc[0][2] + macc[0][3] + macc[1][0] + macc[1][1] + macc[1][2] + macc[1][3];
    for (int c = 0; c < NF; ++c) s += facc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NF>
void run_mix(int threads) {
    double* out; long long* cyc;
    const int blocks = 256, iters = 20000;
    hipMalloc(&out, sizeof(double) * threads * blocks);
    hipMalloc(&cyc, sizeof(long long));
    hipLaunchKernelGGL(mix_kernel<NF>, dim3(blocks), dim3(threads), 0, 0, out, 100, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(mix_kernel<NF>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, sizeof c, hipMemcpyDeviceToHost);
    const int waves_per_simd = threads / 256;
    printf("mix: 1 mfma_f64_16x16x4 + %2d v_fma_f64 per iteration, %d waves/SIMD: %7.1f counter ticks per iteration per SIMD "
           "(matrix alone %d, vector alone %d, sum %d), %8.3f ms\n", NF, waves_per_simd,
           (double)c / iters / waves_per_simd,
           64, 4 * NF, 64 + 4 * NF, ms);
    hipFree(out); hipFree(cyc);
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
    for (int threads : {256, 1024}) {
        run_mix<0>(threads);
        run_mix<4>(threads);
        run_mix<8>(threads);
        run_mix<16>(threads);
        run_mix<32>(threads);
    }
    return 0;
}
