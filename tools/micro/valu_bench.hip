// Micro-benchmark (measurement aid): lone-wave VALU issue/latency and the s_memtime tick rate on MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(float* out, unsigned long long* ticks, int iters, float a, float b) {
  float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  unsigned long long t0 = __builtin_readcyclecounter();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // 8 dependent fma
      x0 = __builtin_fmaf(x0, a, b); x0 = __builtin_fmaf(x0, a, b); x0 = __builtin_fmaf(x0, a, b); x0 = __builtin_fmaf(x0, a, b);
      x0 = __builtin_fmaf(x0, a, b); x0 = __builtin_fmaf(x0, a, b); x0 = __builtin_fmaf(x0, a, b); x0 = __builtin_fmaf(x0, a, b);
    } else if (MODE == 1) {  // 8 independent fma
      x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
      x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
    } else if (MODE == 2) {  // 8 independent exp2
      x0 = __builtin_amdgcn_exp2f(x0); x1 = __builtin_amdgcn_exp2f(x1); x2 = __builtin_amdgcn_exp2f(x2); x3 = __builtin_amdgcn_exp2f(x3);
      x4 = __builtin_amdgcn_exp2f(x4); x5 = __builtin_amdgcn_exp2f(x5); x6 = __builtin_amdgcn_exp2f(x6); x7 = __builtin_amdgcn_exp2f(x7);
    } else if (MODE == 3) {  // 8 independent cmp+cndmask pairs
      x0 = x0 < a ? x0 + b : x0; x1 = x1 < a ? x1 + b : x1; x2 = x2 < a ? x2 + b : x2; x3 = x3 < a ? x3 + b : x3;
      x4 = x4 < a ? x4 + b : x4; x5 = x5 < a ? x5 + b : x5; x6 = x6 < a ? x6 + b : x6; x7 = x7 < a ? x7 + b : x7;
    } else if (MODE == 5) {  // 8 independent packed fma (two fp32 lanes per register pair): 16 fma per iteration
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x0 + 8, x1 + 8}, p5 = {x2 + 8, x3 + 8}, p6 = {x4 + 8, x5 + 8}, p7 = {x6 + 8, x7 + 8};
      const f2 A = {a, a}, B = {b, b};
      for (int j = 0; j < iters; ++j) {
        p0 = __builtin_elementwise_fma(p0, A, B); p1 = __builtin_elementwise_fma(p1, A, B); p2 = __builtin_elementwise_fma(p2, A, B); p3 = __builtin_elementwise_fma(p3, A, B);
        p4 = __builtin_elementwise_fma(p4, A, B); p5 = __builtin_elementwise_fma(p5, A, B); p6 = __builtin_elementwise_fma(p6, A, B); p7 = __builtin_elementwise_fma(p7, A, B);
      }
      x0 = p0.x + p0.y + p4.x + p4.y; x1 = p1.x + p1.y + p5.x + p5.y; x2 = p2.x + p2.y + p6.x + p6.y; x3 = p3.x + p3.y + p7.x + p7.y;
      break;
    } else if (MODE == 6) {  // 8 independent packed mul
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x0 + 8, x1 + 8}, p5 = {x2 + 8, x3 + 8}, p6 = {x4 + 8, x5 + 8}, p7 = {x6 + 8, x7 + 8};
      const f2 A = {a, a};
      for (int j = 0; j < iters; ++j) { p0 *= A; p1 *= A; p2 *= A; p3 *= A; p4 *= A; p5 *= A; p6 *= A; p7 *= A; }
      x0 = p0.x + p0.y + p4.x + p4.y; x1 = p1.x + p1.y + p5.x + p5.y; x2 = p2.x + p2.y + p6.x + p6.y; x3 = p3.x + p3.y + p7.x + p7.y;
      break;
    } else {  // dependent chain through cmp -> select (mask crossing)
      x0 = x0 < a ? x0 * b : x0 + b; x0 = x0 < a ? x0 * b : x0 + b; x0 = x0 < a ? x0 * b : x0 + b; x0 = x0 < a ? x0 * b : x0 + b;
      x0 = x0 < a ? x0 * b : x0 + b; x0 = x0 < a ? x0 * b : x0 + b; x0 = x0 < a ? x0 * b : x0 + b; x0 = x0 < a ? x0 * b : x0 + b;
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = r1 - r0; }
}
template <int MODE>
void run(const char* name, int blocks, int threads, float* out, unsigned long long* ticks) {
  const int iters = 100000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, ticks, 1000, 0.999f, 0.001f);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, ticks, iters, 0.999f, 0.001f);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  unsigned long long h[2]; hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
  printf("%-28s blocks=%5d thr=%3d  %.3f ms  %.2f ns per op  memtime ticks/op %.2f  (memtime %.1f MHz, realtime %.1f MHz)\n", name, blocks, threads, ms,
         ms * 1e6 / (iters * 8.0), (double)h[0] / (iters * 8.0), h[0] / (ms * 1e3), h[1] / (ms * 1e3));
}
int main() {
  float* out; unsigned long long* ticks;
  hipMalloc(&out, 1 << 24); hipMalloc(&ticks, 64);
  for (int thr : {64, 512, 1024}) {  // one workgroup per CU x 4: 1, 2 x 4 ... waves per SIMD resident
    const int blocks = thr == 64 ? 1024 : 256;  // 64: one wave per SIMD; 512 / 1024: one workgroup per CU = 2 / 4 waves per SIMD
    run<0>("dependent fma", blocks, thr, out, ticks);
    run<1>("independent fma", blocks, thr, out, ticks);
    run<2>("independent exp2", blocks, thr, out, ticks);
    run<3>("independent cmp+cndmask+add", blocks, thr, out, ticks);
    run<5>("independent packed fma (per instruction)", blocks, thr, out, ticks);
    run<6>("independent packed mul (per instruction)", blocks, thr, out, ticks);
  }
  return 0;
}
