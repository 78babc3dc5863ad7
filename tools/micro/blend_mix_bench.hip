// Micro-benchmark (measurement aid): what would more waves per SIMD buy the forward blend?  A workgroup of four waves runs a loop whose
// body has the instruction mix of blend_range's steady state (csrc/gsr_hip.hip; tools/loop_isa.py: ~59 packed-fp32, ~78 plain VALU, 8
// v_exp_f32, ~20 ds_read_b128, one ds_write, one barrier per iteration = one 32-entry batch of one tile), on synthetic records in LDS.
// W such workgroups are resident per CU (W = 1 .. 8: W waves per SIMD); reported: the time one workgroup needs per iteration and the
// CU's rate in workgroup-iterations per microsecond.  The tile launch runs FOUR tiles per CU (W = 4); a six-wave tile workgroup would
// put six waves on a SIMD for the same arithmetic per CU (W = 6 here, per wave-iteration).
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o blend_mix_bench blend_mix_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

// DUTY 0: the bare loop.  1: the wave with (iteration & 3) == wave also does the staging wave's share of the real loop - ~35 VALU and ten
// LDS stores (the records of a later batch moved to the exp2 domain and parked) - and everybody meets it at the barrier, as in
// blend_range.  2: the same 35 instructions done by a FIFTH wave that blends nothing (a staging wave; the workgroup then has five).
template <int DUTY>
__global__ __launch_bounds__(DUTY == 2 ? 320 : 256) void k_mix(float* out, int iters) {
  __shared__ float4 sXY[4][16], sAB[4][16], sCO[4][16], sRG[4][16], sBE[4][16];
  __shared__ float sP[2][4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < 64) {
    for (int b = 0; b < 4; ++b) {
      const int k = tid & 15;
      sXY[b][k] = make_float4(3.f + 0.1f * k, 3.2f + 0.1f * k, 4.f - 0.1f * k, 4.1f - 0.1f * k);
      sAB[b][k] = make_float4(-0.05f, -0.06f, 0.01f, 0.012f);
      sCO[b][k] = make_float4(-0.04f, -0.045f, 0.3f + 0.01f * k, 0.35f);
      sRG[b][k] = make_float4(0.5f, 0.6f, 0.4f, 0.3f);
      sBE[b][k] = make_float4(0.2f, 0.1f, 0.f, 0.f);
    }
  }
  for (int k = tid; k < 512; k += 256) (&sP[0][0][0])[k] = 0.999f;
  __syncthreads();
  const float pxf = (float)(lane & 7), pyf = (float)(lane >> 3);
  const f2 px2 = {pxf, pxf}, py2 = {pyf, pyf};
  f2 al[4], om[4];
  for (int u = 0; u < 4; ++u) { al[u] = f2{0.01f, 0.02f}; om[u] = f2{0.99f, 0.98f}; }
  f2 CR = {0, 0}, CG = {0, 0}, CB = {0, 0};
  float Tb = 1.f, Tmin = 1.f;
  unsigned last = 0;
  const int e0 = (wave & 3) * 8;
  float d0 = pxf * 0.01f, d1 = pyf * 0.02f, d2 = 0.3f, d3 = 0.4f, d4 = 0.5f;
  for (int i = 0; i < iters; ++i) {
    if (DUTY && (DUTY == 2 ? wave == 4 : wave == (i & 3))) {  // the staging share: ~35 dependent-ish VALU + 10 LDS stores
#pragma unroll
      for (int r = 0; r < 7; ++r) {
        d0 = __builtin_fmaf(d0, 0.999f, d1); d1 = d1 * 0.998f + d2; d2 = __builtin_fmaf(d2, d0, 0.001f); d3 = d3 * d0; d4 = __builtin_fmaf(d4, 0.5f, d3);
      }
      if (lane < 32) {
        float* raw = reinterpret_cast<float*>(&sRG[(i + 2) & 3][0]);
        const int o = (lane >> 1) * 4 + (lane & 1);
        raw[o] = 0.5f + 1e-9f * d0; raw[o + 2] = 0.4f + 1e-9f * d1;
        float* raw2 = reinterpret_cast<float*>(&sBE[(i + 2) & 3][0]);
        raw2[o] = 0.2f + 1e-9f * d2; raw2[o + 2] = 1e-9f * d3;
        float* raw3 = reinterpret_cast<float*>(&sCO[(i + 2) & 3][0]);
        raw3[o + 2] = 0.3f + 1e-9f * d4;
      }
    }
    if (DUTY == 2 && wave == 4) { __syncthreads(); continue; }
    // stage A: the four segment products, the chain, the weights, the colour sums
    const float P0 = sP[i & 1][0][lane], P1 = sP[i & 1][1][lane], P2 = sP[i & 1][2][lane], P3 = sP[i & 1][3][lane];
    const float t1 = Tb * P0, t2 = t1 * P1, t3 = t2 * P2, t4 = t3 * P3;
    const float Tf = wave == 0 ? Tb : wave == 1 ? t1 : wave == 2 ? t2 : t3;
    const bool alive_in = !(Tf < 0.0001f);
    float T = alive_in ? Tf : 0.f;
    float Tn[8];
    float Tp = T;
#pragma unroll
    for (int u = 0; u < 8; u += 2) { Tn[u] = Tp * om[u >> 1].x; Tn[u + 1] = Tn[u] * om[u >> 1].y; Tp = Tn[u + 1]; }
    const bool alive_out = !(Tp < 0.0001f);
    f2 w[4];
    float Tq = T;
#pragma unroll
    for (int u = 0; u < 8; u += 2) { w[u >> 1] = al[u >> 1] * f2{Tq, Tn[u]}; Tq = Tn[u + 1]; }
    Tmin = alive_out ? Tp : Tmin;
    last += alive_out ? 8u : 0u;
    const int cb = i & 3;
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
      const float4 rg = sRG[cb][(e0 + u) >> 1], be = sBE[cb][(e0 + u) >> 1];
      CR = __builtin_elementwise_fma(f2{rg.x, rg.y}, w[u >> 1], CR);
      CG = __builtin_elementwise_fma(f2{rg.z, rg.w}, w[u >> 1], CG);
      CB = __builtin_elementwise_fma(f2{be.x, be.y}, w[u >> 1], CB);
    }
    asm volatile("" : "+v"(CR), "+v"(CG), "+v"(CB), "+v"(Tmin), "+v"(last));
    // stage E of the next batch
    const int gb = (i + 1) & 3;
    f2 Pp = {1.f, 1.f};
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
      const float4 xy = sXY[gb][(e0 + u) >> 1], ab = sAB[gb][(e0 + u) >> 1], co = sCO[gb][(e0 + u) >> 1];
      const f2 dx = f2{xy.x, xy.y} - px2, dy = f2{xy.z, xy.w} - py2;
      const f2 a2 = {ab.x, ab.y}, b2 = {ab.z, ab.w}, c2 = {co.x, co.y}, o = {co.z, co.w};
      const f2 t = __builtin_elementwise_fma(b2, dy, a2 * dx);
      const f2 p2 = __builtin_elementwise_fma(c2 * dy, dy, t * dx);
      const f2 ao = o * f2{__builtin_amdgcn_exp2f(p2.x), __builtin_amdgcn_exp2f(p2.y)};
      const float alpha0 = fminf(0.99f, ao.x), alpha1 = fminf(0.99f, ao.y);
      const bool keep0 = !(p2.x > 0.f) && !(alpha0 < 1.0f / 255.0f), keep1 = !(p2.y > 0.f) && !(alpha1 < 1.0f / 255.0f);
      al[u >> 1] = f2{keep0 ? alpha0 : 0.f, keep1 ? alpha1 : 0.f};
      om[u >> 1] = f2{1.f, 1.f} - al[u >> 1];
      Pp *= om[u >> 1];
    }
    sP[(i + 1) & 1][wave][lane] = fmaxf(Pp.x * Pp.y, 0.9995f);  // (kept near 1: the loop must not die)
    Tb = fmaxf(t4, 0.5f);
    __syncthreads();
  }
  out[blockIdx.x * 320 + tid] = CR.x + CR.y + CG.x + CG.y + CB.x + CB.y + Tmin + (float)last + Tb + d0 + d4;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 8 * 320 * 4);
  const int iters = 20000;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  printf("blend-loop instruction mix, 256 CUs, W four-wave workgroups resident per CU (= W waves per SIMD), %d iterations each\n", iters);
  auto sweep = [&](auto kernel, int threads, const char* what) {
    printf("-- %s\n", what);
    for (int W = 1; W <= 6; ++W) {
      hipLaunchKernelGGL(kernel, dim3(256 * W), dim3(threads), 0, 0, out, 2000);
      hipDeviceSynchronize();
      float best = 1e9f;
      for (int r = 0; r < 3; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL(kernel, dim3(256 * W), dim3(threads), 0, 0, out, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
      }
      const double us_per_iter = best * 1e3 / iters;  // one workgroup's iteration with W on the CU
      printf("W = %d: %.3f us per workgroup-iteration, %.2f workgroup-iterations per us and CU (%.3f us / W)\n", W, us_per_iter, W / us_per_iter, us_per_iter / W);
    }
  };
  sweep(k_mix<0>, 256, "bare loop (four waves)");
  sweep(k_mix<1>, 256, "+ the staging share on the wave whose turn it is (four waves, as blend_range)");
  sweep(k_mix<2>, 320, "+ the staging share on a fifth wave that blends nothing");
  return 0;
}
