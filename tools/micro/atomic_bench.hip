// Micro-benchmark (measurement aid, not product code): throughput of global float/uint atomics on MI355X by scope,
// address pattern and lanes per instruction.  Build: hipcc --offload-arch=gfx950 -O3 atomic_bench.hip -o atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ inline uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// each wave does `iters` atomic instructions; `lanes` active lanes per instruction write `lanes` consecutive floats of a
// random record (record = 12 floats, like the screen-space accumulator), nrec records.
template <int SCOPE>
__global__ void k_rec(float* buf, uint32_t nrec, int iters, int lanes) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
    const uint32_t r = hash(wave * 7919u + it) % nrec;
    if (lane < lanes) {
      float* p = buf + (size_t)r * 12 + lane;
      if (SCOPE == 0) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (SCOPE == 1) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (SCOPE == 2) unsafeAtomicAdd(p, 1.0f);
      else *p += 1.0f;  // plain RMW (racy): upper bound
    }
  }
}
// every lane its own random uint counter (count-pass pattern)
template <int SCOPE>
__global__ void k_cnt(uint32_t* buf, uint32_t n, int iters) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    uint32_t* p = buf + hash(tid * 31u + it) % n;
    if (SCOPE == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

template <class F>
float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}

int main() {
  const uint32_t nrec = 300000;
  float* buf; hipMalloc(&buf, (size_t)nrec * 12 * 4); hipMemset(buf, 0, (size_t)nrec * 12 * 4);
  uint32_t* cb; hipMalloc(&cb, 65536 * 4); hipMemset(cb, 0, 65536 * 4);
  const int waves = 1024, iters = 650;  // like one backward blend: 1024 tiles x 650 splats
  for (int lanes : {1, 10, 16, 64}) {
    const double ops = (double)waves * iters * lanes;
    float t0 = timeit([&] { hipLaunchKernelGGL(k_rec<0>, dim3(waves), dim3(64), 0, 0, buf, nrec, iters, lanes); });
    float t1 = timeit([&] { hipLaunchKernelGGL(k_rec<1>, dim3(waves), dim3(64), 0, 0, buf, nrec, iters, lanes); });
    float t2 = timeit([&] { hipLaunchKernelGGL(k_rec<2>, dim3(waves), dim3(64), 0, 0, buf, nrec, iters, lanes); });
    float t3 = timeit([&] { hipLaunchKernelGGL(k_rec<3>, dim3(waves), dim3(64), 0, 0, buf, nrec, iters, lanes); });
    printf("rec lanes=%2d  agent %.1f us (%.1f Gop/s)  workgroup %.1f us (%.1f)  unsafe %.1f us (%.1f)  plainRMW %.1f us (%.1f)\n", lanes,
           t0 * 1e3, ops / t0 / 1e6, t1 * 1e3, ops / t1 / 1e6, t2 * 1e3, ops / t2 / 1e6, t3 * 1e3, ops / t3 / 1e6);
  }
  for (int w : {1024, 4096, 16384}) {
    const int it = 650 * 1024 / w;
    const double ops = (double)w * it * 10;
    float t0 = timeit([&] { hipLaunchKernelGGL(k_rec<0>, dim3(w), dim3(64), 0, 0, buf, nrec, it, 10); });
    printf("rec lanes=10 waves=%5d iters=%3d agent %.1f us (%.1f Gop/s)\n", w, it, t0 * 1e3, ops / t0 / 1e6);
  }
  for (uint32_t n : {1024u, 65536u}) {
    const double ops = 1.25e6;
    const int threads = 300000, it = 4;
    float t0 = timeit([&] { hipLaunchKernelGGL(k_cnt<0>, dim3((threads + 255) / 256), dim3(256), 0, 0, cb, n, it); });
    float t1 = timeit([&] { hipLaunchKernelGGL(k_cnt<1>, dim3((threads + 255) / 256), dim3(256), 0, 0, cb, n, it); });
    printf("cnt n=%6u  agent %.1f us (%.1f Gop/s)  workgroup %.1f us (%.1f)\n", n, t0 * 1e3, 1.2e6 / t0 / 1e6, t1 * 1e3, 1.2e6 / t1 / 1e6);
    (void)ops;
  }
  return 0;
}
