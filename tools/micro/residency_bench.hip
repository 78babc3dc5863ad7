// Micro-benchmark (measurement aid): how many workgroups of a given shape actually start together on MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void spin(unsigned long long* out, int spin_us, int lds_words) {
  extern __shared__ float lds[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (lds_words > 0) lds[threadIdx.x % lds_words] = 1.f;
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    out[blockIdx.x * 2] = t0;
    out[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_getreg((4 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
  }
}
int main() {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  printf("CUs %d, maxThreadsPerMP %d, sharedMemPerMP %zu, regsPerMP %d\n", pr.multiProcessorCount, pr.maxThreadsPerMultiProcessor,
         pr.sharedMemPerMultiprocessor, pr.regsPerMultiprocessor);
  unsigned long long* d; hipMalloc(&d, 1 << 20);
  struct Cfg { int blocks, threads, lds; };
  for (Cfg c : {Cfg{1024, 64, 6144}, Cfg{1024, 320, 34820}, Cfg{1024, 320, 20000}, Cfg{1024, 256, 37152}, Cfg{2048, 64, 6144}, Cfg{768, 320, 34820}, Cfg{1024, 192, 40452}}) {
    hipLaunchKernelGGL(spin, dim3(c.blocks), dim3(c.threads), c.lds, 0, d, 20, c.lds / 4);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(c.blocks * 2);
    hipMemcpy(h.data(), d, c.blocks * 16, hipMemcpyDeviceToHost);
    unsigned long long mn = ~0ull; for (int i = 0; i < c.blocks; ++i) mn = std::min(mn, h[2 * i]);
    int late = 0; int perx[8] = {0}; int latex[8] = {0};
    for (int i = 0; i < c.blocks; ++i) { bool l = (h[2 * i] - mn) > 500; late += l; perx[h[2 * i + 1] & 7]++; latex[h[2 * i + 1] & 7] += l; }
    printf("blocks %4d threads %3d lds %5d : %4d start late (>5us)  per-XCD blocks/late:", c.blocks, c.threads, c.lds, late);
    for (int x = 0; x < 8; ++x) printf(" %d/%d", perx[x], latex[x]);
    printf("\n");
  }
  return 0;
}
