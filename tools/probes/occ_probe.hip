// Occupancy probe: resident 512-thread workgroups per CU vs static LDS size and VGPR count.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS_FLOATS, int REGS>
__global__ __launch_bounds__(512) void k(float* out) {
  __shared__ float s[LDS_FLOATS];
  float r[REGS];
  for (int i = 0; i < REGS; ++i) r[i] = out[threadIdx.x + i * 512];
  s[threadIdx.x] = r[0];
  __syncthreads();
  float acc = s[(threadIdx.x * 7) % LDS_FLOATS];
  for (int i = 0; i < REGS; ++i) acc = acc * r[i] + r[(i + 1) % REGS];
  out[threadIdx.x] = acc;
}
template <int L, int R>
void probe() {
  int n = -1;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k<L, R>, 512, 0);
  hipFuncAttributes a;
  hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k<L, R>));
  printf("LDS %6d B  numRegs %3d  -> %d workgroups/CU\n", L * 4, a.numRegs, n);
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("sharedMemPerBlock %zu sharedMemPerMultiprocessor %zu regsPerBlock %d regsPerMultiprocessor %d maxThreadsPerMultiProcessor %d\n",
         p.sharedMemPerBlock, p.sharedMemPerMultiprocessor, p.regsPerBlock, p.regsPerMultiprocessor, p.maxThreadsPerMultiProcessor);
  probe<4096, 8>(); probe<8192, 8>(); probe<10240, 8>(); probe<11648, 8>(); probe<12288, 8>(); probe<16384, 8>();
  probe<4096, 40>(); probe<4096, 56>(); probe<4096, 72>(); probe<4096, 88>(); probe<4096, 100>(); probe<4096, 120>();
  return 0;
}
