// Issue rates of the fp64-class VALU instructions the resamplers use (gfx950), one wave per SIMD and
// four waves per SIMD: cycles per wave-instruction from s_memtime around an unrolled independent stream.
//   hipcc -O3 --offload-arch=gfx950 -o fp64_rates fp64_rates.hip && ./fp64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ void probe(double* out, unsigned long long* cyc, double a0, double b0) {
  double a[8], r[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = a0 + i + threadIdx.x * 1e-3; r[i] = b0 + i; }
  float fr[8];
  int ir[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { fr[i] = static_cast<float>(a[i]); ir[i] = i; }
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) { REP16(asm volatile("v_add_f64 %0, %0, %1" : "+v"(r[i]) : "v"(a[i]));) }
      if (OP == 1) { REP16(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r[i]) : "v"(a[i]));) }
      if (OP == 2) { REP16(asm volatile("v_fract_f64 %0, %1" : "=v"(r[i]) : "v"(a[i]));) }
      if (OP == 3) { REP16(asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(ir[i]) : "v"(a[i]));) }
      if (OP == 4) { REP16(asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(fr[i]) : "v"(a[i]));) }
      if (OP == 5) { REP16(asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(r[i]) : "v"(fr[i]));) }
      if (OP == 6) { REP16(asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(r[i]), "v"(a[i]) : "vcc");) }
      if (OP == 7) { REP16(asm volatile("v_floor_f64 %0, %1" : "=v"(r[i]) : "v"(a[i]));) }
      if (OP == 8) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(r[i]) : "v"(a[i]));) }
      if (OP == 9) { REP16(asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r[i]) : "v"(ir[i]));) }
      if (OP == 10) { REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(fr[i]) : "v"(fr[(i + 1) & 7]));) }
      if (OP == 11) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(ir[i]) : "v"(ir[(i + 1) & 7]));) }
      if (OP == 12) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(ir[i]) : "v"(ir[(i + 1) & 7]));) }
      if (OP == 13) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ir[i]) : "v"(ir[(i + 1) & 7]) : "vcc");) }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += r[i] + fr[i] + ir[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name) {
  double* out;
  unsigned long long* cyc;
  hipMalloc(&out, 256 * 1024 * sizeof(double));
  hipMalloc(&cyc, 256 * sizeof(unsigned long long));
  for (int threads : {256, 1024}) {   // 1 and 4 waves per SIMD, one workgroup per CU
    hipLaunchKernelGGL(probe<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.5, 2.5);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (auto v : h) mx = v > mx ? v : mx;
    const double insts = 64.0 * 8 * 16;           // per wave
    // s_memtime counts at 100 MHz on gfx9: convert with the shader clock measured by the host instead
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.5, 2.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = threads / 256.0;
    // time ~ insts * waves_per_simd * cycles_per_inst / clock  (launch overhead ~5 us ignored: stream is ~100+ us)
    printf("%-16s %4d thr: %.1f us  -> %.2f cycles per wave-instruction at 2.4 GHz (memtime ticks %llu)\n", name,
           threads, ms * 1e3, ms * 1e-3 * 2.4e9 / (insts * waves_per_simd), mx);
  }
  hipFree(out);
  hipFree(cyc);
}

int main() {
  run<0>("v_add_f64"); run<1>("v_mul_f64"); run<8>("v_fma_f64"); run<2>("v_fract_f64"); run<7>("v_floor_f64");
  run<3>("v_cvt_i32_f64"); run<9>("v_cvt_f64_i32"); run<4>("v_cvt_f32_f64"); run<5>("v_cvt_f64_f32");
  run<6>("v_cmp_lt_f64"); run<10>("v_add_f32"); run<11>("v_mul_lo_u32"); run<12>("v_mad_u32_u24");
  run<13>("v_cndmask_b32");
  return 0;
}
