// Microbenchmark: VALU issue rate on gfx950 for the instruction kinds the intersection loop uses.
// Not part of the product; informs the roofline denominator in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  float2v p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
  float2v pa = {a, a}, pb = {b, b};
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (MODE == 0) {  // v_fma_f32
        x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
        x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
      } else if (MODE == 1) {  // v_pk_fma_f32
        p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb);
        p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb);
        p4 = __builtin_elementwise_fma(p4, pa, pb); p5 = __builtin_elementwise_fma(p5, pa, pb);
        p6 = __builtin_elementwise_fma(p6, pa, pb); p7 = __builtin_elementwise_fma(p7, pa, pb);
      } else if (MODE == 2) {  // v_sub / v_mul mix
        x0 = x0 - a; x1 = x1 * b; x2 = x2 - a; x3 = x3 * b; x4 = x4 - a; x5 = x5 * b; x6 = x6 - a; x7 = x7 * b;
      } else if (MODE == 3) {  // v_pk_mul / v_pk_add
        p0 = p0 * pa; p1 = p1 + pb; p2 = p2 * pa; p3 = p3 + pb; p4 = p4 * pa; p5 = p5 + pb; p6 = p6 * pa; p7 = p7 + pb;
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

template <int MODE>
void run(const char* name, int wg_per_cu, int flop_per_instr) {
  int cus = 256, iters = 20000;
  int blocks = cus * wg_per_cu;
  float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 100, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr_per_wave = (double)iters * 64;
  double waves_per_simd = wg_per_cu;  // 4 waves per WG, 4 SIMDs
  double cyc = ms * 1e-3 * 2.4e9;
  double cyc_per_instr_simd = cyc / (instr_per_wave * waves_per_simd);
  double tflops = (double)blocks * 4 * instr_per_wave * 64 * flop_per_instr / (ms * 1e-3) / 1e12;
  printf("%-22s waves/SIMD %d: %.3f ms, %.2f cycles(@2.4GHz)/instr/SIMD, %.1f TFLOP/s\n", name, wg_per_cu, ms, cyc_per_instr_simd, tflops);
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_fma_f32", w, 2);
    run<1>("v_pk_fma_f32", w, 4);
    run<2>("v_sub/v_mul", w, 1);
    run<3>("v_pk_mul/v_pk_add", w, 2);
  }
  return 0;
}
