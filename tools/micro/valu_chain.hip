// Dev microbenchmark: what ONE wave's dependent VALU chain costs on gfx950, alone on its SIMD and beside others.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/valu_chain.hip -o build_ab/valu_chain && build_ab/valu_chain
// (every asm statement that executes a scalar ALU instruction names "scc" among its clobbers: the first cut of the
// last chain did not, the compiler kept its loop compare in SCC across the statement, and the loop never ended —
// caught by the watchdog below, which is why every launch here is waited for with a deadline.)
// Prints cycles per instruction (s_memtime, 100 MHz-independent: shader clock) for chains of `v_fma_f32` with
// ILP 1 / 2 / 4 (independent accumulators), a v_cmp -> v_cndmask chain, a v_rcp chain, and a dependent
// ds_read_b32 chain, at 1, 2, 4, 6, 8 waves per SIMD (blocks of 256 threads = one wave per SIMD of a CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
#include <unistd.h>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ __launch_bounds__(256) void chain(unsigned long long* out, float seed, int iters) {
  __shared__ unsigned int lds[256 * 4];
  float a = seed + threadIdx.x, b = a * 0.5f, c = a * 0.25f, d = a * 0.125f;
  const float m = 1.0000001f, k = 1e-9f;
  unsigned int idx = (threadIdx.x * 4u) & 1023u;
  for (int i = 0; i < 4; i++) lds[threadIdx.x * 4 + i] = ((threadIdx.x * 4 + i) * 4u + 16u) & 4095u;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  // (each chain is ONE asm statement — `.rept 64` — so that the compiler puts nothing between its instructions: between
  // separate asm statements it inserts an s_nop)
  for (int it = 0; it < iters; it++) {
    if (KIND == 0) asm volatile(".rept 64\n\tv_fma_f32 %0, %0, %1, %2\n\t.endr" : "+v"(a) : "v"(m), "v"(k));
    if (KIND == 1) asm volatile(".rept 64\n\tv_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3\n\t.endr" : "+v"(a), "+v"(b) : "v"(m), "v"(k));
    if (KIND == 2) asm volatile(".rept 64\n\tv_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\t.endr" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(k));
    if (KIND == 3) asm volatile(".rept 64\n\tv_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc\n\t.endr" : "+v"(a) : "v"(m), "v"(b) : "vcc");
    if (KIND == 4) asm volatile(".rept 64\n\tv_rcp_f32 %0, %0\n\t.endr" : "+v"(a));
    if (KIND == 5) asm volatile(".rept 64\n\tds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\t.endr" : "+v"(idx) : : "memory");
    if (KIND == 6) asm volatile(".rept 64\n\tv_cmp_lt_f32 vcc, %0, %1\n\ts_and_b64 s[20:21], vcc, exec\n\ts_cbranch_scc0 1f\n\tv_add_f32 %0, %0, %2\n1:\n\t.endr" : "+v"(a) : "v"(m), "v"(k) : "vcc", "s20", "s21", "scc");
    if (KIND == 7) asm volatile(".rept 64\n\tv_mul_f32 %0, %0, %1\n\tv_readfirstlane_b32 s20, %0\n\ts_add_u32 s20, s20, 1\n\tv_add_f32 %0, s20, %0\n\t.endr" : "+v"(a) : "v"(m) : "s20", "scc");
    if (KIND == 8) asm volatile(".rept 64\n\tv_cmp_lt_f32 vcc, %0, %1\n\ts_cbranch_vccz 1f\n\tv_add_f32 %0, %0, %2\n1:\n\t.endr" : "+v"(a) : "v"(m), "v"(k) : "vcc");
    if (KIND == 9) asm volatile(".rept 64\n\tv_cmp_lt_f32 vcc, %0, %1\n\ts_and_saveexec_b64 s[20:21], vcc\n\tv_add_f32 %0, %0, %2\n\ts_or_b64 exec, exec, s[20:21]\n\t.endr" : "+v"(a) : "v"(m), "v"(k) : "vcc", "s20", "s21", "scc");
    if (KIND == 10) asm volatile(".rept 64\n\ts_add_u32 s20, s20, 1\n\t.endr" : : : "s20", "scc");
    if (KIND == 11) asm volatile(".rept 64\n\ts_load_dword s20, %0, 0x0\n\ts_waitcnt lgkmcnt(0)\n\t.endr" : : "s"(out) : "s20", "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  if (a + b + c + d == 12345.678f || idx == 0xdeadbeefu) out[0] = 1; // keep the chains alive
}

// WATCHDOG: every launch is waited for by polling an event with a deadline; the first one that does not finish ends the run
static void wait_or_leave(const char* what, int w) {
  hipEvent_t ev;
  (void)hipEventCreate(&ev);
  (void)hipEventRecord(ev, 0);
  const auto t0 = std::chrono::steady_clock::now();
  while (hipEventQuery(ev) == hipErrorNotReady) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 5.0) {
      printf("\nWATCHDOG: %s at %d waves per SIMD did not finish within 5 s\n", what, w);
      fflush(stdout);
      (void)hipEventDestroy(ev);  // (non-blocking; the buffer is NOT freed: hipFree waits for the device, i.e. for the very
      _exit(3);                   //  kernel that does not end — leaving the process is what takes its queue off the GPU)
    }
    usleep(200);
  }
  (void)hipEventDestroy(ev);
}

template <int KIND>
void run(const char* name, int per_rep) {
  unsigned long long* d;
  hipMalloc(&d, 256 * 8 * 4 * 8 * sizeof(unsigned long long));
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  printf("%-34s", name);
  fflush(stdout);
  for (int w : {1, 2, 4, 6, 8}) {
    const int blocks = cus * w;
    const int iters = 200;
    for (int rep = 0; rep < 2; rep++) {
      chain<KIND><<<blocks, 256>>>(d, 1.0f, iters);
      wait_or_leave(name, w);
    }
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    s /= h.size();
    printf("  w=%d: %6.2f", w, s / (iters * 64.0 * per_rep));
    fflush(stdout);
  }
  printf("   (s_memtime ticks per instruction per wave)\n");
  hipFree(d);
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  printf("start\n");
  run<0>("v_fma dependent (ILP 1)", 1);
  run<1>("v_fma ILP 2", 2);
  run<2>("v_fma ILP 4", 4);
  run<3>("v_cmp -> v_cndmask (vcc) chain", 2);
  run<4>("v_rcp dependent", 1);
  run<5>("ds_read_b32 dependent + wait", 1);
  run<6>("v_cmp, s_and, s_cbranch, v_add", 4);
  run<7>("v_mul, readfirstlane, s_add, v_add", 4);
  run<8>("v_cmp, s_cbranch_vccz, v_add", 3);
  run<9>("v_cmp, s_and_saveexec, v_add, s_or", 4);
  run<10>("s_add_u32 dependent", 1);
  run<11>("s_load_dword + wait (K$ hit)", 1);
  return 0;
}
