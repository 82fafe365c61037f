// Dev microbenchmark: what a DIVERGENT gather instruction costs the vector-memory pipe of a gfx950 CU, by access pattern.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/gather_cost.hip -o build_ab/gather_cost && build_ab/gather_cost
// Background (profiles/r05_config5_cache_model.txt): BASELINE config 5's grid kernel is bound by the NUMBER of its 128-bit
// gathers — 29 TA-busy cycles each, whatever the number of active lanes.  Which part of that is the pipe's fixed rate
// (a wave's 64 addresses / 1024 bytes of return data) and which part follows the pattern?  The table a lane reads is the
// size of config 5's entry runs (0.4 MB: L2-resident, 88 % L1 hits in the product); every lane draws a fresh random run
// per trip (an LCG per lane), as the lanes of a leaf round stand in different cells.
//   P0  four 128-bit loads at base, +16, +32, +48 of the lane's own run               (the product's leaf round)
//   P1  the same bytes, QUAD-cooperative: in load j the four lanes of a quad read the 64 contiguous bytes of lane
//       4 q + j's run (one line per quad and load instead of four)                    (needs a 4 x 4 transpose afterwards)
//   P2  four 64-bit loads (entries compressed to 8 bytes)
//   P3  four 32-bit loads
//   P4  P0 with every second lane switched off                                       (per instruction or per lane?)
//   P5  two 128-bit loads per trip (linearity check: half of P0's instructions)
//   P6  P0 with ALL lanes of a wave on the same run                                   (the coherent limit)
// Shape as the product: 512 threads per workgroup, three workgroups per CU (6 waves per SIMD); result: shader-clock cycles
// per load instruction per CU (wave-level loads issued by one CU / elapsed cycles) and ns per wave-level load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#include <unistd.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr unsigned TABLE_BYTES = 24131u * 16u;  // config 5's entries
constexpr unsigned RUN_SLOTS = (TABLE_BYTES - 64u) / 16u;

template <int P, bool DEP>
__global__ __launch_bounds__(512) void gather(const float4* __restrict__ table, float* out, int trips, unsigned seed, unsigned window_slots) {
  const unsigned lane = threadIdx.x & 63u;
  unsigned s = seed ^ ((blockIdx.x * 512u + threadIdx.x) * 2654435761u);
  if (P == 6) s = seed ^ ((blockIdx.x * 8u + (threadIdx.x >> 6)) * 2654435761u);
  float acc = 0.f;
  const char* tb = reinterpret_cast<const char*>(table);
  if (P == 4 && (lane & 1u)) return;
  for (int t = 0; t < trips; t++) {
    s = s * 1664525u + 1013904223u;
    // 16-byte slot at which this trip's run starts: anywhere in the table (window_slots == 0: L2 traffic dominates) or within
    // the workgroup's own window of it (the L1 holds the three windows of a CU: the hit path)
    const unsigned run = window_slots ? (blockIdx.x * 2731u * 64u) % (RUN_SLOTS - window_slots) + (s >> 8) % window_slots : (s >> 8) % RUN_SLOTS;
    unsigned off = run * 16u;
    if (P == 0 || P == 4 || P == 6) {
      const float4 a = *reinterpret_cast<const float4*>(tb + off), b = *reinterpret_cast<const float4*>(tb + off + 16u),
                   c = *reinterpret_cast<const float4*>(tb + off + 32u), d = *reinterpret_cast<const float4*>(tb + off + 48u);
      acc += a.x + b.y + c.z + d.w;
    } else if (P == 1) {
      // load j: the quad's lanes read bytes [16 (lane & 3), +16) of lane (4 q + j)'s run
      const unsigned o0 = __shfl(off, (lane & ~3u) | 0u), o1 = __shfl(off, (lane & ~3u) | 1u), o2 = __shfl(off, (lane & ~3u) | 2u),
                     o3 = __shfl(off, (lane & ~3u) | 3u);
      const unsigned sub = 16u * (lane & 3u);
      const float4 a = *reinterpret_cast<const float4*>(tb + o0 + sub), b = *reinterpret_cast<const float4*>(tb + o1 + sub),
                   c = *reinterpret_cast<const float4*>(tb + o2 + sub), d = *reinterpret_cast<const float4*>(tb + o3 + sub);
      acc += a.x + b.y + c.z + d.w;
    } else if (P == 2) {
      off = run * 8u;
      const float2 a = *reinterpret_cast<const float2*>(tb + off), b = *reinterpret_cast<const float2*>(tb + off + 8u),
                   c = *reinterpret_cast<const float2*>(tb + off + 16u), d = *reinterpret_cast<const float2*>(tb + off + 24u);
      acc += a.x + b.y + c.x + d.y;
    } else if (P == 3) {
      off = run * 4u;
      const float a = *reinterpret_cast<const float*>(tb + off), b = *reinterpret_cast<const float*>(tb + off + 4u),
                  c = *reinterpret_cast<const float*>(tb + off + 8u), d = *reinterpret_cast<const float*>(tb + off + 12u);
      acc += a + b + c + d;
    } else if (P == 5) {
      const float4 a = *reinterpret_cast<const float4*>(tb + off), b = *reinterpret_cast<const float4*>(tb + off + 16u);
      acc += a.x + b.y;
    }
    // the next trip's address depends on this trip's data only through a value that is always 0 (keeps the loads of
    // DIFFERENT waves the only parallelism, as in the product, where a leaf round's candidates decide what comes next)
    if (DEP) s += (acc == 1.2345e30f) ? 1u : 0u;
  }
  if (acc == 1.2345e30f) out[0] = acc;
}

static void wait_or_leave(const char* what) {
  hipEvent_t ev;
  (void)hipEventCreate(&ev);
  (void)hipEventRecord(ev, 0);
  const auto t0 = std::chrono::steady_clock::now();
  while (hipEventQuery(ev) == hipErrorNotReady) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 10.0) {
      printf("\nWATCHDOG: %s did not finish within 10 s\n", what);
      fflush(stdout);
      (void)hipEventDestroy(ev);
      _exit(3);
    }
    usleep(200);
  }
  (void)hipEventDestroy(ev);
}

template <int P, bool DEP>
static void run(const char* name, int loads_per_trip, const float4* table, float* out, int cus, double mhz, unsigned window_slots) {
  const int trips = 4000;
  const int blocks = cus * 3;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 4; rep++) {
    CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((gather<P, DEP>), dim3(blocks), dim3(512), 0, 0, table, out, trips, 0x9e3779b9u + rep, window_slots);
    CHECK(hipGetLastError());
    CHECK(hipEventRecord(b, 0));
    wait_or_leave(name);
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, a, b));
    if (rep > 0 && ms < best) best = ms;
  }
  const double loads_per_cu = 3.0 * 8.0 * (double)trips * loads_per_trip;  // wave-level loads one CU issues
  const double cycles = best * 1e-3 * mhz * 1e6;
  printf("%-66s %8.3f ms   %6.1f cycles per wave-level load per CU   (%5.2f ns)\n", name, best, cycles / loads_per_cu, best * 1e6 / loads_per_cu);
  fflush(stdout);
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double mhz = prop.clockRate / 1000.0;
  printf("# %s, %d CUs, %.0f MHz (cycles are at this clock); table %u bytes; 3 x 512 threads per CU, 4000 trips per lane\n", prop.name, cus, mhz, TABLE_BYTES);
  std::vector<float> h(TABLE_BYTES / 4 + 64);
  for (size_t i = 0; i < h.size(); i++) h[i] = (float)(i % 977) * 0.001f;
  float4* table = nullptr;
  float* out = nullptr;
  CHECK(hipMalloc(&table, h.size() * 4));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  for (unsigned window_slots : {0u, 512u, 128u}) {
    if (window_slots) printf("\n### every workgroup draws its runs from a window of %u bytes of the table (three windows per CU: L1 hits)\n", window_slots * 16u);
    else printf("\n### runs drawn from the whole table (0.4 MB against 32 KB of L1: the L2 -> L1 path is the bound)\n");
    printf("## trips INDEPENDENT (a wave keeps as many loads in flight as its counters allow: the pipe's throughput)\n");
    run<0, false>("P0 4 x 128-bit, the lane's own run (the product's leaf round)", 4, table, out, cus, mhz, window_slots);
    run<1, false>("P1 4 x 128-bit, quad-cooperative (64 contiguous bytes per quad and load)", 4, table, out, cus, mhz, window_slots);
    run<2, false>("P2 4 x 64-bit, the lane's own run", 4, table, out, cus, mhz, window_slots);
    run<3, false>("P3 4 x 32-bit, the lane's own run", 4, table, out, cus, mhz, window_slots);
    run<4, false>("P4 P0 with every second lane off", 4, table, out, cus, mhz, window_slots);
    run<5, false>("P5 2 x 128-bit, the lane's own run", 2, table, out, cus, mhz, window_slots);
    run<6, false>("P6 4 x 128-bit, all lanes of a wave on one run", 4, table, out, cus, mhz, window_slots);
    printf("## every trip DEPENDS on the one before (four loads in flight per wave: latency shows)\n");
    run<0, true>("P0 4 x 128-bit, the lane's own run", 4, table, out, cus, mhz, window_slots);
    run<3, true>("P3 4 x 32-bit, the lane's own run", 4, table, out, cus, mhz, window_slots);
    run<5, true>("P5 2 x 128-bit, the lane's own run", 2, table, out, cus, mhz, window_slots);
  }
  CHECK(hipFree(table));
  CHECK(hipFree(out));
  return 0;
}
