// Microbenchmark: how fast can all CUs pull L2-resident data into LDS with buffer_load ... lds (the operand movement of
// conv2.hip / corr2.hip), as a function of the bytes a CU keeps in flight?  Measures the roofline that section 7.17 of
// DESIGN.md says bounds the convolution kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/l2_lds_stream.hip -o gpurun_out/l2_lds_stream && gpurun_out/l2_lds_stream
// Each block (256 threads) owns `lds_kb` of LDS split into two halves; per "stage" every wave issues `pieces` 1-KB
// pieces (16 B per lane) from a window of a buffer small enough to stay in L2, then waits for the stage issued one
// stage earlier (depth 1, like the 2-buffer loop) before re-using its half.  No MFMA, no ds_read: a pure DMA stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void dma16(const v4i rsrc, const void* lds, unsigned voff, int soff) {
  const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lptr_t)lds);
  const int so = __builtin_amdgcn_readfirstlane(soff);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(la), "v"(voff), "s"(rsrc), "s"(so) : "memory");
}

// HALF: a piece is 16 rows x 64 B (the lower or upper half of sixteen 128-byte lines, alternating by stage) instead
// of 8 whole lines -- what a 64-byte-row (16 split-fp16 channels per stage) variant of the conv loop would fetch
template <int PIECES, bool HALF = false>
__global__ void __launch_bounds__(256) stream_kernel(const char* buf, int window_bytes, int stages, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long a = (unsigned long long)buf;
  const v4i rs = {(int)(a & 0xffffffffu), (int)((a >> 32) & 0xffffu), window_bytes, 0x00020000};
  const int half = 4 * PIECES * 1024;  // bytes per stage of this block
  // every block walks the window from its own start: all of it stays hot in L2, neighbours share lines
  unsigned off = (unsigned)(((long)blockIdx.x * 7919L * 1024L) % window_bytes);
  auto issue = [&](int h) {
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      if constexpr (HALF) {
        const unsigned o = (off + (unsigned)((wave * PIECES + p) * 2048)) % (unsigned)window_bytes;
        dma16(rs, lds + h * half + (wave * PIECES + p) * 1024, (unsigned)((lane >> 2) * 128 + (lane & 3) * 16 + h * 64), (int)o);
      } else {
        const unsigned o = (off + (unsigned)((wave * PIECES + p) * 1024)) % (unsigned)window_bytes;
        dma16(rs, lds + h * half + (wave * PIECES + p) * 1024, (unsigned)(lane * 16), (int)o);
      }
    }
    if (!HALF || h == 1) off = (off + (HALF ? 2 * half : half)) % (unsigned)window_bytes;
  };
  issue(0);
  for (int s = 0; s < stages; s += 2) {
    issue(1);
    if (PIECES == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (PIECES == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (PIECES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (PIECES == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue(0);
    if (PIECES == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (PIECES == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (PIECES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (PIECES == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (sink != nullptr && threadIdx.x == 0) sink[blockIdx.x] = ((int*)lds)[blockIdx.x & 63];
}

template <int PIECES, bool HALF = false>
static void run(const char* buf, int window, int blocks_per_cu, int cus, int* sink) {
  const int stages = 400;
  const size_t lds = 2 * 4 * PIECES * 1024;
  hipFuncSetAttribute((const void*)stream_kernel<PIECES, HALF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int grid = blocks_per_cu * cus;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_kernel<PIECES, HALF>), dim3(grid), dim3(256), lds, 0, buf, window, stages, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)grid * (stages + 1) * 4 * PIECES * 1024;
  printf("{\"half_lines\": %d, \"pieces_per_wave_stage\": %d, \"blocks_per_cu\": %d, \"kb_in_flight_per_cu\": %d, \"window_mb\": %.1f, \"ms\": %.4f, \"TB_per_s\": %.2f}\n",
         (int)HALF, PIECES, blocks_per_cu, 2 * 4 * PIECES * blocks_per_cu, window / 1048576.0, ms, bytes / ms / 1e9);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  char* buf; int* sink;
  hipMalloc(&buf, 512 << 20);
  hipMemset(buf, 1, 512 << 20);
  hipMalloc(&sink, 1 << 20);
  printf("{\"device\": \"%s\", \"cus\": %d}\n", prop.name, cus);
  for (int window : {2 << 20, 16 << 20, 256 << 20}) {  // L2-resident (per XCD), MALL-resident, HBM
    for (int bpc : {1, 2, 3, 4}) {
      run<2>(buf, window, bpc, cus, sink);
      run<4>(buf, window, bpc, cus, sink);
      run<6>(buf, window, bpc, cus, sink);
      if (bpc <= 2) run<12>(buf, window, bpc, cus, sink);
      if (bpc == 3) { run<3, true>(buf, window, bpc, cus, sink); run<6, true>(buf, window, bpc, cus, sink); }
    }
  }
  hipFree(buf); hipFree(sink);
  return 0;
}
