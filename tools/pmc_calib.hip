// pmc_calib.hip -- known-volume kernels to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE per ACCESS SHAPE on gfx950.
//
// MI355X_MICROARCH.md calibrates one shape only (wide coalesced streaming reads of 16 B per lane: FETCH_SIZE reports
// half the bytes) and says every other shape must be calibrated on a known byte count.  The kernels below request an
// exactly known number of bytes in the shapes this library's kernels use:
//   k_stream_read      16 B per lane, consecutive lanes consecutive addresses (the reference shape)
//   k_gather64         every lane reads ONE 64-byte row (4 x 16 B) at a random row of a table far larger than the
//                      Infinity Cache (msm_chunk_kernel's table gathers) -- and of a table that fits it
//   k_strided_runs     runs of 64 / 128 contiguous bytes, `stride` bytes apart (ntt29_pass_kernel's tile loads: C = 2 or
//                      4 columns of 32 bytes)
//   k_stream_write     16 B per lane streaming stores;  k_scatter4: one 4-byte store per lane at random words;
//   k_runs_write       64-byte runs at random places (the staged scatters' stores)
// Usage (GPU box): rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out_f ... -- ./pmc_calib ; the same with WRITE_SIZE;
// tools/pmc_calib.py divides the requested bytes (printed by this program) by the counters.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

struct alignas(16) V16 { uint32_t x, y, z, w; };

__device__ __forceinline__ uint32_t mix(uint32_t v) {
  v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
  return v;
}
__global__ void k_stream_read(const V16* __restrict__ src, size_t n16, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const V16 v = src[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_gather64(const V16* __restrict__ table, uint32_t rows, uint32_t per_thread, uint32_t* sink) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (uint32_t k = 0; k < per_thread; k++) {
    const uint32_t r = mix(t * 2654435761u + k * 40503u + 12345u) % rows;
    const V16* p = table + 4 * (size_t)r;
    const V16 a = p[0], b = p[1], c = p[2], d = p[3];
    acc ^= a.x ^ b.y ^ c.z ^ d.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_strided_runs(const V16* __restrict__ src, uint32_t run16 /* 16-byte words per run */, size_t stride16,
                               size_t runs, uint32_t* sink) {
  // consecutive lanes take consecutive 16-byte words of a run, then the next run
  uint32_t acc = 0;
  const size_t total = runs * run16;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t run = i / run16, w = i % run16;
    // runs visited in a scrambled order so that neighbouring runs are not neighbours in time
    const size_t rr = (run * 2654435761ull) % runs;
    const V16 v = src[rr * stride16 + w];
    acc ^= v.x ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_stream_write(V16* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = V16{(uint32_t)i, 1, 2, 3};
}
__global__ void k_scatter4(uint32_t* __restrict__ dst, uint32_t words, uint32_t per_thread) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  for (uint32_t k = 0; k < per_thread; k++) dst[mix(t * 2654435761u + k * 40503u + 777u) % words] = t;
}
__global__ void k_runs_write(uint32_t* __restrict__ dst, uint32_t runs_total, uint32_t per_thread) {
  // 16 consecutive lanes write one 64-byte run at a random 64-byte-aligned place
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t g = t >> 4, l = t & 15;
  for (uint32_t k = 0; k < per_thread; k++) dst[(size_t)(mix(g * 2654435761u + k * 40503u + 99u) % runs_total) * 16 + l] = t;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); return 1; } } while (0)

int main() {
  const size_t big = (size_t)1 << 30, small = (size_t)64 << 20;     // 1 GiB (beyond the 256 MiB Infinity Cache), 64 MiB
  void *buf, *sink;
  CK(hipMalloc(&buf, big));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(buf, 1, big));
  const int blocks = 256 * 8, threads = 256;
  const uint32_t nthreads = blocks * threads;
  // every kernel twice: the second run is the one to read (clocks, TLBs warm); requested bytes printed per kernel
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k_stream_read, dim3(blocks), dim3(threads), 0, 0, (const V16*)buf, big / 16, (uint32_t*)sink);
    hipLaunchKernelGGL(k_gather64, dim3(blocks), dim3(threads), 0, 0, (const V16*)buf, (uint32_t)(big / 64), 16u, (uint32_t*)sink);
    hipLaunchKernelGGL(k_gather64, dim3(blocks), dim3(threads), 0, 0, (const V16*)buf, (uint32_t)(small / 64), 16u, (uint32_t*)sink);
    hipLaunchKernelGGL(k_strided_runs, dim3(blocks), dim3(threads), 0, 0, (const V16*)buf, 4u, (size_t)1024, (big / 16) / 1024, (uint32_t*)sink);
    hipLaunchKernelGGL(k_strided_runs, dim3(blocks), dim3(threads), 0, 0, (const V16*)buf, 8u, (size_t)1024, (big / 16) / 1024, (uint32_t*)sink);
    hipLaunchKernelGGL(k_stream_write, dim3(blocks), dim3(threads), 0, 0, (V16*)buf, big / 16);
    hipLaunchKernelGGL(k_scatter4, dim3(blocks), dim3(threads), 0, 0, (uint32_t*)buf, (uint32_t)(big / 4), 16u);
    hipLaunchKernelGGL(k_runs_write, dim3(blocks), dim3(threads), 0, 0, (uint32_t*)buf, (uint32_t)(big / 64), 16u);
    CK(hipDeviceSynchronize());
  }
  const size_t runs = (big / 16) / 1024;
  printf("{\"k_stream_read\": {\"read\": %zu}, \"k_gather64\": [{\"read\": %zu, \"table\": %zu}, {\"read\": %zu, \"table\": %zu}], "
         "\"k_strided_runs\": [{\"read\": %zu, \"run\": 64}, {\"read\": %zu, \"run\": 128}], \"k_stream_write\": {\"write\": %zu}, "
         "\"k_scatter4\": {\"write\": %zu}, \"k_runs_write\": {\"write\": %zu}}\n",
         big, (size_t)nthreads * 16 * 64, big, (size_t)nthreads * 16 * 64, small, runs * 64, runs * 128, big,
         (size_t)nthreads * 16 * 4, (size_t)(nthreads / 16) * 16 * 64);
  return 0;
}
