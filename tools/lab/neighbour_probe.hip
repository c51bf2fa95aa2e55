// What kind of neighbour slows the persistent tile kernels? Three side-stream kernels of the background builder's shape
// (64 workgroups x 256 threads, < 96 VGPRs: they fit beside a backward tile workgroup), each running for `usec`:
//   alu   : a dependent fma chain, no memory at all
//   gather: random 64-byte reads over `table` (`aux` = 0 cached, 16 device scope), as the builder's symmetry test does
//   atomic: random atomicAdd over `table`, as the builder's slot counting does
// hipcc --offload-arch=gfx950 -O3 -shared -fPIC neighbour_probe.hip -o neighbour_probe.so   (tools/lab/neighbour_probe.py drives it)
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

__global__ __launch_bounds__(256) void alu_kernel(float* out, long long ticks) {
  const long long t0 = wall_clock64();
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  while (wall_clock64() - t0 < ticks) {
#pragma unroll
    for (int i = 0; i < 64; ++i) a = __builtin_fmaf(a, b, 1e-7f);
  }
  if (a == 123.456f) out[0] = a;
}

template <bool DEVICE_SCOPE>
__global__ __launch_bounds__(256) void gather_kernel(const int4* table, uint32_t lines, int* out, long long ticks) {
  const long long t0 = wall_clock64();
  uint32_t s = blockIdx.x * 256 + threadIdx.x + 1;
  int acc = 0;
  while (wall_clock64() - t0 < ticks) {
    const uint32_t l0 = lcg(s) % lines, l1 = lcg(s) % lines;
    int4 v0, v1;
    if (DEVICE_SCOPE) {
      v0.x = __hip_atomic_load(&table[l0 * 4].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v1.x = __hip_atomic_load(&table[l1 * 4].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      v0 = table[l0 * 4]; v1 = table[l1 * 4];
    }
    acc += v0.x + v1.x;
  }
  if (acc == 0x7fffffff) out[0] = acc;
}

__global__ __launch_bounds__(256) void atomic_kernel(int* table, uint32_t words, int* out, long long ticks) {
  const long long t0 = wall_clock64();
  uint32_t s = blockIdx.x * 256 + threadIdx.x + 1;
  int acc = 0;
  while (wall_clock64() - t0 < ticks) acc += atomicAdd(&table[lcg(s) % words], 1);
  if (acc == 0x7fffffff) out[0] = acc;
}

extern "C" int probe_launch(int kind, void* table, uint32_t bytes, void* out, double usec, int blocks, void* stream) {
  const long long ticks = static_cast<long long>(usec * 100.0);   // wall_clock64: 100 MHz
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (kind == 0) alu_kernel<<<blocks, 256, 0, st>>>(static_cast<float*>(out), ticks);
  else if (kind == 1) gather_kernel<false><<<blocks, 256, 0, st>>>(static_cast<const int4*>(table), bytes / 64, static_cast<int*>(out), ticks);
  else if (kind == 2) gather_kernel<true><<<blocks, 256, 0, st>>>(static_cast<const int4*>(table), bytes / 64, static_cast<int*>(out), ticks);
  else atomic_kernel<<<blocks, 256, 0, st>>>(static_cast<int*>(table), bytes / 4, static_cast<int*>(out), ticks);
  return static_cast<int>(hipGetLastError());
}
