// probe_hbm_read.hip -- what streaming-read rate does one MI355X sustain for the cache-replay access pattern?
// build: hipcc -O3 --offload-arch=gfx950 scripts/probe_hbm_read.hip -o build_tmp/probe_hbm_read ; run on the GPU box.
// Patterns over NS source arrays of `len` bytes each (NS * len ~ 1.6 GB, the cache arena of a cache-hit step at Bt 32):
//   A  one flat array of NS * len bytes, grid-stride 16-byte loads, U in flight
//   B  NS streams, every thread reads ITS chunk in all NS arrays (the replay kernel's pattern), U in flight
//   C  NS streams, a block owns CH chunks per thread (a contiguous span of 256 * CH * 16 bytes) and walks the sources one after another
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int U> __global__ __launch_bounds__(256) void pat_a(const u32x4* __restrict__ src, size_t n, u32x4* out) {
  u32x4 acc = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u];
  }
  for (; i < n; i += stride) acc ^= src[i];
  if (acc.x == 0x12345678) out[0] = acc;
}
template <int U> __global__ __launch_bounds__(256) void pat_b(const u32x4* const* __restrict__ srcs, int ns, size_t n, u32x4* out) {
  u32x4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    for (int s = 0; s + U <= ns; s += U) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(srcs[s + u] + i);
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u];
    }
  }
  if (acc.x == 0x12345678) out[0] = acc;
}
template <int CH> __global__ __launch_bounds__(256) void pat_c(const u32x4* const* __restrict__ srcs, int ns, size_t n, u32x4* out) {
  u32x4 acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) acc[c] = (u32x4){0, 0, 0, 0};
  for (size_t base = (size_t)blockIdx.x * 256 * CH; base < n; base += (size_t)gridDim.x * 256 * CH) {
    for (int s = 0; s < ns; ++s) {
      const u32x4* p = srcs[s] + base + threadIdx.x;
      u32x4 v[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) v[c] = __builtin_nontemporal_load(p + c * 256);
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] ^= v[c];
    }
  }
  u32x4 t = acc[0];
#pragma unroll
  for (int c = 1; c < CH; ++c) t ^= acc[c];
  if (t.x == 0x12345678) out[0] = t;
}

int main() {
  const int NS = 84;
  const size_t len = (size_t)8192 * 1152 * 2;  // one cache slot at Bt 32
  const size_t n = len / 16;
  char* arena;
  CK(hipMalloc((void**)&arena, NS * len));
  CK(hipMemset(arena, 1, NS * len));
  std::vector<const u32x4*> h(NS);
  for (int s = 0; s < NS; ++s) h[s] = (const u32x4*)(arena + s * len);
  const u32x4** d;
  CK(hipMalloc((void**)&d, NS * sizeof(void*)));
  CK(hipMemcpy(d, h.data(), NS * sizeof(void*), hipMemcpyHostToDevice));
  u32x4* out;
  CK(hipMalloc((void**)&out, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch) {
    launch();
    CK(hipDeviceSynchronize());
    float best = 1e9f, tot = 0;
    for (int r = 0; r < 5; ++r) {
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best; tot += ms;
    }
    printf("%-44s best %7.1f us  %6.0f GB/s   mean %7.1f us\n", name, best * 1e3, NS * len / (best * 1e-3) / 1e9, tot / 5 * 1e3);
  };
  for (int g : {256 * 8, 256 * 16, 256 * 32}) {
    char nm[96];
    snprintf(nm, 96, "A flat U=4 grid %d", g); timeit(nm, [&] { pat_a<4><<<g, 256>>>((const u32x4*)arena, NS * n, out); });
    snprintf(nm, 96, "A flat U=8 grid %d", g); timeit(nm, [&] { pat_a<8><<<g, 256>>>((const u32x4*)arena, NS * n, out); });
    snprintf(nm, 96, "B 84 streams U=4 grid %d", g); timeit(nm, [&] { pat_b<4><<<g, 256>>>(d, NS, n, out); });
    snprintf(nm, 96, "C span CH=4 grid %d", g); timeit(nm, [&] { pat_c<4><<<g, 256>>>(d, NS, n, out); });
    snprintf(nm, 96, "C span CH=8 grid %d", g); timeit(nm, [&] { pat_c<8><<<g, 256>>>(d, NS, n, out); });
  }
  timeit("B 84 streams U=4 grid 4608 (exact)", [&] { pat_b<4><<<4608, 256>>>(d, NS, n, out); });
  timeit("C span CH=2 grid 2304 (exact)", [&] { pat_c<2><<<2304, 256>>>(d, NS, n, out); });
  timeit("C span CH=9 grid 512 (exact)", [&] { pat_c<9><<<512, 256>>>(d, NS, n, out); });
  timeit("C span CH=3 grid 1536 (exact)", [&] { pat_c<3><<<1536, 256>>>(d, NS, n, out); });
  return 0;
}
