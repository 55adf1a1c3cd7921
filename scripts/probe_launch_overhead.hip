// probe_launch_overhead.hip -- what one dependent kernel node of a hipGraph costs on this chip, by shape of the launch.
// build: hipcc -O3 --offload-arch=gfx950 scripts/probe_launch_overhead.hip -o build_tmp/probe_launch_overhead ; run on the GPU box.
// A linear chain of 2000 kernel nodes is captured from a stream and replayed; time per node by events.  Variants: grid 1 / 256 / 1024 workgroups of 512
// threads; 0 or 150 KB of dynamic LDS (one workgroup per CU, as the GEMM tiles); empty body, or every workgroup writes 64 KB (dirty L2 lines at the boundary).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct Big { unsigned long long pad[24]; char* buf; int write; };
__global__ __launch_bounds__(512) void node(const Big p) {
  extern __shared__ char smem[];
  if (p.write) {
    uint4* dst = reinterpret_cast<uint4*>(p.buf + (size_t)blockIdx.x * 65536);
    for (int i = threadIdx.x; i < 4096; i += 512) dst[i] = make_uint4(i, blockIdx.x, 0, 1);
  }
  if (p.pad[0] == 12345) smem[threadIdx.x] = 1;
}
void run(const char* name, int grid, int lds, int write, char* buf) {
  hipStream_t s; CK(hipStreamCreate(&s));
  CK(hipFuncSetAttribute((const void*)node, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  Big p{}; p.buf = buf; p.write = write;
  const int N = 2000;
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(node, dim3(grid), dim3(512), lds, s, p);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-34s %6.2f us per node\n", name, ms * 1e3 / N);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(s));
}
int main() {
  char* buf; CK(hipMalloc(&buf, (size_t)1024 * 65536));
  run("grid 1, no LDS, empty", 1, 0, 0, buf);
  run("grid 256, no LDS, empty", 256, 0, 0, buf);
  run("grid 256, 150 KB LDS, empty", 256, 150 * 1024, 0, buf);
  run("grid 1024, 150 KB LDS, empty", 1024, 150 * 1024, 0, buf);
  run("grid 256, 150 KB LDS, 64 KB/WG", 256, 150 * 1024, 1, buf);
  run("grid 1024, 150 KB LDS, 64 KB/WG", 1024, 150 * 1024, 1, buf);
  return 0;
}
