// Cost of a grid-wide barrier inside one kernel (cooperative launch) versus a kernel boundary on the same stream:
// decides whether fusing the 3-4 GEMMs of a single-molecule MLP into one cooperative kernel can pay.
//   hipcc --offload-arch=gfx950 -O3 -o grid_sync grid_sync.hip && ./grid_sync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ void k_sync(float* x, int n_sync) {
  cg::grid_group g = cg::this_grid();
  float v = x[blockIdx.x * blockDim.x + threadIdx.x];
  for (int s = 0; s < n_sync; ++s) {
    v = v * 1.0001f + 1.0f;
    x[blockIdx.x * blockDim.x + threadIdx.x] = v;
    g.sync();
    v += x[((blockIdx.x + 1) % gridDim.x) * blockDim.x + threadIdx.x];
  }
  x[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
__global__ void k_step(float* x) {
  float v = x[blockIdx.x * blockDim.x + threadIdx.x];
  v = v * 1.0001f + 1.0f + x[((blockIdx.x + 1) % gridDim.x) * blockDim.x + threadIdx.x];
  x[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
int main() {
  for (int blocks : {64, 256, 512}) {
    float* x;
    hipMalloc(&x, blocks * 256 * sizeof(float));
    hipMemset(x, 0, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int n_sync = 50;
    void* args[] = {&x, &n_sync};
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchCooperativeKernel((void*)k_sync, dim3(blocks), dim3(256), args, 0, 0);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
    }
    float ms_sync; hipEventElapsedTime(&ms_sync, e0, e1);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      for (int s = 0; s < 50; ++s) hipLaunchKernelGGL(k_step, dim3(blocks), dim3(256), 0, 0, x);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
    }
    float ms_k; hipEventElapsedTime(&ms_k, e0, e1);
    printf("blocks %4d: grid.sync %.2f us each   kernel boundary %.2f us each   (err %s)\n", blocks, ms_sync * 1e3 / 50, ms_k * 1e3 / 50,
           hipGetErrorString(hipGetLastError()));
    hipFree(x);
  }
  return 0;
}
