// Does global_load_lds_dwordx4 (16-byte LDS-DMA) take source addresses that are only 4- or 8-byte aligned?
//   hipcc --offload-arch=gfx950 -O3 tools/dma16_probe.hip -o tools/_build/dma16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef const __attribute__((address_space(1))) void glb_ptr_t;
__global__ __launch_bounds__(64) void k(const float *in, float *out, int shift, int stride) {
  extern __shared__ float lds[];
  // lane l: 16 bytes from in + shift + l * stride (floats)
  __builtin_amdgcn_global_load_lds((glb_ptr_t *)(in + shift + threadIdx.x * stride), (lds_ptr_t *)lds, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
  const int N = 1 << 16;
  std::vector<float> h(N);
  for (int i = 0; i < N; i++) h[i] = (float)i;
  float *in, *out;
  hipMalloc(&in, N * 4);
  hipMalloc(&out, 256 * 4);
  hipMemcpy(in, h.data(), N * 4, hipMemcpyHostToDevice);
  int bad_total = 0;
  for (int shift = 0; shift < 4; shift++)
    for (int stride : {4, 5, 66}) {
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, in, out, shift, stride);
      std::vector<float> o(256);
      hipMemcpy(o.data(), out, 1024, hipMemcpyDeviceToHost);
      int bad = 0;
      for (int l = 0; l < 64; l++)
        for (int j = 0; j < 4; j++) bad += o[l * 4 + j] != (float)(shift + l * stride + j);
      printf("shift %d floats, lane stride %d floats: %s (%d wrong of 256)\n", shift, stride, bad ? "WRONG" : "ok", bad);
      bad_total += bad;
    }
  return bad_total != 0;
}
