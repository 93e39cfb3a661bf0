// Where does global_load_lds_dwordx4 with an instruction offset put its data?  (gfx950)
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_glds_off.hip -o /tmp/glds_off && /tmp/glds_off
// One wave: M0 = LDS base + 256, offset:1024, source = table of dwords whose value is their own index.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const uint32_t* src, uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 0xffffffffu;
  __syncthreads();
  const uint32_t base = (uint32_t)(uintptr_t)lds + 256;
  const uint32_t voff = threadIdx.x * 16;
  const unsigned long long sa = (unsigned long long)src;
  const unsigned long long ss = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sa >> 32)) << 32) |
                                (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sa);
  const uint32_t sb = __builtin_amdgcn_readfirstlane(base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\ts_waitcnt vmcnt(0)"
               :: "s"(sb), "v"(voff), "s"(ss) : "memory", "m0");
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}
int main() {
  uint32_t h[4096], *d, *o, r[2048];
  for (int i = 0; i < 4096; ++i) h[i] = i;
  (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&o, sizeof(r));
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  (void)hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  int first = -1, last = -1;
  for (int i = 0; i < 2048; ++i) if (r[i] != 0xffffffffu) { if (first < 0) first = i; last = i; }
  printf("LDS dwords written: [%d, %d]  (M0 base at dword 64; +offset would start at dword 320)\n", first, last);
  if (first >= 0) printf("value at first written dword: %u  (source dword index; 256 = the instruction offset applied to the source)\n", r[first]);
  return 0;
}
