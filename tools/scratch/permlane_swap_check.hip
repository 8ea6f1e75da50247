// Standalone check of mi::rows_allreduce_{max,sum} (v_permlane{32,16}_swap) against a host loop.
//   hipcc -O3 --offload-arch=gfx950 tools/scratch/permlane_swap_check.hip -o /tmp/plc && /tmp/plc   -> "bad=0"
// Written when the __builtin_amdgcn_permlane*_swap form was found to be miscompiled (see common.h).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../vllm-omni_amd/csrc/common.h"
__global__ void k(float* o, const float* in) {
  float v = in[threadIdx.x];
  o[threadIdx.x] = mi::rows_allreduce_max(v + 0.f);
  o[64 + threadIdx.x] = mi::rows_allreduce_sum(v);
}
int main() {
  float h[64], r[128], *d, *o;
  for (int i = 0; i < 64; ++i) h[i] = (float)((i * 37) % 64);
  hipMalloc(&d, 256); hipMalloc(&o, 512);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(o, d);
  hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) {
    float mx = fmaxf(fmaxf(h[i & 15], h[16 + (i & 15)]), fmaxf(h[32 + (i & 15)], h[48 + (i & 15)]));
    float sm = h[i & 15] + h[16 + (i & 15)] + h[32 + (i & 15)] + h[48 + (i & 15)];
    if (r[i] != mx || r[64 + i] != sm) { ++bad; if (bad < 6) printf("lane %d got %g %g want %g %g\n", i, r[i], r[64 + i], mx, sm); }
  }
  printf("bad=%d\n", bad);
  return bad != 0;
}
