#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short v4s;
__global__ void k(const short* in, short* out) {
  __shared__ short lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x;
  const int g = lane >> 4, j = lane & 15, q = j >> 2, p = j & 3;
  const short* a = lds + (4 * g + q) * 64 + 4 * p;     // group g: rows 4g..4g+3, columns 0..15
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)a);
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = r[e];
}
int main() {
  short h[64 * 64], o[256];
  for (int r = 0; r < 64; ++r) for (int c = 0; c < 64; ++c) h[r * 64 + c] = (short)(r * 100 + c);
  short *di, *dout;
  hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  for (int lane = 0; lane < 64; lane += 1) { printf("lane %2d:", lane); for (int e = 0; e < 4; ++e) printf(" %4d", o[lane * 4 + e]); printf("\n"); }
  return 0;
}
