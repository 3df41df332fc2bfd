#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, float* out, int n, int shift) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, n*4, 0x00020000);
  f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (threadIdx.x*2 + shift)*4, 0, 0));
  out[threadIdx.x*4+0]=v.x; out[threadIdx.x*4+1]=v.y; out[threadIdx.x*4+2]=v.z; out[threadIdx.x*4+3]=v.w;
}
int main(){
  const int n=256; float h[n]; for(int i=0;i<n;i++) h[i]=i;
  float *a,*o; hipMalloc(&a,n*4); hipMalloc(&o,64*4*4); hipMemcpy(a,h,n*4,hipMemcpyHostToDevice);
  for (int shift : {0,1,127,128,129,130,131}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, o, n, shift);
    hipError_t e = hipDeviceSynchronize();
    float r[256]; hipMemcpy(r,o,sizeof(r),hipMemcpyDeviceToHost);
    printf("shift %d err %d: lane0 %g %g %g %g  lane63 %g %g %g %g\n", shift, (int)e, r[0],r[1],r[2],r[3], r[252],r[253],r[254],r[255]);
  }
  return 0;
}
