#include <hip/hip_runtime.h>
__device__ __forceinline__ double lane_from_left(double x) {   // value of lane-1
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_from_right(double x) {  // value of lane+1
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__global__ void k(const double* a, double* l, double* r, double* l2, double* r2) {
  int i = threadIdx.x; double x = a[i];
  l[i] = lane_from_left(x); r[i] = lane_from_right(x);
  l2[i] = __shfl_up(x, 1, 64); r2[i] = __shfl_down(x, 1, 64);
}
int main(){ double *a,*l,*r,*l2,*r2; hipMalloc(&a,512);hipMalloc(&l,512);hipMalloc(&r,512);hipMalloc(&l2,512);hipMalloc(&r2,512);
 double h[64]; for(int i=0;i<64;i++)h[i]=i+0.5; hipMemcpy(a,h,512,hipMemcpyHostToDevice); k<<<1,64>>>(a,l,r,l2,r2);
 double hl[64],hr[64],hl2[64],hr2[64]; hipMemcpy(hl,l,512,hipMemcpyDeviceToHost); hipMemcpy(hr,r,512,hipMemcpyDeviceToHost);hipMemcpy(hl2,l2,512,hipMemcpyDeviceToHost); hipMemcpy(hr2,r2,512,hipMemcpyDeviceToHost);
 for(int i : {0,1,15,16,31,32,62,63}) printf("lane %d: dppL %.1f dppR %.1f shflUp %.1f shflDown %.1f\n", i, hl[i],hr[i],hl2[i],hr2[i]); return 0; }
