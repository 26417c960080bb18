#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <chrono>
__device__ inline uint32_t umed3(uint32_t a, uint32_t b, uint32_t c){uint32_t r; asm("v_med3_u32 %0, %1, %2, %3":"=v"(r):"v"(a),"v"(b),"v"(c)); return r;}
template<int V> __global__ __launch_bounds__(64) void kern(uint32_t* out, int iters, uint32_t seed){
  uint32_t L[18];
  for(int s=0;s<18;s++) L[s]=seed*(s+1)+threadIdx.x;
  uint32_t key=seed^threadIdx.x;
  for(int i=0;i<iters;i++){
    key = key*1664525u+1013904223u;
    if(V==0){
#pragma unroll
      for(int s=17;s>0;s--) L[s]=umed3(L[s-1],key,L[s]);
      L[0]=min(L[0],key);
    } else if(V==1){
#pragma unroll
      for(int s=17;s>0;s--) L[s]=max(L[s-1],min(key,L[s]));
      L[0]=min(L[0],key);
    } else if (V==2){
#pragma unroll
      for(int s=17;s>0;s--) L[s]=min(L[s]+1u,key);   // 2 simple ops? add+min
      L[0]=min(L[0],key);
    } else {
#pragma unroll
      for(int s=17;s>0;s--) L[s]=min(L[s],key);   // 1 simple VOP2 per slot
      L[0]=min(L[0],key);
    }
  }
  uint32_t acc=0; for(int s=0;s<18;s++) acc^=L[s];
  out[blockIdx.x*64+threadIdx.x]=acc;
}
template<int V> double run(uint32_t* d,int blocks,int iters){
  hipLaunchKernelGGL(kern<V>,dim3(blocks),dim3(64),0,0,d,iters,123u); hipDeviceSynchronize();
  auto t0=std::chrono::high_resolution_clock::now();
  hipLaunchKernelGGL(kern<V>,dim3(blocks),dim3(64),0,0,d,iters,123u); hipDeviceSynchronize();
  auto t1=std::chrono::high_resolution_clock::now();
  return std::chrono::duration<double>(t1-t0).count();
}
int main(){
  uint32_t* d; hipMalloc(&d, 1<<24);
  int iters=20000;
  for(int wps: {1,2,4,8}){
    int blocks=256*4*wps;
    double a=run<0>(d,blocks,iters), b=run<1>(d,blocks,iters), c=run<2>(d,blocks,iters), e=run<3>(d,blocks,iters);
    double ops=(double)wps*iters*18; // per SIMD slot-ops
    printf("waves/SIMD %d: med3 %.2f cyc/op  minmax %.2f cyc/slot  addmin %.2f  min1 %.2f (at 2.4GHz)\n",wps,a*2.4e9/ops,b*2.4e9/ops,c*2.4e9/ops,e*2.4e9/ops);
  }
  return 0;
}
