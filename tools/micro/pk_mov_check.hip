// v_pk_mov_b32 with op_sel: which halves land where (gfx950)?  prints the three variants for a = (1, 2), b = (3, 4)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f hi_hi(v2f a, v2f b) { v2f r; asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ v2f lo_lo(v2f a, v2f b) { v2f r; asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ v2f hi_lo(v2f a, v2f b) { v2f r; asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__global__ void k(float* out)
{
    v2f a, b; a.x = 1.f + threadIdx.x; a.y = 2.f; b.x = 3.f; b.y = 4.f;
    const v2f r0 = hi_hi(a, b), r1 = lo_lo(a, b), r2 = hi_lo(a, b);
    if (threadIdx.x == 0) { out[0] = r0.x; out[1] = r0.y; out[2] = r1.x; out[3] = r1.y; out[4] = r2.x; out[5] = r2.y; }
}
int main()
{
    float* d; float h[6];
    hipMalloc(&d, 24); k<<<1, 64>>>(d); hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("hi_hi (2,4)? %g %g | lo_lo (1,3)? %g %g | hi_lo (2,3)? %g %g\n", h[0], h[1], h[2], h[3], h[4], h[5]);
    return 0;
}
