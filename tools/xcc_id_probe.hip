// Which XCD does workgroup b of a launch run on?  HW_REG_XCC_ID per workgroup, for a launch of 256 workgroups, for the
// next launch after a launch of 3 workgroups in between, and for two launches on two streams at once.
// (Answer recorded in profiles/r04_workgroup_end_times.txt: round-robin from where the LAST dispatch of any kernel
// stopped -- blockIdx.x % 8 says nothing about the XCD once other kernels run in between.)
//   hipcc -O2 --offload-arch=gfx950 tools/xcc_id_probe.hip -o tools/xcc_id_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_xcc(unsigned* o, int spin) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) o[blockIdx.x] = x & 15u;
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
}

static void show(const char* what, const unsigned* h, int n) {
    printf("%-58s", what);
    for (int i = 0; i < n; ++i) printf("%x", h[i]);
    int same = 0; for (int i = 0; i < n; ++i) same += (h[i] & 7u) == (unsigned)((i + h[0]) & 7);
    printf("   round-robin from XCD %u: %d of %d\n", h[0], same, n);
}

int main() {
    unsigned* d; CK(hipMalloc(&d, 4 * 4096));
    unsigned h[1024];
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_xcc, dim3(256), dim3(1024), 0, 0, d, 0); CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, d, 4 * 256, hipMemcpyDeviceToHost)); show("256 workgroups of 1024 threads:", h, 64);
        hipLaunchKernelGGL(k_xcc, dim3(3), dim3(256), 0, 0, d + 1024, 0);
        hipLaunchKernelGGL(k_xcc, dim3(256), dim3(1024), 0, 0, d, 0); CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, d, 4 * 256, hipMemcpyDeviceToHost)); show("the same after a launch of 3 workgroups:", h, 64);
        hipLaunchKernelGGL(k_xcc, dim3(37), dim3(256), 0, s2, d + 1024, 200);
        hipLaunchKernelGGL(k_xcc, dim3(256), dim3(1024), 0, s1, d, 0); CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, d, 4 * 256, hipMemcpyDeviceToHost)); show("the same beside 37 workgroups on another stream:", h, 64);
    }
    int even = 0; for (int i = 0; i < 256; ++i) even += !(h[i] & 1u);
    printf("workgroups on even XCDs in the last launch: %d of 256\n", even);
    return 0;
}
