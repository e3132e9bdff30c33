// Does a process that makes ONE cooperative launch fault in its exit handlers under rocprofv3 -- with no torch and no libvjf_hip.so in it?
//   hipcc --offload-arch=gfx950 -O2 -o tools/coop_exit_repro tools/coop_exit_repro.hip
//   rocprofv3 --kernel-trace -d <dir> -- tools/coop_exit_repro coop     (or: plain)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__global__ void k(int* p) { if (threadIdx.x == 0) atomicAdd(p, 1); }
int main(int argc, char** argv) {
    const bool coop = argc > 1 && !strcmp(argv[1], "coop");
    int* d = nullptr;
    if (hipMalloc(&d, 4) != hipSuccess || hipMemset(d, 0, 4) != hipSuccess) return 2;
    void* args[] = {(void*)&d};
    hipError_t e = coop ? hipLaunchCooperativeKernel((const void*)k, dim3(64), dim3(64), args, 0, nullptr)
                        : hipLaunchKernel((const void*)k, dim3(64), dim3(64), args, 0, nullptr);
    int h = -1;
    if (e == hipSuccess) e = hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("%s launch: %s, count %d\n", coop ? "cooperative" : "plain", hipGetErrorString(e), h);
    (void)hipFree(d);
    return e == hipSuccess && h == 64 ? 0 : 1;
}
