// How the dispatcher places the workgroups of a small launch: packed onto few CUs or spread over all of them?  A launch of W workgroups of
// T threads (V VGPRs' worth of occupancy is not modelled: the kernel is tiny; LDS bytes per workgroup given) where every workgroup notes the
// CU it runs on (HW_ID: XCC, SE, CU) and then idles ~10 us, so that all of them are resident together.  Prints the number of distinct CUs and the
// histogram of workgroups per CU.  A row-block workgroup of another decode needs a whole CU, so the CUs a small launch touches are what it costs.
// Build: hipcc --offload-arch=gfx950 -O3 cu_spread_probe.hip -o cu_spread_probe ; run: ./cu_spread_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void note_cu(uint32_t* out, int spin) {
    extern __shared__ unsigned char lds[];
    if (threadIdx.x == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x] = ((xcc & 0xf) << 16) | (hw & 0xffff);          // HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
        lds[0] = (unsigned char)hw;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
}

int main() {
    uint32_t* out;
    CHECK(hipMalloc(&out, 4096 * 4));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&note_cu), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    struct Cfg { int wgs, threads, lds; const char* what; };
    const Cfg cfgs[] = {{320, 512, 24 * 1024, "tail: 320 x 512 threads, 24 KB"},   {320, 256, 20 * 1024, "query-attention: 320 x 256, 20 KB"},
                        {160, 256, 64 * 1024, "row GEMM, 64-column tile: 160 x 256, 64 KB"}, {160, 256, 16 * 1024, "row GEMM, 16-column tile: 160 x 256, 16 KB"},
                        {64, 256, 16 * 1024, "64 x 256"},  {640, 256, 16 * 1024, "640 x 256, 16 KB"}, {180, 512, 150 * 1024, "row-block kernel: 180 x 512, 150 KB"}};
    for (const Cfg& c : cfgs) {
        CHECK(hipMemset(out, 0xff, 4096 * 4));
        hipLaunchKernelGGL(note_cu, dim3(c.wgs), dim3(c.threads), c.lds, 0, out, 23000);      // ~10 us at 2.3 GHz
        CHECK(hipDeviceSynchronize());
        std::vector<uint32_t> h(c.wgs);
        CHECK(hipMemcpy(h.data(), out, c.wgs * 4, hipMemcpyDeviceToHost));
        std::map<uint32_t, int> per_cu;
        for (uint32_t v : h) per_cu[((v >> 16) << 16) | (v & 0xff00)]++;                     // (xcc, se, sh, cu)
        std::map<int, int> hist;
        for (auto& kv : per_cu) hist[kv.second]++;
        printf("%-56s -> %3zu distinct CUs; workgroups per CU:", c.what, per_cu.size());
        for (auto& kv : hist) printf("  %d x%d", kv.first, kv.second);
        printf("\n");
    }
    return 0;
}
