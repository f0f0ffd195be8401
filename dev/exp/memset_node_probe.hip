// Probe for the round-2 finding: "a hipMemsetAsync captured into a hipGraph replayed a 16-byte pattern of pointer-like
// values instead of zeros once host allocations had happened between capture and replay, without intervening synchronisations".
// Standalone: no engine code.  Captures {poison kernel, memset(0), check kernel} on a stream, instantiates, then -- like an engine fork --
// allocates and frees host memory (new / vectors) and device buffers between replays, replays back to back without synchronising, and
// counts the words the check kernel found non-zero.  A non-zero count here means the runtime's memset node is at fault; zero means the
// round-2 corruption came from the engine's own host code (or from a runtime state this probe does not reach).
// Build: hipcc --offload-arch=gfx950 -O2 memset_node_probe.hip -o memset_node_probe ; run: ./memset_node_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void poison(unsigned* p, size_t n) { for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0xdeadbeefu; }
__global__ void check(const unsigned* p, size_t n, unsigned long long* bad, unsigned* first) {
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (p[i] != 0u) { if (atomicAdd(bad, 1ull) == 0) { first[0] = (unsigned)i; first[1] = p[i]; } }
}

int main() {
    const size_t sizes[] = {64, 4096, 64 * 20 * 9491};          // words: a counter block, a small state array, the [64, 20, V] log-prob tensor
    hipStream_t cap, run;
    CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&run, hipStreamNonBlocking));
    unsigned long long* bad; unsigned* first;
    CK(hipMalloc(&bad, 8)); CK(hipMalloc(&first, 8));
    CK(hipMemset(bad, 0, 8)); CK(hipMemset(first, 0, 8));
    unsigned long long total_bad = 0;
    for (size_t n : sizes) {
        unsigned* buf;
        CK(hipMalloc(&buf, n * 4));
        hipGraph_t g; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(poison, dim3(256), dim3(256), 0, cap, buf, n);
        CK(hipMemsetAsync(buf, 0, n * 4, cap));
        hipLaunchKernelGGL(check, dim3(256), dim3(256), 0, cap, buf, n, bad, first);
        CK(hipStreamEndCapture(cap, &g));
        CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        std::vector<void*> devs;
        for (int rep = 0; rep < 400; ++rep) {
            // what a fork does between capture and replay: host allocations of assorted sizes, a few device allocations
            std::vector<std::vector<char>*> junk;
            for (int j = 0; j < 32; ++j) junk.push_back(new std::vector<char>((size_t)(rand() % 65536) + 16, (char)j));
            if (rep % 16 == 0) { void* d; CK(hipMalloc(&d, 1 << 20)); devs.push_back(d); }
            CK(hipGraphLaunch(ex, run));
            CK(hipGraphLaunch(ex, run));                        // back to back, no synchronisation
            for (auto* v : junk) delete v;
        }
        CK(hipStreamSynchronize(run));
        unsigned long long hb; unsigned hf[2];
        CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hf, first, 8, hipMemcpyDeviceToHost));
        printf("memset node over %zu words, 800 replays with host / device allocations in between: %llu non-zero words seen", n, hb);
        if (hb) printf(" (first: word %u = 0x%08x)", hf[0], hf[1]);
        printf("\n");
        total_bad += hb;
        CK(hipMemset(bad, 0, 8));
        for (void* d : devs) CK(hipFree(d));
        CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g)); CK(hipFree(buf));
    }
    printf("%s\n", total_bad ? "REPRODUCED: the captured memset node wrote non-zero data" : "not reproduced: every replayed memset node zeroed its buffer");
    return 0;
}
