// Micro-benchmark: can the HBM domain of an allocation be steered without holding spacer memory?
// The front end is ~12 % slower when the buffer it reads (IQ, 25.8 GB) and the buffer it writes (soft bits, 3.8 GB)
// share one of the three HBM domains (profiles/r02_hbm_domains.txt).  Round 2 got them apart by allocating four
// candidate pairs (118 GB) and timing every combination.  Here: one input buffer, and output buffers obtained in
// different ways -- plain hipMalloc right behind it, the hipExtMallocWithFlags kinds, a stream-ordered pool
// allocation, physical chunks mapped through the virtual-memory API, hipMalloc behind a spacer -- each timed with the
// data mover of the front end's geometry.  Second part: a chunk-level domain map of the first N GB a fresh process
// gets through hipMemCreate (2 GB chunks, small mover reading chunk i and writing chunk j).
// build: hipcc -O3 --offload-arch=gfx950 alloc_kinds.hip -o alloc_kinds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned v4u __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mover(const char *in, char *out, int n_chunks, int cpw) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    v4u acc = {1u, 2u, 3u, 4u};
    for (int c = 0; c < cpw; c++) {
        const int chunk = wave * cpw + c;
        if (chunk >= n_chunks) break;
        const v4u *p = reinterpret_cast<const v4u *>(in + size_t(chunk) * 20480) + lane;
        v4u v[20];
#pragma unroll
        for (int i = 0; i < 20; i++) v[i] = __builtin_nontemporal_load(p + 64 * i);
#pragma unroll
        for (int i = 0; i < 20; i++) acc += v[i];
        v4u *o = reinterpret_cast<v4u *>(out + size_t(chunk) * 3072) + lane;
#pragma unroll
        for (int i = 0; i < 3; i++) __builtin_nontemporal_store(acc, o + 64 * i);
    }
}

__global__ void fill_noise(unsigned *p, size_t n) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        unsigned x = unsigned(i) * 2654435761u + unsigned(i >> 32) * 40503u + 12345u;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = (x & 0x807fffffu) | 0x3f000000u;
    }
}

static hipEvent_t e0, e1;
static float run(const char *in, char *out, int n_chunks, int cpw = 25) {
    const unsigned grid = unsigned(((n_chunks + cpw - 1) / cpw + 3) / 4);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mover, dim3(grid), dim3(256), 51 * 1024, 0, in, out, n_chunks, cpw);
        hipEventRecord(e1);
        if (hipEventSynchronize(e1) != hipSuccess) return -1.f;
        float t; hipEventElapsedTime(&t, e0, e1);
        if (rep > 0 && t < best) best = t;
    }
    return best;
}

static void report(const char *what, const char *in, char *out, int n_chunks) {
    if (!out) { printf("%-58s not available (%s)\n", what, hipGetErrorString(hipGetLastError())); return; }
    printf("%-58s %p  %.3f ms\n", what, (void *)out, run(in, out, n_chunks));
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int map_gb = argc > 1 ? atoi(argv[1]) : 96;
    const int n_chunks = 16384 * 76;
    const size_t in_bytes = size_t(n_chunks) * 20480, out_bytes = size_t(n_chunks) * 3072;
    hipEventCreate(&e0); hipEventCreate(&e1);
    size_t free_b = 0, total_b = 0;
    hipMemGetInfo(&free_b, &total_b);
    printf("free %.1f GB of %.1f GB\n", free_b / 1e9, total_b / 1e9);
    char *in = nullptr;
    if (hipMalloc(&in, in_bytes) != hipSuccess) { printf("input alloc failed\n"); return 1; }
    hipLaunchKernelGGL(fill_noise, dim3(4096), dim3(256), 0, 0, reinterpret_cast<unsigned *>(in), in_bytes / 4);
    printf("input %p (%.1f GB, hipMalloc, first allocation of the process)\n", (void *)in, in_bytes / 1e9);
    char *o;
    o = nullptr; (void)hipMalloc(&o, out_bytes); report("hipMalloc right behind it", in, o, n_chunks); char *o_plain = o;
    o = nullptr; (void)hipExtMallocWithFlags(reinterpret_cast<void **>(&o), out_bytes, hipDeviceMallocFinegrained); report("hipExtMallocWithFlags(Finegrained)", in, o, n_chunks); if (o) hipFree(o);
    o = nullptr; (void)hipExtMallocWithFlags(reinterpret_cast<void **>(&o), out_bytes, hipDeviceMallocUncached); report("hipExtMallocWithFlags(Uncached)", in, o, n_chunks); if (o) hipFree(o);
    o = nullptr; (void)hipExtMallocWithFlags(reinterpret_cast<void **>(&o), out_bytes, hipDeviceMallocContiguous); report("hipExtMallocWithFlags(Contiguous)", in, o, n_chunks); if (o) hipFree(o);
    o = nullptr; (void)hipMallocAsync(reinterpret_cast<void **>(&o), out_bytes, 0); hipStreamSynchronize(0); report("hipMallocAsync (default pool)", in, o, n_chunks); if (o) { hipFreeAsync(o, 0); hipStreamSynchronize(0); }
    {   // one physical allocation through the virtual-memory API
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
        const size_t sz = (out_bytes + gran - 1) / gran * gran;
        hipMemGenericAllocationHandle_t h;
        void *va = nullptr;
        if (hipMemCreate(&h, sz, &prop, 0) == hipSuccess && hipMemAddressReserve(&va, sz, 0, nullptr, 0) == hipSuccess &&
            hipMemMap(va, sz, 0, h, 0) == hipSuccess) {
            hipMemAccessDesc acc = {};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            hipMemSetAccess(va, sz, &acc, 1);
            printf("(vmm granularity %zu)\n", gran);
            report("hipMemCreate + hipMemMap", in, static_cast<char *>(va), n_chunks);
            hipMemUnmap(va, sz); hipMemRelease(h); hipMemAddressFree(va, sz);
        } else {
            report("hipMemCreate + hipMemMap", in, nullptr, n_chunks);
        }
    }
    // behind spacers of growing size (held only while the candidate is allocated)
    for (int gb : {8, 16, 32, 48, 64, 80}) {
        char *sp = nullptr, *c = nullptr;
        if (hipMalloc(&sp, size_t(gb) << 30) != hipSuccess) { printf("spacer %d GB failed\n", gb); continue; }
        (void)hipMalloc(&c, out_bytes);
        hipFree(sp);
        char what[96];
        snprintf(what, sizeof what, "hipMalloc behind a %d GB spacer (spacer freed)", gb);
        report(what, in, c, n_chunks);
        if (c) hipFree(c);
    }
    // free + re-allocate the plain one: does the driver hand back the same place?
    if (o_plain) hipFree(o_plain);
    o = nullptr; (void)hipMalloc(&o, out_bytes); report("hipMalloc again after freeing everything but the input", in, o, n_chunks); if (o) hipFree(o);
    // the input's own tail as output (same allocation): the worst case by construction
    hipFree(in);

    // ---- part 2: chunk-level domain map of what a fresh process is handed first ----
    {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        const size_t chunk = size_t(2) << 30;
        const int n = map_gb / 2;
        std::vector<hipMemGenericAllocationHandle_t> hs;
        void *va = nullptr;
        if (hipMemAddressReserve(&va, chunk * n, 0, nullptr, 0) != hipSuccess) { printf("reserve failed\n"); return 0; }
        for (int i = 0; i < n; i++) {
            hipMemGenericAllocationHandle_t h;
            if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) break;
            if (hipMemMap(static_cast<char *>(va) + chunk * i, chunk, 0, h, 0) != hipSuccess) { hipMemRelease(h); break; }
            hs.push_back(h);
        }
        const int m = int(hs.size());
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        hipMemSetAccess(va, chunk * m, &acc, 1);
        hipLaunchKernelGGL(fill_noise, dim3(4096), dim3(256), 0, 0, reinterpret_cast<unsigned *>(va), chunk * m / 4);
        hipDeviceSynchronize();
        printf("\n%d chunks of 2 GB through hipMemCreate, mapped in allocation order; us per launch of a mover reading chunk i "
               "(2 GB) and writing the first 315 MB of chunk j; row = i, column = j (every 2nd chunk)\n     ", m);
        const int small = int(chunk / 20480);
        for (int j = 0; j < m; j += 2) printf(" %4d", 2 * j);
        printf("\n");
        for (int i = 0; i < m; i += 2) {
            printf("%4d ", 2 * i);
            for (int j = 0; j < m; j += 2) {
                if (i == j) { printf("    -"); continue; }
                printf(" %4.0f", 1e3f * run(static_cast<char *>(va) + chunk * i, static_cast<char *>(va) + chunk * j, small, 4));
            }
            printf("\n");
            fflush(stdout);
        }
    }
    return 0;
}
