// sppm_host_static.hpp -- sppm_intersect_static and the measured-peak probe.
// Part of the one translation unit sppm_kernels.hip (included there, after the kernels; not a header to
// include anywhere else): the host side of the C ABI declared in include/sppm_hip.h.

// ================================ the entry point that allocates nothing ============================
// sppm_intersect_static: sppm_intersect for callers that cannot allocate (code under HIP graph capture, tight
// serving loops, harnesses that forbid hipMalloc).  sppm_intersect reads the scene from device memory that
// sppm_create allocated; here the sphere table and the hit counter live in a file-scope device array of fixed size.
// The table reaches it as KERNEL ARGUMENTS (chunks of SCRATCH_CHUNK spheres by value), so there is no host-sourced
// copy either: the call is a memset of the counter block and a few kernel launches on the caller's stream.
#define SPPM_STATIC_MAX_SPHERES 512
#define SCRATCH_HEAD 256  // bytes: [0] rays that hit (unsigned long long), rest reserved
#define SCRATCH_CHUNK 40  // spheres per upload launch (3520 bytes of kernel arguments)
static __device__ __align__(256) unsigned char scratch[SCRATCH_HEAD + SPPM_STATIC_MAX_SPHERES * sizeof(sppm_sphere)];
// the array's address on each device (every GPU holds its own copy): found once per device with that device
// current; host code uses these pointers, never the array's name
#define SPPM_MAX_DEVICES 64
static unsigned char* g_scratch_ptr[SPPM_MAX_DEVICES];
static std::mutex g_scratch_mutex;

struct SphereChunk { sppm_sphere s[SCRATCH_CHUNK]; };
__global__ void k_static_put_spheres(SphereChunk chunk, int first, int count, unsigned char* __restrict__ base)
{
    double* dst = (double*)(base + SCRATCH_HEAD) + (size_t)first * 11;
    const double* src = (const double*)chunk.s;
    for (int i = threadIdx.x; i < count * 11; i += blockDim.x) dst[i] = src[i];
}
// k_intersect over the table in the scratch array, plus the number of rays that hit (one atomic per wavefront)
__global__ void __launch_bounds__(256) k_static_intersect(unsigned char* __restrict__ base, int ns, const double* __restrict__ rays, int64_t n,
                                                          int32_t* __restrict__ id, double* __restrict__ out)
{
    DevScene sc;
    sc.sp = (const sppm_sphere*)(base + SCRATCH_HEAD); sc.ns = ns; sc.n_cells = 0;
    STAGE_SCENE_T(sc, 0);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool hit = false;
    if (i < n) {
        Hit h;
        hit = intersect_scene<0>(sc, ld3(rays + 6 * i), ld3(rays + 6 * i + 3), h);
        id[i] = hit ? h.object_id : -1;
        double* o = out + 7 * i;
        o[0] = h.distance;
        if (hit) { o[1] = h.position.x; o[2] = h.position.y; o[3] = h.position.z; o[4] = h.normal.x; o[5] = h.normal.y; o[6] = h.normal.z; }
        else { o[1] = o[2] = o[3] = o[4] = o[5] = o[6] = 0.0; }
    }
    const unsigned long long m = __ballot(hit);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd((unsigned long long*)base, (unsigned long long)__popcll(m));
}

static int static_scratch(unsigned char** out)
{
    int dev = -1;
    HIPCHK(hipGetDevice(&dev));
    if (dev < 0 || dev >= SPPM_MAX_DEVICES) return fail(SPPM_ERR_INVALID, "sppm_intersect_static: device index beyond " + std::to_string(SPPM_MAX_DEVICES - 1));
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        unsigned char* ptr = nullptr;
        HIPCHK(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));  // this device's copy of the array
        if (!ptr) return fail(SPPM_ERR_HIP, "sppm_intersect_static: hipGetSymbolAddress returned a null address");
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}

extern "C" int sppm_intersect_static(const sppm_sphere* h_spheres, int32_t n_spheres, const double* d_rays, int64_t n, int32_t* d_id, double* d_hit,
                                     uint64_t* d_n_hit, void* stream)
{
    if (!h_spheres || n_spheres < 1 || n < 0 || (n > 0 && (!d_rays || !d_id || !d_hit))) return fail(SPPM_ERR_INVALID, "sppm_intersect_static: bad argument");
    if (n_spheres > SPPM_STATIC_MAX_SPHERES)
        return fail(SPPM_ERR_INVALID, "sppm_intersect_static: at most " + std::to_string(SPPM_STATIC_MAX_SPHERES) + " spheres fit the static table (got " +
                                          std::to_string(n_spheres) + "); larger scenes go through sppm_create + sppm_intersect");
    if (n > (1ll << 40)) return fail(SPPM_ERR_INVALID, "sppm_intersect_static: more than 2^40 rays");
    hipStream_t st = (hipStream_t)stream;
    if (st) {  // the array used is the CURRENT device's copy: a stream of another GPU would be handed the wrong address
        int cur = -1;
        hipDevice_t sdev = -1;
        HIPCHK(hipGetDevice(&cur));
        HIPCHK(hipStreamGetDevice(st, &sdev));
        if ((int)sdev != cur) return fail(SPPM_ERR_INVALID, "sppm_intersect_static: the stream belongs to device " + std::to_string((int)sdev) + ", the current device is " + std::to_string(cur));
    }
    unsigned char* ptr = nullptr;
    int rc = static_scratch(&ptr);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(ptr, 0, SCRATCH_HEAD, st));  // the array keeps the last call's contents: the counter must start at zero
    for (int first = 0; first < n_spheres; first += SCRATCH_CHUNK) {
        SphereChunk chunk;
        const int count = std::min(SCRATCH_CHUNK, n_spheres - first);
        memset(&chunk, 0, sizeof chunk);
        memcpy(chunk.s, h_spheres + first, sizeof(sppm_sphere) * (size_t)count);
        hipLaunchKernelGGL(k_static_put_spheres, dim3(1), dim3(256), 0, st, chunk, first, count, ptr);
    }
    HIPCHK(hipGetLastError());
    if (n > 0) {
        hipLaunchKernelGGL(k_static_intersect, dim3(nblk(n, 256)), dim3(256), (size_t)n_spheres * sizeof(sppm_sphere), st, ptr, n_spheres, d_rays, n, d_id, d_hit);
        HIPCHK(hipGetLastError());
    }
    if (d_n_hit) HIPCHK(hipMemcpyAsync(d_n_hit, ptr, sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
    return SPPM_OK;
}

// ================================ measured peak (bench.py) ===========================================
// No counterpart in the reference.  The FP64 vector FMA rate this device sustains, measured beside the data-sheet
// figure (SURVEY.md 8d asks for both denominators): every thread runs 8 independent chains of fused multiply-adds.
__global__ void __launch_bounds__(256) k_fma_peak(double* __restrict__ out, int iters, double a, double b)
{
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
        x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
        x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
    }
    const double s = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
    if (s == 12345.678) out[0] = s;  // keeps the chains alive; never true for the arguments used
}
extern "C" int sppm_measure_fma_peak(void* stream, double* h_tflops)
{
    if (!h_tflops) return fail(SPPM_ERR_INVALID, "sppm_measure_fma_peak: null argument");
    hipStream_t st = (hipStream_t)stream;
    double* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, sizeof(double)));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    const int blocks = 256 * 32, iters = 4096;
    hipLaunchKernelGGL(k_fma_peak, dim3(blocks), dim3(256), 0, st, d, 64, 0.999999, 1e-7);  // warm-up
    HIPCHK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(k_fma_peak, dim3(blocks), dim3(256), 0, st, d, iters, 0.999999, 1e-7);
    HIPCHK(hipEventRecord(e1, st));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *h_tflops = 2.0 * 8.0 * (double)iters * 256.0 * (double)blocks / ((double)ms * 1e-3) / 1e12;
    hipEventDestroy(e0); hipEventDestroy(e1); hipFree(d);
    return SPPM_OK;
}
