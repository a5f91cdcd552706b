// kvq_core.hip -- error state, version and device info of libkvq.so
#include <string.h>

#include "kvq_common.h"

namespace kvq {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(KVQ_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return KVQ_OK;
}

// ---- profiling ring -----------------------------------------------------------------------------------------
static hipEvent_t* g_ev = nullptr;   // 2*g_cap events: [start0, stop0, start1, stop1, ...]
static int g_cap = 0, g_count = 0;

bool prof_begin(hipStream_t st) {
    if (!g_ev || g_count >= g_cap) return false;
    (void)hipEventRecord(g_ev[2 * g_count], st);
    return true;
}
void prof_end(hipStream_t st) {
    (void)hipEventRecord(g_ev[2 * g_count + 1], st);
    ++g_count;
}

// ---- clock probe ----------------------------------------------------------------------------------------------
// One wave per workgroup stores {XCC id, s_memtime (shader-clock cycles), s_memrealtime (constant 100 MHz), hardware id}.  Two
// probes on one stream bracket a region; per XCD (the cycle counter is the XCD's own) d(memtime) / d(memrealtime) x 100 MHz is
// the shader clock the chip HELD over the region, DVFS included.  No product kernel carries a stamp (MI355X_MICROARCH.md, DVFS (6)).
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* __restrict__ out) {
    if (threadIdx.x != 0) return;
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 /* HW_REG_XCC_ID */) | (0 << 6) | ((4 - 1) << 11)) & 0xf;
    const unsigned hwid = __builtin_amdgcn_s_getreg((4 /* HW_REG_HW_ID */) | (0 << 6) | ((32 - 1) << 11));
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    const unsigned long long r = __builtin_amdgcn_s_memrealtime();
    unsigned long long* o = out + (size_t)blockIdx.x * 4;
    o[0] = xcc; o[1] = t; o[2] = r; o[3] = hwid;
}

// dst[r][:] = r < rows ? src[r][:] : 0 for r < rows_padded, in 16-byte pieces (the zero-padded operand copies of a weight-gradient
// product whose token count is not a multiple of the MFMA kernel's 64-deep k-tile)
__global__ __launch_bounds__(256) void pad_rows_kernel(const uint4* __restrict__ src, int64_t rows, int64_t row16, int64_t ld16,
                                                        uint4* __restrict__ dst, int64_t rows_padded) {
    const int64_t total = rows_padded * row16;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / row16, c = i - r * row16;
        uint4 v = {0u, 0u, 0u, 0u};
        if (r < rows) v = src[r * ld16 + c];
        dst[i] = v;
    }
}

}  // namespace kvq

extern "C" {

int kvq_pad_rows(const void* src, int64_t rows, int64_t row_bytes, int64_t src_ld_bytes, void* dst, int64_t rows_padded, void* stream) {
    using namespace kvq;
    KVQ_REQUIRE(src && dst && rows >= 0 && rows_padded >= rows && row_bytes > 0, "kvq_pad_rows: bad argument");
    KVQ_REQUIRE(row_bytes % 16 == 0 && src_ld_bytes % 16 == 0 && src_ld_bytes >= row_bytes && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0,
                "kvq_pad_rows: rows, row stride and both buffers in multiples of 16 bytes");
    const int64_t total = rows_padded * (row_bytes / 16);
    if (total == 0) return KVQ_OK;
    const unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(pad_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint4*>(src), rows,
                       row_bytes / 16, src_ld_bytes / 16, reinterpret_cast<uint4*>(dst), rows_padded);
    return check_launch("pad_rows_kernel");
}

int kvq_graph_census(void* graph, int64_t* counts_host) {
    using namespace kvq;
    KVQ_REQUIRE(graph && counts_host, "kvq_graph_census: null pointer argument");
    size_t n = 0;
    if (hipGraphGetNodes((hipGraph_t)graph, nullptr, &n) != hipSuccess) return fail(KVQ_E_LAUNCH, "hipGraphGetNodes (count) failed");
    for (int i = 0; i < KVQ_GRAPH_NODE_KINDS; ++i) counts_host[i] = 0;
    if (n == 0) return KVQ_OK;
    hipGraphNode_t* nodes = new hipGraphNode_t[n];
    int rc = KVQ_OK;
    if (hipGraphGetNodes((hipGraph_t)graph, nodes, &n) != hipSuccess) rc = fail(KVQ_E_LAUNCH, "hipGraphGetNodes failed");
    for (size_t i = 0; rc == KVQ_OK && i < n; ++i) {
        hipGraphNodeType t;
        if (hipGraphNodeGetType(nodes[i], &t) != hipSuccess) { rc = fail(KVQ_E_LAUNCH, "hipGraphNodeGetType failed"); break; }
        int slot = KVQ_GRAPH_NODE_OTHER;
        if (t == hipGraphNodeTypeKernel) slot = KVQ_GRAPH_NODE_KERNEL;
        else if (t == hipGraphNodeTypeMemset) slot = KVQ_GRAPH_NODE_MEMSET;
        else if (t == hipGraphNodeTypeMemcpy) slot = KVQ_GRAPH_NODE_MEMCPY;
        else if (t == hipGraphNodeTypeEmpty) slot = KVQ_GRAPH_NODE_EMPTY;
        else if (t == hipGraphNodeTypeEventRecord || t == hipGraphNodeTypeWaitEvent) slot = KVQ_GRAPH_NODE_EVENT;
        ++counts_host[slot];
    }
    delete[] nodes;
    return rc;
}

int kvq_clock_probe_rows(void) { return 64; }

int kvq_clock_probe(uint64_t* out, size_t out_bytes, void* stream) {
    using namespace kvq;
    KVQ_REQUIRE(out, "kvq_clock_probe: null pointer argument");
    if (out_bytes < (size_t)64 * 4 * sizeof(uint64_t)) return fail(KVQ_E_WORKSPACE, "kvq_clock_probe: %zu bytes < %zu", out_bytes, (size_t)64 * 32);
    hipLaunchKernelGGL(clock_probe_kernel, dim3(64), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<unsigned long long*>(out));
    return check_launch("clock_probe_kernel");
}

int kvq_prof_enable(int n_pairs) {
    using namespace kvq;
    if (g_ev) {
        for (int i = 0; i < 2 * g_cap; ++i) (void)hipEventDestroy(g_ev[i]);
        delete[] g_ev;
        g_ev = nullptr;
    }
    g_cap = g_count = 0;
    if (n_pairs <= 0) return KVQ_OK;
    g_ev = new hipEvent_t[2 * (size_t)n_pairs];
    for (int i = 0; i < 2 * n_pairs; ++i)
        if (hipEventCreate(&g_ev[i]) != hipSuccess) return fail(KVQ_E_LAUNCH, "hipEventCreate failed");
    g_cap = n_pairs;
    return KVQ_OK;
}

int kvq_prof_read(float* ms_host, int max) {
    using namespace kvq;
    int n = 0;
    for (; n < g_count && n < max; ++n) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_ev[2 * n], g_ev[2 * n + 1]) != hipSuccess)
            return fail(KVQ_E_LAUNCH, "kvq_prof_read: events not complete (synchronise the stream first)");
        ms_host[n] = ms;
    }
    g_count = 0;
    return n;
}

int kvq_version(void) { return KVQ_VERSION; }

const char* kvq_last_error(void) { return kvq::g_err; }

int kvq_device_info(int* cu_count, char* name, size_t name_len) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return kvq::fail(KVQ_E_NODEVICE, "hipGetDevice failed: no usable HIP device");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return kvq::fail(KVQ_E_NODEVICE, "hipGetDeviceProperties failed");
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (name && name_len) {
        strncpy(name, prop.gcnArchName, name_len - 1);
        name[name_len - 1] = 0;
    }
    return KVQ_OK;
}

}  // extern "C"
