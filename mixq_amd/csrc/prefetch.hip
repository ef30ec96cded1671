// mixq_prefetch: pull a device buffer (a layer's weight image) towards the compute units AHEAD of the kernel that will stream it - into the
// MI355X's 256 MB memory-side cache - from a side stream, while another kernel computes.  Why: a GEMM of this library whose weights sit in
// that cache runs its k loop 20 % faster than one whose weights come from HBM (529 vs 633 cycles per k-step, profiles/r04_gemm_trace_warm_cold.txt:
// a CU's window of outstanding requests turns over with the latency of what is behind it), a model streams every layer's weights from HBM
// (7 GB >> 256 MB), and nothing issued from INSIDE the GEMM hides that (deeper rings, touches by its own loader waves: NOTEBOOK.md).  A
// model runner calls this for GEMM j+1's image when it enqueues GEMM j (mixq_amd: layer.prefetch_weights(stream)); the images are k-major,
// so a linear walk runs ahead of the GEMM's own consumption even when it has not finished by the time that GEMM starts.
// One dword per 128-byte line and lane; a handful of light workgroups (no LDS, 16 registers: they fit beside a resident GEMM workgroup).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void prefetch_kernel(const uint32_t* __restrict__ p, size_t lines, uint32_t* __restrict__ sink)
{
    const size_t stride = static_cast<size_t>(gridDim.x) * 256;
    uint32_t acc = 0;
    for (size_t l = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; l < lines; l += stride) acc ^= p[l * 32];
    if (acc == 0x9e3779b9u && sink) *sink = acc;          // (keeps the loads; never true for real weight bytes in practice, harmless if it is)
}

}  // namespace

extern "C" int mixq_prefetch(const void* ptr, long long bytes, mixq_stream_t stream)
{
    if (bytes < 0 || (!ptr && bytes)) return MIXQ_EINVAL;
    if ((reinterpret_cast<uintptr_t>(ptr) & 3) != 0) return MIXQ_EINVAL;
    const size_t lines = static_cast<size_t>(bytes) / 128;
    if (!lines) return MIXQ_OK;
    static uint32_t* sink = nullptr;                      // (never written in practice; one word per process)
    const size_t want = (lines + 255) / 256;
    const unsigned grid = static_cast<unsigned>(want < 128 ? want : 128);
    hipLaunchKernelGGL(prefetch_kernel, dim3(grid), dim3(256), 0, mixq_stream(stream), static_cast<const uint32_t*>(ptr), lines, sink);
    return mixq_launch_status();
}
