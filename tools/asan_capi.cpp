// Host-side sanitizer pass over the C ABI: a small driver, built - like the library it links
// (mixq_amd/libmixq_hip_asan.so: the product sources with -fsanitize=address -fno-gpu-sanitize) - with AddressSanitizer on the HOST
// code.  The Python layer hands raw addresses across this boundary; nothing else checks the C side's host code (argument validation,
// the plan / workspace / per-device tables, launch-geometry arithmetic) for out-of-bounds accesses or use-after-free.
// What it walks: identification, every quantise / detect / dequant / re-tile entry point incl. empty inputs and rejected arguments, the
// kept-outlier-map routes, the fused GEMMs on every operand layout, mixq_linear_forward on a KEPT argument block (re-used, then edited),
// the RMSNorm entry points, the W8A16 pair - each followed by a device synchronise, results cross-checked where a second route exists.
//   build + run: tools/asan_capi.sh (on the GPU box)            exit code 0 and no ASAN report = clean
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../include/mixq_hip.h"

extern "C" int mixq_gemm_amax_supported(int, int, int, int);

static int g_fail = 0;
#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_fail; } } while (0)
#define HIPOK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); exit(2); } } while (0)

template <class T> struct Dev {
    T* p = nullptr; size_t n = 0;
    explicit Dev(size_t n_) : n(n_) { HIPOK(hipMalloc(reinterpret_cast<void**>(&p), (n ? n : 1) * sizeof(T))); HIPOK(hipMemset(p, 0, (n ? n : 1) * sizeof(T))); }
    ~Dev() { (void)hipFree(p); }
    void up(const std::vector<T>& h) { HIPOK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); }
    std::vector<T> down() const { std::vector<T> h(n); HIPOK(hipMemcpy(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost)); return h; }
};

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }
static uint16_t f2h(float f) { _Float16 h = static_cast<_Float16>(f); uint16_t b; memcpy(&b, &h, 2); return b; }
static float h2f(uint16_t b) { _Float16 h; memcpy(&h, &b, 2); return static_cast<float>(h); }

// the kept outlier map of include/mixq_hip.h (mixq_quant_fused_masked) for the first n entries of ind
static std::vector<uint32_t> kept_map(const std::vector<int32_t>& ind, int n, int K) {
    const int W = (K + 31) / 32, head = (W + 1 + 3) / 4 * 4;
    std::vector<uint32_t> m(head + (K + 7) / 8 * 4, 0u);
    std::vector<uint16_t> pos((K + 7) / 8 * 8, 0xffffu);                       // per-column AND-masks: 0 for an outlier column
    for (int j = 0; j < n; ++j) { m[ind[j] >> 5] |= 1u << (ind[j] & 31); pos[ind[j]] = 0; }
    m[W] = static_cast<uint32_t>(n);
    memcpy(m.data() + head, pos.data(), pos.size() * 2);
    return m;
}

int main() {
    char info[64];
    CHECK(mixq_version() > 0);
    CHECK(mixq_device_info(info, 8) == MIXQ_OK && strlen(info) < 8);          // (truncated, terminated)
    CHECK(mixq_device_info(info, sizeof(info)) == MIXQ_OK);
    CHECK(mixq_device_info(nullptr, 16) != MIXQ_OK);
    printf("device: %s\n", info);

    const int M = 100, K = 512, N = 264, n = 5, cap = 16;
    std::vector<uint16_t> hx(static_cast<size_t>(M) * K);
    for (auto& v : hx) v = f2h((static_cast<int>(rnd() % 2001) - 1000) / 250.f);
    std::vector<int32_t> hind = {7, 300, 64, 511, 33};                         // unsorted, as `ind` is after an append
    hind.resize(cap, K - 1);
    for (int r = 0; r < M; ++r) for (int j = 0; j < n; ++j) hx[static_cast<size_t>(r) * K + hind[j]] = f2h(40.f + r);
    Dev<uint16_t> x(hx.size()), x2(hx.size()), xs(M), xs2(M), xo(static_cast<size_t>(M) * cap), xo2(static_cast<size_t>(M) * cap), out(hx.size());
    Dev<int32_t> ind(cap), ndev(1), flag(1), cnt(1), indout(K);
    Dev<uint8_t> q(static_cast<size_t>(112) * K), q2(static_cast<size_t>(112) * K), colflags(K);
    ind.up(hind); ndev.up({n});
    const std::vector<uint32_t> hmap = kept_map(hind, n, K);
    Dev<uint32_t> map(hmap.size()); map.up(hmap);
    const int map_words = static_cast<int>(hmap.size());
    CHECK(map_words == mixq_kept_map_words(K));

    // ---- quantise passes: plain / packed, empty, rejected -------------------------------------------------------------------------------
    x.up(hx);
    CHECK(mixq_find_row_scale(x.p, xs.p, q.p, M, K, K, 8, MIXQ_FMT_PLAIN, nullptr) == MIXQ_OK);
    CHECK(mixq_find_row_scale(x.p, xs.p, q.p, M, K, K, 8, MIXQ_FMT_P16X64, nullptr) == MIXQ_OK);
    CHECK(mixq_find_row_scale(x.p, xs.p, q.p, M, K, K, 4, MIXQ_FMT_PLAIN, nullptr) == MIXQ_OK);
    CHECK(mixq_find_row_scale(x.p, xs.p, q.p, 0, K, K, 8, MIXQ_FMT_PLAIN, nullptr) == MIXQ_OK);
    CHECK(mixq_find_row_scale(nullptr, nullptr, nullptr, 0, K, K, 8, MIXQ_FMT_PLAIN, nullptr) == MIXQ_OK);   // empty inputs may carry null pointers
    CHECK(mixq_find_row_scale(nullptr, xs.p, q.p, M, K, K, 8, MIXQ_FMT_PLAIN, nullptr) == MIXQ_EINVAL);
    CHECK(mixq_find_row_scale(x.p, xs.p, q.p, M, K, K, 5, MIXQ_FMT_PLAIN, nullptr) == MIXQ_EINVAL);
    CHECK(mixq_find_row_scale(x.p, xs.p, q.p, M, 100, 104, 8, MIXQ_FMT_PLAIN, nullptr) == MIXQ_ESHAPE);
    CHECK(mixq_find_row_scale(x.p, xs.p, q.p, M, K, K, 8, 9, nullptr) == MIXQ_EINVAL);
    HIPOK(hipDeviceSynchronize());
    // the in-kernel mask route against the kept-map route, device count below the capacity of `ind`
    x.up(hx); x2.up(hx);
    CHECK(mixq_quant_fused(x.p, ind.p, cap, ndev.p, xs.p, q.p, xo.p, flag.p, M, K, K, cap, 8, 6.f, MIXQ_FMT_P16X64, nullptr) == MIXQ_OK);
    CHECK(mixq_quant_fused_masked(x2.p, ind.p, cap, ndev.p, map.p, map_words, xs2.p, q2.p, xo2.p, flag.p, M, K, K, cap, 8, 6.f, MIXQ_FMT_P16X64, nullptr) == MIXQ_OK);
    HIPOK(hipDeviceSynchronize());
    CHECK(x.down() == x2.down() && xs.down() == xs2.down() && xo.down() == xo2.down());
    {   // (the packed images are block-major and their pad rows are never written: compare the M rows of the plain matrices)
        Dev<uint8_t> pa(static_cast<size_t>(M) * K), pb(static_cast<size_t>(M) * K);
        CHECK(mixq_unpack_operand(q.p, pa.p, M, K, MIXQ_FMT_P16X64, nullptr) == MIXQ_OK);
        CHECK(mixq_unpack_operand(q2.p, pb.p, M, K, MIXQ_FMT_P16X64, nullptr) == MIXQ_OK);
        HIPOK(hipDeviceSynchronize());
        CHECK(pa.down() == pb.down());
    }
    CHECK(mixq_quant_fused_masked(x2.p, ind.p, cap, ndev.p, map.p, (K + 31) / 32 + 1, xs2.p, q2.p, xo2.p, flag.p, M, K, K, cap, 8, 6.f, MIXQ_FMT_P16X64, nullptr) == MIXQ_EINVAL);   // (round 4's layout: bits + count only)
    CHECK(mixq_quant_fused_masked(x2.p, ind.p, cap, ndev.p, nullptr, 0, xs2.p, q2.p, xo2.p, flag.p, M, K, K, cap, 8, 6.f, MIXQ_FMT_P16X64, nullptr) == MIXQ_EINVAL);
    CHECK(mixq_quant_fused(x.p, ind.p, n, nullptr, xs.p, q.p, nullptr, flag.p, M, K, K, cap, 8, 6.f, MIXQ_FMT_PLAIN, nullptr) == MIXQ_EINVAL);   // n > 0 needs x_out
    CHECK(mixq_quant_fused(x.p, ind.p, n, nullptr, xs.p, q.p, xo.p, flag.p, M, K, K, 3, 8, 6.f, MIXQ_FMT_PLAIN, nullptr) == MIXQ_EINVAL);        // ldo < n
    CHECK(mixq_quant_fused(x.p, nullptr, 0, nullptr, xs.p, q.p, nullptr, nullptr, M, K, K, 0, 4, 6.f, MIXQ_FMT_P16X64, nullptr) == MIXQ_OK);
    x.up(hx);
    CHECK(mixq_extract_outliers_zero(x.p, ind.p, n, xo.p, M, K, K, cap, nullptr) == MIXQ_OK);
    CHECK(mixq_extract_outliers_zero(x.p, ind.p, n, xo.p, M, K, K, 2, nullptr) == MIXQ_EINVAL);
    x.up(hx);
    CHECK(mixq_detect_outlier_cols(x.p, 6.f, colflags.p, indout.p, cnt.p, M, K, K, nullptr) == MIXQ_OK);
    HIPOK(hipDeviceSynchronize());
    CHECK(cnt.down()[0] == n);
    CHECK(mixq_detect_outlier_cols(x.p, 6.f, nullptr, indout.p, cnt.p, M, K, K, nullptr) == MIXQ_EINVAL);

    // ---- weights: quantised matrix, re-tiling both ways, column dequantisation ---------------------------------------------------------
    std::vector<int8_t> hw(static_cast<size_t>(N) * K);
    for (auto& v : hw) v = static_cast<int8_t>(static_cast<int>(rnd() % 255) - 127);
    std::vector<uint16_t> hsw(N), hbias(N);
    for (int i = 0; i < N; ++i) { hsw[i] = f2h(0.001f + (rnd() % 100) * 1e-5f); hbias[i] = f2h((static_cast<int>(rnd() % 200) - 100) / 100.f); }
    const int N16 = (N + 15) / 16 * 16;
    Dev<int8_t> w(hw.size()), wback(hw.size());
    Dev<uint8_t> wp(static_cast<size_t>(N16) * K), wf(static_cast<size_t>(N16) * K);
    Dev<uint16_t> sw(N), bias(N), wo(static_cast<size_t>(N) * cap), y(static_cast<size_t>(M) * N), y2(static_cast<size_t>(M) * N), y3(static_cast<size_t>(M) * N);
    w.up(hw); sw.up(hsw); bias.up(hbias);
    CHECK(mixq_pack_operand(w.p, wp.p, N, K, MIXQ_FMT_P16X64, nullptr) == MIXQ_OK);
    CHECK(mixq_pack_operand(w.p, wf.p, N, K, MIXQ_FMT_F16X64, nullptr) == MIXQ_OK);
    CHECK(mixq_unpack_operand(wf.p, wback.p, N, K, MIXQ_FMT_F16X64, nullptr) == MIXQ_OK);
    HIPOK(hipDeviceSynchronize());
    CHECK(wback.down() == hw);
    CHECK(mixq_pack_operand(w.p, wp.p, N, 100, MIXQ_FMT_P16X64, nullptr) == MIXQ_ESHAPE);
    CHECK(mixq_pack_operand(w.p, wp.p, N, K, 7, nullptr) == MIXQ_EINVAL);
    CHECK(mixq_pack_operand(nullptr, wp.p, N, K, MIXQ_FMT_P16X64, nullptr) == MIXQ_EINVAL);
    CHECK(mixq_pack_p16x64(w.p, wp.p, N, K, nullptr) == MIXQ_OK);
    CHECK(mixq_dequant_weight_cols(w.p, sw.p, ind.p, n, wo.p, N, K, cap, 8, nullptr) == MIXQ_OK);
    CHECK(mixq_dequant_weight_cols(w.p, sw.p, ind.p, n, wo.p, N, K, 2, 8, nullptr) == MIXQ_EINVAL);
    CHECK(mixq_dequant_weight_cols(w.p, sw.p, ind.p, 0, wo.p, N, K, cap, 8, nullptr) == MIXQ_OK);

    // ---- fused GEMM: plain operands, P16X64 pair, fragment-order weights - one result ----------------------------------------------------
    x.up(hx);
    Dev<uint8_t> qplain(static_cast<size_t>(M) * K);
    CHECK(mixq_quant_fused(x.p, ind.p, cap, ndev.p, xs.p, qplain.p, xo.p, flag.p, M, K, K, cap, 8, 6.f, MIXQ_FMT_PLAIN, nullptr) == MIXQ_OK);
    x.up(hx);
    CHECK(mixq_quant_fused(x.p, ind.p, cap, ndev.p, xs.p, q.p, xo.p, flag.p, M, K, K, cap, 8, 6.f, MIXQ_FMT_P16X64, nullptr) == MIXQ_OK);
    CHECK(mixq_gemm_i8_fused(reinterpret_cast<int8_t*>(qplain.p), w.p, xs.p, sw.p, xo.p, cap, wo.p, cap, cap, ndev.p, nullptr, 0, bias.p, y.p, N, M, N, K, MIXQ_ACT_NONE, 0, nullptr) == MIXQ_OK);
    CHECK(mixq_gemm_i8_fused(reinterpret_cast<int8_t*>(q.p), reinterpret_cast<int8_t*>(wp.p), xs.p, sw.p, xo.p, cap, wo.p, cap, cap, ndev.p, nullptr, 0, bias.p, y2.p, N, M, N, K, MIXQ_ACT_NONE, MIXQ_X_PACKED | MIXQ_W_PACKED, nullptr) == MIXQ_OK);
    CHECK(mixq_gemm_i8_fused(reinterpret_cast<int8_t*>(q.p), reinterpret_cast<int8_t*>(wf.p), xs.p, sw.p, xo.p, cap, wo.p, cap, cap, ndev.p, nullptr, 0, bias.p, y3.p, N, M, N, K, MIXQ_ACT_NONE, MIXQ_X_PACKED | MIXQ_W_F16X64, nullptr) == MIXQ_OK);
    HIPOK(hipDeviceSynchronize());
    { auto a = y.down(); CHECK(a == y2.down() && a == y3.down()); float s = 0; for (auto v : a) s += h2f(v); CHECK(s == s); }
    CHECK(mixq_gemm_i8_fused(reinterpret_cast<int8_t*>(q.p), reinterpret_cast<int8_t*>(wf.p), xs.p, sw.p, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, 0, nullptr, y3.p, N, M, N, 100, MIXQ_ACT_NONE, MIXQ_X_PACKED | MIXQ_W_F16X64, nullptr) == MIXQ_ESHAPE);
    CHECK(mixq_gemm_i8_fused(nullptr, reinterpret_cast<int8_t*>(wf.p), xs.p, sw.p, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, 0, nullptr, y3.p, N, M, N, K, MIXQ_ACT_NONE, MIXQ_X_PACKED | MIXQ_W_F16X64, nullptr) == MIXQ_EINVAL);
    CHECK(mixq_gemm_i8_fused(reinterpret_cast<int8_t*>(q.p), reinterpret_cast<int8_t*>(wf.p), xs.p, sw.p, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, 0, nullptr, y3.p, N, 0, N, K, MIXQ_ACT_NONE, MIXQ_X_PACKED | MIXQ_W_F16X64, nullptr) == MIXQ_OK);
    CHECK(mixq_gemm_i8_fused(reinterpret_cast<int8_t*>(q.p), reinterpret_cast<int8_t*>(wf.p), xs.p, sw.p, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, 0, nullptr, y3.p, N, M, N, K, MIXQ_ACT_SILU_MUL, MIXQ_X_PACKED | MIXQ_W_F16X64, nullptr) != MIXQ_OK);   // (the multiplier is required)
    (void)mixq_gemm_amax_supported(M, N, K, MIXQ_X_PACKED | MIXQ_W_F16X64);

    // ---- the whole forward on a KEPT argument block: re-used, then edited (another batch size, the kept-map route) -----------------------
    mixq_linear_args a;
    memset(&a, 0, sizeof(a));
    a.x = x.p; a.ldx = K; a.ind = ind.p; a.n_cap = cap; a.n_dev = ndev.p; a.x_scale = xs.p; a.q_x = q.p; a.x_out = xo.p; a.ldxo = cap; a.flag = flag.p;
    a.q_w = wf.p; a.scale_col = sw.p; a.w_out = wo.p; a.ldwo = cap; a.bias = bias.p; a.y = y2.p; a.ldy = N;
    a.M = M; a.N = N; a.K = K; a.bit = 8; a.sigma = 6.f; a.act = MIXQ_ACT_NONE; a.qfmt = MIXQ_FMT_P16X64; a.wfmt = MIXQ_FMT_F16X64;
    for (int rep = 0; rep < 3; ++rep) {
        x.up(hx);
        CHECK(mixq_linear_forward(&a, nullptr) == MIXQ_OK);
        HIPOK(hipDeviceSynchronize());
        CHECK(y2.down() == y.down());
    }
    a.col_mask = map.p; a.col_mask_words = map_words;                          // the frozen layer's kept outlier map
    x.up(hx);
    CHECK(mixq_linear_forward(&a, nullptr) == MIXQ_OK);
    HIPOK(hipDeviceSynchronize());
    CHECK(y2.down() == y.down());
    a.M = 17;                                                                  // a decode-sized batch on the same block
    x.up(hx);
    CHECK(mixq_linear_forward(&a, nullptr) == MIXQ_OK);
    HIPOK(hipDeviceSynchronize());
    { auto full = y.down(), part = y2.down(); CHECK(memcmp(full.data(), part.data(), static_cast<size_t>(17) * N * 2) == 0); }
    a.M = 0;
    CHECK(mixq_linear_forward(&a, nullptr) == MIXQ_OK);
    a.M = M; a.bit = 3;
    CHECK(mixq_linear_forward(&a, nullptr) == MIXQ_EINVAL);
    CHECK(mixq_linear_forward(nullptr, nullptr) == MIXQ_EINVAL);
    a.bit = 8; a.qfmt = MIXQ_FMT_PLAIN;                                        // fragment-order weights need P16X64 activations
    CHECK(mixq_linear_forward(&a, nullptr) != MIXQ_OK);

    // ---- RMSNorm in front of the path -----------------------------------------------------------------------------------------------------
    std::vector<uint16_t> hg(K);
    for (auto& v : hg) v = f2h(0.5f + (rnd() % 100) / 100.f);
    Dev<uint16_t> g(K); g.up(hg);
    x.up(hx);
    CHECK(mixq_rmsnorm(x.p, g.p, out.p, M, K, K, K, 1e-5f, nullptr) == MIXQ_OK);
    CHECK(mixq_rmsnorm(x.p, g.p, out.p, M, K, K, K - 8, 1e-5f, nullptr) == MIXQ_ESHAPE);
    CHECK(mixq_rmsnorm_quant_fused(x.p, g.p, out.p, ind.p, cap, ndev.p, xs.p, q.p, xo.p, flag.p, M, K, K, K, cap, 1e-5f, 8, 6.f, MIXQ_FMT_P16X64, nullptr) == MIXQ_OK);
    CHECK(mixq_rmsnorm_quant_fused_masked(x.p, g.p, x2.p, ind.p, cap, ndev.p, map.p, map_words, xs2.p, q2.p, xo2.p, flag.p, M, K, K, K, cap, 1e-5f, 8, 6.f, MIXQ_FMT_P16X64, nullptr) == MIXQ_OK);
    HIPOK(hipDeviceSynchronize());
    CHECK(out.down() == x2.down() && xs.down() == xs2.down() && xo.down() == xo2.down());
    CHECK(mixq_rmsnorm_quant_fused(x.p, g.p, out.p, ind.p, n, nullptr, xs.p, q.p, nullptr, flag.p, M, K, K, K, cap, 1e-5f, 8, 6.f, MIXQ_FMT_PLAIN, nullptr) == MIXQ_EINVAL);

    // ---- weight-only W8A16 -------------------------------------------------------------------------------------------------------------------
    std::vector<int8_t> hkn(static_cast<size_t>(K) * N);
    for (auto& v : hkn) v = static_cast<int8_t>(static_cast<int>(rnd() % 255) - 127);
    Dev<int8_t> kn(hkn.size()); kn.up(hkn);
    Dev<uint8_t> w16(static_cast<size_t>(N16) * K);
    x.up(hx);
    CHECK(mixq_pack_w8a16(kn.p, w16.p, K, N, nullptr) == MIXQ_OK);
    CHECK(mixq_gemm_w8a16(x.p, K, w16.p, sw.p, bias.p, y3.p, N, M, N, K, nullptr) == MIXQ_OK);
    CHECK(mixq_gemm_w8a16(x.p, K, w16.p, sw.p, bias.p, y3.p, N, 1, N, K, nullptr) == MIXQ_OK);
    CHECK(mixq_gemm_w8a16(x.p, K, w16.p, sw.p, bias.p, y3.p, N, M, N, 100, nullptr) == MIXQ_ESHAPE);
    HIPOK(hipDeviceSynchronize());

    printf(g_fail ? "asan_capi: %d check(s) FAILED\n" : "asan_capi: all checks passed\n", g_fail);
    return g_fail ? 1 : 0;
}
