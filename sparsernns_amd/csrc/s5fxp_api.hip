// s5fxp_api.hip -- host side of libs5fxp.so: the C ABI declared in include/s5fxp.h.
//
// Everything here only validates arguments, packs parameters and enqueues kernels on the
// caller's stream.  No device allocation, no synchronisation, no global mutable state.
#include "../../include/s5fxp.h"
#include "s5fxp_kernels.hpp"
#include "mfma_fused.hpp"
#include "scan_assoc.hpp"

#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

using namespace s5;

namespace {

inline hipStream_t S(void *s) { return reinterpret_cast<hipStream_t>(s); }
inline int hip_rc(hipError_t e) { return e == hipSuccess ? S5FXP_OK : S5FXP_EHIP; }
inline int launch_rc() { return hip_rc(hipGetLastError()); }
inline unsigned ew_grid(int64_t n)
{
    int64_t g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}
inline bool shift_ok(int s) { return s >= 0 && s <= 31; }

// column split of the frame-tiled matmul: 4 waves x mw columns, mw % 4 == 0
inline int mw_for(int M) { return ((M + 3) / 4 + 3) / 4 * 4; }
constexpr int MW_LIMIT = 68;

// dispatch on the compile-time column budget
#define S5_DISPATCH_MW(mw, X24, KERNEL, grid, stream, args)                                                   \
    do {                                                                                                      \
        if ((mw) <= 4) hipLaunchKernelGGL((KERNEL<4, X24>), dim3(grid), dim3(256), 0, stream, args);          \
        else if ((mw) <= 8) hipLaunchKernelGGL((KERNEL<8, X24>), dim3(grid), dim3(256), 0, stream, args);     \
        else if ((mw) <= 16) hipLaunchKernelGGL((KERNEL<16, X24>), dim3(grid), dim3(256), 0, stream, args);   \
        else if ((mw) <= 24) hipLaunchKernelGGL((KERNEL<24, X24>), dim3(grid), dim3(256), 0, stream, args);   \
        else if ((mw) <= 32) hipLaunchKernelGGL((KERNEL<32, X24>), dim3(grid), dim3(256), 0, stream, args);   \
        else if ((mw) <= 48) hipLaunchKernelGGL((KERNEL<48, X24>), dim3(grid), dim3(256), 0, stream, args);   \
        else if ((mw) <= 64) hipLaunchKernelGGL((KERNEL<64, X24>), dim3(grid), dim3(256), 0, stream, args);   \
        else hipLaunchKernelGGL((KERNEL<68, X24>), dim3(grid), dim3(256), 0, stream, args);                   \
    } while (0)

// k_cproj keeps two output tiles in LDS; its column budget stops at 48 (H <= 192)
#define S5_DISPATCH_MW_C(mw, X24, PASS, AT, grid, stream, args)                                                         \
    do {                                                                                                      \
        if ((mw) <= 4) hipLaunchKernelGGL((k_cproj<4, X24, PASS, AT>), dim3(grid), dim3(256), 0, stream, args);         \
        else if ((mw) <= 8) hipLaunchKernelGGL((k_cproj<8, X24, PASS, AT>), dim3(grid), dim3(256), 0, stream, args);    \
        else if ((mw) <= 16) hipLaunchKernelGGL((k_cproj<16, X24, PASS, AT>), dim3(grid), dim3(256), 0, stream, args);  \
        else if ((mw) <= 24) hipLaunchKernelGGL((k_cproj<24, X24, PASS, AT>), dim3(grid), dim3(256), 0, stream, args);  \
        else if ((mw) <= 32) hipLaunchKernelGGL((k_cproj<32, X24, PASS, AT>), dim3(grid), dim3(256), 0, stream, args);  \
        else hipLaunchKernelGGL((k_cproj<48, X24, PASS, AT>), dim3(grid), dim3(256), 0, stream, args);                  \
    } while (0)
constexpr int MW_LIMIT_C = 48;

bool fits24(const int32_t *p, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        if (p[i] < -(1 << 23) || p[i] >= (1 << 23)) return false;
    return true;
}

} // namespace

// -----------------------------------------------------------------------------------------------
extern "C" int s5fxp_version(void) { return S5FXP_VERSION; }

extern "C" const char *s5fxp_strerror(int code)
{
    switch (code) {
    case S5FXP_OK: return "ok";
    case S5FXP_EBADARG: return "bad argument";
    case S5FXP_ENEGSHIFT: return "negative or out-of-range shift (invalid result_exp)";
    case S5FXP_EUNSUPPORTED: return "unsupported configuration";
    case S5FXP_EHIP: return "HIP runtime error";
    case S5FXP_EWORKSPACE: return "workspace or blob too small";
    default: return "unknown error";
    }
}

// -----------------------------------------------------------------------------------------------
// op level
// -----------------------------------------------------------------------------------------------
extern "C" int s5fxp_from_fp(const float *x, int32_t *y, int64_t n, int bits, int exp, int round_mode, void *stream)
{
    if (!x || !y || n < 0 || bits < 1 || bits > 32 || exp < 0 || exp > 31 || round_mode < 0 || round_mode > 2)
        return S5FXP_EBADARG;
    if (n == 0) return S5FXP_OK;
    hipLaunchKernelGGL(k_from_fp, dim3(ew_grid(n)), dim3(256), 0, S(stream), x, y, n, bits, exp, round_mode);
    return launch_rc();
}

extern "C" int s5fxp_to_float(const int32_t *x, float *y, int64_t n, int exp, void *stream)
{
    if (!x || !y || n < 0 || exp < 0 || exp > 31) return S5FXP_EBADARG;
    if (n == 0) return S5FXP_OK;
    hipLaunchKernelGGL(k_to_float, dim3(ew_grid(n)), dim3(256), 0, S(stream), x, y, n, exp);
    return launch_rc();
}

extern "C" int s5fxp_change_cfg(const int32_t *x, int32_t *y, int64_t n, int bits, int exp, int new_bits, int new_exp,
                                void *stream)
{
    if (!x || !y || n < 0 || bits < 1 || bits > 32 || new_bits < 1 || new_bits > 32) return S5FXP_EBADARG;
    if (!shift_ok(exp > new_exp ? exp - new_exp : new_exp - exp)) return S5FXP_ENEGSHIFT;
    if (n == 0) return S5FXP_OK;
    hipLaunchKernelGGL(k_change_cfg, dim3(ew_grid(n)), dim3(256), 0, S(stream), x, y, n, bits, exp, new_bits, new_exp);
    return launch_rc();
}

extern "C" int s5fxp_dense(const int32_t *x, const int32_t *w, const int32_t *bias, int32_t *y, int64_t N, int K, int M,
                           int x_exp, int w_exp, int b_bits, int b_exp, int out_bits, int out_exp, int flags,
                           void *stream)
{
    if (!x || !w || !y || N < 0 || K < 1 || M < 1 || out_bits < 1 || out_bits > 32) return S5FXP_EBADARG;
    if (mw_for(M) > MW_LIMIT) return S5FXP_EUNSUPPORTED;
    if (!shift_ok(x_exp + w_exp - out_exp)) return S5FXP_ENEGSHIFT;
    if (bias && !shift_ok(b_exp > out_exp ? b_exp - out_exp : out_exp - b_exp)) return S5FXP_ENEGSHIFT;
    if (N == 0) return S5FXP_OK;
    DenseArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.N = N; a.K = K; a.M = M; a.mw = mw_for(M);
    a.xb = 32; a.xe = DynExp{x_exp, nullptr}; a.check_inp = 0;
    a.w_exp = w_exp; a.b_bits = b_bits; a.b_exp = b_exp; a.out_bits = out_bits; a.out_exp = out_exp;
    a.relu = flags & 1; a.check24 = 0; a.status = nullptr;
    const unsigned grid = (unsigned)((N + TN - 1) / TN);
    S5_DISPATCH_MW(a.mw, false, k_dense, grid, S(stream), a);
    return launch_rc();
}

extern "C" int s5fxp_dense_csr(const int32_t *x, const int32_t *rowptr, const int32_t *colidx, const int32_t *val,
                               const int32_t *bias, int32_t *y, int64_t N, int K, int M, int x_exp, int w_exp, int b_bits,
                               int b_exp, int out_bits, int out_exp, int flags, void *stream)
{
    if (!x || !rowptr || !y || N < 0 || K < 1 || M < 1 || out_bits < 1 || out_bits > 32) return S5FXP_EBADARG;
    if (!colidx || !val) return S5FXP_EBADARG;
    const size_t smem = (size_t)(64 * (K | 1) + 64 * 65) * 4;
    if (smem > 160 * 1024) return S5FXP_EUNSUPPORTED; // K <= 574: the input tile must fit one CU's LDS
    if (!shift_ok(x_exp + w_exp - out_exp)) return S5FXP_ENEGSHIFT;
    if (bias && !shift_ok(b_exp > out_exp ? b_exp - out_exp : out_exp - b_exp)) return S5FXP_ENEGSHIFT;
    if (N == 0) return S5FXP_OK;
    DenseCsrArgs a{};
    a.x = x; a.rowptr = rowptr; a.colidx = colidx; a.val = val; a.bias = bias; a.y = y; a.N = N; a.K = K; a.M = M;
    a.rs = x_exp + w_exp - out_exp; a.b_bits = b_bits; a.b_exp = b_exp; a.out_bits = out_bits; a.out_exp = out_exp;
    a.relu = flags & 1;
    if (smem > 65536)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_dense_csr), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k_dense_csr, dim3((unsigned)((N + 63) / 64)), dim3(256), smem, S(stream), a);
    return launch_rc();
}

extern "C" int s5fxp_add(const int32_t *x, const int32_t *y, int32_t *out, int64_t n, int64_t y_len, int x_bits,
                         int x_exp, int y_bits, int y_exp, int out_bits, int out_exp, int negate_y, void *stream)
{
    if (!x || !y || !out || n < 0 || y_len < 1 || (n % y_len) != 0) return S5FXP_EBADARG;
    if (!shift_ok(x_exp > out_exp ? x_exp - out_exp : out_exp - x_exp) ||
        !shift_ok(y_exp > out_exp ? y_exp - out_exp : out_exp - y_exp))
        return S5FXP_ENEGSHIFT;
    if (n == 0) return S5FXP_OK;
    hipLaunchKernelGGL(k_add, dim3(ew_grid(n)), dim3(256), 0, S(stream), x, y, out, n, y_len, x_bits, x_exp, y_bits,
                       y_exp, out_bits, out_exp, negate_y);
    return launch_rc();
}

extern "C" int s5fxp_mul(const int32_t *x, const int32_t *y, int32_t *out, int64_t n, int64_t y_len, int x_exp,
                         int y_exp, int out_bits, int out_exp, void *stream)
{
    if (!x || !y || !out || n < 0 || y_len < 1 || (n % y_len) != 0) return S5FXP_EBADARG;
    const int rs = x_exp + y_exp - out_exp;
    if (!shift_ok(rs)) return S5FXP_ENEGSHIFT; // fxparray.py:619-621
    if (n == 0) return S5FXP_OK;
    hipLaunchKernelGGL(k_mul, dim3(ew_grid(n)), dim3(256), 0, S(stream), x, y, out, n, y_len, rs, out_bits);
    return launch_rc();
}

static int cb_common(bool is_mul, const int32_t *x, const int32_t *y, int32_t *out, int64_t n, int64_t y_len,
                     int x_bits, int x_exp, int y_bits, int y_exp, int out_bits, int32_t *out_exp_dev, void *scratch,
                     void *stream)
{
    if (!x || !y || !out || !out_exp_dev || !scratch || n < 1 || y_len < 1 || (n % y_len) != 0) return S5FXP_EBADARG;
    uint32_t *sc = reinterpret_cast<uint32_t *>(scratch);
    int rc = hip_rc(hipMemsetAsync(sc, 0, 32, S(stream)));
    if (rc) return rc;
    if (is_mul) {
        hipLaunchKernelGGL(k_cb_reduce<true>, dim3(ew_grid(n)), dim3(256), 0, S(stream), x, y, n, y_len, x_exp, y_exp, sc);
        hipLaunchKernelGGL(k_cb_finalize, dim3(1), dim3(64), 0, S(stream), sc, x_exp, y_exp, out_bits, 1, out_exp_dev);
        hipLaunchKernelGGL(k_cb_apply<true>, dim3(ew_grid(n)), dim3(256), 0, S(stream), x, y, out, n, y_len, x_bits,
                           y_bits, out_bits, sc);
    } else {
        hipLaunchKernelGGL(k_cb_reduce<false>, dim3(ew_grid(n)), dim3(256), 0, S(stream), x, y, n, y_len, x_exp, y_exp, sc);
        hipLaunchKernelGGL(k_cb_finalize, dim3(1), dim3(64), 0, S(stream), sc, x_exp, y_exp, out_bits, 0, out_exp_dev);
        hipLaunchKernelGGL(k_cb_apply<false>, dim3(ew_grid(n)), dim3(256), 0, S(stream), x, y, out, n, y_len, x_bits,
                           y_bits, out_bits, sc);
    }
    return launch_rc();
}

extern "C" int s5fxp_add_cb(const int32_t *x, const int32_t *y, int32_t *out, int64_t n, int64_t y_len, int x_bits,
                            int x_exp, int y_bits, int y_exp, int out_bits, int32_t *out_exp_dev, void *scratch,
                            void *stream)
{
    return cb_common(false, x, y, out, n, y_len, x_bits, x_exp, y_bits, y_exp, out_bits, out_exp_dev, scratch, stream);
}

extern "C" int s5fxp_mul_cb(const int32_t *x, const int32_t *y, int32_t *out, int64_t n, int64_t y_len, int x_exp,
                            int y_exp, int out_bits, int32_t *out_exp_dev, void *scratch, void *stream)
{
    return cb_common(true, x, y, out, n, y_len, 32, x_exp, 32, y_exp, out_bits, out_exp_dev, scratch, stream);
}

extern "C" int s5fxp_relu(const int32_t *re, const int32_t *im, int32_t *out_re, int32_t *out_im, int64_t n,
                          void *stream)
{
    if (!re || !out_re || n < 0 || (im && !out_im)) return S5FXP_EBADARG;
    if (n == 0) return S5FXP_OK;
    hipLaunchKernelGGL(k_relu, dim3(ew_grid(n)), dim3(256), 0, S(stream), re, im, out_re, out_im, n);
    return launch_rc();
}

extern "C" int s5fxp_sigmoid(const int32_t *x, int32_t *y, int64_t n, int x_bits, int x_exp, int sig_x_exp,
                             int sig_y_exp, const int32_t *lut_host, void *stream)
{
    if (!x || !y || !lut_host || n < 0 || sig_x_exp < 0 || sig_x_exp > 15 || sig_y_exp < 1 || sig_y_exp > 30)
        return S5FXP_EBADARG;
    if (!shift_ok(x_exp > sig_x_exp ? x_exp - sig_x_exp : sig_x_exp - x_exp)) return S5FXP_ENEGSHIFT;
    if (n == 0) return S5FXP_OK;
    Lut8 l;
    for (int i = 0; i < 8; ++i) l.v[i] = lut_host[i];
    hipLaunchKernelGGL(k_sigmoid, dim3(ew_grid(n)), dim3(256), 0, S(stream), x, y, n, x_bits, x_exp, sig_x_exp,
                       sig_y_exp, l);
    return launch_rc();
}

static int launch_scan(ScanArgs &a, hipStream_t st)
{
    const int64_t total = (int64_t)a.B * a.P;
    const unsigned grid = (unsigned)((total + 63) / 64);
    hipLaunchKernelGGL(k_scan_lane<8>, dim3(grid), dim3(64), 0, st, a);
    return launch_rc();
}

extern "C" int s5fxp_scan(const int32_t *bu_re, const int32_t *bu_im, const int32_t *a_re, const int32_t *a_im,
                          int32_t *xs_re, int32_t *xs_im, int B, int L, int P, int a_re_exp, int a_im_exp,
                          int bu_re_exp, int bu_im_exp, int x_re_exp, int x_im_exp, int flags, void *stream)
{
    if (!bu_re || !bu_im || !a_re || !a_im || !xs_re || !xs_im || B < 0 || L < 0 || P < 1) return S5FXP_EBADARG;
    const int sr = bu_re_exp - x_re_exp, si = bu_im_exp - x_im_exp;
    if (!shift_ok(a_re_exp) || !shift_ok(a_im_exp) || !shift_ok(sr < 0 ? -sr : sr) || !shift_ok(si < 0 ? -si : si))
        return S5FXP_ENEGSHIFT;
    if (B == 0 || L == 0) return S5FXP_OK;
    ScanArgs a{};
    a.bu_re = bu_re; a.bu_im = bu_im; a.a_re = a_re; a.a_im = a_im; a.out_re = xs_re; a.out_im = xs_im;
    a.B = B; a.L = L; a.P = P; a.TB = 0; a.ea_re = a_re_exp; a.ea_im = a_im_exp; a.sh_re = sr; a.sh_im = si;
    a.relu = flags & 1; a.run_if = nullptr;
    return launch_scan(a, S(stream));
}

extern "C" int s5fxp_assoc_scan_c64(const float *lambda, const float *bu, float *xs, const float *x0, float *x_last, int B,
                                    int L, int P, int reverse, void *stream)
{
    if (!lambda || !bu || !xs || B < 0 || L < 0 || P < 1) return S5FXP_EBADARG;
    if (B == 0 || L == 0) return S5FXP_OK;
    const int64_t grid = (int64_t)B * ((P + ASSOC_PT - 1) / ASSOC_PT);
    if (grid > 0x7fffffffll) return S5FXP_EBADARG;
    ScanAssocArgs a{};
    a.lambda = reinterpret_cast<const float2 *>(lambda); a.bu = reinterpret_cast<const float2 *>(bu);
    a.xs = reinterpret_cast<float2 *>(xs); a.x0 = reinterpret_cast<const float2 *>(x0);
    a.x_last = reinterpret_cast<float2 *>(x_last); a.B = B; a.L = L; a.P = P; a.reverse = reverse ? 1 : 0;
    hipLaunchKernelGGL(k_scan_assoc_c64, dim3((unsigned)grid), dim3(64 * ASSOC_WAVES), 0, S(stream), a);
    return launch_rc();
}

// -----------------------------------------------------------------------------------------------
// model level
// -----------------------------------------------------------------------------------------------
struct DenseDev {
    int K = 0, M = 0;
    const int32_t *w = nullptr, *bias = nullptr;
    int w_exp = 0, b_bits = 0, b_exp = 0, inp_bits = 0, inp_exp = 0, out_bits = 0, out_exp = 0;
    bool x24 = false;
};

struct LayerDev {
    // norm
    const int32_t *mm = nullptr, *isv = nullptr, *scale = nullptr, *nbias = nullptr;
    s5fxp_norm_desc nd{};
    // ssm
    const int32_t *a_re = nullptr, *a_im = nullptr, *bcat = nullptr, *c_re_t = nullptr, *c_im_t = nullptr, *D = nullptr;
    s5fxp_ssm_desc sd{};
    bool b24 = false, c24 = false;
    bool quad_ok = false; // quad recurrence kernel applicable (P % 16 == 0, coefficients fit 24 bits)
    int32_t quad_xmax = 0; // its exactness bound on |state|
    bool pair_ok = false;  // pair recurrence kernel applicable (scan_quad.hpp k_scan_pair_asm)
    int32_t pair_xmax = 0; // its exactness bound on |state|
    DenseDev out2;
    int l_bits, l_exp, r_bits, r_exp, res_bits, res_exp, sig_x, sig_y;
    int32_t lut[8];
};

struct FastModel; // int8-MFMA path (s5fxp_fast.hpp); nullptr when the model is not eligible

// Experiment / test switches.  They are read from the environment ONCE, in s5fxp_model_create, and live in the handle:
// a forward never consults the environment, so s5fxp_model_recurrence_kernel always reports what s5fxp_model_forward
// runs and two handles with different switches can be used side by side (include/s5fxp.h, "Environment").
struct ModelCfg {
    bool debug_sync = false;  // S5FXP_DEBUG_SYNC: synchronise + check after every stage of the fused forward
    bool no_bn_ext = false;   // S5FXP_NO_BN_EXT: four full BatchNorm reductions instead of the per-channel extremes
    bool no_pair = false;     // S5FXP_NO_PAIR: never the pair recurrence kernel
    bool pair_global = false; // S5FXP_PAIR_GLOBAL: pair kernel fed from an int32 K stream in global memory (no helper wave)
    bool no_pk16 = false;     // S5FXP_NO_PK16: unpacked epilogues in the gate kernel
    bool no_compact = false;  // S5FXP_NO_COMPACT: never run a layer on its live states only (s5fxp_fast.hpp FastLayer)
    bool no_live_lanes = false; // S5FXP_NO_LIVE_LANES: the recurrence streams of a compacted layer keep their padding slots
    bool gate_bn = false;      // S5FXP_GATE_BN: the gate kernel recomputes u = BatchNorm(layer input) instead of reading it (k_cgate_p<.., GBN>;
                               // measured slower: DESIGN.md 4a)
    bool cgate_ft64 = false;   // S5FXP_CGATE_FT64: the packed-epilogue gate kernel on 64-frame tiles with six-wave workgroups (the form before
                               // the 32-frame / three-wave one became the default at dim_scale 0.5)
    int64_t cap_cgate32 = 1280; // S5FXP_WGS_CGATE32: workgroups per launch of the 32-frame gate kernel (5 per CU: all resident)
    bool no_dec_resid = false; // S5FXP_NO_DEC_RESID: the last layer's residual pass as its own launch (proj_p.hpp k_dec_p<.., RESID>)
    int pairl_blocks = 32;    // S5FXP_PAIRL_BLOCKS=16: 16 time blocks per LDS buffer of the LDS-fed pair kernel
    size_t plane_skew = 0;    // S5FXP_PLANE_SKEW=<bytes, multiple of 256>: extra distance between the workspace's planes (experiments)
    int64_t cap_enc = 512, cap_dec = 512, cap_cgate = 512, cap_bproj = 1024, cap_resid = 512; // S5FXP_WGS_*: workgroups per launch
    static ModelCfg from_env()
    {
        ModelCfg c;
        auto on = [](const char *n) { return std::getenv(n) != nullptr; };
        auto cap = [](const char *n, int64_t dflt) {
            const char *e = std::getenv(n);
            const int v = e ? std::atoi(e) : 0;
            return (int64_t)(v > 0 ? v : dflt);
        };
        c.debug_sync = on("S5FXP_DEBUG_SYNC"); c.no_bn_ext = on("S5FXP_NO_BN_EXT"); c.no_pair = on("S5FXP_NO_PAIR");
        c.pair_global = on("S5FXP_PAIR_GLOBAL"); c.no_pk16 = on("S5FXP_NO_PK16"); c.no_compact = on("S5FXP_NO_COMPACT"); c.no_dec_resid = on("S5FXP_NO_DEC_RESID"); c.no_live_lanes = on("S5FXP_NO_LIVE_LANES"); c.gate_bn = on("S5FXP_GATE_BN"); c.cgate_ft64 = on("S5FXP_CGATE_FT64"); c.cap_cgate32 = cap("S5FXP_WGS_CGATE32", c.cap_cgate32);
        { const char *e = std::getenv("S5FXP_PAIRL_BLOCKS"); c.pairl_blocks = e && std::atoi(e) == 16 ? 16 : 32; }
        { const char *e = std::getenv("S5FXP_PLANE_SKEW"); c.plane_skew = e ? ((size_t)std::atoll(e) & ~(size_t)255) : 0; }
        c.cap_enc = cap("S5FXP_WGS_ENC", c.cap_enc); c.cap_dec = cap("S5FXP_WGS_DEC", c.cap_dec);
        c.cap_cgate = cap("S5FXP_WGS_CGATE", c.cap_cgate); c.cap_bproj = cap("S5FXP_WGS_BPROJ", c.cap_bproj);
        c.cap_resid = cap("S5FXP_WGS_RESID", c.cap_resid);
        return c;
    }
};

struct s5fxp_model {
    int n_layers = 0, d_in = 0, H = 0, P = 0, d_out = 0;
    DenseDev enc, dec;
    std::vector<LayerDev> layers;
    int flags = 0;
    ModelCfg cfg;
    FastModel *fast = nullptr;
};

namespace {

struct Packer {
    size_t off = 0;
    char *host = nullptr; // nullptr: size pass
    char *dev = nullptr;
    const int32_t *put(const int32_t *src, size_t n)
    {
        const size_t bytes = n * sizeof(int32_t);
        const int32_t *d = reinterpret_cast<const int32_t *>(dev + off);
        if (host) std::memcpy(host + off, src, bytes);
        off = (off + bytes + 255) & ~(size_t)255;
        return d;
    }
};

bool dense_ok(const s5fxp_dense_desc &d) { return d.weight && d.bias && d.K > 0 && d.M > 0; }

void pack_dense(Packer &p, const s5fxp_dense_desc &d, DenseDev &o, bool allow24)
{
    o.K = d.K; o.M = d.M;
    o.w = p.put(d.weight, (size_t)d.K * d.M);
    o.bias = p.put(d.bias, (size_t)d.M);
    o.w_exp = d.w_exp; o.b_bits = d.b_bits; o.b_exp = d.b_exp; o.inp_bits = d.inp_bits; o.inp_exp = d.inp_exp;
    o.out_bits = d.out_bits; o.out_exp = d.out_exp;
    o.x24 = allow24 && fits24(d.weight, (size_t)d.K * d.M);
}

// Exactness bounds of the fast recurrence kernels (scan_quad.hpp) over the states idx[0..n) of a layer that runs on P_slots
// state slots (all P states, or the live ones of a compacted layer: a dead state's coefficients never meet a non-zero state).
struct ScanBounds {
    bool quad_ok = false, pair_ok = false;
    int32_t quad_xmax = 0, pair_xmax = 0;
};
ScanBounds scan_bounds(const s5fxp_ssm_desc &ssm, const int *idx, int n, int P_slots, bool allow24)
{
    ScanBounds o;
    // scaled coefficients c = A * 2^(16-e) must fit 24 signed bits; |c*x| + 2^16 < 2^31 bounds the state
    const int sre = 16 - ssm.A_re_exp, sim = 16 - ssm.A_im_exp;
    int64_t cmax = 1;
    for (int j = 0; j < n; ++j) {
        const int q = idx[j];
        const int64_t ar = std::llabs((long long)ssm.A_re[q]), ai = std::llabs((long long)ssm.A_im[q]);
        const int smax = sre > sim ? sre : sim;
        if (smax >= 0 && smax < 24) {
            cmax = std::max(cmax, ar << smax);
            cmax = std::max(cmax, ai << smax);
        }
    }
    o.quad_ok = allow24 && (P_slots % 16 == 0) && sre >= 0 && sim >= 0 && sre < 16 && sim < 16 && cmax < (1 << 23);
    o.quad_xmax = (int32_t)std::min<int64_t>(((int64_t(1) << 31) - 1 - 65536) / cmax, (1 << 23) - 1);
    // pair kernel: Bu is folded into the addend of the own product (always an Ai product), so
    //   |Ai| * 2^(16-e) * |x| + 2^16 * (bmax + 1) <= 2^31 - 1   and   |Ar| * 2^(16-e) * |x| <= 2^31 - 1
    // with bmax = the largest |Bu| after the shift to the state exponent (static: its bits minus the shift)
    const int sh_re = ssm.Bu_re_exp - ssm.x_re_exp, sh_im = ssm.Bu_im_exp - ssm.x_im_exp;
    const int bb_re = ssm.Bu_re_bits - sh_re, bb_im = ssm.Bu_im_bits - sh_im; // bits of the shifted Bu
    if (o.quad_ok && P_slots % 32 == 0 && bb_re >= 1 && bb_re <= 16 && bb_im >= 1 && bb_im <= 16) {
        const int64_t lim = (int64_t(1) << 31) - 1;
        int64_t xm = 32766;
        for (int j = 0; j < n; ++j) {
            const int q = idx[j];
            const int64_t ar = std::llabs((long long)ssm.A_re[q]), ai = std::llabs((long long)ssm.A_im[q]);
            const int64_t room_re = lim - 65536 * ((int64_t(1) << (bb_re - 1)) + 1), room_im = lim - 65536 * ((int64_t(1) << (bb_im - 1)) + 1);
            if (ai) xm = std::min({xm, room_re / (ai << sre), room_im / (ai << sim)});
            if (ar) xm = std::min({xm, lim / (ar << sre), lim / (ar << sim)});
        }
        o.pair_xmax = (int32_t)xm;
        o.pair_ok = xm >= 16384; // below that the quad kernel (bound 32767) is the better optimistic choice
    }
    return o;
}

void pack_layer(Packer &p, const s5fxp_layer_desc &l, LayerDev &o, bool allow24)
{
    const int H = l.ssm.H, P = l.ssm.P;
    o.nd = l.norm;
    o.mm = p.put(l.norm.minus_mean, H);
    o.isv = p.put(l.norm.invsq_var, H);
    o.scale = l.norm.scale ? p.put(l.norm.scale, H) : nullptr;
    o.nbias = l.norm.bias ? p.put(l.norm.bias, H) : nullptr;
    o.sd = l.ssm;
    o.a_re = p.put(l.ssm.A_re, P);
    o.a_im = p.put(l.ssm.A_im, P);
    std::vector<int32_t> tmp((size_t)H * 2 * P);
    for (int h = 0; h < H; ++h)
        for (int q = 0; q < P; ++q) {
            tmp[(size_t)h * 2 * P + q] = l.ssm.B_re[(size_t)q * H + h];
            tmp[(size_t)h * 2 * P + P + q] = l.ssm.B_im[(size_t)q * H + h];
        }
    o.bcat = p.put(tmp.data(), tmp.size());
    o.b24 = allow24 && fits24(tmp.data(), tmp.size());
    std::vector<int32_t> t2((size_t)P * H), t3((size_t)P * H);
    for (int h = 0; h < H; ++h)
        for (int q = 0; q < P; ++q) {
            t2[(size_t)q * H + h] = l.ssm.C_re[(size_t)h * P + q];
            t3[(size_t)q * H + h] = l.ssm.C_im[(size_t)h * P + q];
        }
    o.c_re_t = p.put(t2.data(), t2.size());
    o.c_im_t = p.put(t3.data(), t3.size());
    o.c24 = allow24 && fits24(t2.data(), t2.size()) && fits24(t3.data(), t3.size());
    o.D = p.put(l.ssm.D, H);
    {
        std::vector<int> all(P);
        for (int q = 0; q < P; ++q) all[q] = q;
        const ScanBounds sb = scan_bounds(l.ssm, all.data(), P, P, allow24);
        o.quad_ok = sb.quad_ok; o.quad_xmax = sb.quad_xmax; o.pair_ok = sb.pair_ok; o.pair_xmax = sb.pair_xmax;
    }
    pack_dense(p, l.out2, o.out2, allow24);
    o.l_bits = l.l_bits; o.l_exp = l.l_exp; o.r_bits = l.r_bits; o.r_exp = l.r_exp; o.res_bits = l.res_bits;
    o.res_exp = l.res_exp; o.sig_x = l.sig_x_exp; o.sig_y = l.sig_y_exp;
    std::memcpy(o.lut, l.lut, sizeof(o.lut));
}

int validate(const s5fxp_model_desc *d)
{
    if (!d || d->n_layers < 0 || (d->n_layers > 0 && !d->layers) || !dense_ok(d->encoder) || !dense_ok(d->decoder))
        return S5FXP_EBADARG;
    const int H = d->encoder.M;
    if (d->decoder.K != H) return S5FXP_EBADARG;
    if (mw_for(H) > MW_LIMIT_C || mw_for(d->decoder.M) > MW_LIMIT) return S5FXP_EUNSUPPORTED;
    if (8 + 8 * d->n_layers > S5FXP_STATUS_WORDS) return S5FXP_EUNSUPPORTED;
    for (int i = 0; i < d->n_layers; ++i) {
        const s5fxp_layer_desc &l = d->layers[i];
        const s5fxp_ssm_desc &s = l.ssm;
        if (s.H != H || s.P < 1 || !s.A_re || !s.A_im || !s.B_re || !s.B_im || !s.C_re || !s.C_im || !s.D ||
            !l.norm.minus_mean || !l.norm.invsq_var || !dense_ok(l.out2) || l.out2.K != H || l.out2.M != H)
            return S5FXP_EBADARG;
        if (mw_for(2 * s.P) > MW_LIMIT) return S5FXP_EUNSUPPORTED;
        // static shifts: a negative one is a ValueError (fxp_mul) or an undefined XLA shift (fxp_matmul)
        const int sh[] = {s.u_exp + s.B_re_exp - s.Bu_re_exp, s.u_exp + s.B_im_exp - s.Bu_im_exp,
                          s.x_re_exp + s.C_re_exp - s.y_exp,  s.x_im_exp + s.C_im_exp - s.y_exp,
                          s.D_exp + s.u_exp - s.y_exp,        l.l_exp + l.r_exp - l.res_exp,
                          s.A_re_exp,                         s.A_im_exp};
        for (int v : sh)
            if (!shift_ok(v)) return S5FXP_ENEGSHIFT;
        const int d1 = s.Bu_re_exp - s.x_re_exp, d2 = s.Bu_im_exp - s.x_im_exp;
        if (!shift_ok(d1 < 0 ? -d1 : d1) || !shift_ok(d2 < 0 ? -d2 : d2)) return S5FXP_ENEGSHIFT;
        if (l.sig_x_exp < 0 || l.sig_x_exp > 15 || l.sig_y_exp < 1 || l.sig_y_exp > 30) return S5FXP_EUNSUPPORTED;
    }
    return S5FXP_OK;
}

// time blocks (4 steps each) the recurrence kernels keep in flight: streams are padded to a multiple of it and followed by
// that many blocks of readable padding (the pair kernel's ring, the deepest, prefetches that far beyond its run)
constexpr int SCAN_DEPTH = S5_SCANP_ASM_DEPTH;
static_assert(S5_SCANP_ASM_DEPTH % S5_SCAN_ASM_DEPTH == 0, "one padding rule for all recurrence kernels");

} // namespace

#include "s5fxp_fast.hpp"

namespace {

// m == nullptr: size pass (an upper bound: it includes the MFMA-path tensors whenever the model is eligible)
size_t pack_all(const s5fxp_model_desc *d, Packer &p, s5fxp_model *m, bool allow24)
{
    DenseDev tmp_e, tmp_d;
    pack_dense(p, d->encoder, m ? m->enc : tmp_e, allow24);
    for (int i = 0; i < d->n_layers; ++i) {
        LayerDev tmp;
        pack_layer(p, d->layers[i], m ? m->layers[i] : tmp, allow24);
    }
    pack_dense(p, d->decoder, m ? m->dec : tmp_d, allow24);
    if ((allow24 || !m) && fast_eligible(d)) {
        if (m) m->fast = new FastModel();
        pack_fast(p, d, m ? m->fast : nullptr);
    }
    return p.off;
}

} // namespace

extern "C" size_t s5fxp_model_blob_bytes(const s5fxp_model_desc *desc)
{
    if (validate(desc) != S5FXP_OK) return 0;
    Packer p;
    return pack_all(desc, p, nullptr, false);
}

extern "C" int s5fxp_model_create(const s5fxp_model_desc *desc, void *dev_blob, size_t blob_bytes, int flags,
                                  void *stream, s5fxp_model **out)
{
    if (!out || !dev_blob) return S5FXP_EBADARG;
    *out = nullptr;
    if (flags & S5FXP_MODEL_FORCE_CSR) return S5FXP_EUNSUPPORTED; // see include/s5fxp.h: op level only
    int rc = validate(desc);
    if (rc) return rc;
    const size_t need = s5fxp_model_blob_bytes(desc);
    if (blob_bytes < need) return S5FXP_EWORKSPACE;
    s5fxp_model *m = new (std::nothrow) s5fxp_model();
    if (!m) return S5FXP_EBADARG;
    m->n_layers = desc->n_layers;
    m->d_in = desc->encoder.K;
    m->H = desc->encoder.M;
    m->P = desc->n_layers ? desc->layers[0].ssm.P : 0;
    m->d_out = desc->decoder.M;
    m->flags = flags;
    m->cfg = ModelCfg::from_env();
    m->layers.resize(desc->n_layers);
    std::vector<char> host(need);
    Packer p;
    p.host = host.data();
    p.dev = reinterpret_cast<char *>(dev_blob);
    pack_all(desc, p, m, !(flags & S5FXP_MODEL_FORCE_GENERIC));
    // pageable-memory async copies are staged by the runtime before returning, so `host` may go
    rc = hip_rc(hipMemcpyAsync(dev_blob, host.data(), need, hipMemcpyHostToDevice, S(stream)));
    if (rc) {
        delete m;
        return rc;
    }
    rc = hip_rc(hipStreamSynchronize(S(stream))); // creation is not on the hot path
    if (rc) {
        delete m;
        return rc;
    }
    *out = m;
    return S5FXP_OK;
}

extern "C" void s5fxp_model_destroy(s5fxp_model *m)
{
    if (m) delete m->fast;
    delete m;
}
extern "C" int s5fxp_model_out_exp(const s5fxp_model *m) { return m ? m->dec.out_exp : 0; }
extern "C" int s5fxp_model_out_bits(const s5fxp_model *m) { return m ? m->dec.out_bits : 0; }
/* 1 if the int8-MFMA path was packed for this model (it also needs L % 4 == 0 at run time) */
extern "C" int s5fxp_model_is_fast(const s5fxp_model *m) { return m && m->fast ? 1 : 0; }
extern "C" int s5fxp_model_recurrence_xmax(const s5fxp_model *m, int layer)
{
    const int k = s5fxp_model_recurrence_kernel(m, layer);
    if (k < 0) return -1;
    const LayerDev &l = m->layers[layer];
    // a plain forward (no traces, no carry) of a compactable layer runs on its live states: their bounds apply
    const bool compact = m->fast && m->fast->layers[layer].compact_ok && !m->cfg.no_compact;
    const int32_t pair_xmax = compact ? m->fast->layers[layer].c_bounds.pair_xmax : l.pair_xmax;
    const int32_t quad_xmax = compact ? m->fast->layers[layer].c_bounds.quad_xmax : l.quad_xmax;
    if (k >= 3) return pair_xmax;
    if (k == 2) return quad_xmax < 32766 ? quad_xmax : 32766;
    return l.quad_ok ? quad_xmax : 0;
}

extern "C" int s5fxp_model_recurrence_kernel(const s5fxp_model *m, int layer)
{
    if (!m || layer < 0 || layer >= m->n_layers) return -1;
    const LayerDev &l = m->layers[layer];
    if (!m->fast) return l.quad_ok && !(m->flags & S5FXP_MODEL_FORCE_GENERIC) ? 1 : 0;
    // the fused path: the same decision forward_fast takes (and reports in status word [8 + 8*layer + 5])
    const int code = select_rung(m, layer, S5FXP_FWD_DEFER_REDO, false, m->fast->layers[layer].compact_ok && !m->cfg.no_compact).code;
    return code == RK_EXACT ? 1 : code; // no fast rung applies: a quad kernel all the same, the 32-bit chain
}

namespace {

struct WsLayout {
    size_t hA, hB, bq, xs, x1, z, dyn, total;
    int TB; // time blocks per sequence in the scan-native streams (padded to a multiple of SCAN_DEPTH)
};
WsLayout ws_layout(const s5fxp_model *m, int B, int L)
{
    const size_t N = (size_t)B * L;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    WsLayout w{};
    w.TB = ((L + 3) / 4 + SCAN_DEPTH - 1) / SCAN_DEPTH * SCAN_DEPTH;
    size_t off = 0;
    const size_t nh = al(N * m->H * 4);
    // + SCAN_DEPTH blocks of padding: the recurrence kernel's ring buffer reads ahead of the last block
    const size_t ns = al(((size_t)B * w.TB + SCAN_DEPTH) * (m->P ? m->P : 1) * 8 * 4);
    w.hA = off; off += nh;
    w.hB = off; off += nh;
    w.bq = off; off += ns;
    w.xs = off; off += ns;
    w.x1 = off; off += nh;
    w.z = off; off += nh;
    w.dyn = off; off += al(sizeof(LayerDyn) * (size_t)(m->n_layers ? m->n_layers : 1));
    w.total = off;
    return w;
}
} // namespace

extern "C" size_t s5fxp_workspace_bytes(const s5fxp_model *m, int B, int L)
{
    if (!m || B < 1 || L < 1) return 0;
    const size_t g = ws_layout(m, B, L).total;
    const size_t f = m->fast ? fast_ws(m, B, L).total : 0;
    return g > f ? g : f;
}

namespace {

// FxpSequenceLayer.forward (fxpmodel.py:1110-1161) x [first, last) on the generic int32 kernels: BatchNorm exponents
// (reduce -> cross-rank max -> finalize per compute_best op), B projection, recurrence, C projection (+ exact re-run),
// out2 / sigmoid / gate, residual.  h: the first layer's input (read only), hn / h: the ping-pong pair the outputs go to;
// on return h points at the last layer's output and (hb, he) are its configuration.
struct GenericRun {
    const s5fxp_model *m;
    int B, L;
    WsLayout w;
    char *ws;
    LayerDyn *dyn;
    int32_t *status;
    const s5fxp_layer_trace *traces; // indexed by layer - trace_base
    int trace_base;
    const s5fxp_forward_opts *opts;
    hipStream_t st;
    void *stream;
};

int generic_layers(const GenericRun &g, int first, int last, int32_t *&h, int32_t *&hn, int &hb, DynExp &he)
{
    const s5fxp_model *m = g.m;
    const int B = g.B, L = g.L;
    const WsLayout &w = g.w;
    char *ws = g.ws;
    LayerDyn *dyn = g.dyn;
    int32_t *status = g.status;
    const s5fxp_layer_trace *traces = g.traces;
    hipStream_t st = g.st;
    void *stream = g.stream;
    s5fxp_allreduce_max_fn allreduce = g.opts ? g.opts->allreduce : nullptr;
    void *allreduce_ctx = g.opts ? g.opts->allreduce_ctx : nullptr;
    void **scan_events = g.opts ? g.opts->scan_events : nullptr;
    const int32_t *state_in = g.opts ? g.opts->state_in : nullptr;
    int32_t *state_out = g.opts ? g.opts->state_out : nullptr;
    auto I = [&](size_t off) { return reinterpret_cast<int32_t *>(ws + off); };
    const int64_t N = (int64_t)B * L;
    const int H = m->H, P = m->P;
    const int64_t NH = N * H;
    const unsigned tiles = (unsigned)((N + TN - 1) / TN);
    int rc;
    for (int li = first; li < last; ++li) {
        const LayerDev &l = m->layers[li];
        const s5fxp_layer_trace *tr = traces ? &traces[li - g.trace_base] : nullptr;
        LayerDyn *d = dyn + li;
        int32_t *st_exps = status + 8 + 8 * li;
        const s5fxp_ssm_desc &s = l.sd;
        auto mx = [](int a, int b) { return a > b ? a : b; };

        BnArgs bn{};
        bn.mm = l.mm; bn.isv = l.isv; bn.scale = l.scale; bn.bias = l.nbias;
        bn.xb = hb; bn.xe = he;
        bn.mb = l.nd.mean_bits; bn.me = l.nd.mean_exp; bn.b1 = mx(hb, bn.mb);
        bn.ib = l.nd.invsq_var_bits; bn.ie = l.nd.invsq_var_exp; bn.b2 = mx(bn.b1, bn.ib);
        bn.sb = l.nd.scale_bits; bn.se = l.nd.scale_exp; bn.b3 = l.scale ? mx(bn.b2, bn.sb) : bn.b2;
        bn.bb = l.nd.bias_bits; bn.be = l.nd.bias_exp; bn.b4 = l.nbias ? mx(bn.b3, bn.bb) : bn.b3;
        bn.ub = s.u_bits; bn.ue = s.u_exp; bn.out_bits = bn.b4; bn.dyn = d;

        // ---- BatchNorm exponents: reduce -> (cross-rank max) -> finalize, per compute_best op
        const unsigned rg = ew_grid(NH) > 2048 ? 2048 : ew_grid(NH);
        auto hook = [&](int slot, int n) -> int {
            return allreduce ? allreduce(allreduce_ctx, reinterpret_cast<float *>(d->mx + slot), n, stream) : 0;
        };
        hipLaunchKernelGGL(k_bn_reduce<1>, dim3(rg), dim3(256), 0, st, bn, h, NH, H, d);
        if (hook(0, 3)) return S5FXP_EHIP;
        hipLaunchKernelGGL(k_bn_finalize<1>, dim3(1), dim3(64), 0, st, bn, d, status, st_exps);
        hipLaunchKernelGGL(k_bn_reduce<2>, dim3(rg), dim3(256), 0, st, bn, h, NH, H, d);
        if (hook(3, 1)) return S5FXP_EHIP;
        hipLaunchKernelGGL(k_bn_finalize<2>, dim3(1), dim3(64), 0, st, bn, d, status, st_exps);
        if (l.scale) {
            hipLaunchKernelGGL(k_bn_reduce<3>, dim3(rg), dim3(256), 0, st, bn, h, NH, H, d);
            if (hook(4, 1)) return S5FXP_EHIP;
            hipLaunchKernelGGL(k_bn_finalize<3>, dim3(1), dim3(64), 0, st, bn, d, status, st_exps);
        }
        if (l.nbias) {
            hipLaunchKernelGGL(k_bn_reduce<4>, dim3(rg), dim3(256), 0, st, bn, h, NH, H, d);
            if (hook(5, 3)) return S5FXP_EHIP;
            hipLaunchKernelGGL(k_bn_finalize<4>, dim3(1), dim3(64), 0, st, bn, d, status, st_exps);
        }

        // ---- B projection (fused BatchNorm apply + change_cfg), writes the scan-native stream
        const int sh_re = s.Bu_re_exp - s.x_re_exp, sh_im = s.Bu_im_exp - s.x_im_exp;
        {
            BprojArgs a{};
            a.bn = bn; a.x = h; a.w = l.bcat; a.bq = I(w.bq);
            a.tr_bu_re = tr ? tr->Bu_re : nullptr; a.tr_bu_im = tr ? tr->Bu_im : nullptr;
            a.tr_pre_s5 = tr ? tr->pre_s5 : nullptr; a.tr_u = tr ? tr->u : nullptr;
            a.N = N; a.L = L; a.TB = w.TB; a.H = H; a.P = P; a.mw = mw_for(2 * P);
            a.rs_re = s.u_exp + s.B_re_exp - s.Bu_re_exp; a.rs_im = s.u_exp + s.B_im_exp - s.Bu_im_exp;
            a.bre_bits = s.Bu_re_bits; a.bim_bits = s.Bu_im_bits; a.sh_re = sh_re; a.sh_im = sh_im;
            if (l.b24 && s.u_bits <= 24 && bn.out_bits <= 24) S5_DISPATCH_MW(a.mw, true, k_bproj, tiles, st, a);
            else S5_DISPATCH_MW(a.mw, false, k_bproj, tiles, st, a);
        }
        // ---- recurrence.  Fast: quad kernel (3 dependent VALU ops / step), exact while |x| <= xmax; the C
        //      projection checks that bound on every state and, if it fails, the exact 32-bit kernels re-run.
        ScanArgs sl{};
        sl.bu_re = I(w.bq); sl.a_re = l.a_re; sl.a_im = l.a_im; sl.out_re = I(w.xs);
        sl.B = B; sl.L = L; sl.P = P; sl.TB = w.TB; sl.ea_re = s.A_re_exp; sl.ea_im = s.A_im_exp;
        const size_t plane = (size_t)B * P;
        sl.x0_re = state_in ? state_in + (size_t)li * 2 * plane : nullptr;
        sl.x0_im = state_in ? sl.x0_re + plane : nullptr;
        const unsigned lane_grid = (unsigned)(((int64_t)B * P + 63) / 64);
        int32_t xmax = (1 << 23) - 1; // the 24-bit C projection's own limit
        const bool quad = l.quad_ok && !(m->flags & S5FXP_MODEL_FORCE_GENERIC);
        if (scan_events && scan_events[2 * li] && (rc = hip_rc(hipEventRecord((hipEvent_t)scan_events[2 * li], st)))) return rc;
        if (quad) {
            ScanQuadArgs q{};
            q.bq = I(w.bq); q.xs = I(w.xs); q.a_re = l.a_re; q.a_im = l.a_im; q.B = B; q.TB = w.TB; q.P = P;
            q.ea_re = s.A_re_exp; q.ea_im = s.A_im_exp; q.x0_re = sl.x0_re; q.x0_im = sl.x0_im;
            hipLaunchKernelGGL(k_scan_quad_asm, dim3((unsigned)((int64_t)B * P / 16)), dim3(64), 0, st, q, GroupOff{});
            xmax = l.quad_xmax;
        } else {
            sl.run_if = nullptr;
            hipLaunchKernelGGL(k_scan_lane_native, dim3(lane_grid), dim3(64), 0, st, sl);
        }
        if (scan_events && scan_events[2 * li + 1] && (rc = hip_rc(hipEventRecord((hipEvent_t)scan_events[2 * li + 1], st)))) return rc;
        // ---- C projection + D*u + ReLU: pass 0 (24-bit multiplies, range check), then the exact re-run
        {
            CprojArgs a{};
            a.bn = bn; a.x = h; a.xs = I(w.xs); a.w_re = l.c_re_t; a.w_im = l.c_im_t; a.D = l.D;
            a.x1 = I(w.x1); a.tr_ys = tr ? tr->ys : nullptr; a.N = N; a.L = L; a.TB = w.TB; a.H = H; a.P = P;
            a.mw = mw_for(H);
            a.rs_re = s.x_re_exp + s.C_re_exp - s.y_exp; a.rs_im = s.x_im_exp + s.C_im_exp - s.y_exp;
            a.rs_d = s.D_exp + s.u_exp - s.y_exp; a.y_bits = s.y_bits; a.xmax = xmax; a.dynw = d; a.status = status;
            const bool c24 = l.c24 && !(m->flags & S5FXP_MODEL_FORCE_GENERIC);
            if (c24) S5_DISPATCH_MW_C(a.mw, true, 0, int32_t, tiles, st, a);
            else hipMemsetAsync(&d->redo, 0xff, 4, st); // no 24-bit variant: the exact pass does all the work
            if (quad) {
                sl.run_if = &d->redo;
                hipLaunchKernelGGL(k_scan_lane_native, dim3(lane_grid), dim3(64), 0, st, sl);
            }
            S5_DISPATCH_MW_C(a.mw, false, 1, int32_t, tiles, st, a);
            if (state_out) // the raw states are still in the stream (either pass): carry out = state after frame L-1
                hipLaunchKernelGGL(k_state_out, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, st, (const void *)I(w.xs), 0,
                                   (const int32_t *)nullptr, B, L, P, w.TB, state_out + (size_t)li * 2 * plane,
                                   state_out + (size_t)li * 2 * plane + plane, GroupOff{});
            if (tr && (tr->xs_re || tr->xs_im))
                hipLaunchKernelGGL(k_unpack_native, dim3(ew_grid(N * P)), dim3(256), 0, st, (const int32_t *)I(w.xs),
                                   tr->xs_re, tr->xs_im, B, L, P, w.TB);
        }
        // ---- out2 + sigmoid + gate + residual maxima
        {
            const DenseDev &o = l.out2;
            GateArgs a{};
            a.x1 = I(w.x1); a.w = o.w; a.bias = o.bias; a.skip = h; a.z = I(w.z);
            a.tr_out2 = tr ? tr->out2 : nullptr; a.tr_sig = tr ? tr->out2_sigmoid : nullptr;
            a.N = N; a.H = H; a.mw = mw_for(H); a.y_bits = s.y_bits; a.y_exp = s.y_exp;
            a.inp_bits = o.inp_bits; a.inp_exp = o.inp_exp; a.w_exp = o.w_exp; a.b_bits = o.b_bits; a.b_exp = o.b_exp;
            a.out_bits = o.out_bits; a.out_exp = o.out_exp; a.sig_x = l.sig_x; a.sig_y = l.sig_y;
            std::memcpy(a.lut, l.lut, sizeof(a.lut));
            a.l_bits = l.l_bits; a.l_exp = l.l_exp; a.r_bits = l.r_bits; a.r_exp = l.r_exp; a.res_bits = l.res_bits;
            a.res_exp = l.res_exp; a.rs_gate = l.l_exp + l.r_exp - l.res_exp;
            a.skip_bits = hb; a.skip_e = he; a.dynw = d; a.status = status;
            // exponent sanity that the reference would hit as a ValueError / undefined shift
            const bool conv = s.y_bits > o.inp_bits || s.y_exp > o.inp_exp;
            if (!shift_ok((conv ? o.inp_exp : s.y_exp) + o.w_exp - o.out_exp)) return S5FXP_ENEGSHIFT;
            if (o.x24 && s.y_bits <= 24) S5_DISPATCH_MW(a.mw, true, k_out2gate, tiles, st, a);
            else S5_DISPATCH_MW(a.mw, false, k_out2gate, tiles, st, a);
            if (tr && tr->post_GLU) hipMemcpyAsync(tr->post_GLU, a.z, (size_t)NH * 4, hipMemcpyDeviceToDevice, st);
        }
        if (hook(8, 3)) return S5FXP_EHIP;
        hipLaunchKernelGGL(k_res_finalize, dim3(1), dim3(64), 0, st, d, l.res_exp, he, l.res_bits, status, st_exps, 8);
        hipLaunchKernelGGL(k_resid, dim3(ew_grid(NH)), dim3(256), 0, st, (const int32_t *)I(w.z), (const int32_t *)h, hn,
                           tr ? tr->residadd : nullptr, NH, l.res_bits, hb, (const LayerDyn *)d);
        int32_t *sw = h; h = hn; hn = sw;
        hb = l.res_bits;
        he = DynExp{0, &d->res.eo};
    }

    (void)rc;
    return S5FXP_OK;
}

} // namespace

extern "C" int s5fxp_model_forward(const s5fxp_model *m, const int32_t *x, int x_bits, int x_exp, int B, int L,
                                   int32_t *y, void *workspace, size_t workspace_bytes, int32_t *status,
                                   const s5fxp_layer_trace *traces, const s5fxp_forward_opts *opts, void *stream)
{
    if (!m || !x || !y || !workspace || !status || B < 1 || L < 1 || x_bits < 1 || x_bits > 32) return S5FXP_EBADARG;
    const int G = opts && opts->groups > 1 ? opts->groups : 1;
    const size_t ws_one = s5fxp_workspace_bytes(m, B, L);
    if (workspace_bytes < (size_t)G * ws_one) return S5FXP_EWORKSPACE;
    if (G > 1) {
        // Grouped call: G independent reference batches of B sequences each.  The fused kernels take them in ONE set of
        // launches (gridDim.y = G) when nothing couples the groups on the host; otherwise one forward per group.
        if (m->fast && !traces && !opts->allreduce && fast_bn_ext(m) &&
            (((int64_t)L + 3) / 4 + 2 * SCAN_DEPTH) * (m->P ? m->P : 1) * 32 < 0xffffffffll)
            return forward_fast(m, x, x_bits, x_exp, B, L, y, workspace, status, traces, opts, S(stream), G, ws_one);
        const size_t plane = (size_t)m->n_layers * 2 * B * (m->P ? m->P : 1);
        for (int g = 0; g < G; ++g) {
            s5fxp_forward_opts o = *opts;
            o.groups = 1;
            if (o.state_in) o.state_in += g * plane;
            if (o.state_out) o.state_out += g * plane;
            const int rc = s5fxp_model_forward(m, x + (size_t)g * B * L * m->d_in, x_bits, x_exp, B, L, y + (size_t)g * B * L * m->d_out,
                                               reinterpret_cast<char *>(workspace) + g * ws_one, ws_one, status + (size_t)g * S5FXP_STATUS_WORDS,
                                               traces ? traces + (size_t)g * m->n_layers : nullptr, &o, stream);
            if (rc) return rc;
        }
        return S5FXP_OK;
    }
    // the recurrence kernels address one (sequence, state group) run of a stream through a 32-bit buffer extent
    if ((((int64_t)L + 3) / 4 + 2 * SCAN_DEPTH) * (m->P ? m->P : 1) * 32 >= 0xffffffffll) return S5FXP_EBADARG;
    if (m->fast) return forward_fast(m, x, x_bits, x_exp, B, L, y, workspace, status, traces, opts, S(stream));
    s5fxp_allreduce_max_fn allreduce = opts ? opts->allreduce : nullptr;
    void *allreduce_ctx = opts ? opts->allreduce_ctx : nullptr;
    void **scan_events = opts ? opts->scan_events : nullptr;
    const int32_t *state_in = opts ? opts->state_in : nullptr;
    int32_t *state_out = opts ? opts->state_out : nullptr;
    const WsLayout w = ws_layout(m, B, L);
    if (workspace_bytes < w.total) return S5FXP_EWORKSPACE;
    hipStream_t st = S(stream);
    char *ws = reinterpret_cast<char *>(workspace);
    auto I = [&](size_t off) { return reinterpret_cast<int32_t *>(ws + off); };
    LayerDyn *dyn = reinterpret_cast<LayerDyn *>(ws + w.dyn);
    const int64_t N = (int64_t)B * L;
    const int H = m->H, P = m->P;
    const int64_t NH = N * H;
    const unsigned tiles = (unsigned)((N + TN - 1) / TN);
    int rc;
    {
        StatusInit si{};
        si.path = S5FXP_PATH_GENERIC;
        for (int li = 0; li < m->n_layers; ++li) {
            si.rk[li] = m->layers[li].quad_ok && !(m->flags & S5FXP_MODEL_FORCE_GENERIC) ? 1 : 0;
            si.slots[li] = m->P;
            si.stream[li] = m->P;
        }
        hipLaunchKernelGGL(k_clear2, dim3(1), dim3(256), 0, st, status, (int)S5FXP_STATUS_WORDS, (int32_t *)nullptr, 0, si, m->n_layers, GroupOff{});
    }
    if ((rc = hip_rc(hipMemsetAsync(dyn, 0, sizeof(LayerDyn) * (size_t)(m->n_layers ? m->n_layers : 1), st)))) return rc;

    // ---- encoder + ReLU (fxpmodel.py:1263-1266)
    int32_t *h = I(w.hA), *hn = I(w.hB);
    {
        const DenseDev &e = m->enc;
        DenseArgs a{};
        a.x = x; a.w = e.w; a.bias = e.bias; a.y = h; a.N = N; a.K = e.K; a.M = e.M; a.mw = mw_for(e.M);
        a.xb = x_bits; a.xe = DynExp{x_exp, nullptr}; a.inp_bits = e.inp_bits; a.inp_exp = e.inp_exp; a.check_inp = 1;
        a.w_exp = e.w_exp; a.b_bits = e.b_bits; a.b_exp = e.b_exp; a.out_bits = e.out_bits; a.out_exp = e.out_exp;
        a.relu = 1; a.check24 = 1; a.status = status;
        const bool conv = x_bits > e.inp_bits || x_exp > e.inp_exp;
        if (!shift_ok((conv ? e.inp_exp : x_exp) + e.w_exp - e.out_exp)) return S5FXP_ENEGSHIFT;
        if (e.x24) S5_DISPATCH_MW(a.mw, true, k_dense, tiles, st, a);
        else S5_DISPATCH_MW(a.mw, false, k_dense, tiles, st, a);
    }
    int hb = m->enc.out_bits;
    DynExp he{m->enc.out_exp, nullptr};

    {
        GenericRun g{m, B, L, w, ws, dyn, status, traces, 0, opts, st, stream};
        if ((rc = generic_layers(g, 0, m->n_layers, h, hn, hb, he))) return rc;
    }

    // ---- decoder (fxpmodel.py:1437): its input exponent is the last residual's
    {
        const DenseDev &e = m->dec;
        DenseArgs a{};
        a.x = h; a.w = e.w; a.bias = e.bias; a.y = y; a.N = N; a.K = e.K; a.M = e.M; a.mw = mw_for(e.M);
        a.xb = hb; a.xe = he; a.inp_bits = e.inp_bits; a.inp_exp = e.inp_exp; a.check_inp = 1;
        a.w_exp = e.w_exp; a.b_bits = e.b_bits; a.b_exp = e.b_exp; a.out_bits = e.out_bits; a.out_exp = e.out_exp;
        a.relu = 0; a.check24 = 0; a.status = status;
        if (e.x24 && hb <= 24) S5_DISPATCH_MW(a.mw, true, k_dense, tiles, st, a);
        else S5_DISPATCH_MW(a.mw, false, k_dense, tiles, st, a);
    }
    return launch_rc();
}

// FxpSequenceLayer.forward for ONE layer of a created model (fxpmodel.py:1110-1161): BatchNorm -> SSM -> ReLU -> out2 ->
// sigmoid -> gate -> residual compute_best add -> ReLU, on the generic int32 kernels (exact for any int32 operands; the
// fused kernels exist for whole forwards, where the int16 inter-kernel planes pay off).
extern "C" int s5fxp_layer_forward(const s5fxp_model *m, int layer, const int32_t *x, int x_bits, int x_exp, int B, int L,
                                   int32_t *y, int32_t *y_exp_dev, void *workspace, size_t workspace_bytes, int32_t *status,
                                   const s5fxp_layer_trace *trace, const s5fxp_forward_opts *opts, void *stream)
{
    if (!m || !x || !y || !workspace || !status || layer < 0 || layer >= m->n_layers || B < 1 || L < 1 || x_bits < 1 || x_bits > 32)
        return S5FXP_EBADARG;
    if (opts && opts->groups > 1) return S5FXP_EUNSUPPORTED;
    if ((((int64_t)L + 3) / 4 + 2 * SCAN_DEPTH) * (m->P ? m->P : 1) * 32 >= 0xffffffffll) return S5FXP_EBADARG;
    const WsLayout w = ws_layout(m, B, L);
    if (workspace_bytes < w.total) return S5FXP_EWORKSPACE;
    hipStream_t st = S(stream);
    char *ws = reinterpret_cast<char *>(workspace);
    LayerDyn *dyn = reinterpret_cast<LayerDyn *>(ws + w.dyn);
    int rc;
    {
        StatusInit si{};
        si.path = S5FXP_PATH_GENERIC;
        si.rk[layer] = m->layers[layer].quad_ok && !(m->flags & S5FXP_MODEL_FORCE_GENERIC) ? 1 : 0;
        si.slots[layer] = m->P;
        si.stream[layer] = m->P;
        hipLaunchKernelGGL(k_clear2, dim3(1), dim3(256), 0, st, status, (int)S5FXP_STATUS_WORDS, (int32_t *)nullptr, 0, si, m->n_layers, GroupOff{});
    }
    if ((rc = hip_rc(hipMemsetAsync(dyn, 0, sizeof(LayerDyn) * (size_t)m->n_layers, st)))) return rc;
    // streaming carry of a single layer: the arrays are [1][2][B][P] here
    s5fxp_forward_opts o{};
    if (opts) o = *opts;
    const size_t plane2 = (size_t)2 * B * (m->P ? m->P : 1);
    if (o.state_in) o.state_in -= (size_t)layer * plane2;   // generic_layers indexes the carry by layer
    if (o.state_out) o.state_out -= (size_t)layer * plane2;
    int32_t *h = const_cast<int32_t *>(x), *hn = y; // h is only read
    int hb = x_bits;
    DynExp he{x_exp, nullptr};
    GenericRun g{m, B, L, w, ws, dyn, status, trace, layer, &o, st, stream};
    if ((rc = generic_layers(g, layer, layer + 1, h, hn, hb, he))) return rc;
    if (y_exp_dev && (rc = hip_rc(hipMemcpyAsync(y_exp_dev, he.dyn, sizeof(int32_t), hipMemcpyDeviceToDevice, st)))) return rc;
    return launch_rc();
}
extern "C" int s5fxp_model_live_states(const s5fxp_model *m, int layer)
{
    if (!m || layer < 0 || layer >= m->n_layers) return -1;
    if (m->fast) return m->fast->layers[layer].n_live;
    return m->P;
}
extern "C" int s5fxp_model_layer_out_bits(const s5fxp_model *m, int layer)
{
    return m && layer >= 0 && layer < m->n_layers ? m->layers[layer].res_bits : -1;
}

#ifdef S5_PHASE_PROF
// tools/prof_phases.py only (a -DS5_PHASE_PROF build): the accumulated phase clocks of k_enc_p
extern "C" int s5fxp_debug_phase_prof(long long *host_out, int n)
{
    return hip_rc(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_phase_prof), (size_t)n * sizeof(long long)));
}
#endif

#ifdef S5_GATE_CHECK
// experiment builds only (proj_p.hpp gate_check): reads (and optionally clears) the fragment counters
extern "C" int s5fxp_debug_gate_counts(unsigned long long out[4], int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(s5::g_gate_frag), 4 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        const unsigned long long z[4] = {0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(s5::g_gate_frag), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
