/*
 * s5fxp_ref.c -- scalar C restatement of the reference's fixed-point S5 forward.
 *
 * TEST INFRASTRUCTURE ONLY: this is the second, independently written half of the CPU
 * oracle (the first is oracle/fxp_oracle.py).  It is compiled by oracle/Makefile into
 * oracle/_build/libs5fxp_ref.so and may be loaded only by tests/, __graft_entry__.smoke()
 * and the cpu_baseline leg of bench.py.  The product (sparsernns_amd/) never links it.
 *
 * PARITY UNPINNED: the reference (stevenabreu7/SparseRNNs) ships no tests or golden vectors
 * for this path and JAX is not installed here, so neither oracle half can be checked against
 * the reference itself; they are checked against hand-derived known answers and against each
 * other.
 *
 * It consumes the INTEGER model (what the reference's export() emits: fxpmodel.py:368-393,
 * 819-847,946-968,1163-1207) and follows, op by op:
 *   fxparray.py:274-284 (rshift), 310-326 (change_exp), 232-271 (change_cfg), 346-357 (clip),
 *   386-466 (add, incl. "compute_best"), 573-637 (mul), 640-678 (matmul);
 *   fxpmodel.py:27-63 (relu), 97-144 (sigmoid LUT), 147-208 (scan), 331-366 (dense),
 *   610-794 (SSM forward), 890-944 (BatchNorm), 1110-1161 (layer), 1261-1271, 1431-1439.
 * All integer arithmetic is int32 with two's-complement wrap (JAX default, x64 off); all
 * "compute_best" maxima are float32.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int32_t K, M;
    const int32_t *w;    /* [K][M] */
    const int32_t *bias; /* [M] or NULL */
    int32_t w_exp, b_bits, b_exp, inp_bits, inp_exp, out_bits, out_exp;
} ref_dense;

typedef struct {
    int32_t H, P;
    const int32_t *a_re, *a_im; /* [P] */
    const int32_t *b_re, *b_im; /* [P][H] */
    const int32_t *c_re, *c_im; /* [H][P] */
    const int32_t *d;           /* [H] */
    int32_t a_re_exp, a_im_exp, b_re_exp, b_im_exp, c_re_exp, c_im_exp, d_exp;
    int32_t u_bits, u_exp, bu_re_bits, bu_re_exp, bu_im_bits, bu_im_exp;
    int32_t x_re_exp, x_im_exp, y_bits, y_exp;
} ref_ssm;

typedef struct {
    const int32_t *minus_mean, *invsq_var, *scale, *bias; /* [H]; scale/bias may be NULL */
    int32_t mean_bits, mean_exp, isv_bits, isv_exp, scale_bits, scale_exp, bias_bits, bias_exp;
} ref_bn;

typedef struct {
    ref_bn bn;
    ref_ssm ssm;
    ref_dense out2;
    int32_t l_bits, l_exp, r_bits, r_exp, res_bits, res_exp;
    int32_t sig_x_exp, sig_y_exp;
    int32_t lut[8];
} ref_layer;

typedef struct {
    int32_t n_layers;
    ref_dense enc;
    const ref_layer *layers;
    ref_dense dec;
} ref_model;

/* Optional per-layer capture of intermediates (any pointer may be NULL). */
typedef struct {
    int32_t *pre_s5, *u, *bu_re, *bu_im, *xs_re, *xs_im, *ys, *out2, *sigmoid, *post_glu, *residadd;
    int32_t pre_s5_exp, residadd_exp; /* written back */
} ref_layer_trace;

enum { REF_OK = 0, REF_BADARG = -1, REF_NEGSHIFT = -2, REF_UNSUPPORTED = -3 };

/* ---------------------------------------------------------------- int32 helpers (wrap) */
static inline int32_t w_add(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t w_sub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t w_mul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static inline int32_t w_shl(int32_t a, int s) { return (int32_t)((uint32_t)a << s); }
static inline int32_t w_asr(int32_t a, int s) { return a >> s; } /* arithmetic on every target we build for */
static inline int32_t sat(int32_t v, int bits)
{
    int32_t hi = (int32_t)((1u << (bits - 1)) - 1u), lo = -hi - 1;
    return v > hi ? hi : (v < lo ? lo : v);
}
/* fxparray.py:310-326: no clip when the exponent is unchanged, else clip at CURRENT bits */
static inline int32_t chexp(int32_t d, int bits, int e, int e2)
{
    if (e2 == e) return d;
    if (e2 > e) return sat(w_shl(d, e2 - e), bits);
    return sat(w_asr(d, e - e2), bits);
}
/* fxparray.py:232-271 (signed only) */
static inline int32_t chcfg(int32_t d, int bits, int e, int bits2, int e2)
{
    if (bits == bits2 && e == e2) return d;
    d = chexp(d, bits, e, e2);
    return bits > bits2 ? sat(d, bits2) : d;
}
static inline float tofloat(int32_t d, int e) { return (float)d / (float)(1u << e); }

/* max(0, int(ceil(log2(m + eps)))) with a correctly rounded float32 log2 */
static int intbits_f32(float m, float eps)
{
    volatile float v = m + eps;
    float l = (float)log2((double)v);
    int c = (int)ceilf(l);
    return c > 0 ? c : 0;
}

/* ---------------------------------------------------------------- dense (fxpmodel.py:331-366) */
static int dense_forward(const ref_dense *d, const int32_t *x, int x_bits, int x_exp, int64_t N, int32_t *y, int relu)
{
    int conv = (x_bits > d->inp_bits) || (x_exp > d->inp_exp);
    int xe = conv ? d->inp_exp : x_exp;
    int rs = xe + d->w_exp - d->out_exp;
    if (rs < 0 || rs > 31) return REF_NEGSHIFT;
    int K = d->K, M = d->M;
#pragma omp parallel
    {
        int32_t *xr = (int32_t *)malloc(sizeof(int32_t) * (size_t)K);
        int32_t *acc = (int32_t *)malloc(sizeof(int32_t) * (size_t)M);
#pragma omp for schedule(static)
        for (int64_t n = 0; n < N; ++n) {
            const int32_t *xn = x + n * K;
            for (int k = 0; k < K; ++k) xr[k] = conv ? chcfg(xn[k], x_bits, x_exp, d->inp_bits, d->inp_exp) : xn[k];
            for (int m = 0; m < M; ++m) acc[m] = 0;
            for (int k = 0; k < K; ++k) {
                int32_t xv = xr[k];
                if (xv == 0) continue;
                const int32_t *wr = d->w + (size_t)k * M;
                for (int m = 0; m < M; ++m) acc[m] = w_add(acc[m], w_mul(xv, wr[m]));
            }
            int32_t *yn = y + n * M;
            for (int m = 0; m < M; ++m) {
                int32_t v = sat(w_asr(acc[m], rs), d->out_bits);
                if (d->bias) v = sat(w_add(v, chexp(d->bias[m], d->b_bits, d->b_exp, d->out_exp)), d->out_bits);
                yn[m] = (relu && v < 0) ? 0 : v;
            }
        }
        free(xr);
        free(acc);
    }
    return REF_OK;
}

/* ---------------------------------------------------------------- compute_best ops */
/* fxparray.py:420-448; y is a per-channel vector (H) broadcast over N frames, or a full tensor */
static int add_cb(const int32_t *x, int xb, int xe, const int32_t *y, int yb, int ye, int y_is_vec, int64_t N, int H,
                  int ob, int32_t *out, int *out_exp, int relu)
{
    float m = 0.f, mx = 0.f, my = 0.f;
#pragma omp parallel for reduction(max : m, mx, my) schedule(static)
    for (int64_t n = 0; n < N; ++n)
        for (int h = 0; h < H; ++h) {
            float fx = tofloat(x[n * H + h], xe);
            float fy = tofloat(y_is_vec ? y[h] : y[n * H + h], ye);
            volatile float s = fx + fy;
            m = fmaxf(m, fabsf(s));
            mx = fmaxf(mx, fabsf(fx));
            my = fmaxf(my, fabsf(fy));
        }
    int ib = intbits_f32(m, 1e-6f);
    int eo = ob - ib - 1;
    int ia = intbits_f32(mx, 1e-8f), ia2 = intbits_f32(my, 1e-8f);
    if (ia2 > ia) ia = ia2;
    int ea = xe > ye ? xe : ye;
    int ba = ia + ea + 1;
    int xb2 = ba > xb ? ba : xb, yb2 = ba > yb ? ba : yb;
    if (eo < 0 || ea - eo > 31 || eo - ea > 31) return REF_NEGSHIFT;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n)
        for (int h = 0; h < H; ++h) {
            int32_t a = chcfg(x[n * H + h], xb, xe, xb2, ea);
            int32_t b = chcfg(y_is_vec ? y[h] : y[n * H + h], yb, ye, yb2, ea);
            int32_t s = w_add(a, b);
            if (eo > ea) s = w_shl(s, eo - ea);
            else if (eo < ea) s = w_asr(s, ea - eo);
            s = sat(s, ob);
            out[n * H + h] = (relu && s < 0) ? 0 : s;
        }
    *out_exp = eo;
    return REF_OK;
}

/* fxparray.py:601-637 with result_exp="compute_best"; y is a per-channel vector */
static int mul_cb(const int32_t *x, int xb, int xe, const int32_t *y, int yb, int ye, int64_t N, int H, int32_t *out,
                  int *out_bits, int *out_exp)
{
    float m = 0.f;
#pragma omp parallel for reduction(max : m) schedule(static)
    for (int64_t n = 0; n < N; ++n)
        for (int h = 0; h < H; ++h) {
            volatile float p = tofloat(x[n * H + h], xe) * tofloat(y[h], ye);
            m = fmaxf(m, fabsf(p));
        }
    int ob = xb > yb ? xb : yb;
    int eo = ob - intbits_f32(m, 1e-6f) - 1;
    int rs = xe + ye - eo;
    if (rs < 0) return REF_NEGSHIFT; /* ValueError in the reference, fxparray.py:619-621 */
    if (rs > 31 || eo < 0) return REF_NEGSHIFT;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n)
        for (int h = 0; h < H; ++h) out[n * H + h] = sat(w_asr(w_mul(x[n * H + h], y[h]), rs), ob);
    *out_bits = ob;
    *out_exp = eo;
    return REF_OK;
}

/* ---------------------------------------------------------------- SSM (fxpmodel.py:610-794) */
static inline int32_t shiftto(int32_t v, int e, int e2) { return e > e2 ? w_asr(v, e - e2) : w_shl(v, e2 - e); }

/* st: NULL, or this layer's carry [2][B][P] (re plane, im plane): the recurrence starts from it and leaves the state after
 * the last step there (fxpmodel.py:147-172 has the carry as an explicit argument; recurrent_loop passes zeros, :196-207) */
static int ssm_forward(const ref_ssm *s, const int32_t *xin, int xb, int xe, int B, int L, int32_t *ys,
                       ref_layer_trace *tr, int32_t *st)
{
    const int H = s->H, P = s->P;
    const int64_t N = (int64_t)B * L;
    int rs_bre = s->u_exp + s->b_re_exp - s->bu_re_exp, rs_bim = s->u_exp + s->b_im_exp - s->bu_im_exp;
    int rs_cre = s->x_re_exp + s->c_re_exp - s->y_exp, rs_cim = s->x_im_exp + s->c_im_exp - s->y_exp;
    int rs_d = s->d_exp + s->u_exp - s->y_exp;
    if (rs_bre < 0 || rs_bim < 0 || rs_cre < 0 || rs_cim < 0 || rs_d < 0) return REF_NEGSHIFT;
    if (rs_bre > 31 || rs_bim > 31 || rs_cre > 31 || rs_cim > 31 || rs_d > 31) return REF_NEGSHIFT;
    int fail = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        int32_t *u = (int32_t *)malloc(sizeof(int32_t) * (size_t)L * H);
        int32_t *bur = (int32_t *)malloc(sizeof(int32_t) * (size_t)L * P);
        int32_t *bui = (int32_t *)malloc(sizeof(int32_t) * (size_t)L * P);
        int32_t *xr = (int32_t *)malloc(sizeof(int32_t) * (size_t)L * P);
        int32_t *xi = (int32_t *)malloc(sizeof(int32_t) * (size_t)L * P);
        if (!u || !bur || !bui || !xr || !xi) { fail = 1; goto done; }
        /* u = change_cfg(input -> u.bits, u.exp)  (fxpmodel.py:620-624); Bu = u @ B^T (631-644) */
        for (int t = 0; t < L; ++t) {
            const int32_t *xn = xin + ((int64_t)b * L + t) * H;
            int32_t *un = u + (size_t)t * H;
            for (int h = 0; h < H; ++h) un[h] = chcfg(xn[h], xb, xe, s->u_bits, s->u_exp);
            for (int p = 0; p < P; ++p) {
                int32_t ar = 0, ai = 0;
                for (int h = 0; h < H; ++h) {
                    ar = w_add(ar, w_mul(un[h], s->b_re[p * H + h]));
                    ai = w_add(ai, w_mul(un[h], s->b_im[p * H + h]));
                }
                bur[(size_t)t * P + p] = sat(w_asr(ar, rs_bre), s->bu_re_bits);
                bui[(size_t)t * P + p] = sat(w_asr(ai, rs_bim), s->bu_im_bits);
            }
        }
        /* sequential recurrence, no clip (fxpmodel.py:147-172) */
        for (int p = 0; p < P; ++p) {
            int32_t sr = st ? st[(size_t)b * P + p] : 0, si = st ? st[((size_t)B + b) * P + p] : 0, Ar = s->a_re[p], Ai = s->a_im[p];
            for (int t = 0; t < L; ++t) {
                int32_t nr = w_add(w_sub(w_asr(w_mul(Ar, sr), s->a_re_exp), w_asr(w_mul(Ai, si), s->a_re_exp)),
                                   shiftto(bur[(size_t)t * P + p], s->bu_re_exp, s->x_re_exp));
                int32_t ni = w_add(w_add(w_asr(w_mul(Ar, si), s->a_im_exp), w_asr(w_mul(Ai, sr), s->a_im_exp)),
                                   shiftto(bui[(size_t)t * P + p], s->bu_im_exp, s->x_im_exp));
                sr = nr;
                si = ni;
                xr[(size_t)t * P + p] = sr;
                xi[(size_t)t * P + p] = si;
            }
            if (st) {
                st[(size_t)b * P + p] = sr;
                st[((size_t)B + b) * P + p] = si;
            }
        }
        if (tr) {
            size_t off = (size_t)b * L;
            if (tr->u) memcpy(tr->u + off * H, u, sizeof(int32_t) * (size_t)L * H);
            if (tr->bu_re) memcpy(tr->bu_re + off * P, bur, sizeof(int32_t) * (size_t)L * P);
            if (tr->bu_im) memcpy(tr->bu_im + off * P, bui, sizeof(int32_t) * (size_t)L * P);
            if (tr->xs_re) memcpy(tr->xs_re + off * P, xr, sizeof(int32_t) * (size_t)L * P);
            if (tr->xs_im) memcpy(tr->xs_im + off * P, xi, sizeof(int32_t) * (size_t)L * P);
        }
        /* complex ReLU through float32 (fxpmodel.py:30-45), C projection, x2, + D*u (746-793) */
        for (int t = 0; t < L; ++t) {
            int32_t *pr = xr + (size_t)t * P, *pi = xi + (size_t)t * P;
            for (int p = 0; p < P; ++p) {
                float fr = (float)pr[p], fi = (float)pi[p];
                int keep = (fr > 0.f) || (fr == 0.f && fi > 0.f);
                /* f32 -> s32 convert: truncating and saturating */
                int32_t qr = fr >= 2147483648.f ? INT32_MAX : (int32_t)fr;
                int32_t qi = fi >= 2147483648.f ? INT32_MAX : (int32_t)fi;
                pr[p] = keep ? qr : 0;
                pi[p] = keep ? qi : 0;
            }
            const int32_t *un = u + (size_t)t * H;
            int32_t *yn = ys + ((int64_t)b * L + t) * H;
            for (int h = 0; h < H; ++h) {
                int32_t ar = 0, ai = 0;
                for (int p = 0; p < P; ++p) {
                    ar = w_add(ar, w_mul(pr[p], s->c_re[h * P + p]));
                    ai = w_add(ai, w_mul(pi[p], s->c_im[h * P + p]));
                }
                int32_t cr = sat(w_asr(ar, rs_cre), s->y_bits), ci = sat(w_asr(ai, rs_cim), s->y_bits);
                int32_t cx = sat(w_add(cr, w_mul(ci, -1)), s->y_bits);
                int32_t cx2 = w_mul(cx, 2); /* not clipped, fxpmodel.py:765-767 */
                int32_t du = sat(w_asr(w_mul(s->d[h], un[h]), rs_d), s->y_bits);
                yn[h] = sat(w_add(cx2, du), s->y_bits);
            }
        }
    done:
        free(u); free(bur); free(bui); free(xr); free(xi);
    }
    (void)N;
    return fail ? REF_BADARG : REF_OK;
}

/* ---------------------------------------------------------------- sigmoid LUT (fxpmodel.py:97-144) */
static inline int32_t sigmoid_lut(int32_t x, int xb, int xe, int sx, int sy, const int32_t *lut)
{
    int32_t xx = chexp(x, xb, xe, sx);
    int32_t sign = xx > 0 ? 1 : -1;
    int32_t a = xx < 0 ? w_sub(0, xx) : xx;
    int32_t ind = w_asr(a, sx);
    if (ind > 6) ind = 6;
    int32_t mu = a & ((1 << sx) - 1);
    int32_t half = w_add(w_asr(w_mul((1 << sx) - mu, lut[ind]), sx), w_asr(w_mul(mu, lut[ind + 1]), sx));
    return w_add(1 << (sy - 1), w_mul(sign, half));
}

/* ---------------------------------------------------------------- whole model */
/* state: NULL, or [n_layers][2][B][P] int32, read and replaced (the streaming carry) */
int ref_forward_state(const ref_model *m, const int32_t *x, int x_bits, int x_exp, int B, int L, int32_t *y, int *y_bits,
                      int *y_exp, ref_layer_trace *traces, int nthreads, int32_t *state);

int ref_forward(const ref_model *m, const int32_t *x, int x_bits, int x_exp, int B, int L, int32_t *y, int *y_bits,
                int *y_exp, ref_layer_trace *traces, int nthreads)
{
    return ref_forward_state(m, x, x_bits, x_exp, B, L, y, y_bits, y_exp, traces, nthreads, NULL);
}

int ref_forward_state(const ref_model *m, const int32_t *x, int x_bits, int x_exp, int B, int L, int32_t *y, int *y_bits,
                      int *y_exp, ref_layer_trace *traces, int nthreads, int32_t *state)
{
    if (!m || !x || !y || B <= 0 || L <= 0) return REF_BADARG;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    const int H = m->enc.M;
    const int64_t N = (int64_t)B * L;
    size_t sz = sizeof(int32_t) * (size_t)N * H;
    int32_t *h = (int32_t *)malloc(sz), *t1 = (int32_t *)malloc(sz), *t2 = (int32_t *)malloc(sz);
    int32_t *ysb = (int32_t *)malloc(sz), *g = (int32_t *)malloc(sz);
    int rc = REF_BADARG;
    if (!h || !t1 || !t2 || !ysb || !g) goto out;
    rc = dense_forward(&m->enc, x, x_bits, x_exp, N, h, 1); /* encoder + ReLU, fxpmodel.py:1263-1266 */
    if (rc) goto out;
    int hb = m->enc.out_bits, he = m->enc.out_exp;
    for (int li = 0; li < m->n_layers; ++li) {
        const ref_layer *l = &m->layers[li];
        ref_layer_trace *tr = traces ? &traces[li] : NULL;
        /* BatchNorm: four compute_best ops (fxpmodel.py:890-944) */
        int b1 = hb > l->bn.mean_bits ? hb : l->bn.mean_bits, e1, b2, e2;
        rc = add_cb(h, hb, he, l->bn.minus_mean, l->bn.mean_bits, l->bn.mean_exp, 1, N, H, b1, t1, &e1, 0);
        if (rc) goto out;
        rc = mul_cb(t1, b1, e1, l->bn.invsq_var, l->bn.isv_bits, l->bn.isv_exp, N, H, t2, &b2, &e2);
        if (rc) goto out;
        int32_t *cur = t2, *oth = t1;
        if (l->bn.scale) {
            rc = mul_cb(cur, b2, e2, l->bn.scale, l->bn.scale_bits, l->bn.scale_exp, N, H, oth, &b2, &e2);
            if (rc) goto out;
            int32_t *sw = cur; cur = oth; oth = sw;
        }
        if (l->bn.bias) {
            int b3 = b2 > l->bn.bias_bits ? b2 : l->bn.bias_bits;
            rc = add_cb(cur, b2, e2, l->bn.bias, l->bn.bias_bits, l->bn.bias_exp, 1, N, H, b3, oth, &e2, 0);
            if (rc) goto out;
            b2 = b3;
            int32_t *sw = cur; cur = oth; oth = sw;
        }
        if (tr) {
            tr->pre_s5_exp = e2;
            if (tr->pre_s5) memcpy(tr->pre_s5, cur, sz);
        }
        rc = ssm_forward(&l->ssm, cur, b2, e2, B, L, ysb, tr, state ? state + (size_t)li * 2 * B * l->ssm.P : NULL);
        if (rc) goto out;
        if (tr && tr->ys) memcpy(tr->ys, ysb, sz);
        /* x1 = relu(y); g = sigmoid(out2(x1)); z = gate(x1, g)  (fxpmodel.py:1125-1137) */
        int64_t NH = N * H;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < NH; ++i) ysb[i] = ysb[i] < 0 ? 0 : ysb[i];
        rc = dense_forward(&l->out2, ysb, l->ssm.y_bits, l->ssm.y_exp, N, g, 0);
        if (rc) goto out;
        if (tr && tr->out2) memcpy(tr->out2, g, sz);
        int rs = l->l_exp + l->r_exp - l->res_exp;
        if (rs < 0 || rs > 31) { rc = REF_NEGSHIFT; goto out; }
        int32_t *z = oth; /* free buffer */
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < NH; ++i) {
            int32_t s = sigmoid_lut(g[i], l->out2.out_bits, l->out2.out_exp, l->sig_x_exp, l->sig_y_exp, l->lut);
            if (tr && tr->sigmoid) tr->sigmoid[i] = s;
            int32_t a = chcfg(ysb[i], l->ssm.y_bits, l->ssm.y_exp, l->l_bits, l->l_exp);
            int32_t bq = chcfg(s, l->out2.out_bits, l->sig_y_exp, l->r_bits, l->r_exp);
            z[i] = sat(w_asr(w_mul(a, bq), rs), l->res_bits);
        }
        if (tr && tr->post_glu) memcpy(tr->post_glu, z, sz);
        /* residual add with compute_best, then ReLU (fxpmodel.py:1147-1159) */
        int er;
        rc = add_cb(z, l->res_bits, l->res_exp, h, hb, he, 0, N, H, l->res_bits, cur, &er, 0);
        if (rc) goto out;
        if (tr) {
            tr->residadd_exp = er;
            if (tr->residadd) memcpy(tr->residadd, cur, sz);
        }
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < NH; ++i) h[i] = cur[i] < 0 ? 0 : cur[i];
        hb = l->res_bits;
        he = er;
    }
    rc = dense_forward(&m->dec, h, hb, he, N, y, 0);
    if (y_bits) *y_bits = m->dec.out_bits;
    if (y_exp) *y_exp = m->dec.out_exp;
out:
    free(h); free(t1); free(t2); free(ysb); free(g);
    return rc;
}

int ref_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
