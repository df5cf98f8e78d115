// SL-lattice path DP kernels for gfx950.
//
// What is computed (reference: Planning/DP_Path/path_planner.py:66-155):
// every edge between adjacent lattice layers is the quintic
//     l(t) = l1 + (l2-l1) * h0(t/T),  h0(x) = 10x^3 - 15x^4 + 6x^5,  T = s2 - s1
// (zero end slopes/curvatures, path_planner.py:146), sampled at
// get_arange(s1, s2, d_s) and -- as the reference does -- evaluated at the
// ABSOLUTE arc length t = s, not s - s1.  Edge cost =
//     w_d*d_s*sum_k (l_k - mid)^2 + (1-w_d)*d_s*sum_k (l'_k^2 + l''_k^2 + l'''_k^2)
//     + 3*no_path_cost if any sample is closer than ob_dist to any obstacle.
//
// Because the sample abscissae are shared by all n_l^2 edges of a layer, the
// sums over k collapse to per-layer constants (A = sum h0, Bq = sum h0^2,
// H = sum h1^2+h2^2+h3^2) and, with u = l1 - mid, dl = l2 - l1,
//     edge = g(u) + dl*(2*cG*A*u + dl*(cG*Bq + cS*H)) + penalty,
// where g(u) = w_d*(d_s*np.sum(K copies of u^2)) is evaluated in NumPy's own
// summation order so that straight edges (dl = 0) are bit-identical to the
// reference and exact ties between lanes stay exact.  The collision test only
// looks at (sample, obstacle) pairs whose arc-length gap is below ob_dist;
// those few pairs are compacted into LDS once per layer.
//
// All arithmetic is fp64: lattice choices must be index-exact and the
// obstacle penalty (3e4) turns a single mis-rounded distance into a wrong path.
#include "mp_device.h"

#define SL_E_REG 8   // (sample, obstacle) pairs kept in registers on the fused path
#define SL_E_CAP 256 // LDS capacity of the compacted pair list

struct SlDev {
    mp_sl_batch in;
    mp_sl_params p;
    double *cost_map;
    int32_t *index_map;
    int32_t *path_idx;
    int32_t *status;
    int32_t *layers_done;
    double *best_cost;
    double *edge_cost; // K1 only
};

// Per-layer block-shared scalars.
struct LayerSh {
    double A;      // sum_k h0_k
    double Q;      // cG*sum h0^2 + cS*sum(h1^2+h2^2+h3^2)
    double delta;  // np.arange fill step
    int K;         // samples on every edge of the layer
    int n;         // np.arange length before the appended end point
    int n_ent;     // compacted (sample, obstacle) pairs
    int bad;       // MP_SL_E_* raised by this layer, 0 = fine
};

// Samples of one layer: abscissae xk, basis h0 and the per-sample smoothness
// term; then A and Q.  Block-cooperative; ends with a barrier.
__device__ void layer_tables(LayerSh *sh, double *xk, double *h0k, double *qk, double p1s, double p2s,
                             double d_s, double cG, double cS, int n_ob)
{
    const int tid = threadIdx.x;
    if (tid == 0) {
        int n;
        double delta;
        int K = arange_count(p1s, p2s, d_s, MP_SL_K_CAP, &n, &delta);
        sh->bad = 0;
        if (K < 0) { sh->bad = MP_SL_E_CAPACITY; K = 0; }
        if (K == 0 && n_ob > 0 && sh->bad == 0) sh->bad = MP_SL_E_EMPTY_EDGE;
        sh->K = K;
        sh->n = n;
        sh->delta = delta;
        sh->n_ent = 0;
    }
    __syncthreads();
    const int K = sh->K;
    const double T = p2s - p1s;
    for (int k = tid; k < K; k += blockDim.x) {
        double x = arange_value(k, sh->n, p1s, p2s, d_s, sh->delta);
        double tau = x / T;
        double t2 = tau * tau;
        double h0 = t2 * tau * (10.0 + tau * (-15.0 + 6.0 * tau));
        double w = tau - 1.0;
        double h1 = 30.0 * t2 * w * w / T;
        double h2 = 60.0 * tau * (1.0 + tau * (-3.0 + 2.0 * tau)) / (T * T);
        double h3 = 60.0 * (1.0 + tau * (-6.0 + 6.0 * tau)) / (T * T * T);
        xk[k] = x;
        h0k[k] = h0;
        qk[k] = h1 * h1 + h2 * h2 + h3 * h3;
    }
    __syncthreads();
    if (tid == 0) {
        double A = 0.0, Bq = 0.0, H = 0.0;
        for (int k = 0; k < K; ++k) {
            A += h0k[k];
            Bq += h0k[k] * h0k[k];
            H += qk[k];
        }
        sh->A = A;
        sh->Q = cG * Bq + cS * H;
    }
    __syncthreads();
}

// Compacts the (sample k, obstacle o) pairs with |x_k - ob_s| < ob_dist for
// obstacles [o_lo, o_hi) into LDS (arbitrary order; the collision test is an
// OR).  sh->n_ent must be 0 on entry; ends with a barrier.  n_ent may exceed
// SL_E_CAP: entries beyond the capacity are dropped and the caller re-chunks.
__device__ void build_entries(LayerSh *sh, const double *xk, const double *h0k, double *e_h0, double *e_ol,
                              double *e_ds2, const double *ob_s, const double *ob_l, int o_lo, int o_hi,
                              double r2)
{
    const int K = sh->K;
    for (int o = o_lo + threadIdx.x; o < o_hi; o += blockDim.x) {
        double os = ob_s[o], ol = ob_l[o];
        for (int k = 0; k < K; ++k) {
            double ds = xk[k] - os;
            double ds2 = ds * ds;
            if (ds2 < r2) {
                int slot = atomicAdd(&sh->n_ent, 1);
                if (slot < SL_E_CAP) {
                    e_h0[slot] = h0k[k];
                    e_ol[slot] = ol;
                    e_ds2[slot] = ds2;
                }
            }
        }
    }
    __syncthreads();
}

template <int NE>
struct EntReg {
    double h0[NE > 0 ? NE : 1], ol[NE > 0 ? NE : 1], ds2[NE > 0 ? NE : 1];
    // loads entries [e0, e0+NE) of the LDS list, padding with never-hit entries
    __device__ __forceinline__ void load(const double *e_h0, const double *e_ol, const double *e_ds2, int e0,
                                         int n_ent)
    {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            bool ok = e0 + e < n_ent;
            h0[e] = ok ? e_h0[e0 + e] : 0.0;
            ol[e] = ok ? e_ol[e0 + e] : 0.0;
            ds2[e] = ok ? e_ds2[e0 + e] : MP_INF;
        }
    }
    __device__ __forceinline__ bool hit(double l1, double dl, double r2) const
    {
        bool h = false;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            double ll = fma(dl, h0[e], l1);
            double d = ll - ol[e];
            double d2 = fma(d, d, ds2[e]);
            h |= d2 < r2;
        }
        return h;
    }
};

// Source-side per-layer arrays in LDS (4 doubles per source lane).
struct SrcSh {
    double *l;    // lateral offset of the source node
    double *g;    // straight-edge guide cost w_d*d_s*sum (l-mid)^2
    double *ua;   // 2*cG*A*(l-mid)
    double *cp;   // accumulated cost of the source node
};

__device__ __forceinline__ double edge_closed_form(double l2v, double l1, double g, double ua, double Q)
{
    double dl = l2v - l1;
    double t = fma(dl, Q, ua);
    return fma(dl, t, g);
}

// Relaxes the targets owned by this thread for one layer.
//   NE >= 0 : collision test fused in, NE entries in registers
//   NE <  0 : collision bits come from the LDS mask
template <int NE, int SJ>
__device__ __forceinline__ void relax_layer(const SlDev &a, int b, int s, int n_src, const SrcSh &src,
                                            const double *lrow, double Q, double r2, double pen,
                                            const double *e_h0, const double *e_ol, const double *e_ds2,
                                            int n_ent, const unsigned *mask, int mwords, double *cost_cur,
                                            unsigned char *idx8)
{
    const int n_l = a.in.n_l;
    const int sub = threadIdx.x % SJ;
    const int TL = blockDim.x / SJ;
    const int ch = (n_src + SJ - 1) / SJ;
    const int j_lo = sub * ch;
    const int j_hi = min(n_src, j_lo + ch);
    EntReg<(NE > 0 ? NE : 0)> ent;
    if constexpr (NE > 0) ent.load(e_h0, e_ol, e_ds2, 0, n_ent);
    for (int l2 = threadIdx.x / SJ; l2 < n_l; l2 += TL) {
        const double l2v = lrow[l2];
        double best = MP_INF;
        int bi = 0x7fffffff;
        for (int j = j_lo; j < j_hi; ++j) {
            const double l1 = src.l[j];
            double e = edge_closed_form(l2v, l1, src.g[j], src.ua[j], Q);
            bool hit;
            if constexpr (NE > 0) hit = ent.hit(l1, l2v - l1, r2);
            else if constexpr (NE == 0) hit = false;
            else hit = (mask[l2 * mwords + (j >> 5)] >> (j & 31)) & 1u;
            e += hit ? pen : 0.0;
            double c = e + src.cp[j];
            if (c < best) { best = c; bi = j; }
        }
        if (SJ > 1) group_argmin<SJ>(best, bi);
        if (sub == 0) {
            cost_cur[l2] = best;
            int32_t pi = (s == 0) ? -1 : bi;
            idx8[(size_t)s * n_l + l2] = (unsigned char)bi;
            size_t o = ((size_t)b * a.in.n_s + s) * n_l + l2;
            if (a.cost_map) a.cost_map[o] = best;
            if (a.index_map) a.index_map[o] = pi;
        }
    }
}

// Sets mask bit (l2, j) for every edge of the layer that collides with one of
// the entries currently in the LDS list (processed SL_E_REG at a time).
template <int SJ>
__device__ void mask_accumulate(int n_l, int n_src, const SrcSh &src, const double *lrow, double r2,
                                const double *e_h0, const double *e_ol, const double *e_ds2, int n_ent,
                                unsigned *mask, int mwords)
{
    const int sub = threadIdx.x % SJ;
    const int TL = blockDim.x / SJ;
    const int ch = (n_src + SJ - 1) / SJ;
    const int j_lo = sub * ch;
    const int j_hi = min(n_src, j_lo + ch);
    for (int e0 = 0; e0 < n_ent; e0 += SL_E_REG) {
        EntReg<SL_E_REG> ent;
        ent.load(e_h0, e_ol, e_ds2, e0, n_ent);
        for (int l2 = threadIdx.x / SJ; l2 < n_l; l2 += TL) {
            const double l2v = lrow[l2];
            for (int j = j_lo; j < j_hi; ++j) {
                const double l1 = src.l[j];
                if (ent.hit(l1, l2v - l1, r2)) atomicOr(&mask[l2 * mwords + (j >> 5)], 1u << (j & 31));
            }
        }
    }
}

__device__ __forceinline__ double block_min(double v, double *red)
{
    v = wave_min_f64(v);
    const int wave = threadIdx.x / MP_WAVE, nw = (blockDim.x + MP_WAVE - 1) / MP_WAVE;
    __syncthreads();
    if ((threadIdx.x & (MP_WAVE - 1)) == 0) red[wave] = v;
    __syncthreads();
    double r = red[0];
    for (int w = 1; w < nw; ++w) r = red[w] < r ? red[w] : r;
    return r;
}

// Fused SL plan: one workgroup per scenario walks the layers; edge costs never
// leave the chip.  Dynamic LDS layout (doubles first):
//   cost_cur[n_l] src.l[n_l] src.g[n_l] src.ua[n_l] src.cp[n_l] lrow[n_l]
//   xk[K_CAP] h0k[K_CAP] qk[K_CAP] e_h0[E_CAP] e_ol[E_CAP] e_ds2[E_CAP] red[16]
//   LayerSh | mask[n_l*mwords] u32 | idx8[n_s*n_l] u8
template <int SJ>
__global__ void sl_plan_kernel(SlDev a)
{
    extern __shared__ double smem[];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int n_l = a.in.n_l, n_s = a.in.n_s;
    const int mwords = (n_l + 31) / 32;
    double *cost_cur = smem;
    SrcSh src;
    src.l = cost_cur + n_l;
    src.g = src.l + n_l;
    src.ua = src.g + n_l;
    src.cp = src.ua + n_l;
    double *lrow = src.cp + n_l;
    double *xk = lrow + n_l;
    double *h0k = xk + MP_SL_K_CAP;
    double *qk = h0k + MP_SL_K_CAP;
    double *e_h0 = qk + MP_SL_K_CAP;
    double *e_ol = e_h0 + SL_E_CAP;
    double *e_ds2 = e_ol + SL_E_CAP;
    double *red = e_ds2 + SL_E_CAP;
    LayerSh *sh = (LayerSh *)(red + 16);
    unsigned *mask = (unsigned *)(sh + 1);
    unsigned char *idx8 = (unsigned char *)(mask + (size_t)n_l * mwords);

    const double *s_layers = a.in.s_layers + (size_t)b * n_s;
    const double *lbase = a.in.l_map + (size_t)b * a.in.l_bstride;
    const double *ob_s = a.in.ob_s + (size_t)b * a.in.ob_stride;
    const double *ob_l = a.in.ob_l + (size_t)b * a.in.ob_stride;
    const int n_ob = a.in.n_ob[b];
    const double r = a.in.ob_dist[b], r2 = r * r;
    const double d_s = a.p.d_s, w_d = a.p.w_d;
    const double cG = w_d * d_s, cS = (1.0 - w_d) * d_s;
    const double pen = 3.0 * a.p.no_path_cost, abort_thr = 10.0 * a.p.no_path_cost;
    const int mid_l = a.p.mid_l;
    const double ego_s = a.in.ego_sl[2 * b], ego_l = a.in.ego_sl[2 * b + 1];

    int status = MP_SL_FOUND;
    int done = 0;
    for (int s = 0; s < n_s; ++s) {
        const double p1s = (s == 0) ? ego_s : s_layers[s - 1];
        const double p2s = s_layers[s];
        const int n_src = (s == 0) ? 1 : n_l;
        const double *lprev_g = lbase + (size_t)(s == 0 ? n_s - 1 : s - 1) * a.in.l_rstride;
        const double mid = lprev_g[mid_l];
        layer_tables(sh, xk, h0k, qk, p1s, p2s, d_s, cG, cS, n_ob);
        if (sh->bad) { status = sh->bad; break; }
        const int K = sh->K;
        const double A = sh->A, Q = sh->Q;
        // source-side arrays; cost_cur still holds the previous layer's row
        for (int j = tid; j < n_src; j += nt) {
            double l1 = (s == 0) ? ego_l : lprev_g[j];
            double u = l1 - mid;
            src.l[j] = l1;
            src.g[j] = __dmul_rn(w_d, __dmul_rn(d_s, np_sum_const(__dmul_rn(u, u), K)));
            src.ua[j] = 2.0 * cG * A * u;
            src.cp[j] = (s == 0) ? 0.0 : cost_cur[j];
        }
        const double *lcur_g = lbase + (size_t)s * a.in.l_rstride;
        for (int l = tid; l < n_l; l += nt) lrow[l] = lcur_g[l];
        build_entries(sh, xk, h0k, e_h0, e_ol, e_ds2, ob_s, ob_l, 0, n_ob, r2); // barrier inside
        const int n_ent = sh->n_ent;
        if (n_ent <= SL_E_REG) {
            switch (n_ent) {
            case 0: relax_layer<0, SJ>(a, b, s, n_src, src, lrow, Q, r2, pen, e_h0, e_ol, e_ds2, n_ent, mask, mwords, cost_cur, idx8); break;
            case 1: relax_layer<1, SJ>(a, b, s, n_src, src, lrow, Q, r2, pen, e_h0, e_ol, e_ds2, n_ent, mask, mwords, cost_cur, idx8); break;
            case 2: relax_layer<2, SJ>(a, b, s, n_src, src, lrow, Q, r2, pen, e_h0, e_ol, e_ds2, n_ent, mask, mwords, cost_cur, idx8); break;
            case 3: relax_layer<3, SJ>(a, b, s, n_src, src, lrow, Q, r2, pen, e_h0, e_ol, e_ds2, n_ent, mask, mwords, cost_cur, idx8); break;
            case 4: relax_layer<4, SJ>(a, b, s, n_src, src, lrow, Q, r2, pen, e_h0, e_ol, e_ds2, n_ent, mask, mwords, cost_cur, idx8); break;
            case 5:
            case 6: relax_layer<6, SJ>(a, b, s, n_src, src, lrow, Q, r2, pen, e_h0, e_ol, e_ds2, n_ent, mask, mwords, cost_cur, idx8); break;
            default: relax_layer<8, SJ>(a, b, s, n_src, src, lrow, Q, r2, pen, e_h0, e_ol, e_ds2, n_ent, mask, mwords, cost_cur, idx8); break;
            }
        } else {
            // general path: collision bits accumulated into the LDS mask, in
            // obstacle chunks small enough for the pair list
            for (int w = tid; w < n_l * mwords; w += nt) mask[w] = 0u;
            __syncthreads();
            if (n_ent <= SL_E_CAP) {
                mask_accumulate<SJ>(n_l, n_src, src, lrow, r2, e_h0, e_ol, e_ds2, n_ent, mask, mwords);
            } else {
                const int oc = max(1, SL_E_CAP / K);
                for (int o0 = 0; o0 < n_ob; o0 += oc) {
                    __syncthreads();
                    if (tid == 0) sh->n_ent = 0;
                    __syncthreads();
                    build_entries(sh, xk, h0k, e_h0, e_ol, e_ds2, ob_s, ob_l, o0, min(n_ob, o0 + oc), r2);
                    mask_accumulate<SJ>(n_l, n_src, src, lrow, r2, e_h0, e_ol, e_ds2, sh->n_ent, mask, mwords);
                }
            }
            __syncthreads();
            relax_layer<-1, SJ>(a, b, s, n_src, src, lrow, Q, r2, pen, e_h0, e_ol, e_ds2, n_ent, mask, mwords, cost_cur, idx8);
        }
        __syncthreads();
        done = s + 1;
        if (s >= 1) {
            double m = MP_INF;
            for (int l = tid; l < n_l; l += nt) m = cost_cur[l] < m ? cost_cur[l] : m;
            m = block_min(m, red);
            if (m > abort_thr) {
                if (status == MP_SL_FOUND) status = MP_SL_NOT_FOUND;
                if (a.p.early_exit) break;
            }
        }
    }
    __syncthreads();
    // rows that were never relaxed keep the reference's initial values
    for (int i = done * n_l + tid; i < n_s * n_l; i += nt) {
        size_t o = (size_t)b * n_s * n_l + i;
        if (a.cost_map) a.cost_map[o] = 0.0;
        if (a.index_map) a.index_map[o] = -2;
    }
    if (tid < MP_WAVE) {
        // end node: first argmin of the last row, by one wave
        double c = MP_INF;
        int i = 0x7fffffff;
        if (status == MP_SL_FOUND) {
            for (int l = tid; l < n_l; l += MP_WAVE)
                if (cost_cur[l] < c) { c = cost_cur[l]; i = l; }
            group_argmin<MP_WAVE>(c, i);
        }
        int32_t *pidx = a.path_idx + (size_t)b * n_s;
        if (status == MP_SL_FOUND) {
            if (tid == 0) {
                int end_l = i;
                pidx[n_s - 1] = end_l;
                for (int q = n_s - 1; q >= 1; --q) {
                    end_l = idx8[(size_t)q * n_l + end_l];
                    pidx[q - 1] = end_l;
                }
            }
        } else {
            for (int q = tid; q < n_s; q += MP_WAVE) pidx[q] = -1;
        }
        if (tid == 0) {
            a.status[b] = status;
            a.layers_done[b] = done;
            if (a.best_cost) a.best_cost[b] = (status == MP_SL_FOUND) ? c : __builtin_nan("");
        }
    }
}

// K1: every edge cost of one (scenario, layer) per workgroup, written with the
// source index fastest so that consecutive lanes store consecutive doubles.
__global__ void sl_edge_cost_kernel(SlDev a)
{
    __shared__ double xk[MP_SL_K_CAP], h0k[MP_SL_K_CAP], qk[MP_SL_K_CAP];
    __shared__ double e_h0[SL_E_CAP], e_ol[SL_E_CAP], e_ds2[SL_E_CAP];
    __shared__ LayerSh sh;
    const int s = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, nt = blockDim.x;
    const int n_l = a.in.n_l, n_s = a.in.n_s;
    const double *s_layers = a.in.s_layers + (size_t)b * n_s;
    const double *lbase = a.in.l_map + (size_t)b * a.in.l_bstride;
    const double *ob_s = a.in.ob_s + (size_t)b * a.in.ob_stride;
    const double *ob_l = a.in.ob_l + (size_t)b * a.in.ob_stride;
    const int n_ob = a.in.n_ob[b];
    const double r = a.in.ob_dist[b], r2 = r * r;
    const double d_s = a.p.d_s, w_d = a.p.w_d;
    const double cG = w_d * d_s, cS = (1.0 - w_d) * d_s;
    const double pen = 3.0 * a.p.no_path_cost;
    const double p1s = (s == 0) ? a.in.ego_sl[2 * b] : s_layers[s - 1];
    const double p2s = s_layers[s];
    const int n_src = (s == 0) ? 1 : n_l;
    const double *lprev_g = lbase + (size_t)(s == 0 ? n_s - 1 : s - 1) * a.in.l_rstride;
    const double *lcur_g = lbase + (size_t)s * a.in.l_rstride;
    const double mid = lprev_g[a.p.mid_l];
    double *out = a.edge_cost + ((size_t)b * n_s + s) * n_l * n_l;
    layer_tables(&sh, xk, h0k, qk, p1s, p2s, d_s, cG, cS, n_ob);
    if (sh.bad) {
        if (tid == 0) atomicMin(&a.status[b], sh.bad);
        for (int e = tid; e < n_l * n_l; e += nt) out[e] = __builtin_nan("");
        return;
    }
    const int K = sh.K;
    const double A = sh.A, Q = sh.Q;
    build_entries(&sh, xk, h0k, e_h0, e_ol, e_ds2, ob_s, ob_l, 0, n_ob, r2);
    const int n_ent = sh.n_ent;
    for (int e = tid; e < n_l * n_l; e += nt) {
        const int l2 = e / n_l, j = e - l2 * n_l;
        if (j >= n_src) { out[e] = __builtin_nan(""); continue; }
        const double l1 = (s == 0) ? a.in.ego_sl[2 * b + 1] : lprev_g[j];
        const double l2v = lcur_g[l2];
        const double u = l1 - mid;
        const double g = __dmul_rn(w_d, __dmul_rn(d_s, np_sum_const(__dmul_rn(u, u), K)));
        const double ua = 2.0 * cG * A * u;
        double c = edge_closed_form(l2v, l1, g, ua, Q);
        const double dl = l2v - l1;
        bool hit = false;
        if (n_ent <= SL_E_CAP) {
            for (int q = 0; q < n_ent; ++q) {
                double ll = fma(dl, e_h0[q], l1);
                double d = ll - e_ol[q];
                hit |= fma(d, d, e_ds2[q]) < r2;
            }
        } else {
            // pair list overflowed: test every (sample, obstacle) pair directly
            for (int o = 0; o < n_ob; ++o) {
                double os = ob_s[o], ol = ob_l[o];
                for (int k = 0; k < K; ++k) {
                    double ds = xk[k] - os;
                    double ds2 = ds * ds;
                    if (ds2 < r2) {
                        double ll = fma(dl, h0k[k], l1);
                        double d = ll - ol;
                        hit |= fma(d, d, ds2) < r2;
                    }
                }
            }
        }
        c += hit ? pen : 0.0;
        out[e] = c;
    }
}

// K2: layer-wise min-relaxation over a materialised edge-cost tensor.  One
// workgroup per scenario; each wave owns target rows, its 64 lanes stride over
// the sources (coalesced reads) and the row's first-minimum is found with a
// wave-level shuffle argmin.  LDS: cost_prev[n_l] cost_cur[n_l] red[16].
__global__ void sl_dp_kernel(int n_s, int n_l, const double *edge_cost, mp_sl_params p, double *cost_map,
                             int32_t *index_map, int32_t *path_idx, int32_t *status_out,
                             int32_t *layers_done, double *best_cost)
{
    extern __shared__ double smem[];
    double *cost_prev = smem, *cost_cur = smem + n_l, *red = cost_cur + n_l;
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & (MP_WAVE - 1), wave = tid / MP_WAVE, nw = nt / MP_WAVE;
    const double *ec = edge_cost + (size_t)b * n_s * n_l * n_l;
    int32_t *imap = index_map + (size_t)b * n_s * n_l;
    const double abort_thr = 10.0 * p.no_path_cost;
    int status = MP_SL_FOUND, done = 0;
    for (int s = 0; s < n_s; ++s) {
        const double *el = ec + (size_t)s * n_l * n_l;
        if (s == 0) {
            for (int l = tid; l < n_l; l += nt) {
                cost_cur[l] = el[(size_t)l * n_l];
                imap[l] = -1;
            }
        } else {
            for (int l2 = wave; l2 < n_l; l2 += nw) {
                const double *row = el + (size_t)l2 * n_l;
                double c = MP_INF;
                int i = 0x7fffffff;
                for (int j = lane; j < n_l; j += MP_WAVE) {
                    double v = row[j] + cost_prev[j];
                    if (v < c) { c = v; i = j; }
                }
                group_argmin<MP_WAVE>(c, i);
                if (lane == 0) {
                    cost_cur[l2] = c;
                    imap[(size_t)s * n_l + l2] = i;
                }
            }
        }
        __syncthreads();
        double m = MP_INF;
        for (int l = tid; l < n_l; l += nt) {
            double v = cost_cur[l];
            if (cost_map) cost_map[((size_t)b * n_s + s) * n_l + l] = v;
            cost_prev[l] = v;
            m = v < m ? v : m;
        }
        done = s + 1;
        m = block_min(m, red); // barriers inside also publish cost_prev
        if (s >= 1 && m > abort_thr) {
            if (status == MP_SL_FOUND) status = MP_SL_NOT_FOUND;
            if (p.early_exit) break;
        }
    }
    __syncthreads();
    for (int i = done * n_l + tid; i < n_s * n_l; i += nt) {
        if (cost_map) cost_map[(size_t)b * n_s * n_l + i] = 0.0;
        imap[i] = -2;
    }
    __threadfence_block();
    __syncthreads();
    if (tid < MP_WAVE) {
        double c = MP_INF;
        int i = 0x7fffffff;
        if (status == MP_SL_FOUND) {
            for (int l = tid; l < n_l; l += MP_WAVE)
                if (cost_prev[l] < c) { c = cost_prev[l]; i = l; }
            group_argmin<MP_WAVE>(c, i);
        }
        int32_t *pidx = path_idx + (size_t)b * n_s;
        if (status == MP_SL_FOUND) {
            if (tid == 0) {
                int end_l = i;
                pidx[n_s - 1] = end_l;
                for (int q = n_s - 1; q >= 1; --q) {
                    // written by other waves of this workgroup: read past the vector L1
                    end_l = __hip_atomic_load(&imap[(size_t)q * n_l + end_l], __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
                    pidx[q - 1] = end_l;
                }
            }
        } else {
            for (int q = tid; q < n_s; q += MP_WAVE) pidx[q] = -1;
        }
        if (tid == 0) {
            status_out[b] = status;
            layers_done[b] = done;
            if (best_cost) best_cost[b] = (status == MP_SL_FOUND) ? c : __builtin_nan("");
        }
    }
}

// ---------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------
extern void mp_set_error(const char *fmt, ...);

static size_t sl_plan_lds_bytes(int n_s, int n_l)
{
    size_t d = (size_t)6 * n_l + 3 * MP_SL_K_CAP + 3 * SL_E_CAP + 16;
    size_t bytes = d * sizeof(double) + sizeof(LayerSh);
    bytes += (size_t)n_l * ((n_l + 31) / 32) * sizeof(unsigned);
    bytes += (size_t)n_s * n_l;
    return (bytes + 15) & ~(size_t)15;
}

static int check_batch(const mp_sl_batch *in, const mp_sl_params *p)
{
    if (!in || !p) { mp_set_error("null batch/params"); return MP_E_BADARG; }
    if (in->B <= 0 || in->n_s <= 0 || in->n_l <= 0 || in->ob_stride < 0) {
        mp_set_error("non-positive B/n_s/n_l");
        return MP_E_BADARG;
    }
    if (!in->s_layers || !in->l_map || !in->ego_sl || !in->n_ob || !in->ob_dist ||
        (in->ob_stride > 0 && (!in->ob_s || !in->ob_l))) {
        mp_set_error("null input array");
        return MP_E_BADARG;
    }
    if (p->mid_l < 0 || p->mid_l >= in->n_l) { mp_set_error("mid_l out of range"); return MP_E_BADARG; }
    if (!(p->d_s > 0)) { mp_set_error("d_s must be positive"); return MP_E_BADARG; }
    return MP_OK;
}

static int launch_check(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        mp_set_error("%s: %s", what, hipGetErrorString(e));
        return MP_E_LAUNCH;
    }
    return MP_OK;
}

extern "C" int mp_sl_plan(const mp_sl_batch *batch, const mp_sl_params *params, double *cost_map,
                          int32_t *index_map, int32_t *path_idx, int32_t *status, int32_t *layers_done,
                          double *best_cost, void *stream)
{
    int rc = check_batch(batch, params);
    if (rc) return rc;
    if (!path_idx || !status || !layers_done) { mp_set_error("null output array"); return MP_E_BADARG; }
    const int n_l = batch->n_l, n_s = batch->n_s;
    if (n_l > 256 || (size_t)n_s * n_l > 65536) {
        mp_set_error("mp_sl_plan supports n_l <= 256 and n_s*n_l <= 65536; use mp_sl_edge_cost + mp_sl_dp");
        return MP_E_UNSUPPORTED;
    }
    SlDev a;
    a.in = *batch;
    a.p = *params;
    a.cost_map = cost_map;
    a.index_map = index_map;
    a.path_idx = path_idx;
    a.status = status;
    a.layers_done = layers_done;
    a.best_cost = best_cost;
    a.edge_cost = nullptr;
    const size_t lds = sl_plan_lds_bytes(n_s, n_l);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(batch->B);
    if (n_l <= 8) {
        hipLaunchKernelGGL(sl_plan_kernel<8>, grid, dim3(64), lds, st, a);
    } else if (n_l <= 32) {
        int threads = ((4 * n_l + 63) / 64) * 64;
        hipLaunchKernelGGL(sl_plan_kernel<4>, grid, dim3(threads), lds, st, a);
    } else {
        int threads = ((n_l + 63) / 64) * 64;
        hipLaunchKernelGGL(sl_plan_kernel<1>, grid, dim3(threads), lds, st, a);
    }
    return launch_check("mp_sl_plan");
}

__global__ void fill_i32_kernel(int32_t *p, int n, int32_t v)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

extern "C" int mp_sl_edge_cost(const mp_sl_batch *batch, const mp_sl_params *params, double *edge_cost,
                               int32_t *status, void *stream)
{
    int rc = check_batch(batch, params);
    if (rc) return rc;
    if (!edge_cost || !status) { mp_set_error("null output array"); return MP_E_BADARG; }
    if (batch->B > 65535) { mp_set_error("mp_sl_edge_cost: B > 65535"); return MP_E_UNSUPPORTED; }
    SlDev a;
    a.in = *batch;
    a.p = *params;
    a.cost_map = nullptr;
    a.index_map = nullptr;
    a.path_idx = nullptr;
    a.status = status;
    a.layers_done = nullptr;
    a.best_cost = nullptr;
    a.edge_cost = edge_cost;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(fill_i32_kernel, dim3((batch->B + 255) / 256), dim3(256), 0, st, status, batch->B,
                       MP_SL_FOUND);
    hipLaunchKernelGGL(sl_edge_cost_kernel, dim3(batch->n_s, batch->B), dim3(256), 0, st, a);
    return launch_check("mp_sl_edge_cost");
}

extern "C" int mp_sl_dp(int32_t B, int32_t n_s, int32_t n_l, const double *edge_cost,
                        const mp_sl_params *params, double *cost_map, int32_t *index_map,
                        int32_t *path_idx, int32_t *status, int32_t *layers_done, double *best_cost,
                        void *stream)
{
    if (B <= 0 || n_s <= 0 || n_l <= 0) { mp_set_error("non-positive B/n_s/n_l"); return MP_E_BADARG; }
    if (!edge_cost || !params || !index_map || !path_idx || !status || !layers_done) {
        mp_set_error("null array (index_map is required)");
        return MP_E_BADARG;
    }
    size_t lds = ((size_t)2 * n_l + 16) * sizeof(double);
    if (lds > 64 * 1024) { mp_set_error("mp_sl_dp: n_l too large"); return MP_E_UNSUPPORTED; }
    int threads = n_l >= 256 ? 512 : 256;
    hipLaunchKernelGGL(sl_dp_kernel, dim3(B), dim3(threads), lds, (hipStream_t)stream, n_s, n_l, edge_cost,
                       *params, cost_map, index_map, path_idx, status, layers_done, best_cost);
    return launch_check("mp_sl_dp");
}
