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
// Strength reduction.  The sample abscissae are shared by all n_l^2 edges of a
// layer, so the sums over k collapse to per-layer constants (A = sum h0,
// Bq = sum h0^2, H = sum h1^2+h2^2+h3^2) and, with u = l1 - mid, dl = l2 - l1,
//     edge = g(u) + dl*(2*cG*A*u + dl*(cG*Bq + cS*H)) + penalty,
// where g(u) = w_d*(d_s*np.sum(K copies of u^2)) is evaluated in NumPy's own
// summation order so that straight edges (dl = 0) are bit-identical to the
// reference and exact ties between lanes stay exact.  The collision test only
// looks at (sample, obstacle) pairs whose arc-length gap is below ob_dist;
// those few pairs are compacted into LDS once per layer.
//
// Exact pruning.  edge >= cS*H*dl^2 (the guide part is a sum of squares), so a
// predecessor j of target l2 can only win if
//     min_j(cost_prev) + cS*H*dl^2 <= cost of a candidate already evaluated.
// Each target therefore evaluates its nearest predecessor first and then walks
// outwards only while dl^2 stays under that bound.  Because the reference
// evaluates the quintic outside [0, T] on every layer but the first, H grows
// like s^8 and the walk usually stops after a few lanes; results are identical
// to visiting all n_l predecessors (the bound carries explicit rounding slack).
//
// All arithmetic is fp64: lattice choices must be index-exact and the
// obstacle penalty (3e4) turns a single mis-rounded distance into a wrong path.
#include "mp_device.h"

#define SL_E_REG 6   // (sample, obstacle) pairs kept in registers on the fused path
#define SL_ROWS_COMPACT 16 // predecessor rows of the fused kernel's compact form (MP_SL_ROWS overrides)
#define SL_E_CAP 48  // LDS capacity of the compacted pair list

struct SlDev {
    mp_sl_batch in;
    mp_sl_params p;
    double *cost_map;
    int32_t *index_map;
    int32_t *path_idx;
    int32_t *status;
    int32_t *layers_done;
    double *best_cost;
    double *edge_cost; // edge-tensor kernel only
    int32_t pair_cap;  // fused kernel: (sample, obstacle) pairs the LDS pair store holds per scenario
    unsigned long long *n_evaluated; // counting instance of the fused kernel only: edges priced, summed over the batch
    const int32_t *order;            // fused kernel: workgroup i plans scenario order[i] (null: scenario i)
    int32_t idx_rows;                // fused kernel: predecessor rows its LDS holds (one per layer that walks)
};

// Internal status of the fused kernel's compact form (never leaves the library): the plan was
// found but more layers walked than the kernel had predecessor rows for, so the path could not
// be traced back; the launcher's second launch (every row in LDS) plans the scenario again.
#define SL_RETRY (-100)

struct LayerGeom {
    int K;         // samples on every edge of the layer (negative: MP_SL_E_* code)
    int n;         // np.arange length before the appended end point
    double delta;  // np.arange fill step
};

// Per-layer constants of the closed-form edge cost.
struct LayerConst {
    double A;      // sum_k h0_k
    double Q;      // cG*sum h0^2 + cS*sum(h1^2+h2^2+h3^2)
    double cSH;    // cS*sum(h1^2+h2^2+h3^2): lower bound of edge/dl^2
};

// Sample count of one layer.
__device__ __forceinline__ LayerGeom layer_geom(double p1s, double p2s, double d_s, int n_ob)
{
    LayerGeom g;
    g.K = arange_count(p1s, p2s, d_s, MP_SL_K_CAP, &g.n, &g.delta);
    if (g.K < 0) g.K = MP_SL_E_CAPACITY;
    else if (g.K == 0 && n_ob > 0) g.K = MP_SL_E_EMPTY_EDGE;
    return g;
}

// Basis value h0 = 10x^3 - 15x^4 + 6x^5 at x = t/T.
__device__ __forceinline__ double basis_h0(double t, double T)
{
    const double tau = t / T;
    return tau * tau * tau * (10.0 + tau * (-15.0 + 6.0 * tau));
}

// A, Q, cSH of one layer: one thread, samples in index order, so every kernel
// that calls this gets the same bits.
__device__ __forceinline__ LayerConst layer_const(const LayerGeom &g, double p1s, double p2s, double d_s,
                                                  double cG, double cS)
{
    const double T = p2s - p1s;
    // one division per layer: the derivatives' scale factors 1/T, 1/T^2, 1/T^3 (one thread
    // works through a layer's samples, and a double division is some thirty instructions)
    const double i1 = 1.0 / T, i2 = i1 * i1, i3 = i2 * i1;
    double sa = 0.0, sb = 0.0, sq = 0.0;
    for (int k = 0; k < g.K; ++k) {
        const double x = arange_value(k, g.n, p1s, p2s, d_s, g.delta);
        const double tau = x / T;
        const double t2 = tau * tau;
        const double h0 = t2 * tau * (10.0 + tau * (-15.0 + 6.0 * tau));
        const double w = tau - 1.0;
        const double h1 = 30.0 * t2 * w * w * i1;
        const double h2 = 60.0 * tau * (1.0 + tau * (-3.0 + 2.0 * tau)) * i2;
        const double h3 = 60.0 * (1.0 + tau * (-6.0 + 6.0 * tau)) * i3;
        sa += h0;
        sb += h0 * h0;
        sq += h1 * h1 + h2 * h2 + h3 * h3;
    }
    LayerConst c;
    c.A = sa;
    c.cSH = cS * sq;
    c.Q = cG * sb + cS * sq;
    return c;
}

// Calls f(k, x_k, (x_k - os)^2) for every sample k of a layer whose arc length lies
// within ob_dist of the obstacle's.  Only the samples around the obstacle are
// visited (the index window is conservative, the test itself uses the exact
// abscissa).
template <class F>
__device__ __forceinline__ void for_each_near_sample(const LayerGeom &g, double p1s, double p2s, double d_s,
                                                     double os, double r, double r2, F f)
{
    // samples lie in [p1s, p2s] up to rounding: an obstacle further than
    // ob_dist (plus slack) from that span cannot produce a pair
    const double rr = r * 1.000001 + 1e-9;
    if (os < p1s - rr || os > p2s + rr) return;
    int k_lo = 0, k_hi = g.K - 1;
    if (g.K > 8) {
        const double c = (os - p1s) / d_s, w = r / d_s + 2.0;
        k_lo = (c - w > 0.0) ? (int)(c - w) : 0;
        k_hi = (c + w < (double)(g.K - 2)) ? (int)(c + w) : g.K - 2;
    }
    auto test = [&](int k) {
        const double x = arange_value(k, g.n, p1s, p2s, d_s, g.delta);
        const double ds = x - os;
        const double ds2 = ds * ds;
        if (ds2 < r2) f(k, x, ds2);
    };
    for (int k = k_lo; k <= k_hi; ++k) test(k);
    if (k_hi < g.K - 1) test(g.K - 1); // the appended end point sits off the grid
}

// Appends the (sample k, obstacle) pairs of one obstacle to the LDS pair list of a
// layer (arbitrary order; the collision test is an OR).  *n_ent may exceed
// SL_E_CAP: entries beyond the capacity are dropped and callers fall back to
// testing every pair directly.
__device__ __forceinline__ void add_obstacle_entries(int *n_ent, const LayerGeom &g, double p1s, double p2s,
                                                     double d_s, double *e_h0, double *e_ol, double *e_ds2,
                                                     double os, double ol, double r, double r2)
{
    const double T = p2s - p1s;
    for_each_near_sample(g, p1s, p2s, d_s, os, r, r2, [&](int, double x, double ds2) {
        const int slot = atomicAdd(n_ent, 1);
        if (slot < SL_E_CAP) {
            e_h0[slot] = basis_h0(x, T);
            e_ol[slot] = ol;
            e_ds2[slot] = ds2;
        }
    });
}

// Pair list of one layer, obstacles strided over the workgroup starting at
// o_first.  *n_ent must be 0 on entry; the caller synchronises afterwards.
__device__ __forceinline__ void build_entries(int *n_ent, const LayerGeom &g, double p1s, double p2s, double d_s,
                                              double *e_h0, double *e_ol, double *e_ds2, const double *ob_s,
                                              const double *ob_l, int n_ob, double r, double r2,
                                              int o_first = threadIdx.x)
{
    for (int o = o_first; o < n_ob; o += blockDim.x)
        add_obstacle_entries(n_ent, g, p1s, p2s, d_s, e_h0, e_ol, e_ds2, ob_s[o], ob_l[o], r, r2);
}

// One collision test: sample at l1 + dl*h0 against an obstacle at lateral ol
// whose squared arc-length gap is ds2.
__device__ __forceinline__ bool pair_hit(double l1, double dl, double h0, double ol, double ds2, double r2)
{
    double ll = fma(dl, h0, l1);
    double d = ll - ol;
    return fma(d, d, ds2) < r2;
}

// Collision of one edge against everything.  NE >= 0: NE pairs in registers;
// NE == -1: the LDS pair list; NE == -2: every (sample, obstacle) pair (the
// list overflowed).
template <int NE>
struct Collider {
    double h0[NE > 0 ? NE : 1], ol[NE > 0 ? NE : 1], ds2[NE > 0 ? NE : 1];
    const double *e_h0, *e_ol, *e_ds2, *ob_s, *ob_l;
    LayerGeom g;
    double p1s, p2s, d_s, r2;
    int n_ent, n_ob;
    // NE == -1: the pair that hit last time (-1: none yet), tried first the next time; one per
    // side of the walk, whose two sides are usually blocked by different pairs
    mutable int last[2];
    __device__ __forceinline__ void init(const double *eh, const double *eo, const double *ed, int ne,
                                         const LayerGeom &g_, double p1s_, double p2s_, double d_s_,
                                         const double *obs, const double *obl, int nob, double r2_)
    {
        e_h0 = eh; e_ol = eo; e_ds2 = ed; n_ent = ne;
        g = g_; p1s = p1s_; p2s = p2s_; d_s = d_s_; ob_s = obs; ob_l = obl; n_ob = nob; r2 = r2_;
        last[0] = last[1] = -1;
        if constexpr (NE > 0) {
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                bool ok = e < ne; // pad with pairs that can never hit
                h0[e] = ok ? eh[e] : 0.0;
                ol[e] = ok ? eo[e] : 0.0;
                ds2[e] = ok ? ed[e] : MP_INF;
            }
        }
    }
    __device__ __forceinline__ bool hit(double l1, double dl, int side = 0) const
    {
        bool h = false;
        if constexpr (NE > 0) {
#pragma unroll
            for (int e = 0; e < NE; ++e) h |= pair_hit(l1, dl, h0[e], ol[e], ds2[e], r2);
        } else if constexpr (NE == -1) {
            // A walk along a blocked stretch of sources meets the same obstacle edge after edge:
            // the pair that hit last time is tried first, and only a miss looks at the list.
            const int lst = last[side];
            if (lst >= 0 && pair_hit(l1, dl, e_h0[lst], e_ol[lst], e_ds2[lst], r2)) return true;
            // in blocks of 8, so that a wave whose edges have all hit something stops early
            // (crowded layers have hundreds of pairs)
            for (int e0 = 0; e0 < n_ent && !h; e0 += 8) {
                const int e1 = e0 + 8 < n_ent ? e0 + 8 : n_ent;
                for (int e = e0; e < e1; ++e)
                    if (pair_hit(l1, dl, e_h0[e], e_ol[e], e_ds2[e], r2)) {
                        h = true;
                        last[side] = e;
                    }
            }
        } else if constexpr (NE == -2) {
            const double T = p2s - p1s;
            for (int k = 0; k < g.K; ++k) {
                const double x = arange_value(k, g.n, p1s, p2s, d_s, g.delta);
                const double hk = basis_h0(x, T);
                for (int o = 0; o < n_ob; ++o) {
                    const double ds = x - ob_s[o];
                    const double d2 = ds * ds;
                    if (d2 < r2) h |= pair_hit(l1, dl, hk, ob_l[o], d2, r2);
                }
            }
        }
        return h;
    }
    // True only if the edges into lateral offset l2 from EVERY source offset in [la, lb] hit
    // something: the sample of pair e sits at l1 + h0*(l2 - l1), linear in l1, so its squared
    // distance to the obstacle is largest at one end of the range; both ends inside the
    // radius with a margin far above rounding means every source in between hits as well.
    // Conservative (false when in doubt or when no pair list exists); looks at 32 pairs at most.
    __device__ __forceinline__ bool hit_from_all(double l2, double la, double lb) const
    {
        const double r2m = r2 * (1.0 - 1e-9);
        auto all = [&](double h0e, double ole, double ds2e) {
            const double da = fma(l2 - la, h0e, la) - ole, db = fma(l2 - lb, h0e, lb) - ole;
            const double a2 = da * da, b2 = db * db;
            return (a2 > b2 ? a2 : b2) + ds2e < r2m;
        };
        bool h = false;
        if constexpr (NE > 0) {
#pragma unroll
            for (int e = 0; e < NE; ++e) h |= all(h0[e], ol[e], ds2[e]);
        } else if constexpr (NE == -1) {
            const int n = n_ent < 32 ? n_ent : 32;
            for (int e = 0; e < n; ++e) h |= all(e_h0[e], e_ol[e], e_ds2[e]);
        }
        return h;
    }
};

// The (sample, obstacle) pairs of every layer of one scenario, kept in LDS for
// the whole plan.  A pair is a sample of a layer whose arc length lies within
// ob_dist of an obstacle's; it carries what the collision test needs (basis
// value of the sample, lateral offset of the obstacle, squared arc-length gap).
// Building them once, in parallel over the obstacles, leaves nothing about
// obstacles to do inside the sequential layer loop:
//   count()        every thread; adds each obstacle's pairs per layer to cntL
//   place(s_begin) every thread; lays the pairs of layers s_begin.. into the store
//                  from its start: prefix sum of the counts -> offL, then the fill.
//                  Layers that do not fit any more keep their count and get
//                  offL = cap; the layer loop calls place(s) again when it reaches
//                  the first of them (the earlier layers' pairs are dead by then).
//                  A layer that alone exceeds the store gets offL = PAIRS_TOO_MANY.
// cntL must be zero before count(); the caller synchronises between the calls.
#define PAIRS_TOO_MANY 0xFFFFu // offL of a layer that alone has more pairs than the store holds
struct PairStore {
    int n_s, n_ob, cap;
    bool ascending;                 // ego_s <= layP2[0] <= layP2[1] <= ...
    const double *ob_s, *ob_l;      // obstacles of the scenario (global memory)
    bool has_ob;                    // this thread's first obstacle, held in registers
    double my_os, my_ol;
    double ego_s, d_s, r, r2;
    const double *layP2;            // LDS: arc length of every layer
    const int *layK;                // LDS: sample count code of every layer
    int *cntL;                      // LDS: pairs per layer
    unsigned short *offL;           // LDS: first pair of the layer in the store
    unsigned *recL;                 // LDS (may be null): (pairs << 16 | first pair) of a stored layer, ~0u otherwise
    double *h0, *ol, *ds2;          // LDS: the store, cap entries each
    unsigned short *lay;            // LDS (may be null): the layer every stored pair belongs to

    __device__ __forceinline__ LayerGeom geo(int s) const
    {
        LayerGeom g;
        const int kn = layK[s];
        g.K = kn < 0 ? kn : (kn & 0xffff);
        g.n = kn < 0 ? 0 : g.K - (kn >> 16);
        const double p1s = (s == 0) ? ego_s : layP2[s - 1];
        g.delta = (p1s + d_s) - p1s;
        return g;
    }

    // f(s, g, p1s, p2s, os, ol) for every obstacle of this thread and every valid
    // layer >= s_begin whose span the obstacle can reach.  The lanes first look
    // through the layers together (a cheap span test), each noting where ITS layers
    // start; the work then runs per lane on its own layers, so a wave takes as many
    // steps as its busiest obstacle has layers (2-3), not one per layer of the lattice.
    template <class F>
    __device__ __forceinline__ void each_obstacle_layer(int s_begin, F f) const
    {
        const double rr = r * 1.000001 + 1e-9;
        auto visit = [&](int s, double os, double ol_) {
            const LayerGeom g = geo(s);
            if (g.K >= 0) f(s, g, (s == 0) ? ego_s : layP2[s - 1], layP2[s], os, ol_);
        };
        auto one = [&](bool active, double os, double ol_) {
            if (ascending) {
                // arc lengths ascend: the layers an obstacle can reach are those from the
                // first whose end is not below os - rr up to the last whose start is not
                // above os + rr; the first is found by bisection (the same number of steps
                // in every lane), the rest by walking on
                int lo = s_begin, hi = n_s; // first s in [s_begin, n_s) with !(os > layP2[s] + rr)
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (os > layP2[mid] + rr) lo = mid + 1; else hi = mid;
                }
                if (!active) return;
                for (int s = lo; s < n_s; ++s) {
                    const double p1 = (s == 0) ? ego_s : layP2[s - 1];
                    if (os < p1 - rr) break;
                    visit(s, os, ol_);
                }
                return;
            }
            int s_first = 0, cnt = 0;
            bool contiguous = true;
            double p1s = (s_begin == 0) ? ego_s : layP2[s_begin - 1];
            for (int s = s_begin; s < n_s; ++s) {
                const double p2s = layP2[s];
                if (active && !(os < p1s - rr || os > p2s + rr)) {
                    if (cnt == 0) s_first = s;
                    else if (s != s_first + cnt) contiguous = false;
                    ++cnt;
                }
                p1s = p2s;
            }
            if (contiguous) {
                for (int q = 0; q < cnt; ++q) visit(s_first + q, os, ol_);
            } else { // layers out of order: take them one by one
                for (int s = s_begin; s < n_s; ++s) {
                    const double q1 = (s == 0) ? ego_s : layP2[s - 1];
                    if (!(os < q1 - rr || os > layP2[s] + rr)) visit(s, os, ol_);
                }
            }
        };
        // every lane of a wave enters `one` the same number of times
        const int tid = threadIdx.x, nt = blockDim.x;
        // the obstacles of the next round are fetched before this round's are worked on
        bool act_n = nt + tid < n_ob;
        double os_n = act_n ? ob_s[nt + tid] : 0.0, ol_n = act_n ? ob_l[nt + tid] : 0.0;
        one(has_ob, my_os, my_ol);
        for (int o0 = nt; o0 < n_ob; o0 += nt) {
            const bool act = act_n;
            const double os = os_n, ol_ = ol_n;
            const int o = o0 + nt + tid;
            act_n = o < n_ob;
            os_n = act_n ? ob_s[o] : 0.0;
            ol_n = act_n ? ob_l[o] : 0.0;
            one(act, os, ol_);
        }
    }

    __device__ __forceinline__ void count() const
    {
        each_obstacle_layer(0, [&](int s, const LayerGeom &g, double p1s, double p2s, double os, double) {
            int c = 0;
            for_each_near_sample(g, p1s, p2s, d_s, os, r, r2, [&](int, double, double) { ++c; });
            if (c) atomicAdd(&cntL[s], c);
        });
    }

    __device__ __forceinline__ void place(int s_begin) const
    {
        const int tid = threadIdx.x;
        if (tid < MP_WAVE) {
            // first pair of every layer: exclusive prefix sum of the counts, by one wave;
            // the counters of the layers that get stored restart from 0 for the fill pass
            int carry = 0;
            for (int s0 = s_begin; s0 < n_s; s0 += MP_WAVE) {
                const int s = s0 + tid;
                const int c = s < n_s ? cntL[s] : 0;
                int incl = c;
                for (int m = 1; m < MP_WAVE; m <<= 1) {
                    const int up = __shfl_up(incl, m, MP_WAVE);
                    if (tid >= m) incl += up;
                }
                if (s < n_s) {
                    // only layers that fit entirely are stored (their counters restart for
                    // the fill); the others keep their count
                    const int off = carry + incl - c;
                    if (off + c <= cap) {
                        offL[s] = (unsigned short)off;
                        cntL[s] = 0;
                        if (recL) recL[s] = ((unsigned)c << 16) | (unsigned)off;
                    } else {
                        offL[s] = (s == s_begin) ? PAIRS_TOO_MANY : (unsigned short)cap;
                        if (recL) recL[s] = ~0u;
                    }
                }
                carry += __shfl(incl, MP_WAVE - 1, MP_WAVE);
            }
        }
        __syncthreads();
        if (offL[s_begin] == PAIRS_TOO_MANY) return; // nothing fits behind a layer that does not fit itself
        each_obstacle_layer(s_begin, [&](int s, const LayerGeom &g, double p1s, double p2s, double os, double ol_) {
            const double T = p2s - p1s;
            const int off = offL[s];
            if (off >= cap) return; // not stored this time
            for_each_near_sample(g, p1s, p2s, d_s, os, r, r2, [&](int, double x, double d2) {
                const int slot = off + atomicAdd(&cntL[s], 1);
                h0[slot] = basis_h0(x, T);
                ol[slot] = ol_;
                ds2[slot] = d2;
                if (lay) lay[slot] = (unsigned short)s;
            });
        });
        __syncthreads();
    }

    // Pairs of layer s for the layer loop (every thread, same values): lays the store
    // out again from layer s if it ran full before it.  *stored = false when this one
    // layer alone exceeds the store: its pairs must then be found another way.
    __device__ __forceinline__ int layer_pairs(int s, int *off_out, bool *stored) const
    {
        return layer_pairs(s, cntL[s], offL[s], off_out, stored);
    }

    // The same with cntL[s] / offL[s] already read by the caller (a layer ahead).
    __device__ __forceinline__ int layer_pairs(int s, int n, int off, int *off_out, bool *stored) const
    {
        if (off != PAIRS_TOO_MANY && off + n > cap) {
            __syncthreads();
            place(s);
            n = cntL[s];
            off = offL[s];
        }
        *stored = off != PAIRS_TOO_MANY;
        *off_out = *stored ? off : 0;
        return n;
    }
};

// Straight-edge guide cost w_d*(d_s*np.sum(K copies of (l1-mid)^2)).
__device__ __forceinline__ double guide_straight(double u, int K, double w_d, double d_s)
{
    return __dmul_rn(w_d, __dmul_rn(d_s, np_sum_const(__dmul_rn(u, u), K)));
}

// Closed-form edge cost without the obstacle penalty.
__device__ __forceinline__ double edge_closed_form(double dl, double g, double ua, double Q)
{
    double t = fma(dl, Q, ua);
    return fma(dl, t, g);
}

// Source-side per-layer arrays in LDS, 16 B per lane and array so that lanes
// reading consecutive sources touch every bank once.
struct SrcSh {
    double2 *lc;   // (lateral offset of the source node, its accumulated cost)
    double mid;    // centre of the source row
    double two_cga; // 2*cG*A of the layer
    int K;         // samples per edge of the layer
    double w_d, d_s;
};

// Relaxes target l2 = threadIdx.x (lateral offset l2v) of layer s: nearest
// predecessor first, then outward walks that stop once cS*H*dl^2 alone exceeds
// the best cost so far.  w_inv = 1/(cS*H) with its safety factor, or 0 when
// the bound does not apply.  Returns the accumulated cost; *bj_out = predecessor.
// COUNT: *n_eval is raised by one per edge priced (measurement build of the kernel).
template <int NE, bool BOUND_FIRST = false, bool COUNT = false>
__device__ __forceinline__ double relax_target(int n_src, const SrcSh &src, double l2v, double Q, double w_inv,
                                               double m_prev, double g_max, bool sorted, double pen,
                                               const Collider<NE> &col, int *bj_out, int *n_eval = nullptr,
                                               int *n_visit = nullptr, int *n_test = nullptr)
{
    const int l2 = threadIdx.x;
    const int j0 = min(l2, n_src - 1);
    double2 lc = src.lc[j0];
    double best;
    bool hit0;
    {
        const double l1 = lc.x, dl = l2v - lc.x;
        const double u = l1 - src.mid;
        double e = edge_closed_form(dl, guide_straight(u, src.K, src.w_d, src.d_s), src.two_cga * u, Q);
        hit0 = col.hit(l1, dl);
        e += hit0 ? pen : 0.0;
        best = e + lc.y;
    }
    int bj = j0;
    if constexpr (COUNT) ++*n_eval;
    // A target that cannot be entered without a collision (an obstacle reaches a place every
    // incoming edge passes, e.g. the node itself) pays the penalty on every edge.  Its window
    // would stay as wide as the penalty and the walk would price the whole row; instead the
    // penalty becomes part of every candidate's lower bound and drops out of the comparison.
    double pen_all = 0.0;
    if (hit0 && sorted && n_src > 1 && w_inv > 0.0 && col.hit_from_all(l2v, src.lc[0].x, src.lc[n_src - 1].x))
        pen_all = pen * (1.0 - 1e-12);
    // candidates with dl^2 above w2 cannot beat `best`: their cost is at least
    // m_prev + cSH*dl^2 (+ the penalty where no edge avoids it) up to rounding, which the
    // slack terms cover
    // (the window follows `best` down: a free edge found next to a blocked one ends the walk)
    auto window = [&](double bst) {
        double gap = (bst - m_prev) - pen_all;
        gap = (gap > 0.0 ? gap : 0.0) + 1e-12 * (g_max + fabs(bst));
        return gap * w_inv;
    };
    double w2 = w_inv > 0.0 ? window(best) : MP_INF;
    // BOUND_FIRST (layers with many pairs): a candidate is first priced without the obstacle
    // penalty (a lower bound of its cost, since the penalty only adds and rounding is
    // monotone) and its pairs are only looked at if that bound could still win.
    auto consider = [&](int j, double l1, double cp, double dl, int side) {
        bool hit = false;
        if constexpr (!BOUND_FIRST) {
            // the collision test first: a candidate that hits pays the penalty, and where its
            // own cost plus the penalty already leaves no room (a blocked target walking along
            // a blocked stretch of sources) the closed form is not needed
            if constexpr (COUNT) ++*n_test;
            hit = col.hit(l1, dl, side);
            if (hit && w_inv > 0.0) {
                double gap = (best - cp) - pen * (1.0 - 1e-12);
                gap = (gap > 0.0 ? gap : 0.0) + 1e-12 * (g_max + fabs(best));
                if (dl * dl > gap * w_inv) return;
            }
        }
        if constexpr (COUNT) ++*n_eval;
        const double u = l1 - src.mid;
        const double e = edge_closed_form(dl, guide_straight(u, src.K, src.w_d, src.d_s), src.two_cga * u, Q);
        double c;
        if constexpr (BOUND_FIRST) {
            c = e + cp;
            if (!(c < best || (c == best && j < bj))) return; // lower index wins ties
            if (col.hit(l1, dl, side)) c = (e + pen) + cp;
        } else {
            c = (e + (hit ? pen : 0.0)) + cp;
        }
        if (!(c < best || (c == best && j < bj))) return;
        best = c;
        bj = j;
        if (w_inv > 0.0) w2 = window(best);
    };
    // A candidate inside the window is priced only if its own accumulated cost leaves room:
    // its total is at least cp + cSH*dl^2 (the window's argument with cp in place of the
    // layer minimum), so where that already exceeds `best` the closed form and the collision
    // test are skipped.  On a layer whose costs rise steeply away from the ego (the first
    // ones) most of the window falls to this test.
    auto cannot_win = [&](double dl2, double cp) {
        double gap = (best - cp) - pen_all;
        gap = (gap > 0.0 ? gap : 0.0) + 1e-12 * (g_max + fabs(best));
        return w_inv > 0.0 && dl2 > gap * w_inv;
    };
    // outwards on both sides in turn, so that a free edge close by is found (and shrinks
    // the window) before the far side of a blocked stretch has been walked
    int jd = j0 - 1, ju = j0 + 1;
    while (jd >= 0 || ju < n_src) {
        if constexpr (COUNT) ++*n_visit;
        if (jd >= 0) {
            lc = src.lc[jd];
            const double dl = l2v - lc.x;
            if (dl * dl > w2) {
                jd = (sorted && dl > 0.0) ? -1 : jd - 1; // sorted: sources further down are further away
            } else {
                if (!cannot_win(dl * dl, lc.y)) consider(jd, lc.x, lc.y, dl, 0);
                --jd;
            }
        }
        if (ju < n_src) {
            lc = src.lc[ju];
            const double dl = l2v - lc.x;
            if (dl * dl > w2) {
                ju = (sorted && dl < 0.0) ? n_src : ju + 1;
            } else {
                if (!cannot_win(dl * dl, lc.y)) consider(ju, lc.x, lc.y, dl, 1);
                ++ju;
            }
        }
    }
    *bj_out = bj;
    return best;
}

// Layers scenario b uses: all n_s slots, or its own count when the batch has one.
__device__ __forceinline__ int scenario_layers(const mp_sl_batch &in, int b)
{
    if (!in.n_layers) return in.n_s;
    const int n = in.n_layers[b];
    return n < 0 ? 0 : (n > in.n_s ? in.n_s : n);
}

// Outputs of a scenario with no layers: initial maps, no path, MP_SL_SKIPPED.
__device__ __forceinline__ void write_skipped(const SlDev &a, int b, int n_sm, int n_l)
{
    for (int i = threadIdx.x; i < n_sm * n_l; i += blockDim.x) {
        const size_t o = (size_t)b * n_sm * n_l + i;
        if (a.cost_map) a.cost_map[o] = 0.0;
        if (a.index_map) a.index_map[o] = -2;
    }
    for (int q = threadIdx.x; q < n_sm; q += blockDim.x) a.path_idx[(size_t)b * n_sm + q] = -1;
    if (threadIdx.x == 0) {
        a.status[b] = MP_SL_SKIPPED;
        a.layers_done[b] = 0;
        if (a.best_cost) a.best_cost[b] = __builtin_nan("");
    }
}

// Fused SL plan: one workgroup per scenario (one thread per lane of the lattice)
// walks the layers; edge costs never leave the chip.  A prologue computes, in
// parallel, every layer's constants (one thread per layer) and the (sample,
// obstacle) pairs of EVERY layer (one thread per obstacle: count per layer,
// prefix sum, fill), so that nothing about obstacles is left to do inside the
// sequential layer loop: there a thread relaxes its target, publishes its node
// as a source of the next layer in the other half of a double-buffered LDS array,
// and a layer costs one barrier and touches no global memory unless the lateral
// grid changes from layer to layer (that row is prefetched a layer ahead).
// Dynamic LDS layout (n_lp = n_l rounded up to even, cap = pair_cap):
//   lc[2][n_lp] double2 | layA layQ layW layP2 layMid [n_s] each | pairs[3][cap] (basis value,
//   obstacle offset, squared arc-length gap) | red[2][8] redmax[8] | layK[n_s] int (K, bit 16: end
//   point appended) | cntL[n_s] int (pairs per layer) | excL[n_s] int, first_ex[2] int (abort
//   bookkeeping of layers without a barrier) | recL[n_s] u32 | offL[n_s] u16 (first pair of the layer)
//   | laySorted[n_s] u8 (padded to 4) | locL[n_s] u8 (padded to 4: layers that need no exchange)
//   | runL[n_s] u8 (padded to 4) | rowL[n_s] u8 (padded to 4: the layer's predecessor row, 0xff: none)
//   | pair_lay[cap] u16 (padded to 4: layer of every stored pair) | idx8[idx_rows*n_l] u8 (padded to 8)
//   | lrow[n_lp] double (the shared lateral row) | wmask[2][nt] u64 (per wave: lanes hit, per layer of a run;
//   two buffers, taken in turn by the runs)
// Predecessor rows are kept only for layers that looked at other lanes (the others keep every
// lane), handed out in the order the layers are relaxed; a.idx_rows of them fit.
// SHARED_ROW: one lateral row for all layers (l_rstride == 0); MAPS: cost map / index map requested
// COUNT: measurement build that also adds the number of edges it priced to *a.n_evaluated
// RETRY: the second launch of the compact form (plans only the scenarios the first left open; an
// instance of its own so that a profile tells the two launches apart)
template <bool SHARED_ROW, bool MAPS, bool COUNT = false, bool RETRY = false>
__global__ void sl_plan_kernel(SlDev a)
{
    int n_eval = 0, n_eval_seen = 0, n_visit = 0, n_test = 0; // COUNT build only
    if (!MAPS) a.cost_map = nullptr, a.index_map = nullptr;
    extern __shared__ double smem[];
    const int b = a.order ? a.order[blockIdx.x] : blockIdx.x;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int n_l = a.in.n_l, n_sm = a.in.n_s; // n_sm: layer slots per scenario in every array
    const int n_s = scenario_layers(a.in, b);  // layers this scenario uses
    const int wave = tid / MP_WAVE, nw = nt / MP_WAVE;
    const int n_lp = (n_l + 1) & ~1; // keeps the double2 arrays 16-byte aligned
    const int cap = a.pair_cap;
    double2 *lc_all = (double2 *)smem;
    double *layA = (double *)(lc_all + 2 * n_lp);
    double *layQ = layA + n_sm;
    double *layW = layQ + n_sm;
    double *layP2 = layW + n_sm;
    double *layMid = layP2 + n_sm;
    double *pair_h0 = layMid + n_sm, *pair_ol = pair_h0 + cap, *pair_ds2 = pair_ol + cap;
    double *red = pair_ds2 + cap;
    double *redmax = red + 16;
    int *layK = (int *)(redmax + 8);
    int *cntL = layK + n_sm;
    int *excL = cntL + n_sm;      // per layer: waves whose targets all exceed the abort threshold
    int *first_ex = excL + n_sm;  // first layer every wave reported (2 ints, keeps what follows aligned)
    unsigned *recL = (unsigned *)(first_ex + 2); // per layer: pairs << 16 | first pair, ~0u = not in the store
    unsigned short *offL = (unsigned short *)(recL + n_sm);
    unsigned char *laySorted = (unsigned char *)(offL + n_sm + (n_sm & 1));
    unsigned char *locL = laySorted + ((n_sm + 3) & ~3); // 1: the layer is relaxed without looking at other lanes
    unsigned char *runL = locL + ((n_sm + 3) & ~3); // the layer's sample count if it and its successor need no exchange, else 0
    unsigned char *rowL = runL + ((n_sm + 3) & ~3);
    unsigned short *pair_lay = (unsigned short *)(rowL + ((n_sm + 3) & ~3));
    unsigned char *idx8 = (unsigned char *)(pair_lay + cap + (cap & 1));
    // (offsets from smem, so that the pointer stays an LDS pointer)
    double *lrow = (double *)((char *)smem + (((size_t)((char *)idx8 - (char *)smem) + (size_t)a.idx_rows * n_l + 7) & ~(size_t)7));
    unsigned long long *wmask = (unsigned long long *)(lrow + n_lp);

    if (RETRY && a.status[b] != SL_RETRY) return; // second launch: only what the first left open
    if (n_s <= 0) { // nothing to plan for this scenario
        write_skipped(a, b, n_sm, n_l);
        return;
    }
    const double *s_layers = a.in.s_layers + (size_t)b * n_sm;
    const double *lbase = a.in.l_map + (size_t)b * a.in.l_bstride;
    const int64_t rstride = SHARED_ROW ? 0 : a.in.l_rstride;
    const double *ob_s = a.in.ob_s + (size_t)b * a.in.ob_stride;
    const double *ob_l = a.in.ob_l + (size_t)b * a.in.ob_stride;
    const int n_ob = min(max(a.in.n_ob[b], 0), a.in.ob_stride); // never past the scenario's obstacle slots
    const double r = a.in.ob_dist[b], r2 = r * r;
    const double d_s = a.p.d_s, w_d = a.p.w_d;
    const double cG = w_d * d_s, cS = (1.0 - w_d) * d_s;
    const double pen = 3.0 * a.p.no_path_cost, abort_thr = 10.0 * a.p.no_path_cost;
    const int mid_l = a.p.mid_l;
    const double ego_s = a.in.ego_sl[2 * b], ego_l = a.in.ego_sl[2 * b + 1];
    // this thread's obstacle stays in registers for the whole plan
    const bool has_ob = tid < n_ob;
    const double my_os = has_ob ? ob_s[tid] : 0.0, my_ol = has_ob ? ob_l[tid] : 0.0;

    // ---- prologue: per-layer constants, one thread per layer ----------------
    int out_of_order = 0; // a layer that starts after it ends
    for (int s = tid; s < n_s; s += nt) {
        const double p1s = (s == 0) ? ego_s : s_layers[s - 1];
        const double p2s = s_layers[s];
        out_of_order |= !(p1s <= p2s);
        const LayerGeom g = layer_geom(p1s, p2s, d_s, n_ob);
        layK[s] = g.K < 0 ? g.K : (g.K | ((g.n != g.K) << 16));
        layP2[s] = p2s;
        layMid[s] = lbase[(size_t)s * rstride + mid_l];
        laySorted[s] = 1;
        locL[s] = 0;
        runL[s] = 0;
        rowL[s] = 0xff;
        cntL[s] = 0;
        excL[s] = 0;
        if (g.K >= 0) {
            const LayerConst c = layer_const(g, p1s, p2s, d_s, cG, cS);
            layA[s] = c.A;
            layQ[s] = c.Q;
            // the pruning bound needs Q/cSH moderate (rounding of the closed form
            // relative to cSH*dl^2); 0 switches it off for the layer
            layW[s] = (c.cSH > 0.0 && c.Q <= 1e5 * c.cSH) ? (1.0 + 1e-9) / (c.cSH * (1.0 - 1e-9)) : 0.0;
        }
    }
    if (!rstride)
        for (int i = tid; i < n_l; i += nt) lrow[i] = lbase[i]; // the lateral row every layer shares (for the runs below)
    wmask[tid] = 0ull; // both buffers of the runs' masks
    wmask[nt + tid] = 0ull;
    const bool layers_ascend = !__syncthreads_or(out_of_order);
    {
        // largest |l - mid| over the lattice (bounds every straight-edge cost)
        // and, per lattice row, whether it ascends
        double u_loc = 0.0, gap_loc = MP_INF; // gap_loc: smallest spacing of neighbouring lanes (row 0)
        const int rows = rstride ? n_s : 1;
        for (int i = tid; i < rows * n_l; i += nt) {
            const int row = i / n_l, col = i - row * n_l;
            const double *lr = lbase + (size_t)row * rstride;
            const double v = lr[col];
            const double u = fabs(v - lr[mid_l]);
            u_loc = u > u_loc ? u : u_loc;
            if (col > 0 && !(lr[col - 1] <= v)) laySorted[row] = 0;
            if (row == 0 && col > 0) {
                const double gp = v - lr[col - 1];
                gap_loc = gp < gap_loc ? gp : gap_loc;
            }
        }
        const double u_ego = fabs(ego_l - layMid[n_s - 1]);
        u_loc = u_ego > u_loc ? u_ego : u_loc;
        u_loc = -wave_min_f64(-u_loc);
        gap_loc = wave_min_f64(gap_loc);
        if ((tid & (MP_WAVE - 1)) == 0) {
            red[wave] = u_loc;
            redmax[wave] = gap_loc;
        }
        if (tid == 0) first_ex[0] = 0x7fffffff;
    }
    // ---- prologue: the (sample, obstacle) pairs of every layer --------------------
    PairStore ps;
    ps.n_s = n_s; ps.n_ob = n_ob; ps.cap = cap;
    ps.ascending = layers_ascend;
    ps.ob_s = ob_s; ps.ob_l = ob_l;
    ps.has_ob = has_ob; ps.my_os = my_os; ps.my_ol = my_ol;
    ps.ego_s = ego_s; ps.d_s = d_s; ps.r = r; ps.r2 = r2;
    ps.layP2 = layP2; ps.layK = layK; ps.cntL = cntL; ps.offL = offL; ps.recL = recL;
    ps.h0 = pair_h0; ps.ol = pair_ol; ps.ds2 = pair_ds2; ps.lay = pair_lay;
    ps.count();
    __syncthreads();
    if (tid == 0) lc_all[0] = make_double2(ego_l, 0.0); // layer 0: the ego is the only source
    ps.place(0);
    double u_max = red[0];
    for (int w = 1; w < nw; ++w) u_max = red[w] > u_max ? red[w] : u_max;
    const double g_max = uni_f64(cG * (double)MP_SL_K_CAP * u_max * u_max);
    double gap_min = redmax[0];
    for (int w = 1; w < nw; ++w) gap_min = redmax[w] < gap_min ? redmax[w] : gap_min;
    gap_min = uni_f64(gap_min);
    // Layers that need no exchange between lanes.  With one lateral row shared by all layers
    // (ascending, no two lanes equal) a target's own lane is its first candidate, a straight
    // edge that costs at most g_max + pen, so after layer s no cost exceeds
    // c_top(s) = c_top(s-1) + g_max + pen.  The walk of relax_target looks at a neighbour only
    // if its squared lateral distance is within (best - min + slack) * w_inv <= (c_top + slack)
    // * w_inv; once that is below the squared lane spacing no target can look at any other
    // lane: every thread then prices its own straight edge from registers, and the layer needs
    // neither the other lanes' nodes nor the layer minimum nor a barrier.  (The reference's
    // absolute-s evaluation makes w_inv fall like s^-8, so this holds from the first few
    // layers on.)  0 switches the shortcut off.
    // (cG >= 0: straight edges then cost at least nothing, so costs never decrease along a lane)
    const double gap2_min = (!rstride && n_l > 1 && laySorted[0] && gap_min > 0.0 && gap_min < MP_INF && cG >= 0.0)
                                ? gap_min * gap_min * (1.0 - 1e-9) : 0.0;
    __syncthreads(); // red / redmax are reused by the layer loop

    const bool is_target = tid < n_l;
    double l_cur = is_target ? lbase[tid] : 0.0; // row 0, this thread's node
    int status = MP_SL_FOUND;
    int done = 0;
    // Everything a layer needs that was fixed by the prologue is read one layer ahead,
    // before the barrier that ends the layer in front of it, so that the LDS round trips
    // overlap with that layer's reduction and barrier instead of starting the next one.
    struct LayerRegs {
        int kn, n_pairs, pair_off;
        bool sorted;
        double p2s, Q, w_inv, A, mid;
    } cur, nxt;
    // (every lane reads the same words: into scalar registers, see uni_f64)
    auto load_layer = [&](int s, LayerRegs &L) {
        L.kn = uni_i32(layK[s]);
        L.n_pairs = uni_i32(cntL[s]);
        L.pair_off = uni_i32(offL[s]);
        L.sorted = s == 0 || uni_i32(laySorted[rstride ? s - 1 : 0]) != 0;
        L.p2s = uni_f64(layP2[s]);
        L.Q = uni_f64(layQ[s]);
        L.w_inv = uni_f64(layW[s]);
        L.A = uni_f64(layA[s]);
        L.mid = uni_f64(layMid[s == 0 ? n_s - 1 : s - 1]); // layer 0: the last row's centre, as index -1 picks it
    };
    load_layer(0, cur);
    double p1s = ego_s;
    int g_K = -1;           // sample count and row centre g_own was computed for
    double g_mid = 0.0, g_own = 0.0;
    double c_top = MP_INF;  // upper bound of every accumulated cost of the layer being relaxed
    double cost_own = 0.0;  // this thread's node of the previous layer
    bool local = false;     // the layer being relaxed needs no exchange between lanes
    int run_par = 0;        // which of the two mask buffers the next run of layers without exchange takes
    int row_next = 0;       // predecessor rows handed out so far
    bool rows_short = false; // a layer that walked found no row left: the path cannot be traced back here
    auto is_local = [&](double top, double w_inv_l) {
        return gap2_min > 0.0 && w_inv_l > 0.0 && (top + 1e-12 * (g_max + top)) * w_inv_l < gap2_min;
    };
    for (int s = 0; s < n_s; ++s) {
        if (local) {
            // A run of layers that need nothing from the other lanes and whose successor
            // does not either: this lane's straight edge (cost g_own, the value the closed
            // form takes at dl = 0) against the layer's stored pairs, nothing else.  Layers
            // that need more (a pair store to lay out again, another sample count, the
            // layer in front of one that walks, the last layer) leave the run and take the
            // general path below.
            const int s_in = s;
            const int k_run = __builtin_amdgcn_readfirstlane(g_K);
            // the two sums a layer can add to this lane's cost, rounded as the general path
            // rounds them (edge first, then edge + cost)
            const double step_free = g_own + 0.0, step_hit = g_own + pen;
            // Costs never decrease along a lane, so "every cost of the layer exceeds the abort
            // threshold" holds from some layer of the run on; each lane counts the layers at
            // which its own cost was still within the threshold, and the wave looks at the
            // counts once, behind the run.
            int n_within = 0;
            // How far the run goes (at most 64 layers a time): lane i looks at layer s + i and
            // keeps that layer's record (pairs << 16 | first pair), so that the loops below get
            // every layer's record with a lane read instead of an LDS round trip.
            int n_run;
            unsigned rec_q = ~0u;
            {
                const int q = s + (tid & (MP_WAVE - 1));
                int k_q = 0;
                if (q < n_s) {
                    rec_q = recL[q];
                    k_q = runL[q];
                }
                const unsigned long long ok = __ballot(k_q == k_run && k_q != 0 && rec_q != ~0u);
                n_run = __builtin_amdgcn_readfirstlane(ok == ~0ull ? MP_WAVE : __builtin_ctzll(~ok));
            }
            if (n_run > 0) {
                // What happens to a lane in these layers does not depend on the other lanes or
                // on the costs: its straight edge either passes an obstacle or not.  So the
                // pairs of ALL layers of the run are looked at first (they lie in the store
                // layer after layer), giving per layer the lanes that hit something; the costs
                // then follow from those masks, in registers.  pair_hit at dl = 0: the sample
                // sits on this lane.
                const unsigned rec_first = __builtin_amdgcn_readlane(rec_q, 0);
                const unsigned rec_last = __builtin_amdgcn_readlane(rec_q, n_run - 1);
                const int e_begin = (int)(rec_first & 0xffffu);
                const int e_end = (int)(rec_last & 0xffffu) + (int)(rec_last >> 16);
                // Which lanes a pair hits: the lanes whose offset lies close enough to the
                // obstacle's, and since the row ascends and every rounding step of the test
                // fma(d, d, ds2) < r2, d = l - ol, is monotone in |d|, these are consecutive
                // lanes around the obstacle.  So a lane takes a PAIR (not every lane every
                // pair): it finds the first lane at or beyond the obstacle, tests the two lanes
                // next to the obstacle (if neither is hit, none is), bisects for the first and
                // the last lane hit with that very test, and ORs the range into the mask of
                // the pair's layer: one 64-bit mask per layer of the run and wave.  The pairs
                // are spread over the lanes of the whole workgroup (a pair's range may reach
                // into several waves' lanes), and one barrier stands between the ORs and the
                // reads.  The masks are double-buffered by run: a wave that is a run ahead ORs
                // into the other buffer, and a wave zeroes its entries right after reading
                // them, which happens before anyone passes the next run's barrier.
                const int lane = tid & (MP_WAVE - 1), lane0 = tid - lane; // first lattice lane of this wave
                unsigned long long *wm_run = wmask + run_par * nt;
                run_par ^= 1;
                for (int e = e_begin + tid; e < e_end; e += nt) {
                    const double ol_e = pair_ol[e], q_e = pair_ds2[e];
                    const int li = (int)pair_lay[e] - s_in;
                    auto pred = [&](int i) {
                        const double d = lrow[i] - ol_e;
                        return fma(d, d, q_e) < r2;
                    };
                    int c0 = 0, c1 = n_l; // first lane whose offset is not below the obstacle's
                    while (c0 < c1) {
                        const int mid = (c0 + c1) >> 1;
                        if (lrow[mid] < ol_e) c0 = mid + 1; else c1 = mid;
                    }
                    int in = -1;
                    if (c0 < n_l && pred(c0)) in = c0;
                    else if (c0 > 0 && pred(c0 - 1)) in = c0 - 1;
                    if (in < 0) continue;
                    int a0 = 0, a1 = in, b0 = in, b1 = n_l - 1; // first hit in [a0, a1], last hit in [b0, b1]
                    while (a0 < a1 || b0 < b1) {
                        if (a0 < a1) {
                            const int mid = (a0 + a1) >> 1;
                            if (pred(mid)) a1 = mid; else a0 = mid + 1;
                        }
                        if (b0 < b1) {
                            const int mid = (b0 + b1 + 1) >> 1;
                            if (pred(mid)) b0 = mid; else b1 = mid - 1;
                        }
                    }
                    // lanes a0 .. b0 of the lattice, as bits of the 64 of every wave they reach
                    for (int w0 = a0 & ~(MP_WAVE - 1); w0 <= b0; w0 += MP_WAVE) {
                        const int lo = max(a0 - w0, 0), hi = min(b0 - w0, MP_WAVE - 1);
                        const unsigned long long m = (~0ull >> (MP_WAVE - 1 - (hi - lo))) << lo;
                        atomicOr(&wm_run[w0 + li], m);
                    }
                }
                __syncthreads();
                // lane i now reads the mask of layer s + i; the loop below fetches it lane by lane
                const unsigned long long wm_i = wm_run[lane0 + lane];
                wm_run[lane0 + lane] = 0ull;
                const unsigned wm_lo = (unsigned)wm_i, wm_hi = (unsigned)(wm_i >> 32);
                const unsigned long long my_bit = 1ull << lane;
                for (int i = 0; i < n_run; ++i) {
                    // (the lane read returns a signed int: through unsigned, or bit 31 would spread)
                    const unsigned m_lo = (unsigned)__builtin_amdgcn_readlane((int)wm_lo, i);
                    const unsigned m_hi = (unsigned)__builtin_amdgcn_readlane((int)wm_hi, i);
                    const bool hit = (((unsigned long long)m_lo | ((unsigned long long)m_hi << 32)) & my_bit) != 0ull;
                    cost_own = (hit ? step_hit : step_free) + cost_own;
                    n_within += !(cost_own > abort_thr);
                    if (MAPS && is_target) {
                        const size_t o = ((size_t)b * n_sm + s) * n_l + tid;
                        if (a.cost_map) a.cost_map[o] = cost_own;
                        if (a.index_map) a.index_map[o] = tid;
                    }
                    if constexpr (COUNT) {
                        if (is_target) ++n_eval;
                        n_eval_seen = n_eval;
                        if ((tid & (MP_WAVE - 1)) == 0 && a.n_evaluated) atomicAdd(a.n_evaluated + 1, 1ull);
                    }
                    ++s;
                }
            }
            if (s != s_in) {
                // the run's first layer at which every target of this wave was above the threshold
                int last_within = is_target ? n_within : 0;
                for (int m = 32; m >= 1; m >>= 1) last_within = max(last_within, __shfl_xor(last_within, m, MP_WAVE));
                // (lane i reports for layer s_in + i: the run holds at most 64 layers)
                const int q = s_in + (int)(tid & (MP_WAVE - 1));
                if (q >= s_in + last_within && q < s)
                    if (atomicAdd(&excL[q], 1) == nw - 1) atomicMin(first_ex, q);
            }
            if (s != s_in) { // back to the general path at layer s (still one that needs no exchange)
                done = s;
                load_layer(s, cur);
                p1s = layP2[s - 1];
            }
        }
        const int buf = s & 1;
        double m_prev = 0.0; // minimum accumulated cost over the previous layer
        if (s == 1) {
            // largest cost of layer 0, from which the bound c_top grows
            double top0 = redmax[0];
            for (int w = 1; w < nw; ++w) top0 = redmax[w] > top0 ? redmax[w] : top0;
            c_top = uni_f64(top0 < MP_INF ? (top0 + g_max + pen) * (1.0 + 1e-12) : MP_INF);
            local = false; // layer 1 still takes the walk (its minimum was reduced with layer 0's)
            // which of the later layers need no exchange: the bound grows by the same amount
            // per layer, so every layer's can be had now (read from layer 2's end on, behind
            // the barrier that ends this layer)
            for (int q = tid; q < n_s; q += nt) {
                // at least what q - 1 steps of c -> (c + g_max + pen) * (1 + 1e-12) come to
                const double c = q < 2 ? c_top : (c_top + (double)(q - 1) * (g_max + pen)) * (1.0 + 2e-12 * (double)q);
                const bool loc = q >= 2 && is_local(c, layW[q]);
                locL[q] = loc;
                const int kq = layK[q];
                const bool loc_next = q + 1 < n_s && q >= 1 &&
                                      is_local((c + g_max + pen) * (1.0 + 1e-12), layW[q + 1]);
                runL[q] = (loc && loc_next && kq >= 0) ? (unsigned char)(kq & 0xffff) : 0;
            }
        }
        if (s > 0 && !local) {
            m_prev = red[buf * 8];
            for (int w = 1; w < nw; ++w) m_prev = red[buf * 8 + w] < m_prev ? red[buf * 8 + w] : m_prev;
            m_prev = uni_f64(m_prev);
            if (s >= 2 && m_prev > abort_thr) {
                if (status == MP_SL_FOUND) status = MP_SL_NOT_FOUND;
                if (a.p.early_exit) break;
            }
        }
        LayerGeom geo;
        geo.K = cur.kn < 0 ? cur.kn : (cur.kn & 0xffff);
        geo.n = cur.kn < 0 ? 0 : geo.K - (cur.kn >> 16);
        geo.delta = (p1s + d_s) - p1s;
        if (geo.K < 0) { status = geo.K; break; }
        const int n_src = (s == 0) ? 1 : n_l;
        int ent_off;
        bool ent_stored;
        const int n_ent = ps.layer_pairs(s, cur.n_pairs, cur.pair_off, &ent_off, &ent_stored);
        // next row of the lateral grid, in flight while this layer is relaxed
        double l_next = l_cur;
        if (rstride && is_target && s + 1 < n_s) l_next = lbase[(size_t)(s + 1) * rstride + tid];
        const double p2s = cur.p2s;
        const double Q = cur.Q, w_inv = cur.w_inv;
        const bool sorted = cur.sorted;
        SrcSh src;
        src.lc = lc_all + buf * n_lp;
        src.mid = cur.mid;
        src.two_cga = 2.0 * cG * cur.A;
        src.K = geo.K;
        src.w_d = w_d;
        src.d_s = d_s;
        const double *e_h0 = pair_h0 + ent_off, *e_ol = pair_ol + ent_off, *e_ds2 = pair_ds2 + ent_off;
        double best = MP_INF;
        int bj = 0;
        if (is_target && local) {
            // own lane only; the same arithmetic as relax_target's first candidate
            const double u = l_cur - src.mid;
            const double dl = l_cur - l_cur;
            if (geo.K != g_K || src.mid != g_mid) { // the same for all layers of a regular lattice
                g_K = geo.K;
                g_mid = src.mid;
                g_own = guide_straight(u, src.K, src.w_d, src.d_s);
            }
            double e = edge_closed_form(dl, g_own, src.two_cga * u, Q);
            bool hit;
            if (ent_stored) {
                Collider<-1> col;
                col.init(e_h0, e_ol, e_ds2, n_ent, geo, p1s, p2s, d_s, ob_s, ob_l, n_ob, r2);
                hit = col.hit(l_cur, dl);
            } else {
                Collider<-2> col;
                col.init(e_h0, e_ol, e_ds2, n_ent, geo, p1s, p2s, d_s, ob_s, ob_l, n_ob, r2);
                hit = col.hit(l_cur, dl);
            }
            e += hit ? pen : 0.0;
            best = e + cost_own;
            bj = tid;
            if constexpr (COUNT) ++n_eval;
        } else if (is_target) {
#define SL_RELAX(NE)                                                                                     \
    {                                                                                                    \
        Collider<NE> col;                                                                                \
        col.init(e_h0, e_ol, e_ds2, n_ent, geo, p1s, p2s, d_s, ob_s, ob_l, n_ob, r2);                    \
        best = relax_target<NE, false, COUNT>(n_src, src, l_cur, Q, w_inv, m_prev, g_max, sorted, pen, col, &bj, \
                                              &n_eval, &n_visit, &n_test);                               \
    }
            switch (ent_stored ? n_ent : -1) {
            case 0: SL_RELAX(0) break;
            case 1: SL_RELAX(1) break;
            case 2: SL_RELAX(2) break;
            case 3: SL_RELAX(3) break;
            case 4: SL_RELAX(4) break;
            case 5:
            case 6: SL_RELAX(6) break;
            case -1: SL_RELAX(-2) break; // the pair store overflowed before this layer
            default: // more than 6 pairs: the list is read from LDS
                if (n_ent > 24) {
                    Collider<-1> col;
                    col.init(e_h0, e_ol, e_ds2, n_ent, geo, p1s, p2s, d_s, ob_s, ob_l, n_ob, r2);
                    best = relax_target<-1, true, COUNT>(n_src, src, l_cur, Q, w_inv, m_prev, g_max, sorted, pen, col,
                                                         &bj, &n_eval, &n_visit, &n_test);
                } else {
                    SL_RELAX(-1)
                }
                break;
            }
#undef SL_RELAX
        }
        if constexpr (COUNT) {
            // second counter: edges priced one after the other, i.e. the busiest lane of
            // every wave, layer by layer (what the wave's time follows)
            int c = n_eval - n_eval_seen, cv = n_visit, ct = n_test;
            n_eval_seen = n_eval;
            n_visit = n_test = 0;
            for (int m = 32; m >= 1; m >>= 1) {
                c = max(c, __shfl_xor(c, m, MP_WAVE));
                cv = max(cv, __shfl_xor(cv, m, MP_WAVE));
                ct = max(ct, __shfl_xor(ct, m, MP_WAVE));
            }
            if ((tid & (MP_WAVE - 1)) == 0 && a.n_evaluated) {
                atomicAdd(a.n_evaluated + 1, (unsigned long long)c);
                atomicAdd(a.n_evaluated + 2, (unsigned long long)cv); // turns of the walk, busiest lane
                atomicAdd(a.n_evaluated + 3, (unsigned long long)ct); // collision tests in them, busiest lane
            }
        }
        done = s + 1;
        if (s + 1 < n_s) load_layer(s + 1, nxt); // a refill of the pair store re-reads its two fields
        if (s == 1) c_top = (c_top + g_max + pen) * (1.0 + 1e-12); // bound of layer 2
        const bool next_local = s >= 1 && s + 1 < n_s && (s == 1 ? is_local(c_top, nxt.w_inv) : locL[s + 1] != 0);
        // a layer that looked at other lanes keeps its predecessors (layer 0 has none: the ego);
        // whether it did is this very flag, so the trace back and the store cannot disagree
        const bool wants_row = s >= 1 && !local;
        const bool store_row = wants_row && row_next < a.idx_rows;
        rows_short |= wants_row && !store_row;
        if (store_row && tid == 0) rowL[s] = (unsigned char)row_next;
        if (is_target) {
            if (store_row) idx8[(size_t)row_next * n_l + tid] = (unsigned char)bj;
            const size_t o = ((size_t)b * n_sm + s) * n_l + tid;
            if (a.cost_map) a.cost_map[o] = best;
            if (a.index_map) a.index_map[o] = (s == 0) ? -1 : bj;
            // this node as a source of the next layer, unless that layer reads nothing from
            // the other lanes (after the last layer the row is read by the end-node search)
            if (!next_local) lc_all[(buf ^ 1) * n_lp + tid] = make_double2(l_cur, best);
            cost_own = best;
        }
        row_next += store_row;
        if (next_local) {
            // no minimum and no barrier: the abort test of this layer (every cost above the
            // threshold) is kept as a count of the waves for which it holds
            if (s >= 1) {
                const bool all_above = __all(!is_target || best > abort_thr);
                if (all_above && (tid & (MP_WAVE - 1)) == 0) {
                    if (atomicAdd(&excL[s], 1) == nw - 1) atomicMin(first_ex, s);
                }
            }
            // a layer that did walk over the other lanes' nodes must be finished by every
            // wave before the next one overwrites them
            if (!local) __syncthreads();
        } else {
            if (s == 0) {
                const double top = -wave_min_f64(is_target ? -best : MP_INF);
                if ((tid & (MP_WAVE - 1)) == 0) redmax[wave] = top;
            }
            best = wave_min_f64(best);
            if ((tid & (MP_WAVE - 1)) == 0) red[(buf ^ 1) * 8 + wave] = best;
            __syncthreads();
        }
        local = next_local;
        l_cur = l_next;
        p1s = p2s;
        cur = nxt;
    }
    if constexpr (COUNT) {
        // one add per wave
        int c = n_eval;
        for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, MP_WAVE);
        if ((tid & (MP_WAVE - 1)) == 0 && a.n_evaluated) atomicAdd(a.n_evaluated, (unsigned long long)c);
    }
    if (done == n_s && n_s >= 2 && status == MP_SL_FOUND) {
        // the last layer's abort test
        const int buf = n_s & 1;
        double m = red[buf * 8];
        for (int w = 1; w < nw; ++w) m = red[buf * 8 + w] < m ? red[buf * 8 + w] : m;
        if (m > abort_thr) status = MP_SL_NOT_FOUND;
    }
    __syncthreads();
    {
        // layers that ran without a barrier: the first one whose costs all exceeded the
        // threshold is where the reference gives up
        const int s_ex = first_ex[0];
        if (s_ex != 0x7fffffff) {
            if (status >= MP_SL_NOT_FOUND) {
                status = MP_SL_NOT_FOUND;
                if (a.p.early_exit && s_ex + 1 < done) done = s_ex + 1;
            } else if (a.p.early_exit && s_ex < done) {
                // a later layer stopped the loop with an error code (layer `done` was never
                // relaxed), but the reference had returned "Path not found!" at s_ex before
                // it ever looked at that layer (path_planner.py:74-77)
                status = MP_SL_NOT_FOUND;
                done = s_ex + 1;
            }
        }
    }
    // a plan whose path cannot be traced back with the rows at hand is left to the second launch
    if (status == MP_SL_FOUND && rows_short) status = SL_RETRY;
    // rows that were never relaxed keep the reference's initial values
    for (int i = done * n_l + tid; i < n_sm * n_l; i += nt) {
        size_t o = (size_t)b * n_sm * n_l + i;
        if (a.cost_map) a.cost_map[o] = 0.0;
        if (a.index_map) a.index_map[o] = -2;
    }
    if (tid < MP_WAVE) {
        // end node: first argmin of the last row, by one wave
        const double2 *last = lc_all + (done & 1) * n_lp;
        double c = MP_INF;
        int i = 0x7fffffff;
        if (status == MP_SL_FOUND) {
            for (int l = tid; l < n_l; l += MP_WAVE)
                if (last[l].y < c) { c = last[l].y; i = l; }
            group_argmin<MP_WAVE>(c, i);
        }
        int32_t *pidx = a.path_idx + (size_t)b * n_sm;
        if (status == MP_SL_FOUND) {
            if (tid == 0) {
                int end_l = i;
                pidx[n_s - 1] = end_l;
                for (int q = n_s - 1; q >= 1; --q) {
                    const int r = rowL[q]; // 0xff: the layer kept every lane
                    if (r != 0xff) end_l = idx8[(size_t)r * n_l + end_l];
                    pidx[q - 1] = end_l;
                }
            }
            for (int q = n_s + tid; q < n_sm; q += MP_WAVE) pidx[q] = -1;
        } else {
            for (int q = tid; q < n_sm; q += MP_WAVE) pidx[q] = -1;
        }
        if (tid == 0) {
            a.status[b] = status;
            a.layers_done[b] = done;
            if (a.best_cost) a.best_cost[b] = (status == MP_SL_FOUND) ? c : __builtin_nan("");
        }
    }
}

// ---------------------------------------------------------------------------
// Order of the workgroups.  A plan's time is mostly decided by its first layers: where an
// obstacle blocks a stretch of a target's sources the walk prices candidates one after the
// other, and such a scenario takes up to three times as long as the median.  Workgroups are
// handed to the CUs in launch order, so a batch of a few rounds of workgroups ends when the
// last long plan does; with the long ones launched first the short ones fill in behind them
// (on the benchmark batch 0.24 -> 0.18 ms).  The weight is a guess and only decides the
// order: every scenario is planned by the same code whatever its position.
//
// sl_weight_kernel: one wave per scenario, lanes over the obstacles.  For the pairs of the
// first three walking layers, the number of source lanes the pair blocks for a target: the
// sample sits at l1 + h0*(l2 - l1), so the blocked sources span 2*rho / |1 - h0| (rho: what
// is left of the radius at the pair's arc-length gap).  The sum goes into one of 64 bins
// (quarter octaves).
__global__ void sl_weight_kernel(SlDev a, int32_t *key)
{
    const int lane = threadIdx.x & (MP_WAVE - 1);
    const int b = blockIdx.x * (blockDim.x / MP_WAVE) + threadIdx.x / MP_WAVE;
    if (b >= a.in.B) return;
    const int n_l = a.in.n_l, n_sm = a.in.n_s;
    const int n_s = scenario_layers(a.in, b);
    const double *s_layers = a.in.s_layers + (size_t)b * n_sm;
    const double *lbase = a.in.l_map + (size_t)b * a.in.l_bstride;
    const double *ob_s = a.in.ob_s + (size_t)b * a.in.ob_stride;
    const double *ob_l = a.in.ob_l + (size_t)b * a.in.ob_stride;
    const int n_ob = min(max(a.in.n_ob[b], 0), a.in.ob_stride);
    const double r = a.in.ob_dist[b], r2 = r * r, d_s = a.p.d_s;
    double w = 0.0;
    if (n_l > 1 && n_ob > 0 && r > 0.0) {
        const double l_a = lbase[0], l_b = lbase[n_l - 1];
        const double l_lo = fmin(l_a, l_b), l_hi = fmax(l_a, l_b);
        const double gap = (l_hi - l_lo) / (double)(n_l - 1);
        const double lanes_per_m = gap > 0.0 ? 1.0 / gap : 0.0;
        const int s_end = n_s < 3 ? n_s : 3; // layers 1 and 2: their windows are the widest
        // (single precision is plenty for a guess)
        for (int s = 1; s < s_end && gap > 0.0; ++s) {
            const double p1s = s_layers[s - 1], p2s = s_layers[s];
            const double rr = r * 1.000001 + 1e-9;
            if (__ballot(lane < n_ob) == 0ull) break;
            const LayerGeom g = layer_geom(p1s, p2s, d_s, n_ob);
            if (g.K < 0) continue;
            const float inv_T = 1.0f / (float)(p2s - p1s);
            for (int o = lane; o < n_ob; o += MP_WAVE) {
                const double os = ob_s[o], ol = ob_l[o];
                if (os < p1s - rr || os > p2s + rr || !(ol > l_lo - r - 1.0 && ol < l_hi + r + 1.0)) continue;
                for_each_near_sample(g, p1s, p2s, d_s, os, r, r2, [&](int, double x, double ds2) {
                    const float tau = (float)x * inv_T;
                    const float h0 = tau * tau * tau * (10.0f + tau * (-15.0f + 6.0f * tau));
                    const float am = fabsf(1.0f - h0);
                    const float span = 2.0f * sqrtf((float)(r2 - ds2)) * (float)lanes_per_m;
                    // a sample that sits at the target's own offset blocks the target for every
                    // source; the walk treats such targets cheaply, so they count little
                    const float lanes = am > 1e-4f ? span / am : 0.2f * span;
                    w += (double)fminf(lanes, (float)n_l);
                });
            }
        }
    }
    for (int m = 32; m >= 1; m >>= 1) w += __shfl_xor(w, m, MP_WAVE);
    if (lane == 0) key[b] = mp_weight_bin(w);
}

// The same for lattices of up to 16 lanes, whose kernel prices every edge: a scenario's time
// follows its (sample, obstacle) pairs, i.e. obstacles x layers.  One thread per scenario.
__global__ void sl_weight_narrow_kernel(SlDev a, int32_t *key)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.in.B) return;
    const int n_ob = min(max(a.in.n_ob[b], 0), a.in.ob_stride);
    key[b] = mp_weight_bin((double)(n_ob + 1) * (double)scenario_layers(a.in, b));
}

// mp_order_kernel: one workgroup; a counting sort of the scenarios by bin (0..63), heaviest
// bin first (the order inside a bin is whatever the atomics give).
__global__ void mp_order_kernel(int B, const int32_t *key, int32_t *order)
{
    __shared__ int cnt[MP_WEIGHT_BINS], base[MP_WEIGHT_BINS];
    const int tid = threadIdx.x, nt = blockDim.x;
    if (tid < MP_WEIGHT_BINS) cnt[tid] = 0;
    __syncthreads();
    for (int b = tid; b < B; b += nt) atomicAdd(&cnt[key[b] & (MP_WEIGHT_BINS - 1)], 1);
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int k = MP_WEIGHT_BINS - 1; k >= 0; --k) {
            base[k] = run;
            run += cnt[k];
            cnt[k] = 0;
        }
    }
    __syncthreads();
    for (int b = tid; b < B; b += nt) {
        const int k = key[b] & (MP_WEIGHT_BINS - 1);
        order[base[k] + atomicAdd(&cnt[k], 1)] = b;
    }
}

// order[0..B) = the scenarios by descending bin key[0..B) (for the other kernels of the library)
void mp_order_launch(int B, const int32_t *key, int32_t *order, hipStream_t st)
{
    hipLaunchKernelGGL(mp_order_kernel, dim3(1), dim3(1024), 0, st, B, key, order);
}

// ---------------------------------------------------------------------------
// Narrow lattices (n_l <= 16, the reference's default is 5 lanes): one lane per
// EDGE instead of one thread per node.  Lane (t, j) = (tid / W, tid % W), W the
// power of two >= n_l, holds the edge from source j to target t; the targets'
// first minima come from a W-lane (cost, index) merge.  Every edge of a layer is
// evaluated, so no pruning is needed and all lanes are busy.  The (sample,
// obstacle) pairs of all layers are laid out in LDS by the prologue (PairStore,
// in parallel over the obstacles; the store is laid out again from the current
// layer when a crowded scene fills it), so the collision test of a layer is one
// loop over that layer's pairs, the same for every lane.  Same edge arithmetic as
// sl_plan_kernel, bit for bit.
// Dynamic LDS (n_sm = layer slots, cap = pair_cap):
//   layA layQ layP2 layMid [n_sm] | pairs[3][cap] | lc[2][W] double2 | g[2][W] | red[2][16]
//   | hit_w[16] u64 | layK[n_sm] int | cntL[n_sm] int | offL[n_sm] u16 | idx8[n_sm*n_l] u8
template <int W, bool MAPS>
__global__ void sl_plan_narrow_kernel(SlDev a)
{
    if (!MAPS) a.cost_map = nullptr, a.index_map = nullptr;
    extern __shared__ double smem[];
    const int b = a.order ? a.order[blockIdx.x] : blockIdx.x;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & (MP_WAVE - 1), wave = tid / MP_WAVE, nw = nt / MP_WAVE;
    const int n_l = a.in.n_l, n_sm = a.in.n_s;
    const int n_s = scenario_layers(a.in, b);
    double *layA = smem;
    double *layQ = layA + n_sm;
    double *layP2 = layQ + n_sm;
    double *layMid = layP2 + n_sm;
    const int cap = a.pair_cap; // even
    double *pair_h0 = layMid + n_sm + (n_sm & 1); // even offset: the double2 array below stays aligned
    double *pair_ol = pair_h0 + cap, *pair_ds2 = pair_ol + cap;
    double2 *lc_all = (double2 *)(pair_ds2 + cap + (cap & 1));
    double *g_all = (double *)(lc_all + 2 * W);
    double *red = g_all + 2 * W;
    unsigned long long *hit_w = (unsigned long long *)(red + 32); // [16]: per wave, which edges its share of the pairs hits
    int *layK = (int *)(hit_w + 16);
    int *cntL = layK + n_sm;
    unsigned short *offL = (unsigned short *)(cntL + n_sm);
    unsigned char *idx8 = (unsigned char *)(offL + n_sm + (n_sm & 1));

    if (n_s <= 0) {
        write_skipped(a, b, n_sm, n_l);
        return;
    }
    const double *s_layers = a.in.s_layers + (size_t)b * n_sm;
    const double *lbase = a.in.l_map + (size_t)b * a.in.l_bstride;
    const int64_t rstride = a.in.l_rstride;
    const double *ob_s = a.in.ob_s + (size_t)b * a.in.ob_stride;
    const double *ob_l = a.in.ob_l + (size_t)b * a.in.ob_stride;
    const int n_ob = min(max(a.in.n_ob[b], 0), a.in.ob_stride); // never past the scenario's obstacle slots
    const double r = a.in.ob_dist[b], r2 = r * r;
    const double d_s = a.p.d_s, w_d = a.p.w_d;
    const double cG = w_d * d_s, cS = (1.0 - w_d) * d_s;
    const double pen = 3.0 * a.p.no_path_cost, abort_thr = 10.0 * a.p.no_path_cost;
    const int mid_l = a.p.mid_l;
    const double ego_s = a.in.ego_sl[2 * b], ego_l = a.in.ego_sl[2 * b + 1];
    const int t = tid / W, j = tid - t * W;

    int out_of_order = 0; // a layer that starts after it ends
    for (int s = tid; s < n_s; s += nt) {
        const double p1s = (s == 0) ? ego_s : s_layers[s - 1];
        const double p2s = s_layers[s];
        out_of_order |= !(p1s <= p2s);
        const LayerGeom g = layer_geom(p1s, p2s, d_s, n_ob);
        layK[s] = g.K < 0 ? g.K : (g.K | ((g.n != g.K) << 16));
        layP2[s] = p2s;
        layMid[s] = lbase[(size_t)s * rstride + mid_l];
        cntL[s] = 0;
        if (g.K >= 0) {
            const LayerConst c = layer_const(g, p1s, p2s, d_s, cG, cS);
            layA[s] = c.A;
            layQ[s] = c.Q;
        }
    }
    const bool layers_ascend = !__syncthreads_or(out_of_order);
    PairStore ps;
    ps.n_s = n_s; ps.n_ob = n_ob; ps.cap = cap;
    ps.ascending = layers_ascend;
    ps.ob_s = ob_s; ps.ob_l = ob_l;
    ps.has_ob = tid < n_ob;
    ps.my_os = ps.has_ob ? ob_s[tid] : 0.0;
    ps.my_ol = ps.has_ob ? ob_l[tid] : 0.0;
    ps.ego_s = ego_s; ps.d_s = d_s; ps.r = r; ps.r2 = r2;
    ps.layP2 = layP2; ps.layK = layK; ps.cntL = cntL; ps.offL = offL; ps.recL = nullptr;
    ps.h0 = pair_h0; ps.ol = pair_ol; ps.ds2 = pair_ds2; ps.lay = nullptr;
    ps.count();
    __syncthreads();
    auto layer_geo = [&](int s) {
        LayerGeom g;
        const int kn = layK[s];
        g.K = kn < 0 ? kn : (kn & 0xffff);
        g.n = kn < 0 ? 0 : g.K - (kn >> 16);
        const double p1s = (s == 0) ? ego_s : layP2[s - 1];
        g.delta = (p1s + d_s) - p1s;
        return g;
    };
    {
        const LayerGeom g0 = layer_geo(0);
        if (g0.K >= 0 && tid == 0) {
            lc_all[0] = make_double2(ego_l, 0.0);
            g_all[0] = guide_straight(ego_l - layMid[n_s - 1], g0.K, w_d, d_s);
        }
    }
    ps.place(0); // ends with a barrier

    int status = MP_SL_FOUND, done = 0;
    for (int s = 0; s < n_s; ++s) {
        const int buf = s & 1;
        if (s > 0) {
            double m_prev = red[buf * 16];
            for (int w = 1; w < nw; ++w) m_prev = red[buf * 16 + w] < m_prev ? red[buf * 16 + w] : m_prev;
            if (s >= 2 && m_prev > abort_thr) {
                if (status == MP_SL_FOUND) status = MP_SL_NOT_FOUND;
                if (a.p.early_exit) break;
            }
        }
        const LayerGeom geo = layer_geo(s);
        if (geo.K < 0) { status = geo.K; break; }
        const int K = geo.K;
        const int n_src = (s == 0) ? 1 : n_l;
        const bool valid = t < n_l && j < n_src;
        const double p1s = (s == 0) ? ego_s : layP2[s - 1];
        const double p2s = layP2[s];
        const double l2v = t < n_l ? lbase[(size_t)s * rstride + t] : 0.0;
        const double2 lcj = valid ? lc_all[buf * W + j] : make_double2(0.0, 0.0);
        const double gj = valid ? g_all[buf * W + j] : 0.0;
        const double dl = l2v - lcj.x;
        int ent_off;
        bool ent_stored;
        const int n_ent = ps.layer_pairs(s, &ent_off, &ent_stored);
        bool hit = false;
        if (ent_stored && n_l * n_l <= 32) {
            // Up to 5 lanes: the edges of a layer fill at most half of the (single) wave, so
            // the wave holds every edge 64/Es times (Es = 4, 16 or 32 lanes per copy) and each
            // copy takes every (64/Es)-th pair; a ballot brings the copies' verdicts together.
            // A crowded scene runs several waves per scene (they share the work of laying out
            // the pairs); every wave then takes its share of the layer's pairs as well and the
            // verdicts meet in LDS.
            const double *e_h0 = pair_h0 + ent_off, *e_ol = pair_ol + ent_off, *e_ds2 = pair_ds2 + ent_off;
            const int n_e = n_l * n_l, Es = n_e <= 4 ? 4 : (n_e <= 16 ? 16 : 32), P = MP_WAVE / Es;
            const int ee = lane & (Es - 1), part = wave * P + lane / Es;
            const int t2 = ee / n_l, j2 = ee - t2 * n_l;
            const bool e_valid = ee < n_e && j2 < n_src;
            // the edge's end points: the values the lanes of the (t, j) layout hold
            const double l2b = e_valid ? lbase[(size_t)s * rstride + t2] : 0.0;
            const double l1b = e_valid ? lc_all[buf * W + j2].x : 0.0;
            const double dlb = l2b - l1b;
            bool h = false;
            for (int e = part; e < n_ent; e += P * nw) h |= pair_hit(l1b, dlb, e_h0[e], e_ol[e], e_ds2[e], r2);
            unsigned long long m = __ballot(h && e_valid);
            for (int p = 1; p < P; ++p) m |= m >> (p * Es); // bit e of the low copy: any copy hit
            if (nw > 1) {
                if (lane == 0) hit_w[wave] = m;
                __syncthreads();
                m = hit_w[0];
                for (int w = 1; w < nw; ++w) m |= hit_w[w];
            }
            hit = valid && ((m >> (t * n_l + j)) & 1ull);
        } else if (ent_stored) {
            const double *e_h0 = pair_h0 + ent_off, *e_ol = pair_ol + ent_off, *e_ds2 = pair_ds2 + ent_off;
            for (int e = 0; e < n_ent; ++e) hit |= pair_hit(lcj.x, dl, e_h0[e], e_ol[e], e_ds2[e], r2);
        } else if (3 * cap >= 2 * K) {
            // This one layer has more pairs than the store holds (a very crowded scene).
            // The store is idle then (the layers after it are laid out later), so it takes
            // the layer's sample abscissae and basis values, and the obstacles are streamed:
            // each round the lanes look at 64 obstacles, a ballot keeps those within reach
            // of the layer's span, and for each of them (a wave-uniform loop) every lane
            // tests its own edge at the samples around the obstacle.
            double *x_tab = pair_h0, *h_tab = pair_h0 + K;
            __syncthreads();
            for (int k = tid; k < K; k += nt) {
                const double x = arange_value(k, geo.n, p1s, p2s, d_s, geo.delta);
                x_tab[k] = x;
                h_tab[k] = basis_h0(x, p2s - p1s);
            }
            __syncthreads();
            const double rr = r * 1.000001 + 1e-9; // span test with slack; the pair test itself is exact
            const double inv_ds = 1.0 / d_s, k_half = r * inv_ds + 0.75; // window slack: 3/4 of a spacing
            for (int o0 = 0; o0 < n_ob; o0 += MP_WAVE) {
                const int o = o0 + lane;
                double os = 0.0, ol = 0.0;
                bool near = false;
                if (o < n_ob) {
                    os = ob_s[o];
                    ol = ob_l[o];
                    near = !(os < p1s - rr || os > p2s + rr);
                }
                unsigned long long m = __ballot(near);
                while (m) {
                    const int sl = __builtin_ctzll(m);
                    m &= m - 1;
                    const double uos = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(os), sl),
                                                        __builtin_amdgcn_readlane(__double2loint(os), sl));
                    const double uol = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ol), sl),
                                                        __builtin_amdgcn_readlane(__double2loint(ol), sl));
                    // grid sample k sits at p1s + k*d_s up to rounding, so only |k - c| < r/d_s
                    // can pass; the appended end point is tested on its own
                    int k_lo = 0, k_hi = K - 1;
                    if (K > 8) {
                        const double c = (uos - p1s) * inv_ds;
                        k_lo = (c - k_half > 0.0) ? (int)(c - k_half) : 0;
                        k_hi = (c + k_half < (double)(K - 2)) ? (int)(c + k_half) : K - 2;
                        k_lo = __builtin_amdgcn_readfirstlane(k_lo);
                        k_hi = __builtin_amdgcn_readfirstlane(k_hi);
                    }
                    for (int k = k_lo; k <= k_hi; ++k) {
                        const double ds = x_tab[k] - uos;
                        const double ds2 = ds * ds;
                        if (ds2 < r2) hit |= pair_hit(lcj.x, dl, h_tab[k], uol, ds2, r2);
                    }
                    if (k_hi < K - 1) {
                        const double ds = x_tab[K - 1] - uos;
                        const double ds2 = ds * ds;
                        if (ds2 < r2) hit |= pair_hit(lcj.x, dl, h_tab[K - 1], uol, ds2, r2);
                    }
                }
            }
        } else { // a store too small even for the sample tables: test every pair directly
            Collider<-2> col;
            col.init(nullptr, nullptr, nullptr, 0, geo, p1s, p2s, d_s, ob_s, ob_l, n_ob, r2);
            hit = col.hit(lcj.x, dl);
        }

        const double mid = layMid[s == 0 ? n_s - 1 : s - 1];
        const double two_cga = 2.0 * cG * layA[s];
        double e = edge_closed_form(dl, gj, two_cga * (lcj.x - mid), layQ[s]);
        e += hit ? pen : 0.0;
        double c = valid ? e + lcj.y : MP_INF;
        int bj = valid ? j : 0x7fffffff;
        group_argmin<W>(c, bj);
        const bool owner = j == 0 && t < n_l;
        if (owner) {
            idx8[(size_t)s * n_l + t] = (unsigned char)bj;
            const size_t o = ((size_t)b * n_sm + s) * n_l + t;
            if (a.cost_map) a.cost_map[o] = c;
            if (a.index_map) a.index_map[o] = (s == 0) ? -1 : bj;
            lc_all[(buf ^ 1) * W + t] = make_double2(l2v, c);
        }
        done = s + 1;
        if (s + 1 < n_s) {
            const LayerGeom gn = layer_geo(s + 1);
            if (gn.K >= 0) {
                if (owner) g_all[(buf ^ 1) * W + t] = guide_straight(l2v - layMid[s], gn.K, w_d, d_s);
            }
        }
        const double bm = wave_min_f64(owner ? c : MP_INF);
        if (lane == 0) red[(buf ^ 1) * 16 + wave] = bm;
        __syncthreads();
    }
    if (done == n_s && n_s >= 2 && status == MP_SL_FOUND) {
        const int buf = n_s & 1;
        double m = red[buf * 16];
        for (int w = 1; w < nw; ++w) m = red[buf * 16 + w] < m ? red[buf * 16 + w] : m;
        if (m > abort_thr) status = MP_SL_NOT_FOUND;
    }
    __syncthreads();
    for (int i = done * n_l + tid; i < n_sm * n_l; i += nt) {
        const size_t o = (size_t)b * n_sm * n_l + i;
        if (a.cost_map) a.cost_map[o] = 0.0;
        if (a.index_map) a.index_map[o] = -2;
    }
    if (tid < MP_WAVE) {
        const double2 *last = lc_all + (done & 1) * W;
        double c = MP_INF;
        int i = 0x7fffffff;
        if (status == MP_SL_FOUND) {
            if (tid < n_l) { c = last[tid].y; i = tid; }
            group_argmin<MP_WAVE>(c, i);
        }
        int32_t *pidx = a.path_idx + (size_t)b * n_sm;
        if (status == MP_SL_FOUND) {
            if (tid == 0) {
                int end_l = i;
                pidx[n_s - 1] = end_l;
                for (int q = n_s - 1; q >= 1; --q) {
                    end_l = idx8[(size_t)q * n_l + end_l];
                    pidx[q - 1] = end_l;
                }
            }
            for (int q = n_s + tid; q < n_sm; q += MP_WAVE) pidx[q] = -1;
        } else {
            for (int q = tid; q < n_sm; q += MP_WAVE) pidx[q] = -1;
        }
        if (tid == 0) {
            a.status[b] = status;
            a.layers_done[b] = done;
            if (a.best_cost) a.best_cost[b] = (status == MP_SL_FOUND) ? c : __builtin_nan("");
        }
    }
}

// Edge tensor: every edge cost of one (scenario, layer) per workgroup, written
// with the source index fastest so that consecutive lanes store consecutive
// doubles.  Same arithmetic as the fused kernel, bit for bit.
__global__ void sl_edge_cost_kernel(SlDev a)
{
    __shared__ double e_h0[SL_E_CAP], e_ol[SL_E_CAP], e_ds2[SL_E_CAP];
    __shared__ LayerConst lay;
    __shared__ int n_ent_sh;
    const int s = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, nt = blockDim.x;
    const int n_l = a.in.n_l, n_sm = a.in.n_s;
    const int n_s = scenario_layers(a.in, b);
    double *out = a.edge_cost + ((size_t)b * n_sm + s) * n_l * n_l;
    if (s >= n_s) { // layer slot this scenario does not use
        for (int e = tid; e < n_l * n_l; e += nt) out[e] = __builtin_nan("");
        return;
    }
    const double *s_layers = a.in.s_layers + (size_t)b * n_sm;
    const double *lbase = a.in.l_map + (size_t)b * a.in.l_bstride;
    const double *ob_s = a.in.ob_s + (size_t)b * a.in.ob_stride;
    const double *ob_l = a.in.ob_l + (size_t)b * a.in.ob_stride;
    const int n_ob = min(max(a.in.n_ob[b], 0), a.in.ob_stride); // never past the scenario's obstacle slots
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
    const LayerGeom geo = layer_geom(p1s, p2s, d_s, n_ob);
    if (geo.K < 0) {
        if (tid == 0) atomicMin(&a.status[b], geo.K);
        for (int e = tid; e < n_l * n_l; e += nt) out[e] = __builtin_nan("");
        return;
    }
    if (tid == 0) {
        lay = layer_const(geo, p1s, p2s, d_s, cG, cS);
        n_ent_sh = 0;
    }
    __syncthreads();
    const double A = lay.A, Q = lay.Q;
    build_entries(&n_ent_sh, geo, p1s, p2s, d_s, e_h0, e_ol, e_ds2, ob_s, ob_l, n_ob, r, r2);
    __syncthreads();
    const int n_ent = n_ent_sh;
    Collider<-1> col_list;
    Collider<-2> col_all;
    col_list.init(e_h0, e_ol, e_ds2, n_ent, geo, p1s, p2s, d_s, ob_s, ob_l, n_ob, r2);
    col_all.init(e_h0, e_ol, e_ds2, n_ent, geo, p1s, p2s, d_s, ob_s, ob_l, n_ob, r2);
    if (n_src == n_l && (n_l & 1) == 0 && (2 * nt) % n_l == 0 && n_l <= 2 * nt) {
        // two neighbouring sources per thread, kept for all of its edges: what depends on the
        // source alone leaves the loop and a thread stores 16 bytes per target row
        const int hp = n_l / 2, jp = tid % hp, j = 2 * jp;
        const double l1a = lprev_g[j], l1b = lprev_g[j + 1];
        const double ua = l1a - mid, ub = l1b - mid;
        const double ga = guide_straight(ua, geo.K, w_d, d_s), gb = guide_straight(ub, geo.K, w_d, d_s);
        const double xa = 2.0 * cG * A * ua, xb = 2.0 * cG * A * ub;
        for (int l2 = tid / hp; l2 < n_l; l2 += nt / hp) {
            const double l2v = lcur_g[l2];
            const double dla = l2v - l1a, dlb = l2v - l1b;
            double ca = edge_closed_form(dla, ga, xa, Q), cb = edge_closed_form(dlb, gb, xb, Q);
            const bool ha = n_ent <= SL_E_CAP ? col_list.hit(l1a, dla) : col_all.hit(l1a, dla);
            const bool hb = n_ent <= SL_E_CAP ? col_list.hit(l1b, dlb) : col_all.hit(l1b, dlb);
            ca += ha ? pen : 0.0;
            cb += hb ? pen : 0.0;
            *(double2 *)(out + (size_t)l2 * n_l + j) = make_double2(ca, cb);
        }
        return;
    }
    if (n_src == n_l && nt % n_l == 0) {
        // the workgroup's size is a multiple of the row length: a thread keeps one source
        // for all of its edges, and what depends on the source alone leaves the loop
        const int j = tid % n_l;
        const double l1 = lprev_g[j];
        const double u = l1 - mid;
        const double g = guide_straight(u, geo.K, w_d, d_s), ua = 2.0 * cG * A * u;
        for (int l2 = tid / n_l; l2 < n_l; l2 += nt / n_l) {
            const double dl = lcur_g[l2] - l1;
            double c = edge_closed_form(dl, g, ua, Q);
            const bool hit = n_ent <= SL_E_CAP ? col_list.hit(l1, dl) : col_all.hit(l1, dl);
            c += hit ? pen : 0.0;
            out[(size_t)l2 * n_l + j] = c;
        }
        return;
    }
    for (int e = tid; e < n_l * n_l; e += nt) {
        const int l2 = e / n_l, j = e - l2 * n_l;
        if (j >= n_src) { out[e] = __builtin_nan(""); continue; }
        const double l1 = (s == 0) ? a.in.ego_sl[2 * b + 1] : lprev_g[j];
        const double u = l1 - mid;
        const double dl = lcur_g[l2] - l1;
        double c = edge_closed_form(dl, guide_straight(u, geo.K, w_d, d_s), 2.0 * cG * A * u, Q);
        const bool hit = n_ent <= SL_E_CAP ? col_list.hit(l1, dl) : col_all.hit(l1, dl);
        c += hit ? pen : 0.0;
        out[e] = c;
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

#define SL_DP_ROWS 8 // target rows whose loads a wave issues together

// DP relaxation over a materialised edge-cost tensor.  One workgroup per
// scenario; each wave owns target rows, its 64 lanes stride over the sources
// (coalesced reads) and the row's first-minimum is found with a wave-level
// shuffle argmin.  LDS: cost_prev[n_l] cost_cur[n_l] red[16].
__global__ void sl_dp_kernel(int n_sm, int n_l, const int32_t *n_layers, const double *edge_cost,
                             mp_sl_params p, double *cost_map, int32_t *index_map, int32_t *path_idx,
                             int32_t *status_out, int32_t *layers_done, double *best_cost)
{
    extern __shared__ double smem[];
    double *cost_prev = smem, *cost_cur = smem + n_l, *red = cost_cur + n_l;
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & (MP_WAVE - 1), wave = tid / MP_WAVE, nw = nt / MP_WAVE;
    const int n_s = n_layers ? min(max(n_layers[b], 0), n_sm) : n_sm;
    const double *ec = edge_cost + (size_t)b * n_sm * n_l * n_l;
    int32_t *imap = index_map + (size_t)b * n_sm * n_l;
    if (n_s <= 0) {
        for (int i = tid; i < n_sm * n_l; i += nt) {
            if (cost_map) cost_map[(size_t)b * n_sm * n_l + i] = 0.0;
            imap[i] = -2;
        }
        for (int q = tid; q < n_sm; q += nt) path_idx[(size_t)b * n_sm + q] = -1;
        if (tid == 0) {
            status_out[b] = MP_SL_SKIPPED;
            layers_done[b] = 0;
            if (best_cost) best_cost[b] = __builtin_nan("");
        }
        return;
    }
    const double abort_thr = 10.0 * p.no_path_cost;
    int status = MP_SL_FOUND, done = 0;
    for (int s = 0; s < n_s; ++s) {
        const double *el = ec + (size_t)s * n_l * n_l;
        if (s == 0) {
            for (int l = tid; l < n_l; l += nt) {
                cost_cur[l] = el[(size_t)l * n_l];
                imap[l] = -1;
            }
        } else if ((n_l & 1) == 0 && n_l <= 2 * MP_WAVE) {
            // Rows of up to 128 sources, 16-byte aligned: a lane holds the sources 2*lane and
            // 2*lane+1 of a row (one 16-byte load), and the loads of SL_DP_ROWS rows are issued
            // before the first of them is reduced, so that a wave has that many kilobytes in
            // flight instead of one row at a time.
            const int pj = 2 * lane;
            const bool has = pj < n_l;
            const double cp0 = has ? cost_prev[pj] : 0.0, cp1 = has ? cost_prev[pj + 1] : 0.0;
            for (int l0 = wave; l0 < n_l; l0 += nw * SL_DP_ROWS) {
                double2 e[SL_DP_ROWS];
#pragma unroll
                for (int q = 0; q < SL_DP_ROWS; ++q) {
                    const int l2 = l0 + q * nw;
                    e[q] = make_double2(MP_INF, MP_INF);
                    if (has && l2 < n_l) e[q] = *(const double2 *)(el + (size_t)l2 * n_l + pj);
                }
                double c[SL_DP_ROWS];
                int i[SL_DP_ROWS];
#pragma unroll
                for (int q = 0; q < SL_DP_ROWS; ++q) { // rows past the end hold +inf: harmless
                    const double v0 = e[q].x + cp0, v1 = e[q].y + cp1;
                    const bool second = v1 < v0;
                    c[q] = has ? (second ? v1 : v0) : MP_INF;
                    i[q] = has ? (second ? pj + 1 : pj) : 0x7fffffff;
                }
                double rc;
                int ri;
                wave_argmin_rows8(c, i, rc, ri);
                const int l2 = l0 + lane * nw; // lane q < 8 holds row q of the batch
                if (lane < SL_DP_ROWS && l2 < n_l) {
                    cost_cur[l2] = rc;
                    imap[(size_t)s * n_l + l2] = ri;
                }
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
            if (cost_map) cost_map[((size_t)b * n_sm + s) * n_l + l] = v;
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
    for (int i = done * n_l + tid; i < n_sm * n_l; i += nt) {
        if (cost_map) cost_map[(size_t)b * n_sm * n_l + i] = 0.0;
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
        int32_t *pidx = path_idx + (size_t)b * n_sm;
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
            for (int q = n_s + tid; q < n_sm; q += MP_WAVE) pidx[q] = -1;
        } else {
            for (int q = tid; q < n_sm; q += MP_WAVE) pidx[q] = -1;
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

// rows: predecessor rows the kernel keeps (one per layer that walks; all n_s in the full-table form)
static size_t sl_plan_lds_bytes(int n_s, int n_l, int cap, int rows)
{
    size_t d = (size_t)4 * ((n_l + 1) & ~1) + 5 * (size_t)n_s + 3 * (size_t)cap + 24;
    size_t bytes = d * sizeof(double) + (4 * (size_t)n_s + 2) * sizeof(int) + ((size_t)n_s + (n_s & 1)) * sizeof(short);
    bytes += 4 * (((size_t)n_s + 3) & ~(size_t)3) + (size_t)rows * n_l;
    bytes += 2 * ((size_t)cap + (cap & 1));                     // layer of every stored pair
    bytes = (bytes + 7) & ~(size_t)7;
    bytes += ((size_t)((n_l + 1) & ~1) + 2 * (size_t)((n_l + 63) / 64) * 64) * sizeof(double); // lateral row, run masks
    return (bytes + 15) & ~(size_t)15;
}

#define SLN_CAP_MAX 384 // narrow kernel: upper limit of its pair store per wave (9 KB)

static size_t sl_narrow_lds_bytes(int n_s, int n_l, int W, int cap)
{
    size_t d = (size_t)4 * n_s + (n_s & 1) + 3 * (size_t)cap + (cap & 1) + 4 * W + 2 * W + 32 + 16;
    size_t bytes = d * sizeof(double) + 2 * (size_t)n_s * sizeof(int) + ((size_t)n_s + (n_s & 1)) * sizeof(short) +
                   (size_t)n_s * n_l;
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

static int sl_plan_launch(const mp_sl_batch *batch, const mp_sl_params *params, double *cost_map,
                          int32_t *index_map, int32_t *path_idx, int32_t *status, int32_t *layers_done,
                          double *best_cost, unsigned long long *n_evaluated, int32_t *order_ws, void *stream,
                          bool keys_ready = false);

extern "C" int mp_sl_plan(const mp_sl_batch *batch, const mp_sl_params *params, double *cost_map,
                          int32_t *index_map, int32_t *path_idx, int32_t *status, int32_t *layers_done,
                          double *best_cost, void *stream)
{
    return sl_plan_launch(batch, params, cost_map, index_map, path_idx, status, layers_done, best_cost, nullptr,
                          nullptr, stream);
}

extern "C" int mp_sl_plan_ordered(const mp_sl_batch *batch, const mp_sl_params *params, double *cost_map,
                                  int32_t *index_map, int32_t *path_idx, int32_t *status, int32_t *layers_done,
                                  double *best_cost, int32_t *order_ws, void *stream)
{
    return sl_plan_launch(batch, params, cost_map, index_map, path_idx, status, layers_done, best_cost, nullptr,
                          order_ws, stream);
}

// mp_sl_plan_ordered for the whole replan, whose front end has already left every scene's weight
// bin (sl_weight_narrow_kernel's value) in order_ws[0..B); lattices of more than 16 lanes take
// their own weights as usual.
int mp_sl_plan_keyed(const mp_sl_batch *batch, const mp_sl_params *params, double *cost_map, int32_t *index_map,
                     int32_t *path_idx, int32_t *status, int32_t *layers_done, double *best_cost,
                     int32_t *order_ws, void *stream)
{
    return sl_plan_launch(batch, params, cost_map, index_map, path_idx, status, layers_done, best_cost, nullptr,
                          order_ws, stream, true);
}

extern "C" int mp_sl_plan_counted(const mp_sl_batch *batch, const mp_sl_params *params, int32_t *path_idx,
                                  int32_t *status, int32_t *layers_done, double *best_cost,
                                  uint64_t *n_evaluated, void *stream)
{
    if (!n_evaluated) { mp_set_error("mp_sl_plan_counted: null counter"); return MP_E_BADARG; }
    if (batch && batch->n_l <= 16) {
        mp_set_error("mp_sl_plan_counted: lattices of up to 16 lanes evaluate every edge; nothing to count");
        return MP_E_UNSUPPORTED;
    }
    return sl_plan_launch(batch, params, nullptr, nullptr, path_idx, status, layers_done, best_cost,
                          (unsigned long long *)n_evaluated, nullptr, stream);
}

static int sl_plan_launch(const mp_sl_batch *batch, const mp_sl_params *params, double *cost_map,
                          int32_t *index_map, int32_t *path_idx, int32_t *status, int32_t *layers_done,
                          double *best_cost, unsigned long long *n_evaluated, int32_t *order_ws, void *stream,
                          bool keys_ready)
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
    a.pair_cap = 0;
    a.idx_rows = 0;
    a.n_evaluated = n_evaluated;
    a.order = nullptr;
    // MP_SL_NARROW=0 sends narrow lattices through the one-thread-per-node kernel as
    // well (the two must agree bit for bit; used by the tests)
    const bool narrow_ok = mp_knob_sl_narrow() != 0;
    if (n_l <= 16 && narrow_ok && !n_evaluated) {
        const int W = n_l <= 2 ? 2 : (n_l <= 4 ? 4 : (n_l <= 8 ? 8 : 16));
        int threads = ((n_l * W + 63) / 64) * 64;
        int cap_max = SLN_CAP_MAX;
        if (batch->ob_stride > 128 && n_l * n_l <= 32) {
            // a crowded scene: laying out thousands of pairs and testing them against every edge
            // is the work, and one wave per scene leaves the launch waiting for its slowest
            // scene; two waves share it, with a store twice the size (four waves were slower: what a
            // layer costs besides its pairs is paid by every wave)
            threads = 128;
            cap_max = 2 * SLN_CAP_MAX;
        }
        int ncap = (7 * batch->ob_stride) / 2;
        ncap = ncap > cap_max ? cap_max : ncap;
        if (mp_knob_sl_pair_cap() >= 0) ncap = mp_knob_sl_pair_cap(); // tests: force refills
        ncap = ncap < 2 ? 2 : (ncap > 2048 ? 2048 : ncap);
        ncap += ncap & 1;
        a.pair_cap = ncap;
        const size_t lds = sl_narrow_lds_bytes(n_s, n_l, W, ncap);
        if (lds > 64 * 1024) { mp_set_error("mp_sl_plan: too many layers for LDS"); return MP_E_UNSUPPORTED; }
        const dim3 grid(batch->B), block(threads);
        hipStream_t st = (hipStream_t)stream;
        if (order_ws && batch->B > MP_ORDER_MIN_BATCH && mp_knob_order() != 0) {
            if (!keys_ready)
                hipLaunchKernelGGL(sl_weight_narrow_kernel, dim3((batch->B + 255) / 256), dim3(256), 0, st, a, order_ws);
            mp_order_launch(batch->B, order_ws, order_ws + batch->B, st);
            a.order = order_ws + batch->B;
        }
        const bool nmaps = cost_map || index_map;
#define SLN_LAUNCH(WW)                                                                              \
    if (nmaps) hipLaunchKernelGGL((sl_plan_narrow_kernel<WW, true>), grid, block, lds, st, a);      \
    else hipLaunchKernelGGL((sl_plan_narrow_kernel<WW, false>), grid, block, lds, st, a)
        switch (W) {
        case 2: SLN_LAUNCH(2); break;
        case 4: SLN_LAUNCH(4); break;
        case 8: SLN_LAUNCH(8); break;
        default: SLN_LAUNCH(16); break;
        }
#undef SLN_LAUNCH
        return launch_check("mp_sl_plan");
    }
    // pair store: 5 pairs per obstacle slot (an obstacle of radius r meets about
    // 2r/d_s + 1 samples, a few of them twice where two layers share a sample; the benchmark's
    // scenarios have 146 pairs on average and 227 at most for 50 obstacles, and laying the
    // store out again for the quarter of them that 175 entries did not hold cost 5 %),
    // fewer if LDS would not hold it; layers past its end test their pairs directly
    const int threads = ((n_l + 63) / 64) * 64;
    const bool shared_row = batch->l_rstride == 0;
    const bool maps = cost_map || index_map;
    // Predecessor rows in LDS.  With one lateral row for all layers only the first few layers
    // look at other lanes (the rest keep every lane and need no row), so the plain instance
    // holds SL_ROWS_COMPACT rows: at 64 layers x 128 lanes that is 2 KB instead of 8, and the
    // CU holds the workgroups its registers allow.  A scenario in which more layers walk is
    // planned again by a second launch with every row in LDS (only such scenarios do anything
    // in it).  The other instances (maps written, counting, a lateral grid per layer: every
    // layer may walk) keep every row.
    const int rows_compact = mp_knob_sl_rows() > 0 ? mp_knob_sl_rows() : SL_ROWS_COMPACT; // tests: force the second launch
    const bool compact = shared_row && !maps && !n_evaluated && n_s > rows_compact;
    const int rows = compact ? rows_compact : n_s;
    int cap = 5 * batch->ob_stride;
    if (mp_knob_sl_pair_cap() >= 0) cap = mp_knob_sl_pair_cap(); // tests: force the overflow path
    else {
        // registers allow 16 waves per CU; a store that lets LDS hold fewer workgroups than
        // that costs more (idle SIMDs) than laying a smaller store out again every few layers
        // (512 B less than the CU's share: the kernel's static LDS and the allocation granule)
        const size_t budget = (size_t)10 * 1024 * (threads / 64) - 512, fixed = sl_plan_lds_bytes(n_s, n_l, 0, rows);
        const int fit = budget > fixed ? (int)((budget - fixed) / (3 * sizeof(double) + 2)) : 0;
        const int floor_cap = 128;
        if (cap > fit) cap = fit > floor_cap ? fit : (cap < floor_cap ? cap : floor_cap);
    }
    cap = cap < 2 ? 2 : (cap > 16384 ? 16384 : cap);
    cap &= ~1;
    while (cap > 2 && sl_plan_lds_bytes(n_s, n_l, cap, rows) > 60 * 1024) cap = (cap / 2 + 1) & ~1;
    a.pair_cap = cap;
    a.idx_rows = rows;
    const size_t lds = sl_plan_lds_bytes(n_s, n_l, cap, rows);
    if (lds > 160 * 1024) { mp_set_error("mp_sl_plan: lattice too large for LDS"); return MP_E_UNSUPPORTED; }
    // the second launch of the compact form: every row in LDS, a store sized for what is left
    int cap_full = cap;
    while (cap_full > 2 && sl_plan_lds_bytes(n_s, n_l, cap_full, n_s) > 60 * 1024) cap_full = (cap_full / 2 + 1) & ~1;
    const size_t lds_full = sl_plan_lds_bytes(n_s, n_l, cap_full, n_s);
    if (lds_full > 160 * 1024) { mp_set_error("mp_sl_plan: lattice too large for LDS"); return MP_E_UNSUPPORTED; }
    const dim3 grid(batch->B), block(threads);
    hipStream_t st = (hipStream_t)stream;
    if (order_ws && !n_evaluated && batch->B > MP_ORDER_MIN_BATCH && mp_knob_order() != 0) {
        // heaviest scenarios first: weights -> order_ws[0..B), order -> order_ws[B..2B)
        hipLaunchKernelGGL(sl_weight_kernel, dim3((batch->B + 3) / 4), dim3(4 * MP_WAVE), 0, st, a, order_ws);
        mp_order_launch(batch->B, order_ws, order_ws + batch->B, st);
        a.order = order_ws + batch->B;
    }
    if (n_evaluated) {
        if (shared_row) hipLaunchKernelGGL((sl_plan_kernel<true, false, true>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((sl_plan_kernel<false, false, true>), grid, block, lds, st, a);
    } else if (shared_row && !maps) {
        hipLaunchKernelGGL((sl_plan_kernel<true, false>), grid, block, lds, st, a);
        if (compact) {
            a.order = nullptr;
            a.pair_cap = cap_full;
            a.idx_rows = n_s;
            hipLaunchKernelGGL((sl_plan_kernel<true, false, false, true>), grid, block, lds_full, st, a);
        }
    } else if (shared_row) hipLaunchKernelGGL((sl_plan_kernel<true, true>), grid, block, lds, st, a);
    else if (!maps) hipLaunchKernelGGL((sl_plan_kernel<false, false>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((sl_plan_kernel<false, true>), grid, block, lds, st, a);
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
    a.pair_cap = 0;
    a.idx_rows = 0;
    a.n_evaluated = nullptr;
    a.order = nullptr;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(fill_i32_kernel, dim3((batch->B + 255) / 256), dim3(256), 0, st, status, batch->B,
                       MP_SL_FOUND);
    hipLaunchKernelGGL(sl_edge_cost_kernel, dim3(batch->n_s, batch->B), dim3(256), 0, st, a);
    return launch_check("mp_sl_edge_cost");
}

extern "C" int mp_sl_dp(int32_t B, int32_t n_s, int32_t n_l, const int32_t *n_layers,
                        const double *edge_cost, const mp_sl_params *params, double *cost_map,
                        int32_t *index_map, int32_t *path_idx, int32_t *status, int32_t *layers_done,
                        double *best_cost, void *stream)
{
    if (B <= 0 || n_s <= 0 || n_l <= 0) { mp_set_error("non-positive B/n_s/n_l"); return MP_E_BADARG; }
    if (!edge_cost || !params || !index_map || !path_idx || !status || !layers_done) {
        mp_set_error("null array (index_map is required)");
        return MP_E_BADARG;
    }
    size_t lds = ((size_t)2 * n_l + 16) * sizeof(double);
    if (lds > 64 * 1024) { mp_set_error("mp_sl_dp: n_l too large"); return MP_E_UNSUPPORTED; }
    int threads = n_l >= 256 ? 512 : 256;
    hipLaunchKernelGGL(sl_dp_kernel, dim3(B), dim3(threads), lds, (hipStream_t)stream, n_s, n_l, n_layers,
                       edge_cost, *params, cost_map, index_map, path_idx, status, layers_done, best_cost);
    return launch_check("mp_sl_dp");
}
