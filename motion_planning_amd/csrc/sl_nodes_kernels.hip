// SL lattice edge costs for lattices whose nodes carry their own arc length.
//
// The reference reads s_map[s][l] and l_map[s][l] per node
// (Planning/DP_Path/path_planner.py:104-123), so nothing ties the nodes of a
// layer to one arc length; SLMap never builds such a lattice (sl_map.py:69-78),
// which is why the fused kernels assume a layer abscissa and share the sample
// set among a layer's edges.  This kernel takes the general lattice: one thread
// per edge evaluates cal_cost as the reference writes it (quintic coefficients,
// the four sampled orders, NumPy's pairwise sums, the distance of every sample
// to every obstacle), with no limit on the samples of an edge.  It is the slow,
// general companion of sl_edge_cost_kernel; mp_sl_dp relaxes its edge tensor.
#pragma clang fp contract(off)
#include "mp_device.h"

struct NodeDev {
    int32_t n_s, n_l, ob_stride;
    const double *s_map, *l_map; // [B][n_s][n_l]
    const double *ego_sl;
    const double *ob_s, *ob_l;
    const int32_t *n_ob;
    const double *ob_dist;
    const int32_t *n_layers;
    mp_sl_params p;
    double *edge_cost;
    int32_t *status;
};

// One sampled quantity of an edge: f(k) for sample k, summed in np.sum's order.
struct EdgeTerm {
    const double *a;     // quintic coefficients
    double t_st, t_en, dt, delta, mid;
    int n, order;        // order 0: (l - mid)^2, 1..3: the derivative squared
    __device__ __forceinline__ double operator()(int k) const
    {
        const double x = arange_value(k, n, t_st, t_en, dt, delta);
        const double v = quintic_eval(a, x, order);
        const double d = order == 0 ? v - mid : v;
        return d * d;
    }
};

struct EdgeTermAt {
    const EdgeTerm &f;
    int lo;
    __device__ __forceinline__ double operator()(int i) const { return f(lo + i); }
};

// np.sum of f(0 .. n-1): blocks of up to 128 elements with eight running sums,
// longer ranges split in halves (NumPy's pairwise summation).  The halving is
// walked with an explicit stack: 2^20 samples are 13 levels deep.
__device__ double edge_np_sum(const EdgeTerm &f, int n)
{
    struct Frame {
        int lo, n, state;
        double left;
    } stk[16];
    int sp = 0;
    stk[0] = Frame{0, n, 0, 0.0};
    double ret = 0.0;
    while (sp >= 0) {
        Frame &t = stk[sp];
        if (t.n <= 128) {
            ret = np_sum_gen(t.n, EdgeTermAt{f, t.lo});
            --sp;
            continue;
        }
        int n2 = t.n / 2;
        n2 -= n2 % 8;
        if (t.state == 0) {
            t.state = 1;
            stk[sp + 1] = Frame{t.lo, n2, 0, 0.0};
            ++sp;
        } else if (t.state == 1) {
            t.left = ret;
            t.state = 2;
            stk[sp + 1] = Frame{t.lo + n2, t.n - n2, 0, 0.0};
            ++sp;
        } else {
            ret = t.left + ret;
            --sp;
        }
    }
    return ret;
}

#define SLN_K_MAX (1 << 20) // samples per edge (2^20 * d_s of arc length)

__global__ void sl_edge_cost_nodes_kernel(NodeDev a)
{
    const int s = blockIdx.x, b = blockIdx.y;
    const int n_l = a.n_l, n_sm = a.n_s;
    int n_s = n_sm;
    if (a.n_layers) n_s = min(max(a.n_layers[b], 0), n_sm);
    double *out = a.edge_cost + ((size_t)b * n_sm + s) * n_l * n_l;
    if (s >= n_s) {
        for (int e = threadIdx.x; e < n_l * n_l; e += blockDim.x) out[e] = __builtin_nan("");
        return;
    }
    const double *s_map = a.s_map + (size_t)b * n_sm * n_l, *l_map = a.l_map + (size_t)b * n_sm * n_l;
    const double *ob_s = a.ob_s + (size_t)b * a.ob_stride, *ob_l = a.ob_l + (size_t)b * a.ob_stride;
    const int n_ob = min(max(a.n_ob[b], 0), a.ob_stride);
    const double r = a.ob_dist[b];
    const double d_s = a.p.d_s, w_d = a.p.w_d, pen = 3.0 * a.p.no_path_cost;
    const int n_src = s == 0 ? 1 : n_l;
    // guide row: the predecessor's; the ego edge reads row -1, i.e. the last one (path_planner.py:112)
    const double mid = l_map[(size_t)(s == 0 ? n_s - 1 : s - 1) * n_l + a.p.mid_l];
    int err = 0;
    for (int e = threadIdx.x; e < n_l * n_l; e += blockDim.x) {
        const int l2 = e / n_l, l1 = e - l2 * n_l;
        if (l1 >= n_src) { out[e] = __builtin_nan(""); continue; }
        const double p1s = s == 0 ? a.ego_sl[2 * b] : s_map[(size_t)(s - 1) * n_l + l1];
        const double p1l = s == 0 ? a.ego_sl[2 * b + 1] : l_map[(size_t)(s - 1) * n_l + l1];
        const double p2s = s_map[(size_t)s * n_l + l2], p2l = l_map[(size_t)s * n_l + l2];
        double co[6];
        quintic_coef(p1l, 0.0, 0.0, p2l, 0.0, 0.0, p2s - p1s, co);
        EdgeTerm f;
        f.a = co; f.t_st = p1s; f.t_en = p2s; f.dt = d_s; f.mid = mid;
        const int K = arange_count(p1s, p2s, d_s, SLN_K_MAX, &f.n, &f.delta);
        if (K < 0) { err = MP_SL_E_CAPACITY; out[e] = __builtin_nan(""); continue; }
        if (K == 0 && n_ob > 0) { err = err ? err : MP_SL_E_EMPTY_EDGE; out[e] = __builtin_nan(""); continue; }
        f.order = 0;
        const double guide = d_s * edge_np_sum(f, K);
        double sm[3];
        for (int o = 1; o <= 3; ++o) {
            f.order = o;
            sm[o - 1] = edge_np_sum(f, K);
        }
        const double smooth = d_s * ((sm[0] + sm[1]) + sm[2]);
        double ob_cost = 0.0;
        if (n_ob > 0) {
            // cal_dist_arr per obstacle, the smallest of them against ob_dist (path_planner.py:117-124)
            double dmin = MP_INF;
            for (int k = 0; k < K; ++k) {
                const double x = arange_value(k, f.n, p1s, p2s, d_s, f.delta);
                const double ll = quintic_eval(co, x, 0);
                for (int o = 0; o < n_ob; ++o) {
                    const double dx = x - ob_s[o], dy = ll - ob_l[o];
                    const double d = __dsqrt_rn(dx * dx + dy * dy);
                    dmin = d < dmin ? d : dmin;
                }
            }
            ob_cost = dmin < r ? pen : 0.0;
        }
        out[e] = (w_d * guide + (1.0 - w_d) * smooth) + ob_cost;
    }
    if (err) atomicMin(&a.status[b], err); // codes are negative: the capacity code wins over the empty edge
}

__global__ void sln_fill_i32_kernel(int32_t *p, int n, int32_t v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

extern void mp_set_error(const char *fmt, ...);

extern "C" int mp_sl_edge_cost_nodes(int32_t B, int32_t n_s, int32_t n_l, const double *s_map, const double *l_map,
                                     const double *ego_sl, const double *ob_s, const double *ob_l,
                                     int32_t ob_stride, const int32_t *n_ob, const double *ob_dist,
                                     const int32_t *n_layers, const mp_sl_params *params, double *edge_cost,
                                     int32_t *status, void *stream)
{
    if (B <= 0 || n_s <= 0 || n_l <= 0 || ob_stride < 0) { mp_set_error("non-positive B/n_s/n_l"); return MP_E_BADARG; }
    if (!s_map || !l_map || !ego_sl || !n_ob || !ob_dist || !params || !edge_cost || !status ||
        (ob_stride > 0 && (!ob_s || !ob_l))) {
        mp_set_error("null array");
        return MP_E_BADARG;
    }
    if (params->mid_l < 0 || params->mid_l >= n_l) { mp_set_error("mid_l out of range"); return MP_E_BADARG; }
    if (!(params->d_s > 0)) { mp_set_error("d_s must be positive"); return MP_E_BADARG; }
    if (B > 65535) { mp_set_error("mp_sl_edge_cost_nodes: B > 65535"); return MP_E_UNSUPPORTED; }
    NodeDev a{n_s, n_l, ob_stride, s_map, l_map, ego_sl, ob_s, ob_l, n_ob, ob_dist, n_layers, *params, edge_cost,
              status};
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sln_fill_i32_kernel, dim3((B + 255) / 256), dim3(256), 0, st, status, B, MP_SL_FOUND);
    hipLaunchKernelGGL(sl_edge_cost_nodes_kernel, dim3(n_s, B), dim3(256), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mp_set_error("mp_sl_edge_cost_nodes: %s", hipGetErrorString(e)); return MP_E_LAUNCH; }
    return MP_OK;
}
