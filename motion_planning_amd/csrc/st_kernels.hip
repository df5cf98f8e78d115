// ST-lattice speed DP and ST blocked-interval kernels for gfx950.
//
// Reference: Planning/DP_Speed/speed_planner.py:57-162 (DP) and
// Planning/DP_Speed/st_map.py:24-60 (blocked intervals).
//
// The DP is layer-sequential and history dependent: acceleration and jerk of
// an edge read the predecessor indices already fixed for earlier layers
// (speed_planner.py:124-136).  Every term of the edge cost is evaluated with
// the reference's own operation order and without fused multiply-adds
// (__dadd_rn/__dmul_rn/__ddiv_rn), so costs and therefore every first-minimum
// choice are bit-identical to the Python planner.  Terms that depend only on
// an integer difference (speed on j-p, acceleration on j-2p+pp, jerk on
// j-3p+3pp-ppp) are tabulated once per scenario in LDS, which removes all
// divisions from the edge loop without changing a single bit.
#include <algorithm>
#include <type_traits>

#include "mp_device.h"

#ifdef ST_COUNT
// Development aid, built only with -DST_COUNT (tools/count_st.py): what the walks of the
// split relaxation do.  0 wave steps, 1 lane steps (edges priced), 2 wave steps that took
// the quiet obstacle sum, 4 lanes stopped by the bound, 5 wave passes, 8 first wave steps,
// 9 / 10 later lane steps that did / did not improve the lane's best; 11 / 12 cells of a second
// sweep that were not priced / all of them (their priced edges count as lane steps too), 13 those
// of them that were too far ahead for their intervals to be looked at.
__device__ unsigned long long st_cnt[16];
#define ST_CNT_WAVE(i) do { const unsigned long long m_ = __ballot(1); if ((int)(threadIdx.x & 63) == __ffsll((long long)m_) - 1) atomicAdd(&st_cnt[i], 1ull); } while (0)
#define ST_CNT_LANE(i) atomicAdd(&st_cnt[i], 1ull)
extern "C" int mp_debug_st_counters(unsigned long long *out, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(st_cnt), sizeof(st_cnt)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(st_cnt), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#else
#define ST_CNT_WAVE(i) do {} while (0)
#define ST_CNT_LANE(i) do {} while (0)
#endif

struct StDev {
    int32_t n_t, n_s_max, r_max, r_tab; // r_tab: interval slots per cell in the LDS cell tables
    int32_t lanes_per_cell;             // 4 or 8 (st_lanes_per_cell)
    int32_t r_pitch;                    // interval slots per time layer in LDS (layers with more are read from global memory)
    int32_t k_prefix;                   // > 0: cells 0..k_prefix are planned first, for a bound on the plan's cost (see st_dp_body)
    int32_t rec_cells;                  // cells the LDS cell records are laid out for (n_s_max, or k_prefix + 1)
    int32_t n_one;                      // with a first sweep: the widest grid that is planned in one sweep all the same
    int32_t tab_half;                   // the acceleration / jerk tables hold differences -tab_half..tab_half (jerk of layers >= 3: twice that)
    const int32_t *n_s;
    const double *s_down, *s_up;
    const int32_t *n_reg;
    const double *cur_vel;
    double speed_lim, ds, dt, dt2, dt3; // dt2 = dt**2, dt3 = dt**3 as Python evaluates them
    double *cost;
    int32_t *pre;
    int32_t *path_s_ind, *path_len;
    const int32_t *order; // workgroup i plans scenario order[i] (null: scenario i)
};

// Blocked intervals of one time layer as seen from one cell (arc length s).
// speed_planner.py:144-150 adds, per interval and in order: 1e8 if the cell is
// inside; else 1e3*(front+s-down)^2 if the interval starts ahead within the
// speed-dependent look-ahead `front`; else 1e3*(25+s-up)^2 if it ended less
// than 25 m behind.  Only the middle case depends on the edge, so each interval
// is reduced to (dn, cst): the term is the front formula when s >= dn - front
// and cst otherwise, with dn = +inf for intervals that are not ahead of the
// cell (inf - front = inf, so the front test then never fires) and cst = the
// inside / behind constant or 0.  Evaluating the terms in interval order with
// the reference's own operations keeps the sum bit-identical.
template <int RT>
struct StCell {
    double dn[RT > 0 ? RT : 1], cst[RT > 0 ? RT : 1];
    bool monotone; // the obstacle cost cannot decrease as the edge speed grows

    __device__ __forceinline__ static void classify(double s, double dnk, double uk, double *dn_out, double *cst_out)
    {
        double back = 0.0;
        if (s > uk && s <= __dadd_rn(uk, 25.0)) {
            double d = __dsub_rn(__dadd_rn(25.0, s), uk);
            back = __dmul_rn(1e3, __dmul_rn(d, d));
        }
        if (s >= dnk && s <= uk) { *dn_out = MP_INF; *cst_out = 1e8; }
        else if (s < dnk) { *dn_out = dnk; *cst_out = back; }
        else { *dn_out = MP_INF; *cst_out = back; }
    }

    __device__ __forceinline__ void init(double s, const double *down, const double *up, int R)
    {
        monotone = true;
#pragma unroll
        for (int k = 0; k < RT; ++k) {
            if (k < R) {
                classify(s, down[k], up[k], &dn[k], &cst[k]);
                // an interval whose end precedes its start could trade a large
                // "behind" constant for a small front term
                if (dn[k] != MP_INF && cst[k] != 0.0) monotone = false;
            } else { dn[k] = MP_INF; cst[k] = 0.0; } // padding adds exact zeros
        }
    }

    // in-order sum of the constants: what the obstacle terms of an edge into the cell add up to
    // at least, if the cell is monotone (padding adds exact zeros)
    __device__ __forceinline__ double quiet(double) const
    {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < RT; ++k) acc = (k == 0) ? cst[0] : __dadd_rn(acc, cst[k]);
        return acc;
    }

    __device__ __forceinline__ double ob(double s, double front) const
    {
        const double fs = __dadd_rn(front, s);
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < RT; ++k) {
            const double d = __dsub_rn(fs, dn[k]);
            const double t = (s >= __dsub_rn(dn[k], front)) ? __dmul_rn(1e3, __dmul_rn(d, d)) : cst[k];
            acc = (k == 0) ? t : __dadd_rn(acc, t); // 0.0 + t == t: every term is >= +0.0
        }
        return acc;
    }
};

// Any number of intervals: same terms, intervals re-read from LDS per edge.
struct StCellLds {
    const double *down, *up;
    int R;
    bool monotone;
    __device__ __forceinline__ void init(double s, const double *down_, const double *up_, int R_)
    {
        down = down_; up = up_; R = R_;
        monotone = true;
        for (int k = 0; k < R; ++k) {
            double dnk, c;
            StCell<1>::classify(s, down[k], up[k], &dnk, &c);
            if (dnk != MP_INF && c != 0.0) monotone = false;
        }
    }
    __device__ __forceinline__ double quiet(double s) const
    {
        double acc = 0.0;
        for (int k = 0; k < R; ++k) {
            double dnk, c;
            StCell<1>::classify(s, down[k], up[k], &dnk, &c);
            acc = __dadd_rn(acc, c);
        }
        return acc;
    }
    __device__ __forceinline__ double ob(double s, double front) const
    {
        const double fs = __dadd_rn(front, s);
        double acc = 0.0;
        for (int k = 0; k < R; ++k) {
            double dnk, c;
            StCell<1>::classify(s, down[k], up[k], &dnk, &c);
            const double d = __dsub_rn(fs, dnk);
            const double t = (s >= __dsub_rn(dnk, front)) ? __dmul_rn(1e3, __dmul_rn(d, d)) : c;
            acc = __dadd_rn(acc, t);
        }
        return acc;
    }
};

// LDS rows of the DP.  What the edge loop reads together sits together, so one 16-byte
// read serves two tables: fs[d] = (front[d], speed[d]) and, per cell of a cost row,
// (cost, history word).  The history word of predecessor p of the current layer packs
// 8*(pp - 2p) in the low half and 8*(3pp - ppp - 3p) in the high half (on layer 2, whose
// jerk has no third predecessor, 8*(pp - 2p) again): byte offsets into the acceleration
// and jerk tables, written once per cell when its predecessor is fixed (st_store_hist)
// instead of two dependent reads per edge.
//
// Lanes per cell of the split relaxation.  Pruning makes the walks short, so in a large
// batch (throughput) the cost of a pass over the cells counts as much as the balance of the
// lanes: grids of up to 128 cells give a cell 4 lanes (two passes of 64 cells instead of four
// of 32 on a 97-cell grid), wider ones 8.  A small batch (latency) always takes 8 lanes: a layer
// is then one pass (measured, one plan per call, replayed graph: 29 / 61 / 101 cells take
// 50 / 60 / 73 us with 8 lanes and a workgroup as wide as the grid needs, 56 / 78 / 96 us with 4).
int mp_knob_st_lanes();
int mp_knob_st_prefix();
static inline int st_lanes_per_cell(int B, int n_s_max)
{
    if (mp_knob_st_lanes() > 0) return mp_knob_st_lanes(); // development aid
    return (B > 512 && n_s_max <= 128) ? 4 : 8;
}

// Acceleration and jerk terms of an edge as functions of an integer difference of cell indices
// (speed_planner.py:124-136), with the reference's operations: what the LDS tables hold.
__device__ __forceinline__ double st_acc_term(const StDev &a, int n, double *acv_out)
{
    const double acv = __ddiv_rn(__dmul_rn(__dmul_rn(1.0, (double)n), a.ds), a.dt2);
    *acv_out = acv;
    return __dmul_rn(__dmul_rn(acv, acv), a.dt);
}
__device__ __forceinline__ double st_jerk2_term(const StDev &a, double acv)
{
    const double jk = __ddiv_rn(acv, a.dt);
    return __dmul_rn(__dmul_rn(jk, jk), a.dt);
}
__device__ __forceinline__ double st_jerk3_term(const StDev &a, int m)
{
    const double jk = __ddiv_rn(__dmul_rn(__dmul_rn(1.0, (double)m), a.ds), a.dt3);
    return __dmul_rn(__dmul_rn(jk, jk), a.dt);
}

struct StSh {
    double *ch_prev, *ch_cur, *fs, *acc, *jerk2, *jerk3, *down, *up;
    short *pre16;
    int half; // acc / jerk2: index n + half; jerk3: index m + 2 * half
    __device__ __forceinline__ double &cost_prev(int p) const { return ch_prev[2 * p]; }
    __device__ __forceinline__ double &cost_cur(int j) const { return ch_cur[2 * j]; }
    __device__ __forceinline__ double &front(int d) const { return fs[2 * d]; }
    __device__ __forceinline__ double &speed(int d) const { return fs[2 * d + 1]; }
};

// History word of cell j of layer i, whose predecessor bp has just been fixed, as the next
// layer reads it when j is a predecessor there (see the LDS rows above): next to the cell's cost.
__device__ __forceinline__ int st_store_hist(const StSh &sh, int i, int N, int j, int bp)
{
    const int lo = 8 * (bp - 2 * j);
    int hi = lo; // layer 2 reads the acceleration offset twice (its jerk has no third predecessor)
    if (i >= 2) hi = 8 * (3 * bp - (int)sh.pre16[(size_t)(i - 1) * N + bp] - 3 * j);
    const int word = (lo & 0xffff) | (hi << 16);
    *(int *)(sh.ch_cur + 2 * j + 1) = word;
    return word;
}

// The cells a relaxation call takes: j_lo..j_hi of the layer.  u_cap: a number the plan's final
// cost cannot exceed (MP_INF: none known); a candidate whose cost is certain to exceed it need
// not be priced (see st_dp_body).  keep_cost (may be null): where the cells' costs of this layer
// are kept beside the cost row, indexed by cell.
struct StRange {
    int n_s; // cells of the scenario's grid
    int j_lo, j_hi;
    double u_cap;
    double *keep_cost;
};

// All cells of time layer i >= 2 owned by this thread.
//
// Exact pruning: every term of the edge cost is non-negative and floating-point
// addition is monotone, so predecessor p of cell j costs at least
// (ob(d) + speed[d]) + cost_prev[p], d = j - p.  Each cell evaluates p = j
// first (standing still) and walks p downwards, i.e. towards higher speeds; a
// predecessor whose bound exceeds the best cost so far is skipped, and because
// ob(d) and (above the speed limit) speed[d] never decrease with d, the walk
// stops once (ob(d) + speed[d]) + min(cost_prev) exceeds it.  Skipped
// predecessors are strictly worse, so cost, first-minimum index and every tie
// are exactly those of the full scan.
// down / up: the layer's intervals, in LDS or (a layer with more intervals than LDS has slots
// for) in global memory; every call site passes one kind, so the loads are typed.
template <class CELL>
__device__ __forceinline__ double st_relax_layer(const StDev &a, const StSh &sh, const StRange &rg, const double *down,
                                                 const double *up, int i, int n_s, int N, int R,
                                                 int d_lim, double m_prev, double *cost_g, int32_t *pre_g)
{
    const double ds = a.ds;
    const short *pre1 = sh.pre16 + (size_t)(i - 1) * N;
    const short *pre2 = sh.pre16 + (size_t)(i >= 2 ? i - 2 : 0) * N;
    double m_loc = MP_INF;
    for (int j = rg.j_lo + threadIdx.x; j <= rg.j_hi; j += blockDim.x) {
        const double s = __dmul_rn((double)j, ds);
        CELL cell;
        cell.init(s, down, up, R);
        double best = MP_INF;
        int bp = -1;
        for (int p = j; p >= 0; --p) {
            const int d = j - p;
            const double cp = sh.cost_prev(p);
            const double base = __dadd_rn(cell.ob(s, sh.front(d)), sh.speed(d));
            if (bp >= 0) {
                if (cell.monotone && d >= d_lim && __dadd_rn(base, m_prev) > best) break;
                if (__dadd_rn(base, cp) > best) continue;
            }
            const int pp = pre1[p];
            const int n = d - (p - pp);
            double jc;
            if (i == 2) jc = sh.jerk2[n + sh.half];
            else {
                const int ppp = pre2[pp];
                jc = sh.jerk3[(j - 3 * p + 3 * pp - ppp) + 2 * sh.half];
            }
            const double tot = __dadd_rn(__dadd_rn(base, sh.acc[n + sh.half]), jc);
            const double c = __dadd_rn(tot, cp);
            if (bp < 0 || c <= best) { best = c; bp = p; } // lower index wins ties
        }
        sh.cost_cur(j) = best;
        sh.pre16[(size_t)i * N + j] = (short)bp;
        st_store_hist(sh, i, N, j, bp);
        if (rg.keep_cost) rg.keep_cost[j] = best;
        if (cost_g) cost_g[(size_t)i * N + j] = best;
        if (pre_g) pre_g[(size_t)i * N + j] = bp;
        m_loc = best < m_loc ? best : m_loc;
    }
    return m_loc;
}

// The cells j_lo..j_hi of a layer when a bound u_cap on the plan's final cost is known (the
// second sweep of st_dp_body), one thread per cell.  Nearly all of these cells lie far ahead or
// behind an obstacle and cost more than the bound whatever their predecessor: the constants of
// the intervals a cell is inside of or has just passed are part of every edge into it (where no
// interval trades its constant for a front term), so if their in-order sum plus the cheapest
// cell of the layer before already exceeds the bound the cell keeps an infinite cost and itself
// as predecessor, unpriced.  The others are walked as st_relax_layer walks them, with the bound
// in the stop and skip tests, and from the last cell of the layer before that costs no more than
// the bound (reach[i - 1]; a predecessor above the bound decides nothing): a cell far ahead of
// it starts at a speed whose cost alone ends the walk at once.  reach[i] takes the cells of this
// layer within the bound.
template <class CELL>
__device__ __forceinline__ double st_relax_layer_capped(const StDev &a, const StSh &sh, const StRange &rg,
                                                        const double *down, const double *up, int i, int N, int R,
                                                        int d_lim, double m_prev, int *reach)
{
    const double ds = a.ds, u_cap = rg.u_cap;
    const short *pre1 = sh.pre16 + (size_t)(i - 1) * N;
    const short *pre2 = sh.pre16 + (size_t)(i >= 2 ? i - 2 : 0) * N;
    const int j_reach = reach[i - 1] >= 0 ? reach[i - 1] : N;
    double m_loc = MP_INF;
    for (int j = rg.j_lo + threadIdx.x; j <= rg.j_hi; j += blockDim.x) {
        double best = MP_INF;
        int bp = j;
        ST_CNT_LANE(12);
        // Too far ahead of every cell within the bound: the slowest edge that reaches the cell from
        // one of them is above the speed limit (from there on the speed term only grows with the
        // distance), and that term alone, on the cheapest cell of the layer before, exceeds the
        // bound; every candidate would be stopped or skipped by the tests below, so the cell's
        // intervals are not even looked at.  (Whole waves of cells at the far end of the grid.)
        const int d0 = j - min(j, j_reach);
        if (d0 >= d_lim && __dadd_rn(sh.speed(d0), m_prev) > u_cap) {
            ST_CNT_LANE(13);
            sh.cost_cur(j) = best;
            sh.pre16[(size_t)i * N + j] = (short)bp;
            st_store_hist(sh, i, N, j, bp);
            continue;
        }
        const double s = __dmul_rn((double)j, ds);
        CELL cell;
        cell.init(s, down, up, R);
        const double ob_quiet = cell.quiet(s);
        if (cell.monotone && __dadd_rn(ob_quiet, m_prev) > u_cap) {
            ST_CNT_LANE(11);
        } else {
            for (int p = min(j, j_reach); p >= 0; --p) {
                const int d = j - p;
                const double cp = sh.cost_prev(p);
                const double base = __dadd_rn(cell.ob(s, sh.front(d)), sh.speed(d));
                const double cap = best < u_cap ? best : u_cap;
                if (cell.monotone && d >= d_lim && __dadd_rn(base, m_prev) > cap) break;
                if (__dadd_rn(base, cp) > cap) continue;
                ST_CNT_LANE(1);
                const int pp = pre1[p];
                const int n = d - (p - pp);
                // (the tables of a launch with a first sweep end where its cells' differences end;
                // the few edges priced here compute what lies beyond)
                double jc, ac, acv;
                const bool n_in = abs(n) <= sh.half;
                if (n_in) ac = sh.acc[n + sh.half];
                else ac = st_acc_term(a, n, &acv);
                if (i == 2) {
                    if (n_in) jc = sh.jerk2[n + sh.half];
                    else jc = st_jerk2_term(a, acv);
                } else {
                    const int m = j - 3 * p + 3 * pp - pre2[pp];
                    if (abs(m) <= 2 * sh.half) jc = sh.jerk3[m + 2 * sh.half];
                    else jc = st_jerk3_term(a, m);
                }
                const double tot = __dadd_rn(__dadd_rn(base, ac), jc);
                const double c = __dadd_rn(tot, cp);
                if (c <= best) { best = c; bp = p; } // lower index wins ties
            }
        }
        if (best <= u_cap) atomicMax(reach + i, j);
        sh.cost_cur(j) = best;
        sh.pre16[(size_t)i * N + j] = (short)bp;
        st_store_hist(sh, i, N, j, bp);
        m_loc = best < m_loc ? best : m_loc;
    }
    return m_loc;
}

// Per-cell interval records in LDS, one per cell: r_tab pairs (dn, cst); then two ints: 16
// times the smallest predecessor distance at which any interval takes the front formula,
// and a word that is non-zero when the cell's obstacle cost is not monotone in the edge
// speed; then, as u16 x 8, 16 times that distance for each interval (0xffff: never).
// Without interval slots (r_tab = 0) a record is the two ints alone.
struct StTab {
    double *rec;
    int stride; // doubles per record
    int r_tab;
    __device__ __forceinline__ double *cell(int j) const { return rec + (size_t)j * stride; }
    __device__ __forceinline__ int *quiet(int j) const { return (int *)(cell(j) + 2 * r_tab); }
};
__host__ __device__ __forceinline__ int st_rec_stride(int r_tab) { return r_tab > 0 ? 2 * r_tab + 4 : 1; }

// Bytes of dynamic LDS in front of the cell records (see st_dp_kernel).
__host__ __device__ __forceinline__ size_t st_tab_offset(int N, int r_max, int n_t, int tab_half)
{
    size_t b = ((size_t)6 * N + 8 * (tab_half + 1) + 2 * r_max + n_t + 16) * sizeof(double) + ((size_t)n_t * N + 1) * sizeof(short) +
               2 * sizeof(int);
    return (b + 15) & ~(size_t)15;
}
__host__ __device__ __forceinline__ size_t st_tab_bytes(int N, int r_tab)
{
    return (size_t)N * st_rec_stride(r_tab) * sizeof(double);
}
// (StDev.rec_cells: the cells the records are laid out for: the whole grid, or, in a launch with
// a first sweep, its cells 0..k_prefix only: the second sweep keeps a cell's
// intervals in registers, and a scenario of such a launch that is planned in one sweep has the
// kept rows' room behind the records for its other cells; see st_dp_body.)
// Behind the records, with a first sweep: its cells' costs of every layer (double [n_t][k+1]), the
// layers' minima over them (double [n_t]) and, per layer, the last cell that costs no more than the
// bound (int [n_t]).  (Their history words are written again from the predecessor rows.)
__host__ __device__ __forceinline__ size_t st_keep_offset(int N, int r_slots, int n_t, int r_tab, int rec_cells,
                                                          int tab_half)
{
    const size_t b = st_tab_offset(N, r_slots, n_t, tab_half) + st_tab_bytes(rec_cells, r_tab);
    return (b + 15) & ~(size_t)15;
}
__host__ __device__ __forceinline__ size_t st_keep_bytes(int n_t, int k_prefix)
{
    if (k_prefix <= 0) return 0;
    const size_t cells = (size_t)n_t * (k_prefix + 1);
    return (cells + n_t) * sizeof(double) + n_t * sizeof(int);
}

// All cells of time layer i >= 2, ST_SPLIT lanes per cell.
//
// A thread per cell leaves most lanes idle (cell j has j+1 predecessors and a
// wave runs as long as its longest cell).  Here the lanes of a wave hold
// neighbouring cells, and the ST_SPLIT lanes of a cell take predecessors
// p = j-r, j-r-ST_SPLIT, ... (r = lane within the cell): the waves of a workgroup then
// carry almost equal work, and neighbouring lanes read neighbouring table
// entries.  Each lane keeps its own first minimum (walking towards lower p with
// "<="), the partial results are merged on (cost, p), so the lowest p among
// equal costs wins exactly as in the reference's list.index(min(...))
// (speed_planner.py:102-111).  The monotone bound of st_relax_layer ends a
// lane's walk early; what it skips is strictly worse than that lane's best.
//
// Whether an interval ahead of a cell takes the front formula, s >= dn - front[d], can only
// turn from false to true as the predecessor distance d grows (front[] does not decrease and
// rounding is monotone), so the classification of a layer finds, with that very test, the
// first distance at which it holds; the edge loop then compares integers.  Pruning makes the
// walks short (three or four steps per lane on a 97-cell grid), so what a pass over the
// cells costs outside the walk matters as much as the walk: a cell's intervals, fire
// distances and flags are one record, read with constant offsets from one address.
// Records are indexed from the range's first cell.
template <int RT, int ST_SPLIT>
__device__ __forceinline__ double st_relax_layer_split(const StDev &a, const StSh &sh, const StTab &tb,
                                                       const StRange &rg, int i,
                                                       int n_s, int N, int R, int d_lim, double m_prev,
                                                       double *cost_g, int32_t *pre_g)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    const double ds = a.ds;
    const int n1 = sh.half;
    // (every record's quiet word and flag word are at their reset values: set at the start of the
    // plan and put back by whoever read them last, see below)
    const double front_step = sh.front(n_s > 1 ? 1 : 0);
    const double inv_front_step = front_step > 0.0 ? 1.0 / front_step : 0.0; // for an estimate only
    // q / n_c by multiplication: exact while q * n_c < 2^32 (q < 8 * n_c, n_c <= 1024); a
    // single-cell range has no 32-bit multiplier and needs none
    const int j_lo = rg.j_lo, n_c = rg.j_hi - rg.j_lo + 1;
    const unsigned n_c_magic = n_c > 1 ? 0xffffffffu / (unsigned)n_c + 1u : 0u;
    for (int q = tid; q < n_c * RT; q += nt) {
        const int k = n_c > 1 ? (int)__umulhi((unsigned)q, n_c_magic) : q, j = j_lo + (q - k * n_c);
        double *rec = tb.cell(j - j_lo);
        double dnk = MP_INF, ck = 0.0; // slots beyond the layer's intervals never fire and add exact zeros
        unsigned e_fire = 0xffffu;
        if (k < R) {
            const double s = __dmul_rn((double)j, ds);
            StCell<1>::classify(s, sh.down[k], sh.up[k], &dnk, &ck);
            if (dnk != MP_INF) {
                // the estimate only has to be close; the two loops make it exact
                int dq = n_s;
                if (front_step > 0.0) {
                    const double g = (dnk - s) * inv_front_step;
                    dq = g < (double)n_s ? (g > 0.0 ? (int)g : 0) : n_s;
                }
                while (dq > 0 && s >= __dsub_rn(dnk, sh.front(dq - 1))) --dq;
                while (dq < n_s && !(s >= __dsub_rn(dnk, sh.front(dq)))) ++dq;
                e_fire = (unsigned)dq * 16u;
                atomicMin(tb.quiet(j - j_lo), (int)e_fire);
                // an interval whose end precedes its start could trade a large "behind"
                // constant for a small front term
                if (ck != 0.0) atomicOr(tb.quiet(j - j_lo) + 1, 1);
            }
        }
        rec[2 * k] = dnk;
        rec[2 * k + 1] = ck;
        ((unsigned short *)(rec + 2 * tb.r_tab + 1))[k] = (unsigned short)e_fire;
    }
    __syncthreads();
    const int r = tid & (ST_SPLIT - 1), g = tid / ST_SPLIT, G = nt / ST_SPLIT;
    const double *jerk_row = (i == 2) ? sh.jerk2 + n1 : sh.jerk3 + 2 * n1;
    const int e_lim = d_lim * 16;
    double m_loc = MP_INF;
    // cells are taken from the far end downwards: whatever is left over for the last,
    // partly filled pass are then the cells next to s = 0, which have the fewest
    // predecessors, instead of the ones with the most
    // ... and passes alternate the direction in which the lane groups take their cells: a
    // wave's position in the workgroup also fixes the SIMD it runs on, so a fixed direction
    // would hand the longest cells of every pass to the same SIMD
    bool flip = false;
    for (int jb = 0; jb < n_c; jb += G, flip = !flip) {
        const int j = rg.j_hi - (jb + (flip ? G - 1 - g : g));
        const bool live = j >= j_lo;
        double best = MP_INF;
        int bp = 0x7fffffff;
        if (live) {
            const double *rec = tb.cell(j - j_lo);
            double dn[RT > 0 ? RT : 1], cst[RT > 0 ? RT : 1];
            unsigned fire[(RT + 1) / 2 > 0 ? (RT + 1) / 2 : 1];
#pragma unroll
            for (int k = 0; k < RT; ++k) {
                const double2 v = ((const double2 *)rec)[k];
                dn[k] = v.x;
                cst[k] = v.y;
            }
#pragma unroll
            for (int k = 0; k < (RT + 1) / 2; ++k) fire[k] = ((const unsigned *)(rec + 2 * tb.r_tab + 1))[k];
            const int2 qf = *(const int2 *)(rec + 2 * tb.r_tab);
            // (this lane group is the only reader of the cell's quiet and flag words in this layer:
            // back to their reset values for the next layer's classification, behind a barrier)
            if (RT > 0 && r == 0) *(int2 *)(rec + 2 * tb.r_tab) = make_int2(0x7fffffff, 0);
            const int e_quiet = RT > 0 ? qf.x : 0;
            const int e_brk = (RT == 0 || qf.y == 0) ? e_lim : 0x7fffffff;
            const double s = __dmul_rn((double)j, ds);
            // what the obstacle sum comes to while no interval takes the front formula: the
            // same additions in the same order with every term at its constant
            double ob_quiet = 0.0;
#pragma unroll
            for (int k = 0; k < RT; ++k) ob_quiet = (k == 0) ? cst[0] : __dadd_rn(ob_quiet, cst[k]);
            const int ej = j * 16;
            const char *ch_j = (const char *)sh.ch_prev + ej;           // (cost, history) of p = j
            const char *acc_j = (const char *)(sh.acc + (j + n1));      // acc[(j - 2p + pp) + n1] = acc_j + low half
            const char *jerk_j = (const char *)(jerk_row + j);
            int be = -1;
            // one step of the walk at e = 16 * (j - p); FIRST: the lane's first step, where
            // nothing can be pruned yet and the candidate is the best so far whatever it costs
            auto step = [&](int e, auto first) -> bool {
                constexpr bool FIRST = decltype(first)::value;
                const double2 fsv = *(const double2 *)((const char *)sh.fs + e);
                ST_CNT_WAVE(0); ST_CNT_LANE(1);
                if (FIRST) ST_CNT_WAVE(8);
                // the lanes of a wave hold neighbouring cells at nearly the same distance, so
                // "no lane needs the front formula" is common (slow edges, obstacles far ahead)
                // and decided for the whole wave
                double obv;
                if (RT > 0 && __ballot(e >= e_quiet) == 0ull) { obv = ob_quiet; ST_CNT_WAVE(2); }
                else {
                    const double fsum = __dadd_rn(fsv.x, s);
                    obv = 0.0;
#pragma unroll
                    for (int k = 0; k < RT; ++k) {
                        const unsigned ek = (k & 1) ? (fire[k >> 1] >> 16) : (fire[k >> 1] & 0xffffu);
                        const double x = __dsub_rn(fsum, dn[k]);
                        const double t = ((unsigned)e >= ek) ? __dmul_rn(1e3, __dmul_rn(x, x)) : cst[k];
                        obv = (k == 0) ? t : __dadd_rn(obv, t); // 0.0 + t == t: every term is >= +0.0
                    }
                }
                const double base = __dadd_rn(obv, fsv.y);
                if constexpr (!FIRST)
                    if (e >= e_brk && __dadd_rn(base, m_prev) > best) { ST_CNT_LANE(4); return true; }
                // acceleration on j - 2p + pp, jerk on j - 3p + 3pp - ppp (speed_planner.py:124-136)
                const double2 chv = *(const double2 *)(ch_j - e);
                const int h = __double2loint(chv.y);
#ifdef ST_EXP_NOGATHER
                // timing-only build (tools/st_conflicts.sh; results are wrong): both table reads at
                // one address per cell, to see what their data-dependent addresses cost
                const double av = *(const double *)(acc_j + (h & 0));
                const double jv = *(const double *)(jerk_j + (h & 0));
#else
                const double av = *(const double *)(acc_j + (int)(short)(h & 0xffff));
                const double jv = *(const double *)(jerk_j + (h >> 16));
#endif
                const double tot = __dadd_rn(__dadd_rn(base, av), jv);
                const double c = __dadd_rn(tot, chv.x);
                if (!FIRST) { if (c <= best) ST_CNT_LANE(9); else ST_CNT_LANE(10); }
                if (FIRST || c <= best) { best = c; be = e; }
                return false;
            };
            int e = r * 16;
            if (e <= ej) {
                step(e, std::true_type());
                for (e += 16 * ST_SPLIT; e <= ej; e += 16 * ST_SPLIT)
                    if (step(e, std::false_type())) break;
            }
            if (be >= 0) bp = j - (be >> 4);
        }
        ST_CNT_WAVE(5);
        group_argmin<ST_SPLIT>(best, bp);
        if (live && r == 0) {
            sh.cost_cur(j) = best;
            sh.pre16[(size_t)i * N + j] = (short)bp;
            st_store_hist(sh, i, N, j, bp);
            if (rg.keep_cost) rg.keep_cost[j] = best;
            if (cost_g) cost_g[(size_t)i * N + j] = best;
            if (pre_g) pre_g[(size_t)i * N + j] = bp;
            m_loc = best < m_loc ? best : m_loc;
        }
    }
    return m_loc;
}

// One workgroup per scenario, one thread per s cell.  Dynamic LDS (N = n_s_max):
//   fs[N][2] row0[N][2] row1[N][2] acc[2W+2] jerk2[2W+2] jerk3[4W+4]   (W = tab_half)
//   down[n_t][r_pitch] up[n_t][r_pitch] last_col[n_t] red[2][8] | pre16[n_t][N] i16 | d_lim
//   | cell records [N][2 r_tab + 4] (see StTab)
// TABLES: the cost / predecessor tables are written to global memory.
// RTK >= 0: an instance for batches with at most RTK blocked intervals per layer (layers
// with fewer are padded with intervals that add exact zeros); it holds one relaxation path
// instead of nine, needs fewer registers and runs at a higher occupancy.  RTK < 0: every path.
template <bool TABLES, int RTK, int SPLITK>
__device__ __forceinline__ void st_dp_body(StDev a)
{
    if (!TABLES) a.cost = nullptr, a.pre = nullptr;
    extern __shared__ double smem[];
    const int b = a.order ? a.order[blockIdx.x] : blockIdx.x;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int N = a.n_s_max, n_t = a.n_t, R_max = a.r_max;
    const int n_s = a.n_s[b];
    if (n_s <= 0 || n_s > N) { // no grid for this scenario: no path
        for (int i = tid; i < n_t; i += nt) a.path_s_ind[(size_t)b * n_t + i] = -1;
        if (tid == 0) a.path_len[b] = 0;
        return;
    }
    const int wave = tid / MP_WAVE, nw = nt / MP_WAVE;
    StSh sh;
    sh.fs = smem;
    sh.ch_prev = sh.fs + 2 * N;
    sh.ch_cur = sh.ch_prev + 2 * N;
    const int W = a.tab_half;
    sh.half = W;
    sh.acc = sh.ch_cur + 2 * N;        // index n + W, n in [-W, W]
    sh.jerk2 = sh.acc + 2 * (W + 1);   // same index, layer i == 2
    sh.jerk3 = sh.jerk2 + 2 * (W + 1); // index m + 2W, layers i >= 3
    // blocked intervals of every time layer, r_pitch slots each, staged once at the start; a layer
    // with more intervals than that (it takes the thread-per-cell relaxation anyway) reads them
    // from global memory
    const int pitch = a.r_pitch;
    const int R_slots = pitch * n_t;
    double *down_all = sh.jerk3 + 4 * (W + 1), *up_all = down_all + R_slots;
    sh.down = down_all;
    sh.up = up_all;
    double *last_col = up_all + R_slots;
    double *red = last_col + n_t; // [2][8]: the layer minimum per wave, buffers alternating by layer
    sh.pre16 = (short *)(red + 16);
    int *d_lim_sh = (int *)(sh.pre16 + (size_t)n_t * N + (((size_t)n_t * N) & 1));
    StTab tb; // placed by offset from smem so the pointers stay LDS pointers
    tb.rec = smem + st_tab_offset(N, R_slots, n_t, W) / 8;
    tb.r_tab = a.r_tab;
    tb.stride = st_rec_stride(a.r_tab);

    const double ds = a.ds, dt = a.dt, lim = a.speed_lim;
    const double v0 = a.cur_vel[b];
    const double dt2 = a.dt2, dt3 = a.dt3;
    double *cost_g = a.cost ? a.cost + (size_t)b * n_t * N : nullptr;
    int32_t *pre_g = a.pre ? a.pre + (size_t)b * n_t * N : nullptr;

    if (tid == 0) *d_lim_sh = n_s; // first d whose speed exceeds the limit
    __syncthreads();
    // tables (speed_planner.py:120-136, 151-160): terms that depend only on an
    // integer difference, evaluated with the reference's operations
    for (int d = tid; d < n_s; d += nt) {
        double v = __ddiv_rn(__dmul_rn(__dmul_rn(1.0, (double)d), ds), dt);
        sh.front(d) = __dmul_rn(4.5, v);
        double sc;
        if (v > lim) {
            sc = __dmul_rn(__dmul_rn(100.0, __dmul_rn(v, v)), dt);
            double vm = d > 0 ? __ddiv_rn(__dmul_rn(__dmul_rn(1.0, (double)(d - 1)), ds), dt) : 0.0;
            if (d == 0 || !(vm > lim)) *d_lim_sh = d;
        } else {
            double dv = __ddiv_rn(__dsub_rn(v, lim), lim);
            sc = __dmul_rn(__dmul_rn(-100.0, dv), dt);
        }
        sh.speed(d) = sc;
    }
    const int h_tab = min(n_s - 1, W); // differences this grid can ask the tables for
    for (int q = tid; q < 2 * h_tab + 1; q += nt) {
        const int n = q - h_tab;
        double acv;
        sh.acc[n + W] = st_acc_term(a, n, &acv);
        sh.jerk2[n + W] = st_jerk2_term(a, acv);
    }
    for (int q = tid; q < 4 * h_tab + 1; q += nt) {
        const int m = q - 2 * h_tab;
        sh.jerk3[m + 2 * W] = st_jerk3_term(a, m);
    }
    // row 0: cost 0, pre_ind -1 (st_map.py:86-91)
    for (int j = tid; j < N; j += nt) {
        sh.cost_prev(j) = 0.0;
        sh.pre16[j] = -1;
        if (cost_g) cost_g[j] = 0.0;
        if (pre_g) pre_g[j] = -1;
    }
    if (tid == 0) last_col[0] = 0.0;
    // Two sweeps (see below) pay when the first covers less than half of the grid; a narrower grid
    // is planned in one (n_one: up to twice the first sweep's cells, as far as the records, running
    // on into the room of the kept rows, and the tables go).
    const int kA = a.k_prefix + 1; // cells of the first sweep
    bool two_sweeps = false;
    if constexpr (!TABLES)
        if (a.k_prefix > 0) {
            two_sweeps = n_s > a.n_one;
        }
    // quiet and flag word of every cell record at their reset values (whoever reads them puts them back)
    for (int j = tid; j < (two_sweeps ? kA : n_s); j += nt) {
        tb.quiet(j)[0] = 0x7fffffff;
        tb.quiet(j)[1] = 0;
    }
    for (int q = tid; q < (n_t - 1) * pitch; q += nt) {
        const int i = 1 + q / pitch, k = q - (i - 1) * pitch;
        const int R_i = min(max(a.n_reg[(size_t)b * n_t + i], 0), R_max);
        if (R_i <= pitch && k < R_i) {
            down_all[i * pitch + k] = a.s_down[((size_t)b * n_t + i) * R_max + k];
            up_all[i * pitch + k] = a.s_up[((size_t)b * n_t + i) * R_max + k];
        }
    }
    __syncthreads();
    const int d_lim = *d_lim_sh;

    // One sweep over the time layers for the cells rg.j_lo..rg.j_hi.  Cell j only looks at cells
    // 0..j of the layer before, so the cells 0..k can be planned through all layers on their own
    // (first sweep: their costs of every layer are kept), and the cheapest of
    // them in the last layer bounds the plan's final cost from above: the end state is the
    // cheaper of the cheapest cell of the last layer and the cheapest last cell of any layer, so
    // it costs at most u.  Costs never decrease along a chain of predecessors (every term of an
    // edge is >= 0 and rounding is monotone), so a cell that costs more than u is neither on the
    // plan nor the predecessor that decides any cell of it; the second sweep (cells k+1.., with
    // row i-1 of the first sweep's cells copied in beside its own) therefore stops a walk as soon
    // as everything left of it is certain to cost more than min(lane's best, u).  Cells that
    // cost at most u come out exactly as without the bound (costs and predecessors), cells
    // above it with some value above it, and the end state and its chain are the same.  Most of a
    // grid is such cells (far ahead, behind obstacles), and their walks were the long ones.
    auto sweep = [&](const StRange &rg0, auto capped_tag, bool second, double *keep_c, int kA,
                     double *m_keep, int *reach) -> double {
        constexpr bool CAPPED = decltype(capped_tag)::value;
        double m_prev = 0.0; // minimum of cost_prev
        // the interval count of a layer is fetched while the layer before it is relaxed
        int R_next = min(max(a.n_reg[(size_t)b * n_t + 1], 0), R_max);
        for (int i = 1; i < n_t; ++i) {
            const int R = R_next;
            const bool staged = R <= pitch; // else: in global memory
            sh.down = down_all + i * pitch;
            sh.up = up_all + i * pitch;
            const double *g_down = a.s_down + ((size_t)b * n_t + i) * R_max, *g_up = a.s_up + ((size_t)b * n_t + i) * R_max;
            if (i + 1 < n_t) R_next = min(max(a.n_reg[(size_t)b * n_t + i + 1], 0), R_max);
            StRange rg = rg0;
            if (keep_c && !second) { // first sweep: this layer's row of the kept costs
                rg.keep_cost = keep_c + (size_t)i * kA;
            }
            double m_loc = MP_INF;
            if (i == 1) {
                // only p_j = 0 (speed_planner.py:96-100)
                for (int j = rg.j_lo + tid; j <= rg.j_hi; j += nt) {
                    if constexpr (CAPPED)
                        if (j >= d_lim && sh.speed(j) > rg.u_cap) {
                            // (the speed term of the one edge into the cell is above the bound already)
                            sh.cost_cur(j) = MP_INF;
                            sh.pre16[(size_t)i * N + j] = 0;
                            st_store_hist(sh, i, N, j, 0);
                            continue;
                        }
                    const double s = __dmul_rn((double)j, ds);
                    double v = __ddiv_rn(__dmul_rn(__dmul_rn(1.0, (double)j), ds), dt);
                    double ac = __ddiv_rn(__dsub_rn(v, v0), dt);
                    double jk = __ddiv_rn(ac, dt);
                    StCellLds cell;
                    double ob;
                    if (staged) {
                        cell.init(s, sh.down, sh.up, R);
                        ob = cell.ob(s, sh.front(j));
                    } else {
                        cell.init(s, g_down, g_up, R);
                        ob = cell.ob(s, sh.front(j));
                    }
                    double best = __dadd_rn(__dadd_rn(__dadd_rn(ob, sh.speed(j)), __dmul_rn(__dmul_rn(ac, ac), dt)),
                                            __dmul_rn(__dmul_rn(jk, jk), dt));
                    sh.cost_cur(j) = best;
                    sh.pre16[(size_t)i * N + j] = 0;
                    st_store_hist(sh, i, N, j, 0);
                    if (rg.keep_cost) rg.keep_cost[j] = best;
                    if (cost_g) cost_g[(size_t)i * N + j] = best;
                    if (pre_g) pre_g[(size_t)i * N + j] = 0;
                    if constexpr (CAPPED)
                        if (best <= rg.u_cap) atomicMax(reach + i, j);
                    m_loc = best < m_loc ? best : m_loc;
                }
            } else if constexpr (CAPPED) {
                bool done = false;
                if constexpr (RTK >= 0)
                    if (R <= RTK && staged) {
                        m_loc = st_relax_layer_capped<StCell<RTK>>(a, sh, rg, sh.down, sh.up, i, N, R, d_lim, m_prev, reach);
                        done = true;
                    }
                if (!done) {
                    if (staged) m_loc = st_relax_layer_capped<StCellLds>(a, sh, rg, sh.down, sh.up, i, N, R, d_lim, m_prev, reach);
                    else m_loc = st_relax_layer_capped<StCellLds>(a, sh, rg, g_down, g_up, i, N, R, d_lim, m_prev, reach);
                }
            } else if (RTK >= 0) {
                if constexpr (RTK >= 0) {
                    if (R <= RTK && RTK <= a.r_tab && staged)
                        m_loc = st_relax_layer_split<RTK, SPLITK>(a, sh, tb, rg, i, n_s, N, R, d_lim, m_prev, cost_g, pre_g);
                    else if (staged)
                        m_loc = st_relax_layer<StCellLds>(a, sh, rg, sh.down, sh.up, i, n_s, N, R, d_lim, m_prev, cost_g, pre_g);
                    else
                        m_loc = st_relax_layer<StCellLds>(a, sh, rg, g_down, g_up, i, n_s, N, R, d_lim, m_prev, cost_g, pre_g);
                }
            } else if (R <= 8 && (R == 7 ? 8 : R) <= a.r_tab && staged) { // 7 intervals run as 8 with one padded slot
#define ST_SPLIT_CASE(RT)                                                                                   \
    m_loc = a.lanes_per_cell == 4 ? st_relax_layer_split<RT, 4>(a, sh, tb, rg, i, n_s, N, R, d_lim, m_prev, cost_g, pre_g)   \
                    : st_relax_layer_split<RT, 8>(a, sh, tb, rg, i, n_s, N, R, d_lim, m_prev, cost_g, pre_g)
                switch (R) {
                case 0: ST_SPLIT_CASE(0); break;
                case 1: ST_SPLIT_CASE(1); break;
                case 2: ST_SPLIT_CASE(2); break;
                case 3: ST_SPLIT_CASE(3); break;
                case 4: ST_SPLIT_CASE(4); break;
                case 5: ST_SPLIT_CASE(5); break;
                case 6: ST_SPLIT_CASE(6); break;
                default: ST_SPLIT_CASE(8); break;
                }
#undef ST_SPLIT_CASE
            } else if (!staged) {
                m_loc = st_relax_layer<StCellLds>(a, sh, rg, g_down, g_up, i, n_s, N, R, d_lim, m_prev, cost_g, pre_g);
            } else {
                switch (R) {
                case 0: m_loc = st_relax_layer<StCell<0>>(a, sh, rg, sh.down, sh.up, i, n_s, N, R, d_lim, m_prev, cost_g, pre_g); break;
                case 1: m_loc = st_relax_layer<StCell<1>>(a, sh, rg, sh.down, sh.up, i, n_s, N, R, d_lim, m_prev, cost_g, pre_g); break;
                case 2: m_loc = st_relax_layer<StCell<2>>(a, sh, rg, sh.down, sh.up, i, n_s, N, R, d_lim, m_prev, cost_g, pre_g); break;
                case 3: m_loc = st_relax_layer<StCell<3>>(a, sh, rg, sh.down, sh.up, i, n_s, N, R, d_lim, m_prev, cost_g, pre_g); break;
                case 4: m_loc = st_relax_layer<StCell<4>>(a, sh, rg, sh.down, sh.up, i, n_s, N, R, d_lim, m_prev, cost_g, pre_g); break;
                case 5: m_loc = st_relax_layer<StCell<5>>(a, sh, rg, sh.down, sh.up, i, n_s, N, R, d_lim, m_prev, cost_g, pre_g); break;
                case 6: m_loc = st_relax_layer<StCell<6>>(a, sh, rg, sh.down, sh.up, i, n_s, N, R, d_lim, m_prev, cost_g, pre_g); break;
                case 7:
                case 8: m_loc = st_relax_layer<StCell<8>>(a, sh, rg, sh.down, sh.up, i, n_s, N, R, d_lim, m_prev, cost_g, pre_g); break;
                default: m_loc = st_relax_layer<StCellLds>(a, sh, rg, sh.down, sh.up, i, n_s, N, R, d_lim, m_prev, cost_g, pre_g); break;
                }
            }
            // padding cells keep the initial values
            for (int j = n_s + tid; j < N; j += nt) {
                if (cost_g) cost_g[(size_t)i * N + j] = 0.0;
                if (pre_g) pre_g[(size_t)i * N + j] = -1;
            }
            // second sweep: the first sweep's cells of this layer beside the cells just relaxed, so
            // that the row is whole for the next layer (and, after the last one, for the end state)
            if (second)
                for (int j = tid; j < kA; j += nt) {
                    sh.cost_cur(j) = keep_c[(size_t)i * kA + j];
                    st_store_hist(sh, i, N, j, sh.pre16[(size_t)i * N + j]); // (the word it had in the first sweep)
                }
            // (a layer that takes the thread-per-cell relaxation has no barrier of its own: a wave
            // may be writing the next layer's minimum while another still reads this one's, hence
            // the two buffers)
            double *red_i = red + 8 * (i & 1);
            m_loc = wave_min_f64(m_loc);
            if ((tid & (MP_WAVE - 1)) == 0) red_i[wave] = m_loc;
            __syncthreads();
            if (tid == 0 && rg.j_hi == n_s - 1) last_col[i] = sh.cost_cur(n_s - 1);
            m_prev = red_i[0];
            for (int w = 1; w < nw; ++w) m_prev = red_i[w] < m_prev ? red_i[w] : m_prev;
            if (m_keep) {
                // the whole row's minimum: the first sweep leaves its own per layer, the second joins them
                if (second) m_prev = m_keep[i] < m_prev ? m_keep[i] : m_prev;
                else if (tid == 0) m_keep[i] = m_prev;
            }
            // the two cost rows swap roles: the next layer writes the row this one read
            double *row = sh.ch_prev;
            sh.ch_prev = sh.ch_cur;
            sh.ch_cur = row;
        }
        return m_prev;
    };

    const StRange whole{n_s, 0, n_s - 1, MP_INF, nullptr};
    if constexpr (!TABLES) if (two_sweeps) {
        // (kept rows: behind the cell records; see st_keep_offset)
        double *keep_c = smem + st_keep_offset(N, R_slots, n_t, a.r_tab, a.rec_cells, W) / 8;
        double *m_keep = keep_c + (size_t)n_t * kA;
        int *reach = (int *)(m_keep + n_t);
        const StRange first{n_s, 0, kA - 1, MP_INF, nullptr};
        const double u = sweep(first, std::false_type(), false, keep_c, kA, m_keep, nullptr);
        // per layer, the last of the first sweep's cells within the bound (the second sweep's cells join in)
        for (int i = 1 + wave; i < n_t; i += nw) {
            int jm = -1;
            for (int j0 = 0; j0 < kA; j0 += MP_WAVE) {
                const int j = j0 + (tid & (MP_WAVE - 1));
                const unsigned long long m = __ballot(j < kA && keep_c[(size_t)i * kA + j] <= u);
                if (m) jm = j0 + 63 - __clzll((long long)m);
            }
            if ((tid & (MP_WAVE - 1)) == 0) reach[i] = jm;
        }
        __syncthreads(); // the first sweep is done with the rows and the minimum buffers
        const StRange rest{n_s, kA, n_s - 1, u, nullptr};
        sweep(rest, std::true_type(), true, keep_c, kA, m_keep, reach);
    }
    if (!two_sweeps) sweep(whole, std::false_type(), false, nullptr, 0, nullptr, nullptr);
    __syncthreads(); // last_col complete
    // end state and backtrack (speed_planner.py:63-89)
    if (tid < MP_WAVE) {
        // cheapest cell of the last layer, the first of equals: by one wave
        double min_j = MP_INF;
        int arg_j = 0x7fffffff;
        for (int j = tid; j < n_s; j += MP_WAVE) {
            const double c = sh.cost_prev(j);
            if (c < min_j || arg_j == 0x7fffffff) { min_j = c; arg_j = j; }
        }
        group_argmin<MP_WAVE>(min_j, arg_j);
        if (tid == 0) {
        double min_i = last_col[1];
        int arg_i = 1;
        for (int i = 2; i < n_t; ++i)
            if (last_col[i] < min_i) { min_i = last_col[i]; arg_i = i; }
        int end_i, end_j;
        if (min_i <= min_j) { end_i = arg_i; end_j = n_s - 1; }
        else { end_i = n_t - 1; end_j = arg_j; }
        int32_t *out = a.path_s_ind + (size_t)b * n_t;
        for (int i = end_i + 1; i < n_t; ++i) out[i] = -1;
        int j = end_j;
        out[end_i] = end_j;
        for (int i = end_i; i >= 1; --i) {
            j = sh.pre16[(size_t)i * N + j];
            out[i - 1] = j;
        }
        a.path_len[b] = end_i + 1;
        }
    }
}

template <bool TABLES>
__global__ void st_dp_kernel(StDev a)
{
    st_dp_body<TABLES, -1, 0>(a);
}

// The kernel that holds every path, without the tables: 101 registers would leave it at four
// waves per SIMD; at five (96 registers, 12 B of scratch) the whole replan's speed search, whose
// workgroups mostly wait for LDS round trips and barriers, takes 0.115 instead of 0.134 ms per
// 4096 scenes (six waves: 56 B of scratch, 0.119; measured while it still planned in one sweep).
__attribute__((amdgpu_waves_per_eu(5, 5))) __global__ void st_dp_plain_kernel(StDev a)
{
    st_dp_body<false, -1, 0>(a);
}

#ifndef ST_WAVES_PER_EU
#define ST_WAVES_PER_EU 6
#endif
template <int RTK, int SPLITK> // SPLITK lanes per cell (st_lanes_per_cell)
__attribute__((amdgpu_waves_per_eu(ST_WAVES_PER_EU, ST_WAVES_PER_EU))) __global__ void st_dp_r_kernel(StDev a)
{
    st_dp_body<false, RTK, SPLITK>(a);
}

// ---------------------------------------------------------------------------
// ST blocked intervals
// ---------------------------------------------------------------------------

// Smallest double x with sqrt(x) >= r under round-to-nearest: then
// "sqrt(d2) < r" (the reference's test, Utils/tool.py:58-59 + st_map.py:51-52)
// is exactly "d2 < x", and the scan needs no square roots.
__device__ __forceinline__ double sqrt_threshold(double r)
{
    if (!(r > 0.0)) return 0.0; // sqrt(d2) < r never holds for r <= 0
    double x = __dmul_rn(r, r);
    for (int it = 0; it < 8 && __dsqrt_rn(x) >= r; ++it) x = nextafter(x, 0.0);
    for (int it = 0; it < 16 && __dsqrt_rn(x) < r; ++it) x = nextafter(x, MP_INF);
    return x;
}

struct RegDev {
    int32_t n_t, p_max, n_st_max, n_dy_max, r_max;
    int32_t dy_fstride; // frames stored per moving obstacle in dy_s / dy_l (>= n_t)
    const double *path_s, *path_l;
    const int32_t *n_path;
    const double *st_s, *st_l, *st_r;
    const int32_t *n_st;
    const double *dy_s, *dy_l, *dy_r;
    const int32_t *n_dy;
    double *s_down, *s_up;
    int32_t *n_reg, *status;
    // whole replan only (null otherwise): the grid width of every scene and where the scene's
    // weight bin for the ordered speed search goes (st_weight_kernel's value, from the counts
    // this kernel has in hand)
    const int32_t *key_n_s;
    int32_t *key_out;
};

__device__ __forceinline__ int st_weight_bin(int n_s, int regs, int n_t)
{
    const double n = (double)max(n_s, 0);
    return mp_weight_bin(n * n * (4.0 + (double)regs / (double)(n_t - 1)));
}

// One workgroup per scenario; the sampled path sits in LDS.  For every time
// layer the obstacles are visited in the reference's order (statics, then
// dynamics) in chunks of blockDim threads, one obstacle per thread scanning
// the path; hits are appended in obstacle order with a ballot prefix count.
__global__ void st_regions_kernel(RegDev a)
{
    extern __shared__ double smem[];
    __shared__ int wave_cnt[16];
    __shared__ int base_sh;
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & (MP_WAVE - 1), wave = tid / MP_WAVE, nw = nt / MP_WAVE;
    // counts clamped to the slots the arrays have
    const int P = min(max(a.n_path[b], 0), a.p_max);
    const int n_st = a.n_st ? min(max(a.n_st[b], 0), a.n_st_max) : 0, n_dy = a.n_dy ? min(max(a.n_dy[b], 0), a.n_dy_max) : 0;
    double *ps = smem, *pl = smem + a.p_max;
    for (int k = tid; k < P; k += nt) {
        ps[k] = a.path_s[(size_t)b * a.p_max + k];
        pl[k] = a.path_l[(size_t)b * a.p_max + k];
    }
    __syncthreads();
    int overflow = 0, regs_sum = 0;
    const int n_ob = n_st + n_dy;
    for (int i = 0; i < a.n_t; ++i) {
        if (tid == 0) base_sh = 0;
        __syncthreads();
        for (int o0 = 0; o0 < n_ob; o0 += nt) {
            const int o = o0 + tid;
            bool hit = false;
            double dn = 0.0, u = 0.0;
            if (o < n_ob) {
                double ox, oy, rad;
                if (o < n_st) {
                    ox = a.st_s[(size_t)b * a.n_st_max + o];
                    oy = a.st_l[(size_t)b * a.n_st_max + o];
                    rad = a.st_r[(size_t)b * a.n_st_max + o];
                } else {
                    const size_t q = (size_t)b * a.n_dy_max + (o - n_st);
                    ox = a.dy_s[q * a.dy_fstride + i];
                    oy = a.dy_l[q * a.dy_fstride + i];
                    rad = a.dy_r[q];
                }
                const double thr = sqrt_threshold(rad);
                for (int k = 0; k < P; ++k) {
                    double dx = __dsub_rn(ox, ps[k]), dy = __dsub_rn(oy, pl[k]);
                    double d2 = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
                    if (d2 < thr) {
                        if (!hit) { dn = ps[k]; hit = true; }
                        u = ps[k];
                    }
                }
            }
            const unsigned long long bal = __ballot(hit);
            if (lane == 0) wave_cnt[wave] = __popcll(bal);
            __syncthreads();
            int pos = base_sh;
            for (int w = 0; w < wave; ++w) pos += wave_cnt[w];
            pos += __popcll(bal & ((1ull << lane) - 1ull));
            if (hit) {
                if (pos < a.r_max) {
                    a.s_down[((size_t)b * a.n_t + i) * a.r_max + pos] = dn;
                    a.s_up[((size_t)b * a.n_t + i) * a.r_max + pos] = u;
                } else overflow = 1;
            }
            __syncthreads();
            if (tid == 0) {
                int tot = 0;
                for (int w = 0; w < nw; ++w) tot += wave_cnt[w];
                base_sh += tot;
            }
            __syncthreads();
        }
        if (tid == 0) {
            a.n_reg[(size_t)b * a.n_t + i] = min(base_sh, a.r_max);
            if (i >= 1) regs_sum += min(base_sh, a.r_max);
        }
        __syncthreads();
    }
    overflow = __syncthreads_or(overflow);
    if (tid == 0) {
        a.status[b] = overflow ? MP_ST_E_CAPACITY : MP_ST_OK;
        if (a.key_out) a.key_out[b] = st_weight_bin(a.key_n_s[b], regs_sum, a.n_t);
    }
}


// Scan of the path for one obstacle: arc length of the first and of the last
// path point closer than the radius (st_map.py:48-60).
__device__ __forceinline__ bool scan_path(const double *ps, const double *pl, int P, double ox, double oy,
                                          double thr, double *dn_out, double *u_out)
{
    bool hit = false;
    double dn = 0.0, u = 0.0;
    for (int k = 0; k < P; ++k) {
        double dx = __dsub_rn(ox, ps[k]), dy = __dsub_rn(oy, pl[k]);
        double d2 = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
        if (d2 < thr) {
            if (!hit) { dn = ps[k]; hit = true; }
            u = ps[k];
        }
    }
    *dn_out = dn;
    *u_out = u;
    return hit;
}

// Same result as st_regions_kernel with the scans done once.  A static obstacle
// blocks the same interval at every time layer, so it is scanned once instead of
// n_t times; (obstacle, layer) items whose circle cannot reach the path's
// bounding box are dropped before any scan; the surviving items are gathered in
// a work list so that every lane scans; the per-layer lists are then written in
// the reference's order from the stored results.  Items: q < n_st the statics,
// then q = n_st + o*n_t + i for frame i of moving obstacle o.
// Dynamic LDS: ps[p_max] pl[p_max] | res_dn[n_items_max] res_u[n_items_max] |
//   work[n_items_max] u16 | flag[n_items_max] u8
__global__ void st_regions_once_kernel(RegDev a, int n_items_max)
{
    extern __shared__ double smem[];
    __shared__ int n_work_sh, regs_sh;
    __shared__ double box_sh[4][16];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & (MP_WAVE - 1), wave = tid / MP_WAVE, nw = nt / MP_WAVE;
    // counts clamped to the slots the arrays have
    const int P = min(max(a.n_path[b], 0), a.p_max);
    const int n_st = a.n_st ? min(max(a.n_st[b], 0), a.n_st_max) : 0, n_dy = a.n_dy ? min(max(a.n_dy[b], 0), a.n_dy_max) : 0;
    const int n_t = a.n_t;
    double *ps = smem, *pl = smem + a.p_max;
    double *res_dn = pl + a.p_max, *res_u = res_dn + n_items_max;
    unsigned short *work = (unsigned short *)(res_u + n_items_max);
    unsigned char *flag = (unsigned char *)(work + n_items_max + (n_items_max & 1));
    // path into LDS and its bounding box
    double s_lo = MP_INF, s_hi = -MP_INF, l_lo = MP_INF, l_hi = -MP_INF;
    for (int k = tid; k < P; k += nt) {
        const double s = a.path_s[(size_t)b * a.p_max + k], l = a.path_l[(size_t)b * a.p_max + k];
        ps[k] = s;
        pl[k] = l;
        s_lo = s < s_lo ? s : s_lo; s_hi = s > s_hi ? s : s_hi;
        l_lo = l < l_lo ? l : l_lo; l_hi = l > l_hi ? l : l_hi;
    }
    s_lo = wave_min_f64(s_lo); s_hi = -wave_min_f64(-s_hi);
    l_lo = wave_min_f64(l_lo); l_hi = -wave_min_f64(-l_hi);
    if (lane == 0) { box_sh[0][wave] = s_lo; box_sh[1][wave] = s_hi; box_sh[2][wave] = l_lo; box_sh[3][wave] = l_hi; }
    if (tid == 0) n_work_sh = regs_sh = 0;
    __syncthreads();
    for (int w = 0; w < nw; ++w) {
        s_lo = box_sh[0][w] < s_lo ? box_sh[0][w] : s_lo; s_hi = box_sh[1][w] > s_hi ? box_sh[1][w] : s_hi;
        l_lo = box_sh[2][w] < l_lo ? box_sh[2][w] : l_lo; l_hi = box_sh[3][w] > l_hi ? box_sh[3][w] : l_hi;
    }
    const int n_items = n_st + n_dy * n_t;
    auto item = [&](int q, double *ox, double *oy, double *rad) {
        if (q < n_st) {
            *ox = a.st_s[(size_t)b * a.n_st_max + q];
            *oy = a.st_l[(size_t)b * a.n_st_max + q];
            *rad = a.st_r[(size_t)b * a.n_st_max + q];
        } else {
            const int o = (q - n_st) / n_t, i = (q - n_st) - o * n_t;
            const size_t g = (size_t)b * a.n_dy_max + o;
            *ox = a.dy_s[g * a.dy_fstride + i];
            *oy = a.dy_l[g * a.dy_fstride + i];
            *rad = a.dy_r[g];
        }
    };
    // items that can reach the path's bounding box (margin: the radius plus slack
    // well above the rounding of the distance test)
    for (int q = tid; q < n_items; q += nt) {
        double ox, oy, rad;
        item(q, &ox, &oy, &rad);
        flag[q] = 0;
        const double m = fabs(rad) * (1.0 + 1e-9) + 1e-9;
        const bool can = P > 0 && ox >= s_lo - m && ox <= s_hi + m && oy >= l_lo - m && oy <= l_hi + m;
        if (can) work[atomicAdd(&n_work_sh, 1)] = (unsigned short)q;
    }
    __syncthreads();
    const int n_work = n_work_sh;
    for (int w = tid; w < n_work; w += nt) {
        const int q = work[w];
        double ox, oy, rad, dn, u;
        item(q, &ox, &oy, &rad);
        const bool hit = scan_path(ps, pl, P, ox, oy, sqrt_threshold(rad), &dn, &u);
        flag[q] = hit;
        res_dn[q] = dn;
        res_u[q] = u;
    }
    __syncthreads();
    // per-layer lists in the reference's order: statics, then moving obstacles.  A wave takes
    // whole layers (a list is an ordered compaction of the layer's flags: ballots and a
    // running count in the wave's own registers), so the waves do not meet until the end.
    int overflow = 0, regs_loc = 0;
    const int n_ob = n_st + n_dy;
    for (int i = wave; i < n_t; i += nw) {
        int base = 0;
        for (int o0 = 0; o0 < n_ob; o0 += MP_WAVE) {
            const int o = o0 + lane;
            const int q = o < n_st ? o : n_st + (o - n_st) * n_t + i;
            const bool hit = o < n_ob && flag[q];
            const unsigned long long bal = __ballot(hit);
            const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
            if (hit) {
                if (pos < a.r_max) {
                    a.s_down[((size_t)b * n_t + i) * a.r_max + pos] = res_dn[q];
                    a.s_up[((size_t)b * n_t + i) * a.r_max + pos] = res_u[q];
                } else overflow = 1;
            }
            base += __popcll(bal);
        }
        if (lane == 0) a.n_reg[(size_t)b * n_t + i] = min(base, a.r_max);
        if (i >= 1) regs_loc += min(base, a.r_max);
    }
    if (lane == 0 && regs_loc) atomicAdd(&regs_sh, regs_loc);
    overflow = __syncthreads_or(overflow);
    if (tid == 0) {
        a.status[b] = overflow ? MP_ST_E_CAPACITY : MP_ST_OK;
        if (a.key_out) a.key_out[b] = st_weight_bin(a.key_n_s[b], regs_sh, n_t);
    }
}

// A plan's time follows the square of its grid width (and its intervals): one thread per
// scenario bins that, for batches whose grids differ (mp_st_dp_ordered).
__global__ void st_weight_kernel(int B, int n_t, const int32_t *n_s, const int32_t *n_reg, int32_t *key)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int regs = 0;
    for (int i = 1; i < n_t; ++i) regs += max(n_reg[(size_t)b * n_t + i], 0);
    key[b] = st_weight_bin(n_s[b], regs, n_t);
}

// ---------------------------------------------------------------------------
extern void mp_set_error(const char *fmt, ...);

static int st_launch_check(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        mp_set_error("%s: %s", what, hipGetErrorString(e));
        return MP_E_LAUNCH;
    }
    return MP_OK;
}

static int st_dp_launch(int32_t B, int32_t n_t, int32_t n_s_max, const int32_t *n_s, const double *s_down,
                        const double *s_up, const int32_t *n_reg, int32_t r_max, const double *cur_vel,
                        double speed_lim, double ds, double dt, double *cost, int32_t *pre,
                        int32_t *path_s_ind, int32_t *path_len, int32_t *order_ws, void *stream,
                        bool keys_ready = false);

extern "C" int mp_st_dp(int32_t B, int32_t n_t, int32_t n_s_max, const int32_t *n_s, const double *s_down,
                        const double *s_up, const int32_t *n_reg, int32_t r_max, const double *cur_vel,
                        double speed_lim, double ds, double dt, double *cost, int32_t *pre,
                        int32_t *path_s_ind, int32_t *path_len, void *stream)
{
    return st_dp_launch(B, n_t, n_s_max, n_s, s_down, s_up, n_reg, r_max, cur_vel, speed_lim, ds, dt, cost, pre,
                        path_s_ind, path_len, nullptr, stream);
}

extern "C" int mp_st_dp_ordered(int32_t B, int32_t n_t, int32_t n_s_max, const int32_t *n_s, const double *s_down,
                                const double *s_up, const int32_t *n_reg, int32_t r_max, const double *cur_vel,
                                double speed_lim, double ds, double dt, double *cost, int32_t *pre,
                                int32_t *path_s_ind, int32_t *path_len, int32_t *order_ws, void *stream)
{
    return st_dp_launch(B, n_t, n_s_max, n_s, s_down, s_up, n_reg, r_max, cur_vel, speed_lim, ds, dt, cost, pre,
                        path_s_ind, path_len, order_ws, stream);
}

// mp_st_dp_ordered for the whole replan, whose interval kernel has already left every scene's
// weight bin in order_ws[0..B) (RegDev.key_out).
int mp_st_dp_keyed(int32_t B, int32_t n_t, int32_t n_s_max, const int32_t *n_s, const double *s_down,
                   const double *s_up, const int32_t *n_reg, int32_t r_max, const double *cur_vel,
                   double speed_lim, double ds, double dt, double *cost, int32_t *pre, int32_t *path_s_ind,
                   int32_t *path_len, int32_t *order_ws, void *stream)
{
    return st_dp_launch(B, n_t, n_s_max, n_s, s_down, s_up, n_reg, r_max, cur_vel, speed_lim, ds, dt, cost, pre,
                        path_s_ind, path_len, order_ws, stream, true);
}

static int st_dp_launch(int32_t B, int32_t n_t, int32_t n_s_max, const int32_t *n_s, const double *s_down,
                        const double *s_up, const int32_t *n_reg, int32_t r_max, const double *cur_vel,
                        double speed_lim, double ds, double dt, double *cost, int32_t *pre,
                        int32_t *path_s_ind, int32_t *path_len, int32_t *order_ws, void *stream, bool keys_ready)
{
    if (B <= 0 || n_t < 2 || n_s_max <= 0 || r_max < 0) { mp_set_error("bad B/n_t/n_s_max/r_max"); return MP_E_BADARG; }
    if (!n_s || !n_reg || !cur_vel || !path_s_ind || !path_len || (r_max > 0 && (!s_down || !s_up))) {
        mp_set_error("null array");
        return MP_E_BADARG;
    }
    if (!(ds > 0) || !(dt > 0) || !(speed_lim > 0)) { mp_set_error("ds, dt, speed_lim must be positive"); return MP_E_BADARG; }
    // n_s_max: the history word holds 8*(pp-2p) and 8*(3pp-ppp-3p) as signed 16-bit halves and
    // fire distances are u16 values of 16*d, so 24*n_s_max must stay below 32768 whatever LDS holds
    static_assert(24 * 1024 < 32768, "history word / fire distance encoding");
    if (n_t > 64 || n_s_max > 1024 || r_max > 256) {
        mp_set_error("mp_st_dp supports n_t <= 64, n_s_max <= 1024, r_max <= 256");
        return MP_E_UNSUPPORTED;
    }
    // cell tables: as many interval slots as fit next to everything else (at most r_max)
    int r_tab = r_max > 0 ? r_max : 1;
    if (r_tab > 8 || r_tab == 7) r_tab = 8; // the split path keeps a cell's intervals in registers (7 run as 8)
    while (r_tab > 0 && st_tab_bytes(n_s_max, r_tab) > 40 * 1024) --r_tab;
    if (mp_knob_st_generic() == 2) r_tab = 0; // tests: no cell tables, every layer with intervals by a thread per cell
    // the intervals of every layer sit in LDS from the start (a layer then begins without a
    // barrier): as many slots per layer as the caller's bound when that is a few kilobytes, else
    // the 8 the lanes-per-cell relaxation takes; a layer with more is read from global memory
    const int r_slots_1 = r_max > 0 ? r_max : 1;
    const int r_pitch = (size_t)2 * r_slots_1 * n_t * sizeof(double) <= 6 * 1024 ? r_slots_1 : 8;
    int rk = r_max == 7 ? 8 : r_max;
    if (!(B >= 2048 && rk <= 8 && rk <= r_tab && mp_knob_st_generic() <= 0) || cost || pre) rk = -1;
    // Large batches (throughput; a single plan would only gain barriers) plan the nearest cells
    // first (see st_dp_body): cells 0..k, k = half of what a vehicle reaches that never exceeds
    // the speed limit ((cells per layer at the first speed above the limit) x (layers) / 2).  Any
    // k gives the same plans; a small one makes the first sweep cheap and its bound loose (on the
    // benchmark's 97-cell grids, 4096 plans: k = 40 / 30 / 20 / 10 -> 0.226 / 0.218 / 0.207 /
    // 0.220 ms).  Not when the maps are wanted: those hold every cell's exact cost.
    int k_prefix = 0;
    if (B > 512 && !cost && !pre && mp_knob_st_generic() != 2 && n_t >= 3) {
        int d_over = 0;
        while (d_over < n_s_max && !(1.0 * d_over * ds / dt > speed_lim)) ++d_over;
        int k = d_over * (n_t - 1) / 2;
        // ... rounded down to whole waves of the first sweep (its lanes are the issue slots a pass
        // costs, filled or not: 16 cells instead of 21 on the benchmark, 0.174 -> 0.162 ms)
        const int per_wave = MP_WAVE / st_lanes_per_cell(B, n_s_max);
        if (k + 1 > per_wave) k = (k + 1) / per_wave * per_wave - 1;
        // (a first sweep over most of the grid would cost what the whole costs, and the second comes
        // on top; the kernel decides per scenario as well, by its own grid width)
        if (2 * (k + 1) > n_s_max) k = 0;
        // (the whole replan's scenes, 0.26 intervals per layer on average, grids of 74..100 cells, on
        // the kernel for every path: ST stage 0.107 ms in one sweep, 0.076 with a first sweep over 16
        // cells, 0.084 over 32, 0.111 over 48)
        if (mp_knob_st_prefix() >= 0) k = mp_knob_st_prefix(); // tests: 0 = one sweep
        if (k > 0 && k + 8 < n_s_max) k_prefix = k;
    }
    // (records for the first sweep's cells only: with 13 KB of LDS a CU holds the twelve workgroups
    // the instances' registers allow)
    const int rec_cells = k_prefix > 0 ? k_prefix + 1 : n_s_max;
    // the widest grid that is still planned in one sweep, and with it how far the tables must go
    // (the second sweep computes what it needs beyond them)
    int n_one = n_s_max, tab_half = n_s_max - 1;
    if (k_prefix > 0) {
        const int room = rec_cells >= n_s_max ? n_s_max
                       : (int)((st_tab_bytes(rec_cells, r_tab) + st_keep_bytes(n_t, k_prefix)) /
                               (st_rec_stride(r_tab) * sizeof(double)));
        n_one = std::min(2 * (k_prefix + 1) - 1, room);
        if (rec_cells < n_s_max) tab_half = std::max(n_one - 1, k_prefix);
    }
    StDev a{n_t, n_s_max, r_slots_1, r_tab, st_lanes_per_cell(B, n_s_max), r_pitch, k_prefix, rec_cells, n_one, tab_half, n_s, s_down, s_up, n_reg, cur_vel, speed_lim, ds, dt,
            pow(dt, 2.0), pow(dt, 3.0), cost, pre, path_s_ind, path_len, nullptr};
    int threads = n_s_max <= 32 ? 64 : (n_s_max <= 64 ? 128 : 256);
    // few scenarios (less than two workgroups per CU): latency counts, not throughput, so a
    // grid gets the lanes that make a layer one pass (8 per cell), up to 512
    if (B <= 512) threads = n_s_max <= 32 ? 256 : 512;
    // with a first sweep a layer's pass is over its cells only (and most of the others are never walked)
    if (k_prefix > 0 && (k_prefix + 1) * a.lanes_per_cell <= 128) threads = std::min(threads, 128);
    if (mp_knob_st_threads() > 0) threads = mp_knob_st_threads(); // development aid
    size_t lds = st_keep_offset(n_s_max, r_pitch * n_t, n_t, r_tab, rec_cells, tab_half) + st_keep_bytes(n_t, k_prefix);
    if (lds > 160 * 1024) { mp_set_error("mp_st_dp: grid too large for LDS"); return MP_E_UNSUPPORTED; }
    const dim3 grid(B), block(threads);
    hipStream_t st = (hipStream_t)stream;
    if (order_ws && B > MP_ORDER_MIN_BATCH && mp_knob_order() != 0) {
        // widest grids first: weights -> order_ws[0..B), order -> order_ws[B..2B)
        if (!keys_ready)
            hipLaunchKernelGGL(st_weight_kernel, dim3((B + 255) / 256), dim3(256), 0, st, B, n_t, n_s, n_reg, order_ws);
        mp_order_launch(B, order_ws, order_ws + B, st);
        a.order = order_ws + B;
    }
    // large batches with few intervals per layer (r_max is the caller's bound) take the
    // instance built for that bound and grid width class; MP_ST_GENERIC=1 keeps them on the
    // kernel that holds every path
#define ST_R_LAUNCH(RT)                                                                      \
    if (a.lanes_per_cell == 4) hipLaunchKernelGGL((st_dp_r_kernel<RT, 4>), grid, block, lds, st, a); \
    else hipLaunchKernelGGL((st_dp_r_kernel<RT, 8>), grid, block, lds, st, a)
    if (cost || pre) hipLaunchKernelGGL(st_dp_kernel<true>, grid, block, lds, st, a);
    else switch (rk) {
        case 0: ST_R_LAUNCH(0); break;
        case 1: ST_R_LAUNCH(1); break;
        case 2: ST_R_LAUNCH(2); break;
        case 3: ST_R_LAUNCH(3); break;
        case 4: ST_R_LAUNCH(4); break;
        case 5: ST_R_LAUNCH(5); break;
        case 6: ST_R_LAUNCH(6); break;
        case 8: ST_R_LAUNCH(8); break;
        default:
            // (a small batch leaves the GPU mostly empty: the build without scratch is the quicker one there)
            if (B > 512) hipLaunchKernelGGL(st_dp_plain_kernel, grid, block, lds, st, a);
            else hipLaunchKernelGGL(st_dp_kernel<false>, grid, block, lds, st, a);
            break;
    }
#undef ST_R_LAUNCH
    return st_launch_check("mp_st_dp");
}

// mp_st_regions with dy_s / dy_l rows of dy_fstride >= n_t frames per obstacle.
int mp_st_regions_strided(int32_t B, int32_t n_t, const double *path_s, const double *path_l,
                          const int32_t *n_path, int32_t p_max, const double *st_s, const double *st_l,
                          const double *st_r, const int32_t *n_st, int32_t n_st_max, const double *dy_s,
                          const double *dy_l, const double *dy_r, const int32_t *n_dy, int32_t n_dy_max,
                          int32_t dy_fstride, int32_t r_max, double *s_down, double *s_up, int32_t *n_reg,
                          int32_t *status, void *stream, const int32_t *key_n_s = nullptr,
                          int32_t *key_out = nullptr);

extern "C" int mp_st_regions(int32_t B, int32_t n_t, const double *path_s, const double *path_l,
                             const int32_t *n_path, int32_t p_max, const double *st_s, const double *st_l,
                             const double *st_r, const int32_t *n_st, int32_t n_st_max, const double *dy_s,
                             const double *dy_l, const double *dy_r, const int32_t *n_dy, int32_t n_dy_max,
                             int32_t r_max, double *s_down, double *s_up, int32_t *n_reg, int32_t *status,
                             void *stream)
{
    return mp_st_regions_strided(B, n_t, path_s, path_l, n_path, p_max, st_s, st_l, st_r, n_st, n_st_max, dy_s,
                                 dy_l, dy_r, n_dy, n_dy_max, n_t, r_max, s_down, s_up, n_reg, status, stream);
}

int mp_st_regions_strided(int32_t B, int32_t n_t, const double *path_s, const double *path_l,
                          const int32_t *n_path, int32_t p_max, const double *st_s, const double *st_l,
                          const double *st_r, const int32_t *n_st, int32_t n_st_max, const double *dy_s,
                          const double *dy_l, const double *dy_r, const int32_t *n_dy, int32_t n_dy_max,
                          int32_t dy_fstride, int32_t r_max, double *s_down, double *s_up, int32_t *n_reg,
                          int32_t *status, void *stream, const int32_t *key_n_s, int32_t *key_out)
{
    if (dy_fstride < n_t) { mp_set_error("fewer stored frames than time layers"); return MP_E_BADARG; }
    if (B <= 0 || n_t <= 0 || p_max <= 0 || r_max <= 0 || n_st_max < 0 || n_dy_max < 0) {
        mp_set_error("bad sizes");
        return MP_E_BADARG;
    }
    if (!path_s || !path_l || !n_path || !s_down || !s_up || !n_reg || !status ||
        (n_st_max > 0 && (!st_s || !st_l || !st_r || !n_st)) ||
        (n_dy_max > 0 && (!dy_s || !dy_l || !dy_r || !n_dy))) {
        mp_set_error("null array");
        return MP_E_BADARG;
    }
    size_t lds = (size_t)2 * p_max * sizeof(double);
    if (lds > 128 * 1024) { mp_set_error("mp_st_regions: p_max too large"); return MP_E_UNSUPPORTED; }
    RegDev a{n_t, p_max, n_st_max, n_dy_max, r_max, dy_fstride, path_s, path_l, n_path, st_s, st_l, st_r,
             n_st_max > 0 ? n_st : nullptr, dy_s, dy_l, dy_r, n_dy_max > 0 ? n_dy : nullptr,
             s_down, s_up, n_reg, status, key_n_s, key_n_s ? key_out : nullptr};
    int n_ob = n_st_max + n_dy_max;
    int threads = n_ob <= 64 ? 64 : (n_ob <= 128 ? 128 : 256);
    // every scan once, results kept in LDS, when they fit (MP_ST_REGIONS_ONCE=0: per-layer scans)
    const int n_items_max = n_st_max + n_dy_max * n_t;
    const size_t lds_once = lds + (size_t)n_items_max * 19 + 16;
    if (n_items_max <= 65535 && lds_once <= 64 * 1024 && mp_knob_st_regions_once() != 0) {
        threads = n_items_max <= 64 ? 64 : (n_items_max <= 128 ? 128 : 256);
        hipLaunchKernelGGL(st_regions_once_kernel, dim3(B), dim3(threads), lds_once, (hipStream_t)stream, a,
                           n_items_max);
        return st_launch_check("mp_st_regions");
    }
    hipLaunchKernelGGL(st_regions_kernel, dim3(B), dim3(threads), lds, (hipStream_t)stream, a);
    return st_launch_check("mp_st_regions");
}
