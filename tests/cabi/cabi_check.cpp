// Calls libmplattice through its C ABI from plain C++ (HIP runtime only: no Python, no
// torch): the known-answer SL plan of the reference's default lattice, the known-answer
// ST plan, and one whole replan through mp_plan_scenes with the arena laid out from
// mp_plan_array_info.  Prints "cabi ok" and exits 0 when everything matches.
//
// Known answers (captured from the reference planner, see tests/golden):
//   SL: layers 8/16/28 m, 5 lanes, obstacles (9.5,0.2) (16,-1) (21.5,1.2) radius 1 -> lanes [4,4,3]
//   ST: 11 x 29 grid, five moving intervals, v0 = 5 -> s indices [0,0,1,2,3,4,5,6,7,8,9]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mplattice.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
#define CHECK_MP(x) do { int rc_ = (x); if (rc_ != MP_OK) { std::printf("mplattice error %d (%s) at %s:%d\n", rc_, mp_last_error(), __FILE__, __LINE__); return 1; } } while (0)
#define EXPECT(c) do { if (!(c)) { std::printf("FAILED: %s at %s:%d\n", #c, __FILE__, __LINE__); return 1; } } while (0)

template <class T>
static T *to_device(const std::vector<T> &v)
{
    T *p = nullptr;
    if (hipMalloc((void **)&p, v.size() * sizeof(T) + 8) != hipSuccess) return nullptr;
    if (hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return p;
}

template <class T>
static std::vector<T> to_host(const T *p, size_t n)
{
    std::vector<T> v(n);
    (void)hipMemcpy(v.data(), p, n * sizeof(T), hipMemcpyDeviceToHost);
    return v;
}

int main()
{
    EXPECT(mp_version() == MP_VERSION);
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));

    // ---- SL path DP, one scenario --------------------------------------------------
    {
        std::vector<double> s_layers{8.0, 16.0, 28.0}, l_row(5), ego{0.0, 0.0};
        for (int i = 0; i < 5; ++i) l_row[i] = (i == 4) ? 1.45 : i * (2.9 / 4) + -1.45;
        std::vector<double> ob_s{9.5, 16.0, 21.5}, ob_l{0.2, -1.0, 1.2}, ob_dist{1.0};
        std::vector<int32_t> n_ob{3};
        mp_sl_batch b;
        std::memset(&b, 0, sizeof(b));
        b.B = 1; b.n_s = 3; b.n_l = 5; b.ob_stride = 3;
        b.s_layers = to_device(s_layers); b.l_map = to_device(l_row); b.l_bstride = 5; b.l_rstride = 0;
        b.ego_sl = to_device(ego); b.ob_s = to_device(ob_s); b.ob_l = to_device(ob_l);
        b.n_ob = to_device(n_ob); b.ob_dist = to_device(ob_dist); b.n_layers = nullptr;
        mp_sl_params prm{0.5, 0.5, 1e4, 1, 2};
        double *cost_map; int32_t *index_map, *path_idx, *status, *layers_done; double *best;
        CHECK_HIP(hipMalloc((void **)&cost_map, 15 * 8)); CHECK_HIP(hipMalloc((void **)&index_map, 15 * 4));
        CHECK_HIP(hipMalloc((void **)&path_idx, 3 * 4)); CHECK_HIP(hipMalloc((void **)&status, 4));
        CHECK_HIP(hipMalloc((void **)&layers_done, 4)); CHECK_HIP(hipMalloc((void **)&best, 8));
        CHECK_MP(mp_sl_plan(&b, &prm, cost_map, index_map, path_idx, status, layers_done, best, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        auto path = to_host(path_idx, 3);
        auto cm = to_host(cost_map, 15);
        EXPECT(to_host(status, 1)[0] == MP_SL_FOUND);
        EXPECT(path[0] == 4 && path[1] == 4 && path[2] == 3);
        EXPECT(std::fabs(cm[0] - 3.8117455703292382) < 1e-12 && cm[2] == 0.0);
        EXPECT(std::fabs(cm[14] - 30025.887995570331) < 1e-8);
        // argument errors come back as codes, nothing is enqueued
        EXPECT(mp_sl_plan(nullptr, &prm, cost_map, index_map, path_idx, status, layers_done, best, stream) == MP_E_BADARG);
        std::printf("SL plan: lanes %d %d %d, cost of the end node %.6f\n", path[0], path[1], path[2], to_host(best, 1)[0]);
    }
    // ---- ST speed DP, one scenario -------------------------------------------------
    {
        const int n_t = 11, n_s = 29, R = 5;
        const double s0[R] = {12, 20, 6, 25, 15}, v[R] = {2, 0, 4, -1, 1};
        std::vector<double> down(n_t * R), up(n_t * R), cur_vel{5.0};
        for (int i = 0; i < n_t; ++i)
            for (int k = 0; k < R; ++k) { down[i * R + k] = s0[k] + v[k] * 0.5 * i; up[i * R + k] = down[i * R + k] + 3.0; }
        std::vector<int32_t> n_reg(n_t, R), ns{n_s};
        int32_t *path, *len;
        CHECK_HIP(hipMalloc((void **)&path, n_t * 4)); CHECK_HIP(hipMalloc((void **)&len, 4));
        CHECK_MP(mp_st_dp(1, n_t, n_s, to_device(ns), to_device(down), to_device(up), to_device(n_reg), R,
                          to_device(cur_vel), 25.0 * 5.0 / 18.0, 1.0, 0.5, nullptr, nullptr, path, len, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        auto p = to_host(path, n_t);
        EXPECT(to_host(len, 1)[0] == 11);
        const int want[11] = {0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9};
        for (int i = 0; i < n_t; ++i) EXPECT(p[i] == want[i]);
        std::printf("ST plan: 11 layers, last s index %d\n", p[10]);
    }
    // ---- whole replan of two scenes --------------------------------------------------
    {
        const int B = 2, NK = 7, NST = 3, NDY = 1, F = 11;
        std::vector<double> kx(B * NK), ky(B * NK, 0.0), vec(B, 0.0), ego_vel{5.0, 0.0, 4.0, 0.3};
        for (int b = 0; b < B; ++b)
            for (int i = 0; i < NK; ++i) kx[b * NK + i] = -2.0 + 4.5 * i;
        std::vector<int32_t> n_knots{NK, NK}, n_st{3, 0}, n_dy{0, 1};
        // scene 0: the known-answer scene in Cartesian form (straight line: s = x, l = y)
        std::vector<double> st_pos{8.0, 0.2, 14.0, -1.0, 19.0, 1.2, 0, 0, 0, 0, 0, 0};
        std::vector<double> st_rad{1.0, 1.0, 1.0, 0, 0, 0}, dy_rad{0.0, 0.8}, dy_pos(B * NDY * F * 2, 0.0);
        for (int f = 0; f < F; ++f) { dy_pos[(1 * F + f) * 2] = 15.0 + 1.0 * 0.5 * f; dy_pos[(1 * F + f) * 2 + 1] = 3.0; }
        mp_scene_batch sc{to_device(kx), to_device(ky), to_device(n_knots), to_device(vec), to_device(ego_vel),
                          to_device(st_pos), to_device(st_rad), to_device(n_st), to_device(dy_pos),
                          to_device(dy_rad), to_device(n_dy)};
        mp_plan_dims d{B, NK, NST, NDY, F, 5, 64, 14, 208, 11, 102, 4, 54, 0};
        mp_plan_params prm;
        mp_plan_default_params(&prm);
        const int64_t bytes = mp_plan_arena_bytes(&d);
        EXPECT(bytes > 0);
        void *arena;
        CHECK_HIP(hipMalloc(&arena, bytes));
        CHECK_MP(mp_plan_scenes(&sc, &d, &prm, arena, bytes, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        auto find = [&](const char *want, int64_t shape[4]) -> int64_t {
            for (int i = 0; i < mp_plan_array_count(); ++i) {
                const char *name; int32_t dt; int64_t off;
                if (mp_plan_array_info(&d, i, &name, &dt, shape, &off) == MP_OK && std::string(name) == want) return off;
            }
            return -1;
        };
        int64_t shp[4];
        const int64_t o_status = find("status", shp), o_path = find("path_idx", shp);
        EXPECT(o_status >= 0 && o_path >= 0 && shp[0] == B && shp[1] == 14);
        auto status = to_host((const int32_t *)((char *)arena + o_status), B);
        auto path = to_host((const int32_t *)((char *)arena + o_path), B * 14);
        const int64_t o_np = find("n_prof", shp), o_vv = find("vv", shp), o_fx = find("final_x", shp);
        auto n_prof = to_host((const int32_t *)((char *)arena + o_np), B);
        auto vv = to_host((const double *)((char *)arena + o_vv), B * 54);
        auto fx = to_host((const double *)((char *)arena + o_fx), B * 54);
        EXPECT(status[0] == MP_PLAN_OK && status[1] == MP_PLAN_OK);
        EXPECT(path[0] == 4 && path[1] == 4 && path[2] == 3 && path[3] == -1);   // the known-answer lanes
        EXPECT(n_prof[0] >= 2 && n_prof[0] <= 51 && vv[0] == 5.0);               // profile starts at the ego speed
        EXPECT(fx[n_prof[0] - 1] > fx[0]);
        std::printf("replan: scene 0 lanes %d %d %d, %d path points; scene 1 %d path points\n", path[0], path[1],
                    path[2], n_prof[0], n_prof[1]);
    }
    std::printf("cabi ok\n");
    return 0;
}
