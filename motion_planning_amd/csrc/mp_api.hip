// Library-wide pieces of the C ABI: version and last-error text.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "../../include/mplattice.h"

static thread_local char g_err[512] = "";

void mp_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int mp_version(void) { return MP_VERSION; }
extern "C" const char *mp_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------
// Testing knobs: read from the environment ONCE (first use), changeable afterwards
// only through mp_set_knob.  -1 = not set (the library's normal choice).
// ---------------------------------------------------------------------------
enum { KNOB_SL_NARROW, KNOB_SL_PAIR_CAP, KNOB_ST_REGIONS_ONCE, KNOB_ST_THREADS, KNOB_RL_THREADS, KNOB_FE_THREADS,
       KNOB_ST_GENERIC, KNOB_ORDER, KNOB_ST_LANES, KNOB_SL_ROWS, KNOB_CV_THREADS, KNOB_ST_PREFIX, KNOB_COUNT };
static const char *const g_knob_names[KNOB_COUNT] = {"MP_SL_NARROW", "MP_SL_PAIR_CAP", "MP_ST_REGIONS_ONCE",
                                                     "MP_ST_THREADS", "MP_RL_THREADS", "MP_FE_THREADS",
                                                     "MP_ST_GENERIC", "MP_ORDER", "MP_ST_LANES", "MP_SL_ROWS",
                                                     "MP_CV_THREADS", "MP_ST_PREFIX"};
static std::atomic<int> g_knobs[KNOB_COUNT];
static std::once_flag g_knobs_once;

// a workgroup-size knob is honoured only if it is a whole number of waves within the
// launch limit; anything else reads as "not set"
static int knob_clean(int k, int v)
{
    if (v < 0) return -1;
    if (k == KNOB_ST_THREADS || k == KNOB_RL_THREADS || k == KNOB_FE_THREADS || k == KNOB_CV_THREADS)
        return (v >= 64 && v <= 1024 && v % 64 == 0) ? v : -1;
    if (k == KNOB_ST_LANES) return (v == 4 || v == 8) ? v : -1; // lanes per cell of the speed DP
    if (k == KNOB_SL_ROWS) return (v >= 1 && v <= 255) ? v : -1;
    return v;
}

static void knobs_from_env()
{
    for (int k = 0; k < KNOB_COUNT; ++k) {
        const char *e = getenv(g_knob_names[k]);
        g_knobs[k].store(e && *e ? knob_clean(k, atoi(e)) : -1, std::memory_order_relaxed);
    }
}

static int knob(int k)
{
    std::call_once(g_knobs_once, knobs_from_env);
    return g_knobs[k].load(std::memory_order_relaxed);
}

int mp_knob_sl_narrow() { return knob(KNOB_SL_NARROW); }
int mp_knob_sl_pair_cap() { return knob(KNOB_SL_PAIR_CAP); }
int mp_knob_st_regions_once() { return knob(KNOB_ST_REGIONS_ONCE); }
int mp_knob_st_threads() { return knob(KNOB_ST_THREADS); }
int mp_knob_rl_threads() { return knob(KNOB_RL_THREADS); }
int mp_knob_fe_threads() { return knob(KNOB_FE_THREADS); }
int mp_knob_cv_threads() { return knob(KNOB_CV_THREADS); }
int mp_knob_st_prefix() { return knob(KNOB_ST_PREFIX); }
int mp_knob_st_generic() { return knob(KNOB_ST_GENERIC); }
int mp_knob_order() { return knob(KNOB_ORDER); }
int mp_knob_st_lanes() { return knob(KNOB_ST_LANES); }
int mp_knob_sl_rows() { return knob(KNOB_SL_ROWS); }

extern "C" int mp_set_knob(const char *name, int32_t value)
{
    if (!name) { mp_set_error("mp_set_knob: null name"); return MP_E_BADARG; }
    std::call_once(g_knobs_once, knobs_from_env);
    for (int k = 0; k < KNOB_COUNT; ++k)
        if (strcmp(name, g_knob_names[k]) == 0) {
            const int v = knob_clean(k, value);
            if (value >= 0 && v < 0) { mp_set_error("mp_set_knob: %s does not accept %d", name, (int)value); return MP_E_BADARG; }
            g_knobs[k].store(v, std::memory_order_relaxed);
            return MP_OK;
        }
    mp_set_error("mp_set_knob: unknown knob %s", name);
    return MP_E_BADARG;
}
