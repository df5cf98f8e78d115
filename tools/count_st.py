"""Counters of the split ST relaxation (development aid; needs a library built with -DST_COUNT,
named by MP_LIB_PATH)."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, '.')
from motion_planning_amd import batch as mb, _lib
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
sc = bench.make_scenarios(0, B)
args = [mb.to_dev(np.full(B, 97, np.int32), torch.int32), mb.to_dev(sc["down"]), mb.to_dev(sc["up"]),
        mb.to_dev(np.full((B, 11), R, np.int32), torch.int32), mb.to_dev(sc["cur_vel"])]
lib = _lib.lib()
out = (C.c_ulonglong * 16)()
lib.mp_debug_st_counters(None, 1)
mb.st_dp(*args, bench.SPEED_LIM, want_tables=False, n_s_max=97)
torch.cuda.synchronize()
lib.mp_debug_st_counters(out, 0)
names = ["wave steps", "lane steps", "quiet wave steps", "wave skip iterations", "lanes broke (full test)", "wave passes",
         "lanes the speed-only test would stop", "wave steps the speed-only test would skip", "first wave steps",
         "later steps that improved", "later steps that did not",
         "lanes speed+cp > best", "wave steps all lanes speed+cp > best", "lanes quiet+speed+cp > best", "wave steps all lanes quiet+speed+cp > best"]
for i, n in enumerate(names):
    print("%-45s %12d  per plan %.1f" % (n, out[i], out[i] / B))
