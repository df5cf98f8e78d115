"""Counters of the split ST relaxation (development aid; needs a library built with -DST_COUNT,
named by MP_LIB_PATH):

    cd motion_planning_amd/csrc && hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off \
        -DST_COUNT -c -o build/st_count.o st_kernels.hip && hipcc --offload-arch=gfx950 -shared -fPIC \
        -o ../libmplattice_count.so build/st_count.o $(ls build/*.o | grep -v -e st_kernels -e st_count)
    MP_LIB_PATH=$PWD/motion_planning_amd/libmplattice_count.so python tools/count_st.py [B] [R]
"""
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
names = {0: "wave steps", 1: "lane steps (edges priced)", 2: "wave steps on the quiet obstacle sum",
         4: "lanes stopped by the bound", 5: "wave passes", 8: "first wave steps",
         9: "later lane steps that improved", 10: "later lane steps that did not",
         11: "second sweep: cells not priced (constants above the bound)", 12: "second sweep: cells",
         13: "second sweep: cells too far ahead of every cell within the bound"}
for i, n in names.items():
    print("%-45s %12d  per plan %.1f" % (n, out[i], out[i] / B))
