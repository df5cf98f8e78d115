"""Host time to enqueue one step of the lattice benchmark against the GPU time it takes
(development aid): the loop that only enqueues, then the same loop waited for."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from motion_planning_amd import batch as mb

class G:  # no exchange
    active = False
    def submit_parts(self, parts): pass

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sc = bench.make_scenarios(0, B)
prm = mb.sl_params(early_exit=False, n_l=bench.N_L)
for n in (1, 2):
    loop = bench.SplitLoop(sc, prm, G(), n, False)
    for _ in range(5):
        loop.step()
    torch.cuda.synchronize()
    steps = 200
    t0 = time.perf_counter()
    for _ in range(steps):
        loop.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("B=%d, %d part(s): host enqueues a step in %.3f ms, the GPU finishes one every %.3f ms" %
          (B, n, (t1 - t0) / steps * 1e3, (t2 - t0) / steps * 1e3))
