"""Is the CPU port bench.py times (oracle.planner_forward) cost-equivalent to the thing it stands for?  SURVEY 8(d) / BASELINE.md
section 3 ask for the true reference and the restatement timed side by side on the build container's threads (the reference
cannot travel to the GPU box).  Three shapes, same state_dict and inputs for both, median of 7 timed forwards after 2 warm-ups,
all the container's threads.  BUILD CONTAINER ONLY (imports /root/reference through oracle/_ref_import.py):

    python tools/cpu_equivalence.py > profiles/r06a/cpu_equivalence.txt
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import magat_oracle as orc                         # noqa: E402
from oracle._ref_import import import_reference, make_config   # noqa: E402
from magat_pathplanning_amd.synthetic import comm_gso, fov_states  # noqa: E402


def med(fn, reps=7, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    _, classes = import_reference()
    threads = torch.get_num_threads()
    print("torch %s, %d threads (os.cpu_count() = %s)" % (torch.__version__, threads, os.cpu_count()))
    print("%-34s %12s %12s %8s %14s" % ("shape", "reference ms", "oracle ms", "ratio", "max|dlogit|"))
    worst = 0.0
    for B, N, mw, K, P, mode in ((8, 100, 50, 3, 4, "BottomNeck_skipConcat"), (32, 20, 28, 3, 4, "BottomNeck_only"),
                                 (64, 10, 20, 2, 1, "BottomNeck_only")):
        cfg = make_config(num_agents=N, nGraphFilterTaps=K, nAttentionHeads=P, bottleneckMode=mode)
        torch.manual_seed(5)
        model = classes[mode](cfg).eval()
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        x, S = fov_states(B, N, seed=1), comm_gso(B, N, mw, seed=2)

        def ref():
            with torch.no_grad():
                model.addGSO(S.clone())
                return model(x)

        def port():
            with torch.no_grad():
                return orc.planner_forward(x, S.clone(), sd, cfg)

        err = float((ref() - port()).abs().max())
        tr, tp = med(ref), med(port)
        worst = max(worst, tp / tr)
        print("%-34s %12.1f %12.1f %8.3f %14.3g" % ("B=%d N=%d K=%d P=%d %s" % (B, N, K, P, mode.replace("BottomNeck_", "")), tr * 1e3, tp * 1e3,
                                                   tp / tr, err))
    print("the port takes up to %.2fx the reference's time on these shapes: bench.py's cpu_baseline (the port) UNDERSTATES the "
          "reference's CPU throughput by at most that factor" % worst)


if __name__ == "__main__":
    main()
