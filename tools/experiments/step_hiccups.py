"""Where do the odd slow stretches of a bench.py run fall?  (round 6: the timed region of the 32768-scene lines was 25-30 ms longer than 200 x the
sustained step in some runs and not in others - profiles/r06_ab_timed_region.txt.)

Builds bench.py's workload without its spin-up, then issues groups of G steps, each group followed by a device synchronisation, for T seconds; prints the
median group time and every group that took more than 1.5 x the median with the time since the first launch at which it started.

    python tools/experiments/step_hiccups.py [--config 3] [--seconds 3] [--group 10]
"""
import os, sys, time, json

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench


def main():
    argv = sys.argv[1:]
    def take(flag, default, cast):
        if flag in argv:
            i = argv.index(flag); v = cast(argv[i + 1]); del argv[i:i + 2]; return v
        return default
    seconds = take("--seconds", 3.0, float)
    group = take("--group", 10, int)
    args = bench.parse(argv + ["--spinup", "0", "--no-cpu-baseline", "--no-companions"])
    work = bench.HipWorkload(args, 0, torch.device("cuda:0"))
    torch.cuda.synchronize()
    t_first = time.perf_counter()
    rec = []
    while time.perf_counter() - t_first < seconds:
        t0 = time.perf_counter()
        for _ in range(group):
            work.step()
        torch.cuda.synchronize()
        rec.append((t0 - t_first, time.perf_counter() - t0))
    d = sorted(r[1] for r in rec)
    med = d[len(d) // 2]
    slow = [(round(t, 4), round(x * 1e3, 3)) for t, x in rec if x > 1.5 * med]
    print(json.dumps({"workload": work.metric_name(), "group": group, "groups": len(rec), "median_group_ms": med * 1e3, "median_step_ms": med * 1e3 / group,
                      "first_groups_ms": [round(x * 1e3, 3) for _, x in rec[:8]],
                      "slow_groups (start s since first launch, ms)": slow[:40], "n_slow": len(slow),
                      "excess_ms_total": round(sum(x - med for _, x in rec if x > 1.5 * med) * 1e3, 2)}))


if __name__ == "__main__":
    main()
