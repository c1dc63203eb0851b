"""Where does a bench step's wall time go?  Reads a rocprofv3 ``--kernel-trace`` CSV of ``bench.py`` and prints, for the
last few steps (a step ends with the fused-AdamW kernel): wall time, per-queue busy time, time where NO queue runs a
kernel (host-bound gaps), launches, and the largest idle gaps with the kernels around them.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d out -o x -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline
    python tools/trace_gaps.py out/**/x_kernel_trace.csv
"""
from __future__ import annotations

import csv
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = name.replace("void ", "").replace("vc::", "")
    return name[:70]


def main(path: str, last: int = 6) -> None:
    rows = list(csv.DictReader(open(path)))
    ev = []
    for r in rows:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Stream_Id", "")))
    ev.sort()
    ends = [i for i, e in enumerate(ev) if "clip_adamw_kernel" in e[2]]   # virconv_amd.optim.ClipAdamW; else the stock fused AdamW:
    if len(ends) < last + 1:
        ends = [i for i, e in enumerate(ev) if "multi_tensor_apply" in e[2] and "Adam" in e[2] or "fused_adam" in e[2].lower()]
    if len(ends) < last + 1:
        ends = [i for i, e in enumerate(ev) if "multi_tensor_apply" in e[2]]
    # one step may run several multi-tensor kernels back to back: keep the last of each burst
    marks = [i for j, i in enumerate(ends) if j + 1 == len(ends) or ends[j + 1] - i > 20]
    marks = marks[-(last + 1):]
    print(f"{len(ev)} kernels, {len(marks) - 1} steps analysed")
    for a, b in zip(marks[:-1], marks[1:]):
        seg = ev[a + 1:b + 1]
        t0, t1 = ev[a][1], ev[b][1]
        busy = defaultdict(int)
        for s, e, _, q, st in seg:
            busy[(q, st)] += e - s
        # union of busy intervals
        iv = sorted((s, e) for s, e, *_ in seg)
        cov, cur_s, cur_e = 0, None, None
        gaps = []
        for s, e in iv:
            if cur_e is None:
                cur_s, cur_e = s, e
                if s > t0:
                    gaps.append((s - t0, t0, s))
            elif s <= cur_e:
                cur_e = max(cur_e, e)
            else:
                gaps.append((s - cur_e, cur_e, s))
                cov += cur_e - cur_s
                cur_s, cur_e = s, e
        cov += cur_e - cur_s
        print(f"step: wall {1e-6 * (t1 - t0):.3f} ms, {len(seg)} launches, some-queue-busy {1e-6 * cov:.3f} ms, all-idle {1e-6 * (t1 - t0 - cov):.3f} ms; "
              + ", ".join(f"q{q}/s{st}: {1e-6 * v:.3f} ms" for (q, st), v in sorted(busy.items(), key=lambda kv: -kv[1])))
    # gap histogram + biggest gaps of the last step
    seg = ev[marks[-2] + 1:marks[-1] + 1]
    byq = defaultdict(list)
    for e in seg:
        byq[(e[3], e[4])].append(e)
    mq = max(byq, key=lambda k: sum(e[1] - e[0] for e in byq[k]))
    lst = sorted(byq[mq])
    g = [(lst[i + 1][0] - lst[i][1], lst[i][2], lst[i + 1][2]) for i in range(len(lst) - 1)]
    tot = sum(max(x[0], 0) for x in g)
    print(f"main queue {mq}: {len(lst)} kernels, sum of inter-kernel gaps {1e-6 * tot:.3f} ms "
          f"(<2us: {sum(1 for x in g if x[0] < 2000)}, 2-5us: {sum(1 for x in g if 2000 <= x[0] < 5000)}, 5-20us: {sum(1 for x in g if 5000 <= x[0] < 20000)}, >20us: {sum(1 for x in g if x[0] >= 20000)})")
    for d, a, b in sorted(g, reverse=True)[:25]:
        print(f"  gap {1e-3 * d:8.1f} us  after {short(a)}  before {short(b)}")
    agg = defaultdict(lambda: [0, 0])
    for s, e, n, q, st in seg:
        k = (short(n)[:60], (q, st) == mq)
        agg[k][0] += 1
        agg[k][1] += e - s
    print("kernels of the last step (count, total us, on main queue):")
    for (n, on), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
        print(f"  {c:4d} {1e-3 * t:9.1f} {'main' if on else 'side'}  {n}")


    if "--sequence" in sys.argv:
        # every kernel of the last step in start order: offset from the step's first kernel, duration, queue, name
        t0 = min(e[0] for e in seg)
        qn = {k: i for i, k in enumerate(sorted(byq, key=lambda k: -sum(e[1] - e[0] for e in byq[k])))}
        print("sequence of the last step (start us, duration us, queue rank by busy time: 0 = main, name):")
        for s, e, n, q, st in sorted(seg):
            print(f"  {1e-3 * (s - t0):9.1f} {1e-3 * (e - s):8.1f}  q{qn[(q, st)]}  {short(n)}")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    main(args[0], int(args[1]) if len(args) > 1 else 6)
