"""GPU-side timeline of a streamed store-to-store run from a rocprofv3 trace.

    cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/tl -- \
        python3 $R/bench.py --workload config4 --scratch /dev/shm --no-cpu-baseline --engine-format --output-compression blosc-zstd
    python tools/probes/gpu_timeline.py gpurun_out/tl

Groups the kernels into decode / deskew / RL / encode / other and the copies into H2D / D2H, takes the window between the
first and the last RL launch of the streamed plate (the bench's resident-kernel timing before it is cut off by its gap),
and prints: busy time of each group, the union of all kernel intervals (the GPU executing anything), what fraction of the
window no kernel runs, and the largest idle gaps with what ran either side of them.
"""

from __future__ import annotations

import csv
import glob
import json
import sys


def load(d):
    ks, cs = [], []
    for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    for f in glob.glob(f"{d}/**/*memory_copy_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            cs.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "?"), int(r.get("Bytes", 0) or 0)))
    return sorted(ks), sorted(cs)


def group(name):
    if "decode_blocks" in name or "unshuffle" in name:
        return "decode"
    if "encode_blocks" in name or "frames_kernel" in name:
        return "encode"
    if "rl_fused" in name or "correlate" in name:
        return "rl"
    if "deskew" in name:
        return "deskew"
    return "other"


def union(iv):
    total, end = 0, None
    for a, b in sorted(iv):
        if end is None or a > end:
            total += b - a
            end = b
        elif b > end:
            total += b - end
            end = b
    return total


def main():
    d = sys.argv[1]
    ks, cs = load(d)
    rl = [k for k in ks if group(k[2]) == "rl"]
    # the streamed plate = the last long run of RL launches: cut at the largest gap between consecutive RL launches
    gaps = [(rl[i + 1][0] - rl[i][1], i) for i in range(len(rl) - 1)]
    cut = max(gaps)[1] + 1 if gaps else 0
    first_half, second_half = rl[:cut], rl[cut:]
    part = second_half if len(second_half) >= len(first_half) else first_half
    t0, t1 = part[0][0], part[-1][1]
    win = [k for k in ks if k[1] > t0 and k[0] < t1]
    wc = [c for c in cs if c[1] > t0 and c[0] < t1]
    span = t1 - t0
    out = {"window_ms": round(span / 1e6, 2), "rl_launches_in_window": len(part)}
    for g in ("decode", "deskew", "rl", "encode", "other"):
        iv = [(max(a, t0), min(b, t1)) for a, b, n in win if group(n) == g]
        out[f"{g}_busy_ms"] = round(union(iv) / 1e6, 2)
        out[f"{g}_launches"] = len(iv)
    allk = [(max(a, t0), min(b, t1)) for a, b, _ in win]
    busy = union(allk)
    out["any_kernel_ms"] = round(busy / 1e6, 2)
    out["no_kernel_frac"] = round(1 - busy / span, 3)
    out["sum_of_groups_ms"] = round(sum(out[f"{g}_busy_ms"] for g in ("decode", "deskew", "rl", "encode", "other")), 2)
    for direction in sorted({c[2] for c in wc}):
        iv = [(max(a, t0), min(b, t1)) for a, b, dd, _ in wc if dd == direction]
        out[f"copy_{direction}_ms"] = round(union(iv) / 1e6, 2)
        out[f"copy_{direction}_GB"] = round(sum(c[3] for c in wc if c[2] == direction) / 1e9, 2)
    # idle gaps
    ivs = sorted(allk)
    gaps, end, last = [], ivs[0][1], None
    names = sorted(win)
    cur_end, cur_name = names[0][1], names[0][2]
    for a, b, n in names[1:]:
        if a > cur_end:
            gaps.append((a - cur_end, group(cur_name), group(n)))
        if b > cur_end:
            cur_end, cur_name = b, n
    gaps.sort(reverse=True)
    out["largest_idle_gaps_ms (after -> before)"] = [(round(g / 1e6, 2), x, y) for g, x, y in gaps[:12]]
    out["idle_gap_total_ms"] = round(sum(g for g, _, _ in gaps) / 1e6, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
