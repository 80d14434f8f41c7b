"""Turn the two rocprofv3 --pmc passes of tools/pmc_traffic.sh into a stamped traffic record.

    python tools/traffic_stamp.py gpurun_out/<dir> --key fused --workload config2 [--kernel rl_fused_sep_kernel]

Reads ``<dir>/FETCH_SIZE/**/counter_collection.csv`` and ``<dir>/WRITE_SIZE/...``, averages the
counter over the launches of the named kernel, applies the gfx950 corrections of
MI355X_MICROARCH.md (section HBM: FETCH_SIZE reports half the bytes of wide streaming reads -> x2; both
counters are in KiB) and writes

* ``<dir>/traffic_<key>.json``  -- the record ``bench.py`` reads from ``profiles/traffic.json``:
  bytes per launch, the kernel's full symbol, the workload and ``source_sha16`` =
  ``_lib.kernel_source_sha16()`` of the tree that was profiled (``bench.py`` reports ``traffic: null``
  as soon as the kernels' sources differ from it);
* ``<dir>/pmc_<key>.csv``       -- the per-kernel means, for ``profiles/``.

``--merge profiles/traffic.json`` also folds the record into that file (run in the build container,
where ``git rev-parse HEAD`` is available for the ``git_head`` field).
"""

from __future__ import annotations

import argparse
import collections
import csv
import glob
import json
import subprocess
import sys

from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def counter_means(d: Path, counter: str):
    agg = collections.defaultdict(list)
    meta = {}
    for f in sorted(glob.glob(str(d / counter / "**" / "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
            meta[r["Kernel_Name"]] = (r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"),
                                      r.get("Grid_Size"), r.get("Workgroup_Size"))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}, meta


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir", type=Path)
    ap.add_argument("--key", default="fused")
    ap.add_argument("--workload", default="config2")
    ap.add_argument("--kernel", default="rl_fused_sep_kernel")
    ap.add_argument("--merge", type=Path, default=None)
    ap.add_argument("--source", default=None, help="where the CSV will live under profiles/")
    args = ap.parse_args()

    from shrimpy_amd._lib import kernel_source_sha16

    fetch, meta = counter_means(args.dir, "FETCH_SIZE")
    write, _ = counter_means(args.dir, "WRITE_SIZE")
    rows = []
    for name in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(name, (0.0, 0))
        w, nw = write.get(name, (0.0, 0))
        rd, wr = 2.0 * f * 1024.0, w * 1024.0
        rows.append(dict(kernel=name, launches=max(nf, nw), FETCH_SIZE_KiB=f, WRITE_SIZE_KiB=w,
                         read_bytes=rd, write_bytes=wr, hbm_bytes=rd + wr,
                         vgpr=meta.get(name, ("",) * 5)[0], sgpr=meta.get(name, ("",) * 5)[1],
                         lds=meta.get(name, ("",) * 5)[2]))
    out_csv = args.dir / f"pmc_{args.key}.csv"
    with open(out_csv, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0]) if rows else ["kernel"])
        w.writeheader()
        w.writerows(rows)
    hits = [r for r in rows if args.kernel in r["kernel"]]
    if not hits:
        raise SystemExit(f"no launch of {args.kernel!r} in {args.dir}")
    main_row = max(hits, key=lambda r: r["launches"])
    try:
        head = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, capture_output=True,
                              text=True, check=True).stdout.strip()
    except Exception:  # noqa: BLE001 -- no .git on the GPU box
        head = None
    rec = {
        "workload": args.workload,
        "kernel": main_row["kernel"],
        "hbm_bytes_per_launch": main_row["hbm_bytes"],
        "read_bytes_per_launch": main_row["read_bytes"],
        "write_bytes_per_launch": main_row["write_bytes"],
        "launches_averaged": main_row["launches"],
        "source_sha16": kernel_source_sha16(),
        "git_head": head,
        "note": "(2 * FETCH_SIZE + WRITE_SIZE) * 1024, separate --pmc passes (tools/pmc_traffic.sh); "
                "gfx950 corrections per MI355X_MICROARCH.md section HBM",
        "source": args.source or str(out_csv),
    }
    (args.dir / f"traffic_{args.key}.json").write_text(json.dumps(rec, indent=1))
    print(json.dumps(rec))
    if args.merge:
        doc = json.loads(args.merge.read_text()) if args.merge.exists() else {}
        if head and not rec["git_head"]:
            rec["git_head"] = head
        doc[args.key] = rec
        args.merge.write_text(json.dumps(doc, indent=1) + "\n")


if __name__ == "__main__":
    main()
