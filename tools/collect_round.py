"""Copy what ``tools/round_end.sh <tag>`` (and the secondary-kernel profile) left under ``gpurun_out/`` into
``profiles/`` and finish the traffic record with the commit it was measured on.

    python tools/collect_round.py r02 [--secondary gpurun_out/r02_secondary_stats3]
"""
import argparse
import csv
import glob
import json
import shutil
import subprocess
import sys

from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--secondary", default=None, help="rocprofv3 output directory of tools/bench_kernels.py")
    args = ap.parse_args()
    tag, out, prof = args.tag, ROOT / "gpurun_out", ROOT / "profiles"
    newest = max(glob.glob(str(out / f"{tag}_stats" / "**" / "*kernel_stats.csv"), recursive=True), key=lambda f: Path(f).stat().st_mtime)
    shutil.copy(newest, prof / f"{tag}_kernel_stats.csv")
    shutil.copy(out / f"{tag}_pmc" / "pmc_fused.csv", prof / f"{tag}_pmc_traffic_fused.csv")
    for name in ("bench", "bench_config4", "bench_config5"):
        shutil.copy(out / f"{tag}_{name}.json", prof / f"{tag}_{name}.json")
    doc = json.loads((out / f"{tag}_traffic.json").read_text())
    head = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    doc["fused"]["git_head"] = head
    doc["fused"]["source"] = f"profiles/{tag}_pmc_traffic_fused.csv"
    (prof / "traffic.json").write_text(json.dumps(doc, indent=1) + "\n")
    from shrimpy_amd._lib import kernel_source_sha16

    print("traffic stamp", doc["fused"]["source_sha16"], "tree", kernel_source_sha16(), "bytes", doc["fused"]["hbm_bytes_per_launch"])
    first = list(csv.DictReader(open(newest)))[0]
    log = (out / f"{tag}_stats.log").read_text()
    i = log.rfind('"launch_ms": ')
    print("rocprofv3:", first["Name"][:60], first["Calls"], float(first["AverageNs"]) / 1e6, "ms;  HIP events in the same run:",
          log[i + 13:i + 22])
    b = json.loads((prof / f"{tag}_bench.json").read_text())
    print("config2", b["value"], b["ms_per_step"], b["roofline"]["frac"], b["roofline"]["traffic"], b["cpu_baseline"]["value"])
    for name in ("bench_config4", "bench_config5"):
        d = json.loads((prof / f"{tag}_{name}.json").read_text())
        print(name, d["ms_per_step"], d["roofline"]["frac"], d["config"]["store_to_store"]["s_per_unit"])
    if (out / f"{tag}_secondary_kernels.jsonl").exists():
        shutil.copy(out / f"{tag}_secondary_kernels.jsonl", prof / f"{tag}_secondary_kernels.jsonl")
    if args.secondary:
        f = max(glob.glob(str(Path(args.secondary) / "**" / "*kernel_stats.csv"), recursive=True), key=lambda p: Path(p).stat().st_mtime)
        rows = list(csv.DictReader(open(f)))
        keep = [r for r in rows if ("(anonymous namespace)::" in r["Name"] and "at::" not in r["Name"]) or "lsr::" in r["Name"]
                or r["Name"].startswith(("fft_rtc", "transpose_rtc", "r2c_", "c2r_", "bluestein"))]
        with open(prof / f"{tag}_secondary_kernel_stats.csv", "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=rows[0].keys())
            w.writeheader()
            w.writerows(keep)


if __name__ == "__main__":
    main()
