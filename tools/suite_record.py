"""Write ``profiles/<tag>_gpu_suite.txt`` from the last GPU-box run of the suite and of ``smoke()``:

    python tools/suite_record.py r03 [--dir gpurun_out/r3]

expects ``<dir>/final_suite.log`` (pytest output + ``rc=``), ``<dir>/final_trace.txt`` (``LSR_TEST_TRACE``) and
``<dir>/final_smoke.log`` (``smoke()`` output + ``kernel_source_sha16 <hash>``), as the round's last gpurun call leaves them."""
import argparse
import glob
import json
import subprocess
import sys

from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--dir", default="gpurun_out/final")
    args = ap.parse_args()
    d = ROOT / args.dir
    head = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    from shrimpy_amd._lib import kernel_source_sha16

    tree = kernel_source_sha16()
    suite = [ln for ln in (d / "final_suite.log").read_text().splitlines()
             if "lsr-test" not in ln and ln.strip() and not ln.strip().startswith(".")]
    smoke = (d / "final_smoke.log").read_text().strip()
    # the node ids as the run itself printed them (tests/conftest.py writes them to the real stderr, which the
    # command redirected into the same log)
    import re

    trace = "\n".join(m for ln in (d / "final_suite.log").read_text().splitlines()
                      for m in re.findall(r"\[lsr-test[^\]]*\] (?:start|done|session finished)[^\[]*", ln))
    stamp = json.loads((ROOT / "profiles" / "traffic.json").read_text())["fused"].get("source_sha16")
    earlier = []
    for f in sorted(glob.glob(str(d / "suite*.log")), key=lambda p: Path(p).stat().st_mtime):
        tail = [ln for ln in Path(f).read_text().splitlines() if " passed" in ln or ln.startswith("rc=")]
        earlier.append(f"{Path(f).name}: " + " | ".join(tail[-2:]))
    text = f"""`python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider` on a fresh MI355X box (gpurun), then smoke().
git HEAD when this record was written: {head}
kernel_source_sha16 of this tree (csrc/*.hip, *.hpp, Makefile, include/lsrecon.h): {tree}
kernel_source_sha16 printed by the run: see the smoke section; stamp of profiles/traffic.json: {stamp}

---- summary
{chr(10).join(suite[-4:])}

---- smoke
{smoke}

---- earlier full-suite runs of this round, each on its own fresh box
{chr(10).join(earlier) if earlier else "(none kept)"}

---- every test with its start / finish time (tests/conftest.py, LSR_TEST_TRACE)
{trace}"""
    out = ROOT / "profiles" / f"{args.tag}_gpu_suite.txt"
    out.write_text(text)
    print(out, "tree", tree, "stamp", stamp, "|", suite[-2] if len(suite) > 1 else suite)


if __name__ == "__main__":
    main()
