"""Instruction ledger of a kernel's main loop, from the compiler's listing.

    python tools/isa_ledger.py --tu rl_fused_ysep --define LSR_YSEP_PZ=9 --kernel 'rl_fused_ysep_kernelILi9ELi7ELi8ELi2ELb0E'
    python tools/isa_ledger.py --asm listing.s --kernel <substring of the mangled name>

Finds the kernel, takes its largest innermost-or-not loop (the label .. back-edge span with the most instructions: the
per-plane body of the stencil kernels) and counts its instructions by class -- what the SQ counters report as one
number (SQ_INSTS_VALU) broken down into arithmetic (packed / scalar FMAs, multiplies, adds), moves, masks
(v_cmp / v_cndmask), integer / address arithmetic, lane traffic (v_readlane / v_writelane / DPP moves), and next to them
LDS, vector-memory, scalar and waits.  Static counts of one trip through the body; conditional sub-blocks inside the body
(border tiles) are listed separately when --split-branches is given.
"""

from __future__ import annotations

import argparse
import collections
import re
import subprocess
import sys
import tempfile

from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "shrimpy_amd" / "csrc"
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only"]

_LABEL = re.compile(r"^(\.LBB\d+_\d+):")
_BRANCH = re.compile(r"^\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)|^\s*s_branch\s+(\.LBB\d+_\d+)")


def classify(op: str) -> str:
    if op.startswith(("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32")):
        return "valu: packed f32 arithmetic (v_pk_fma / mul / add)"
    if op.startswith(("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_mac_f32", "v_rcp_f32", "v_max_f32", "v_min_f32",
                      "v_fma_f64", "v_add_f64", "v_mul_f64", "v_cvt_")):
        return "valu: scalar float arithmetic"
    if op.startswith(("v_pk_mov_b32", "v_mov_b32", "v_mov_b64", "v_accvgpr", "v_swap")):
        return "valu: moves (v_mov / v_pk_mov / accvgpr)"
    if op.startswith(("v_cmp", "v_cndmask")):
        return "valu: masks (v_cmp / v_cndmask)"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane", "v_permlane", "v_bpermute", "ds_bpermute", "ds_swizzle")) or "dpp" in op:
        return "valu: lane traffic (readlane / writelane / dpp)"
    if op.startswith(("v_add_u32", "v_add_co", "v_addc", "v_sub_u32", "v_subrev", "v_lshl", "v_lshr", "v_ashr", "v_and_b32", "v_or_b32", "v_or3",
                      "v_xor", "v_mul_lo", "v_mul_hi", "v_mul_u32", "v_mad_u", "v_mad_i", "v_add3", "v_lshl_add", "v_add_lshl", "v_bfe", "v_bfi",
                      "v_min_u", "v_max_u", "v_min_i", "v_max_i", "v_sub_co", "v_subb", "v_not", "v_alignb", "v_perm")):
        return "valu: integer / address arithmetic"
    if op.startswith("v_"):
        return "valu: other"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "lds: reads"
    if op.startswith("ds_"):
        return "lds: writes / atomics"
    if op.startswith(("global_load_lds", "buffer_load") ) and "lds" in op:
        return "vmem: LDS-DMA loads"
    if op.startswith(("global_load", "flat_load", "buffer_load", "scratch_load")):
        return "vmem: loads"
    if op.startswith(("global_store", "flat_store", "buffer_store", "scratch_store", "global_atomic")):
        return "vmem: stores / atomics"
    if op.startswith("s_waitcnt"):
        return "scalar: s_waitcnt"
    if op.startswith("s_barrier"):
        return "scalar: s_barrier"
    if op.startswith(("s_cbranch", "s_branch")):
        return "scalar: branches"
    if op.startswith("s_nop"):
        return "scalar: s_nop"
    if op.startswith("s_"):
        return "scalar: other SALU / SMEM"
    return "other"


def kernel_lines(text: str, needle: str) -> list[str]:
    lines = text.splitlines()
    start = None
    for i, ln in enumerate(lines):
        head = ln.split(";")[0].rstrip()
        if head.endswith(":") and not head.startswith((".", " ", "\t")) and needle in head:
            start = i
            break
    if start is None:
        raise SystemExit(f"no kernel whose name contains {needle!r}")
    out = []
    for ln in lines[start + 1:]:
        if ln.strip().startswith("s_endpgm"):
            out.append(ln)
            break
        out.append(ln)
    return out


def instructions(lines):
    """[(index, label or None, opcode, text)] for the instructions and labels of a kernel."""
    out = []
    for ln in lines:
        m = _LABEL.match(ln)
        if m:
            out.append(("label", m.group(1), ln))
            continue
        s = ln.strip()
        if not s or s.startswith((";", ".", "//")):
            continue
        out.append(("ins", s.split()[0], s))
    return out


def loops(items):
    """Back edges: (first index, last index) of every span that starts at a label and ends at a branch back to it."""
    at = {name: i for i, (kind, name, _) in enumerate(items) if kind == "label"}
    spans = []
    for i, (kind, op, text) in enumerate(items):
        if kind != "ins":
            continue
        m = _BRANCH.match(text)
        if not m:
            continue
        target = m.group(1) or m.group(2)
        if target in at and at[target] < i:
            spans.append((at[target], i))
    return spans


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--tu")
    ap.add_argument("--define", action="append", default=[])
    ap.add_argument("--asm")
    ap.add_argument("--kernel", required=True)
    ap.add_argument("--per", type=float, default=None, help="divide the counts by this (e.g. trips per 64 voxels) for a second column")
    ap.add_argument("--per-label", default="per unit")
    args = ap.parse_args()
    if args.asm:
        text = Path(args.asm).read_text()
    else:
        with tempfile.TemporaryDirectory() as tmp:
            out = Path(tmp) / "k.s"
            cmd = [HIPCC, *FLAGS, *[f"-D{d}" for d in args.define], f"-I{CSRC}", str(CSRC / f"{args.tu}.hip"), "-o", str(out)]
            subprocess.run(cmd, check=True, capture_output=True)
            text = out.read_text()
    items = instructions(kernel_lines(text, args.kernel))
    spans = loops(items)
    if not spans:
        raise SystemExit("the kernel has no loop")
    first, last = max(spans, key=lambda s: sum(1 for k in items[s[0]:s[1] + 1] if k[0] == "ins"))
    body = [it for it in items[first:last + 1] if it[0] == "ins"]
    counts = collections.Counter(classify(op) for _, op, _ in body)
    total = sum(counts.values())
    valu = sum(v for k, v in counts.items() if k.startswith("valu"))
    print(f"kernel  {args.kernel}")
    print(f"main loop: {total} instructions ({valu} VALU) between {items[first][1]} and its back edge; "
          f"{len(spans)} loops in the kernel, {sum(1 for k in items if k[0] == 'ins')} instructions in all")
    width = max(len(k) for k in counts)
    for k, v in sorted(counts.items(), key=lambda kv: (kv[0].split(':')[0] != 'valu', -kv[1])):
        extra = f"  {v / args.per:8.1f} {args.per_label}" if args.per else ""
        print(f"  {k:{width}s} {v:6d}  {100.0 * v / total:5.1f} % of the body{extra}")
    ops = collections.Counter(op for _, op, _ in body if classify(op).startswith("valu"))
    print("  most frequent VALU opcodes: " + ", ".join(f"{op} x{n}" for op, n in ops.most_common(12)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
