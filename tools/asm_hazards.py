"""Static check of the hand-pipelined kernels' register discipline: no instruction may touch a VGPR that a vector-memory
load still in flight is going to write.

The stencil kernels (csrc/rl_fused_sep.hip, rl_fused_ysep.hip, correlate_sep.hip, correlate_dense.hip) issue their global
loads as inline asm and wait with hand-counted ``s_waitcnt vmcnt(N)``; the hardware has no interlock between a VMEM load
and a later VALU instruction on the same register -- the wait is the only thing that orders them.  The compiler knows
nothing of the loads in flight: a destination whose value is dead *to it* (a placeholder load of the prologue, the
prefetch of a plane past the last one) is a free register, and whatever it puts there is overwritten when the load
lands.  Round 4 met this twice (sums computed into such a register above the kernel's final wait; LDS offsets computed
into the destination of a prologue load) -- wrong results that changed from run to run, in two of ~120 kernel instances.
This tool finds such code in the compiler's output instead of on the GPU:

    python tools/asm_hazards.py                      # every stencil translation unit, all compiled tap counts
    python tools/asm_hazards.py --tu rl_fused_sep --pz 9
    python tools/asm_hazards.py --asm some_kernel.s  # an existing hipcc -S --cuda-device-only listing

Model (gfx9 / CDNA: loads, LDS-DMA loads and stores share ONE in-order counter, vmcnt): forward data flow over the
kernel's basic blocks; the state maps every VGPR that is the destination of a load in flight to the smallest number of
vector-memory operations issued after that load on any path (``s_waitcnt vmcnt(N)`` retires exactly the loads with at
least N younger operations; the minimum over paths is the conservative side).  An instruction that reads or writes such
a register is reported -- except a load into the same register (returns are in order, so the younger value wins).
Exit status 1 if anything is reported.
"""

from __future__ import annotations

import argparse
import re
import subprocess
import sys
import tempfile

from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "shrimpy_amd" / "csrc"
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only"]
# translation unit -> (define, compiled tap counts): shrimpy_amd/csrc/Makefile
UNITS = {
    "rl_fused_sep": ("LSR_FUSED_PZ", (3, 5, 7, 9, 11, 13, 15)),
    "rl_fused_ysep": ("LSR_YSEP_PZ", (3, 5, 7, 9, 11)),
    "correlate_sep": ("LSR_SEP_PZ", (3, 5, 7, 9, 11, 13, 15)),
    "correlate_dense": ("LSR_DENSE_PZ", (3, 5, 7, 9, 11)),
}
CAP = 64  # vmcnt is a 6-bit counter: "at least 64 younger operations" is as good as retired

_VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
_LABEL = re.compile(r"^(\.LBB\d+_\d+):")
_VMCNT = re.compile(r"vmcnt\((\d+)\)")
_VMEM_PREFIX = ("global_", "flat_", "buffer_", "scratch_", "tbuffer_")


def vregs(text: str) -> list[int]:
    out = []
    for m in _VREG.finditer(text):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


class Instr:
    __slots__ = ("line", "text", "op", "is_vmem", "load_dst", "touched", "dont_care", "wait", "target", "cond", "ends")

    def __init__(self, line: int, text: str):
        self.line, self.text = line, text
        parts = text.split(None, 1)
        self.op = parts[0]
        operands = parts[1] if len(parts) > 1 else ""
        operands = operands.split(";")[0]
        self.is_vmem = self.op.startswith(_VMEM_PREFIX)
        self.load_dst: list[int] = []
        if self.is_vmem and "_load_" in self.op and "_lds_" not in self.op:
            first = operands.split(",")[0]
            self.load_dst = vregs(first)
        elif self.is_vmem and "_atomic_" in self.op and ("sc0" in operands.split() or " glc" in operands):
            self.load_dst = vregs(operands.split(",")[0])     # an atomic that returns the old value
        self.touched = vregs(operands) if not self.op.startswith("s_") or self.op in ("s_nop",) else []
        # 64-bit integer arithmetic whose result is used in its low half only (32-bit offsets widened for an add): the
        # compiler leaves the high half of the pair operand undefined and the allocator gives it any register -- one with a
        # load in flight included.  Reading garbage nobody uses disturbs nothing: the HIGH register of a pair SOURCE of
        # these instructions is not reported.
        self.dont_care = set()
        if self.op in ("v_mad_u64_u32", "v_mad_i64_i32", "v_lshl_add_u64"):
            ops = [o.strip() for o in operands.split(",")]
            for o in ops[1:]:
                m = re.fullmatch(r"v\[(\d+):(\d+)\]", o)
                if m and int(m.group(2)) == int(m.group(1)) + 1:
                    self.dont_care.add(int(m.group(2)))
            self.dont_care -= set(vregs(ops[0]))
        self.wait = None
        if self.op == "s_waitcnt":
            m = _VMCNT.search(operands)
            if m:
                self.wait = int(m.group(1))
            elif re.fullmatch(r"\s*(0x[0-9a-f]+|\d+)\s*", operands):    # raw immediate: vmcnt = bits 3:0 and 15:14
                imm = int(operands.strip(), 0)
                self.wait = (imm & 0xF) | ((imm >> 14) & 0x3) << 4
        self.target, self.cond = None, False
        if self.op == "s_branch":
            self.target = operands.strip()
        elif self.op.startswith("s_cbranch_"):
            self.target, self.cond = operands.strip(), True
        self.ends = self.op in ("s_endpgm",)


def split_kernels(lines: list[str]):
    """(name, [(line number, text)]) per kernel of an AMDGPU assembly listing."""
    name, body, out = None, [], []
    for i, raw in enumerate(lines, 1):
        if name is None:
            m = re.match(r"^(_Z\w+):\s*(;.*)?$", raw)
            if m:
                name, body = m.group(1), []
            continue
        t = raw.strip()
        if not t or t.startswith(";"):
            continue
        if _LABEL.match(t):
            body.append((i, t))
            continue
        if t.startswith("."):
            if t.startswith(".Lfunc_end"):
                out.append((name, body))
                name = None
            continue
        body.append((i, t))
    return out


def analyse(body) -> list[dict]:
    # ---- basic blocks
    instrs, label_at = [], {}
    for line, t in body:
        m = _LABEL.match(t)
        if m:
            label_at[m.group(1)] = len(instrs)
            rest = t[m.end():].strip()
            if not rest or rest.startswith(";"):
                continue
            t = rest
        instrs.append(Instr(line, t))
    leaders = {0} | set(label_at.values())
    for k, ins in enumerate(instrs):
        if ins.target is not None or ins.ends:
            leaders.add(k + 1)
    leaders = sorted(x for x in leaders if x < len(instrs))
    block_of = {}
    blocks = []
    for b, s in enumerate(leaders):
        e = leaders[b + 1] if b + 1 < len(leaders) else len(instrs)
        blocks.append((s, e))
        block_of[s] = b
    succ = []
    for s, e in blocks:
        last = instrs[e - 1]
        nxt = []
        if last.target is not None and last.target in label_at:
            nxt.append(block_of[label_at[last.target]])
        if not last.ends and (last.target is None or last.cond) and e < len(instrs):
            nxt.append(block_of[e])
        succ.append(nxt)

    # ---- data flow: state = {vgpr: (younger operations (min over paths), line of the load)}
    def transfer(state, s, e, report):
        state = dict(state)
        for k in range(s, e):
            ins = instrs[k]
            if ins.wait is not None:
                state = {r: v for r, v in state.items() if v[0] < ins.wait}
                continue
            if state and ins.touched:
                own = set(ins.load_dst)
                for r in ins.touched:
                    if r in state and r not in own and r not in ins.dont_care:
                        report(ins, r, state[r])
            if ins.is_vmem:
                state = {r: (min(v[0] + 1, CAP), v[1]) for r, v in state.items() if v[0] + 1 < CAP}
                for r in ins.load_dst:
                    state[r] = (0, ins.line)
        return state

    def join(a, b):
        if a is None:
            return dict(b)
        out = dict(a)
        for r, v in b.items():
            if r not in out or v[0] < out[r][0]:
                out[r] = v
        return out

    inp = [None] * len(blocks)
    inp[0] = {}
    work = [0]
    rounds = 0
    while work:
        b = work.pop()
        rounds += 1
        if rounds > 200000:
            raise RuntimeError("the data flow did not converge")
        out = transfer(inp[b], *blocks[b], lambda *a: None)
        for n in succ[b]:
            merged = join(inp[n], out)
            if merged != inp[n]:
                inp[n] = merged
                if n not in work:
                    work.append(n)
    found = {}

    def report(ins, reg, st):
        found.setdefault((ins.line, reg), dict(line=ins.line, instr=ins.text, vgpr=reg, load_line=st[1], younger=st[0]))

    for b, (s, e) in enumerate(blocks):
        if inp[b] is not None:
            transfer(inp[b], s, e, report)
    # a kernel of this family must not leave the register file at all: spill traffic is vector-memory traffic the
    # hand-counted waits do not know about, and an out-of-line call moves array arguments to scratch memory
    for ins in instrs:
        if ins.op.startswith("scratch_") or ins.op == "s_swappc_b64":
            found.setdefault((ins.line, -1), dict(line=ins.line, instr=ins.text, vgpr=-1, load_line=0, younger=0,
                                                  kind="scratch memory / call"))
            break
    return sorted(found.values(), key=lambda d: (d["line"], d["vgpr"]))


def check_listing(path: Path) -> list[dict]:
    out = []
    for name, body in split_kernels(path.read_text().splitlines()):
        for h in analyse(body):
            h["kernel"] = name
            out.append(h)
    return out


def compile_unit(tu: str, pz: int, workdir: Path) -> Path:
    define, _ = UNITS[tu]
    out = workdir / f"{tu}_pz{pz}.s"
    cmd = [HIPCC, *FLAGS, f"-D{define}={pz}", str(CSRC / f"{tu}.hip"), "-o", str(out)]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"{' '.join(cmd)} failed:\n{proc.stderr[-2000:]}")
    return out


def check_unit(tu: str, pz: int, workdir: Path) -> tuple[str, int, int, list[dict]]:
    listing = compile_unit(tu, pz, workdir)
    kernels = split_kernels(listing.read_text().splitlines())
    return tu, pz, len(kernels), check_listing(listing)


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--tu", action="append", choices=sorted(UNITS), help="translation unit(s); default: all")
    ap.add_argument("--pz", action="append", type=int, help="z tap count(s); default: all compiled")
    ap.add_argument("--asm", help="check an existing assembly listing instead of compiling")
    ap.add_argument("--jobs", type=int, default=4)
    args = ap.parse_args()
    if args.asm:
        hazards = check_listing(Path(args.asm))
        for h in hazards:
            print(f"{h['kernel']}: line {h['line']}: `{h['instr']}` touches v{h['vgpr']}, destination of the load at line "
                  f"{h['load_line']} ({h['younger']} younger operations)")
        print(f"{args.asm}: {len(hazards)} hazard(s)")
        return 1 if hazards else 0
    todo = [(tu, pz) for tu in (args.tu or sorted(UNITS)) for pz in UNITS[tu][1] if not args.pz or pz in args.pz]
    total = 0
    with tempfile.TemporaryDirectory(prefix="lsr_asm_") as tmp, ThreadPoolExecutor(max(1, args.jobs)) as pool:
        for tu, pz, nk, hazards in pool.map(lambda a: check_unit(*a, Path(tmp)), todo):
            print(f"{tu}.hip pz={pz}: {nk} kernels, {len(hazards)} hazard(s)")
            for h in hazards[:20]:
                if h.get("kind"):
                    print(f"    {h['kernel']}: line {h['line']}: `{h['instr']}`: {h['kind']}")
                    continue
                print(f"    {h['kernel']}: line {h['line']}: `{h['instr']}` touches v{h['vgpr']}, destination of the load at "
                      f"line {h['load_line']} ({h['younger']} younger operations)")
            if len(hazards) > 20:
                print(f"    ... and {len(hazards) - 20} more")
            total += len(hazards)
    print(f"{len(todo)} translation units checked, {total} hazard(s)")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
