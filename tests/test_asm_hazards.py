"""tools/asm_hazards.py: the static check that no instruction touches a VGPR a vector-memory load in flight is going to
write (the hand-pipelined kernels wait with hand-counted ``s_waitcnt vmcnt(N)``; the hardware has no interlock).  The
analyser on small listings with a known answer, then the compiler's actual output for two translation units -- the
instances round 4 found broken on the GPU before the tool existed."""

import shutil
import sys

from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))

import asm_hazards as ah  # noqa: E402


def _kernel(lines):
    body = ["_Z6kernelv:"] + ["\t" + ln if not ln.startswith(".LBB") else ln for ln in lines] + ["\ts_endpgm", ".Lfunc_end0:"]
    return ah.split_kernels(body)[0][1]


def test_a_write_to_the_destination_of_a_load_in_flight_is_reported():
    h = ah.analyse(_kernel(["global_load_dword v5, v1, s[0:1]", "v_mov_b32_e32 v5, 0", "s_waitcnt vmcnt(0)"]))
    assert [(x["vgpr"], x["instr"].split()[0]) for x in h] == [(5, "v_mov_b32_e32")]
    # ... and so is a read before the wait; after the wait both are fine
    h = ah.analyse(_kernel(["global_load_dwordx4 v[4:7], v1, s[0:1]", "v_add_f32_e32 v9, v6, v6", "s_waitcnt vmcnt(0)",
                            "v_add_f32_e32 v9, v6, v6", "v_mov_b32_e32 v4, 0"]))
    assert [(x["vgpr"], x["line"]) for x in h] == [(6, 3)]


def test_vmcnt_retires_loads_in_issue_order_and_stores_count():
    base = ["global_load_dword v5, v1, s[0:1]", "global_load_dword v6, v1, s[0:1] offset:4",
            "global_store_dword v1, v9, s[2:3]"]
    # two operations are younger than the load of v5: vmcnt(2) has retired it, vmcnt(3) has not
    assert ah.analyse(_kernel(base + ["s_waitcnt vmcnt(2)", "v_mov_b32_e32 v5, 0"])) == []
    assert len(ah.analyse(_kernel(base + ["s_waitcnt vmcnt(3)", "v_mov_b32_e32 v5, 0"]))) == 1
    assert len(ah.analyse(_kernel(base + ["s_waitcnt vmcnt(2)", "v_mov_b32_e32 v6, 0"]))) == 1      # v6 is younger
    # LDS-DMA loads have no register destination but count; a reload of the same register is in order
    assert ah.analyse(_kernel(["global_load_dword v5, v1, s[0:1]", "global_load_lds_dwordx4 v1, s[0:1]",
                               "global_load_dword v5, v1, s[0:1]", "s_waitcnt vmcnt(0)", "v_mov_b32_e32 v5, 0"])) == []
    # a combined wait, and lgkmcnt alone retires nothing
    assert ah.analyse(_kernel(base[:1] + ["s_waitcnt vmcnt(0) lgkmcnt(0)", "v_mov_b32_e32 v5, 0"])) == []
    assert len(ah.analyse(_kernel(base[:1] + ["s_waitcnt lgkmcnt(0)", "v_mov_b32_e32 v5, 0"]))) == 1


def test_the_state_follows_branches_and_loops_on_the_conservative_side():
    # the wait sits on ONE arm of a branch: the other arm still has the load in flight at the join
    h = ah.analyse(_kernel(["global_load_dword v5, v1, s[0:1]", "s_cbranch_scc1 .LBB0_2", "s_waitcnt vmcnt(0)", ".LBB0_2:",
                            "v_mov_b32_e32 v5, 0"]))
    assert len(h) == 1
    # a store on one arm only must not count towards retiring the load (fewest younger operations over the paths)
    h = ah.analyse(_kernel(["global_load_dword v5, v1, s[0:1]", "s_cbranch_scc1 .LBB0_2", "global_store_dword v1, v9, s[2:3]",
                            ".LBB0_2:", "s_waitcnt vmcnt(1)", "v_mov_b32_e32 v5, 0"]))
    assert len(h) == 1
    # loop-carried: the prefetch issued at the bottom of the body is in flight at its top
    h = ah.analyse(_kernel([".LBB0_1:", "v_add_f32_e32 v9, v5, v9", "s_waitcnt vmcnt(0)", "global_load_dword v5, v1, s[0:1]",
                            "s_cbranch_scc1 .LBB0_1"]))
    assert [(x["vgpr"], x["instr"].split()[0]) for x in h] == [(5, "v_add_f32_e32")]
    # the undefined high half of a widened 32-bit offset is not a use
    assert ah.analyse(_kernel(["global_load_dword v29, v1, s[0:1]", "v_mad_u64_u32 v[58:59], s[0:1], v64, s3, v[28:29]",
                               "s_waitcnt vmcnt(0)"])) == []


@pytest.mark.skipif(shutil.which(ah.HIPCC) is None, reason="needs hipcc")
@pytest.mark.parametrize("tu,pz", [("correlate_sep", 13), ("rl_fused_sep", 5), ("correlate_dense", 11)])
def test_the_compilers_output_for_the_instances_that_were_broken_is_clean(tmp_path, tu, pz):
    """correlate_sep <13,9,9,UPDATE>: LDS offsets computed in the destination of a prologue load ("=v" operands);
    rl_fused_sep <5,15,STATS>: the reduction's lane arithmetic above the final wait; correlate_dense <11,9,UPDATE>: the
    iteration lambda was not inlined, its staging registers lived in scratch memory and were stored there before the
    loads into them had landed.  `python tools/asm_hazards.py` checks all 24 translation units (profiles/r04_asm_hazards.txt)."""
    _, _, kernels, hazards = ah.check_unit(tu, pz, tmp_path)
    assert kernels >= 8
    assert hazards == [], hazards[:5]
    # none of these kernels may leave the register file: a hand-counted wait does not cover the compiler's scratch traffic
    text = (tmp_path / f"{tu}_pz{pz}.s").read_text()
    assert "scratch_store" not in text and "s_swappc_b64" not in text


@pytest.mark.skipif(shutil.which(ah.HIPCC) is None, reason="needs hipcc")
def test_every_stencil_translation_unit_is_free_of_hazards(tmp_path):
    """the three instances above guard what broke once; a toolchain bump or a new specialisation can bring the
    hazard back anywhere.  The whole sweep -- every SEP / DENSE / FUSED / YSEP translation unit the Makefile builds, loop
    bodies included (the analyser carries the in-flight loads around back edges) -- runs here, eight compilations side by
    side (~1 minute), and as ``make -C shrimpy_amd/csrc hazards``."""
    from concurrent.futures import ThreadPoolExecutor

    todo = [(tu, pz) for tu in sorted(ah.UNITS) for pz in ah.UNITS[tu][1]]
    assert len(todo) == 24
    with ThreadPoolExecutor(8) as pool:
        results = list(pool.map(lambda a: ah.check_unit(*a, tmp_path), todo))
    bad = {f"{tu} pz={pz}": hazards[:3] for tu, pz, _, hazards in results if hazards}
    assert not bad, bad
    assert sum(nk for _, _, nk, _ in results) >= 24 * 8
    # the Makefile's tap counts and the tool's are the same lists
    mk = (ROOT / "shrimpy_amd" / "csrc" / "Makefile").read_text()
    for var, unit in (("SEP_PZ", "correlate_sep"), ("DENSE_PZ", "correlate_dense"), ("FUSED_PZ", "rl_fused_sep"), ("YSEP_PZ", "rl_fused_ysep")):
        line = next(ln for ln in mk.splitlines() if ln.startswith(var + " :="))
        assert tuple(int(v) for v in line.split(":=")[1].split()) == ah.UNITS[unit][1], var
