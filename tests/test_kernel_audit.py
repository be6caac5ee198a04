"""Static checks on the generated code of the fused smoother (no GPU needed: hipcc cross-compiles).

The kernel prefetches with loads the compiler cannot see; tools/check_hidden_loads.py verifies on
the assembly that no compiler-generated instruction copies a register while such a load is pending
inside the time loop and that no variant spills to scratch memory."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not installed")
def test_fused_smoother_has_no_copies_of_inflight_registers():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_hidden_loads.py")],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert " 0 findings" in p.stdout


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not installed")
def test_2d_tile_kernels_do_not_spill(tmp_path):
    """The 2D smoother keeps a whole tile per wave in registers; a variant that spilled would move
    it through scratch memory.  The code-object metadata must show no private segment and no
    spilled scalar registers for any k_smooth2d_cols instantiation."""
    out = tmp_path / "smooth2d.s"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                    "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", str(out),
                    os.path.join(ROOT, "multigrid_amd", "csrc", "mgrt_smooth2d.hip")],
                   check=True, capture_output=True, timeout=900)
    import re
    name, seen, bad = None, 0, []
    for line in out.read_text().splitlines():
        m = re.match(r"\s+\.name:\s+(\S*k_smooth2d_cols\S*)", line)
        if m:
            name = m.group(1)
            seen += 1
        m = re.match(r"\s+\.(private_segment_fixed_size|sgpr_spill_count|vgpr_spill_count):\s+(\d+)", line)
        if m and name and int(m.group(2)) > 0:
            bad.append((name, m.group(1), m.group(2)))
        if re.match(r"\s+\.wavefront_size:", line):
            name = None
    assert seen >= 16, "no k_smooth2d_cols kernels found in the assembly"
    assert not bad, bad[:5]


def test_scalar_register_hazard_check_finds_what_it_should():
    """The audit's second check on synthetic listings: a vector-ALU write of a descriptor register fewer than five
    wait states before a buffer instruction from inline asm is reported -- directly in front, across a branch to the
    block of the instruction -- and not when s_nop pads it, when a scalar instruction wrote the register, or when the
    buffer instruction is the compiler's own (which the compiler pads itself)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import check_hidden_loads as audit
    finally:
        sys.path.pop(0)

    def listing(body):
        return ["_ZN1x16k_smooth3d_fusedIfEEvv:"] + body + ["\ts_endpgm"]

    store = ["\t;;#ASMSTART", "\tbuffer_store_dword v0, v1, s[8:11], 0 offen", "\t;;#ASMEND"]
    assert len(audit.sgpr_hazards(listing(["\tv_readfirstlane_b32 s9, v3", "\tv_add_f32_e32 v0, v0, v0"] + store))) == 1
    assert audit.sgpr_hazards(listing(["\tv_readfirstlane_b32 s9, v3", "\ts_nop 4"] + store)) == []
    assert audit.sgpr_hazards(listing(["\tv_readfirstlane_b32 s12, v3"] + store)) == []
    assert audit.sgpr_hazards(listing(["\ts_add_u32 s8, s8, s20", "\ts_addc_u32 s9, s9, s21"] + store)) == []
    assert audit.sgpr_hazards(listing(["\tv_readfirstlane_b32 s9, v3", "\tbuffer_store_dword v0, v1, s[8:11], 0 offen"])) == []
    # the write sits at the end of another block that branches to the store's block
    across = ["\tv_cmp_lt_f32_e64 s[10:11], v0, v1", "\ts_cbranch_scc1 .LBB0_2", "\ts_nop 7", "\ts_branch .LBB0_3", ".LBB0_2:"] + store + [".LBB0_3:"]
    assert len(audit.sgpr_hazards(listing(across))) == 1
    # carry-out destinations count as well
    assert len(audit.sgpr_hazards(listing(["\tv_add_co_u32_e64 v2, s[8:9], v2, v3"] + store))) == 1
