#!/usr/bin/env python3
"""Audit of the fused smoother's generated code: loads issued from inline asm are invisible to the
compiler, so a register with such a load pending must not be read by compiler-generated code before
the kernel's own s_waitcnt.  The source guarantees this everywhere except inside the inner
(fast) time loop, where it depends on the register allocator not copying the prefetch registers.
This script compiles the file to assembly and fails if, inside any depth-2 loop of a
k_smooth3d_fused kernel, a compiler-generated instruction reads a register that an asm load in
that loop writes before a wait has retired that load (the counter is modelled: operations retire
in issue order, vmcnt(N) leaves the youngest N in flight).

Usage: tools/check_hidden_loads.py [hipcc flags...]   (exit status 1 on a finding)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRCS = [os.path.join(ROOT, "multigrid_amd", "csrc", n) for n in ("mgrt_smooth_fused.hip", "mgrt_smooth_pair.hip")]


# The double-precision sweep on the corrected field (<double, ITERS 2, SR 2, NW 12, PD 2, ..., COR>) is allowed its
# scratch memory: the loop model below sees the compiler's scratch operations like every other instruction, so a spill
# of a register with a load pending would still be a finding.
SCRATCH_ALLOWED = re.compile(r"k_smooth3d_fusedIdLi2ELi2ELi12ELi2ELb[01]ELb0ELb1E")


def regs(tok):
    """v12 or v[12:13] -> {12, 13}"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def main():
    text = []
    with tempfile.TemporaryDirectory() as tmp:
        for src in SRCS:
            out = os.path.join(tmp, os.path.basename(src) + ".s")
            cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                   "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", out, src] + sys.argv[1:]
            subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
            text += open(out).read().splitlines()
    kernel, bad, nk, loops = None, [], 0, 0
    per_kernel = {}
    in_asm, depth2 = False, False
    loop_lines = []

    def flush():
        """Replays the loop body twice (the second time with the operations the first left in flight)
        against a model of the vector-memory counter: operations retire in issue order, a wait with
        vmcnt(N) retires all but the youngest N.  Any instruction outside the asm blocks that names a
        register whose asm load has not retired yet is a finding."""
        nonlocal loop_lines, loops
        if not loop_lines:
            return
        loops += 1
        per_kernel[kernel] = per_kernel.get(kernel, 0) + 1
        queue, pending, seen = [], set(), set()
        for _ in range(2):
            for asm, line in loop_lines:
                text_ = line.split(";")[0].strip()
                if not text_ or text_.endswith(":") or text_.startswith("."):
                    continue
                op = text_.split()[0]
                m = re.match(r"s_waitcnt\s+(.*)", text_)
                if m:
                    n = re.search(r"vmcnt\((\d+)\)", m.group(1))
                    if n:
                        while len(queue) > int(n.group(1)):
                            pending -= queue.pop(0)
                    continue
                is_vmem = op.startswith(("buffer_", "global_", "flat_", "scratch_"))
                if asm:
                    if is_vmem:
                        m = re.match(r"buffer_load_(?:dword(?:x2)?|ubyte)\s+(\S+),", text_)
                        d = regs(m.group(1)) if m else set()
                        queue.append(d)
                        pending |= d
                    continue
                named = set()
                for tok in re.findall(r"v\[\d+:\d+\]|v\d+", text_):
                    named |= regs(tok)
                if named & pending and (kernel, text_) not in seen:
                    seen.add((kernel, text_))
                    bad.append((kernel, text_))
                if is_vmem:
                    queue.append(set())
        loop_lines = []

    for line in text:
        m = re.match(r"^(_Z\S*k_smooth3d_fused\S*):", line)
        if m:
            flush()
            kernel, depth2 = m.group(1), False
            nk += 1
            continue
        if kernel is None:
            continue
        if "s_endpgm" in line:
            flush()
            kernel = None
            continue
        if ";;#ASMSTART" in line:
            in_asm = True
            continue
        if ";;#ASMEND" in line:
            in_asm = False
            continue
        if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", line):
            # block labels carry the loop annotation on this or the next comment line
            depth2_here = "Depth=2" in line
            if not depth2_here and depth2:  # a block of the enclosing loop, or outside every loop
                flush()
                depth2 = False
            if depth2_here:
                depth2 = True
        elif "Inner Loop Header: Depth=2" in line or ("in Loop:" in line and "Depth=2" in line):
            depth2 = True
        if depth2:
            loop_lines.append((in_asm, line))
    flush()
    # A kernel that spills would move prefetch registers through scratch memory behind the counted
    # waits: the metadata must show no private segment for any of these kernels.
    name = None
    for line in text:
        m = re.match(r"\s+\.name:\s+(\S*k_smooth3d_fused\S*)", line)
        if m:
            name = m.group(1)
        m = re.match(r"\s+\.private_segment_fixed_size:\s+(\d+)", line)
        if m:
            if name and int(m.group(1)) > 0 and not SCRATCH_ALLOWED.search(name):
                bad.append((name, "uses %s bytes of scratch memory per lane" % m.group(1)))
            name = None
    if os.environ.get("MGRT_AUDIT_VERBOSE"):
        for k, n in sorted(per_kernel.items()):
            print("check_hidden_loads: %d inner loop(s) in %s" % (n, k[:110]))
    if nk == 0 or loops == 0:
        print("check_hidden_loads: found %d kernels, %d inner loops -- the audit did not run" % (nk, loops))
        return 1
    for k, l in bad:
        print("check_hidden_loads: %s: use of a register with a hidden load pending: %s" % (k[:70], l))
    print("check_hidden_loads: %d kernels, %d inner loops, %d findings" % (nk, loops, len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
