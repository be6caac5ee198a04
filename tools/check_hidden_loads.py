#!/usr/bin/env python3
"""Audit of the fused smoother's generated code: loads issued from inline asm are invisible to the
compiler, so a register with such a load pending must not be read by compiler-generated code before
the kernel's own s_waitcnt.  The source guarantees this everywhere except inside the inner
(fast) time loop, where it depends on the register allocator not copying the prefetch registers.
This script compiles the file to assembly and fails if, inside any depth-2 loop of a
k_smooth3d_fused kernel, a compiler-generated instruction reads a register that an asm load in
that loop writes before a wait has retired that load (the counter is modelled: operations retire
in issue order, vmcnt(N) leaves the youngest N in flight).  Second check: no memory instruction from inline asm
reads a scalar register (its descriptor) fewer than five wait states after a vector-ALU instruction wrote it -- the
compiler pads its own memory instructions for that, but does not look inside asm (sgpr_hazards below).

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


def sregs(tok):
    """s12 or s[12:15] -> {12, ..., 15}"""
    m = re.fullmatch(r"s(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def sgpr_hazards(text):
    """A vector-memory instruction must not read a scalar register that a vector-ALU instruction wrote fewer than
    five wait states earlier; the compiler pads its own memory instructions with s_nop, but it does not look inside
    an asm block.  For every buffer instruction that comes from inline asm: walk back through up to five wait states
    (an instruction is one, s_nop N is N + 1) along every path that leads there -- the fall-through and every branch
    to a label on the way -- and report a v_ instruction whose scalar destination overlaps the descriptor."""
    findings, kernel, ins, labels, in_asm = [], None, [], {}, False

    def check_kernel():
        targets = {}
        for i, (txt, _) in enumerate(ins):
            m = re.match(r"s_c?branch\w*\s+(\S+)", txt)
            if m:
                targets.setdefault(m.group(1), []).append(i)
        label_at = {}
        for name, pos in labels.items():
            label_at.setdefault(pos, []).append(name)

        def walk(i, budget, desc, seen, out):
            """instructions before position i (exclusive)"""
            while budget > 0:
                for name in label_at.get(i, []):  # a label sits in front of instruction i: its other predecessors
                    for b in targets.get(name, []):
                        if (b, budget) not in seen:
                            seen.add((b, budget))
                            walk(b + 1, budget, desc, seen, out)
                i -= 1
                if i < 0:
                    return
                txt = ins[i][0]
                op = txt.split()[0]
                if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
                    return  # nothing falls through from here
                if op.startswith("v_"):
                    ops = [o.strip() for o in txt[len(op):].split(",")]
                    dst = ops[:2] if ("_co_" in op or op.startswith(("v_div_scale", "v_mad_u64", "v_mad_i64"))) else ops[:1]
                    for d in dst:
                        if sregs(d) & desc:
                            out.append(txt)
                m = re.match(r"s_nop\s+(\d+)", txt)
                budget -= int(m.group(1)) + 1 if m else 1

        for i, (txt, asm) in enumerate(ins):
            if not asm:
                continue
            m = re.match(r"buffer_\w+\s+\S+,\s*\S+,\s*(s\[\d+:\d+\])", txt)
            if not m:
                continue
            out = []
            walk(i, 5, sregs(m.group(1)), set(), out)
            for w in out:
                findings.append((kernel, "%s   <- fewer than 5 wait states after   %s" % (txt, w)))

    for line in text:
        m = re.match(r"^(_Z\S*k_smooth3d_fused\S*):", line)
        if m:
            kernel, ins, labels = m.group(1), [], {}
            continue
        if kernel is None:
            continue
        if ";;#ASMSTART" in line:
            in_asm = True
            continue
        if ";;#ASMEND" in line:
            in_asm = False
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        t = line.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        ins.append((t, in_asm))
        if t.startswith("s_endpgm"):
            check_kernel()
            kernel = None
    return findings


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
    hazards = sgpr_hazards(text)
    for k, l in hazards:
        print("check_hidden_loads: %s: scalar-register hazard: %s" % (k[:70], l))
    if os.environ.get("MGRT_AUDIT_VERBOSE"):
        for k, n in sorted(per_kernel.items()):
            print("check_hidden_loads: %d inner loop(s) in %s" % (n, k[:110]))
    if nk == 0 or loops == 0:
        print("check_hidden_loads: found %d kernels, %d inner loops -- the audit did not run" % (nk, loops))
        return 1
    for k, l in bad:
        print("check_hidden_loads: %s: use of a register with a hidden load pending: %s" % (k[:70], l))
    print("check_hidden_loads: %d kernels, %d inner loops, %d findings" % (nk, loops, len(bad) + len(hazards)))
    return 1 if bad or hazards else 0


if __name__ == "__main__":
    sys.exit(main())
