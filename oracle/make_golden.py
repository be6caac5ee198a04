#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference's own kernels (oracle/_ref/libmg_ref.so).

Run in the build container, where /root/reference is present:
    make -C oracle ref && python oracle/make_golden.py
Each fixture holds inputs-by-recipe (problem id, extents, parameters), the residual history
||f - A u||_2 (exact, fp64 accumulation) before the first cycle and after each one, and the final
solution field.  TEST INFRASTRUCTURE ONLY.
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

# (name, problem, shape, dtype, zfull, omega, a1, a2, v, cycles, mode)
CASES = []
for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
    for n in (17, 33, 65):
        CASES.append(("2d_%d_%s_v22" % (n, tag), 1, (n, n), dt, 0, 1.0, 2, 2, 1, 6, "mg"))
    for n in (5, 9, 17, 33):
        for zfull in (0, 1):
            CASES.append(("3d_%d_%s_v22_z%s" % (n, tag, "full" if zfull else "ref"), 0, (n, n, n), dt, zfull,
                          1.0, 2, 2, 1, 6, "mg"))
    # class-default omega (multigridsolver3D.h:44), W-cycle, uneven extents
    CASES.append(("3d_17x9x33_%s_w_omega23" % tag, 0, (33, 9, 17), dt, 1, 2.0 / 3.0, 3, 1, 2, 3, "mg"))
    CASES.append(("2d_33x17_%s_w_omega23" % tag, 1, (17, 33), dt, 0, 2.0 / 3.0, 3, 1, 2, 3, "mg"))
    # the program's default solve: FMG, a1=a2=3, v=2, iterations=3 (programstate.cpp:150-163)
    CASES.append(("3d_17_%s_fmg_default" % tag, 0, (17, 17, 17), dt, 0, 1.0, 3, 3, 2, 3, "fmg"))
    CASES.append(("3d_17_%s_fmg_zfull" % tag, 0, (17, 17, 17), dt, 1, 1.0, 3, 3, 2, 3, "fmg"))
    CASES.append(("2d_33_%s_fmg_default" % tag, 1, (33, 33), dt, 0, 1.0, 3, 3, 2, 3, "fmg"))
# the CPU-runnable case of the benchmark set: 2D 257^2 Dirichlet V(2,2), float: history plus field
CASES.append(("2d_257_f32_v22", 1, (257, 257), np.float32, 0, 1.0, 2, 2, 1, 6, "mg"))


def main():
    ref = _oracle.load("mgr")
    if ref is None:
        sys.exit("oracle/_ref/libmg_ref.so is missing: run `make -C oracle ref` where /root/reference exists")
    os.makedirs(OUT, exist_ok=True)
    for name, prob, shape, dt, zfull, omega, a1, a2, v, cycles, mode in CASES:
        f, sol = ref.discretize(prob, shape, dt)
        cell = ref.box_cells(shape)
        res = np.zeros(shape, dt)
        hist = []
        if mode == "mg":
            u = np.zeros(shape, dt)
            ref.residual(cell, res, u, f)
            hist.append(ref.l2norm_exact(res))
            for _ in range(cycles):
                ref.mg(u, f, dt(omega), a1, a2, v, 1, zfull)
                ref.residual(cell, res, u, f)
                hist.append(ref.l2norm_exact(res))
        else:
            u = ref.fmg(f, dt(omega), a1, a2, v, cycles, zfull)
            ref.residual(cell, res, u, f)
            hist.append(ref.l2norm_exact(res))
        err = ref.diff(u, sol)
        np.savez_compressed(
            os.path.join(OUT, name + ".npz"),
            problem=prob, shape=np.array(shape), zfull=zfull, omega=omega, a1=a1, a2=a2, v=v, cycles=cycles,
            mode=mode, func=f, history=np.array(hist, dtype=np.float64), u=u,
            l2err_tree=ref.l2norm_tree(err), linferr_tree=ref.linfnorm_tree(err),
            l2res_tree=ref.l2norm_tree(res), linfres_tree=ref.linfnorm_tree(res))
        print("%-28s %s" % (name, " ".join("%.6e" % h for h in hist)))


if __name__ == "__main__":
    main()
