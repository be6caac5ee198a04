#!/usr/bin/env python3
"""Benchmark of the V-cycle engine on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config NAME]

A step is one V(2,2) cycle (omega = 1, pre/post smoothing 2+2, one coarse solve per level, 3D
smoother over every z plane) on fields already resident in HBM.  Prints ONE JSON line:
value = finest-grid points x cycles / second over all GPUs.

--config picks the workload (default 3d513, the headline: 3D 513^3 Poisson, Dirichlet box, float):
2d257, 2d4097mixed, 3d257neumann, 3d513, 3d513f64, 3d1025f64.  Every line carries the roofline
of that workload's dominant kernel and the CPU oracle timed on the same extents.

For N > 1 there is one process per GPU (3d513 only): started by the caller with torch.distributed.run,
or, when WORLD_SIZE is not set, started here as child processes before anything touches a GPU.  The
grid is cut into N slabs along y (mgrt_slab_plan of libmgrt.so: halos over RCCL, a whole cycle of a
rank replayed from one hipGraph); total work is fixed: strong scaling.  After that timed region a second
one, reported beside the line's value as "grown_grid", runs the same plan on 513 x (512 N + 1) x 513 points
(every rank keeps 512 rows: weak scaling), and a third the 1025^3 double-precision grid cut N ways,
reported as "fixed_grid_1025_f64" (--no-weak skips both).  On a box with fewer than N GPUs
the ranks share the devices and the halos go through host callbacks over gloo (a rehearsal of the path,
marked as such in the output, not a measurement).  Each rank does its work in a child process and watches
it: when an attempt fails or stops making progress on any rank, all ranks go on to the next one (RCCL with
the cycle in a graph, RCCL with direct launches, the same without the extra timed regions, host callbacks),
and the line says which attempts failed.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak

D_, N_ = 2, 3  # MGRT_CELL_DIRICHLET, MGRT_CELL_NEUMANN
CONFIGS = {
    # name: (shape (nz, ny, nx) or (ny, nx), dtype, faces x-lo x-hi y-lo y-hi z-lo z-hi, oracle problem, workload)
    "2d257": ((257, 257), "f32", (D_,) * 6, 1, "2D Poisson 257^2 Dirichlet box"),
    "2d4097mixed": ((4097, 4097), "f32", (D_, D_, N_, N_, D_, D_), 1,
                    "2D Poisson 4097^2, Dirichlet x faces, Neumann (frozen, as in the reference) y faces"),
    "3d257neumann": ((257,) * 3, "f32", (N_,) * 6, 0, "3D Poisson 257^3, all faces Neumann (frozen, as in the reference)"),
    "3d513": ((513,) * 3, "f32", (D_,) * 6, 0, "3D Poisson 513^3 Dirichlet box"),
    "3d513f64": ((513,) * 3, "f64", (D_,) * 6, 0, "3D Poisson 513^3 Dirichlet box"),
    "3d1025f64": ((1025,) * 3, "f64", (D_,) * 6, 0, "3D Poisson 1025^3 Dirichlet box"),
}


def make_problem(shape, dtype, faces, device, rows=None):
    """The reference's test problems sampled as discretize_func does (functionhandler.cpp:25-45,64-84):
    3D tridimF1/tridimS1 (programstate.cpp:114-122), 2D sinfunc2/sinfunc2sol (:84-94).  Interior
    F*h*h in `real`, Dirichlet cells the exact solution, Neumann cells the solution times h.
    rows = (lo, hi): only those y rows of the 3D field (a rank's slab), the same values."""
    import math
    import torch
    # Coordinates i/(n-1) in `real`, widened to double for the function bodies (M_PI makes them double).
    # The sines and cosines of an axis are taken by the C library on the host, as the reference takes
    # them, so that the field equals the oracle's bit for bit in either precision; the products follow the
    # reference's left-to-right order and run on the device (IEEE double, no contraction between torch ops).
    ax = [(torch.arange(n, dtype=dtype) / (n - 1)).double().tolist() for n in shape]
    ylo, yhi = rows if rows else (0, shape[-2])
    if rows:
        ax[-2] = ax[-2][ylo:yhi]
    pi = math.pi

    def on_axis(fn, axis):
        t = torch.tensor([fn(c) for c in ax[axis]], dtype=torch.float64, device=device)
        return t.reshape([-1 if a == axis else 1 for a in range(len(shape))])

    if len(shape) == 3:
        sz, sy, sx = (on_axis(lambda c: math.sin(pi * c), a) for a in range(3))
        sol = sx * sy * sz                          # tridimS1: sin(pi x) * sin(pi y) * sin(pi z)
        rhs = (-3.0 * pi * pi) * sx * sy * sz       # tridimF1: -3.0 * M_PI * M_PI * sin * sin * sin
    else:
        s26 = on_axis(lambda c: math.sin(26 * pi * c), 1)
        c50 = on_axis(lambda c: math.cos(50 * pi * c), 0)
        s1 = on_axis(lambda c: math.sin(pi * c), 1)
        # sinfunc2sol spells its arguments M_PI*26*x and M_PI*50*y: the same doubles, 26 and 50 being exact
        sol = (s26 + c50 + s1).expand(len(ax[0]), len(ax[1]))
        rhs = ((-26 * 26 * pi * pi) * s26 - (50 * 50 * pi * pi) * c50 - (pi * pi) * s1).expand(len(ax[0]), len(ax[1]))
    ny = shape[-2]
    dh = torch.tensor(1.0 / (shape[-1] - 1), dtype=dtype, device=device)
    f = rhs.to(dtype) * dh * dh
    sol = sol.to(dtype)
    # faces in the reference's fill order (rectangularborderhandler.cpp:59-62,103-108): later ones win
    nd = len(shape)
    order = [(nd - 3, 4), (nd - 3, 5), (nd - 1, 0), (nd - 1, 1), (nd - 2, 2), (nd - 2, 3)] if nd == 3 else \
        [(0, 2), (0, 3), (1, 0), (1, 1)]
    for axis, face in order:
        if axis == nd - 2 and (ylo > 0 if face % 2 == 0 else yhi < ny):
            continue  # that face of the grid is not among these rows
        idx = [slice(None)] * nd
        idx[axis] = 0 if face % 2 == 0 else -1
        idx = tuple(idx)
        f[idx] = sol[idx] if faces[face] == D_ else sol[idx] * dh
    return f.contiguous()


def pmc_traffic(kernel, n, dtype, fields):
    """HBM bytes per launch of `kernel` at extent n from the committed counter passes (profiles/
    pmc_traffic.json, written by tools/pmc_traffic.py from separate FETCH_SIZE and WRITE_SIZE
    rocprofv3 runs).  `fields` maps positions in the kernel's template argument list to the values
    that pick the variant timed here.  None when no pass for this size is on file: counters cannot
    be read from inside the timed process."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            for row in json.load(fh):
                args = [a.strip() for a in row["template"].split(",")]
                if row["kernel"] == kernel and row["n"] == n and row["dtype"] == dtype and \
                        all(i < len(args) and args[i] == v for i, v in fields.items()):
                    return row["traffic_bytes"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def cpu_threads():
    """Threads for the CPU leg: what the host grants this job.  OMP_NUM_THREADS, where the box sets it,
    is this job's share of a machine whose affinity mask shows every core; otherwise the mask."""
    n = len(os.sched_getaffinity(0))
    try:
        n = min(n, int(os.environ["OMP_NUM_THREADS"]))
    except (KeyError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(name, shape=None, budget_s=12.0):
    """The oracle's C restatement (oracle/libmg_oracle.so, OpenMP) timed on this host on the same
    extents (`shape` when --n overrides the configuration's), dtype and problem: V(2,2) cycles from a
    zero guess for about `budget_s` seconds."""
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle
    o = _oracle.load("mgo")
    if o is None:
        return None
    cfg_shape, dt, faces, problem, _ = CONFIGS[name]
    shape = tuple(shape) if shape else cfg_shape
    npdt = np.float32 if dt == "f32" else np.float64
    threads = cpu_threads()
    share = ""
    if "OMP_NUM_THREADS" in os.environ and threads < len(os.sched_getaffinity(0)):
        share = "; %d threads = this job's share of the host (OMP_NUM_THREADS), whose affinity mask shows %d CPUs" % (
            threads, len(os.sched_getaffinity(0)))
    o.set_num_threads(threads)
    note = ""
    if faces != (D_,) * 6:
        o.set_box_faces(faces)  # the box with these face types on every level, as RectangularBorderHandler paints it
    f, _ = o.discretize(problem, shape, npdt)
    u = np.zeros(shape, npdt)
    t0 = time.perf_counter()
    o.mg(u, f, npdt(1.0), 2, 2, 1, 1, True)  # warm: page in, spin threads up
    first = time.perf_counter() - t0
    reps, t0 = 0, time.perf_counter()
    while reps < 1 or (time.perf_counter() - t0 + first < budget_s and reps < 1000):
        o.mg(u, f, npdt(1.0), 2, 2, 1, 1, True)
        reps += 1
    dt_s = (time.perf_counter() - t0) / reps
    o.set_box_faces(None)
    dof = float(np.prod(shape))
    return {"value": dof / dt_s, "unit": "DOF/s", "cores": threads, "kind": "port",
            "sample": "%d V(2,2) cycles of the same problem at %s (%s, full-z), %.3f s each%s%s"
                      % (reps, "x".join(map(str, shape)), dt, dt_s, note, share)}


def spawn_ranks(args, argv):
    """python bench.py --gpus N with no launcher around it: start N ranks as children (nothing in this
    process has touched a GPU), pass their output through and leave with their exit code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


ATTEMPTS = (
    # what a rank's worker is asked to do, most wanted first; a later one is tried when an earlier one
    # fails or stops making progress (a collective that never returns cannot be left from inside)
    ("RCCL, cycle in one graph", []),
    ("RCCL, direct launches", ["--no-graph"]),
    ("RCCL, direct launches, the line's own workload only", ["--no-graph", "--no-weak"]),
    ("host callbacks over gloo, the line's own workload only", ["--no-graph", "--no-weak", "--transport", "gloo"]),
)
PHASE_LIMIT_DEFAULT_S = float(os.environ.get("MGRT_BENCH_PHASE_LIMIT_S", "150"))
# "start": the first import of torch on a fresh box pages the image in.  "imported": the rendezvous of an
# attempt.  A rank whose worker died early in the attempt before arrives there at once and must outwait the
# ranks that first have to notice their own hang (one phase limit) and then import (one start limit): with
# anything shorter, import jitter decides whether that rank moves an attempt ahead of the others for good.
PHASE_LIMIT_S = {"start": 420.0, "imported": 420.0 + 2 * PHASE_LIMIT_DEFAULT_S}


def supervise(argv):
    """One rank of N > 1: run the work in a child (this process never touches a GPU), watch the phases it
    reports, and move every rank on to the next, more conservative attempt when one fails or hangs.  All
    ranks walk the same list: a hang in a collective stops all of them, and a rank whose worker died early
    waits in the next attempt's rendezvous for the others.  Only the output of the attempt that finished is
    passed on."""
    import tempfile
    rank = int(os.environ.get("RANK", "0"))
    failures = []
    for k, (what, extra) in enumerate(ATTEMPTS):
        with tempfile.TemporaryDirectory(prefix="mgrt_bench_") as tmp:
            phase_file = os.path.join(tmp, "phase")
            open(phase_file, "w").close()
            env = dict(os.environ, MGRT_BENCH_PHASE_FILE=phase_file, MGRT_BENCH_ATTEMPT=str(k),
                       MGRT_BENCH_EARLIER="; ".join(failures))
            with open(os.path.join(tmp, "out"), "w+") as out:
                child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker"] + argv + extra,
                                         env=env, stdout=out)
                phase, since, why = "start", time.monotonic(), None
                while child.poll() is None and phase != "done":
                    time.sleep(0.25)
                    now = open(phase_file).read().split()
                    if now and now[-1] != phase:
                        phase, since = now[-1], time.monotonic()
                    if phase != "done" and time.monotonic() - since > PHASE_LIMIT_S.get(phase, PHASE_LIMIT_DEFAULT_S):
                        why = "no progress for %.0f s in phase '%s'" % (time.monotonic() - since, phase)
                        child.kill()
                        child.wait()
                        break
                now = open(phase_file).read().split()
                if now and now[-1] == "done":
                    # The result is out: pass it on now.  How the process ends after that (closing the process
                    # group, unloading the GPU runtime) is not the bench's business and gets a few seconds.
                    out.seek(0)
                    for line in out:  # the result line to stdout; what libraries printed there goes to stderr
                        (sys.stdout if line.startswith("{") else sys.stderr).write(line)
                    sys.stdout.flush()
                    try:
                        child.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        child.kill()
                        child.wait()
                    return 0
                why = why or "worker left with status %s in phase '%s'" % (child.returncode, now[-1] if now else "start")
        failures.append("%s: %s" % (what, why))
        print("bench.py rank %d: %s; %s" % (rank, failures[-1], "trying: " + ATTEMPTS[k + 1][0] if k + 1 < len(ATTEMPTS)
                                             else "giving up"), file=sys.stderr, flush=True)
    return 1


def phase(name):
    path = os.environ.get("MGRT_BENCH_PHASE_FILE")
    if path:
        with open(path, "a") as fh:
            fh.write(name + "\n")
    # rehearsing the supervisor: MGRT_BENCH_TEST_FAIL="attempt:rank:phase-prefix:exit|hang,..."
    for item in filter(None, os.environ.get("MGRT_BENCH_TEST_FAIL", "").split(",")):
        a, r, ph, how = item.split(":")
        if a == os.environ.get("MGRT_BENCH_ATTEMPT") and r == os.environ.get("RANK") and name.startswith(ph):
            if how == "exit":
                os._exit(3)
            while True:
                time.sleep(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="3d513")
    ap.add_argument("--n", type=int, default=None, help="override the finest extent per axis (2^k+1)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 with fewer than N GPUs: ranks share the devices, gloo with host staging; checks that "
                         "the path runs, the number is not a measurement (chosen by itself when GPUs are short)")
    ap.add_argument("--no-weak", action="store_true",
                    help="N > 1: skip the separately timed legs beside the line's own (the grid that grows with the "
                         "ranks, the 1025^3 double-precision grid)")
    ap.add_argument("--crossover", type=int, default=129 * 129 * 257,
                    help="N > 1: levels with at most this many points are solved whole on every rank")
    ap.add_argument("--transport", choices=("rccl", "gloo"), default="rccl",
                    help="N > 1: how halos travel (gloo: blocking host callbacks; chosen by itself when RCCL does not come up)")
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    if args.gpus > 1 and not args.worker:
        sys.exit(supervise(sys.argv[1:]))

    import torch
    import torch.distributed as dist
    from multigrid_amd import _lib
    phase("imported")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    ndev = torch.cuda.device_count()  # counting devices does not initialise one
    if ndev < 1:
        sys.exit("bench.py: no GPU visible")
    rehearse = args.rehearse or (world > 1 and ndev < world)
    if rehearse:
        local = local % ndev
    if world > 1 and (len(CONFIGS[args.config][0]) != 3 or set(CONFIGS[args.config][2]) != {D_}):
        sys.exit("bench.py: the multi-GPU path runs the 3D Dirichlet boxes (3d513, 3d513f64, 3d1025f64)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        # every attempt of the supervisor gets a rendezvous of its own on the launcher's store
        attempt = os.environ.get("MGRT_BENCH_ATTEMPT", "0")
        if os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True":
            base = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), world, False)
        else:
            base = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]) + int(attempt), world, rank == 0)
        dist.init_process_group("gloo", store=dist.PrefixStore("mgrt_bench_attempt_" + attempt, base), rank=rank,
                                world_size=world)
    _lib.init(local)
    phase("setup")

    shape, dtname, faces, _, workload = CONFIGS[args.config]
    if args.n:
        shape = (args.n,) * len(shape)
        workload = workload.replace("%d^%d" % (CONFIGS[args.config][0][0], len(shape)), "%d^%d" % (args.n, len(shape)))
    tdt = torch.float32 if dtname == "f32" else torch.float64
    w = 4 if dtname == "f32" else 8
    n = shape[-1]
    if world == 1:
        f = make_problem(shape, tdt, faces, dev)
        u = torch.zeros_like(f)
    stream = torch.cuda.Stream()
    sync = torch.zeros(1, device=dev)
    torch.cuda.synchronize()  # the fields were produced on the default stream

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.all_reduce(sync.cpu())
            torch.cuda.synchronize()

    layout, weak, big, spread = None, None, None, None
    if world > 1:
        # mgrt_slab_plan: this rank's slab, halos over RCCL on the plan's stream, a cycle per graph launch.
        # torch.distributed (gloo) only hands the RCCL id round and closes the timed region.
        use_rccl = [not rehearse and args.transport == "rccl"]
        note = ["host callbacks over gloo" + (" (rehearsal)" if rehearse else "")]

        def make_plan(shp, tdt):
            def create(transport, rid):
                return _lib.SlabPlan(shp, tdt, transport=transport, rank=rank, nranks=world, rccl_id=rid, faces=faces,
                                     zfull=True, crossover=args.crossover, use_graph=not args.no_graph)
            cyc = None
            if use_rccl[0]:
                # every rank must end up on the same transport: agree on whether RCCL came up everywhere
                err = ""
                try:
                    box = [_lib.rccl_unique_id() if rank == 0 else None]
                except _lib.MgrtError as e:
                    box, err = [None], str(e)
                dist.broadcast_object_list(box, src=0)
                if box[0] is not None:
                    try:
                        cyc = create("rccl", box[0])
                    except _lib.MgrtError as e:
                        err = str(e)
                bad = torch.tensor([0 if cyc is not None else 1])
                dist.all_reduce(bad)
                if bad.item():
                    if cyc is not None:
                        cyc.close()
                        cyc = None
                    use_rccl[0] = False
                    note[0] = "host callbacks over gloo: RCCL did not come up on %d rank(s)%s" % (
                        bad.item(), " (" + err + ")" if err else "")
                else:
                    note[0] = "RCCL"
            return cyc if cyc is not None else create(_lib.GlooTransport(), None)

        def run_slabs(shp, tdt=tdt):
            phase("plan-%d-rows" % shp[1])
            cyc = make_plan(shp, tdt)
            fl = make_problem(shp, tdt, faces, dev, rows=cyc.rows()[:2])  # this rank's rows, halo included
            ul = torch.zeros_like(fl)
            torch.cuda.synchronize()
            ra = cyc.vcycle(ul, fl, ncycles=0, history=True)[0]
            phase("warmup-%d-rows" % shp[1])
            cyc.vcycle(ul, fl, 1.0, 2, 2, 1, args.warmup)
            barrier()
            phase("timing-%d-rows" % shp[1])
            t0 = time.perf_counter()
            cyc.vcycle(ul, fl, 1.0, 2, 2, 1, args.steps)
            barrier()
            phase("timed-%d-rows" % shp[1])
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            rb = cyc.vcycle(ul, fl, ncycles=0, history=True)[0]
            lo, hi, a, b = cyc.rows()
            lay = {"rank0_rows_owned": b - a, "halo_rows": 4, "levels_cut": cyc.distributed_levels,
                   "whole_on_every_rank_from_points": args.crossover, "cycle_in_one_graph": cyc.uses_graph,
                   "transport": note[0], **({"rccl_version": _lib.rccl_version()} if use_rccl[0] else {})}
            cyc.close()
            return t.item(), ra, rb, lay

        dt, r0, r1, layout = run_slabs(shape)
        if not args.no_weak:
            # The other question, beside the line's own value: a grid that grows with the ranks, so that every
            # rank keeps the rows of a one-GPU problem and only exchanges and gathered levels come on top.
            wshape = (shape[0], (shape[1] - 1) * world + 1, shape[2])
            need = 6.0 * wshape[0] * (shape[1] + 7) * wshape[2] * w * -(-world // ndev)  # fields of the ranks of one device
            have = torch.cuda.get_device_properties(local).total_memory
        if not args.no_weak and need > 0.8 * have:
            weak = {"skipped": "the ranks sharing a device would need %.0f GB of its %.0f GB" % (need / 1e9, have / 1e9)}
        elif not args.no_weak:
            wdt, wr0, wr1, wlay = run_slabs(wshape)
            wdof = float(wshape[0]) * wshape[1] * wshape[2]
            weak = {"workload": "3D Poisson %dx%dx%d (z,y,x), %s, V(2,2): %d rows a rank" % (wshape + (dtname, shape[1] - 1)),
                    "value": wdof * args.steps / wdt, "unit": "DOF/s", "ms_per_step": wdt / args.steps * 1e3,
                    "scaling": "weak", "layout": wlay, "residual_before": wr0, "residual_after_warmup_and_steps": wr1}
        if args.config == "3d513" and not args.no_weak:
            # the configuration named for a whole node, 1025^3 in double precision, cut the same way (fixed grid)
            bshape = CONFIGS["3d1025f64"][0]
            need = 6.0 * bshape[0] * ((bshape[1] - 1) // world + 9) * bshape[2] * 8 * -(-world // ndev)
            if need > 0.8 * torch.cuda.get_device_properties(local).total_memory:
                big = {"skipped": "the ranks sharing a device would need %.0f GB" % (need / 1e9)}
            else:
                bdt, br0, br1, blay = run_slabs(bshape, torch.float64)
                bdof = float(bshape[0]) * bshape[1] * bshape[2]
                big = {"workload": CONFIGS["3d1025f64"][4] + ", f64, V(2,2)", "value": bdof * args.steps / bdt, "unit": "DOF/s",
                       "ms_per_step": bdt / args.steps * 1e3, "scaling": "strong", "layout": blay,
                       "residual_before": br0, "residual_after_warmup_and_steps": br1}
        if rank == 0:  # for the kernel timed alone below
            f = make_problem(shape, tdt, faces, dev)
            u = torch.zeros_like(f)
            torch.cuda.synchronize()

    phase("kernel")

    def time_kernel(u):
        # The dominant kernel of the workload: the pre-smoothing launch of the finest level, the longest
        # of the cycle, timed alone with events on the stream it runs on (on arrays of its own).
        g = _lib.grid(shape, faces=faces)
        dof = 1.0
        for s in shape:
            dof *= s
        work = torch.empty_like(u)
        if len(shape) == 2:
            cshape = tuple((s + 1) // 2 for s in shape)
            gc = _lib.grid(cshape, faces=faces)
            rc = torch.empty(cshape, dtype=tdt, device=dev)
            launch = lambda a, b: _lib.smooth_sweep_residual_restrict(g, gc, a, rc, b, f, 1.0, 2)
            alg_bytes = (3.0 * dof + float(rc.numel())) * w
            kernel = ("k_smooth2d_cols", "2D sweep of 2 red-black iterations that also writes the restricted residual")
            fields = {0: "float" if w == 4 else "double", 1: "2", 5: "2"}
        elif w == 4 and _lib.restrict_sweep_supported(g, tdt, 2):
            # the headline workload: the sweep also forms the residual and restricts it to the coarse grid
            cshape = tuple((s + 1) // 2 for s in shape)
            gc = _lib.grid(cshape, faces=faces)
            rc = torch.empty(cshape, dtype=tdt, device=dev)
            ec = torch.zeros(cshape, dtype=tdt, device=dev)
            resid = torch.empty_like(u)  # only the rows next to tile seams pass through it

            def part(mask, a, b):  # the operation is two launches: 1 = the sweep kernel, 2 = the seam rows' restriction
                _lib.sweep_parts(mask)
                try:
                    _lib.smooth_sweep_residual_restrict(g, gc, a, rc, b, f, 1.0, 2, r=resid)
                finally:
                    _lib.sweep_parts(3)
            launch = lambda a, b: part(1, a, b)
            alg_bytes = (3.0 * dof + float(rc.numel())) * w  # read u, read f, write the smoothed field, write the coarse rhs
            kernel = ("k_smooth3d_fused", "3D sweep of 2 red-black iterations with the residual pass and its restriction")
            fields = {0: "float", 1: "2", 5: "true", 6: "true", 7: "false", 10: "true"}
        else:
            launch = lambda a, b: _lib.smooth_sweep(g, a, b, f, 1.0, 2)
            alg_bytes = 3.0 * dof * w  # read u, read f, write the smoothed field
            kernel = ("k_smooth3d_fused", "3D sweep of 2 red-black iterations")
            fields = {0: "float" if w == 4 else "double", 1: "2", 5: "true", 6: "false", 7: "false"}
            if w == 4 and 60 * 60 <= shape[1] * shape[2] <= 300 * 300:
                # latency-bound level: two stages of one iteration each, two planes a time step
                kernel = ("k_smooth3d_fused_pair", kernel[1] + " (two stages, two planes a step)")
                fields = {0: "float", 3: "true"}
        # Every launch between its own pair of events on the stream it runs on, after untimed ones (the first
        # launches on freshly allocated arrays run 10-30 % long, the first tens of milliseconds after start-up 1-2 %).  For the headline workload the
        # launches that follow the sweep on the finest level of a cycle are issued in between (the seam rows'
        # restriction, the sweep on the corrected field), so that the sweep finds memory as it does in the
        # cycle; a third event after them closes the finest level's share of a cycle.
        follow = None
        if kernel[1].endswith("its restriction"):
            def follow(a, b):  # a: the smoothed field just written, b: the array the cycle continues in
                part(2, a, b)
                _lib.correct_smooth_sweep(g, gc, b, a, ec, f, 1.0, 2)
        reps, warm = (20, 20) if dof < 6e8 else (6, 2)  # some tens of milliseconds of untimed launches either way
        pairs = []
        for i in range(warm + reps):
            a, b = (work, u) if follow or i % 2 == 0 else (u, work)
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(stream)
            launch(a, b)
            e1.record(stream)
            if follow:
                follow(a, b)
                e2.record(stream)
            pairs.append((e0, e1, e2))
        pairs[-1][2 if follow else 1].synchronize()
        torch.cuda.synchronize()
        # a plane with more tiles than the device runs at once is swept in bands of tile rows, a launch each
        launches = 1
        if len(shape) == 3:
            if follow:
                part(1, work, u)
            else:
                launch(work, u)
            launches = max(1, _lib.last_sweep_launches())
            torch.cuda.synchronize()
        t_kernel = sum(x.elapsed_time(y) for x, y, _ in pairs[warm:]) * 1e-3 / reps
        level0 = None
        if follow:
            t_l0 = sum(x.elapsed_time(z) for x, _, z in pairs[warm:]) * 1e-3 / reps
            l0_bytes = 2.0 * (3.0 * dof + float(rc.numel())) * w
            level0 = {"ms": t_l0 * 1e3, "compulsory_bytes": l0_bytes, "achieved": l0_bytes / t_l0 / 1e9, "unit": "GB/s",
                      "frac": l0_bytes / t_l0 / 1e9 / HBM_PEAK_GBS,
                      "what": "the finest level's three launches of a cycle back to back between events: sweep x2 with "
                              "residual and restriction (u, f in; u, coarse rhs out), the seam rows' restriction, sweep x2 "
                              "on u + 4 P(e) (u, e, f in; u out); every field counted once per launch that needs it"}
        return {"t": t_kernel / launches, "level0": level0, "alg_bytes": alg_bytes / launches, "kernel": kernel, "fields": fields,
                "dof": dof, "launches": launches}

    with torch.cuda.stream(stream):
        if world == 1:
            # The kernel first: the cycles then start on a device that has been busy for some tens of milliseconds.
            # Straight after start-up the first ten to twenty cycles run 1-2 % long (W = 5: 1.60-1.61 ms a cycle in the
            # timed region at 513^3, W >= 30: 1.583, the per-cycle median either way), which says something about
            # the device's start-up and nothing about the cycle.
            timed = time_kernel(torch.zeros_like(f))
            plan = _lib.Plan(shape, tdt, faces=faces, zfull=True, use_graph=not args.no_graph)
            r0 = plan.vcycle(u, f, ncycles=0, history=True)[0]
            plan.vcycle(u, f, 1.0, 2, 2, 1, args.warmup)
            barrier()
            t0 = time.perf_counter()
            plan.vcycle(u, f, 1.0, 2, 2, 1, args.steps)
            barrier()
            dt = time.perf_counter() - t0
            r1 = plan.vcycle(u, f, ncycles=0, history=True)[0]
            # Outside the timed region: the same number of cycles again, every one between its own pair of
            # events on the stream it runs on, for the spread over the steps.
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
            marks[0].record(stream)
            for k in range(args.steps):
                plan.vcycle(u, f, 1.0, 2, 2, 1, 1)
                marks[k + 1].record(stream)
            marks[-1].synchronize()
            each = sorted(a.elapsed_time(b) for a, b in zip(marks, marks[1:]))
            if each:
                spread = {"min": each[0], "median": each[len(each) // 2], "max": each[-1], "steps": len(each),
                          "what": "every cycle between its own events, a second run of the same steps after the timed region"}
            plan.close()
        elif rank == 0:
            timed = time_kernel(u)

    if rank == 0:
        t_kernel, level0, alg_bytes, kernel, fields, dof = (timed[k] for k in ("t", "level0", "alg_bytes", "kernel", "fields", "dof"))
        achieved = alg_bytes / t_kernel / 1e9
        traffic = pmc_traffic(kernel[0], n, dtname, fields)
        # algorithmic bytes of a whole unfused V(2,2) cycle: 22 N w + 3 Nc w per level, 16 N w coarsest
        sizes, m = [], list(shape)
        while True:
            pts = 1
            for s in m:
                pts *= s
            sizes.append(pts)
            if min(m) <= 3:
                break
            m = [(s + 1) // 2 for s in m]
        cyc_bytes = sum((22 * a + 3 * b) * w for a, b in zip(sizes, sizes[1:])) + 16 * sizes[-1] * w
        dims = "%dD %d^%d" % (len(shape), n, len(shape))
        out = {
            "metric": "%s V-cycle DOF/s" % dims, "value": dof * args.steps / dt, "unit": "DOF/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": dtname, "data": "synthetic",
            **({"rehearsal": "%d ranks on %d GPU(s) over gloo: not a measurement" % (world, ndev)} if rehearse else {}),
            "config": {"workload": "%s, V(2,2), omega=1%s" % (workload, ", smoother over all z planes" if len(shape) == 3 else ""),
                       "name": args.config, "levels": len(sizes), "graph": not args.no_graph,
                       "parallelism": "single GPU" if world == 1 else "%d y-slabs, halo exchange: %s" % (world, note[0]),
                       "layout": layout,
                       "residual_before": r0, "residual_after_warmup_and_steps": r1},
            "vcycles_per_s": args.steps / dt,
            **({"grown_grid": weak} if weak else {}),
            **({"fixed_grid_1025_f64": big} if big else {}),
            "cycle_gbs_unfused_model": cyc_bytes * args.steps / dt / 1e9,
            "roofline": {"bound": "hbm", "kernel": "%s: %s, finest level" % kernel,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_s": t_kernel,
                         **({"launches_per_sweep": timed["launches"],
                             "note": "the plane has more tiles than the device runs at once: the sweep is issued as bands of tile "
                                     "rows, a launch each; bytes and time are means over the launches"}
                            if timed["launches"] > 1 else {})},
            **({"level0": level0} if level0 else {}),
            **({"step_ms": spread} if spread else {}),
        }
        if world == 1 and not args.no_cpu:
            del f, u, timed
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline(args.config, shape)
        if world > 1 and os.environ.get("MGRT_BENCH_ATTEMPT"):  # every region of this line comes from this attempt
            out["config"]["attempt"] = "rehearsal" if rehearse else ATTEMPTS[int(os.environ["MGRT_BENCH_ATTEMPT"])][0]
        if os.environ.get("MGRT_BENCH_EARLIER"):
            out["config"]["attempts_that_failed"] = os.environ["MGRT_BENCH_EARLIER"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        phase("done")
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
