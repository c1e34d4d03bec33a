#!/usr/bin/env python3
"""bench.py -- DoFs/s of consecutive Laplace vmult applies (the reference's bmop protocol).

One "step" = one operator apply  dst = A src  over the whole mesh, preceded by the pointer swap of
bmop.cu:142-146 (dst = 0.1 initially; K x { swap(dst,src); vmult(dst,src) }).  Workload at N = 1:
BASELINE.json configs[1]: DEGREE_FE=4, DIMENSION=3, uniform cube, n = 54 cells per direction
(157 464 cells, 217^3 = 10 218 313 DoFs), double.  For N > 1 the global cube has
n(N) = round(54 N^(1/3)) cells per direction (configs[3]: N = 8 -> 108^3 cells, 81 182 737 DoFs) and is
sharded by z-slabs, one process per GPU, interface-plane sums over RCCL (weak scaling).

Prints ONE JSON line on rank 0.  Inputs are resident in HBM when the timed region starts.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "dealii-cuda_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# torch / numpy / pymfgpu are imported by _imports(), AFTER the launcher branch of main(): the parent of a multi-GPU
# run must not load anything that could touch the GPU before it has started its children.
torch = dist = np = mf = SlabExchange = slab_ranges = None


def _imports():
    global torch, dist, np, mf, SlabExchange, slab_ranges
    import torch as _torch  # FIRST: libmfgpu.so must bind to the HIP runtime torch already loaded
    import torch.distributed as _dist
    import numpy as _np
    import pymfgpu as _mf
    from pymfgpu.parallel import SlabExchange as _SE, slab_ranges as _sr
    torch, dist, np, mf, SlabExchange, slab_ranges = _torch, _dist, _np, _mf, _SE, _sr


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment (the way the driver starts it): start N fresh
    rank processes (torch.distributed.run, one per GPU, rendezvous on 127.0.0.1) as CHILDREN of this process, which
    has made no GPU call and makes none; relay rank 0's JSON line; exit non-zero if the launcher or any rank does."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print(f"[bench] no WORLD_SIZE: launching {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    pr = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in pr.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if pr.returncode != 0 or line is None:
        print(f"[bench] multi-GPU run failed: launcher exit code {pr.returncode}"
              + ("" if line is not None else ", no result line from rank 0"), file=sys.stderr, flush=True)
        raise SystemExit(pr.returncode if pr.returncode != 0 else 1)
    print(line, flush=True)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(n_dofs, n_cells, nd, s):
    """src read once, dst written once, one coefficient value and one 32-bit dof
    index per local dof / quadrature point."""
    return 2 * s * n_dofs + n_cells * nd * (s + 4)


def csrc_sha16():
    """hash of the library sources the running libmfgpu.so was (supposed to be) built from"""
    import glob
    import hashlib
    hsh = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "dealii-cuda_amd", "csrc", "*")) + [os.path.join(ROOT, "include", "mfgpu.h")]):
        hsh.update(open(f, "rb").read())
    return hsh.hexdigest()[:16]


def parity_vector(n_dofs):
    """the vector of the GPU-vs-CPU figure: seeded, non-constant, every cell contributes"""
    return np.random.default_rng(20240229).standard_normal(n_dofs)


def cpu_baseline(args, budget_s=12.0):
    """oracle/cpu_ref.c (a port, not the reference: the reference CPU path needs deal.II) on the host cores: SAME mesh
    as the GPU run by default, same protocol (bmop-cpu.cc:137-155), bounded sample: as many vmults as fit in
    ~budget_s (>= 2).  Also returns the result of ONE apply to parity_vector(), for the GPU-vs-CPU check."""
    from oracle import cpu_ref
    from oracle import mf_oracle as o

    n = args.cpu_cells or args.cells
    mesh = mf.Mesh.uniform(3, args.degree, n)
    a = mesh.arrays()
    od = o.Desc(3, args.degree, mesh.n_dofs, a["loc2glob"], a["JxW"], a["inv_jac"],
                o.coefficient_value(a["quadrature_points"]), a["constrained_dofs"], None, np.float64,
                a["shape_values"], a["shape_gradients"])
    ref = cpu_ref.CpuRef(od, cpu_ref.structured_cell_colors([n] * 3))
    share = cpu_ref.cpu_share()
    cpu_ref.set_threads(share)  # one thread per CPU this process owns (affinity mask / cgroup quota)
    # warm-up (page faults, thread pool) on a seeded RANDOM vector, kept for the parity figure: the protocol's vector
    # of 0.1s lies in the operator's kernel away from the boundary (grad const = 0) and would test 11 % of the cells
    x0 = parity_vector(mesh.n_dofs)
    y1 = ref.vmult(x0)
    x = np.full(mesh.n_dofs, 0.1)
    t0 = time.perf_counter()
    k = 0
    while True:
        x = ref.vmult(x)
        x *= 0.1 / np.abs(x).max()  # keep the iterate finite (outside the reference's protocol, cheap)
        k += 1
        t = time.perf_counter() - t0
        if (k >= 2 and t > budget_s) or k >= 100:
            break
    return {"value": mesh.n_dofs * k / t, "unit": "DoFs/s", "cores": int(ref.threads), "kind": "port",
            "sample": f"{k} vmult of p={args.degree} 3D uniform n={n} ({mesh.n_dofs} DoFs), oracle/cpu_ref.c "
                      f"(SIMD over 8 cells, OpenMP {ref.threads} threads = the process's CPU share of "
                      f"{os.cpu_count()} host CPUs), {t:.1f} s"}, n, y1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--ramp-steps", type=int, default=2000,
                    help="untimed applies BEFORE the warm-up steps (0: none): an MI355X that idled through the set-up needs "
                         "~0.2 s of load to reach its sustained clocks; the first --steps applies are timed cold and "
                         "reported beside the headline (config.cold_start)")
    ap.add_argument("--degree", type=int, default=4)
    ap.add_argument("--cells", type=int, default=54, help="cells per direction at N=1")
    ap.add_argument("--cpu-cells", type=int, default=0, help="cells per direction of the CPU sample (0: as --cells)")
    ap.add_argument("--float", action="store_true", help="BMOP_USE_FLOATS")
    ap.add_argument("--mode", default="cxx", choices=["cxx", "p2p", "pair", "allreduce"],
                    help="N > 1: exchange of the slab interface planes. cxx (default): the library's C++ path behind "
                         "the C-ABI (mfgpu_vmult_dist: RCCL send/recv on a side stream, overlapped with pass 2); the "
                         "others: the Python test double pymfgpu/parallel.py over torch.distributed")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--kernel", default="auto", choices=["auto", "pencils", "pencils_x", "planes", "planes_2w"],
                    help="mfgpu_desc.kernel: cell-loop kernel family (measurements; the default is the library's choice)")
    ap.add_argument("--segments", type=int, default=0,
                    help="mfgpu_desc.cell_loop_segments (0 = the library's choice, 1 = no overlap of pass 2 with the cell loop)")
    ap.add_argument("--batch-cells", type=int, default=0)
    ap.add_argument("--batch-dofs", type=int, default=0)
    ap.add_argument("--colored", action="store_true", help="coloured-scatter mode instead of two-pass")
    ap.add_argument("--general-jacobian", type=float, default=0.0, metavar="EPS",
                    help="SURVEY.md 8(f) N3: same mesh, but a full inverse Jacobian per quadrature point "
                         "(synthetic deformation F = h (I + EPS R), R random in [-1,1]); 1 GPU")
    ap.add_argument("--adaptive", type=int, default=0, metavar="NREF",
                    help="configs[2]: bmop -DADAPTIVE_GRID mesh with hanging nodes instead of the uniform cube (1 GPU)")
    ap.add_argument("--renumber", action="store_true",
                    help="second line, NOT the headline: renumber the mesh's dofs batch-major first (mfgpu_suggest_renumbering, "
                         "the MatrixFree::renumber_dofs analogue); the reference and the default run keep the caller's numbering")
    ap.add_argument("--no-second-line", action="store_true", help="skip the renumbered second line of the default run")
    ap.add_argument("--ball", type=int, default=-1, metavar="NREF",
                    help="bmop -DBALL_GRID: hyper_ball with NREF global refinements (unstructured, general-geometry path; 1 GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus, sys.argv[1:])  # before any import that could touch the GPU
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("MFGPU_BENCH_ECHO_RANK"):  # tests/test_bench_launcher.py
        print(f"[bench] rank {rank} of {world} started", file=sys.stderr, flush=True)
    _imports()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # tests/test_gpu_dist_rccl_path.py runs the N > 1 path on ONE GPU: every rank on device 0, the torch process group
    # over gloo, the library's RCCL calls on a preloaded test double (never set outside that test)
    one_gpu_test = world > 1 and os.environ.get("MFGPU_BENCH_TEST_ONE_GPU") == "1"
    if one_gpu_test:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu_test:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    nt = mf.F32 if args.float else mf.F64
    tdt = torch.float32 if args.float else torch.float64
    s = 4 if args.float else 8
    p = args.degree
    # weak scaling: the global cube has ~world x the cells; with the default cell count (a multiple of 6) the edge stays a
    # multiple of 6, which the 3x2x2-cell batches of p = 4 tile without ragged batches (54, 66, 84, 108 cells for
    # 1, 2, 4, 8 GPUs: C4 at 8); bench.py --cells 64 runs 9 % below --cells 66 for that reason
    n_glob = int(round(args.cells * world ** (1.0 / 3.0)))
    if world > 1 and args.cells % 6 == 0:
        n_glob = 6 * int(round(args.cells * world ** (1.0 / 3.0) / 6.0))
    zb, ze = slab_ranges(n_glob, world)[rank]
    if args.ball >= 0:
        if world != 1:
            raise SystemExit("--ball is a single-GPU configuration")
        mesh = mf.Mesh.ball(3, p, args.ball, number_type=nt)
    elif args.adaptive:
        if world != 1:
            raise SystemExit("--adaptive is a single-GPU configuration")
        mesh = mf.Mesh.adaptive(3, p, args.adaptive, number_type=nt)
    else:
        mesh = mf.Mesh.uniform(3, p, n_glob, slab=(zb, ze), number_type=nt)
    if args.renumber:
        if world != 1:
            raise SystemExit("--renumber is a single-GPU measurement")
        mesh.desc.max_cells_per_batch = args.batch_cells
        mesh.desc.max_dofs_per_batch = args.batch_dofs
        mesh.renumber(mesh.suggest_renumbering())
    mesh.desc.max_cells_per_batch = args.batch_cells
    mesh.desc.max_dofs_per_batch = args.batch_dofs
    mesh.desc.cell_loop_segments = args.segments
    mesh.desc.kernel = {"auto": mf.KERNEL_AUTO, "pencils": mf.KERNEL_PENCILS, "pencils_x": mf.KERNEL_PENCILS_X,
                        "planes": mf.KERNEL_PLANES, "planes_2w": mf.KERNEL_PLANES_2W}[args.kernel]
    if args.colored:
        mesh.desc.flags |= mf.COLORED_SCATTER
    if args.general_jacobian:
        if world != 1 or args.adaptive or args.float:
            raise SystemExit("--general-jacobian: single GPU, uniform connectivity, double")
        a = mesh.arrays()
        nc, ndl = mesh.n_cells, (p + 1) ** 3
        rng = np.random.default_rng(0)
        h = 1.0 / a["inv_jac"].reshape(nc, 1, 1, 1)
        F = (np.eye(3) + args.general_jacobian * rng.uniform(-1.0, 1.0, (nc, ndl, 3, 3))) * h
        jxw = a["JxW"].reshape(nc, ndl) / h.reshape(nc, 1) ** 3 * np.linalg.det(F)
        gdesc, keep = mf.make_desc(3, p, mesh.n_dofs, a["loc2glob"], jxw, np.linalg.inv(F), None, a["constrained_dofs"],
                                   a["shape_values"], a["shape_gradients"], quadrature_points=a["quadrature_points"],
                                   max_cells_per_batch=args.batch_cells, max_dofs_per_batch=args.batch_dofs)
        del F
        op = mf.Operator(gdesc, (keep, mesh))
    else:
        op = mf.Operator(mesh.desc, mesh)
    stats = op.plan_stats()
    N_loc = mesh.n_dofs
    nd = (p + 1) ** 3
    n_dofs_glob = (p * n_glob + 1) ** 3
    n_cells_glob = n_glob ** 3
    if args.adaptive or args.ball >= 0:
        n_dofs_glob, n_cells_glob = mesh.n_dofs, mesh.n_cells

    dst = torch.full((N_loc,), 0.1, device=dev, dtype=tdt)  # bmop.cu:140
    src = torch.zeros(N_loc, device=dev, dtype=tdt)
    stream = torch.cuda.current_stream().cuda_stream
    exch, cxx = None, None
    if world > 1 and args.mode == "cxx":
        # RCCL unique id from rank 0 to everybody (over the torch process group), then the library's own communicator
        idt = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(mf.dist_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, 0)
        try:
            cxx = mf.Dist(mesh, rank, world, unique_id=bytes(idt.cpu().numpy().tobytes()))
            cxx.attach(op)
            failed = 0.0
        except mf.MfgpuError as e:
            print(f"[bench] rank {rank}: mfgpu_dist_create failed ({e}); falling back to --mode p2p", file=sys.stderr)
            cxx, failed = None, 1.0
        flag = torch.tensor([failed], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if flag.item() != 0.0:  # all ranks or none
            cxx = None
            args.mode = "p2p"
            exch = SlabExchange(mesh, rank, world, dev, tdt, "p2p")
    if cxx is not None:
        # the first multi-GPU run is also the first run of the RCCL transport (no multi-GPU box during development):
        # check one distributed apply against the dense all-reduce of the interface planes before anything is timed
        chk = SlabExchange(mesh, rank, world, dev, tdt, "allreduce")
        xs = torch.linspace(0.5, 1.5, N_loc, device=dev, dtype=tdt)
        y1, y2 = torch.empty_like(xs), torch.empty_like(xs)
        err = float("inf")
        try:
            cxx.vmult(op, y1, xs, stream)
            torch.cuda.synchronize()
            op.vmult(y2, xs, stream)
            chk.exchange_add(y2)
            torch.cuda.synchronize()
            err = float((y1 - y2).norm() / y2.norm())
        except (mf.MfgpuError, RuntimeError) as e:
            print(f"[bench] rank {rank}: mfgpu_vmult_dist failed ({e})", file=sys.stderr)
        bad = torch.tensor([0.0 if err <= (1e-5 if args.float else 1e-12) else 1.0], device=dev)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        del xs, y1, y2
        if bad.item() != 0.0:  # all ranks or none: time the torch.distributed exchange instead, and say so in the line
            print(f"[bench] rank {rank}: mfgpu_vmult_dist failed or disagrees with the all-reduce exchange of the "
                  f"interface planes (rel. error {err:.3e}); falling back to --mode p2p", file=sys.stderr)
            cxx = None
            args.mode = "p2p"
            exch = SlabExchange(mesh, rank, world, dev, tdt, "p2p")
        del chk
    elif world > 1 and exch is None:
        exch = SlabExchange(mesh, rank, world, dev, tdt, args.mode)

    def step():
        nonlocal dst, src
        dst, src = src, dst  # GpuVector::swap
        if cxx is not None:
            cxx.vmult(op, dst, src, stream)
            return
        op.vmult(dst, src, stream)
        if exch is not None:
            exch.exchange_add(dst)

    def renorm():
        # the un-normalised protocol overflows double after ~100 applies (values grow by ||A|| per
        # apply); rescale OUTSIDE the timed region so warm-up + profile legs never hit inf
        nonlocal dst
        m = dst.abs().max()
        if world > 1:
            dist.all_reduce(m, op=dist.ReduceOp.MAX)
        dst.mul_(0.1 / m)

    # Clock ramp.  The GPU idles while the host builds mesh and plan; its power management then takes ~0.1-0.3 s of load
    # to reach the sustained clocks (C2: 0.153-0.159 ms per vmult for 100 applies right after set-up, 0.137-0.140 after
    # 2000 untimed applies).  The same K applies timed cold, right now, are reported as
    # config.cold_start; then --ramp-steps untimed applies (renormalised every 50: the un-normalised protocol
    # overflows after ~100), then the contract's W warm-up and K timed steps.  Same counts on every rank.
    cold_ms = None
    if args.ramp_steps > 0:
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        tc = time.perf_counter()
        n_cold = min(args.steps, 100)
        for _ in range(n_cold):
            step()
        torch.cuda.synchronize()
        cold_ms = (time.perf_counter() - tc) / n_cold * 1e3
        renorm()
        for i in range(args.ramp_steps):
            step()
            if i % 50 == 49:
                renorm()
        renorm()
    for _ in range(args.warmup):
        step()
    renorm()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    tt = torch.tensor([t], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    t = float(tt.item())
    finite = bool(torch.isfinite(dst).all().item())

    # ---- roofline leg: per-launch HIP-event timing of the cell-loop kernel on the launch stream
    # In the state the timed region ran in: a first pass creates the handle's events (host work between the launches,
    # and the host synchronisations above, let the clocks sag: 20 launches measured right here read 112-121 us where the
    # steady state is 96-98), then untimed applies back to the steady state, then the measured launches with nothing
    # but the event records between them.
    n_prof = min(args.steps, 50)

    def apply_local(count):
        nonlocal dst, src
        for i in range(count):
            dst, src = src, dst
            op.vmult(dst, src, stream)
            if i % 50 == 49:
                dst.mul_(0.1 / dst.abs().max())

    renorm()
    op.profile_enable(True)
    apply_local(n_prof)
    op.profile_read()
    op.profile_read_pass2()
    op.profile_enable(False)
    renorm()
    apply_local(min(args.ramp_steps, 500))
    renorm()
    op.profile_enable(True)
    apply_local(n_prof)
    k_ms, n_v = op.profile_read()
    p2_ms = op.profile_read_pass2()
    op.profile_enable(False)
    launches = n_v * stats["n_launches"]
    b_alg_loc = algorithmic_bytes(N_loc, mesh.n_cells, nd, s)
    general = bool(args.general_jacobian) or args.ball >= 0
    if general:  # per quadrature point: the 6 entries of the symmetric a JxW J J^T instead of one scalar
        b_alg_loc += mesh.n_cells * nd * 5 * s
    achieved = b_alg_loc * n_v / (k_ms * 1e-3) / 1e9  # GB/s, == (B_alg/launch) / (avg launch duration)
    # HBM traffic of the dominant kernel: NOT measured by this run (PMC counters need rocprofv3 passes of their own,
    # tools/profile_bench.sh); copied from the committed summary of the last such passes IF it was taken on this
    # workload, kernel and library source (hash of csrc/), else null.  `traffic_source` says where it came from.
    traffic, traffic_step, traffic_source = None, None, None
    tj = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tj):
        try:
            tr = json.load(open(tj))
            if (tr.get("workload") == f"p{p}_3d_n{n_glob}_{'f32' if args.float else 'f64'}" and world == 1
                    and op.kernel_name() + "<" in tr.get("kernel", "") and tr.get("csrc_sha16") == csrc_sha16()
                    and not args.adaptive and not args.colored and not args.batch_cells and not args.batch_dofs
                    and not general and args.kernel == "auto" and not args.renumber):
                traffic = tr["hbm_bytes_per_launch"]
                traffic_step = tr.get("step_hbm_bytes_per_vmult")
                traffic_source = {"file": "profiles/traffic_latest.json", "profile": tr.get("profile"),
                                  "csrc_sha16": tr.get("csrc_sha16")}
        except Exception:
            traffic = None

    out = {
        "metric": f"DoFs/s on {args.steps}x Laplace vmult (bmop), p={p} 3D uniform",
        "value": n_dofs_glob * args.steps / t,
        "unit": "DoFs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * t / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if args.float else "f64",
        "data": "synthetic",
        "config": {"workload": (f"bmop: DEGREE_FE={p}, DIMENSION=3, ADAPTIVE_GRID n_ref={args.adaptive}, hanging nodes, "
                                f"{n_cells_glob} cells, {n_dofs_glob} DoFs") if args.adaptive else
                               (f"bmop -DBALL_GRID: DEGREE_FE={p}, DIMENSION=3, hyper_ball, {args.ball} global refinements, "
                                f"{n_cells_glob} cells, {n_dofs_glob} DoFs, MappingQ1 (full inverse Jacobian per quadrature "
                                f"point); roofline bytes count the folded 6-entry metric (48 B / point)") if args.ball >= 0 else
                               (f"bmop without MATRIX_FREE_UNIFORM_MESH: DEGREE_FE={p}, DIMENSION=3, {n_glob}^3 cells, "
                                f"{n_dofs_glob} DoFs, full inverse Jacobian per quadrature point (synthetic, eps="
                                f"{args.general_jacobian}); roofline bytes count the folded 6-entry metric (48 B / point)")
                               if args.general_jacobian else
                               (f"bmop: DEGREE_FE={p}, DIMENSION=3, MATRIX_FREE_UNIFORM_MESH, hyper_cube(-1,1), "
                                f"{n_glob}^3 cells, {n_dofs_glob} DoFs, {world} z-slab(s)"),
                   "cells_per_dir": n_glob, "n_dofs": n_dofs_glob, "n_cells": n_cells_glob,
                   "parallelism": f"slab{world}" + (f"/{args.mode}" if world > 1 else ""),
                   "dof_numbering": "batch-major (mfgpu_suggest_renumbering)" if args.renumber else "caller's (lexicographic)",
                   "plan": stats, "finite": finite,
                   # the same K applies timed right after set-up, before the clock ramp (rank 0's clock; not the headline)
                   "clock_ramp_steps": args.ramp_steps,
                   "cold_start": None if cold_ms is None else
                   {"ms_per_step": cold_ms, "value": n_dofs_glob / (cold_ms * 1e-3), "steps": min(args.steps, 100)}},
        # `achieved` / `frac`: the dominant kernel (the cell loop), algorithmic bytes per launch over its average launch
        # duration, as the bench contract defines it.  Pass 2 writes the shared and constrained dofs, so the cell loop alone
        # does not move all of B_alg: `vmult_frac` divides by the time of ALL kernels of a vmult (cell loop + pass 2, HIP
        # events on the launch stream), `step_frac` by ms_per_step.  `traffic`: PMC bytes of the cell-loop kernel per launch,
        # `traffic_step`: of all kernels of a vmult.
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "vmult_achieved": b_alg_loc * n_v / ((k_ms + p2_ms) * 1e-3) / 1e9,
                     "vmult_frac": b_alg_loc * n_v / ((k_ms + p2_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "step_frac": b_alg_loc / (1e-3 * (1e3 * t / args.steps)) / 1e9 / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_step": traffic_step, "traffic_source": traffic_source,
                     "kernel": op.kernel_name(), "launches": launches,
                     "avg_launch_us": 1e3 * k_ms / max(launches, 1),
                     "alg_bytes_per_launch": b_alg_loc / stats["n_launches"],
                     "kernel_ms_per_vmult": k_ms / max(n_v, 1), "pass2_ms_per_vmult": p2_ms / max(n_v, 1)},
    }
    if rank == 0 and world == 1 and not args.no_cpu and not args.adaptive and not general and not args.renumber:
        cb, n_cpu, y_cpu = cpu_baseline(args)
        out["cpu_baseline"] = cb
        if n_cpu == n_glob and not args.float:
            # GPU path vs the CPU path on the same mesh, one apply to the same seeded random vector (north_star:
            # "matching the reference CPU path ... to a stated floating-point tolerance": 1e-12 relative l2 in double)
            src.copy_(torch.from_numpy(parity_vector(N_loc)))
            op.vmult(dst, src, stream)
            torch.cuda.synchronize()
            y_gpu = dst.cpu().numpy()
            out["gpu_vs_cpu_rel_l2"] = float(np.linalg.norm(y_gpu - y_cpu) / np.linalg.norm(y_cpu))
    if (rank == 0 and world == 1 and not args.no_second_line and not args.renumber and not args.adaptive and not general
            and not args.colored):
        # SECOND line, not the headline: the same workload after the optional batch-major dof renumbering
        # (mfgpu_suggest_renumbering, the MatrixFree::renumber_dofs analogue the reference does not use).  `value`
        # above is measured in the caller's numbering.
        try:
            del op
            mesh2 = mf.Mesh.uniform(3, p, n_glob, number_type=nt)
            mesh2.desc.max_cells_per_batch, mesh2.desc.max_dofs_per_batch = args.batch_cells, args.batch_dofs
            mesh2.renumber(mesh2.suggest_renumbering())
            mesh2.desc.max_cells_per_batch, mesh2.desc.max_dofs_per_batch = args.batch_cells, args.batch_dofs
            mesh2.desc.kernel = mesh.desc.kernel
            op2 = mf.Operator(mesh2.desc, mesh2)
            dst.fill_(0.1)
            src.zero_()

            def apply2(count):
                nonlocal dst, src
                for i in range(count):
                    dst, src = src, dst
                    op2.vmult(dst, src, stream)
                    if i % 50 == 49:
                        dst.mul_(0.1 / dst.abs().max())

            op2.profile_enable(True)  # (creates the handle's events; see the roofline leg above)
            apply2(min(args.steps, 50))
            op2.profile_read()
            op2.profile_enable(False)
            dst.mul_(0.1 / dst.abs().max())
            apply2(args.ramp_steps)  # same protocol as the first line: clock ramp, W warm-up steps, K timed steps
            apply2(args.warmup)
            dst.mul_(0.1 / dst.abs().max())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                dst, src = src, dst
                op2.vmult(dst, src, stream)
            torch.cuda.synchronize()
            t2 = time.perf_counter() - t0
            dst.mul_(0.1 / dst.abs().max())
            apply2(min(args.ramp_steps, 500))
            dst.mul_(0.1 / dst.abs().max())
            op2.profile_enable(True)
            apply2(min(args.steps, 50))
            k2, n2 = op2.profile_read()
            out["second_line_renumbered"] = {
                "note": "NOT the headline: same workload, dofs renumbered batch-major first (opt-in mfgpu_suggest_renumbering)",
                "value": n_dofs_glob * args.steps / t2, "unit": "DoFs/s", "ms_per_step": 1e3 * t2 / args.steps,
                "kernel_ms_per_vmult": k2 / max(n2, 1), "kernel": op2.kernel_name(),
                "roofline_frac": b_alg_loc * n2 / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBS}
        except Exception as e:  # the second line must never cost the first
            out["second_line_renumbered"] = {"error": str(e)}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
