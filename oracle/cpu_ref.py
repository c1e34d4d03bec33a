"""ctypes wrapper of oracle/cpu_ref.c (checker / cpu_baseline only; see the C file's header)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libcpu_ref.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        _lib = C.CDLL(path)
        _lib.cpu_ref_vmult.restype = C.c_int
        _lib.cpu_ref_bmop.restype = C.c_int
    return _lib


def cpu_share():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container on a
    128-thread host may own 16 of them: 128 OpenMP threads would then time-share 16 cores)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def set_threads(n):
    lib().cpu_ref_set_threads(int(n))


def greedy_cell_colors(loc2glob, n_dofs):
    """cell colouring with no shared dof inside a colour (stand-in for deal.II partition_color)."""
    nc = loc2glob.shape[0]
    dof_mask = np.zeros(n_dofs, dtype=np.uint64)
    color = np.zeros(nc, dtype=np.int64)
    for c in range(nc):
        forb = np.bitwise_or.reduce(dof_mask[loc2glob[c]])
        k = 0
        while (int(forb) >> k) & 1:
            k += 1
        color[c] = k
        dof_mask[loc2glob[c]] |= np.uint64(1 << k)
    return color


def structured_cell_colors(cells_per_dir):
    """2^dim parity colouring of a lexicographically ordered Cartesian mesh (x fastest)."""
    dims = list(cells_per_dir)
    idx = np.arange(int(np.prod(dims)))
    color = np.zeros(len(idx), dtype=np.int64)
    for d, nd_ in enumerate(dims):
        color |= ((idx % nd_) & 1) << d
        idx = idx // nd_
    return color


class CpuRef:
    def __init__(self, od, colors):
        """od: oracle Desc (double, uniform J0, no hanging nodes)"""
        assert od.uniform_j0 and od.constraint_mask is None
        self.od = od
        order = np.argsort(colors, kind="stable").astype(np.uint32)
        ncol = int(colors.max()) + 1
        off = np.zeros(ncol + 1, dtype=np.uint32)
        off[1:] = np.cumsum(np.bincount(colors, minlength=ncol))
        self.order, self.off, self.ncol = order, off, ncol
        f = lambda a, dt: np.ascontiguousarray(a, dtype=dt)
        self.arr = dict(l2g=f(od.loc2glob, np.uint32), coef=f(od.coefficient, np.float64), jxw=f(od.JxW, np.float64),
                        j0=f(od.inv_jac, np.float64), sv=f(od.shape_values, np.float64),
                        sg=f(od.shape_gradients, np.float64), con=f(od.constrained, np.uint32))

    def _args(self):
        a, od = self.arr, self.od
        p = lambda x: x.ctypes.data_as(C.c_void_p)
        return [C.c_int(od.dim), C.c_int(od.n), C.c_uint32(od.n_dofs), p(a["l2g"]), p(a["coef"]), p(a["jxw"]),
                p(a["j0"]), p(a["sv"]), p(a["sg"]), p(a["con"]), C.c_uint32(len(a["con"])), p(self.off),
                C.c_int(self.ncol), p(self.order)]

    def vmult(self, src):
        src = np.array(src, dtype=np.float64)
        dst = np.empty_like(src)
        self.threads = lib().cpu_ref_vmult(*self._args(), dst.ctypes.data_as(C.c_void_p), src.ctypes.data_as(C.c_void_p))
        return dst

    def bmop(self, n_iter, init=0.1):
        a, b = np.empty(self.od.n_dofs), np.zeros(self.od.n_dofs)
        w = lib().cpu_ref_bmop(*self._args(), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                               C.c_double(init), C.c_int(n_iter))
        return a if w == 0 else b
