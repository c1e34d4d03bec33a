"""ctypes binding of the C-ABI in include/mfgpu.h (test / bench harness plumbing).

The product is the shared library dealii-cuda_amd/lib/libmfgpu.so (hand-written HIP kernels
behind a C-ABI) and the C++ shim in dealii-cuda_amd/host/.  This module only loads that library;
it has NO fallback: if the library is missing or a call fails it raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MFGPU_LIB selects another build of the SAME sources (tools/stamps.py: lib/libmfgpu_diag.so)
LIB_PATH = os.environ.get("MFGPU_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libmfgpu.so")

F64, F32 = 0, 1
UNIFORM_J0, HANGING_NODES, COLORED_SCATTER = 1, 2, 1 << 8
KERNEL_AUTO, KERNEL_PENCILS, KERNEL_PENCILS_X, KERNEL_PLANES, KERNEL_PLANES_2W = 0, 1, 2, 3, 4  # Desc.kernel


class MfgpuError(RuntimeError):
    pass


class Desc(C.Structure):
    """mirror of struct mfgpu_desc"""
    _fields_ = [
        ("dim", C.c_int32), ("degree", C.c_int32), ("number_type", C.c_int32), ("flags", C.c_uint32),
        ("n_dofs", C.c_uint32), ("n_cells", C.c_uint32),
        ("loc2glob", C.c_void_p), ("constraint_mask", C.c_void_p),
        ("JxW", C.c_void_p), ("inv_jac", C.c_void_p), ("coefficient", C.c_void_p),
        ("quadrature_points", C.c_void_p), ("shape_values", C.c_void_p), ("shape_gradients", C.c_void_p),
        ("constraint_weights", C.c_void_p), ("constrained_dofs", C.c_void_p),
        ("n_constrained", C.c_uint32), ("max_cells_per_batch", C.c_uint32), ("max_dofs_per_batch", C.c_uint32),
        ("kernel", C.c_uint32), ("cell_loop_segments", C.c_uint32), ("max_workgroups", C.c_uint32),
    ]


# every symbol include/mfgpu.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "mfgpu_create", "mfgpu_vmult", "mfgpu_vmult_add", "mfgpu_n_dofs", "mfgpu_memory_consumption",
    "mfgpu_destroy", "mfgpu_last_error", "mfgpu_plan_stats", "mfgpu_kernel_name", "mfgpu_compute_inverse_diagonal", "mfgpu_set_constrained_values",
    "mfgpu_vec_sadd", "mfgpu_vec_equ", "mfgpu_vec_scale", "mfgpu_vec_divide", "mfgpu_vec_invert", "mfgpu_vec_mul",
    "mfgpu_vec_dot", "mfgpu_vec_l2_norm", "mfgpu_vec_add_and_dot", "mfgpu_vec_all_zero", "mfgpu_profile_enable", "mfgpu_profile_read", "mfgpu_profile_read_pass2",
    "mfgpu_plan_create", "mfgpu_plan_destroy", "mfgpu_plan_array_u32", "mfgpu_plan_lmap", "mfgpu_plan_bflags",
    "mfgpu_vec_alloc", "mfgpu_vec_free", "mfgpu_vec_fill", "mfgpu_vec_from_host", "mfgpu_vec_to_host",
    "mfgpu_device_synchronize", "mfgpu_device_memory_info", "mfgpu_mesh_create_uniform", "mfgpu_mesh_create_adaptive", "mfgpu_mesh_create_ball", "mfgpu_mesh_create_from_leaves",
    "mfgpu_mesh_cell_levels", "mfgpu_mesh_destroy",
    "mfgpu_mesh_desc", "mfgpu_mesh_dof_coords", "mfgpu_mesh_interface_dofs",
    "mfgpu_dist_unique_id", "mfgpu_dist_create", "mfgpu_dist_connect_local", "mfgpu_dist_attach", "mfgpu_dist_schedule",
    "mfgpu_vmult_dist_begin", "mfgpu_vmult_dist_end", "mfgpu_vmult_dist", "mfgpu_dist_destroy",
    "mfgpu_transfer_create", "mfgpu_transfer_create_from_meshes", "mfgpu_transfer_prolongate",
    "mfgpu_transfer_restrict_and_add", "mfgpu_transfer_memory_consumption", "mfgpu_transfer_destroy",
    "mfgpu_mesh_transfer_patches", "mfgpu_suggest_renumbering", "mfgpu_mesh_renumber",
    "mfgpu_level_create", "mfgpu_level_operator", "mfgpu_level_vmult_interface_down", "mfgpu_level_vmult_interface_up",
    "mfgpu_level_destroy", "mfgpu_index_pairs_create", "mfgpu_vec_copy_pairs", "mfgpu_index_pairs_destroy",
    "mfgpu_mesh_create_adaptive_mg", "mfgpu_mg_hierarchy_create", "mfgpu_mg_n_levels", "mfgpu_mg_level_mesh",
    "mfgpu_mg_edge_dofs", "mfgpu_mg_copy_pairs", "mfgpu_mg_transfer_arrays", "mfgpu_mg_hierarchy_destroy",
]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MfgpuError(f"{LIB_PATH} not found: build it with __graft_entry__.build() "
                             "(make -C dealii-cuda_amd); there is no fallback path")
        L = C.CDLL(LIB_PATH)
        L.mfgpu_last_error.restype = C.c_char_p
        L.mfgpu_create.argtypes = [C.POINTER(Desc), C.POINTER(C.c_void_p)]
        L.mfgpu_vmult.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mfgpu_vmult_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mfgpu_n_dofs.argtypes = [C.c_void_p]
        L.mfgpu_n_dofs.restype = C.c_uint32
        L.mfgpu_memory_consumption.argtypes = [C.c_void_p]
        L.mfgpu_memory_consumption.restype = C.c_size_t
        L.mfgpu_destroy.argtypes = [C.c_void_p]
        L.mfgpu_destroy.restype = None
        L.mfgpu_plan_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.mfgpu_device_memory_info.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.mfgpu_kernel_name.argtypes = [C.c_void_p]
        L.mfgpu_kernel_name.restype = C.c_char_p
        L.mfgpu_compute_inverse_diagonal.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.mfgpu_set_constrained_values.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
        vp, d, z, i = C.c_void_p, C.c_double, C.c_size_t, C.c_int
        L.mfgpu_vec_sadd.argtypes = [vp, d, d, vp, z, i, vp]
        L.mfgpu_vec_equ.argtypes = [vp, d, vp, z, i, vp]
        L.mfgpu_vec_scale.argtypes = [vp, vp, z, i, vp]
        L.mfgpu_vec_divide.argtypes = [vp, vp, z, i, vp]
        L.mfgpu_vec_invert.argtypes = [vp, z, i, vp]
        L.mfgpu_vec_mul.argtypes = [vp, d, z, i, vp]
        L.mfgpu_vec_dot.argtypes = [vp, vp, z, i, vp, C.POINTER(d)]
        L.mfgpu_vec_l2_norm.argtypes = [vp, z, i, vp, C.POINTER(d)]
        L.mfgpu_vec_add_and_dot.argtypes = [vp, d, vp, vp, z, i, vp, C.POINTER(d)]
        L.mfgpu_vec_all_zero.argtypes = [vp, z, i, vp, C.POINTER(i)]
        L.mfgpu_profile_enable.argtypes = [C.c_void_p, C.c_int]
        L.mfgpu_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.mfgpu_profile_read_pass2.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.mfgpu_plan_create.argtypes = [C.POINTER(Desc), C.POINTER(C.c_void_p)]
        L.mfgpu_plan_destroy.argtypes = [C.c_void_p]
        L.mfgpu_plan_destroy.restype = None
        L.mfgpu_plan_array_u32.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.mfgpu_plan_array_u32.restype = C.c_int64
        L.mfgpu_plan_lmap.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.mfgpu_plan_lmap.restype = C.c_int64
        L.mfgpu_plan_bflags.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.mfgpu_plan_bflags.restype = C.c_int64
        L.mfgpu_vec_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_int]
        L.mfgpu_vec_free.argtypes = [C.c_void_p]
        L.mfgpu_vec_fill.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_void_p]
        L.mfgpu_vec_from_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.mfgpu_vec_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.mfgpu_mesh_create_uniform.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_uint32), C.c_double, C.c_double,
                                                C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]
        L.mfgpu_mesh_create_adaptive.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.mfgpu_mesh_create_ball.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.mfgpu_mesh_create_from_leaves.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]
        L.mfgpu_mesh_cell_levels.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.mfgpu_mesh_cell_levels.restype = C.c_int64
        L.mfgpu_mesh_destroy.argtypes = [C.c_void_p]
        L.mfgpu_mesh_destroy.restype = None
        L.mfgpu_mesh_desc.argtypes = [C.c_void_p, C.POINTER(Desc)]
        L.mfgpu_mesh_dof_coords.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.mfgpu_mesh_dof_coords.restype = C.c_int64
        L.mfgpu_mesh_interface_dofs.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.mfgpu_mesh_interface_dofs.restype = C.c_int64
        L.mfgpu_transfer_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                                            C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p)]
        L.mfgpu_transfer_create_from_meshes.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        L.mfgpu_transfer_prolongate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mfgpu_transfer_restrict_and_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mfgpu_transfer_memory_consumption.argtypes = [C.c_void_p]
        L.mfgpu_transfer_memory_consumption.restype = C.c_size_t
        L.mfgpu_mesh_transfer_patches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mfgpu_mesh_transfer_patches.restype = C.c_int64
        L.mfgpu_suggest_renumbering.argtypes = [C.POINTER(Desc), C.c_void_p]
        L.mfgpu_mesh_renumber.argtypes = [C.c_void_p, C.c_void_p]
        L.mfgpu_level_create.argtypes = [C.POINTER(Desc), C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
        L.mfgpu_level_operator.argtypes = [C.c_void_p]
        L.mfgpu_level_operator.restype = C.c_void_p
        L.mfgpu_level_vmult_interface_down.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mfgpu_level_vmult_interface_up.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mfgpu_mesh_create_adaptive_mg.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.mfgpu_mg_hierarchy_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.mfgpu_mg_n_levels.argtypes = [C.c_void_p]
        L.mfgpu_mg_level_mesh.argtypes = [C.c_void_p, C.c_int]
        L.mfgpu_mg_level_mesh.restype = C.c_void_p
        L.mfgpu_mg_edge_dofs.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.mfgpu_mg_edge_dofs.restype = C.c_int64
        L.mfgpu_mg_copy_pairs.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.mfgpu_mg_copy_pairs.restype = C.c_int64
        L.mfgpu_mg_transfer_arrays.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.mfgpu_mg_transfer_arrays.restype = C.c_int64
        L.mfgpu_mg_hierarchy_destroy.argtypes = [C.c_void_p]
        L.mfgpu_mg_hierarchy_destroy.restype = None
        L.mfgpu_index_pairs_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
        L.mfgpu_vec_copy_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.mfgpu_index_pairs_destroy.argtypes = [C.c_void_p]
        L.mfgpu_index_pairs_destroy.restype = None
        L.mfgpu_level_destroy.argtypes = [C.c_void_p]
        L.mfgpu_level_destroy.restype = None
        L.mfgpu_transfer_destroy.argtypes = [C.c_void_p]
        L.mfgpu_transfer_destroy.restype = None
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise MfgpuError(f"mfgpu error {rc}: {lib().mfgpu_last_error().decode()}")


def np_dtype(number_type):
    return np.float64 if number_type == F64 else np.float32


def _view(ptr, count, dtype):
    if count <= 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count)


class Mesh:
    """Host-side stand-in for Triangulation + DoFHandler + ConstraintMatrix (mfgpu_mesh_*)."""

    def __init__(self, handle, keep=None):
        self._h = handle
        self._keep = keep
        self.desc = Desc()
        _check(lib().mfgpu_mesh_desc(self._h, C.byref(self.desc)))

    @classmethod
    def uniform(cls, dim, degree, n_per_dir, lo=-1.0, hi=1.0, slab=(0, 0), number_type=F64):
        if np.isscalar(n_per_dir):
            n_per_dir = [int(n_per_dir)] * dim
        arr = (C.c_uint32 * 3)(*(list(n_per_dir) + [1] * (3 - len(n_per_dir))))
        h = C.c_void_p()
        _check(lib().mfgpu_mesh_create_uniform(dim, degree, arr, lo, hi, slab[0], slab[1], number_type, C.byref(h)))
        return cls(h)

    @classmethod
    def adaptive(cls, dim, degree, n_ref, number_type=F64):
        h = C.c_void_p()
        _check(lib().mfgpu_mesh_create_adaptive(dim, degree, n_ref, number_type, C.byref(h)))
        return cls(h)

    @classmethod
    def adaptive_mg(cls, dim, degree, n_ref, number_type=F64):
        """the ADAPTIVE_GRID recipe with 2:1 balance over vertices too (what the multigrid hierarchy needs)"""
        h = C.c_void_p()
        _check(lib().mfgpu_mesh_create_adaptive_mg(dim, degree, n_ref, number_type, C.byref(h)))
        return cls(h)

    @classmethod
    def ball(cls, dim, degree, n_ref, number_type=F64):
        h = C.c_void_p()
        _check(lib().mfgpu_mesh_create_ball(dim, degree, n_ref, number_type, C.byref(h)))
        return cls(h)

    @classmethod
    def from_leaves(cls, dim, degree, leaves, number_type=F64):
        lv = np.ascontiguousarray(leaves, dtype=np.uint32).reshape(-1, 4)
        h = C.c_void_p()
        _check(lib().mfgpu_mesh_create_from_leaves(dim, degree, lv.ctypes.data, len(lv), number_type, C.byref(h)))
        return cls(h)

    def cell_levels(self):
        p = C.c_void_p()
        cnt = lib().mfgpu_mesh_cell_levels(self._h, C.byref(p))
        return _view(p.value, cnt, np.uint32).reshape(-1, 4).copy()

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None and getattr(self, "_owner", None) is None:
            _lib.mfgpu_mesh_destroy(self._h)  # (_lib is None at interpreter shutdown)
            self._h = None

    # ---- numpy views of the description arrays (valid while the mesh lives)
    @property
    def n_dofs(self):
        return int(self.desc.n_dofs)

    @property
    def n_cells(self):
        return int(self.desc.n_cells)

    @property
    def nd(self):
        return (self.desc.degree + 1) ** self.desc.dim

    def arrays(self):
        d = self.desc
        dt = np_dtype(d.number_type)
        nd, nc, n = self.nd, d.n_cells, d.degree + 1
        out = dict(
            loc2glob=_view(d.loc2glob, nc * nd, np.uint32).reshape(nc, nd),
            JxW=_view(d.JxW, nc * nd, dt).reshape(nc, nd),
            inv_jac=(_view(d.inv_jac, nc, dt) if d.flags & UNIFORM_J0
                     else _view(d.inv_jac, nc * nd * d.dim * d.dim, dt).reshape(nc, nd, d.dim, d.dim)),
            quadrature_points=_view(d.quadrature_points, nc * nd * d.dim, dt).reshape(nc, nd, d.dim),
            shape_values=_view(d.shape_values, n * n, dt),
            shape_gradients=_view(d.shape_gradients, n * n, dt),
            constraint_weights=_view(d.constraint_weights, n * n, np.float64),
            constrained_dofs=_view(d.constrained_dofs, d.n_constrained, np.uint32),
            constraint_mask=_view(d.constraint_mask, nc, np.uint32) if d.constraint_mask else None,
        )
        return out

    def dof_coords(self):
        p = C.c_void_p()
        cnt = lib().mfgpu_mesh_dof_coords(self._h, C.byref(p))
        return _view(p.value, cnt, np.float64).reshape(-1, self.desc.dim)

    def suggest_renumbering(self):
        """new_index[old] = new, batch-major for the operator's plan (mfgpu_suggest_renumbering); host only"""
        out = np.zeros(self.n_dofs, dtype=np.uint32)
        _check(lib().mfgpu_suggest_renumbering(C.byref(self.desc), out.ctypes.data))
        return out

    def renumber(self, new_index):
        """DoFHandler::renumber_dofs on the stand-in mesh (in place).  self.desc is refilled from the mesh: set the
        tuning knobs (kernel, max_*_per_batch, ...) again afterwards."""
        ni = np.ascontiguousarray(new_index, dtype=np.uint32)
        assert ni.size == self.n_dofs
        _check(lib().mfgpu_mesh_renumber(self._h, ni.ctypes.data))
        _check(lib().mfgpu_mesh_desc(self._h, C.byref(self.desc)))

    def transfer_patches(self, fine: "Mesh"):
        """(coarse_cell_dofs, fine_patch_dofs) of the level pair (self, fine): host arrays, no GPU"""
        p, dim = self.desc.degree, self.desc.dim
        cd = np.zeros((self.n_cells, (p + 1) ** dim), dtype=np.uint32)
        fd = np.zeros((self.n_cells, (2 * p + 1) ** dim), dtype=np.uint32)
        rc = lib().mfgpu_mesh_transfer_patches(self._h, fine._h, cd.ctypes.data, fd.ctypes.data)
        if rc < 0:
            _check(int(rc))
        return cd, fd

    def interface_dofs(self, which):
        p = C.c_void_p()
        cnt = lib().mfgpu_mesh_interface_dofs(self._h, which, C.byref(p))
        return _view(p.value, cnt, np.uint32).copy()


def make_desc(dim, degree, n_dofs, loc2glob, JxW, inv_jac, coefficient, constrained,
              shape_values, shape_gradients, number_type=F64, constraint_mask=None,
              constraint_weights=None, quadrature_points=None, max_cells_per_batch=0, max_dofs_per_batch=0,
              colored=False, kernel=0, cell_loop_segments=0, max_workgroups=0):
    """Build a Desc from numpy arrays; returns (desc, keepalive list)."""
    dt = np_dtype(number_type)
    keep = []

    def ptr(a, dtype):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=dtype)
        keep.append(a)
        return a.ctypes.data

    d = Desc()
    d.dim, d.degree, d.number_type = dim, degree, number_type
    l2g = np.ascontiguousarray(loc2glob, dtype=np.uint32)
    d.n_dofs = int(n_dofs)
    d.n_cells = l2g.size // ((degree + 1) ** dim)
    # one scalar J^-1 per cell (reference MATRIX_FREE_UNIFORM_MESH), or J^-1[dim][dim] per quadrature point
    uniform = np.asarray(inv_jac).size == d.n_cells
    assert uniform or np.asarray(inv_jac).size == l2g.size * dim * dim
    d.flags = ((UNIFORM_J0 if uniform else 0) | (HANGING_NODES if constraint_mask is not None else 0)
               | (COLORED_SCATTER if colored else 0))
    d.loc2glob = ptr(l2g, np.uint32)
    d.constraint_mask = ptr(constraint_mask, np.uint32)
    d.JxW = ptr(JxW, dt)
    d.inv_jac = ptr(inv_jac, dt)
    d.coefficient = ptr(coefficient, dt)
    d.quadrature_points = ptr(quadrature_points, dt)
    d.shape_values = ptr(shape_values, dt)
    d.shape_gradients = ptr(shape_gradients, dt)
    d.constraint_weights = ptr(constraint_weights, np.float64)
    c = np.ascontiguousarray(constrained, dtype=np.uint32)
    d.constrained_dofs = ptr(c, np.uint32)
    d.n_constrained = c.size
    d.max_cells_per_batch = max_cells_per_batch
    d.max_dofs_per_batch = max_dofs_per_batch
    d.kernel = kernel
    d.cell_loop_segments = cell_loop_segments
    d.max_workgroups = max_workgroups
    return d, keep


class Plan:
    """Host-only plan (no GPU): mfgpu_plan_*"""

    def __init__(self, desc: Desc, keep=None):
        self._keep = keep
        self.nd = (desc.degree + 1) ** desc.dim
        h = C.c_void_p()
        _check(lib().mfgpu_plan_create(C.byref(desc), C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.mfgpu_plan_destroy(self._h)
            self._h = None

    def _u32(self, what):
        p = C.c_void_p()
        cnt = lib().mfgpu_plan_array_u32(self._h, what, C.byref(p))
        if cnt < 0:
            _check(int(cnt))
        return _view(p.value, cnt, np.uint32).copy()

    batch_cell_off = property(lambda s: s._u32(0))
    batch_dof_off = property(lambda s: s._u32(1))
    color_batch_off = property(lambda s: s._u32(2))
    cell_order = property(lambda s: s._u32(3))
    bdofs = property(lambda s: s._u32(4) & np.uint32(0x7fffffff))       # global ids
    bdofs_constrained = property(lambda s: (s._u32(4) >> np.uint32(31)).astype(bool))
    orphans = property(lambda s: s._u32(5))
    batch_nint = property(lambda s: s._u32(6))
    halo_off = property(lambda s: s._u32(7))
    sdofs = property(lambda s: s._u32(8))
    s_off = property(lambda s: s._u32(9))
    s_idx = property(lambda s: s._u32(10))
    chunks = property(lambda s: s._u32(11).reshape(-1, 4))   # {sdofs position, count | k << 16, gstarts offset, offset in group}
    gstarts = property(lambda s: s._u32(12))
    # plane plans (apply_planes3): fixed-size per-batch records
    pr_dofs = property(lambda s: s._u32(13))   # dof lists, bit 31 = constrained
    pr_idx = property(lambda s: s._u32(14))    # index runs
    pr_hn = property(lambda s: s._u32(15))     # hanging-node records of the batches of masked cells
    pr_hn_slot = property(lambda s: s._u32(16))  # per plane batch: index of its record in pr_hn, or 0xffffffff

    @property
    def lmap(self):
        p = C.c_void_p()
        cnt = lib().mfgpu_plan_lmap(self._h, C.byref(p))
        return _view(p.value, cnt, np.uint16).reshape(-1, self.nd).copy()

    @property
    def bflags(self):
        p = C.c_void_p()
        cnt = lib().mfgpu_plan_bflags(self._h, C.byref(p))
        return _view(p.value, cnt, np.uint8).copy()


class DeviceVector:
    """GpuVector<Number> pieces on the path (gpu_vec.h): owning device buffer."""

    def __init__(self, n, number_type=F64):
        self.n, self.number_type = int(n), number_type
        p = C.c_void_p()
        _check(lib().mfgpu_vec_alloc(C.byref(p), self.n, number_type))
        self.ptr = p.value

    def __del__(self):
        if getattr(self, "ptr", None) and _lib is not None:
            _lib.mfgpu_vec_free(self.ptr)
            self.ptr = None

    def fill(self, value, stream=None):
        _check(lib().mfgpu_vec_fill(self.ptr, self.n, self.number_type, float(value), stream))

    def from_host(self, a):
        a = np.ascontiguousarray(a, dtype=np_dtype(self.number_type))
        assert a.size == self.n
        _check(lib().mfgpu_vec_from_host(self.ptr, a.ctypes.data, self.n, self.number_type))

    def to_host(self):
        a = np.empty(self.n, dtype=np_dtype(self.number_type))
        _check(lib().mfgpu_vec_to_host(a.ctypes.data, self.ptr, self.n, self.number_type))
        return a

    def swap(self, other):
        self.ptr, other.ptr = other.ptr, self.ptr
        self.n, other.n = other.n, self.n

    # ---- GpuVector BLAS-1 and reductions (gpu_vec.h:105-157), SURVEY.md 8(f) N2
    def _a(self, other=None):
        if other is not None:
            assert other.n == self.n and other.number_type == self.number_type
        return self.n, self.number_type

    def sadd(self, s, a, w, stream=None):      # this = s*this + a*w
        _check(lib().mfgpu_vec_sadd(self.ptr, float(s), float(a), w.ptr, *self._a(w), stream))

    def add(self, a, w, stream=None):          # this += a*w
        self.sadd(1.0, a, w, stream)

    def equ(self, a, w, stream=None):          # this = a*w
        _check(lib().mfgpu_vec_equ(self.ptr, float(a), w.ptr, *self._a(w), stream))

    def scale(self, w, stream=None):           # this[i] *= w[i]
        _check(lib().mfgpu_vec_scale(self.ptr, w.ptr, *self._a(w), stream))

    def divide(self, w, stream=None):          # operator/=
        _check(lib().mfgpu_vec_divide(self.ptr, w.ptr, *self._a(w), stream))

    def invert(self, stream=None):
        _check(lib().mfgpu_vec_invert(self.ptr, *self._a(), stream))

    def mul(self, a, stream=None):             # operator*=
        _check(lib().mfgpu_vec_mul(self.ptr, float(a), *self._a(), stream))

    def dot(self, w, stream=None):             # operator*
        r = C.c_double()
        _check(lib().mfgpu_vec_dot(self.ptr, w.ptr, *self._a(w), stream, C.byref(r)))
        return r.value

    def l2_norm(self, stream=None):
        r = C.c_double()
        _check(lib().mfgpu_vec_l2_norm(self.ptr, *self._a(), stream, C.byref(r)))
        return r.value

    def add_and_dot(self, a, x, w, stream=None):  # this += a*x; return this . w
        r = C.c_double()
        self._a(x)
        _check(lib().mfgpu_vec_add_and_dot(self.ptr, float(a), x.ptr, w.ptr, *self._a(w), stream, C.byref(r)))
        return r.value

    def all_zero(self, stream=None):
        r = C.c_int()
        _check(lib().mfgpu_vec_all_zero(self.ptr, *self._a(), stream, C.byref(r)))
        return bool(r.value)


class Operator:
    """LaplaceOperatorGpu surface over the C-ABI handle."""

    def __init__(self, desc: Desc, keep=None):
        self._keep = keep
        self.number_type = desc.number_type
        h = C.c_void_p()
        _check(lib().mfgpu_create(C.byref(desc), C.byref(h)))
        self._h = h

    def __del__(self):
        self.clear()

    def clear(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.mfgpu_destroy(self._h)
            self._h = None

    def n(self):
        return int(lib().mfgpu_n_dofs(self._h))

    m = n

    def vmult(self, dst, src, stream=None):
        _check(lib().mfgpu_vmult(self._h, _ptr(dst), _ptr(src), stream))

    def vmult_add(self, dst, src, stream=None):
        _check(lib().mfgpu_vmult_add(self._h, _ptr(dst), _ptr(src), stream))

    def memory_consumption(self):
        return int(lib().mfgpu_memory_consumption(self._h))

    def plan_stats(self):
        s = (C.c_uint64 * 8)()
        _check(lib().mfgpu_plan_stats(self._h, s))
        keys = ["n_batches", "n_launches", "batch_dofs", "max_batch_dofs", "max_batch_cells", "n_orphans",
                "first_touch_or_shared_dofs", "rmw_adds_or_halo_slots"]
        return dict(zip(keys, [int(v) for v in s]))

    def kernel_name(self):
        return lib().mfgpu_kernel_name(self._h).decode()

    def compute_inverse_diagonal(self, inv_diag, stream=None):
        """LaplaceOperatorGpu::compute_diagonal + get_diagonal_inverse (laplace_operator_gpu.h:401-429)"""
        _check(lib().mfgpu_compute_inverse_diagonal(self._h, _ptr(inv_diag), stream))

    def set_constrained_values(self, vec, value, stream=None):
        """ConstraintHandlerGpu::set_constrained_values (constraint_handler_gpu.cu:126-137)"""
        _check(lib().mfgpu_set_constrained_values(self._h, _ptr(vec), float(value), stream))

    def profile_enable(self, on=True):
        _check(lib().mfgpu_profile_enable(self._h, int(on)))

    def profile_read(self):
        ms, nv = C.c_double(), C.c_uint64()
        _check(lib().mfgpu_profile_read(self._h, C.byref(ms), C.byref(nv)))
        return ms.value, int(nv.value)

    def profile_read_pass2(self):
        ms = C.c_double()
        _check(lib().mfgpu_profile_read_pass2(self._h, C.byref(ms)))
        return ms.value


class MgHierarchy:
    """Level hierarchy of an adaptive stand-in mesh (mfgpu_mg_*): level meshes, refinement-edge dofs, transfer arrays,
    copy_to_mg pairs.  Host only."""

    def __init__(self, mesh: "Mesh"):
        h = C.c_void_p()
        _check(lib().mfgpu_mg_hierarchy_create(mesh._h, C.byref(h)))
        self._h = h
        self.n_levels = int(lib().mfgpu_mg_n_levels(h))

    def level_mesh(self, level):
        m = Mesh.__new__(Mesh)
        m._owner = self  # borrowed: the level meshes belong to the hierarchy (Mesh.__del__ leaves them alone)
        m._h = C.c_void_p(lib().mfgpu_mg_level_mesh(self._h, level))
        m.desc = Desc()
        _check(lib().mfgpu_mesh_desc(m._h, C.byref(m.desc)))
        return m

    def edge_dofs(self, level):
        p = C.c_void_p()
        n = lib().mfgpu_mg_edge_dofs(self._h, level, C.byref(p))
        return _view(p.value, n, np.uint32).copy()

    def copy_pairs(self, level):
        a, b = C.c_void_p(), C.c_void_p()
        n = lib().mfgpu_mg_copy_pairs(self._h, level, C.byref(a), C.byref(b))
        return _view(a.value, n, np.uint32).copy(), _view(b.value, n, np.uint32).copy()

    def transfer_arrays(self, level, nd, nfd):
        a, b = C.c_void_p(), C.c_void_p()
        n = lib().mfgpu_mg_transfer_arrays(self._h, level, C.byref(a), C.byref(b))
        return _view(a.value, n * nd, np.uint32).reshape(n, nd).copy(), _view(b.value, n * nfd, np.uint32).reshape(n, nfd).copy()

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.mfgpu_mg_hierarchy_destroy(self._h)
            self._h = None


class Level:
    """Level operator with refinement edges + the interface matrices (mfgpu_level_*)."""

    def __init__(self, desc: Desc, edge_dofs, keep=None):
        self._keep = keep
        e = np.ascontiguousarray(edge_dofs, dtype=np.uint32)
        h = C.c_void_p()
        _check(lib().mfgpu_level_create(C.byref(desc), e.ctypes.data if e.size else None, e.size, C.byref(h)))
        self._h = h
        self.number_type = desc.number_type

    def vmult(self, dst, src, stream=None):
        _check(lib().mfgpu_vmult(lib().mfgpu_level_operator(self._h), _ptr(dst), _ptr(src), stream))

    def compute_inverse_diagonal(self, inv_diag, stream=None):
        _check(lib().mfgpu_compute_inverse_diagonal(lib().mfgpu_level_operator(self._h), _ptr(inv_diag), stream))

    def vmult_interface_down(self, dst, src, stream=None):
        _check(lib().mfgpu_level_vmult_interface_down(self._h, _ptr(dst), _ptr(src), stream))

    def vmult_interface_up(self, dst, src, stream=None):
        _check(lib().mfgpu_level_vmult_interface_up(self._h, _ptr(dst), _ptr(src), stream))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.mfgpu_level_destroy(self._h)
            self._h = None


class Transfer:
    """MGTransferMatrixFreeGpu between two globally refined levels (mfgpu_transfer_*)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_arrays(cls, dim, degree, coarse_cell_dofs, fine_patch_dofs, n_coarse_dofs, n_fine_dofs, coarse_dirichlet,
                    number_type=F64, prolongation_1d=None):
        cd = np.ascontiguousarray(coarse_cell_dofs, dtype=np.uint32)
        fd = np.ascontiguousarray(fine_patch_dofs, dtype=np.uint32)
        di = np.ascontiguousarray(coarse_dirichlet, dtype=np.uint32)
        p1 = None if prolongation_1d is None else np.ascontiguousarray(prolongation_1d, dtype=np.float64)
        n_cells = cd.size // ((degree + 1) ** dim)
        assert fd.size == n_cells * (2 * degree + 1) ** dim
        h = C.c_void_p()
        _check(lib().mfgpu_transfer_create(dim, degree, number_type, n_cells, cd.ctypes.data, fd.ctypes.data, n_coarse_dofs,
                                           n_fine_dofs, di.ctypes.data if di.size else None, di.size,
                                           None if p1 is None else p1.ctypes.data, C.byref(h)))
        return cls(h)

    @classmethod
    def from_meshes(cls, coarse: "Mesh", fine: "Mesh"):
        h = C.c_void_p()
        _check(lib().mfgpu_transfer_create_from_meshes(coarse._h, fine._h, C.byref(h)))
        return cls(h)

    def prolongate(self, dst_fine, src_coarse, stream=None):
        _check(lib().mfgpu_transfer_prolongate(self._h, _ptr(dst_fine), _ptr(src_coarse), stream))

    def restrict_and_add(self, dst_coarse, src_fine, stream=None):
        _check(lib().mfgpu_transfer_restrict_and_add(self._h, _ptr(dst_coarse), _ptr(src_fine), stream))

    def memory_consumption(self):
        return int(lib().mfgpu_transfer_memory_consumption(self._h))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.mfgpu_transfer_destroy(self._h)
            self._h = None


def _ptr(v):
    if isinstance(v, DeviceVector):
        return v.ptr
    if hasattr(v, "data_ptr"):  # torch tensor on the GPU
        return v.data_ptr()
    return int(v)


def synchronize():
    _check(lib().mfgpu_device_synchronize())


def device_memory_info():
    """(free, total) bytes of the current device"""
    f, t = C.c_size_t(), C.c_size_t()
    _check(lib().mfgpu_device_memory_info(C.byref(f), C.byref(t)))
    return f.value, t.value


def dist_unique_id() -> bytes:
    """rank 0: the 128-byte RCCL unique id every rank passes to Dist (mfgpu_dist_unique_id)"""
    buf = C.create_string_buffer(128)
    _check(lib().mfgpu_dist_unique_id(buf))
    return buf.raw


class Dist:
    """mfgpu_dist: exchange of a z-slab's interface planes with its two neighbours (include/mfgpu.h).
    unique_id = the bytes of dist_unique_id() (RCCL transport, collective over all ranks), or None (in-process
    transport: connect neighbours with connect_local)."""

    def __init__(self, mesh: "Mesh", rank: int, world: int, unique_id=None):
        lo = np.ascontiguousarray(mesh.interface_dofs(0), dtype=np.uint32)
        up = np.ascontiguousarray(mesh.interface_dofs(1), dtype=np.uint32)
        con = np.ascontiguousarray(mesh.arrays()["constrained_dofs"], dtype=np.uint32)
        self._keep = (lo, up, con)
        self.number_type = mesh.desc.number_type
        d = C.c_void_p()
        L = lib()
        L.mfgpu_dist_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                        C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        _check(L.mfgpu_dist_create(unique_id, rank, world, lo.ctypes.data if lo.size else None, lo.size,
                                   up.ctypes.data if up.size else None, up.size, con.ctypes.data if con.size else None,
                                   con.size, mesh.n_dofs, self.number_type, C.byref(d)))
        self._d = d
        for f in ("mfgpu_dist_connect_local", "mfgpu_dist_attach"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_void_p]
        L.mfgpu_vmult_dist_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mfgpu_vmult_dist.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mfgpu_vmult_dist_end.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mfgpu_dist_destroy.argtypes = [C.c_void_p]

    def __del__(self):
        if getattr(self, "_d", None) and _lib is not None:
            _lib.mfgpu_dist_destroy(self._d)
            self._d = None

    def connect_local(self, upper: "Dist"):
        """in-process transport: `upper` is the slab above this one"""
        _check(lib().mfgpu_dist_connect_local(self._d, upper._d))

    def attach(self, op: "Operator"):
        _check(lib().mfgpu_dist_attach(self._d, op._h))

    def schedule(self):
        """(interface_first, r1_end, r2_begin, n_batches): mfgpu_dist_schedule"""
        info = (C.c_uint32 * 4)()
        lib().mfgpu_dist_schedule.argtypes = [C.c_void_p, C.c_void_p]
        _check(lib().mfgpu_dist_schedule(self._d, info))
        return bool(info[0]), int(info[1]), int(info[2]), int(info[3])

    def vmult_begin(self, op, dst, src, stream=None):
        _check(lib().mfgpu_vmult_dist_begin(op._h, self._d, _ptr(dst), _ptr(src), stream))

    def vmult_end(self, op, dst, stream=None):
        _check(lib().mfgpu_vmult_dist_end(op._h, self._d, _ptr(dst), stream))

    def vmult(self, op, dst, src, stream=None):
        _check(lib().mfgpu_vmult_dist(op._h, self._d, _ptr(dst), _ptr(src), stream))
