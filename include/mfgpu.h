/* mfgpu.h -- C-ABI of the MI355X-native matrix-free Laplace operator apply.
 *
 * Drop-in boundary for ONE hot path of kalj/dealii-cuda: LaplaceOperatorGpu::vmult
 * (reference laplace_operator_gpu.h:216-223) as driven by bmop.cu:134-153.
 *
 * The reference has no FFI layer: its boundary is a C++ template API whose inputs come
 * from deal.II (DoFHandler, ConstraintMatrix, FEValues, ShapeInfo).  deal.II objects cannot
 * cross a C-ABI, so this ABI sits exactly at the OUTPUT of MatrixFreeGpu::reinit
 * (matrix_free_gpu.cu:448-563) and ConstraintHandlerGpu::reinit
 * (constraint_handler_gpu.cu:68-95): plain host arrays, plain device pointers, sizes.
 * All file:line citations are relative to the reference tree.
 *
 * Conventions: every function returns 0 on success or a negative MFGPU_E* code (no C++
 * exception crosses the boundary; the reference throws dealii::ExcMessage,
 * cuda_utils.cuh:15-25); mfgpu_last_error() gives the message of the calling thread's
 * last failure.  Handles are independent (no process-global shape tables, unlike
 * matrix_free_gpu.h:45-48), thread-compatible, not thread-safe per handle.  All calls on ONE handle must be ordered on
 * a single stream (or serialised by the caller with events): every vmult uses the handle's halo buffer, so two
 * vmults of one handle in flight on different streams race on it.
 * `stream` arguments are hipStream_t passed as void* (NULL = default stream).
 */
#ifndef MFGPU_H
#define MFGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFGPU_OK 0
#define MFGPU_EINVAL (-1)   /* bad argument / inconsistent description        */
#define MFGPU_EHIP (-2)     /* HIP runtime error (message has the HIP string) */
#define MFGPU_ENOMEM (-3)
#define MFGPU_EUNSUPPORTED (-4) /* (dim, degree, number type) not instantiated */

/* number_type */
#define MFGPU_F64 0
#define MFGPU_F32 1 /* reference: -DBMOP_USE_FLOATS, bmop.cu:60-64 */

/* flags */
#define MFGPU_UNIFORM_J0 (1u << 0)    /* inv_jac holds ONE scalar J^-1[0][0] per cell
                                         (MATRIX_FREE_UNIFORM_MESH, matrix_free_gpu.cu:332-334,
                                         fee_gpu.cuh:225-241).  Without it: the reference's default
                                         geometry path, a full J^-1 per quadrature point (fee_gpu.cuh:
                                         235-241,275-281) -- implemented in two-pass
                                         scatter mode (2D and 3D, with or without hanging nodes); with
                                         MFGPU_COLORED_SCATTER: MFGPU_EUNSUPPORTED                  */
#define MFGPU_HANGING_NODES (1u << 1) /* constraint_mask is given (MATRIX_FREE_HANGING_NODES,
                                         fee_gpu.cuh:333-335,349-351)                           */

#define MFGPU_COLORED_SCATTER (1u << 8) /* scatter mode: one launch per batch colour with first-touch
                                         stores and later-colour adds (the reference's use_coloring idea,
                                         matrix_free_gpu.h:374-379, fee_gpu.cuh:359-362, at batch level).
                                         Default is the two-pass mode: one sweep over all batches, batch-
                                         surface partial sums reduced per dof by a second kernel.           */

typedef struct mfgpu_handle mfgpu_handle; /* replaces MatrixFreeGpu + coefficient + ConstraintHandlerGpu
                                             inside LaplaceOperatorGpu (laplace_operator_gpu.h:85-95) */

/* Output of MatrixFreeGpu::reinit / ConstraintHandlerGpu::reinit as plain HOST arrays.
 * Per-cell arrays are UNPADDED (row length n^dim, not the power-of-two `rowlength` of
 * matrix_free_gpu.cu:483) and hold all cells of all colours in one array; colouring is
 * done by the library (coloring.cc:8-33 is replaced).                       */
typedef struct mfgpu_desc {
  int32_t dim;         /* 2 or 3                                    (bmop.cu:53-57)   */
  int32_t degree;      /* FE_Q degree p, n = p+1 points per direction (bmop.cu:47-51) */
  int32_t number_type; /* MFGPU_F64 / MFGPU_F32: type of every `const void*` array and of the vectors */
  uint32_t flags;
  uint32_t n_dofs;  /* vector length                              (matrix_free_gpu.cu:494)     */
  uint32_t n_cells; /* active cells                               (matrix_free_gpu.cu:495)     */
  const uint32_t *loc2glob;        /* [n_cells * n^dim] lexicographic (x fastest), hanging-node
                                      entries already substituted   (matrix_free_gpu.cu:287-300) */
  const uint32_t *constraint_mask; /* [n_cells] or NULL              (hanging_nodes.cuh:23-53)   */
  const void *JxW;                 /* [n_cells * n^dim]              (matrix_free_gpu.cu:315-322) */
  const void *inv_jac;             /* UNIFORM_J0: [n_cells]; else [n_cells * n^dim * dim*dim],
                                      per quadrature point row-major J^-1[d1][d2]
                                      (matrix_free_gpu.cu:324-338)                                */
  const void *coefficient;         /* [n_cells * n^dim] values at quadrature points, or NULL to
                                      evaluate 1/(0.05+2|x|^2) from quadrature_points on the device
                                      (laplace_operator_gpu.h:191-211, poisson_common.h:149-151)   */
  const void *quadrature_points;   /* [n_cells * n^dim * dim] (x,y,z per point) or NULL
                                      (matrix_free_gpu.cu:305-313)                                */
  const void *shape_values;        /* [n*n], index dof*n + q         (matrix_free_gpu.cu:502-509) */
  const void *shape_gradients;     /* [n*n], index dof*n + q         (matrix_free_gpu.cu:510-513) */
  const double *constraint_weights; /* [n*n] row-major W[i][j] or NULL (hanging_nodes.cuh:580-598) */
  const uint32_t *constrained_dofs; /* ascending list of every constrained DoF, Dirichlet AND
                                       hanging                        (constraint_handler_gpu.cu:68-95) */
  uint32_t n_constrained;
  /* tuning knobs, 0 = library default (no reference counterpart; replaces
     cells_per_block_shmem, matrix_free_gpu.h:298-315) */
  uint32_t max_cells_per_batch;
  uint32_t max_dofs_per_batch;
  uint32_t kernel; /* MFGPU_KERNEL_*: which cell-loop kernel family to use; 0 = the library's choice */
  uint32_t cell_loop_segments; /* two-pass mode: the cell loop runs as this many launches over consecutive batch
                                  ranges; the shared-dof sums (pass 2) a finished range completes run on a side
                                  stream of the handle, next to the following range's cells.  0 = the library's
                                  choice, 1 = one launch, pass 2 strictly after it                              */
  uint32_t max_workgroups;     /* cap on the resident workgroups of the persistent cell-loop launches (0 = as many
                                  as fit on the device): leaves room for other work on the GPU; also lets small
                                  meshes exercise the multi-batch loop of a workgroup                          */
} mfgpu_desc;

/* mfgpu_desc.kernel (all variants compute the same operator; non-default ones exist for tests and measurements) */
#define MFGPU_KERNEL_AUTO 0
#define MFGPU_KERNEL_PENCILS 1   /* apply_batches: a thread owns a 1D pencil, 2 workgroups per CU (2D; coloured mode)   */
#define MFGPU_KERNEL_PENCILS_X 2 /* apply_batches_x: the same for 3 workgroups per CU (3D two-pass; hanging nodes)      */
#define MFGPU_KERNEL_PLANES 3    /* apply_planes3: a thread owns a 2D plane, one wave per batch, one wave per SIMD (3D, p = 2..4) */
#define MFGPU_KERNEL_PLANES_2W 4 /* apply_planes4: the same plan and records with half the resources per wave -- one LDS
                                    transpose array aliased with the batch array, <= 256 registers -- for two waves per
                                    SIMD; measures equal to apply_planes3 on MI355X (profiles/r03_notes.md), not the default */

/* ---- operator (replaces LaplaceOperatorGpu::reinit / vmult / vmult_add / clear) -------- */

/* laplace_operator_gpu.h:120-151 (reinit): copies the description to the device, builds the
 * batch plan, folds coefficient * J0^2 * JxW (general geometry: the symmetric coefficient * JxW * J^-1 J^-T per
 * quadrature point).  Uses the current HIP device.  Fails (MFGPU_EHIP) if no HIP device / kernel image is
 * available: there is no CPU fallback.                                                        */
int mfgpu_create(const mfgpu_desc *desc, mfgpu_handle **out);

/* laplace_operator_gpu.h:216-223: dst = A src.  dst, src: device vectors of n_dofs Numbers.
 * Unlike the reference (const_cast at :293,302) src is never written.  Asynchronous.         */
int mfgpu_vmult(mfgpu_handle *h, void *dst_dev, const void *src_dev, void *stream);

/* laplace_operator_gpu.h:286-303: dst += A src on free rows, dst_c += src_c on constrained rows. */
int mfgpu_vmult_add(mfgpu_handle *h, void *dst_dev, const void *src_dev, void *stream);

/* laplace_operator_gpu.h:53-54 m() / n() */
uint32_t mfgpu_n_dofs(const mfgpu_handle *h);

/* laplace_operator_gpu.h:434-445 */
size_t mfgpu_memory_consumption(const mfgpu_handle *h);

/* laplace_operator_gpu.h:110-117 (clear) / matrix_free_gpu.cu:566-596 (free) */
void mfgpu_destroy(mfgpu_handle *h);

const char *mfgpu_last_error(void);

/* Plan statistics (for DESIGN/bench reporting and the CPU-side host-logic tests).
 * stats[0]=n_batches [1]=cell-loop launches per vmult (colours; 1 in two-pass mode) [2]=total batch dofs
 * [3]=max dofs/batch [4]=max cells/batch [5]=n_orphan_dofs
 * [6]=first-touch stores (coloured) / shared dofs (two-pass) [7]=RMW adds (coloured) / halo slots (two-pass) */
int mfgpu_plan_stats(const mfgpu_handle *h, uint64_t stats[8]);

/* Name of the cell-loop kernel this operator launches ("apply_batches_x", "apply_batches", ...): the
 * kernel the roofline figures of bench.py and the rocprofv3 summaries under profiles/ refer to.      */
const char *mfgpu_kernel_name(const mfgpu_handle *h);

/* Average device time of the cell-loop kernels of the most recent mfgpu_vmult* calls, measured
 * with hipEvents on the launch stream when profiling is enabled (bench.py roofline leg).       */
int mfgpu_profile_enable(mfgpu_handle *h, int on);
int mfgpu_profile_read(mfgpu_handle *h, double *kernel_ms_total, uint64_t *n_vmults);
/* ... and of the pass-2 kernels (reduce_classes) of the same mfgpu_vmult / mfgpu_vmult_add calls: 0 in coloured mode. */
int mfgpu_profile_read_pass2(mfgpu_handle *h, double *pass2_ms_total);

/* ---- host-only plan (no GPU needed): same planner the handle uses -------------------------
 * Lets CPU tests check the batching / colouring / first-touch logic.                         */
typedef struct mfgpu_plan mfgpu_plan;
int mfgpu_plan_create(const mfgpu_desc *desc, mfgpu_plan **out);
void mfgpu_plan_destroy(mfgpu_plan *p);
/* what: 0 batch_cell_off[n_batches+1] 1 batch_dof_off[n_batches+1] 2 color_batch_off[n_colors+1]
 *       3 cell_order[n_cells] (plan position -> caller cell) 4 bdofs[total] 5 orphans[n_orphans]
 *       6 batch_nint[n_batches] 7 halo_off[n_batches+1] 8 sdofs[n_shared] 9 s_off[n_shared+1] 10 s_idx
 *       11 chunks[4*n_chunks] {sdofs position, count | k<<16, gstarts offset, offset in group} 12 gstarts
 *       plane plans (apply_planes3): 13 dof-list records [n_plane_batches * slots * 64] 14 index-run records
 *       15 hanging-node records of the batches of masked cells (layout: mfgpu_internal.h p_hn_rows)
 *       16 per plane batch the index of its record in 15, or 0xffffffff (a batch of cells without a mask)
 * returns element count, *ptr = host pointer valid until mfgpu_plan_destroy                  */
int64_t mfgpu_plan_array_u32(const mfgpu_plan *p, int what, const uint32_t **ptr);
int64_t mfgpu_plan_lmap(const mfgpu_plan *p, const uint16_t **ptr);   /* [n_cells*n^dim], plan order */
int64_t mfgpu_plan_bflags(const mfgpu_plan *p, const uint8_t **ptr);  /* bit0 constrained, bit1 add */

/* ---- optional: a dof numbering that suits the operator (deal.II's MatrixFree::renumber_dofs; the reference does not
 * renumber, and nothing here requires it).  new_index[old dof] = new dof, batch-major: the dofs of a batch are one
 * contiguous run (coalesced gathers and stores of the cell loop), the dofs shared between batches follow in the order
 * pass 2 walks them.  The caller renumbers ITS DoFHandler (DoFHandler::renumber_dofs; stand-in: mfgpu_mesh_renumber)
 * and describes the renumbered mesh to mfgpu_create; vectors then live in the new numbering.  Host only.          */
int mfgpu_suggest_renumbering(const mfgpu_desc *desc, uint32_t *new_index /* [n_dofs] */);

/* ---- what a CG / Chebyshev caller needs from the operator besides vmult -------------
 * LaplaceOperatorGpu::compute_diagonal + get_diagonal_inverse (laplace_operator_gpu.h:401-429): writes
 * 1 / diag(A) into inv_diag[n_dofs] (device, operator's number type); the local diagonal of every cell
 * (DiagonalLocalOperator, :355-399) is distributed like a cell result, including the transposed hanging-node
 * resolution, constrained rows are set to 1 before the inversion (:412-414).                              */
int mfgpu_compute_inverse_diagonal(mfgpu_handle *h, void *inv_diag, void *stream);
/* ConstraintHandlerGpu::set_constrained_values (constraint_handler_gpu.cu:126-137): vec[c] = value for every
 * constrained dof c of the description.                                                                   */
int mfgpu_set_constrained_values(mfgpu_handle *h, void *vec, double value, void *stream);

/* ---- GpuVector BLAS-1 and reductions (gpu_vec.h:105-157, gpu_vec.cu:222-617) -------
 * v, w, x: device vectors of n elements of number_type.  The element-wise operations are asynchronous on
 * `stream`; the reductions block until the result is on the host (as the reference's do, gpu_vec.cu:556-560)
 * and accumulate in double in a fixed order (deterministic).                                               */
int mfgpu_vec_sadd(void *v, double s, double a, const void *w, size_t n, int number_type, void *stream);  /* v = s v + a w   gpu_vec.cu:308-314 */
int mfgpu_vec_equ(void *v, double a, const void *w, size_t n, int number_type, void *stream);             /* v = a w         :346-352 */
int mfgpu_vec_scale(void *v, const void *w, size_t n, int number_type, void *stream);                     /* v[i] *= w[i]    :320-325 */
int mfgpu_vec_divide(void *v, const void *w, size_t n, int number_type, void *stream);                    /* v[i] /= w[i]    :328-333 */
int mfgpu_vec_invert(void *v, size_t n, int number_type, void *stream);                                   /* v[i] = 1/v[i]   :336-342 */
int mfgpu_vec_mul(void *v, double a, size_t n, int number_type, void *stream);                            /* v *= a          :357-362 */
int mfgpu_vec_dot(const void *v, const void *w, size_t n, int number_type, void *stream, double *result); /* v . w           :542-563 */
int mfgpu_vec_l2_norm(const void *v, size_t n, int number_type, void *stream, double *result);            /* sqrt(v . v)     :367-369 */
int mfgpu_vec_add_and_dot(void *v, double a, const void *x, const void *w, size_t n, int number_type,
                          void *stream, double *result);                              /* v += a x; return v . w   :597-617 */
int mfgpu_vec_all_zero(const void *v, size_t n, int number_type, void *stream, int *result);              /* :512-540 */

/* ---- GpuVector pieces that are on the path (gpu_vec.h:44,69,84-88,164-172) -------------- */
int mfgpu_vec_alloc(void **dev, size_t n, int number_type);               /* gpu_vec.cu:166-182, zero-filled */
int mfgpu_vec_free(void *dev);
int mfgpu_vec_fill(void *dev, size_t n, int number_type, double value, void *stream); /* vec_init, gpu_vec.cu:281-291 */
int mfgpu_vec_from_host(void *dev, const void *host, size_t n, int number_type);
int mfgpu_vec_to_host(void *host, const void *dev, size_t n, int number_type);
int mfgpu_device_synchronize(void);
/* free / total bytes of the current device (hipMemGetInfo): lets callers and tests check that
 * mfgpu_destroy / mfgpu_vec_free return everything mfgpu_create / mfgpu_vec_alloc took              */
int mfgpu_device_memory_info(size_t *free_bytes, size_t *total_bytes); /* bmop.cu:148 */

/* ---- single-node multi-GPU mode (no reference counterpart: the reference is single-GPU) --------
 * One process per GPU; the mesh is cut into z-slabs (mfgpu_mesh_create_uniform's slab arguments), each rank owns the
 * vector entries of its slab including both interface planes.  After the slab's cell loop an interface plane holds
 * only the rank's own cells' contributions; the exchange adds the neighbour's, leaving the full sum on both sharers
 * (constrained rows are identity rows on both sides and are not summed).  Transport: RCCL point-to-point send / recv
 * with the two z-neighbours on a side stream, overlapped with pass 2 of the slab's other dofs.                       */
typedef struct mfgpu_dist mfgpu_dist;
/* rank 0: a 128-byte RCCL unique id to hand to every rank (any out-of-band channel) */
int mfgpu_dist_unique_id(void *id128);
/* collective over all ranks when id128 != NULL and world > 1 (ncclCommInitRank on the current HIP device).
 * lower_ids / upper_ids: local dof ids of the slab's lower / upper interface plane in the order both neighbours
 * agree on (mfgpu_mesh_interface_dofs); constrained: the description's constrained_dofs.  id128 == NULL: no RCCL
 * communicator; the neighbours are connected in-process with mfgpu_dist_connect_local (tests).                    */
int mfgpu_dist_create(const void *id128, int rank, int world, const uint32_t *lower_ids, uint32_t n_lower,
                      const uint32_t *upper_ids, uint32_t n_upper, const uint32_t *constrained_dofs,
                      uint32_t n_constrained, uint32_t n_dofs, int number_type, mfgpu_dist **out);
int mfgpu_dist_connect_local(mfgpu_dist *lower_rank, mfgpu_dist *upper_rank);
/* tells the operator which dofs the exchange needs first (re-orders its pass 2); once per (operator, dist) pair */
int mfgpu_dist_attach(mfgpu_dist *d, mfgpu_handle *h);
/* the schedule mfgpu_dist_attach chose: info[0] = 1: interface-first (the batches [0, info[1]) and
 * [info[2], info[3]) touch an interface plane and run first, their pass 2, the pack and the exchange run on a side
 * stream next to the interior batches [info[1], info[2]) and the rest of pass 2); 0: whole cell loop, then the exchange
 * next to pass 2 of the non-interface dofs (thin slabs, segmented or coloured cell loops)                              */
int mfgpu_dist_schedule(const mfgpu_dist *d, uint32_t info[4]);
/* dst = A src on the slab, then the exchange.  _begin: cell loop, pass 2 of the interface dofs, start of the
 * transfers (side stream), pass 2 of the rest; _end: wait for the transfers, add.  mfgpu_vmult_dist = both (RCCL
 * transport or world == 1; with the in-process transport call _begin on every slab before any _end).             */
int mfgpu_vmult_dist_begin(mfgpu_handle *h, mfgpu_dist *d, void *dst_dev, const void *src_dev, void *stream);
int mfgpu_vmult_dist_end(mfgpu_handle *h, mfgpu_dist *d, void *dst_dev, void *stream);
int mfgpu_vmult_dist(mfgpu_handle *h, mfgpu_dist *d, void *dst_dev, const void *src_dev, void *stream);
void mfgpu_dist_destroy(mfgpu_dist *d);

/* ---- multigrid level transfer (MGTransferMatrixFreeGpu, mg_transfer_matrix_free_gpu.h:140-252,
 * mg_transfer_matrix_free_gpu.cu:391-660) between two globally refined levels.  The boundary is the output of
 * MGTransferMatrixFreeGpu::build (:150-330): per coarse cell its (p+1)^dim level dofs and the (2p+1)^dim level dofs of
 * the patch of its children, both lexicographic (level_dof_indices), the coarse level's Dirichlet dofs
 * (dirichlet_indices) and the 1D prolongation matrix (shape_values; NULL = FE_Q on Gauss-Lobatto nodes).  The level
 * operators are ordinary mfgpu_handles of the level meshes (LaplaceOperatorGpu::reinit(dof_handler,
 * mg_constrained_dofs, level), laplace_operator_gpu.h:154-186).                                                      */
typedef struct mfgpu_transfer mfgpu_transfer;
int mfgpu_transfer_create(int dim, int degree, int number_type, uint32_t n_coarse_cells,
                          const uint32_t *coarse_cell_dofs /* [n_coarse_cells * (p+1)^dim]  */,
                          const uint32_t *fine_patch_dofs  /* [n_coarse_cells * (2p+1)^dim] */, uint32_t n_coarse_dofs,
                          uint32_t n_fine_dofs, const uint32_t *coarse_dirichlet, uint32_t n_coarse_dirichlet,
                          const double *prolongation_1d /* [(2p+1) * (p+1)], fine index major, or NULL */,
                          mfgpu_transfer **out);
/* :595-627  dst_fine = P (src_coarse with the coarse Dirichlet dofs read as 0); every fine dof is written */
int mfgpu_transfer_prolongate(mfgpu_transfer *t, void *dst_fine_dev, const void *src_coarse_dev, void *stream);
/* :631-660  dst_coarse += P^T src_fine on the non-Dirichlet coarse dofs (floating-point atomics, as the reference) */
int mfgpu_transfer_restrict_and_add(mfgpu_transfer *t, void *dst_coarse_dev, const void *src_fine_dev, void *stream);
size_t mfgpu_transfer_memory_consumption(const mfgpu_transfer *t); /* :333-347 */
void mfgpu_transfer_destroy(mfgpu_transfer *t);

/* Level operator with refinement edges (laplace_operator_gpu.h:154-186, 306-352): `desc` describes the level mesh with
 * the level's Dirichlet dofs as constrained_dofs (no hanging nodes on a level mesh, :174-176), edge_dofs are the level's
 * refinement-edge dofs (MGConstrainedDoFs::get_refinement_edge_indices).  mfgpu_level_operator is the level matrix
 * (constrained rows = Dirichlet + edge dofs: vmult, inverse diagonal ... through the ordinary calls; owned by the level);
 * the interface matrices are what deal.II's Multigrid::set_edge_matrices takes (poisson_mg.cu:367-375).               */
typedef struct mfgpu_level mfgpu_level;
int mfgpu_level_create(const mfgpu_desc *desc, const uint32_t *edge_dofs, uint32_t n_edge, mfgpu_level **out);
mfgpu_handle *mfgpu_level_operator(mfgpu_level *level);
int mfgpu_level_vmult_interface_down(mfgpu_level *level, void *dst_dev, const void *src_dev, void *stream); /* :306-330 */
int mfgpu_level_vmult_interface_up(mfgpu_level *level, void *dst_dev, const void *src_dev, void *stream);   /* :332-352 */
void mfgpu_level_destroy(mfgpu_level *level);

/* copy_to_mg / copy_from_mg (mg_transfer_matrix_free_gpu.cu:690-760): dst[dst_idx[i]] = src[src_idx[i]] for the index
 * pairs of copy_indices (active dof <-> level dof), kept on the device                                              */
typedef struct mfgpu_index_pairs mfgpu_index_pairs;
int mfgpu_index_pairs_create(const uint32_t *dst_idx, const uint32_t *src_idx, uint32_t n, mfgpu_index_pairs **out);
int mfgpu_vec_copy_pairs(const mfgpu_index_pairs *p, void *dst_dev, const void *src_dev, int number_type, void *stream);
void mfgpu_index_pairs_destroy(mfgpu_index_pairs *p);

/* ---- deal.II stand-in for the setup side (host only) --------------------------------------
 * Produces what Triangulation + DoFHandler + ConstraintMatrix + FEValues + ShapeInfo hand to
 * MatrixFreeGpu::reinit, for the meshes bmop uses (bmop_common.h:108-120).                    */
typedef struct mfgpu_mesh mfgpu_mesh;

/* hyper_cube(lo,hi) subdivided n_per_dir[d] times per direction (generalises refine_global,
 * poisson_common.h:62-64 + bmop_common.h:119); cells_z_begin/end select a z-slab (last
 * direction) of cells for the multi-GPU partition: DoFs are renumbered slab-locally.         */
int mfgpu_mesh_create_uniform(int dim, int degree, const uint32_t *n_per_dir, double lo, double hi,
                              uint32_t slab_begin, uint32_t slab_end, int number_type,
                              mfgpu_mesh **out);
/* bmop_common.h:49-105 pseudo_adaptive_refinement on the cube (ADAPTIVE_GRID), n_ref as in
 * bmop's argv; octree with 2:1 balance and hanging-node constraints.                          */
int mfgpu_mesh_create_adaptive(int dim, int degree, int n_ref, int number_type, mfgpu_mesh **out);
/* BALL domain of bmop / poisson (-DBALL_GRID; poisson_common.h:65-70, bmop_common.h:108-120): hyper_ball (unit
 * ball, 5 / 7 coarse cells), spherical manifold on the boundary, n_ref global refinements, MappingQ1.  Unstructured;
 * the description has a full J^-1 per quadrature point (no MFGPU_UNIFORM_J0).                                     */
int mfgpu_mesh_create_ball(int dim, int degree, int n_ref, int number_type, mfgpu_mesh **out);
/* MGTransferMatrixFreeGpu::build for two stand-in meshes (uniform cubes n and 2n cells per direction, or ball meshes
 * of n_ref and n_ref + 1): the transfer between them, in the fine mesh's number type                               */
int mfgpu_transfer_create_from_meshes(const mfgpu_mesh *coarse, const mfgpu_mesh *fine, mfgpu_transfer **out);
/* the index arrays that call hands to mfgpu_transfer_create, on the host (no GPU needed): coarse_cell_dofs
 * [n_cells * (p+1)^dim], fine_patch_dofs [n_cells * (2p+1)^dim], n_cells = the coarse mesh's; returns n_cells     */
int64_t mfgpu_mesh_transfer_patches(const mfgpu_mesh *coarse, const mfgpu_mesh *fine, uint32_t *coarse_cell_dofs,
                                    uint32_t *fine_patch_dofs);
/* DoFHandler::renumber_dofs on the stand-in: loc2glob, constrained dofs, dof coordinates, interface planes */
int mfgpu_mesh_renumber(mfgpu_mesh *m, const uint32_t *new_index);
/* the same recipe with Triangulation::limit_level_difference_at_vertices (2:1 over vertices too), which the reference's
 * multigrid programs set (poisson_mg.cu:131): required by mfgpu_mg_hierarchy_create                                  */
int mfgpu_mesh_create_adaptive_mg(int dim, int degree, int n_ref, int number_type, mfgpu_mesh **out);
/* Multigrid level hierarchy of an adaptive octree stand-in mesh (host only): what distribute_mg_dofs +
 * MGConstrainedDoFs + MGTransferMatrixFreeGpu::build provide on a locally refined mesh (poisson_mg.cu:152,199-209,
 * 325-326).  Level l = all cells of level l; see csrc/mfgpu_mg_hierarchy.cpp.  The level meshes belong to the hierarchy. */
typedef struct mfgpu_mg_hierarchy mfgpu_mg_hierarchy;
int mfgpu_mg_hierarchy_create(const mfgpu_mesh *adaptive, mfgpu_mg_hierarchy **out);
int mfgpu_mg_n_levels(const mfgpu_mg_hierarchy *h);
const mfgpu_mesh *mfgpu_mg_level_mesh(const mfgpu_mg_hierarchy *h, int level);
int64_t mfgpu_mg_edge_dofs(const mfgpu_mg_hierarchy *h, int level, const uint32_t **ptr); /* refinement-edge dofs */
/* copy_indices of the level: returns n, (active dof, level dof) pairs */
int64_t mfgpu_mg_copy_pairs(const mfgpu_mg_hierarchy *h, int level, const uint32_t **active_dofs, const uint32_t **level_dofs);
/* arrays for mfgpu_transfer_create(level - 1 -> level): returns the number of refined cells of level - 1 */
int64_t mfgpu_mg_transfer_arrays(const mfgpu_mg_hierarchy *h, int level, const uint32_t **coarse_cell_dofs,
                                 const uint32_t **fine_patch_dofs);
void mfgpu_mg_hierarchy_destroy(mfgpu_mg_hierarchy *h);
/* same setup from an explicit one-irregular set of octree leaves (level, cx, cy, cz) x n_leaves on
 * hyper_cube(-1,1): lets tests build the awkward small cases of test_hanging_nodes_gpu.cu:297-331 */
int mfgpu_mesh_create_from_leaves(int dim, int degree, const uint32_t *leaves, uint32_t n_leaves,
                                  int number_type, mfgpu_mesh **out);
/* (level, cx, cy, cz) of every cell in mesh order [n_cells*4]; empty for uniform meshes */
int64_t mfgpu_mesh_cell_levels(const mfgpu_mesh *m, const uint32_t **ptr);
void mfgpu_mesh_destroy(mfgpu_mesh *m);
/* fills *desc with pointers into the mesh (valid until mfgpu_mesh_destroy) */
int mfgpu_mesh_desc(const mfgpu_mesh *m, mfgpu_desc *desc);
/* support-point coordinates of every DoF [n_dofs*dim] (double) */
int64_t mfgpu_mesh_dof_coords(const mfgpu_mesh *m, const double **ptr);
/* multi-GPU slabs: local indices of the DoFs on the lower / upper slab interface plane, in the
 * same (lexicographic) order on both neighbours; which = 0 lower, 1 upper                      */
int64_t mfgpu_mesh_interface_dofs(const mfgpu_mesh *m, int which, const uint32_t **ptr);

#ifdef __cplusplus
}
#endif
#endif /* MFGPU_H */
