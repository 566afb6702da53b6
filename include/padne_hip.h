/*
 * padne_hip.h -- C ABI of libpadne_hip.so, the MI355X (gfx950) implementation of
 * the padne solver hot path.
 *
 * The reference (atx/padne) has no FFI for this path: the seam is the Python
 * module padne/solver.py.  Each entry point below names the reference code whose
 * arithmetic it replaces (file:line in the reference tree); INTEGRATION.md shows
 * the ctypes stub a padne maintainer would add to call them.
 *
 * Conventions
 *   - every function returns 0 on success, a negative PADNE_E_* code on failure;
 *     padne_last_error() returns a thread-local human readable message.
 *     Nothing throws across the boundary.
 *   - "host" pointers are caller-owned host memory; "dev" pointers are raw device
 *     addresses (hipMalloc / torch tensor.data_ptr()) valid on the context's GPU.
 *   - all calls are blocking unless the name ends in _async.
 *   - indices are int32 inside a matrix (nnz < 2^31), sizes are int64.
 *   - all floating point data that crosses this boundary is IEEE binary64 (solver.py:21, DTYPE = float64), and so are
 *     assembly, the solver's own products, residuals and dot products; only the multigrid V-cycle INSIDE the
 *     preconditioner runs on single-precision copies of its operators (it has to be a fixed SPD operator, not an
 *     accurate one; PADNE_AMG_F64=1 keeps it in binary64).
 */
#ifndef PADNE_HIP_H
#define PADNE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PADNE_ABI_VERSION 1

#define PADNE_OK            0
#define PADNE_E_INVALID    -1   /* bad argument (null pointer, negative size, index out of range) */
#define PADNE_E_HIP        -2   /* a HIP runtime call failed */
#define PADNE_E_NOMEM      -3
#define PADNE_E_NONMANIFOLD -4  /* triangle soup is not an oriented manifold (mesh.py:342-343 ValueError) */
#define PADNE_E_NOTCONVERGED -5 /* PCG hit max_iter (solution still returned) */
#define PADNE_E_COMM       -6   /* RCCL not available / communicator failure */
#define PADNE_E_BREAKDOWN  -7   /* PCG breakdown: matrix not SPD (p.Ap <= 0) or NaN */
#define PADNE_E_TOOLARGE   -8   /* a count exceeds the 32-bit index space of the CSR structures (nnz, slot offsets) */
#define PADNE_E_NOCOARSEN  -9   /* multigrid setup: the aggregation no longer shrinks the operator.  Internal to the
                                   solve (which then preconditions with the diagonal, info.levels = 0); returned only by
                                   the entry points that expose the hierarchy itself (padne_amg_apply, padne_amg_level) */

typedef struct padne_ctx padne_ctx;   /* device, stream, workspaces, optional RCCL communicator */
typedef struct padne_csr padne_csr;   /* device-resident CSR matrix (f64 values, i32 indices)  */
typedef struct padne_kkt padne_kkt;   /* device-resident plan of solve_system for one assembled system */

/* ---- library / context ------------------------------------------------------------------- */
int         padne_abi_version(void);
const char *padne_last_error(void);
/* number of visible GPUs, or a negative error code */
int         padne_device_count(void);
/* create a context on `device` with its own non-blocking stream */
int         padne_ctx_create(int device, padne_ctx **out);
int         padne_ctx_destroy(padne_ctx *ctx);
int         padne_ctx_synchronize(padne_ctx *ctx);
/* raw hipStream_t of the context (for event timing by the caller) */
void       *padne_ctx_stream(padne_ctx *ctx);

/* ---- multi-GPU (RCCL over xGMI; one process per GPU) --------------------------------------
 * No reference counterpart: the reference is single-process.
 * padne_comm_unique_id fills 128 bytes on one rank; the caller broadcasts them
 * (torch.distributed) and every rank calls padne_ctx_comm_init. */
int padne_comm_unique_id(void *id128);
int padne_ctx_comm_init(padne_ctx *ctx, const void *id128, int rank, int world_size);
int padne_ctx_comm_rank(padne_ctx *ctx, int *rank, int *world_size);
/* communication this process has issued since the library was loaded: calls[0..2] / bytes[0..2] = the COLLECTIVES
 * all-reduce (f64 scalars), all-gather of f64 values, all-gather of f32 values; calls[3] / bytes[3] = peer-to-peer halo
 * exchanges (every rank stores its exported values into the other ranks' mailboxes: no collective).  Bytes are this
 * rank's contributions.  Bookkeeping for DESIGN.md section 6 (what a multi-GPU solve costs), also counted for the
 * in-process team. */
int padne_comm_call_counts(long long calls[4], long long bytes[4]);
/* kernels and asynchronous fills this process has queued through the library since it was loaded (all contexts).  What a
 * solve costs in LAUNCHES is read off the difference around it: below a million unknowns per GPU a solve is bound by its
 * launches, not by bytes (bench.py: rank_proxy). */
int padne_launch_count(long long *count);
/* The same collectives through a transport of the CALLER (gloo, MPI, ...) instead of RCCL: ranks RCCL cannot connect
 * -- two processes on one GPU -- or a host that has no RCCL.  `allgather(user, send, recv, bytes_per_rank)` gathers
 * bytes_per_rank bytes of HOST memory from every rank into recv (rank order; send may lie inside recv) and returns 0;
 * any other value fails the call that needed it with PADNE_E_COMM.  Every collective of the library then goes device ->
 * host -> callback -> device (sums are formed in rank order: the same bits on every rank, as with RCCL); it is the
 * slow path by construction -- what makes it usable is the peer-to-peer halo exchange below, which takes the four
 * exchanges of a CG iteration out of the collectives. */
typedef int (*padne_allgather_fn)(void *user, const void *send, void *recv, int64_t bytes_per_rank);
int padne_ctx_comm_init_host(padne_ctx *ctx, int rank, int world_size, padne_allgather_fn allgather, void *user);
/* Peer-to-peer halo exchange between PROCESSES (one per GPU, or several on one GPU): every rank owns a mailbox in
 * device memory -- a ring of exchange entries of world_size * slots_per_rank 8-byte cells behind a page of arrival
 * flags -- allocated uncached and shared through hipIpc.  A halo exchange is then: every rank STORES its exported
 * values straight into every rank's mailbox and, when its stores are out (system-scope release), writes the exchange's
 * sequence number into its flag there; the receiver's next kernel on the stream waits for the flags of all senders
 * (bounded: PADNE_P2P_TIMEOUT_MS, default 20 s, then the solve returns PADNE_E_COMM) and copies the entry behind its
 * owned values.  No collective, no host involvement, and the interior tiles of the product the exchange is for run
 * while the stores travel.  Protocol, collective over the ranks of a communicator (RCCL or host):
 *   padne_ctx_p2p_export   allocate this rank's mailbox, write its 64-byte hipIpc handle to handle64
 *   -- the caller all-gathers the handles (rank order) --
 *   padne_ctx_p2p_import   open the other ranks' mailboxes (handles = world_size * 64 bytes); from here on exchanges
 *                          whose plan has at most slots_per_rank slots per rank go peer to peer
 *   padne_ctx_p2p_selftest one real exchange of known values through the mailboxes (2 s bound): *ok = 1 if every rank's
 *                          stores and flags arrived here -- the caller gathers the verdicts and closes unless all say yes
 *   padne_ctx_p2p_close    back to the all-gather (also done by padne_ctx_destroy); collective like the ones above
 * PADNE_NO_P2P=1 keeps the all-gather although mailboxes exist (A/B, tests). */
int padne_ctx_p2p_export(padne_ctx *ctx, int32_t slots_per_rank, void *handle64);
int padne_ctx_p2p_import(padne_ctx *ctx, const void *handles, int32_t n_handles);
int padne_ctx_p2p_selftest(padne_ctx *ctx, int32_t *ok);
int padne_ctx_p2p_close(padne_ctx *ctx);

/* Halo plan of a row-partitioned matrix (layer partition).  Every vector the
 * local matrix multiplies is laid out [n_owned owned entries | world_size * m exchanged entries];
 * before each product the rank copies its `n_export` values export_idx[k] (local indices) into its
 * own segment and one ncclAllGather (m doubles per rank) fills the rest, so a column that refers to
 * unknown export_idx[k] of rank r is column n_owned + r*m + k.  The local matrix therefore has
 * n_owned + world_size*m columns and its first n_owned rows are the owned equations (further rows
 * must be empty).  With a communicator set, all PCG dot products are summed over the ranks by
 * ncclAllReduce on the context stream.  n_owned < 0 clears the plan. */
int padne_ctx_set_halo(padne_ctx *ctx, int64_t n_owned, int32_t m, int32_t n_export,
                       const int32_t *export_idx_host);

/* ---- device memory helpers (thin, so that callers need no HIP binding of their own) -------- */
int padne_dev_alloc(padne_ctx *ctx, int64_t bytes, void **dev_out);
int padne_dev_free(padne_ctx *ctx, void *dev);
int padne_dev_upload(padne_ctx *ctx, void *dev_dst, const void *host_src, int64_t bytes);
int padne_dev_download(padne_ctx *ctx, void *host_dst, const void *dev_src, int64_t bytes);
int padne_dev_memset(padne_ctx *ctx, void *dev, int value, int64_t bytes);

/* ---- matrices ------------------------------------------------------------------------------ */
/* upload a host CSR (scipy layout: int32 indptr[n_rows+1], int32 indices[nnz], f64 data[nnz]).
 * Replaces L.tocsc() as the hand-off of the assembled system, solver.py:772.
 * The columns of a row need not ascend (scipy's canonical form does; the layout does not demand it): the upload remembers a
 * matrix whose rows do not, and the paths that count on column order (the order-preserving relabel of padne_csr_reduce)
 * leave such a matrix to the general ones, which sort. */
int padne_csr_from_host(padne_ctx *ctx, int64_t n_rows, int64_t n_cols,
                        const int32_t *indptr, const int32_t *indices, const double *data,
                        padne_csr **out);
int padne_csr_destroy(padne_csr *m);
int padne_csr_shape(const padne_csr *m, int64_t *n_rows, int64_t *n_cols, int64_t *nnz);
/* copy back to caller-allocated host arrays (sizes from padne_csr_shape) */
int padne_csr_to_host(padne_ctx *ctx, const padne_csr *m, int32_t *indptr, int32_t *indices, double *data);

/* Assemble the global system matrix L in the reference layout and sign.
 *   mesh part  : HalfEdge.cotan (mesh.py:124-139), laplace_operator (solver.py:171-213),
 *                process_mesh_laplace_operators (solver.py:563-575)
 *   lumped part: the COO stamps produced by stamp_network_into_system / setup_ground_node
 *                (solver.py:469-560), already in global indices, in stamp order.
 * xy[n_vert][2], tri[n_tri][3] hold mesh-LOCAL vertex ids; mesh m owns vertices
 * [mesh_vertex_offset[m], mesh_vertex_offset[m+1]) (VertexIndexer, solver.py:221-229) and
 * triangles [mesh_tri_offset[m], mesh_tri_offset[m+1]).  Duplicate (row, col) stamps are summed
 * in stamp order after the mesh contribution; exact zeros are not stored (solver.py:187-190).
 * Returns PADNE_E_NONMANIFOLD where Mesh.from_triangle_soup raises ValueError (mesh.py:342-343).
 * The matrix arrays are allocated for an upper bound of the entries (one per triangle corner, two per vertex, the
 * stamps) and the rows are written once, in place; padne_csr_shape reports the exact count.  PADNE_E_TOOLARGE when
 * that bound exceeds the 32-bit index space of a CSR matrix (about 268 M mesh vertices).  The single-pass row kernel
 * finds its offsets with a scan that runs inside it and whose waits are bounded; on a chip shared with other work such a
 * wait can run out -- the rows are then built again in two passes (lengths, scan, fill: the same bits, about three times
 * the time), never an error of the call. */
int padne_assemble_system(padne_ctx *ctx, int64_t n_unknowns,
                          int64_t n_vert, const double *xy_host,
                          int64_t n_tri, const int32_t *tri_host,
                          int64_t n_mesh, const int64_t *mesh_vertex_offset,
                          const int64_t *mesh_tri_offset, const double *conductance,
                          int64_t n_coo, const int64_t *coo_row, const int64_t *coo_col,
                          const double *coo_val,
                          padne_csr **out);
/* Structured test / benchmark mesh generated on the device (no reference counterpart: it stands where the CGAL mesher
 * stands, padne/mesh.py:662-795, for the synthetic configs of SURVEY.md section 8d): nx * ny vertices spaced h from
 * (origin_x, origin_y), row-major, interior vertices displaced by U(-jitter h, +jitter h), cells split by alternating
 * diagonals into counter-clockwise triangles.  The displacements are output 2 v + 1 and 2 v + 2 of numpy's PCG64
 * stream whose 128-bit state and increment are pcg64_state_inc[0..3] = state hi, state lo, inc hi, inc lo
 * (default_rng(seed).bit_generator.state), so the arrays equal padne_amd.synthetic.jittered_grid bit for bit.
 * xy_dev: 2 nx ny doubles, tri_dev: 6 (nx-1)(ny-1) int32, both device memory (padne_dev_alloc). */
int padne_generate_grid_mesh(padne_ctx *ctx, int64_t nx, int64_t ny, double h, double jitter, double origin_x,
                             double origin_y, const uint64_t *pcg64_state_inc, void *xy_dev, void *tri_dev);

/* The same for ONE RANK'S PIECE of a mesh that is partitioned across GPUs (flags bit 0): the triangles are those that
 * touch a vertex the rank owns, so the vertices of the ring around the owned region have incomplete fans and the
 * manifold test (PADNE_E_NONMANIFOLD) is switched off -- the rows of ring vertices are dropped by the caller
 * (padne_csr_relabel); validate the whole mesh on one rank instead.  flags = 0 is padne_assemble_system.
 * With either entry point `xy_host` / `tri_host` may also be DEVICE pointers (e.g. filled by padne_generate_grid_mesh):
 * the kernels then read the caller's arrays, the copy the matrix keeps for padne_csr_power_density is made device to device
 * on the context's second stream beside them, and nothing crosses PCIe (the arrays must stay valid until the call returns). */
int padne_assemble_system_ex(padne_ctx *ctx, int64_t n_unknowns, int64_t n_vert, const double *xy_host,
                             int64_t n_tri, const int32_t *tri_host, int64_t n_mesh,
                             const int64_t *mesh_vertex_offset, const int64_t *mesh_tri_offset,
                             const double *conductance, int64_t n_coo, const int64_t *coo_row,
                             const int64_t *coo_col, const double *coo_val, int32_t flags, padne_csr **out);

/* out = scale * P^T M P restricted to kept indices: entry (i,j,v) of M becomes
 * (map[i], map[j], scale*v) if both maps are >= 0; duplicates are summed.  Used to turn the
 * reference's indefinite KKT system into the SPD system A = -L_vv on the free potentials
 * (ground eliminated, voltage-source-tied nodes merged; DESIGN.md "reduction"). */
int padne_csr_reduce(padne_ctx *ctx, const padne_csr *m, const int32_t *map_host,
                     int64_t n_out, double scale, padne_csr **out);
/* Rectangular form of the same operation with separate row and column maps (row_map has m.n_rows entries,
 * col_map m.n_cols): out (n_rows_out x n_cols_out) = scale * R^T M C.  The row-partitioned solve uses it
 * to keep exactly the owned rows of a rank while the columns keep the exchange slots (DESIGN.md
 * "multi-GPU"). */
int padne_csr_relabel(padne_ctx *ctx, const padne_csr *m, const int32_t *row_map_host, int64_t n_rows_out,
                      const int32_t *col_map_host, int64_t n_cols_out, double scale, padne_csr **out);
/* rows of `top` followed by the rows of `bottom` (equal column counts) */
int padne_csr_vstack(padne_ctx *ctx, const padne_csr *top, const padne_csr *bottom, padne_csr **out);

/* ---- SpMV ---------------------------------------------------------------------------------- */
/* y = M x, host vectors (the reference's residual product L_csc @ v, solver.py:775) */
int padne_spmv(padne_ctx *ctx, const padne_csr *m, const double *x_host, double *y_host);
/* y = M x on device vectors, enqueued `repeat` times on the context stream; blocking at the end */
int padne_spmv_dev(padne_ctx *ctx, const padne_csr *m, const void *x_dev, void *y_dev, int repeat);
/* Y = M X for 8 right-hand sides at once, device arrays interleaved as X[i*8 + j] = entry i of vector j
 * (X: n_cols x 8, Y: n_rows x 8); every column of Y is bit-identical to padne_spmv_dev on that vector.
 * The batched solve (padne_solve_spd with n_rhs a multiple of 8) is built on it. */
int padne_spmm8_dev(padne_ctx *ctx, const padne_csr *m, const void *x_dev, void *y_dev, int repeat);
/* residual: returns ||M x - b||_2 (device), host vectors in */
int padne_residual_norm(padne_ctx *ctx, const padne_csr *m, const double *x_host,
                        const double *b_host, double *norm_out);

/* ---- solve --------------------------------------------------------------------------------- */
typedef struct padne_solve_opts {
    double  rtol;        /* stop when ||b - A x||_2 <= max(rtol*||b||_2, atol); if the TRUE residual
                            stagnates above that (binary64 evaluation floor of b - A x, ~1e-12 at
                            N = 5 M) the solve ends successfully within 10x of the request and
                            reports what it reached in padne_solve_info.rel_residual           */
    double  atol;
    int32_t max_iter;
    int32_t precond;     /* 0 = Jacobi, 1 = smoothed-aggregation multigrid V-cycle (the hierarchy is built on
                            first use and cached on the matrix; the cycle runs in single precision inside
                            the double-precision CG unless PADNE_AMG_F64 is set; on a row-partitioned
                            matrix it is one hierarchy over all ranks, or block-Jacobi with
                            padne_csr_set_preconditioner_block)                                    */
    int32_t check_every; /* iterations enqueued between host convergence polls (0 = auto) */
    int32_t flags;       /* bit0: x holds an initial guess (otherwise x0 = 0)
                            bit1: time sampled SpMV launches with HIP events -> info.spmv_seconds
                            bit2: rebuild everything derived from the matrix inside this call (multigrid
                                  hierarchy, single-precision copies, x-window plan of the SpMV)   */
} padne_solve_opts;

typedef struct padne_solve_info {
    int32_t iterations;
    int32_t restarts;        /* true-residual restarts taken                     */
    double  rel_residual;    /* final TRUE ||b - A x|| / ||b||                    */
    double  abs_residual;
    double  solve_seconds;   /* device time of the iteration loop (HIP events)    */
    double  spmv_seconds;    /* average device time of one SpMV launch (flags bit1) */
    int32_t status;          /* PADNE_OK / PADNE_E_NOTCONVERGED / PADNE_E_BREAKDOWN */
    int32_t n_rhs;
    double  precond_setup_seconds; /* device time of the multigrid setup done inside this call (0 if cached) */
    double  operator_complexity;   /* sum of nnz over the levels / nnz of the fine matrix                    */
    int32_t levels;                /* multigrid levels (0 with Jacobi)                                       */
    int32_t precond_fallbacks;     /* right-hand sides redone with Jacobi after a multigrid breakdown/stall  */
} padne_solve_info;

/* Preconditioned CG on an SPD CSR matrix: replaces scipy.sparse.linalg.spsolve in
 * solve_system (solver.py:773) once the system is reduced.  b, x: host f64[n_rhs][n]
 * (row-major, one right-hand side after another).  With the multigrid preconditioner on one GPU, groups of
 * 8 right-hand sides (remainders of 5-7 zero-padded to 8, a remainder of exactly 4 in a group of width 4) advance in
 * lockstep: one pass over the matrix and the hierarchy per iteration for the whole group; results do not depend on the
 * grouping beyond the tolerance.  PADNE_E_NOTCONVERGED still returns the best iterate in x and the residual reached in info. */
int padne_solve_spd(padne_ctx *ctx, const padne_csr *a, const double *b_host, double *x_host,
                    int32_t n_rhs, const padne_solve_opts *opts, padne_solve_info *info);
/* same with device-resident b and x */
int padne_solve_spd_dev(padne_ctx *ctx, const padne_csr *a, const void *b_dev, void *x_dev,
                        int32_t n_rhs, const padne_solve_opts *opts, padne_solve_info *info);

/* ---- solve_system as a whole (solver.py:767-780: L.tocsc(), spsolve, residual) ---------------------------------------
 * The reference hands the indefinite KKT matrix (multiplier rows of voltage sources / regulators / the ground,
 * solver.py:493-538, 544-560) to SuperLU.  Here it is reduced to the SPD system A y = b on the free potentials
 *  and a padne_kkt plan keeps that reduction on the device: the index map, A = -P^T L P with its
 * multigrid hierarchy, and the N-vectors r, v.  The host describes the reduction by O(#constraints) lists:
 *   elim_sorted[n_elim]   potentials without a reduced unknown of their own, ascending: potentials known outright (the
 *                         ground, nodes tied to it by sources, pins of floating copper) and the members of source-tied
 *                         groups other than the group's representative;
 *   tied_member / tied_rep[n_tied]  those members (ascending) and the representative they are numbered through;
 *   index_map_host        optional int32[N] (NULL: built on the device as  i - #{e in elim : e < i}): a map the caller
 *                         made itself; n_free = number of reduced unknowns.
 *   flags                 bit 0: number the reduced unknowns by horizontal strips of the meshes `L` was assembled from
 *                         (mesh, strip of about three vertex spacings, x; unknowns that are no vertex behind them) -- the
 *                         band numbering the SpMV's x windows need when the mesher numbered the vertices in insertion
 *                         order (CGAL).  Internal to the plan: r and v keep the caller's numbering.  Needs a matrix from
 *                         padne_assemble_system (it carries its mesh); PADNE_E_INVALID if the keys do not fit
 *                         (>= 65535 meshes, > 65535 strips in a mesh): the caller may then pass its own map.
 * `L` is borrowed and must outlive the plan. */
int padne_kkt_create(padne_ctx *ctx, const padne_csr *L, int64_t n_potential, int64_t n_elim,
                     const int64_t *elim_sorted, int64_t n_tied, const int64_t *tied_member, const int64_t *tied_rep,
                     const int32_t *index_map_host, int64_t n_free, int32_t flags, padne_kkt **out);
int padne_kkt_destroy(padne_kkt *plan);
/* borrowed handle of the reduced SPD matrix (for introspection; do not destroy) */
int padne_kkt_matrix(const padne_kkt *plan, const padne_csr **reduced_out);
/* Stage 1 of a solve.  r_host[N]: the right-hand side (solver.py:757-760, 478-541), uploaded ONCE and in parallel with the
 * multigrid setup of A (built on the first call and kept with the plan; opts.flags bit 2 rebuilds it).
 * known_idx / known_val: the known part c of the potentials (v = c + P y): non-zero only on members of constraint groups.
 * Extra right-hand sides k = 0..n_extra-1 (regulator gain columns gamma_k, solver.py:537-538) as sparse columns in
 * ORIGINAL row indices, CSR-like: entries extra_ptr[k]..extra_ptr[k+1] of extra_row / extra_val; b_k = P^T gamma_k.
 * The device forms b = -P^T (r - L c), solves A y = b and A z_k = b_k (zero right-hand sides are skipped; the relative
 * tolerance is tightened so that ||b - A y|| <= abs_residual_target where opts.rtol ||b|| would be looser: the reference's
 * absolute residual bar, tests/test_solver.py:2083-2089; 0 = off), expands v = c + P y and Z_k = P z_k, and returns
 * probe_out[(1 + n_extra)][n_probe]: rho = r - L v, then L Z_k, at the probed unknowns (the members of the constraint
 * groups) -- what the host needs to peel the multiplier currents from.  v stays on the device.
 * Returns PADNE_E_NOTCONVERGED like padne_solve_spd (the iterate is kept, padne_kkt_finish may follow). */
int padne_kkt_solve(padne_ctx *ctx, padne_kkt *plan, const double *r_host, int64_t n_known, const int64_t *known_idx,
                    const double *known_val, int32_t n_extra, const int64_t *extra_ptr, const int64_t *extra_row,
                    const double *extra_val, int64_t n_probe, const int64_t *probe_idx, double *probe_out,
                    const padne_solve_opts *opts, double abs_residual_target, padne_solve_info *info);
/* Stage 2: v += sum_k extra_coeff[k] Z_k (regulator currents), v[mult_idx] = mult_val (the recovered multiplier
 * currents: sources, regulators, ground row), then residual_norm = ||L v - r||_2 on the ORIGINAL system
 * (solver.py:775) while v[N] travels to v_host -- its only crossing of PCIe. */
int padne_kkt_finish(padne_ctx *ctx, padne_kkt *plan, int32_t n_extra, const double *extra_coeff, int64_t n_mult,
                     const int64_t *mult_idx, const double *mult_val, double *v_host, double *residual_norm_out);

/* Row-partitioned runs, optional: attach the rank's owned x owned diagonal block; with precond = 1 the
 * multigrid hierarchy is then built on that block only (block-Jacobi with multigrid blocks, no
 * communication inside the cycle, 3-6x more CG iterations than the default hierarchy over all ranks).
 * Borrowed handle; null detaches. */
int padne_csr_set_preconditioner_block(padne_csr *a, padne_csr *block);

/* z = M^-1 r: one V-cycle of the multigrid preconditioner (built on first use and cached on `a`);
 * host vectors.  Exposed for tests: M must be symmetric positive definite for PCG to apply. */
int padne_amg_apply(padne_ctx *ctx, padne_csr *a, const double *r_host, double *z_host);

/* introspection: borrowed handle of a hierarchy operator (which: 0 = A_l, 1 = P_l, 2 = R_l = P_l^T);
 * it stays valid as long as `a` does and must NOT be destroyed */
int padne_amg_level(padne_ctx *ctx, padne_csr *a, int level, int which, const padne_csr **out);

/* ---- connection snapping ------------------------------------------------------------------- */
/* index of the nearest point of xy[n_points][2] for every query point (squared Euclidean distance in binary64,
 * ties to the smallest index): what NodeIndexer.create asks of its per-layer KD-trees
 * (_construct_kdtrees solver.py:356-396, query solver.py:425).  Brute force on the device: at a million vertices
 * the KD-tree build is two thirds of the host time of solve(). */
int padne_nearest_vertex(padne_ctx *ctx, int64_t n_points, const double *xy_host, int64_t n_query,
                         const double *query_host, int64_t *index_out_host);
/* The same, and for every query the number of points at EXACTLY the minimum distance (tie_count_out_host, >= 1).  The
 * reference's KD-tree (solver.py:389-392, query :425) returns whichever of several equidistant vertices its traversal meets
 * first; a caller that wants the reference's choice re-resolves the queries with a count above 1 with that tree (they are
 * rare: a connection exactly between vertices of an unjittered grid) -- NodeIndexer.create does. */
int padne_nearest_vertex_ties(padne_ctx *ctx, int64_t n_points, const double *xy_host, int64_t n_query,
                              const double *query_host, int64_t *index_out_host, int32_t *tie_count_out_host);

/* ---- post-processing ----------------------------------------------------------------------- */
/* per-face power density  p = sigma*|grad V|^2 with the reference's barycentric difference
 * quotient (compute_triangle_gradient solver.py:689-725, compute_power_density :728-745).
 * Also performs the scatter of produce_layer_solutions (solver.py:596-598): `potential` is the
 * global solution vector, vertex v of mesh m reads potential[mesh_vertex_offset[m] + v]. */
int padne_power_density(padne_ctx *ctx, int64_t n_vert, const double *xy_host,
                        int64_t n_tri, const int32_t *tri_host,
                        int64_t n_mesh, const int64_t *mesh_vertex_offset,
                        const int64_t *mesh_tri_offset, const double *conductance,
                        const double *potential_host, double *power_out_host);

/* The same power densities for the mesh an assembled system was built from: padne_assemble_system leaves the mesh
 * on the device with the matrix, so only the potentials travel (n_vert doubles up, n_tri doubles down).
 * potential_host: the first n_vert entries of the solution in the global vertex numbering. */
int padne_csr_power_density(padne_ctx *ctx, const padne_csr *m, const double *potential_host, double *power_out_host);

/* per-face gradient of the linear interpolant of `potential` (compute_triangle_gradient,
 * solver.py:689-725), faces visited as (v3, v1, v2) like Face.vertices (mesh.py:320-325) */
int padne_face_gradient(padne_ctx *ctx, int64_t n_vert, const double *xy_host,
                        int64_t n_tri, const int32_t *tri_host,
                        int64_t n_mesh, const int64_t *mesh_vertex_offset,
                        const int64_t *mesh_tri_offset, const double *potential_host,
                        double *gx_out_host, double *gy_out_host);

/* ---- introspection for benchmarks ---------------------------------------------------------- */
/* algorithmic bytes of one CSR SpMV: 12*nnz + 20*n_rows + 4   */
int64_t padne_spmv_algorithmic_bytes(const padne_csr *m);
/* average device time (seconds) of `repeat` back-to-back SpMV launches measured with HIP
 * events on the context stream, after `warmup` untimed launches */
int padne_spmv_time(padne_ctx *ctx, const padne_csr *m, const void *x_dev, void *y_dev,
                    int warmup, int repeat, double *seconds_per_launch);

/* the same two for the 8-vector product: 12*nnz + 4*n_rows + 16*n_rows*8 + 4 bytes */
int64_t padne_spmm8_algorithmic_bytes(const padne_csr *m);
int padne_spmm8_time(padne_ctx *ctx, const padne_csr *m, const void *x_dev, void *y_dev,
                     int warmup, int repeat, double *seconds_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* PADNE_HIP_H */
