/*
 * hfmi.h -- C ABI of libhfmi.so, the MI355X (gfx950) device layer under the
 * hippyflow model-based projectors' randomized double-pass eigensolve.
 *
 * The reference (hippyflow, /root/reference) has no FFI of its own: its boundary
 * is a set of duck-typed Python protocols over the
 * third-party hippylib package, whose only native pieces are a C++ MultiVector
 * and a C++ mt19937 "parRandom" compiled by dolfin.  Each entry point below
 * names the reference call site / hippylib symbol it stands in for.
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary; every function returns 0 on
 *     success or a negative hfmi_status; hfmi_last_error() gives the text.
 *   - all arithmetic is IEEE fp64.
 *   - one hfmi_ctx per GPU; a context (and the objects made from it) is not
 *     thread-safe; independent contexts may be used from different threads or
 *     processes.  All work is enqueued on the context's HIP stream.
 *   - a BLOCK is hippylib's MultiVector: nvec vectors of length N, each vector
 *     contiguous in HBM (column-major N x nvec with leading dimension ld,
 *     ld % 32 == 0, 256-byte aligned columns, rows N..ld-1 kept at zero).
 *     Snapshot matrices (n snapshots of length N; PODProjector.py:340-357) and
 *     stacked Jacobians ((ndata*q) rows of length N; operatorWrappers.py:62-64)
 *     are blocks too: one vector per snapshot / per Jacobian row.
 *   - host arrays are caller-owned, dense, C-ordered fp64.
 */
#ifndef HFMI_H
#define HFMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HFMI_VERSION 100

typedef enum {
  HFMI_OK = 0,
  HFMI_ERR_INVALID = -1,      /* bad argument / shape mismatch (the reference asserts) */
  HFMI_ERR_HIP = -2,          /* a HIP runtime call failed */
  HFMI_ERR_NO_DEVICE = -3,    /* no usable gfx950 device */
  HFMI_ERR_NUMERIC = -4,      /* breakdown (e.g. Gram matrix not SPD after shifting) */
  HFMI_ERR_CALLBACK = -5,     /* a host callback operator returned non-zero */
  HFMI_ERR_NOT_CONVERGED = -6,/* iterative kernel hit its iteration cap */
  HFMI_ERR_COMM = -7          /* communicator failure (RCCL error, peer rank missing, time-out) */
} hfmi_status;

/* host <-> block layouts */
#define HFMI_LAYOUT_VECTORS 0 /* host (nvec, N): one vector per row  -- u_data / q_data / J (PODProjector.py:224-225) */
#define HFMI_LAYOUT_DENSE 1   /* host (N, nvec): mv_to_dense layout  -- utilities/mv_utilities.py:31-41 */

typedef struct hfmi_ctx hfmi_ctx;
typedef struct hfmi_block hfmi_block;
typedef struct hfmi_csr hfmi_csr;
typedef struct hfmi_op hfmi_op;
typedef struct hfmi_comm hfmi_comm;

/* ---------------------------------------------------------------- context */
const char* hfmi_last_error(void);
int hfmi_version(void);
const char* hfmi_build_tag(void);   /* identity of the kernel sources (hash); keys the PMC records bench.py may use */
int hfmi_device_count(int* count);
int hfmi_ctx_create(int device, hfmi_ctx** out);
int hfmi_ctx_destroy(hfmi_ctx* ctx);
/* adopt an external HIP stream (e.g. torch.cuda.current_stream().cuda_stream); NULL = own stream */
int hfmi_ctx_set_stream(hfmi_ctx* ctx, void* hip_stream);
int hfmi_ctx_get_stream(hfmi_ctx* ctx, void** hip_stream);
int hfmi_ctx_synchronize(hfmi_ctx* ctx);
int hfmi_ctx_device_info(hfmi_ctx* ctx, char* name, int name_len, int* compute_units, int64_t* hbm_bytes);
int hfmi_ctx_pci_bus_id(hfmi_ctx* ctx, char* buf, int len);   /* "0000:5d:00.0": keys the sysfs clock / power files bench.py reads */
/* HIP-event timer on the context's stream (bench.py measures kernels with it) */
int hfmi_timer_start(hfmi_ctx* ctx);
int hfmi_timer_stop(hfmi_ctx* ctx, double* milliseconds); /* synchronises */

/* ---------------------------------------------------------------- blocks
 * hippylib MultiVector(vector, nvec) and its copy constructor. */
int hfmi_block_create(hfmi_ctx* ctx, int64_t N, int nvec, hfmi_block** out); /* zero-filled */
/* wrap device memory owned by the caller (e.g. a torch tensor): ld % 32 == 0, ld >= N,
 * dptr 128-byte aligned; rows N..ld-1 are zeroed by the call. */
int hfmi_block_wrap(hfmi_ctx* ctx, double* dptr, int64_t N, int nvec, int64_t ld, hfmi_block** out);
/* view of vectors [first, first+count) of a block (MultiVector.__getitem__) */
int hfmi_block_view(hfmi_block* parent, int first, int count, hfmi_block** out);
int hfmi_block_destroy(hfmi_block* b);
int hfmi_block_info(const hfmi_block* b, int64_t* N, int* nvec, int64_t* ld, double** dptr);
int hfmi_block_upload(hfmi_block* b, const double* host, int layout);
int hfmi_block_download(const hfmi_block* b, double* host, int layout);
/* streaming ingest: the reference fills its snapshot / Jacobian blocks sample by sample from host PDE solves
 * (PODProjector.py:343-357; activeSubspaceProjector.py:178-221).  hfmi_block_upload_async copies from PINNED host memory
 * (hfmi_host_alloc_pinned) on the context's ingest stream and returns at once; *ticket names the upload.
 * hfmi_ingest_wait(ticket): the pinned buffer of that upload may be overwritten (host wait).  hfmi_ingest_fence: work
 * enqueued afterwards on the compute stream sees every upload made so far (device-side wait, the host is not blocked).
 * b is normally a view (hfmi_block_view) of the vectors of one sample. */
int hfmi_host_alloc_pinned(size_t bytes, void** out);
int hfmi_host_free_pinned(void* p);
int hfmi_block_upload_async(hfmi_block* b, const double* host_pinned, int layout, int64_t* ticket);
int hfmi_ingest_wait(hfmi_ctx* ctx, int64_t ticket);
int hfmi_ingest_fence(hfmi_ctx* ctx);
int hfmi_block_zero(hfmi_block* b);                                 /* MultiVector.zero */
int hfmi_block_copy(hfmi_block* dst, const hfmi_block* src);        /* copy constructor */
int hfmi_block_scale(hfmi_block* b, double alpha);                  /* vector *= alpha */
int hfmi_block_axpy(hfmi_block* y, double alpha, const hfmi_block* x); /* vector.axpy, all vectors */
int hfmi_block_norms(const hfmi_block* b, double* host_norms);      /* MultiVector.norm("l2") */

/* a1: the probe draw -- hp.parRandom.normal(sigma, Omega)
 * (activeSubspaceProjector.py:433-443,536-551; PODProjector.py:365-374;
 * KLEProjector.py:151-160).  Counter-based Philox4x32-10 + Box-Muller: every GPU
 * regenerates the same Omega from (seed, stream), replacing collective.bcast.
 * Rows 4g .. 4g+3 of vector j come from the counter (g, j, stream) under the key
 * seed: four 32-bit uniforms -> two radius/angle pairs (oracle/philox.py). */
int hfmi_randn_fill(hfmi_block* b, uint64_t seed, uint32_t stream, double sigma);
/* the raw 32-bit stream behind it (bit-exact parity test): out[nvec][ceil(N/4)][4] */
int hfmi_philox_raw(hfmi_block* shape_of, uint64_t seed, uint32_t stream, uint32_t* host_out);

/* synthetic config-2 input: C (N x N block) = Matern-3/2 covariance
 * sigma^2 (1 + a) exp(-a), a = sqrt(3) d_ij / ell, over the first N nodes of an nx x ny grid on the unit square */
int hfmi_block_fill_matern32(hfmi_block* C, int nx, int ny, double sigma, double ell);

/* MultiVector.dot_mv / dot_v: out[i*nvecB + j] = <A_i, B_j>  (row-major nvecA x nvecB) */
int hfmi_block_dot(const hfmi_block* A, const hfmi_block* B, double* host_out);
/* MvDSmatMult / MultiVector.reduce: Y = alpha * A * S + beta * Y, S host (nvecA x nvecY) row-major */
int hfmi_block_gemm_small(const hfmi_block* A, const double* host_S, double alpha, double beta, hfmi_block* Y);

/* ---------------------------------------------------------------- sparse
 * CSR matrix (prior.M, prior.R; PODProjectorFromData.M_csr, PODProjector.py:695-697). */
int hfmi_csr_create(hfmi_ctx* ctx, int64_t nrows, int64_t ncols, int64_t nnz, const int64_t* indptr,
                    const int32_t* indices, const double* data, hfmi_csr** out);
int hfmi_csr_destroy(hfmi_csr* m);

/* ---------------------------------------------------------------- operators
 * The reference's linear-operator protocol (mult / matMvMult / init_vector;
 * SURVEY.md section 8b) as a tagged object.  apply = hp.MatMvMult(A, W, Y). */

/* a2: hp.LowRankOperator(ones/n, snapshots)  -> Y = scale * X (X^T W)
 *     (PODProjector.py:359-361).  X: block, one vector per snapshot. */
int hfmi_op_snapshot_gram(hfmi_ctx* ctx, const hfmi_block* X, double scale, hfmi_op** out);
/*     general diagonal: hp.LowRankOperator(d, U) -> Y = U diag(d) (U^T W)  (prior.Hlr as B / B^-1,
 *     activeSubspaceProjector.py:455-459; priorPreconditionedProjector.py:48-55).  host_d: one weight per vector of U. */
int hfmi_op_low_rank(hfmi_ctx* ctx, const hfmi_block* U, const double* host_d, hfmi_op** out);
/* a3: sample-averaged Jacobian Gram  Y = scale * sum_i J_i^T Gamma^{-1} J_i W
 *     (MeanJTJfromDataOperator.mult, operatorWrappers.py:95-114; JTJ summed by
 *     SummedListOperator / SeriallySampledJacobianOperator,
 *     activeSubspaceProjector.py:82-95,163-248).  J: block of ndata*q vectors
 *     (row o of sample i is vector i*q+o); gamma_inv host q x q or NULL. */
int hfmi_op_jtj(hfmi_ctx* ctx, const hfmi_block* J, int ndata, int q, const double* host_gamma_inv,
                double scale, hfmi_op** out);
/*     output-space counterpart  Y = scale * sum_i J_i J_i^T W  (JJT, jacobian.py:169-193;
 *     activeSubspaceProjector.py:625-673); acts on blocks of length q. */
int hfmi_op_jjt(hfmi_ctx* ctx, const hfmi_block* J, int ndata, int q, double scale, hfmi_op** out);
/* a4: explicit dense symmetric operator (config 2 covariance; npToDolfinOperator,
 *     operatorWrappers.py:19-52).  C: block of N vectors of length N (symmetric). */
int hfmi_op_dense_sym(hfmi_ctx* ctx, const hfmi_block* C, hfmi_op** out);
/* a4/a9: sparse operator  Y = M W  (prior.M.mult, prior.R.mult; hp.MatMvMult(B, decoder, encoder)) */
int hfmi_op_csr(hfmi_ctx* ctx, const hfmi_csr* M, hfmi_op** out);
/*     solver object for an SPD CSR matrix: Y = M^{-1} W to a relative residual rel_tol per vector
 *     (prior.Msolver behind hp.Solver2Operator, KLEProjector.py:163-164).  Jacobi-preconditioned
 *     Chebyshev iteration on a row-major copy of the block (one kernel per step, no inner products;
 *     the spectrum of D^-1 M is bracketed once per matrix: Gershgorin + the Lanczos matrix of one
 *     scalar CG run); Jacobi-preconditioned block CG when that bracket is too wide or does not
 *     deliver the tolerance. */
int hfmi_op_csr_pcg(hfmi_ctx* ctx, const hfmi_csr* M, double rel_tol, int max_iter, hfmi_op** out);
/*     what the last solve of such an operator did: steps taken, method (0 block CG, 1 Chebyshev),
 *     and the bracket of the spectrum of D^-1 M in use (0, 0: none) */
int hfmi_op_solver_info(const hfmi_op* op, int* iterations, int* method, double* lmin, double* lmax);
/*     Y = c (b (a W))  (MassPreconditionedCovarianceOperator M C M, KLEProjector.py:47-69) */
int hfmi_op_compose3(hfmi_ctx* ctx, hfmi_op* a, hfmi_op* b, hfmi_op* c, hfmi_op** out);
/*     host black box (FEniCS PDE solves, sparse LU ...): W and Y in HFMI_LAYOUT_VECTORS
 *     (k, N) host arrays; return non-zero to abort.  This is how any object with the
 *     reference's mult/matMvMult protocol plugs into the device solve. */
typedef int (*hfmi_host_apply_fn)(void* user, const double* W_host, double* Y_host, int64_t N, int k);
int hfmi_op_host_callback(hfmi_ctx* ctx, hfmi_host_apply_fn fn, void* user, int64_t N, hfmi_op** out);
/*     For a callback that treats the vectors independently (a sparse-LU / Krylov solve per vector: prior.Rsolver,
 *     activeSubspaceProjector.py:447-450; prior.Msolver): invoke it on slabs of `vectors` vectors.  The slabs go
 *     through pinned double buffers and the device->host copy of slab i+1 and the host->device copy of slab i-1
 *     overlap the host work on slab i.  0 (default) = one call with the whole block. */
int hfmi_op_host_set_chunk(hfmi_op* op, int vectors);
/*     average of a device operator over the ranks of a communicator is done by the
 *     caller between applies (CollectiveOperator, collectiveOperator.py:31-38): a
 *     post-apply hook called with the result block, e.g. an RCCL all-reduce. */
typedef int (*hfmi_post_apply_fn)(void* user, hfmi_block* Y);
int hfmi_op_set_post_apply(hfmi_op* op, hfmi_post_apply_fn fn, void* user);
/*     the same average done natively: the result block of every apply (and the k x k Rayleigh quotient of the
 *     Gram-form solves) is all-reduced over `comm` on the context's stream, no host code inside the solve.
 *     reduce_op HFMI_REDUCE_SUM | HFMI_REDUCE_AVG; comm NULL detaches. */
int hfmi_op_set_collective(hfmi_op* op, hfmi_comm* comm, int reduce_op);
int hfmi_op_apply(hfmi_op* op, const hfmi_block* W, hfmi_block* Y, int accumulate);
int hfmi_op_destroy(hfmi_op* op);

/* ---------------------------------------------------------------- communicator
 * The reference's sample-parallel collective (hippyflow/collectives/collective.py): one process per GPU.
 *   _allReduce_array, collective.py:61-71          -> hfmi_allreduce_host
 *   allReduce of a MultiVector, :98-111 (k Allreduce calls of length N, through host copies)
 *                                                   -> hfmi_allreduce: ONE RCCL all-reduce of the N x k block in
 *                                                      HBM on the context's stream, 1/P of 'avg' fused (ncclAvg)
 *   bcast of a MultiVector, :144-152 (k Bcast calls) -> hfmi_bcast
 *   comm.Get_size / Get_rank, :52-58                -> hfmi_comm_info
 * Bootstrap: rank 0 calls hfmi_comm_unique_id and ships the HFMI_UNIQUE_ID_BYTES bytes to the other ranks (any
 * channel: mpi4py bcast, a file -- hfmi_comm_init_from_file does the file exchange itself); every rank then calls
 * hfmi_comm_init_rank with its own context.  Transports (hfmi_comm_info): 1 = RCCL over xGMI (each rank its own
 * GPU), 2 = direct peer access through HIP IPC staging buffers (ranks sharing a GPU, or HFMI_COMM_TRANSPORT=p2p),
 * 0 = host-only (ctx NULL on every rank: host payloads and barriers only).  Ranks of one communicator live on one
 * node unless HFMI_COMM_TRANSPORT=rccl.  A peer that never arrives fails the call after HFMI_COMM_TIMEOUT_S
 * (300 s) with HFMI_ERR_COMM instead of hanging.  The transport is agreed among the ranks: if librccl does not load,
 * ncclCommInitRank fails or the first all-reduce does not give the right sum on ANY rank, ALL ranks use the p2p
 * transport (which works across GPUs through HIP IPC and is stream-ordered: counters in the node segment, written and
 * polled from the GPUs; HFMI_P2P_SYNC=host|stream overrides the choice).  The id file of hfmi_comm_init_from_file is
 * created 0600 with O_EXCL, read only if it is this user's, and removed before any rank returns. */
#define HFMI_UNIQUE_ID_BYTES 256
#define HFMI_REDUCE_SUM 0
#define HFMI_REDUCE_AVG 1
#define HFMI_REDUCE_MAX 2
int hfmi_comm_unique_id(void* id_out);
int hfmi_comm_init_rank(hfmi_ctx* ctx_or_null, const void* id, int nranks, int rank, hfmi_comm** out);
int hfmi_comm_init_from_file(hfmi_ctx* ctx_or_null, const char* path, int nranks, int rank, hfmi_comm** out);
int hfmi_comm_info(const hfmi_comm* comm, int* nranks, int* rank, int* transport);
/* one line of JSON: the transport, WHY it was chosen (e.g. "fell back from rccl: the first ncclAllReduce failed on rank 3;
 * all ranks agreed on p2p"), the RCCL library in use, every rank's PCI bus id, how the p2p path synchronises */
int hfmi_comm_describe(const hfmi_comm* comm, char* buf, int len);
/* the transport decision as a pure function of the table the ranks publish (test hook for the CPU suite): has_device[p],
 * rccl_ok[p], device_ids[p] for p < nranks; *transport = 0 host / 1 rccl / 2 p2p, or -1 for an inconsistent table */
int hfmi_comm_decide_transport(int nranks, const int* has_device, const int* rccl_ok, const char* const* device_ids,
                               int force_p2p, int* transport, char* reason, int reason_len);
int hfmi_comm_barrier(hfmi_comm* comm);               /* drains the context's stream, then meets the other ranks */
int hfmi_allreduce(hfmi_comm* comm, hfmi_block* Y, int reduce_op);           /* in place, stream-ordered */
int hfmi_bcast(hfmi_comm* comm, hfmi_block* Y, int root);
int hfmi_allreduce_host(hfmi_comm* comm, double* v, int64_t count, int reduce_op);   /* in place */
int hfmi_bcast_host(hfmi_comm* comm, void* v, int64_t nbytes, int root);
int hfmi_comm_destroy(hfmi_comm* comm);

/* ---------------------------------------------------------------- QR (a7)
 * MultiVector.orthogonalize() / Borthogonalize(B): thin QR with Q^T B Q = I,
 * R upper triangular with positive diagonal (unique, so Q equals the
 * reference's MGS Q to round-off).  B, BQ, host_R may be NULL.
 * method: HFMI_QR_CHOL = (shifted) Cholesky-QR, repeated until orthonormal;
 *         HFMI_QR_MGS  = column-by-column Gram-Schmidt with the reference's
 *         Rutishauser re-orthogonalisation test (dependent columns zeroed). */
#define HFMI_QR_CHOL 0
#define HFMI_QR_MGS 1
#define HFMI_QR_AUTO 2 /* CHOL, falling back to MGS on breakdown */
int hfmi_borth_qr(hfmi_block* Q, hfmi_op* B, hfmi_block* BQ, double* host_R, int method, int* passes);

/* ---------------------------------------------------------------- Rayleigh-Ritz (a8)
 * np.linalg.eigh(T) + descending sort: symmetric k x k (host, row-major; the
 * symmetric part is used), eigenvalues descending, eigenvectors in the columns of V
 * (row-major k x k).  sort_by_abs is a flag word: bit 0 = order by |d|; bit 1 = HFMI_EIG_JACOBI.
 * k <= 256, default: Householder tridiagonalisation + divide and conquer on one compute unit
 * (hfmi_eig_dc.hip) -- the algorithm family of the LAPACK routine behind np.linalg.eigh, absolute
 * accuracy eps ||T||.  HFMI_EIG_JACOBI: one-workgroup parallel cyclic Jacobi in LDS (slower; small
 * eigenvalues of graded positive definite matrices to high RELATIVE accuracy).
 * 256 < k <= 16384 (the n x n Gram problem of the deterministic POD, la.eigh at PODProjector.py:821, any number of
 * snapshots): the same algorithm family over the whole GPU (hfmi_eig_blocked.hip) -- panel Householder
 * tridiagonalisation with the trailing update on the fp64 MFMA, divide and conquer with the leaves on one compute unit
 * each and the upper merges on all of them, block-reflector back-transformation.  Non-finite entries: HFMI_ERR_NUMERIC
 * (np.linalg.eigh raises LinAlgError).  HFMI_EIG_LARGE=jacobi in the environment selects the two-sided Jacobi of
 * rounds 2-4 (hfmi_eig_large.hip; up to 4096). */
#define HFMI_EIG_SORT_ABS 1
#define HFMI_EIG_JACOBI 2
int hfmi_sym_eig_small(hfmi_ctx* ctx, const double* host_T, int k, int sort_by_abs, double* host_d,
                       double* host_V);
/* The same with only the nvec leading eigenvectors (in output order) returned; host_V is k x nvec row-major.  What the
 * deterministic POD uses of la.eigh(G): U[:, :u_rank] (PODProjector.py:821-826).  Beyond 256 the back-transformation and
 * the read-back run over nvec columns instead of k. */
int hfmi_sym_eig_leading(hfmi_ctx* ctx, const double* host_T, int k, int sort_by_abs, int nvec, double* host_d,
                         double* host_V);
/* la.eigh(X^T (M X)) of the deterministic POD in one call (PODProjector.py:818-826: UtMU = u_data @ M @ u_data.T, eigh,
 * U[:, :u_rank]): the n x n Gram matrix of two blocks of n vectors is formed on the device and handed to the
 * eigensolver there; host_d receives the n eigenvalues, host_V the nvec leading eigenvectors (n x nvec row-major). */
int hfmi_block_gram_eig(const hfmi_block* A, const hfmi_block* B, int sort_by_abs, int nvec, double* host_d,
                        double* host_V);

/* np.linalg.svd(R) of the small factor inside hp.accuracyEnhancedSVD (activeSubspaceProjector.py:813-834,1026):
 * R (host, k x k row-major) = U diag(sigma) V^T, sigma descending; U, V row-major k x k (columns = vectors).
 * One-workgroup one-sided Jacobi in LDS (full relative accuracy of small singular values). */
int hfmi_svd_small(hfmi_ctx* ctx, const double* host_R, int k, double* host_sigma, double* host_U, double* host_V);

/* ---------------------------------------------------------------- full solves (a5, a6)
 * hp.doublePass(A, Omega, r, s) / hp.doublePassG(A, B, Binv, Omega, r, s):
 * Omega has k >= r vectors and is not modified; on return host_d[r] holds the
 * eigenvalues (descending) and U (r vectors) the (B-)orthonormal eigenvectors.
 * Everything stays on the device between the first apply and the final U.
 * flags: bit 0 = sort by |d|;  bit 1 = use HFMI_QR_MGS;  bit 3 = Jacobi instead of divide and conquer for the
 * k x k Rayleigh-Ritz problem;  bit 2 = form T = (A Q)^T Q literally (by default, for
 * operators of Gram form A = scale X^T Gamma X the same matrix is formed as scale (X Q)^T Gamma (X Q), which skips
 * the second N x k block product and shrinks the rank average of that pass to k x k). */
int hfmi_double_pass(hfmi_op* A, const hfmi_block* Omega, int r, int s, int flags, double* host_d,
                     hfmi_block* U);
int hfmi_double_pass_g(hfmi_op* A, hfmi_op* B, hfmi_op* Binv, const hfmi_block* Omega, int r, int s,
                       int flags, double* host_d, hfmi_block* U);

/* ---------------------------------------------------------------- instrumentation
 * Kernel-level entry points used by bench.py / the parity tests:
 *   C (nvecA x nvecB, device partial-summed, returned on host) = A^T B with an explicit split count
 *   (0 = library default) and the average kernel time of `reps` back-to-back launches. */
int hfmi_bench_tsgemm_tn(const hfmi_block* A, const hfmi_block* B, int nsplit, int reps, double* host_C,
                         double* avg_ms);
int hfmi_bench_tsgemm_nn(const hfmi_block* A, const double* host_S, hfmi_block* Y, int reps, double* avg_ms);
/* C (M x N) = op(A) op(B), column-major host operands with their natural leading dimensions (A: ta ? K x M : M x K; B: tb ? N x K : K x N),
 * on the general fp64 MFMA product of the eigensolver: the N x N x N congruence products of the deterministic POD's N-dimensional route
 * (la.eigh of PODProjector.py:812-833 reformulated in the state dimension when the snapshots outnumber it: hippyflow_amd/projectors.py) */
int hfmi_dense_matmul(hfmi_ctx* ctx, int M, int N, int K, int ta, int tb, const double* host_A, const double* host_B, double* host_C);
/* the general fp64 MFMA product inside the whole-GPU eigensolver (trailing rank-2k updates, Q S of the merges, block reflectors of
 * the back-transformation; la.eigh(G), PODProjector.py:812-833): C (M x N) = op(A) op(B), column-major host operands with their
 * natural leading dimensions, average kernel time of `reps` launches; host_C may be null */
int hfmi_bench_dgemm(hfmi_ctx* ctx, int M, int N, int K, int ta, int tb, int reps, const double* host_A, const double* host_B,
                     double* host_C, double* avg_ms);
/* fp64 MFMA / fp64 FMA / HBM-copy micro-benchmarks (peak denominators measured in the same job) */
int hfmi_bench_peaks(hfmi_ctx* ctx, double* mfma_f64_tflops, double* fma_f64_tflops, double* hbm_copy_gbs);
/* the same MFMA loop with a copy kernel streaming HBM beside it on a second stream: the ceiling of the power-limited regime the
 * big contractions run in (bench.py: roofline.frac_of_in_job_loaded_peak) */
int hfmi_bench_loaded_peak(hfmi_ctx* ctx, double* mfma_f64_tflops, double* hbm_copy_gbs);
/* the MFMA loop on full-mantissa Gaussian operands rotated through the registers every iteration -- alone, and beside the streaming
 * copy: the ceiling the contractions can reach on the solve's data under the power limit (SURVEY section 8d "fp64 MFMA
 * micro-benchmark run in the same job"; bench.py: roofline.frac_of_in_job_random_operand_peak[_while_streaming]) */
/* read-only 16-byte stream over 2 GiB: the HBM rate a contraction that only reads its big operand can reach (the copy of
 * hfmi_bench_peaks also writes); bench.py: roofline.frac_of_in_job_read_peak for the HBM-bound kernel-point shapes */
int hfmi_bench_hbm_read(hfmi_ctx* ctx, double* hbm_read_gbs);
int hfmi_bench_random_peaks(hfmi_ctx* ctx, double* mfma_f64_tflops, double* mfma_f64_tflops_while_streaming, double* hbm_copy_gbs);
/* per-launch HIP-event timing over a region of ordinary calls (bench.py's roofline numbers come from the
 * timed region itself): between begin and end every tsgemm_tn / tsgemm_nn launch is bracketed by events on
 * the context's stream.  end() synchronises and returns one record per distinct (kernel, shape):
 * kind[g] 0 = k_tsgemm_tn (or k_tsgemm_ss for skinny x skinny shapes), 1 = k_tsgemm_nn; shape[3*g..] = (short-side rows m, columns k, long axis N);
 * total milliseconds, launches, and the ALGORITHMIC flops / bytes of one launch. */
/* kernel tuning knobs for in-process A/B measurements (defaults are the measured winners; scripts/gemm_ab.py,
 * scripts/ss_ab.py, scripts/nn_tt_probe.py, scripts/nn_waves_ab.py): ("waves", 8|4|44) tsgemm_tn workgroup shape (44 = two 4-wave workgroups
 * per CU); ("rem4", 1|0) last column tile of <= 12 columns as 4-column groups on the 4x4x4 MFMA | as a full 16-column tile;
 * ("probe", 0..3) timing-only diagnostic of tsgemm_tn (bit 0: the streamed operand re-reads one address, bit 1: no staging /
 * barriers; results are garbage -- scripts/tn_probe.py); ("nn_waves", 0|4|8) and ("nn_tt", 0..3)
 * tsgemm_nn workgroup / wave-tile height (0 = automatic); ("nn_hybrid", 0|1) split only the tail row tiles; ("nn_res", 1|0) small matrix resident in LDS with persistent
 * workgroups when it fits (short reductions: Q R^-1, U = Q V); ("ss", 0|1) route skinny x skinny contractions to
 * tsgemm_ss; ("ss_percu", 1..4) resident tsgemm_ss workgroups per CU assumed when the grid is sized; ("eig", 0|1) Rayleigh-Ritz
 * eigensolver of every call: divide and conquer | Jacobi; ("chol", 0|1) Cholesky + inverse of the QR passes: blocked MFMA kernel |
 * column-at-a-time kernels; ("nn_res_tt", 0|1|2) tile height of the LDS-resident nn product: by round count | table | one less;
 * ("tn_hybrid", 0|1) tsgemm_tn with more row blocks than CUs: uniform split | whole rounds coarsely split + finely split tail;
 * ("prof_level", 1|2) what a profiling region records: 2 = every contraction and every phase (default), 1 = contractions of at
 * least 2 Gflop only (each record is a pair of stream events, 2-4 us of idle GPU between dependent kernels: scripts/prof_level_ab.py);
 * ("comm_panels", 0..8) row panels of an operator application whose
 * rank reduction overlaps the rest of the product (0 / 1 = one all-reduce after the product; default 4). */
int hfmi_tuning_set(const char* key, int value);
/* phases of hfmi_double_pass[_g], accumulated between hfmi_profile_begin and hfmi_profile_end (milliseconds, summed
 * over the solves in the region; device phases by HIP events on the context's stream, the HOST_* legs by the host's
 * wall clock -- they are part of the phase that called the host operator, normally BINV) */
#define HFMI_PHASE_APPLY 0        /* power-iteration applies of A (incl. their all-reduce) */
#define HFMI_PHASE_BINV 1         /* applies of B^-1 */
#define HFMI_PHASE_QR 2           /* (B-)orthogonalisation */
#define HFMI_PHASE_RAYLEIGH 3     /* second pass: T = Q^T A Q */
#define HFMI_PHASE_EIG 4          /* small eigensolve */
#define HFMI_PHASE_BACK 5         /* U = Q V */
#define HFMI_PHASE_ALLREDUCE 6    /* rank reductions enqueued by the solve (inside APPLY / RAYLEIGH) */
#define HFMI_PHASE_HOST_D2H 7     /* host callback: waiting for device -> pinned host copies */
#define HFMI_PHASE_HOST_FN 8      /* host callback: inside the host function */
#define HFMI_PHASE_HOST_H2D 9     /* host callback: issuing / draining pinned host -> device copies */
#define HFMI_PHASE_ALLREDUCE_AUX 10 /* rank reductions of row panels on the auxiliary stream, overlapped with the product that
                                     * makes the next panel; HFMI_PHASE_ALLREDUCE then holds only what the main stream waited */
#define HFMI_PHASE_COUNT 11
int hfmi_profile_phases(hfmi_ctx* ctx, double* ms_out /* HFMI_PHASE_COUNT */);   /* after hfmi_profile_end */
int hfmi_profile_begin(hfmi_ctx* ctx);
int hfmi_profile_end(hfmi_ctx* ctx, int max_groups, int* ngroups, int* kind, int64_t* shape, double* ms,
                     int64_t* launches, double* flops_per_launch, double* bytes_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* HFMI_H */
