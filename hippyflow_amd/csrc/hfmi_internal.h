// Internal declarations shared by the libhfmi translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "hfmi.h"

// ------------------------------------------------------------------ errors
void hfmi_set_error(const char* fmt, ...);
#define HFMI_FAIL(code, ...)      \
  do {                            \
    hfmi_set_error(__VA_ARGS__);  \
    return (code);                \
  } while (0)
#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      hfmi_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return HFMI_ERR_HIP;                                                                     \
    }                                                                                          \
  } while (0)
#define HFMI_TRY(expr)        \
  do {                        \
    int _s = (expr);          \
    if (_s != HFMI_OK) return _s; \
  } while (0)

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
// an A/B or diagnostic switch from the environment: on when set to anything but "" or "0"
static inline bool env_flag(const char* name) {
  const char* e = getenv(name);
  return e && e[0] && !(e[0] == '0' && e[1] == 0);
}

// ------------------------------------------------------------------ objects
enum { WS_PART = 0, WS_G, WS_STAGE, WS_MISC, WS_MGS, WS_COMM, WS_INGEST, WS_NSLOTS };
#define HFMI_INGEST_RING 8

struct hfmi_block {
  hfmi_ctx* ctx;
  double* p;
  int64_t N;
  int nvec;
  int64_t ld;
  bool owner;
};

// Small dense matrices live in one fixed device arena (row-major, ld = SM_LD).
#define SM_MAXK 256
#define SM_LD 256
enum { SM_GRAM = 0, SM_R, SM_RINV, SM_RTOT, SM_T, SM_V, SM_TMP, SM_TMP2, SM_AUX, SM_NSLOTS };

struct hfmi_status_words {  // device-resident, read back by the host after small kernels
  double min_pivot_ratio;   // min_j pivot_j / G_jj
  double gram_dev;          // || D^-1/2 G D^-1/2 - I ||_F  (orthonormality defect of the input)
  double offdiag;           // Jacobi: final off-diagonal norm / Frobenius norm
  int shifted;              // Cholesky needed a diagonal shift
  int failed;               // breakdown even after shifting / not converged
  int sweeps;               // Jacobi sweeps used
  int pad;
  long long tick[8];        // shader-clock section timings of the last small kernel (HFMI_DEBUG_TIMING=1 prints them)
};

struct hfmi_comm;
struct xfer_state;
struct hfmi_ctx {
  int device;
  hipStream_t stream;
  bool own_stream;
  hipEvent_t ev0, ev1;
  hipStream_t aux_stream;         // status read-backs that overlap work queued on `stream`
  hipEvent_t ev_status;
  hipEvent_t ev_side;             // small device -> host copies taken off the main stream (late QR checks, eigenvalues)
  int num_cus;
  void* ws[WS_NSLOTS];
  size_t ws_bytes[WS_NSLOTS];
  double* small;                  // SM_NSLOTS * SM_MAXK * SM_LD doubles
  hfmi_status_words* status_dev;  // device
  hfmi_status_words* status_host; // pinned
  void* pinned;                   // pinned host staging
  size_t pinned_bytes;
  std::vector<hfmi_block*> tmp_blocks;  // cached temporaries for the fused solves
  // storage of destroyed blocks kept for the next hfmi_block_create of the same size (a solve per step returns a fresh
  // U and the caller drops the previous one: without the pool every step pays a hipFree -- a device synchronisation --
  // and a hipMalloc); entries <= 2 GB, at most 16 entries / 8 GB, flushed when an allocation fails
  struct pool_entry { void* p; size_t bytes; };
  std::vector<pool_entry> pool;
  size_t pool_bytes;
  // optional per-launch event timing of the two MFMA kernel classes
  bool profiling;
  struct prof_rec { int kind; int64_t m, k, N; hipEvent_t e0, e1; double flops, bytes; };
  std::vector<prof_rec> prof;
  // phases of the fused solves (HFMI_PHASE_*, include/hfmi.h): event pairs on the stream while profiling, plus the
  // host-side wall clock of the host-callback legs
  struct phase_rec { int phase; hipEvent_t e0, e1; };
  std::vector<phase_rec> phase_events;
  double phase_ms[HFMI_PHASE_COUNT];
  // second pinned staging area (host-callback operators: double-buffered W and Y slabs)
  // row-panel hook of tsgemm_nn (hfmi_gemm_nn.hip): when set, a large product is issued as a few launches over consecutive
  // row ranges and the hook is called after each with the rows that are final -- hfmi_op_apply reduces those rows over the
  // ranks on the auxiliary stream while the next panel is computed
  int (*nn_hook)(void* user, double* Y, int64_t ldy, int r, int64_t row0, int64_t rows);
  void* nn_hook_user;
  int nn_hook_panels;
  bool nn_hook_called;
  // set around ONE launch_tsgemm_nn call by a caller that knows its small matrix is upper triangular (Q <- Q R^-1 of the QR):
  // the resident-S kernel then skips the column tiles that are structurally zero at each reduction step (bit-identical results)
  bool nn_upper_hint;
  hipEvent_t ev_panel[8], ev_join;
  // streaming ingest: uploads from pinned host memory on their own stream, a ring of completion events = tickets
  hipStream_t ingest_stream;
  hipEvent_t ev_ingest[HFMI_INGEST_RING];
  int64_t ingest_seq;
  void* late_pinned;              // status words / R_jj table of an orthogonalisation pass taken on trust (hfmi_api.hip)
  std::vector<hfmi_comm*> watched_comms;   // communicators whose device-side error word the host synchronisation points check
  void* pinned_cb;
  size_t pinned_cb_bytes;
  hipEvent_t ev_cb[4];            // D2H done x2, H2D done x2
  xfer_state* xfer;               // pinned ring + host threads of the large host <-> device transfers (hfmi_xfer.hip), lazily created
};
// large transfers between the caller's pageable arrays and device memory, pipelined through pinned chunks (hfmi_xfer.hip)
int xfer_d2h(hfmi_ctx* ctx, void* host, const void* dev, size_t bytes);   // returns when the host array is complete
int xfer_h2d(hfmi_ctx* ctx, void* dev, const void* host, size_t bytes);   // returns when the host array has been read
void xfer_destroy(hfmi_ctx* ctx);
int phase_begin(hfmi_ctx* ctx, int phase);     // returns a record index or -1 when not profiling
void phase_end(hfmi_ctx* ctx, int idx);
int prof_start(hfmi_ctx* ctx, int kind, int64_t m, int64_t k, int64_t N);  // returns record index or -1
int prof_stop(hfmi_ctx* ctx, int idx);

static inline double* sm_ptr(hfmi_ctx* c, int slot) { return c->small + (size_t)slot * SM_MAXK * SM_LD; }
int ctx_ws(hfmi_ctx* ctx, int slot, size_t bytes, void** out);
int ctx_pinned(hfmi_ctx* ctx, size_t bytes, void** out);
int ctx_tmp_block(hfmi_ctx* ctx, int idx, int64_t N, int nvec, hfmi_block** out);

struct hfmi_csr {
  hfmi_ctx* ctx;
  int64_t nrows, ncols, nnz;
  int64_t* indptr;
  int32_t* indices;
  double* data;
  double* inv_diag;  // Jacobi preconditioner (lazy)
  // ELLPACK image (slot-major: entry s of row i at [s * nrows + i]) built at creation when the rows are short and
  // even (FEM matrices): consecutive lanes then read consecutive (index, value) pairs.  ell_w == 0: CSR kernel only.
  int ell_w;
  int32_t* ell_idx;
  double* ell_val;
  // spectrum of the Jacobi-scaled matrix D^-1 A for the Chebyshev solve (hfmi_cheb.hip): an upper bound from Gershgorin's
  // circles (computed from the host arrays at creation), a lower bound from the Lanczos matrix of one scalar CG run on the
  // device (lazily, first solve); cheb_state: 0 = not estimated yet, 1 = usable, -1 = do not use (estimate failed / a solve
  // did not reach its tolerance)
  double gersh_lmax, cheb_lmin, cheb_lmax;
  int cheb_state;
};

enum hfmi_op_kind { OP_SNAPSHOT_GRAM, OP_JTJ, OP_JJT, OP_DENSE_SYM, OP_CSR, OP_CSR_PCG, OP_COMPOSE3, OP_HOST };

struct hfmi_op {
  hfmi_ctx* ctx;
  hfmi_op_kind kind;
  hfmi_block X;          // snapshot / Jacobian / dense block (by value: a view, not owned)
  int ndata, q;
  double* gamma_inv;     // device q x q row-major (ld = round_up(q,16)) or null
  double* weights;       // device, one per vector of X (general diagonal of a low-rank operator) or null
  double scale;
  const hfmi_csr* csr;
  double rel_tol;
  int max_iter;
  int last_iters;
  int last_method;        // sparse solver: 0 = block CG, 1 = Chebyshev (hfmi_op_solver_info)
  hfmi_op *a, *b, *c;
  hfmi_host_apply_fn host_fn;
  void* host_user;
  int64_t host_N;
  int host_chunk;        // > 0: the callback is invoked on slabs of this many vectors (pipelined with the copies)
  hfmi_post_apply_fn post_fn;
  void* post_user;
  hfmi_comm* comm;       // rank average / sum of the result block (hfmi_op_set_collective), null = none
  int comm_op;
  bool reduced_by_panels; // the last apply already reduced its result over the ranks, panel by panel (hfmi_api.hip)
};
// in-place all-reduce of `count` doubles of device memory on the communicator's context stream (hfmi_comm.hip)
int comm_allreduce_device(hfmi_comm* c, double* data, int64_t count, int op);
int comm_allreduce_device_on(hfmi_comm* c, double* data, int64_t count, int op, void* hip_stream /* null = the context's */);
int comm_transport(const hfmi_comm* c);   // 0 host, 1 rccl, 2 p2p
// HFMI_ERR_COMM if a stream-ordered collective of this communicator gave up (time-out) or met a peer that had; no device call
int comm_check_error(hfmi_comm* c);
// make the p2p staging buffer large enough for a collective of `bytes` NOW (a collective: every rank calls it with the same
// size), so that it never regrows while row panels of an application are in flight on the auxiliary stream
int comm_reserve_stage(hfmi_comm* c, size_t bytes);
void comm_forget_ctx(hfmi_comm* c);        // the context is going away first: stop referring to it
void ctx_watch_comm(hfmi_ctx* ctx, hfmi_comm* c);
void ctx_unwatch_comm(hfmi_ctx* ctx, hfmi_comm* c);
int ctx_check_comm(hfmi_ctx* ctx);        // comm_check_error over the watched communicators

// ------------------------------------------------------------------ kernel launchers (hfmi_gemm.hip)
// C (m x k) = scale * A^T B (+ beta * C); A: N x m, B: N x k column-major blocks.
// C is addressed as C[i*rs + j*cs] (device); out_colmajor selects the partial layout that makes
// the final write coalesced when rs == 1.
int launch_tsgemm_tn(hfmi_ctx* ctx, const double* A, int64_t lda, int m, const double* B, int64_t ldb, int k,
                     int64_t N, double scale, double beta, double* C, int64_t rs, int64_t cs, int nsplit_req);
// skinny x skinny variant (hfmi_skinny.hip): both operands staged through LDS, m, k <= 160 and m + k <= 288 columns
bool tsgemm_ss_applicable(int m, int k, bool same);
void tsgemm_ss_set_percu(int v);
void tsgemm_ss_set_blocked(int v);
int launch_tsgemm_ss(hfmi_ctx* ctx, const double* A, int64_t lda, int m, const double* B, int64_t ldb, int k,
                     int64_t N, double scale, double beta, double* C, int64_t rs, int64_t cs, int nsplit_req);
// C[i*rs + j*cs] = scale * sum_sp part[sp][..] + beta * C, fixed summation order
int launch_reduce_partials(hfmi_ctx* ctx, const double* part, int nsplit, int64_t pstride, int inner_ld, bool tr, int m,
                           int k, double scale, double beta, double* C, int64_t rs, int64_t cs);
// Y (N x r) = alpha * A (N x m) * S (m x r, device row-major, ld = lds, zero padded to 16 cols) + beta * Y
int launch_tsgemm_nn(hfmi_ctx* ctx, const double* A, int64_t lda, int m, const double* S, int lds, int r,
                     double alpha, double beta, double* Y, int64_t ldy, int64_t N);

// ------------------------------------------------------------------ kernel launchers (hfmi_misc.hip)
int launch_fill(hfmi_ctx* ctx, double* p, int64_t N, int nvec, int64_t ld, double value, bool include_pad);
int launch_zero_pad(hfmi_ctx* ctx, double* p, int64_t N, int nvec, int64_t ld);
int launch_row_scale(hfmi_ctx* ctx, double* G, int ld, int m, int k, const double* w);
int launch_matern32(hfmi_ctx* ctx, double* p, int64_t N, int nvec, int64_t ld, int nx, int ny, double sigma, double ell);
int launch_copy(hfmi_ctx* ctx, double* dst, int64_t ldd, const double* src, int64_t lds, int64_t N, int nvec);
int launch_scale(hfmi_ctx* ctx, double* p, int64_t ld, int64_t N, int nvec, double alpha);
int launch_axpy(hfmi_ctx* ctx, double* y, int64_t ldy, double alpha, const double* x, int64_t ldx, int64_t N, int nvec);
int launch_randn(hfmi_ctx* ctx, double* p, int64_t N, int nvec, int64_t ld, uint64_t seed, uint32_t stream, double sigma);
int launch_philox_raw(hfmi_ctx* ctx, uint32_t* out, int64_t N, int nvec, uint64_t seed, uint32_t stream);
// dense (N x nvec row-major, contiguous) <-> block
int launch_dense_to_block(hfmi_ctx* ctx, const double* dense, double* p, int64_t ld, int64_t N, int nvec);
int launch_block_to_dense(hfmi_ctx* ctx, const double* p, int64_t ld, double* dense, int64_t N, int nvec);
int launch_block_to_dense_ld(hfmi_ctx* ctx, const double* p, int64_t ld, double* dense, int64_t ldd, int64_t N, int nvec);
int launch_csr_spmm(hfmi_ctx* ctx, const hfmi_csr* M, const double* X, int64_t ldx, double* Y, int64_t ldy, int nvec,
                    bool accumulate);
int launch_csr_diag_inv(hfmi_ctx* ctx, hfmi_csr* M);
// Chebyshev iteration on row-major (nrows x k) work arrays (hfmi_cheb.hip)
int launch_cheb_first(hfmi_ctx* ctx, const double* B, double* X1, double* X0, const double* inv_diag, int64_t nrows, int k, double inv_theta);
int launch_cheb_step(hfmi_ctx* ctx, const hfmi_csr* M, const double* B, const double* Xc, double* Xp, int k, double c1, double c2, bool resid);
int launch_rm_colsq(hfmi_ctx* ctx, const double* A, int64_t nrows, int k, double* out);
int launch_dots_final(hfmi_ctx* ctx, const double* part, int nchunks, int nvec, double* out);
// Y = M X fused with the per-column dots dots[j] = <X_j, Y_j> (PCG: p . A p); needs the ELL image
int launch_ell_spmm_dot(hfmi_ctx* ctx, const hfmi_csr* M, const double* X, int64_t ldx, double* Y, int64_t ldy, int nvec,
                        double* dots);
// One fused PCG update per column j (alpha_j = rz_j / pap_j): y += alpha p, r -= alpha ap, then
// rz_new_j = <r, D^-1 r> and rr_j = <r, r> (device outputs, fixed summation order)
int launch_pcg_update(hfmi_ctx* ctx, double* y, int64_t ldy, double* r, int64_t ldr, const double* p, int64_t ldp,
                      const double* ap, int64_t ldap, const double* inv_diag, int64_t N, int nvec, const double* rz,
                      const double* pap, double* rz_new, double* rr);
// p_j = D^-1 r_j + (rz_new_j / rz_j) p_j
int launch_pcg_direction(hfmi_ctx* ctx, double* p, int64_t ldp, const double* r, int64_t ldr, const double* inv_diag,
                         int64_t N, int nvec, const double* rz_new, const double* rz);
// out[j] = <A_j, B_j> for j < nvec (device out)
int launch_col_dots(hfmi_ctx* ctx, const double* A, int64_t lda, const double* B, int64_t ldb, int64_t N, int nvec,
                    double* out);
// per-column updates used by PCG / MGS (coefficients on device)
// y_j += sign * (num_j / den_j) * x_j   (den may be null -> coefficient num_j)
int launch_col_axpy_dev(hfmi_ctx* ctx, double* y, int64_t ldy, const double* x, int64_t ldx, int64_t N, int nvec,
                        const double* num, const double* den, double sign);
// p_j = z_j + (num_j/den_j) * p_j
int launch_col_xpby_dev(hfmi_ctx* ctx, double* p, int64_t ldp, const double* z, int64_t ldz, int64_t N, int nvec,
                        const double* num, const double* den);
// z_j = inv_diag .* r_j
int launch_diag_scale(hfmi_ctx* ctx, double* z, int64_t ldz, const double* r, int64_t ldr, const double* inv_diag,
                      int64_t N, int nvec);
// small q x q (row-major) applied to each sample's q x k slab of G in place: G_i <- Gamma G_i
int launch_gamma_apply(hfmi_ctx* ctx, double* G, int ldg, int ndata, int q, int k, const double* gamma, int ldgam);
int launch_gamma_apply_cm(hfmi_ctx* ctx, const double* Gc, double* out, int64_t ld, int ndata, int q, int k, const double* gamma,
                          int ldgam);

// ------------------------------------------------------------------ small dense kernels (hfmi_small.hip)
// G (k x k, SM slot) = R^T R; writes R (upper), Rinv (upper); if rtot_accumulate, Rtot <- R * Rtot.
// Status words (min pivot ratio, defect, shifted/failed) land in ctx->status_dev.
int launch_chol_inv(hfmi_ctx* ctx, int k, int slot_gram, int slot_r, int slot_rinv, int slot_rtot, int rtot_mode,
                    int full_r, double shift_rel, double pivot_tol);
// T (k x k) -> eigenvalues (sorted descending) into dvals (device, k), eigenvectors into slot_v (columns).
int launch_jacobi_eig(hfmi_ctx* ctx, int k, int slot_t, int slot_v, double* dvals, int sort_by_abs);
// the same contract by Householder tridiagonalisation + divide and conquer (hfmi_eig_dc.hip): LAPACK dsyevd's algorithm
// family, i.e. what the reference's np.linalg.eigh runs; absolute accuracy eps ||T||
int launch_dc_eig(hfmi_ctx* ctx, int k, int slot_t, int slot_v, double* dvals, int sort_by_abs);
// dispatcher: method 0 = default (divide and conquer unless tuning "eig" = 1), 1 = Jacobi (high relative accuracy)
int launch_sym_eig(hfmi_ctx* ctx, int k, int slot_t, int slot_v, double* dvals, int sort_by_abs, int method);
int eig_tuning_set(const char* key, int value);   // 1 = key handled
int chol_tuning_set(const char* key, int value);  // "chol": 0 = blocked MFMA kernel (default), 1 = column-at-a-time kernels
int launch_chol_mfma(hfmi_ctx* ctx, int k, int slot_gram, int slot_r, int slot_rinv, int slot_rtot, int rtot_mode, int full_r,
                     double shift_rel, double pivot_tol);   // hfmi_chol.hip
int api_tuning_set(const char* key, int value);   // 1 = key handled ("comm_panels")
int launch_small_set_identity(hfmi_ctx* ctx, int k, int slot);
// slot_c (k x r, zero padded to 16 columns) = slot_a (k x k) * slot_b[:, :r]
int launch_small_matmul(hfmi_ctx* ctx, int k, int r, int slot_a, int slot_b, int slot_c);
// R (k x k, slot_r) = U diag(s) V^T: singular values descending in svals (device), U -> slot_u, V -> slot_v (columns)
int launch_jacobi_svd(hfmi_ctx* ctx, int k, int slot_r, int slot_u, int slot_v, double* svals);

constexpr int HFMI_EIG_MAXN = 16384;     // largest symmetric eigenproblem (hfmi_sym_eig_small / _leading, hfmi_block_gram_eig)
// symmetric eigensolve for 256 < n <= HFMI_EIG_MAXN: host in, host out.  hfmi_eig_blocked.hip (panel tridiagonalisation on the MFMA, divide
// and conquer, block-reflector back-transformation); HFMI_EIG_LARGE=jacobi selects the two-sided Jacobi of hfmi_eig_large.hip
// nvec < 0 or > n: all eigenvectors; otherwise host_V is n x nvec (the leading eigenvectors in output order)
// dev_T: the matrix in device memory (n x n row-major) instead of host_T
int sym_eig_large(hfmi_ctx* ctx, const double* host_T, int n, int sort_by_abs, double* host_d, double* host_V, int nvec = -1,
                  const double* dev_T = nullptr);

int eig_dgemm_bench(hfmi_ctx* ctx, int M, int N, int K, int ta, int tb, int reps, const double* host_A, const double* host_B,
                    double* host_C, double* avg_ms);
// micro-benchmarks
int launch_bench_peaks(hfmi_ctx* ctx, double* mfma_tflops, double* fma_tflops, double* copy_gbs);
int launch_bench_loaded_peak(hfmi_ctx* ctx, double* mfma_tflops, double* copy_gbs);
int launch_bench_read(hfmi_ctx* ctx, double* read_gbs);
int launch_bench_random_peaks(hfmi_ctx* ctx, double* mfma_tflops, double* mfma_tflops_streaming, double* copy_gbs);
