/*
 * ttn.h — C ABI of the MI355X-native TT/QTT core-arithmetic backend ("libttn_hip.so").
 *
 * This is the drop-in boundary for ONE hot path of TensorTrainNumerics.jl v1.1.3: TT-operator
 * apply, TT dot / hadamard / + / scalar*, orthogonalize and the tt_compress! rounding sweep.  The
 * reference has no FFI of its own (it is pure Julia; L1 is reached by multiple dispatch), so every
 * entry point below cites the Julia method it replaces; the Julia-side `ccall` binding a maintainer
 * would add is shown in INTEGRATION.md and julia/TTNBackend.jl.
 *
 * Conventions
 *   - plain C, no C++/torch types; all integers are int64_t, all payload is double (fp64);
 *   - arrays are COLUMN-MAJOR exactly as the reference stores them:
 *       vector core  k : (n_k, r_{k-1}, r_k)            offset i + n*(a + r_{k-1}*b)
 *       operator core k: (n_k, n_k, R_{k-1}, R_k)       offset i + n*(j + n*(a + R_{k-1}*b))
 *     (src/tt_tools.jl:23-29, :48-54); rank vectors have length d+1, `ot` vectors length d;
 *   - site numbers (`k`, `center`) are 1-based like the reference;
 *   - every function returns int: 0 = ok, <0 = argument error (the Julia shim maps these to the
 *     AssertionError the reference throws), >0 = HIP runtime error (hipError_t value).
 *     Functions never throw and never abort.  ttn_last_error_string() describes the last failure.
 *   - the library is usable from any host thread; calls are serialised internally on one HIP stream.
 */
#ifndef TTN_H
#define TTN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes (negative = argument errors, mirrored as AssertionError on the host side) ---- */
#define TTN_OK                 0
#define TTN_ERR_DIMS          -1   /* "Incompatible dimensions"  (tt_operations.jl:11,102,240,344) */
#define TTN_ERR_BOND_INDEX    -2   /* "k must be in 1:(N-1)"     (tt_tools.jl:744)                  */
#define TTN_ERR_SWEEPS        -3   /* "sweeps must be >= 1"      (tt_tools.jl:773)                  */
#define TTN_ERR_CENTER        -4   /* "Impossible orthogonalization" (tt_tools.jl:513)              */
#define TTN_ERR_CAPACITY      -5   /* destination handle too small for the result ranks             */
#define TTN_ERR_ARG           -6   /* null pointer / non-positive size / bad handle                 */
#define TTN_ERR_NOT_INIT      -7   /* ttn_init was not called (or failed)                           */
#define TTN_ERR_UNSUPPORTED   -8   /* shape outside what the kernels support (stated in DESIGN.md)  */
#define TTN_ERR_NO_CONVERGENCE -9  /* Jacobi SVD hit its sweep limit on some bond                   */
#define TTN_ERR_SINGULAR      -10  /* als_linsolve: a local system K is singular (LAPACK's SingularException)  */

/* ---- library lifetime ------------------------------------------------------------------------ */
int         ttn_init(int device);            /* binds to HIP device `device`, creates the stream   */
int         ttn_finalize(void);
const char* ttn_version(void);
const char* ttn_last_error_string(void);
int         ttn_sync(void);                  /* hipStreamSynchronize on the library stream          */
int         ttn_device_count(int* n);        /* hipGetDeviceCount (does not initialise a device)    */

/* ---- device-resident handles ------------------------------------------------------------------
 * A ttn_tt is a BATCH of `batch` TT vectors with common d/dims and a common per-bond rank CAPACITY;
 * each train carries its own current ranks (they live on the device because tt_compress! makes them
 * data dependent).  One contiguous fp64 arena: train b, core k at  data + b*stride + off[k].
 * A ttn_tto is one TT operator shared by every train of a batch.
 * Replaces: TTvector / TToperator containers (src/tt_tools.jl:23-29, :48-54).
 */
typedef struct ttn_tt_s*  ttn_tt_t;
typedef struct ttn_tto_s* ttn_tto_t;

int ttn_tt_create(int64_t d, const int64_t* dims, const int64_t* cap_rks, int64_t batch, ttn_tt_t* out);
int ttn_tt_free(ttn_tt_t h);
/* cores[k] points at n_k*rks[k]*rks[k+1] doubles (host memory, column-major) */
int ttn_tt_upload(ttn_tt_t h, int64_t b, const double* const* cores, const int64_t* rks, const int64_t* ot);
/* copies train `src_b` (cores + ranks + ot) over every other train of the batch, on the device */
int ttn_tt_replicate(ttn_tt_t h, int64_t src_b);
/* current ranks / ot flags of train b (synchronises the stream) */
int ttn_tt_ranks(ttn_tt_t h, int64_t b, int64_t* rks, int64_t* ot);
/* bound[m] = max over the trains of the CURRENT rank of bond m (synchronises); also tightens the host-side rank
 * bounds the capacity checks of the other calls use (they only know upper bounds after data-dependent truncations) */
int ttn_tt_max_ranks(ttn_tt_t h, int64_t* bound);
/* cores[k] must have room for n_k*rks[k]*rks[k+1] doubles with the CURRENT ranks (see ttn_tt_ranks) */
int ttn_tt_download(ttn_tt_t h, int64_t b, double* const* cores);
int ttn_tt_batch(ttn_tt_t h, int64_t* batch);
/* device copy dst <- src (same d/dims, dst capacity >= src current ranks) */
int ttn_tt_copy(ttn_tt_t dst, ttn_tt_t src);

int ttn_tto_create(int64_t d, const int64_t* dims, const int64_t* rks, const double* const* cores, ttn_tto_t* out);
int ttn_tto_free(ttn_tto_t h);

/* ---- the hot path on handles (every op runs on all trains of the batch, asynchronously) -------- */

/* y = A * x       replaces *(A::TToperator, v::TTvector), src/tt_operations.jl:101-111 */
int ttn_apply(ttn_tto_t A, ttn_tt_t x, ttn_tt_t y);

/* tt_compress!(psi, max_bond; truncerr, sweeps)   src/tt_tools.jl:772-789 (+ :743-768 per bond and the
 * effective _svdtrunc, src/tt_cross_interpolation.jl:149-166).  The reference's trailing
 * `orthogonalize(psi; i=k)` (:769) is not computed: tt_compress! discards its value (:779,:785). */
int ttn_compress(ttn_tt_t psi, int64_t max_bond, double truncerr, int64_t sweeps);

/* Device-side status of the handle (synchronises).  Every call that runs bond steps or local solves on psi — ttn_compress,
 * ttn_apply_compress, ttn_bond_truncate, ttn_sweep, ttn_apply_sweep, ttn_swap_sites, ttn_hadamard_ttm, ttn_ttv_decomp, the linear
 * solvers — records per train the FIRST condition its kernels meet: TTN_ERR_CAPACITY (a rank outgrew its slot),
 * TTN_ERR_NO_CONVERGENCE (a Jacobi SVD / eigensolver hit its sweep limit), TTN_ERR_SINGULAR (singular local system).  The record is
 * STICKY per handle: it survives later calls and is returned — and cleared — by this query, so a chain of asynchronous calls needs
 * one query at its end.  If non-null, total_jacobi_sweeps[b] receives the number of Jacobi sweeps train b used in the last call
 * (diagnostics). */
int ttn_compress_status(ttn_tt_t psi, int64_t* total_jacobi_sweeps);
/* One query for a whole chain (synchronises once): the most severe failure code recorded on ANY live handle (left in place
 * there: ttn_compress_status of that handle still reports and clears it) or on a handle that was FREED before anybody queried it
 * (ttn_tt_free folds an unread code into a library-level word, cleared here).  A device-resident chain that creates and frees
 * temporaries (RK4 stages, Krylov vectors) needs this once per time step / iteration instead of one query per compress. */
int ttn_status_all(void);

/* Rank bounds of tt_compress! (k = 0) or one _tt_bond_truncate! (k = 1-based bond).  The reference keeps
 * r = min(length(s), max_bond) singular values (tt_cross_interpolation.jl:152,164), so the rank of a rank-deficient
 * bond can GROW up to min(n_k r_{k-1}, n_{k+1} r_{k+1}, max_bond).  need[m] >= rks[m] is the capacity a handle / a
 * host buffer must have for bond m; fin[m] bounds the ranks after the call.  Either output may be null. */
int ttn_compress_rank_bound(int64_t d, const int64_t* dims, const int64_t* rks, int64_t max_bond, int64_t sweeps, int64_t k,
                            int64_t* need, int64_t* fin);

/* _tt_bond_truncate!(psi, k; max_bond, truncerr) without the discarded orthogonalize; k is 1-based */
int ttn_bond_truncate(ttn_tt_t psi, int64_t k, int64_t max_bond, double truncerr);

/* --- core-wise sharded chains (one segment of the chain per GPU, boundary-core hand-offs between neighbours).
 * A segment is an ordinary handle whose boundary ranks need not be 1.
 * ttn_sweep: the bond steps of src/tt_tools.jl:780-785 restricted to one direction over the bonds k_first .. k_last
 * (1-based like ttn_bond_truncate; descending when k_first > k_last).
 * ttn_tt_core_extent / _export / _import: move core k (1-based) of every train to / from a dense device buffer
 * [batch][dims[k]*bound_left*bound_right] plus its two ranks [batch][2] (device pointers, e.g. of the tensor handed to
 * ncclSend/ncclRecv); import sets the host-side rank bounds the sender reports. */
int ttn_sweep(ttn_tt_t psi, int64_t k_first, int64_t k_last, int64_t max_bond, double truncerr);
/* The fused form for a segment: ttn_apply_begin gives y the ranks of A*x (no core is written); a boundary core may then be imported
 * into y (ttn_tt_core_import); ttn_apply_sweep runs ONE L->R pass over the bonds k_first <= ... <= k_last with the cores right of
 * the moving front still virtual (built from A and x inside the bond step, like ttn_apply_compress) — every core of the range is
 * real afterwards.  first_core_real != 0: core k_first was imported and is used as it is; 0: it is written out first.
 * ttn_stream_handle: the hipStream_t all calls are enqueued on, for event-based ordering against the caller's own streams. */
int ttn_apply_begin(ttn_tto_t A, ttn_tt_t x, ttn_tt_t y);
int ttn_apply_sweep(ttn_tto_t A, ttn_tt_t x, ttn_tt_t y, int64_t k_first, int64_t k_last, int64_t max_bond, double truncerr, int first_core_real);
int ttn_stream_handle(void** stream);
int ttn_tt_core_extent(ttn_tt_t h, int64_t k, int64_t* doubles_per_train, int64_t* bound_left, int64_t* bound_right);
int ttn_tt_core_export(ttn_tt_t h, int64_t k, double* dev_buf, int64_t* dev_rks2);
int ttn_tt_core_import(ttn_tt_t h, int64_t k, const double* dev_buf, const int64_t* dev_rks2, int64_t bound_left, int64_t bound_right);

/* --- site-swap chains: the two-site SVD step with the physical indices of the two cores exchanged
 * (_ttm_swap!, src/tt_operations.jl:365-382; _swap_adjacent_sites, src/qtt_tools.jl:660-695): factors U and S*Vt.
 * All physical dimensions of the train must be equal (QTT: 2) and n * rank capacity <= 256.
 * ttn_hadamard_ttm: z = hadamard_ttm(x, y; tol, rmax) (src/tt_operations.jl:398-422; d(d-1)/2 swaps + d site-wise
 *   contractions, rank rule = the relative tail norm of tt_cross_interpolation.jl:149-166).  work_cap = rank capacity of the
 *   2d working slots; a rank above it (or above z's capacity) is reported by ttn_compress_status as TTN_ERR_CAPACITY.
 * ttn_swap_sites: the swap list of reorder (src/qtt_tools.jl:763-769) in place; swaps[i] = k (1-based) exchanges sites
 *   k and k+1; rank rule = count(s > threshold * s[1]) (at least 1), or every singular value when threshold == 0. */
int ttn_hadamard_ttm(ttn_tt_t x, ttn_tt_t y, ttn_tt_t z, double tol, int64_t rmax, int64_t work_cap);
int ttn_swap_sites(ttn_tt_t x, int64_t nswaps, const int64_t* swaps, double threshold);

/* z_b = ttv_decomp(tensor_b; index, tol) (src/tt_tools.jl:186-252): hierarchical SVD of `batch` dense tensors of shape
 * z.dims (HOST memory, [batch][prod(dims)], column-major like a Julia Array), root at site `index` (1-based), singular values
 * below `tol` discarded (absolute).  Ranks are bounded by the handle's capacity: a larger rank is reported by
 * ttn_compress_status as TTN_ERR_CAPACITY.  Sets the orthogonality flags -1 / 0 / +1 like the reference.  Synchronises. */
int ttn_ttv_decomp(ttn_tt_t z, const double* tensors, int64_t index, double tol);

/* --- als_linsolve(A, b, tt_start; sweep_count) (src/solvers/als.jl:161-222): x = the ALS iterate after
 * sweep_count half sweeps, for every train of the batch (one operator, `batch` right-hand sides b and start trains x0).
 * x receives orthogonalize(x0) first (als.jl:174) and keeps x0's ranks (als.jl:177); the local systems are assembled
 * densely and solved by LU with partial pivoting like the reference's `K \ Pb` (als.jl:58-70).  All trains of x0 must carry
 * the same ranks.  Local systems of up to 2048 unknowns (n r_{i-1} r_i: ranks up to 32 for n = 2) run as ONE persistent launch, one
 * workgroup per train; larger ones — up to 65 536 unknowns; BASELINE config C5 names ranks up to 128 = 32 768 unknowns, an 8.6 GB K —
 * run in the GRID form: assembly and the blocked LU on the whole chip, the half sweeps walked by the host, one train after the other
 * (csrc/ttn_als_grid.h).  ttn_compress_status(x) reports a singular local system (TTN_ERR_SINGULAR). */
int ttn_als_linsolve(ttn_tto_t A, ttn_tt_t b, ttn_tt_t x0, ttn_tt_t x, int64_t sweep_count);

/* mals_linsolve(A, b, tt_start; tol, rmax) (src/solvers/mals.jl:240-312): one forward and one backward half sweep of
 * two-site solves, ranks adapted by the truncated SVD of every local solution (sv_trunc, clamped to rmax).  x receives
 * orthogonalize(x0) first; x's CAPACITY bounds the ranks (a larger rank: TTN_ERR_CAPACITY through ttn_compress_status) and
 * must keep every two-site system n_i cap_i n_{i+1} cap_{i+2} <= 2048. */
int ttn_mals_linsolve(ttn_tto_t A, ttn_tt_t b, ttn_tt_t x0, ttn_tt_t x, double tol, int64_t rmax);

/* dmrg_linsolve(A, b, tt_start; N = 2, tol, sweep_schedule, rmax_schedule) (src/solvers/dmrg.jl:388-472): two-site sweeps
 * (windows 1..d-2 forward with right_core_move!, d-1..2 backward with left_core_move!, dmrg.jl:187-232) walked through the
 * reference's stage schedule — sweep s ends stage j when s == sweep_schedule[j], the sweep that would end the last stage is
 * the closing solve at window 1 — with the ranks cut by cut_off_index (dmrg.jl:179-185) clamped to the stage's rmax.
 * Local systems up to 2048 unknowns are assembled densely and solved by LU (the reference's `K_full` + `K \ Pb` branch,
 * dmrg.jl:57-62, :173-175), larger ones matrix-free by conjugate gradients (see ttn_dmrg_linsolve_it).  sweep_schedule must be positive and
 * strictly increasing (anything else does not terminate in the reference); at most 32 full sweeps per call.  Capacity and
 * status as for ttn_mals_linsolve.  N = 1 is ttn_als_linsolve's territory and not offered here. */
int ttn_dmrg_linsolve(ttn_tto_t A, ttn_tt_t b, ttn_tt_t x0, ttn_tt_t x, double tol, int64_t n_stages, const int64_t* sweep_schedule,
                      const int64_t* rmax_schedule);

/* The same with the reference's local-solver keywords (dmrg.jl:392-396, :92-97): every local system with `it_solver != 0` or with
 * more than `itslv_thresh` unknowns is solved MATRIX-FREE by conjugate gradients on the symmetrised local operator (dmrg.jl:99-171:
 * the three-tensor sandwich G (x) Amid (x) H as fp64 MFMA GEMMs; KrylovKit's `linsolve(...; issymmetric, isposdef, tol, maxiter)`):
 * start vector = the current two-site block, stop at ||residual||_2 < linsolv_tol (absolute) or after linsolv_maxiter iterations.
 * The reference's defaults are it_solver = 1, linsolv_maxiter = 200, linsolv_tol = max(sqrt(tol), 1e-8), itslv_thresh = 256.
 * Dense systems are limited to 2048 unknowns; anything larger is solved matrix-free whatever the keywords say, so two-site
 * systems up to n_i cap_i = n_{i+1} cap_{i+2} = 256 (ranks 128 for n = 2: 65 536 unknowns, BASELINE config C5) are in reach.
 * ttn_dmrg_linsolve itself = it_solver 0, itslv_thresh 2048.  ttn_dmrg_cg_iterations: CG iterations per train of the last call. */
int ttn_dmrg_linsolve_it(ttn_tto_t A, ttn_tt_t b, ttn_tt_t x0, ttn_tt_t x, double tol, int64_t n_stages, const int64_t* sweep_schedule,
                         const int64_t* rmax_schedule, int it_solver, int64_t linsolv_maxiter, double linsolv_tol, int64_t itslv_thresh);
int ttn_dmrg_cg_iterations(int64_t batch, int64_t* iters);

/* --- TDVP local contractions (src/solvers/tdvp.jl:29-43, :205-208), batched, Float64 (cplx = 0) or ComplexF64
 * (cplx = 1: interleaved re/im pairs, as Julia stores them).  Tensors are column-major in the layouts tdvp1sweep! / tdvp2sweep! hold
 * them in — sites (l, s, r) = permutedims(ttv_vec[k], (2,1,3)), operator cores (a, s, b, s') = permutedims(tto_vec[k], (3,1,4,2))
 * (:52-53) — `batch` of each back to back; the operator core(s) may be one shared tensor (m_shared != 0).
 *   _applyH1_lsr      HAC[α,s,β]      = FL[α,a,α'] AC[α',s',β'] M[a,s,b,s'] FR[β',b,β]             FL (Dl,a,Dl)  AC (Dl,d,Dr)  M (a,d,b,d)  FR (Dr,b,Dr)
 *   _applyH0          HC[α,β]         = FL[α,a,α'] C[α',β'] FR[β',a,β]                             FL (Dl,a,Dl)  C (Dl,Dr)     FR (Dr,a,Dr)
 *   _update_left_env  FLnext[α,a,β]   = FL[α',a',β'] A[β',s',β] M[a',s,a,s'] conj(A[α',s,α])       A (Dl,d,Dr)  M (a_in,d,a_out,d)  FL (Dl,a_in,Dl) -> (Dr,a_out,Dr)
 *   _update_right_env FRprev[α,a,β]   = A[α,s',α'] FR[α',a',β'] M[a,s,a',s'] conj(A[β,s,β'])       A (Dl,d,Dr)  M (a_out,d,a_in,d)  FR (Dr,a_in,Dr) -> (Dl,a_out,Dl)
 *   _applyH2_lsr      HAAC[α,s1,s2,β] = FL[α,a,α'] AAC[α',s1',s2',β'] M1[a,s1,b,s1'] M2[b,s2,c,s2'] FR[β',c,β]
 * Each is a chain of fp64 MFMA GEMMs over strided views of the arrays as they lie (no permuted copies); a complex product is four
 * real ones.  The first five take DEVICE pointers (asynchronous on the library stream); ttn_tdvp_contract_f64 stages HOST arrays:
 * op = 0 applyH1, 1 applyH0, 2 update_left_env, 3 update_right_env, 4 applyH2; dims7 = {Dl, d (d1), Dr, a, b, c, d2} with a = a_in,
 * b = a_out for the environment updates; FL / FR / M1 / M2 that an op does not use are ignored (may be null); X = AC / C / A / AAC. */
int ttn_tdvp_apply_h1(int cplx, int64_t batch, int64_t Dl, int64_t d, int64_t Dr, int64_t a, int64_t b, const double* FL, const double* AC,
                      const double* M, const double* FR, double* HAC, int m_shared);
int ttn_tdvp_apply_h0(int cplx, int64_t batch, int64_t Dl, int64_t Dr, int64_t a, const double* FL, const double* C, const double* FR, double* HC);
int ttn_tdvp_update_left_env(int cplx, int64_t batch, int64_t Dl, int64_t d, int64_t Dr, int64_t a_in, int64_t a_out, const double* A, const double* M,
                             const double* FL, double* FLnext, int m_shared);
int ttn_tdvp_update_right_env(int cplx, int64_t batch, int64_t Dl, int64_t d, int64_t Dr, int64_t a_out, int64_t a_in, const double* A, const double* M,
                              const double* FR, double* FRprev, int m_shared);
int ttn_tdvp_apply_h2(int cplx, int64_t batch, int64_t Dl, int64_t d1, int64_t d2, int64_t Dr, int64_t a, int64_t b, int64_t c, const double* FL,
                      const double* AAC, const double* M1, const double* M2, const double* FR, double* HAAC, int m_shared);
int ttn_tdvp_contract_f64(int op, int cplx, int64_t batch, const int64_t* dims7, const double* FL, const double* FR, const double* X, const double* M1,
                          const double* M2, double* out, int m_shared);

/* --- dense moves of the TDVP sweeps on one local matrix (src/solvers/tdvp.jl:76-80, :120-126: qr(Aqr), qr(A'); :252, :276: the svd inside
 * _svdtrunc), Float64 (cplx = 0) or ComplexF64 (cplx = 1, interleaved), column-major, DEVICE pointers, the library's stream.
 * ttn_dense_qr: Householder as LAPACK's geqr2 + org2r: A (m x n, overwritten) -> Q (m x r), R (r x n), r = min(m, n); asynchronous.
 * ttn_dense_svd: one-sided Jacobi, m >= n (pass the conjugate transpose otherwise): A (overwritten) -> U (m x n), s (n, descending),
 * Vh (n x n), A = U diag(s) Vh; synchronises (the caller reads s to choose the rank); TTN_ERR_NO_CONVERGENCE after 60 sweeps. */
int ttn_dense_qr(int cplx, int64_t m, int64_t n, double* A, double* Q, double* R);
int ttn_dense_svd(int cplx, int64_t m, int64_t n, double* A, double* U, double* s, double* Vh);

/* fused convenience for the benchmark op  tt_compress!(A*x, max_bond)  (src/solvers/euler.jl:55) */
int ttn_apply_compress(ttn_tto_t A, ttn_tt_t x, ttn_tt_t y, int64_t max_bond, double truncerr, int64_t sweeps);

/* out[b] = dot(a_b, b_b)      src/tt_operations.jl:239-250 ; out is HOST memory, length batch (synchronises) */
int ttn_dot(ttn_tt_t a, ttn_tt_t b, double* out);
/* HIP-event time of the KERNEL of the last ttn_dot / ttn_norm / ttn_orthogonalize call alone (ttn_dot itself goes on to copy the
 * results to the host and synchronises, which an event pair around the call would include) — what bench.py --op reports */
int ttn_last_launch_ms(float* ms);
/* out[b] = norm(a_b) = sqrt(max(dot(a,a),0))   src/tt_operations.jl:465-470 */
int ttn_norm(ttn_tt_t a, double* out);

/* z = hadamard(x, y)   src/tt_operations.jl:343-361 */
int ttn_hadamard(ttn_tt_t x, ttn_tt_t y, ttn_tt_t z);
/* z = x + y            src/tt_operations.jl:10-35 (also the body of add!, :37-66) */
int ttn_add(ttn_tt_t x, ttn_tt_t y, ttn_tt_t z);
/* y = a * x            src/tt_operations.jl:256-266 (scales the first core with ot==0, else core 1;
 *                      a == 0 gives the all-zero train with ot reset to 0) */
int ttn_scale(double a, ttn_tt_t x, ttn_tt_t y);
/* y_b = a[b] * x_b with one factor per train (a is HOST memory, length batch): what `(1 / sqrt(dot(u, u))) * u` needs on a
 * batch (src/solvers/euler.jl:83-85, :205-207) */
int ttn_scale_batch(const double* a, ttn_tt_t x, ttn_tt_t y);
/* y = orthogonalize(x; i=center)   src/tt_tools.jl:511-543 ; center is 1-based.  QTT trains of rank <= 64 take three kernels
 * (csrc/ttn_ortho_ramp.h: one wave per train over the rank-ramp sites; csrc/ttn_ortho512.h: Cholesky-QR steps with a measured
 * orthogonality check over the tall sites and the centre core; the 1024-thread k_orthogonalize for the left sweep and for trains the
 * other two refuse) and the call reads one word back between them (it synchronises the library stream once); every other train
 * class is one asynchronous launch.  TTN_ORTHO512 = 0 / 1, TTN_ORTHO_RAMP = 0, TTN_ORTHO_CHOLQR = 0 / 1 switch routes off for A/B runs. */
int ttn_orthogonalize(ttn_tt_t x, int64_t center, ttn_tt_t y);

/* ---- parity instrumentation: singular values seen by the last ttn_compress / ttn_bond_truncate --
 * After ttn_sv_capture(h, 1), each bond step stores ALL singular values of its merged matrix (sorted,
 * descending, before truncation).  `step` counts bond steps of the last call from 0
 * (L->R k=1..N-1, then R->L k=N-1..1, per sweep).  Returns the count in *n (<= cap). */
int ttn_sv_capture(ttn_tt_t h, int enable);
int ttn_sv_get(ttn_tt_t h, int64_t b, int64_t step, double* out, int64_t cap, int64_t* n);

/* ---- timing on the library stream (HIP events) ------------------------------------------------- */
int ttn_timer_begin(void);
int ttn_timer_end(float* ms);   /* synchronises */
/* event slots (0..4095) recorded on the library stream without synchronising; elapsed() synchronises */
int ttn_event_record(int64_t slot);
int ttn_event_elapsed(int64_t slot_a, int64_t slot_b, float* ms);

/* kernel unit-test hook: C (m x n row-major, host, in/out) = alpha*op(A)*op(B) + beta*C computed by the device-side
 * workgroup GEMM (fp64 MFMA) every dense kernel is built on; ta/tb: operand stored transposed */
int ttn_selftest_gemm(int64_t m, int64_t n, int64_t k, const double* A, const double* B, double* C, double alpha, double beta,
                      int ta, int tb);

/* self-test of the symmetric eigensolver used by the Gram routes (csrc/ttn_eig_kernels.h): G host, n x n (n = 64 or 128),
 * column-major, symmetric positive definite; sig[nev] = sqrt of the nev largest eigenvalues (descending), X[128*r] = sig_j * u_j (r <= 64);
 * ticks_rc[0] = device clock ticks (s_memtime), [1] = return code of the device routine, [2..5] = ticks of the four phases
 * (tridiagonalisation, bisection, twisted factorisations, back-transformation); ticks_rc has 6 entries. */
int ttn_selftest_eig128(const double* G, int64_t n, int64_t r, int64_t nev, double* sig, double* X, int64_t* ticks_rc);
/* micro-benchmark hook: the same workgroup GEMM on ONE compute unit, `reps` times back to back on device-resident zeros;
 * cycles_out receives the shader-clock cycles (s_memtime) of the whole loop.  Used to price the dense phases against the
 * per-CU fp64 MFMA peak (128 flop/clk/CU). */
int ttn_bench_gemm(int64_t m, int64_t n, int64_t k, int ta, int tb, int64_t reps, int64_t* cycles_out);
/* same for the LDS-resident building blocks on an n x n SPD test matrix (n <= 128): what = 0 set-up only, 1 set-up + Cholesky,
 * 2 set-up + one-sided Jacobi of the matrix columns; sweeps_out (may be NULL) receives the Jacobi sweep count of the last rep */
int ttn_bench_lds(int what, int64_t n, int64_t reps, int64_t* cycles_out, int64_t* sweeps_out);

/* diagnostic: with TTN_PROF=1 in the environment ttn_compress records s_memtime ticks per phase
 * (merge, scale, LQ, Jacobi, sort/rank, split) for every train; out8 receives train b's 16 counters */
int ttn_prof_get(int64_t b, int64_t* out8);
/* per bond step (first 120 steps): (p << 32) | jacobi_sweeps, p = short side of the merged matrix */
int ttn_prof_steps(int64_t b, int64_t* out120);
int ttn_prof_fine(int64_t b, int64_t* out64);       /* TTN_PROF_STEP=k: 64 fine-grained cycle counters of bond step k of train b */
/* diagnostics of the last ttn_orthogonalize that took the multi-launch form (csrc/ttn_ortho_ramp.h, ttn_ortho512.h): the four state
 * words of train b = {next site of the right-to-left sweep, buffer of the last right factor, buffer of the last left factor,
 * 1 if k_ortho512 finished the train (0: the 1024-thread kernel took it over from `next site`)}. */
int ttn_debug_ortho_state(int64_t b, int64_t* out4);

/* ---- stateless host-pointer entry points: the literal drop-ins for one train --------------------
 * Each uploads, runs the handle op above, and downloads.  Output cores are caller-allocated:
 *   apply     : Y_cores[k] sized n_k*(A_rks[k]*X_rks[k])*(A_rks[k+1]*X_rks[k+1])   (as zeros_tt would)
 *   hadamard  : Z_cores[k] sized n_k*(rx*ry)_k*(rx*ry)_{k+1}
 *   add       : Z_cores[k] sized with ranks rx+ry (ends forced to 1)
 *   compress  : in/out cores sized n_k*need[k]*need[k+1] with need from ttn_compress_rank_bound (= the input
 *               ranks unless a rank-deficient bond can grow); on entry they hold the input cores compactly,
 *               on exit the new cores compactly; `rks` is updated in place
 *   orthogonalize: Y_cores sized for X_rks (output ranks never exceed the input's) ; Y_rks / Y_ot are outputs
 */
int ttn_apply_f64(int64_t d, const int64_t* dims,
                  const double* const* A_cores, const int64_t* A_rks,
                  const double* const* X_cores, const int64_t* X_rks,
                  double* const* Y_cores);
int ttn_dot_f64(int64_t d, const int64_t* dims,
                const double* const* A_cores, const int64_t* A_rks,
                const double* const* B_cores, const int64_t* B_rks, double* out);
int ttn_hadamard_f64(int64_t d, const int64_t* dims,
                     const double* const* X_cores, const int64_t* X_rks,
                     const double* const* Y_cores, const int64_t* Y_rks,
                     double* const* Z_cores);
int ttn_add_f64(int64_t d, const int64_t* dims,
                const double* const* X_cores, const int64_t* X_rks,
                const double* const* Y_cores, const int64_t* Y_rks,
                double* const* Z_cores);
int ttn_scale_f64(int64_t d, const int64_t* dims, double a,
                  const double* const* X_cores, const int64_t* X_rks, const int64_t* X_ot,
                  double* const* Y_cores, int64_t* Y_ot);
int ttn_orthogonalize_f64(int64_t d, const int64_t* dims,
                          const double* const* X_cores, const int64_t* X_rks, int64_t center,
                          double* const* Y_cores, int64_t* Y_rks, int64_t* Y_ot);
int ttn_compress_f64(int64_t d, const int64_t* dims, double* const* cores, int64_t* rks,
                     int64_t max_bond, double truncerr, int64_t sweeps);
int ttn_bond_truncate_f64(int64_t d, const int64_t* dims, double* const* cores, int64_t* rks,
                          int64_t k, int64_t max_bond, double truncerr);
/* op = x -> tt_compress!(A * x, max_bond) (src/solvers/euler.jl:55, the operator krylov_linsolve / the time steppers iterate) as ONE
 * stateless call: A * x is never materialised, neither in HBM nor over PCIe (fused apply).  Y_cores[k] sized n_k * cap_k * cap_{k+1}
 * with cap from ttn_apply_compress_rank_bound (the final ranks a sweep can leave: min(A_rks .* X_rks, max_bond, what the dimensions
 * allow)); Y_rks (d + 1) receives the ranks, the cores come back compactly with those ranks. */
int ttn_apply_compress_rank_bound(int64_t d, const int64_t* dims, const int64_t* A_rks, const int64_t* X_rks, int64_t max_bond, int64_t sweeps,
                                  int64_t* cap);
int ttn_apply_compress_f64(int64_t d, const int64_t* dims, const double* const* A_cores, const int64_t* A_rks,
                           const double* const* X_cores, const int64_t* X_rks, double* const* Y_cores, int64_t* Y_rks,
                           int64_t max_bond, double truncerr, int64_t sweeps);

/* r_and_d_to_rks(rks, dims; rmax)   src/tt_tools.jl:407-425 (host-side integer helper, bit-exact) */
int ttn_r_and_d_to_rks(int64_t d, const int64_t* dims, int64_t n_rks, const int64_t* rks, int64_t rmax, int64_t* out);

#ifdef __cplusplus
}
#endif
#endif /* TTN_H */
