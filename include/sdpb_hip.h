/* sdpb_hip.h — C ABI of the MI355X-native interior-point step for SDPB's sdp_solve.
 *
 * The reference has no plugin/FFI layer; the seam this library replaces is the C++
 * member SDP_Solver::step() (src/sdp_solve/SDP_Solver.hxx:94-112, defined in
 * src/sdp_solve/SDP_Solver/run/step/step.cxx:51-229) together with the pre-step half
 * of the loop body of SDP_Solver::run() (src/sdp_solve/SDP_Solver/run/run.cxx:380-435).
 * A thin C++ shim inside sdpb keeps the CLI, the SDP reader and the writers and calls
 * these entry points instead (INTEGRATION.md shows the binding).
 *
 * Conventions
 *  - Every call returns int: 0 ok; 1 a Cholesky factorisation met a matrix that is not
 *    positive definite (message names the block/parity exactly like the reference's
 *    RUNTIME_ERRORs: cholesky_decomposition.cxx:22-25, compute_Q.cxx:36-38,
 *    initialize_schur_complement_solver.cxx:100-103); 2 out of device memory; 3 HIP or
 *    collective failure; 4 bad argument.  sdpb_hip_last_error() returns the text; the
 *    shim rethrows it as RUNTIME_ERROR so main()'s abort path (src/sdpb/main.cxx:180-188)
 *    is unchanged.
 *  - Numbers cross the boundary as decimal strings (what the SDP files and sdpb's
 *    outputs contain, Json_Block_Data_Parser.hxx:26-36, print_iteration.cxx:91-104);
 *    lists are whitespace-separated.  They are converted exactly (then truncated to
 *    the working precision) — El::BigFloat(str) semantics.
 *  - Calls are synchronous and not re-entrant per context; one host thread per GPU.
 *  - There is no CPU path: sdpb_hip_create fails with code 3 if no GPU is present.
 */
#ifndef SDPB_HIP_H
#define SDPB_HIP_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sdpb_hip_ctx sdpb_hip_ctx;

/* Replaces Block_Info (sizes: src/sdp_solve/Block_Info.hxx:54-119), the SDP_Solver
 * constructor (src/sdp_solve/SDP_Solver/SDP_Solver.cxx:3-51) and
 * initialize_bigint_syrk_context (run.cxx:250-255).  dims[j] = m_j, num_points[j] =
 * K_j (block_info_<j>.json), N = length of b.  Blocks are assigned to ranks by a
 * deterministic cost model (analogue of compute_block_grid_mapping.hxx:58-183);
 * rank/world_size describe this process (one process per GPU).  device_id < 0 keeps
 * the current device.  precision_bits is sdpb's --precision (Solver_Parameters.cxx:20-26): it is
 * rounded up to the next compiled mantissa width (128 ... 2048 bits; code 4 with a message naming
 * the range beyond that), as GMP rounds a precision up to whole limbs. */
int sdpb_hip_create(int precision_bits, int num_blocks, const int *dims, const int *num_points, int N, int device_id,
                    int rank, int world_size, sdpb_hip_ctx **out);
void sdpb_hip_destroy(sdpb_hip_ctx *ctx);
/* ctx may be NULL: message of the last failed sdpb_hip_create on this thread. */
const char *sdpb_hip_last_error(sdpb_hip_ctx *ctx);

/* Solver_Parameters (src/sdp_solve/Solver_Parameters.hxx:13-30).  name is the sdpb
 * option name: dualityGapThreshold, primalErrorThreshold, dualErrorThreshold,
 * initialMatrixScalePrimal, initialMatrixScaleDual, feasibleCenteringParameter,
 * infeasibleCenteringParameter, stepLengthReduction, maxComplementarity,
 * minPrimalStep, minDualStep.  Defaults are Solver_Parameters.cxx:10-157. */
int sdpb_hip_set_param(sdpb_hip_ctx *ctx, const char *name, const char *value);
int sdpb_hip_set_flags(sdpb_hip_ctx *ctx, long max_iterations, int find_primal_feasible, int find_dual_feasible,
                       int detect_primal_feasible_jump, int detect_dual_feasible_jump);

/* struct SDP (src/sdp_solve/SDP.hxx:74-122).  Row-major lists exactly as in
 * block_data_<j>.json: bilinear_bases_even/odd[row][k], B[p][n], c[p]. Blocks owned by
 * another rank are accepted and ignored, so every rank may be fed the whole SDP. */
int sdpb_hip_set_block(sdpb_hip_ctx *ctx, int j, const char *bilinear_bases_even, const char *bilinear_bases_odd,
                       const char *B, const char *c);
/* Same block, with B (row-major P x N) and c passed as IEEE doubles (converted exactly):
 * bulk path for inputs whose entries are dyadic rationals (synthetic benchmarks). */
int sdpb_hip_set_block_f64(sdpb_hip_ctx *ctx, int j, const char *bilinear_bases_even, const char *bilinear_bases_odd,
                           const double *B, const double *c);
/* objectives.json: "b" and "constant" (src/sdp_solve/SDP/read_objectives.cxx:22-36). */
int sdpb_hip_set_objective(sdpb_hip_ctx *ctx, const char *b, const char *constant);

/* x = 0, y = 0, X = initialMatrixScalePrimal * I, Y = initialMatrixScaleDual * I
 * (SDP_Solver.cxx:23-38). */
int sdpb_hip_init_state(sdpb_hip_ctx *ctx);

/* One pass of the loop body of SDP_Solver::run (run.cxx:380-435): objectives,
 * Cholesky of X and Y, bilinear pairings, residues and errors, feasibility/termination
 * test, then SDP_Solver::step.  *terminated = 1 when the reference loop would `break`
 * (reason via sdpb_hip_terminate_reason); checkpointing stays with the caller's loop,
 * --maxRuntime and the graceful stop are tested inside (sdpb_hip_set_max_runtime,
 * sdpb_hip_request_stop).  Three host synchronisation points per call.  Collective over
 * all ranks. */
int sdpb_hip_iterate(sdpb_hip_ctx *ctx, int *terminated);
/* SDP_Solver_Terminate_Reason (src/sdp_solve/SDP_Solver_Terminate_Reason.hxx:9-21) in
 * declaration order (0 PrimalDualOptimal ... 5 MaxComplementarityExceeded, 6 MaxIterationsExceeded,
 * 7 MaxRuntimeExceeded, 8 PrimalStepTooSmall, 9 DualStepTooSmall, 10 SIGTERM_Received), -1 while running; the string is operator<<'s text
 * (SDP_Solver_Terminate_Reason.cxx:4-45). */
int sdpb_hip_terminate_reason(sdpb_hip_ctx *ctx);
const char *sdpb_hip_terminate_string(sdpb_hip_ctx *ctx);

/* Per-iteration scalars under their iterations.json keys (print_iteration.cxx:91-104):
 * mu P-obj D-obj gap P-err p-err D-err R-err P-step D-step beta Q_cond_number
 * max_block_cond_number block_name; and out.txt keys (src/sdpb/save_solution.cxx:32-37):
 * primalObjective dualObjective dualityGap primalError dualError.  Writes a
 * NUL-terminated decimal string; returns 4 if buf is too small (*needed = size). */
int sdpb_hip_get_scalar(sdpb_hip_ctx *ctx, const char *name, char *buf, size_t buflen, size_t *needed);

/* Solver state and work arrays, column-major, one decimal per line.  which: x X y Y
 * (SDP_Solver.hxx:28-43; checkpoint / save_solution.cxx:67-150), dx dX dy dY,
 * dual_residues c_minus_By (save_c_minus_By.hxx:18-47) primal_residues primal_residue_p, Q S AXinv AY ...  j = global block
 * index (must be owned by this rank), parity 0/1 for block-diagonal members. */
int sdpb_hip_get_array(sdpb_hip_ctx *ctx, const char *which, int j, int parity, char *buf, size_t buflen,
                       size_t *needed);
/* Inject x, X, y or Y (text checkpoint: load_text_checkpoint.cxx:6-44); dx / dy for sdpb_hip_schur_solve; c = the primal
 * objective entries of block j (SDP.hxx:84-100) as decimals, for callers that fed B and c as doubles (sdpb_hip_set_block_f64)
 * although c is no double. */
int sdpb_hip_set_array(sdpb_hip_ctx *ctx, const char *which, int j, int parity, const char *values);

/* The part of the step that approx_objective and outer_limits reuse
 * (approx_objective/setup_solver.cxx:204-220, outer_limits/compute_optimal/compute_optimal.cxx:188-215):
 * from the CURRENT X and Y (e.g. loaded with sdpb_hip_set_array from a text checkpoint) build
 *   schur_complement_cholesky  L_j = chol(S_j)       -> sdpb_hip_get_array("L", j)
 *   schur_off_diagonal         P_j = L_j^{-1} B_j    -> sdpb_hip_get_array("PT", j)  (transposed, N x P_j)
 *   Cholesky(Q)                                      -> sdpb_hip_get_array("Q")      (lower factor)
 * without modifying x, X, y, Y; same errors as sdpb_hip_iterate.  sdpb_hip_schur_solve then solves
 * the Schur complement equation (solve_schur_complement_equation.cxx:16-79) for right-hand sides
 * placed with sdpb_hip_set_array("dx", j) / ("dy"); the solution replaces them. */
int sdpb_hip_schur_solver_init(sdpb_hip_ctx *ctx);
int sdpb_hip_schur_solve(sdpb_hip_ctx *ctx);

/* ---- binary number path -------------------------------------------------------
 * The same entry points with every number as a fixed-width record in GMP's mpf_t layout, so
 * a C++ caller holding El::BigFloat (= mpf_t, fmpz_BigFloat_convert.hxx:9,13) never formats
 * decimals: one record = 2 + limbs64 64-bit words,
 *   word 0 = (int64) _mp_size (signed count of used limbs; 0 = zero),
 *   word 1 = (int64) _mp_exp  (exponent in 64-bit limbs),
 *   words 2.. = _mp_d[0 .. limbs64), least significant limb first (unused limbs ignored),
 *   value = sign * (sum_i d[i] 2^(64 i)) * 2^(64 (_mp_exp - |_mp_size|)).
 * Conversions truncate toward zero like mpf; reading back is exact when
 * limbs64 >= sdpb_hip_limbs(ctx)/2 + 1 (GMP's own _mp_prec + 1 at the same --precision).
 * Shapes and orders are those of the text entry points (B row-major P x N, arrays column-major). */
int sdpb_hip_set_block_mpf(sdpb_hip_ctx *ctx, int j, int limbs64, const unsigned long long *bilinear_bases_even,
                           const unsigned long long *bilinear_bases_odd, const unsigned long long *B,
                           const unsigned long long *c);
int sdpb_hip_set_objective_mpf(sdpb_hip_ctx *ctx, int limbs64, const unsigned long long *b, const unsigned long long *constant);
/* *count receives the number of elements; nothing is written when capacity < *count. */
int sdpb_hip_get_array_mpf(sdpb_hip_ctx *ctx, const char *which, int j, int parity, int limbs64, unsigned long long *out,
                           size_t capacity, size_t *count);
int sdpb_hip_set_array_mpf(sdpb_hip_ctx *ctx, const char *which, int j, int parity, int limbs64,
                           const unsigned long long *values, size_t count);

/* Rank that owns block j (block_info.block_indices in the reference). */
int sdpb_hip_block_owner(sdpb_hip_ctx *ctx, int j);
/* 32-bit limbs of the device mantissa chosen for precision_bits. */
int sdpb_hip_limbs(sdpb_hip_ctx *ctx);
/* Fraction bits FB of the fixed-point image of the normalised P' that the exact integer
 * Q' = P'^T P' is formed from (the reference keeps El::gmp::Precision() bits,
 * compute_Q.cxx:107, Matrix_Normalizer.cxx:174-192; here, with FX = limbs - 2 rounded up to a multiple of
 * four, FB = 32 FX - 25 at --precision 400 ... 1024 (FX = 16, 24, 32, Toom-4 x Karatsuba image: 487 bits
 * at 400 ... 512, 743 at 640 ... 768, 999 at 1024), 32 FX - 17 above (Toom-4 image: 1263 bits at 1280,
 * 1519 at 1536, 2031 at 2048) and 32 (limbs-2) - 7 at 128 and 256 bits (two Karatsuba levels)): inputs of
 * sdpb_hip_op_int_syrk obey |v| < 2^FB. */
int sdpb_hip_fx_frac_bits(sdpb_hip_ctx *ctx);
/* Measurement aid (bench/profiling only, never on the solve path): average HIP-event time in
 * ms of `reps` launches of one kernel of the iteration on synthetic device-resident operands.
 * op = "syrk": the exact integer Q' = P'^T P' (k_syrk_fx / k_syrk_fx2 + k_syrk_reduce) for a
 * `a` x `b` fixed-point image of pseudo-random pieces.
 * op = "trsm" (a = b = 0): P = L^{-1} B (k_trsm_rlt_panel over all panels) with the solver's own
 * blocks and the Schur-complement factors of its last iteration; needs one iteration first. */
int sdpb_hip_bench_op(sdpb_hip_ctx *ctx, const char *op, int a, int b, int reps, double *ms);

/* Cross-GPU exchange (world_size > 1), replacing the El::mpi collectives listed in
 * SURVEY.md §2a.  The library hands DEVICE pointers it owns to these callbacks:
 *   allreduce_sum_u64: in-place SUM over ranks of count uint64 (the fixed-point image
 *     of the partial Q = P^T P; restore_and_reduce.cxx:137-212 in the reference);
 *   allgather_bytes: rank-ordered gather of `bytes` from every rank (small vectors and
 *     scalars: dy, column norms, mu, errors, lambda_min; El::mpi::AllReduce sites).
 * Return 0 on success.  With RCCL: ncclAllReduce(ncclUint64, ncclSum) / ncclAllGather. */
typedef struct
{
  int (*allreduce_sum_u64)(void *user, void *dev_ptr, size_t count);
  int (*allgather_bytes)(void *user, const void *dev_send, void *dev_recv, size_t bytes);
  void *user;
} sdpb_hip_collectives;
int sdpb_hip_set_collectives(sdpb_hip_ctx *ctx, const sdpb_hip_collectives *c);

/* The same exchange on RCCL over xGMI inside the library (one process per GPU): the
 * collectives are enqueued on the library's own stream, with no host synchronisation.
 * Rank 0 obtains an id, the host program hands those bytes to every rank (MPI_Bcast, a
 * file, torch.distributed, ...), then EVERY rank of the context's world calls
 * sdpb_hip_rccl_init (collective).  Replaces the El::mpi collectives of SDP_Solver::step
 * (restore_and_reduce.cxx:137-212, compute_search_direction.cxx:74, step_length.cxx:39,
 * compute_feasible_and_termination.cxx:66-69).  sdpb_hip_comm_name: "none" (1 rank),
 * "rccl", "callbacks" or "unset". */
#define SDPB_HIP_RCCL_ID_BYTES 128
int sdpb_hip_rccl_unique_id(char id[SDPB_HIP_RCCL_ID_BYTES]);
int sdpb_hip_rccl_init(sdpb_hip_ctx *ctx, const char id[SDPB_HIP_RCCL_ID_BYTES]);
const char *sdpb_hip_comm_name(sdpb_hip_ctx *ctx);
/* Self-test of the RCCL binding on the current device with a one-rank communicator: an
 * all-gather of `bytes` bytes and a SUM all-reduce of bytes/8 64-bit lanes through the same
 * code path the solver uses must return the data unchanged.  0 = ok. */
int sdpb_hip_rccl_selftest(size_t bytes);
/* The same transport with `world` ranks, outside any solver (collective: every rank calls it with the id rank 0
 * obtained from sdpb_hip_rccl_unique_id): rank-ordered all-gather, in-place 64-bit SUM all-reduce of `bytes`, and
 * in-place broadcasts from every root issued alternately from two streams — the three collectives and the stream
 * pattern the iteration uses in place of the El::mpi calls of restore_and_reduce.cxx:137-212 and of the distributed
 * El::Cholesky (initialize_schur_complement_solver.cxx:95-103) — each verified against the expected bytes.  A
 * launcher runs it once in a short-lived process under a timeout (sdpb_amd/rccl_preflight.py) before it hands the
 * iteration to the in-library exchange.  Returns 0 or 3 with the reason in sdpb_hip_last_error(NULL). */
int sdpb_hip_rccl_preflight(const char id[SDPB_HIP_RCCL_ID_BYTES], int rank, int world_size, size_t bytes);

/* --maxRuntime (Solver_Parameters.hxx:27): the wall-clock test of
 * compute_feasible_and_termination.cxx:51-56, taken between MaxIterationsExceeded and
 * PrimalStepTooSmall; with several ranks every rank follows rank 0's clock (":66-69").
 * The clock starts at the first sdpb_hip_iterate after sdpb_hip_init_state. */
int sdpb_hip_set_max_runtime(sdpb_hip_ctx *ctx, double seconds);
/* --maxSharedMemory (src/sdpb/SDPB_Parameters, consumed in SDP_Solver::run: run.cxx:79-181 sizes the shared-memory
 * windows of the Q stage with it, BigInt_Shared_Memory_Syrk_Context.cxx:149-215 splits the output into windows that
 * fit, bigint_syrk_blas.cxx:200-220 loops over them).  Here the stage's scratch is the partial planes of the exact
 * integer syrk (row splits x limb planes x tile-packed lower triangle of Q'): when they exceed `bytes` Q' is computed
 * in chunks of output tiles that fit -- bit-identical results, the chunks only reuse one bounded buffer.  0 = default:
 * what is free on the device when the solver is created, minus a reserve, at most 1/8 of the device.  May be called any
 * time between iterations.  (Environment override for tests and GPUs shared by several ranks: SDPB_HIP_SYRK_PART_BYTES.) */
int sdpb_hip_set_max_shared_memory(sdpb_hip_ctx *ctx, unsigned long long bytes);
/* The rank's memory plan as a JSON object: bytes per array class ("bytes": psd_state_and_scratch, bases_and_pairings,
 * schur_blocks, B, P, P_fixed_point_image, Q, syrk_partial_planes, vectors_and_small), the syrk plan ("syrk": tiles,
 * chunks, tiles_per_chunk, row_splits, rows_per_split, partial_bytes, partial_bytes_unbounded, budget_bytes and where
 * the budget comes from) and the device's free/total bytes -- what the reference prints at --verbosity 2 from
 * run.cxx:79-181 (its memory estimates per node).  Same calling convention as sdpb_hip_timers. */
int sdpb_hip_memory_plan(sdpb_hip_ctx *ctx, char *buf, size_t buflen, size_t *needed);
/* Graceful stop (run.cxx:332-355): async-signal-safe, may be called from a SIGTERM handler
 * on any rank; the next sdpb_hip_iterate of EVERY rank returns terminated with reason
 * "SIGTERM signal received" and leaves x, X, y, Y untouched for the checkpoint. */
void sdpb_hip_request_stop(sdpb_hip_ctx *ctx);
/* Stage timers synchronise the stream at every stage boundary, so they are off unless asked
 * for (the reference's timers cost only at --verbosity >= 2); also SDPB_HIP_PROFILE=1. */
int sdpb_hip_set_profiling(sdpb_hip_ctx *ctx, int on);
/* Host synchronisation points executed so far (3 per iteration: termination test,
 * corrector centering parameter, step lengths). */
long sdpb_hip_host_syncs(sdpb_hip_ctx *ctx);
/* Progress record for a watchdog: the reference's step is collective over COMM_WORLD and a
 * rank that falls out of step hangs the others inside MPI; here a second host thread may poll this while
 * sdpb_hip_iterate is running (lock-free, the only entry point with that property besides
 * sdpb_hip_request_stop).  out[0] iteration, [1] host synchronisation points passed, [2] collectives
 * enqueued on the exchange so far, [3] 64-bit FNV-1a hash of their (kind, bytes, root) sequence, [4] kind of
 * the last one (1 all-gather, 2 all-reduce, 3 broadcast), [5] its bytes, [6] its root + 1 (0: rootless),
 * [7] asynchronous error code of the transport (ncclCommGetAsyncError; 0 none).  The same hash travels in
 * the result block of every synchronisation point: ranks whose sequences differ ALL fail with code 3
 * "collective sequence mismatch" (replaces nothing in the reference — MPI offers no such check). */
int sdpb_hip_progress(sdpb_hip_ctx *ctx, unsigned long long out[8]);

/* Accumulated wall time per stage in ms as a JSON object; names follow the reference's
 * Scoped_Timer hierarchy below "run.iter_*." (SURVEY.md §5). */
int sdpb_hip_timers(sdpb_hip_ctx *ctx, char *buf, size_t buflen, size_t *needed);

/* ---- load balancing from measured block costs (the reference's block_timings file) ----
 * sdpb_hip_create_with_costs: as sdpb_hip_create, with one non-negative cost per block read from a
 * block_timings file of an earlier run (Block_Info/read_block_costs.cxx:14-59: <sdpDir>/block_timings or
 * <checkpointDir>/block_timings, one integer per line); the block -> rank plan is the same
 * longest-processing-time greedy on those costs instead of the analytic model (NULL = analytic).
 * sdpb_hip_block_timings: microseconds[num_blocks], per iteration, for the blocks THIS rank owns (0
 * elsewhere: sum over ranks, write_timing.cxx:34-68 writes rank 0's gathered column).  Like the reference
 * (compute_Q.cxx:40-53 cholesky_<j> + solve_<j>; the syrk split by block size because all blocks are
 * processed together, bigint_syrk/Readme.md:325-342) a block's cost is its own Cholesky(S_j) + its own
 * P_j = L_j^{-1} B_j + its rows' share of the Q syrk.  Blocks run batched on the GPU, so the first two are
 * MEASURED on the device: while an iteration is profiled (sdpb_hip_set_profiling) every workgroup of
 * those kernels adds its residence time (100 MHz wall clock) to its block's counter, and the stage's
 * wall time is divided among the blocks in proportion to the measured times
 * (sdpb_hip_block_clock_ticks returns the raw counters).
 * The unit is microseconds (a GPU block costs well under the reference's 1 ms resolution: in ms every
 * block of the benchmark SDP would read 0 or 1); both readers only use the ratios, and ties between
 * equally loaded ranks go to the rank holding fewer blocks, so a file full of zeros still spreads out. */
int sdpb_hip_create_with_costs(int precision_bits, int num_blocks, const int *dims, const int *num_points, int N,
                               int device_id, int rank, int world_size, const long long *block_costs, sdpb_hip_ctx **out);
int sdpb_hip_block_timings(sdpb_hip_ctx *ctx, long long *microseconds);
int sdpb_hip_block_clock_ticks(sdpb_hip_ctx *ctx, unsigned long long *cholesky_ticks, unsigned long long *solve_ticks);
int sdpb_hip_plan_blocks_with_costs(int num_blocks, const long long *block_costs, int world_size, int *owners);

/* Block -> rank plan without a context or a GPU (pure host logic). owners[j] out. */
int sdpb_hip_plan_blocks(int num_blocks, const int *dims, const int *num_points, int N, int world_size, int *owners);

/* ---- operator-level entry points used by the parity tests ------------------- */
/* Device arithmetic on one pair of numbers: op in add sub mul div sqrt. */
int sdpb_hip_op_scalar(sdpb_hip_ctx *ctx, const char *op, const char *a, const char *b, char *buf, size_t buflen,
                       size_t *needed);
/* Exact integer Q = P^T P with the fixed-point syrk kernel (the semantics of
 * BigInt_Shared_Memory_Syrk_Context::bigint_syrk_blas, bigint_syrk_blas.cxx:183-302).
 * P: rows x cols decimal integers, column-major; result: lower triangle of the
 * cols x cols product, column-major, one integer per line. */
int sdpb_hip_op_int_syrk(sdpb_hip_ctx *ctx, int rows, int cols, const char *P, char *buf, size_t buflen,
                         size_t *needed);
/* syrk_Q as an operator (compute_Q.cxx:94-132: Matrix_Normalizer column norms,
 * normalize_and_shift, the exact integer syrk, check_normalized_Q_diagonal, restore_Q) — the
 * stage the reference unit-tests in test/src/unit_tests/cases/calculate_matrix_square.test.cxx.
 * P: rows x cols decimals, column-major; result: lower triangle of Q = P^T P, cols x cols
 * column-major (upper part zero), one decimal per line.  Returns 1 with the reference's
 * "Normalized Q should have ones on diagonal" text if the check fails. */
int sdpb_hip_op_syrk_Q(sdpb_hip_ctx *ctx, int rows, int cols, const char *P, char *buf, size_t buflen, size_t *needed);
/* min_eigenvalue (step_length/min_eigenvalue.cxx:8-33: El::HermitianEig + El::Min) as an operator: the smallest eigenvalue of
 * the symmetric n x n matrix A (column-major decimals) through the kernels of the step length (Householder tridiagonalisation,
 * bisection + Newton on the shifted tridiagonal matrix).  The parity tests aim it at what whole iterations reach only late in a
 * convergent run: spectra clustered within 2^-k of one value.  Result: one decimal. */
int sdpb_hip_op_min_eigenvalue(sdpb_hip_ctx *ctx, int n, const char *A, char *buf, size_t buflen, size_t *needed);
/* Host-side fixed-point exchange image helpers (used by the world_size-2 gloo tests to
 * check the u64-lane reduction without a GPU): encode a two's-complement integer given
 * in decimal into `planes` 32-bit limbs widened to uint64 lanes; decode after a lane-wise
 * sum with carry propagation. */
int sdpb_hip_host_encode_u64(const char *decimal_integer, int planes, unsigned long long *lanes);
int sdpb_hip_host_decode_u64(const unsigned long long *lanes, int planes, char *buf, size_t buflen, size_t *needed);

/* Measured HBM ceiling of the device the caller has selected: a streaming copy of `bytes`
 * (read + write counted) repeated `reps` times by a plain 16-byte-per-lane kernel; the
 * best repetition in GB/s.  bench.py reports it next to the 8 TB/s data-sheet peak
 * (SURVEY.md section 8d asks for a measured copy ceiling). */
int sdpb_hip_copy_bandwidth(size_t bytes, int reps, double *gb_per_s);

#ifdef __cplusplus
}
#endif
#endif /* SDPB_HIP_H */
