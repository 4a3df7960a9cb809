// rdyn_kernels.h -- launch interface between the C-ABI layer (rdyn_api.cpp) and the HIP kernels.
#ifndef RDYN_KERNELS_H
#define RDYN_KERNELS_H

#include <hip/hip_runtime.h>
#include "rdyn_device.h"

// All strides are in doubles.  Inputs: x(s, k) = x[s * in_ss + k * in_sj].
struct RdynSweepArgs
{
  const RdynChainConst* chain;  // device pointer
  const double *q, *dq, *ddq;   // dq / ddq may be null (treated as zero)
  int64_t n_samples;
  int64_t in_ss, in_sj;
  double* tau;                  // may be null in regressor mode
  int64_t tau_ss, tau_sj;
  double* Y;
  int64_t y_ss, y_sr, y_sc;
  double* M;
  int64_t m_ss, m_se;
  // regressor mode only: optional measured torque, copied into regressor column P (the "b" column of the
  // normal equations) with the Y addressing; same layout as q.
  const double* bcol;
  int bcol_col;                 // column that receives bcol; 0 = column P (right after the regressor)
  // torque mode only: optional external wrenches, links x 6 per sample ([force; torque] in the link's own frame,
  // applied TO the link; getWrench, primitives_impl.h:1225); element e of sample s at ext[s * ext_ss + e * ext_se]
  const double* ext;
  int64_t ext_ss, ext_se;
  // RDYN_MODE_REGRESSOR_EXPAND only (a chain longer than what the kernels sweep: `chain` is its reduced companion): the ten columns of
  // chain link g are the columns of reduced link expand_red_of[g] times the constant 10 x 10 block X_g (rdyn_chain.hpp), zero for links
  // upstream of the first input joint (expand_red_of[g] < 0).  expand_X: device, [expand_n][10][10] row-major X_g(a, p).
  const double* expand_X;
  int expand_n;
  int expand_red_of[RDYN_MAX_JOINTS];  // (ints: a dynamically indexed byte of the kernel arguments is a VECTOR load, whose wait drains every store in flight)
  // RDYN_MODE_REGRESSOR_EXPAND_STAGED: the row-contiguous layout the wave's LDS tile is copied out to -- 1: per-sample images
  // (stride_col == rows, a link's block is one run of 10 rows doubles per sample), 2: stacked matrix (stride_sample == rows, a column is
  // one run of 64 rows doubles per wave)
  int expand_stage;
  // per-sample images through a run-time row map (k_image_sweep<.., MAP>: input joints in any order -- `chain` is then the sorted view,
  // rdyn_chain.hpp -- and joints that are not input joints anywhere in the chain): row_map[f] = the caller's input index of chain joint
  // f -- where q / Dq / DDq are read, tau is written, and the row of the image its values land in --, -1 for a joint that is none
  int row_map[RDYN_MAX_SWEPT_JOINTS];
  // k_image_sweep<.., EXPAND>: the links of the full chain that ride on body f are expand_first[f] .. expand_first[f + 1] - 1 (chain order;
  // the links below expand_first[0] sit upstream of the first input joint: zero blocks)
  int expand_first[RDYN_MAX_SWEPT_JOINTS + 1];
  // torque / inertia launches: the sample-major records (tau: n, M: n x n doubles per sample) through the wave's LDS tile, written in whole
  // lines (rdyn_record_stage.h); decided by the host: natural strides, 128-byte aligned output
  int staged;
  // k_image_sweep: chunk permutation of the workgroups (0 = none), see rdyn_image_impl.h
  unsigned blk_mul;
};

// split / jerk sweeps (rdyn_kin_ext.hip); every output record is links x 6
struct RdynKinExtArgs
{
  const RdynChainConst* chain;
  const double *q, *dq, *ddq, *dddq;
  int64_t n_samples, in_ss, in_sj;
  int64_t out_ss, out_se;
  double* dtw_lin;
  double* dtw_nonlin;
  double* ddtw;
  double* ddtw_lin;     // getDDTwistLinearPart
  double* ddtw_nonlin;  // getDDTwistNonLinearPart
  double* wrench;       // getWrench: base-frame link wrenches
  const double* ext;    // wrench only, may be null: external wrenches, links x 6, element e of sample s at ext[s * ext_ss + e * ext_se]
  int64_t ext_ss, ext_se;
  // rdyn_long_kin.hip only (chains of more than RDYN_MAX_SWEPT_JOINTS joints): the constants and the joint torques of the wrench pass
  // (getJointTorque with external wrenches; layout of q)
  const RdynLongChainConst* chain_long;
  double* tau;
  int64_t tau_ss, tau_sj;
  int staged;  // sample-major records through wave-private LDS, whole lines (rdyn_record_stage.h); decided by the host: natural strides, line-aligned outputs
  int ext_staged;  // ... and the external wrenches read as whole lines into the same tile (16-byte aligned, natural stride)
};
hipError_t rdyn_launch_base_ext(int n_joints, const RdynKinExtArgs& a, hipStream_t st);
hipError_t rdyn_launch_long_ext(int n_joints, const RdynKinExtArgs& a, hipStream_t st);  // a.chain_long; any chain length

// batched local inverse kinematics (rdyn_ik.hip)
struct RdynIkArgs
{
  const RdynChainConst* chain;
  const double* T_target;            // 12 per pose, column-major 3x4 [R | p]; element e of pose s at [s * tt_ss + e * tt_se]
  int64_t tt_ss, tt_se;
  const double* seed;                // n_active per pose, x(s, k) = seed[s * in_ss + k * in_sj]
  double* sol;                       // same addressing as seed (may alias it)
  int64_t n_samples, in_ss, in_sj;
  double weight[6];                  // all 1 for computeLocalIk
  double q_min[RDYN_MAX_SWEPT_JOINTS];     // per CHAIN joint
  double q_max[RDYN_MAX_SWEPT_JOINTS];
  // a constant frame behind the last swept joint (a chain served through its reduced companion: the fixed frames after the last
  // input joint): T_tool = T_last [tail_R | tail_t]; tail_R row-major
  int has_tail;
  double tail_R[9], tail_t[3];
  double toll;
  double damping;                    // Levenberg term: damping^2 is added to the diagonal of J'WJ (0 = the reference's QP)
  int max_iter;
  int it_stage;                      // > 0: resume launch -- continue the poses a first launch left unfinished after it_stage updates
  int* status;                       // per pose, may be null: 1 converged, 0 not within max_iter, < 0 QP failure
  int* iterations;                   // per pose, may be null: QP updates performed
};
hipError_t rdyn_launch_local_ik(int n_joints, const RdynIkArgs& a, hipStream_t st);

// getFrameDistance family (frame_distance.h) on pairs of frames; element e of record s at base[s * X_ss + e * X_se]
struct RdynFrameDistanceArgs
{
  const double *T_wa, *T_wb;   // 12 per frame, column-major 3x4 [R | p]
  int64_t n, t_ss, t_se;
  int kind;                    // 0 getFrameDistance, 1 getFrameDistanceQuat, 2 getFrameDistanceQuatJac
  double* distance;            // 6 per pair
  int64_t d_ss, d_se;
  double* jacobian;            // kind 2 only, may be null: 36 per pair, column-major 6 x 6
  int64_t j_ss, j_se;
};
hipError_t rdyn_launch_frame_distance(const RdynFrameDistanceArgs& a, hipStream_t st);

// Base-frame kinematics outputs; record element e of sample s at out[s * X_ss + e * out_se].
struct RdynKinArgs
{
  const RdynChainConst* chain;
  const double *q, *dq, *ddq;
  int64_t n_samples;
  int64_t in_ss, in_sj;
  int64_t out_se;
  double* T_bt;    int64_t tb_ss;   // 12 per sample
  double* T_links; int64_t tl_ss;   // 12 * links per sample
  double* J;       int64_t j_ss;    // 6 * n_active per sample
  int j_link;                       // Jacobian reference link (chain link index); n_joints = the tool (getJacobian)
  double* twists;                   // 6 * links per sample
  double* dtwists; int64_t tw_ss;
  // rdyn_long_kin.hip only (chains of more than RDYN_MAX_SWEPT_JOINTS joints): the constants; j_up = input joints upstream of link
  // j_link (the first j_up input columns of the Jacobian are filled, primitives_impl.h:965-972)
  const RdynLongChainConst* chain_long;
  int j_up;
  // sample-major records through wave-private LDS, written in whole lines (rdyn_record_stage.h): natural record strides (out_se == 1,
  // X_ss = the record length) and every output pointer 128-byte aligned -- decided by the host; n_active sizes the Jacobian's tile
  int staged;
  int n_active;
};
hipError_t rdyn_launch_long_base(const RdynKinArgs& a, hipStream_t st);

// regressor (+ fused torque) / joint inertia of a chain with more input joints than the unrolled kernels sweep (rdyn_long_local.hip)
struct RdynLongLocalArgs
{
  const RdynLongChainConst* chain_long;
  const double *q, *dq, *ddq;   // dq / ddq may be null (zero)
  int64_t n_samples, in_ss, in_sj;
  double* tau;                  // regressor mode, may be null
  int64_t tau_ss, tau_sj;
  double* Y;
  int64_t y_ss, y_sr, y_sc;
  double* M;
  int64_t m_ss, m_se;
  // regressor mode, may be null: the measured torque (layout of q), copied into regressor column bcol_col with the Y addressing -- the
  // "b" column of the chunk images rdyn_regressor_gram hands to k_gram
  const double* bcol;
  int bcol_col;
  // regressor mode: 1 = per-sample images (stride_row 1, stride_col n_active), 2 = the stacked matrix (stride_sample n_active) -- a link's
  // block through a wave-private LDS tile two columns at a time, copied out 16 bytes per lane; 0 = stores from the computing lane
  int stage;
  int n_active;
};
size_t rdyn_long_local_lds_bytes(int mode, int n_joints);
hipError_t rdyn_launch_long_local(int mode, int n_joints, const RdynLongLocalArgs& a, hipStream_t st);  // mode: RDYN_MODE_REGRESSOR / RDYN_MODE_INERTIA

// Gram / normal equations of a column-major rows x P matrix (rdyn_gram.hip)
struct RdynGramArgs
{
  const double* A;
  const double* b;     // may be null
  int64_t rows, lda;
  int P;
  int accumulate;      // slabs += instead of slabs =
  int add_to_output;   // finish: G/c/bb += instead of =
  double* slabs;       // [blocks][NT * 256]
  double* G;
  double* c;
  double* bb;
  // Block structure of the element-major regressor image: rows [j * row_block, (j + 1) * row_block) belong to
  // input joint j and are structurally zero in columns < first_col[j] (primitives_impl.h:690-691, 1341-1347).
  // row_block == 0: no structure assumed.  Column blocks (16 wide) left of first_col[j] are neither loaded nor multiplied.
  int64_t row_block;
  int first_col[RDYN_MAX_SWEPT_JOINTS];
  // finish only: > 0 = the slabs hold the columns in the wave-pair kernel's order [tau_meas | link desc_nj - 1 | ... | link 0]
  // (rdyn_duo_gram.hip: the zero band of every row group then ends at a 16-column boundary more often); 0 = natural order
  int desc_nj;
  int desc_k;   // finish only, with desc_nj > 0: component columns in front of that order ([C | tau_meas | links descending]; P counts them)
  int group_stride;  // k_gram only: every group_stride-th 16-row group (0 / 1 = all): the subsample pass of the preconditioned route
  int slab_nb;  // finish only: 16-column blocks of the slabs' tile layout if it is wider than P + 1 columns need (0 = derive from P)
  const int* run_flag;  // finish only, may be null: device word; 0 = leave at once (conditional second round of rdyn_cholqr.hip)
  int col_shift;        // finish only: the slabs' column space is the natural order shifted right by col_shift (rdyn_cholqr.hip)
};
// normal equations of the reduced chain -> of the chain (rdyn_chain.hpp; rdyn_gram.hip: k_gram_expand)
struct RdynGramExpandArgs
{
  const double *G_red, *c_red, *bb_red;  // (10 n_red + K)^2, 10 n_red + K, 1
  const double* X;                       // device: [n_joints][10][10]
  int red_of[RDYN_MAX_JOINTS];   // per CHAIN joint (the chain may be longer than what the kernels sweep)
  int n_joints, n_red, n_comp_cols;
  int add_to_output;
  double *G, *c, *bb;                    // (10 n_joints + K)^2, 10 n_joints + K, 1 (c, bb may be null)
};
hipError_t rdyn_launch_gram_expand(const RdynGramExpandArgs& a, hipStream_t st);
hipError_t rdyn_launch_set_double(double* p, double v, hipStream_t st);  // *p = v, ordered on the stream
// fused regressor -> Gram persistent kernel (rdyn_fused_gram.hip)
struct RdynFusedGramArgs
{
  RdynSweepArgs sweep;   // chain, q/dq/ddq/bcol, n_samples, input strides (Y fields ignored)
  int n_active;
  int first_col[RDYN_MAX_SWEPT_JOINTS];
  double* images;        // [blocks][(P + 1) * n * 256] per-workgroup tile images
  double* slabs;         // [blocks][NT * 256] per-workgroup Gram slabs
  int debug;             // timing experiments only (RDYN_FUSED_DEBUG): bit 0 skip phase 1 after the first tile, bit 1 skip phase 2
};
hipError_t rdyn_launch_regressor_gram_fused(int n_joints, const RdynFusedGramArgs& a, int blocks, hipStream_t st);

// per-joint additive components (rdyn_components.hip); constants already sanitised by the API
#define RDYN_MAX_COMPONENTS 30
struct RdynComponent
{
  int type, joint;
  double min_velocity, max_velocity;
  double parameters[3];
};
// LDS-resident regressor -> Gram kernel (rdyn_lds_gram.hip): 16 samples per wave, packed column-major tile in LDS
struct RdynLdsGramArgs
{
  const RdynChainConst* chain;
  const double *q, *dq, *ddq, *bcol;   // bcol may be null
  int64_t n_samples, in_ss, in_sj;
  int n_active;
  // the caller's input index behind tile row r (rdyn_chain.hpp: sorted view): q, Dq, DDq, tau_meas of row r are read at in_map[r] * in_sj;
  // the identity for input joints in chain order
  int in_map[8];
  // one-lane-per-sample sweepers (rdyn_duo_gram.hip, KIN): sweep_lanes != 0 selects them (the host then sized the dynamic LDS for the
  // exchange area behind the tiles, RDYN_KIN_XCH_BYTES); sw_rows[3 (w - 1) + slot] = the tile rows sweeper wave w = 1 .. 7 computes (99 =
  // none), balanced over the waves by the cost of a row (a row of joint l is carried through the links l .. n - 1)
  int sweep_lanes;
  int sw_rows[21];
  int all_revolute;                    // every chain joint is revolute (selects the sweeper without joint-kind selects)
  int first_col[RDYN_MAX_SWEPT_JOINTS];      // per input joint: 10 * chain index
  int lds_off[RDYN_MAX_SWEPT_JOINTS];        // per link: byte offset of its first column in the tile
  int lds_stride[RDYN_MAX_SWEPT_JOINTS];     // per link: bytes between its columns = (16 m_f + 4) * 8
  int lds_m[RDYN_MAX_SWEPT_JOINTS];          // per link: number of input joints whose rows can be non-zero (stored rows = 16 m_f)
  int lds_off_b;                       // byte offset of column P (measured torque), 16 n rows
  int lds_dummy_off;                   // pipelined kernel only: 64 x 8 bytes where lanes drop rows a link does not store
  int tile_bytes;                      // one wave's tile
  double* slabs;
  int debug;                           // timing experiments only (RDYN_FUSED_DEBUG): bit 0 sweep only the first tile, bit 1 no Gram phase
  int tile_stride;                     // wave-pair kernels: sweep every tile_stride-th 16-sample tile only (0 / 1 = all): the subsample pass of the preconditioned route
  const int* run_flag;                 // k_regressor_tsqr and its tree only: null, or a device word -- 0 = leave at once
  // wave-pair kernel only (rdyn_duo_gram.hip): the per-joint component columns [Y | C | tau_meas] of rdyn_identification_gram.
  // Column P + k of the tile belongs to component comp_col_comp[k] and is non-zero only in the rows of that component's joint:
  // it is stored as ONE 16-row group (stride 160 bytes) at lds_off_c + 160 k; the measured torque moves to column P + n_comp_cols.
  int n_comps, n_comp_cols;
  int lds_off_c, comp_stride;          // component column k at lds_off_c + comp_stride * k (160, or 144 in the compact layout)
  int col_shift;                       // k_regressor_pgram_solo only: columns of padding in front of the natural order in the consumer's column space
  int comp_row_step;                   // 0: a component column stores its own joint's 16-row group only; 128: all row groups (rectangular tile of rdyn_tsqr_wide.hip)
  signed char comp_col_row[96];        // per component column: the input joint (row group) it belongs to
  RdynComponent comps[RDYN_MAX_COMPONENTS];
};
hipError_t rdyn_launch_regressor_gram_lds(int n_cols, const RdynLdsGramArgs& a, int blocks, size_t lds_bytes, hipStream_t st);
// the same kernel software-pipelined inside the wave (rdyn_pipe_gram.hip): chains of 2..6 joints
bool rdyn_regressor_gram_pipe_supported(int n_cols);
hipError_t rdyn_launch_regressor_gram_pipe(int n_cols, const RdynLdsGramArgs& a, int blocks, size_t lds_bytes, hipStream_t st);
// the two streams on two co-resident waves (rdyn_duo_gram.hip): chains of 2..7 joints, 512-thread workgroups
bool rdyn_regressor_gram_duo_supported(int n_cols);
// doubles per sample and exchange buffer: up to 6 joints 30 (the b-matrix too; the kernel uses 21 of them without component columns), 12 at 7
#define RDYN_KIN_XCH_BYTES(n_joints) (2 * ((n_joints) <= 6 ? 30 : 12) * 64 * 8)
#define RDYN_KIN_XCH_BYTES_XV(xv) (2 * (xv) * 64 * 8)  // (pass B of the R factor: 21 doubles per sample up to 6 joints, 12 at 7)
// the one-lane-per-sample sweepers exist for this shape: 0 no, 4 / 2 = with the standard / the compact tile layout (column padding in doubles)
int rdyn_regressor_gram_duo_kin_pad(int n_joints, int n_comp_cols);
// n_cols = 10 * chain joints; a.n_comp_cols extra component columns may add at most one 16-column block
bool rdyn_regressor_gram_duo_supports_components(int n_cols, int n_comp_cols);
hipError_t rdyn_launch_regressor_gram_duo(int n_cols, const RdynLdsGramArgs& a, int blocks, size_t lds_bytes, hipStream_t st);
// preconditioned CholeskyQR (rdyn_cholqr.hip): R factor of [A | b] with the heavy pass on the fp64 matrix cores.
//   W = R1^-1 of a Householder factor R1 of a row SUBSAMPLE (rdyn_tsqr.hip);  G2 = (A W)'(A W) over ALL rows: sweep -> LDS tile -> the
//   consumer wave multiplies every 16-row group by W (MFMA) and accumulates the Gram of the product (MFMA);  R = chol(G2) R1.
// xb = 1: one more 16-column block for the component columns of rdyn_identification_tsqr (chains of <= 6 joints)
int rdyn_cholqr_kin_pad(int n_joints, int xb, int pairs);  // pass B with the one-lane-per-sample sweepers: tile padding (4 / 2), 0 = not served
int rdyn_cholqr_pairs(int n_joints, int tile_bytes, int xb);  // 4: W in LDS beside four tiles; -4: four tiles, W in global memory; 2: two pairs on four SIMDs, -1: four waves that sweep and consume their own compact tile (7 joints + components); 0: unsupported
size_t rdyn_cholqr_w_doubles(int n_joints, int xb);           // W in MFMA operand order
// R <- qr([R ; R_new]); rows_new > 0: R_new is zero below that many rows (an expanded factor); any n1 whose packed R_new fits 156 KB of LDS
hipError_t rdyn_launch_cholqr_fold(const double* R_new, double* R, int n1, hipStream_t st, int rows_new = 0);
hipError_t rdyn_launch_regressor_pgram(int n_joints, const RdynLdsGramArgs& a, const double* W, const int* run_flag, int blocks, int pairs, hipStream_t st);
// the same pass over a materialised column-major matrix [A | b] (rdyn_tsqr): n_cols + (b != null) <= 96 columns, natural column order
hipError_t rdyn_launch_pgram_rows(const double* A, const double* b, int64_t rows, int64_t lda, int n_cols, const double* W, double* slabs,
                                  const int* run_flag, int blocks, hipStream_t st);
// R1 (n1 x n1 upper, column-major) -> T = R1 re-triangularised without its deferred columns (zmask <- that set), W = T^-1 in MFMA
// operand order.  row_scale: R1 is the factor of one row in row_scale^2 (the subsample), T is scaled to all rows.
// col_shift: columns of padding in FRONT of the natural order in the consumer's column space (rdyn_cholqr_col_shift).
// flags (device ints): [0] run round 1, [1] run the stand-by Householder factorisation, [2] run round 0.  The preconditioner of a
// round whose growth factor gamma (Q T reproduces the columns of A with a relative error of about u gamma; *gamma_out, may be null)
// exceeds 1e4 calls the round off and the stand-by in.
// V: T^-1 in natural order (n1 x n1, for the factor kernel's own evaluation of gamma).
// Input: the triangular factor R1, or (Gs != null) the Gram matrix [Gs cs; cs' bbs] of the subsample's rows (P x P, P, 1; P = n1 - 1).
// nb_w: 16-column blocks of the consumer's column space (W is written, zero-padded, for all nb_w (nb_w + 1) / 2 operand blocks)
hipError_t rdyn_launch_cholqr_precond(const double* R1, const double* Gs, const double* cs, const double* bbs, int n1, int col_shift, int nb_w,
                                      double row_scale, double* T, double* W, double* V, int* zmask, int* flags, int round, const int* run_flag,
                                      double* gamma_out, hipStream_t st, double* wide_sq = nullptr);  // wide_sq: 2 n1^2 doubles of workspace for n1 > rdyn_cholqr_max_cols_lds()
int rdyn_cholqr_max_cols();      // widest factor (right-hand side included) of the dense steps of the preconditioned route (112)
int rdyn_cholqr_max_cols_lds();  // ... with their two squares in LDS (96): what the fused regressor routes take
int rdyn_cholqr_col_shift(int n_joints, int xb);
hipError_t rdyn_launch_regressor_pgram_solo(const RdynLdsGramArgs& a, const double* W, const int* run_flag, int blocks, int pairs, hipStream_t st);  // rdyn_pgram_solo.hip
int rdyn_cholqr_solo_col_shift(int n_joints, int n_comp_cols);  // pairs == -1 at 7 joints + components (k_regressor_pgram_solo: compact tile, every wave sweeps and consumes)
// G2 = [G c; c' bb] -> R = chol(G2) T (n1 x n1 upper, column-major; zero rows at the confirmed null columns); flags[round] = 1 when the
// round is not accepted (rho_out, may be null: [0] the conditioning measure of the equilibrated Q, [2] gamma on the norms of all rows);
// round 0 clears flags[1]
hipError_t rdyn_launch_cholqr_factor(const double* G, const double* c, const double* bb, int n1, int has_b, const double* T, const double* V, const int* zmask,
                                     double* R, int* flags, int round, const int* run_flag, double* rho_out, hipStream_t st, double* wide_sq = nullptr);
// factor of the reduced chain -> factor of the chain: R = qr(R_red diag(E, I_K, 1)) (a.X, a.red_of, a.n_joints, a.n_red, a.n_comp_cols used)
hipError_t rdyn_launch_cholqr_expand(const RdynGramExpandArgs& a, const double* R_red, double* R, hipStream_t st);
size_t rdyn_cholqr_expand_lds_bytes(int n_joints, int n_red, int n_comp_cols);  // dynamic LDS of that launch (limit: 156 KB)
// tall-skinny QR (rdyn_tsqr.hip): R factor of [A | b] without forming A'A
int rdyn_tsqr_padded_cols(int n_cols_with_rhs);              // 16 / 32 / 48 / 64, 0 = unsupported
size_t rdyn_tsqr_workspace_doubles(int nc, int blocks);
int rdyn_regressor_tsqr_cols(int n_joints, int n_comp_cols);  // factor width of a rdyn_launch_regressor_tsqr call (0: unsupported)
// a.run_flag (device int, may be null): every kernel of the call leaves at once when it reads 0; tree_fan: factors folded per wave and
// tree level (2: the shallowest dependent chain; 16: three launches in all -- the stand-by call of the preconditioned route)
hipError_t rdyn_launch_regressor_tsqr(int n_joints, const RdynLdsGramArgs& a, int blocks, size_t lds_bytes, double* workspace, double* R, int accumulate,
                                      hipStream_t st, int tree_fan = 2);
hipError_t rdyn_launch_tsqr_rows(const double* A, const double* b, int64_t rows, int64_t lda, int n_cols, int blocks, double* workspace, double* R,
                                 int accumulate, hipStream_t st, const int* run_flag = nullptr, int tree_fan = 2);
// the shapes beyond the register-resident folds (rdyn_tsqr_wide.hip): factor packed in LDS, up to 112 columns
int rdyn_tsqr_wide_max_cols();
size_t rdyn_regressor_tsqr_wide_lds_bytes(int n1, int n_active);  // 0: the 16-sample tile does not fit beside the factor
size_t rdyn_tsqr_wide_workspace_doubles(int n1, int blocks);
// a: the rectangular tile layout (comp_row_step = 128); factor width 10 n_joints + a.n_comp_cols + 1; blocks <= 256; a.run_flag as above
hipError_t rdyn_launch_regressor_tsqr_wide(int n_joints, const RdynLdsGramArgs& a, int blocks, double* workspace, double* R, int accumulate, hipStream_t st);
hipError_t rdyn_launch_tsqr_wide_rows(const double* A, const double* b, int64_t rows, int64_t lda, int n_cols, int blocks, double* workspace, double* R,
                                      int accumulate, const int* run_flag, hipStream_t st);
hipError_t rdyn_launch_tsqr_fold_factors(const double* factors, int count, int64_t stride, int n1, double* scratch, double* R, int accumulate, hipStream_t st);
int rdyn_gram_blocks_for(int P);
hipError_t rdyn_launch_gram(const RdynGramArgs& a, int blocks, hipStream_t st);
hipError_t rdyn_launch_gram_finish(const RdynGramArgs& a, int blocks, hipStream_t st);

struct RdynComponentArgs
{
  const double *q, *dq;
  int64_t n_samples, in_ss, in_sj;
  int n_active, n_comps;
  double* C;            // may be null
  int64_t c_ss, c_sr, c_sc;
  double* tau;          // may be null; += component torque, same layout as q
  RdynComponent comps[RDYN_MAX_COMPONENTS];
};
hipError_t rdyn_launch_components(const RdynComponentArgs& a, hipStream_t st);

// pieces of rdyn_identification_tsqr for the multi-device form (rdyn_api.cpp): widths of the swept / the chain's factor, the factor of
// the swept chain alone, the expansion (+ accumulation) of a swept factor
int rdyn_internal_tsqr_widths(const rdyn_chain* c, const rdyn_component* comps, int n_comps, int* n1s, int* n1, int* expands);
int rdyn_internal_tsqr_swept(const rdyn_chain* c, const rdyn_component* comps, int n_comps, const rdyn_batch* b, const double* tau_meas, double* R_swept,
                             void* workspace, size_t workspace_bytes);
int rdyn_internal_tsqr_expand(const rdyn_chain* c, const rdyn_component* comps, int n_comps, const double* R_swept, double* R, int accumulate,
                              double* scratch, void* stream);
// y[i] += x[i], i < n, ordered on the stream (rdyn_gram.hip)
hipError_t rdyn_launch_add_doubles(double* y, const double* x, int64_t n, hipStream_t st);

// every getter of a sample in one launch (rdyn_kernels.hip: k_sample_all): the argument blocks of the single-purpose kernels, one per role;
// a role whose outputs are all null leaves at once
struct RdynAllArgs
{
  int64_t n_samples;
  RdynKinArgs frames, jacobian, twists;
  RdynSweepArgs torque, torque_nl, inertia, regressor;
};
hipError_t rdyn_launch_sample_all(int n_joints, const RdynAllArgs& a, hipStream_t st);

enum { RDYN_MODE_REGRESSOR = 0, RDYN_MODE_TORQUE = 1, RDYN_MODE_INERTIA = 2, RDYN_MODE_REGRESSOR_GRAM = 3, RDYN_MODE_REGRESSOR_EXPAND = 4,
       RDYN_MODE_REGRESSOR_EXPAND_STAGED = 5 };

hipError_t rdyn_launch_local_sweep(int n_joints, int mode, const RdynSweepArgs& a, hipStream_t st);
hipError_t rdyn_launch_base_sweep(int n_joints, const RdynKinArgs& a, hipStream_t st);
hipError_t rdyn_launch_rowpair_sweep(int n_joints, int n_active, const RdynSweepArgs& a, hipStream_t st);
// per-sample Eigen image (y_sr == 1, y_sc == n_active) and stacked matrix (y_ss == n_active): one thread per sample, link blocks staged
// through LDS and written in whole lines (rdyn_image.hip / rdyn_image_impl.h).  fix_mask: bit f set = chain joint f is not an input
// joint; the input joints are the others, in chain order.  Compiled patterns: <= 1 fixed head joint, <= 3 fixed tail joints.
bool rdyn_image_supported(int n_joints, unsigned fix_mask, int64_t y_ss, bool multi);
hipError_t rdyn_launch_image_sweep(int n_joints, unsigned fix_mask, const RdynSweepArgs& a, hipStream_t st, int mapped = 0);  // mapped: 1 = run-time row map, 2 = + expansion to a longer chain
bool rdyn_image_expand_supported(int n_red, int n_full, int64_t y_ss);
bool rdyn_image_map_supported(int n_joints, unsigned fix_mask, int64_t y_ss);  // per-sample images through a run-time row map (any input order, fixed joints anywhere)
hipError_t rdyn_launch_image_sweep_multi(int n_joints, unsigned fix_mask, bool stacked, const RdynSweepArgs* table, int n_items, int64_t max_samples,
                                         hipStream_t st);  // 2..8 input joints
hipError_t rdyn_launch_local_sweep_multi(int n_joints, int mode, const RdynSweepArgs* table, int n_items, int64_t max_samples, hipStream_t st);

#endif
