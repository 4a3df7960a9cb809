/*
 * rdyn.h -- C-ABI of librdyn_hip.so: batched rigid-body dynamics of a serial chain on AMD MI355X (gfx950).
 *
 * Drop-in boundary for ONE path of CNR-STIIMA-IRAS/rosdyn: the public surface of `class rosdyn::Chain`
 * (rosdyn_core/include/rosdyn_core/primitives.h:235-555) restricted to
 *   getTransformation(s) / getJacobian / getTwist / getDTwist / getJointTorque /
 *   getJointTorqueNonLinearPart / getRegressor / getJointInertia / getNominalParameters,
 * evaluated for N samples (q, Dq, DDq) per call instead of one.  The reference has no FFI layer; a
 * maintainer binds these symbols from `rosdyn::Chain` (see INTEGRATION.md, and the ready-made C++
 * facade rosdyn_amd/csrc/rosdyn_chain_facade.hpp which keeps the reference's method names).
 *
 * Contract = the STATELESS function (chain, q, Dq, DDq) -> outputs that a freshly constructed reference
 * Chain computes on its first call (the reference's value caches, primitives_impl.h:886/985/1088, are
 * not reproduced).  All arithmetic is IEEE fp64.  Conventions are the reference's: spatial vectors are
 * [linear; angular] (spacevect_algebra.h:44-52); twists/Jacobians are expressed in the base frame with
 * the link's own origin as reference point; the regressor has 10 columns per chain joint INCLUDING fixed
 * joints, parameter order [m, m cx, m cy, m cz, Ixx, Ixy, Ixz, Iyy, Iyz, Izz] with the inertia taken
 * about the link origin (primitives_impl.h:399-417, 1295-1355); rows exist only for the active (input)
 * joints, in input order (primitives_impl.h:1352).
 *
 * Memory: every `const double*` / `double*` in a batched call is a DEVICE pointer (hipMalloc or a
 * torch.cuda tensor's data_ptr()).  Nothing is copied to or from the host, no allocation and no
 * synchronisation happens inside a batched call (graph-capturable) except the one-time upload of a
 * chain's constants (~4 KB) the first time a chain is used on a device.
 *
 * Error handling: every function returns an rdyn_status; rdyn_last_error() returns the thread-local
 * message.  Where the reference throws (std::runtime_error("Base link not found") primitives_impl.h:603,
 * "Tool link not found" :610, std::invalid_argument("Input data dimensions mismatch") :1302) the same
 * text is the message and the facade re-throws the same exception types.
 */
#ifndef RDYN_H
#define RDYN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RDYN_VERSION 100
/* Chain joints INCLUDING fixed ones (reference: m_joints_number, primitives_impl.h:638).  The reference's default build is
 * unbounded (rosdyn_core/CMakeLists.txt:12-16: MAX_NUM_AXES = -1); here a chain may have up to RDYN_MAX_JOINTS joints.
 *  - Chains of up to RDYN_MAX_SWEPT_JOINTS joints: kernels instantiated per joint count, every per-link quantity in registers.
 *  - Longer chains (long because of FIXED frames as a rule: a tool changer, a camera mount, the flange / tool0 frames of the public
 *    UR and Panda models), input joints listed in ANY order (setInputJointsName, primitives_impl.h:705-737):
 *      the by-link kinematic outputs -- rdyn_transformation, rdyn_jacobian(_link), rdyn_twist, rdyn_twist_parts, rdyn_jerk_parts,
 *      rdyn_wrench, rdyn_joint_torque_ext -- run on kernels with a run-time link loop (rdyn_long_kin.hip), any number of input joints;
 *      rdyn_regressor, rdyn_joint_torque(_nonlinear), rdyn_joint_inertia, rdyn_nominal_parameters, the normal equations, the R factors
 *      and rdyn_local_ik sweep the REDUCED COMPANION -- the input joints (at most RDYN_MAX_SWEPT_JOINTS) with the fixed frames folded
 *      into the neighbouring bodies, rdyn_chain_reduction -- and restore the columns of every folded link exactly (Y_f = Y_body X_f);
 *      with MORE input joints than that (up to RDYN_MAX_JOINTS of them; rdyn_long_local.hip: rolled link and row loops, the per-joint
 *      state in wave-private LDS) rdyn_regressor (+ its fused torque), rdyn_joint_inertia, the joint torques (read off the wrench
 *      recursion) and every kinematic output are served, rdyn_regressor_gram and rdyn_regressor_tsqr for 11 input joints (110 + 1 columns:
 *      what the Gram kernel and the widest R factor hold; chunk images); rdyn_local_ik, component columns and wider normal equations /
 *      factors answer RDYN_ERR_UNSUPPORTED. */
#define RDYN_MAX_JOINTS 32
#define RDYN_MAX_SWEPT_JOINTS 10

typedef enum rdyn_status
{
  RDYN_OK = 0,
  RDYN_ERR_INVALID_ARGUMENT = 1, /* NULL pointers, negative sizes, dimension mismatch                    */
  RDYN_ERR_BASE_NOT_FOUND = 2,   /* "Base link not found"   primitives_impl.h:603                        */
  RDYN_ERR_TOOL_NOT_FOUND = 3,   /* "Tool link not found"   primitives_impl.h:610                        */
  RDYN_ERR_URDF = 4,             /* malformed URDF XML                                                   */
  RDYN_ERR_UNSUPPORTED = 5,      /* more than RDYN_MAX_JOINTS chain joints, or a shape an entry point does not serve */
  RDYN_ERR_JOINT_NOT_FOUND = 6,  /* setInputJointsName: "Joint named '%s' not found" primitives_impl.h:734 */
  RDYN_ERR_NO_DEVICE = 7,        /* no HIP device / HIP runtime failure at start-up                      */
  RDYN_ERR_HIP = 8               /* a HIP call failed (message carries hipGetErrorString)                */
} rdyn_status;

/* rosdyn::Joint::Type, primitives.h:67 (same numeric values) */
typedef enum rdyn_joint_type { RDYN_REVOLUTE = 0, RDYN_PRISMATIC = 1, RDYN_FIXED = 2 } rdyn_joint_type;

/* urdf::Joint::type values as urdfdom defines them; mapped by primitives_impl.h:74-83:
 * revolute|continuous -> REVOLUTE, prismatic -> PRISMATIC, everything else -> FIXED */
typedef enum rdyn_urdf_joint_type
{
  RDYN_URDF_UNKNOWN = 0, RDYN_URDF_REVOLUTE = 1, RDYN_URDF_CONTINUOUS = 2, RDYN_URDF_PRISMATIC = 3,
  RDYN_URDF_FLOATING = 4, RDYN_URDF_PLANAR = 5, RDYN_URDF_FIXED = 6
} rdyn_urdf_joint_type;

typedef struct rdyn_chain rdyn_chain; /* opaque; immutable after creation except rdyn_chain_set_input_joints */

/* ---- flat POD description of one serial chain, base -> tool (what Chain::init extracts from the urdf tree,
 *      primitives_impl.h:600-636).  For callers that already hold a parsed urdf::Model. ------------------- */
typedef struct rdyn_joint_desc
{
  char name[64];
  int32_t urdf_type;         /* rdyn_urdf_joint_type */
  double origin_xyz[3];      /* parent_to_joint_origin_transform.position            (primitives_impl.h:54) */
  double origin_quat[4];     /* parent_to_joint_origin_transform.rotation  x,y,z,w   (urdf_parser.h:44-50)  */
  double axis[3];            /* urdf joint axis, not necessarily normalised          (primitives_impl.h:55-59) */
  int32_t has_limits;        /* urdf_joint->limits != NULL                           (primitives_impl.h:87)  */
  double lower, upper, velocity, effort;
} rdyn_joint_desc;

typedef struct rdyn_link_desc
{
  char name[64];
  int32_t has_inertial;      /* urdf_link->inertial != NULL                          (primitives_impl.h:291) */
  double mass;
  double com_xyz[3];         /* inertial->origin.position                            (primitives_impl.h:305-307) */
  double com_quat[4];        /* inertial->origin.rotation x,y,z,w                    (primitives_impl.h:309-314) */
  double ixx, ixy, ixz, iyy, iyz, izz;
} rdyn_link_desc;

typedef struct rdyn_chain_desc
{
  int32_t n_joints;                  /* chain joints incl. fixed; links = n_joints + 1 */
  const rdyn_joint_desc* joints;     /* [n_joints], joint i connects links[i] -> links[i+1] */
  const rdyn_link_desc* links;       /* [n_joints + 1], links[0] = base link */
  double gravity[3];                 /* base-frame gravity vector; the reference's default is zero (primitives.h:346) */
} rdyn_chain_desc;

/* ---- construction (host only) --------------------------------------------------------------------- */
/* rosdyn::createChain(urdf::ModelInterface, base, tool, gravity)  primitives.h:566, primitives_impl.h:1518.
 * urdf_xml is the robot_description string; a minimal reader reproduces the urdfdom behaviour the
 * reference relies on (rpy->quaternion, default axis (1,0,0), missing origin/inertial). */
int rdyn_chain_from_urdf(const char* urdf_xml, const char* base_link, const char* tool_link,
                         const double gravity[3], rdyn_chain** out);
int rdyn_chain_from_desc(const rdyn_chain_desc* desc, rdyn_chain** out);
/* Chain::clone() primitives_impl.h:552 -- an independent deep copy */
int rdyn_chain_clone(const rdyn_chain* chain, rdyn_chain** out);
void rdyn_chain_destroy(rdyn_chain* chain);
const char* rdyn_last_error(void);

/* ---- introspection (host only); reference getters primitives.h:364-447 ------------------------------ */
int rdyn_chain_links_number(const rdyn_chain* chain);          /* getLinksNumber        */
int rdyn_chain_joints_number(const rdyn_chain* chain);         /* getJointsNumber       */
int rdyn_chain_active_joints_number(const rdyn_chain* chain);  /* getActiveJointsNumber */
int rdyn_chain_moveable_joints_number(const rdyn_chain* chain);
const char* rdyn_chain_link_name(const rdyn_chain* chain, int i);            /* getLinksName().at(i)          */
const char* rdyn_chain_joint_name(const rdyn_chain* chain, int i);           /* chain order, incl. fixed      */
const char* rdyn_chain_moveable_joint_name(const rdyn_chain* chain, int i);  /* getMoveableJointName(i)       */
const char* rdyn_chain_active_joint_name(const rdyn_chain* chain, int i);    /* getActiveJointName(i)         */
int rdyn_chain_joint_type(const rdyn_chain* chain, int i);                   /* rdyn_joint_type of chain joint i */
int rdyn_chain_gravity(const rdyn_chain* chain, double gravity[3]);          /* getGravity                    */
/* Chain::setInputJointsName primitives_impl.h:705.  Unknown names -> RDYN_ERR_JOINT_NOT_FOUND and the
 * chain is left unchanged (the reference returns false and continues in an inconsistent state). */
int rdyn_chain_set_input_joints(rdyn_chain* chain, const char* const* names, int n_names);
/* getQMax/getQMin/getDQMax/getDDQMax/getTauMax of the active joints (primitives_impl.h:85-143, 768-776);
 * any pointer may be NULL */
int rdyn_chain_limits(const rdyn_chain* chain, double* q_max, double* q_min, double* dq_max, double* ddq_max, double* tau_max);
/* The object tree behind a reference Chain (getJoints() / getLinks(), primitives.h:62-232), read only.
 * Chain joint i (chain order, fixed joints included): R_pj row-major 3x3 and t_pj of parent <- joint (primitives_impl.h:54, 68), the
 * normalised axis in the joint frame (:55-59), limits = {q_max, q_min, Dq_max, DDq_max, tau_max} (:85-143).  Joint::getTransformation(q):
 * revolute R_pj (I + sin q K + (1 - cos q) K^2), t_pj; prismatic R_pj, t_pj + R_pj axis q (:38-47).
 * Chain link i (0 = base link): Link::getNominalParameters [m, m c, Ixx Ixy Ixz Iyy Iyz Izz about the link origin] (:399-417), mass,
 * centre of gravity in the link frame.  Any output may be NULL. */
int rdyn_chain_joint_constants(const rdyn_chain* chain, int i, double R_pj[9], double t_pj[3], double axis[3], double limits[5]);
int rdyn_chain_link_parameters(const rdyn_chain* chain, int i, double pi[10], double* mass, double cog[3]);
/* Chain::getNominalParameters primitives_impl.h:1382 -> pi[10 * joints_number] (HOST pointer) */
int rdyn_nominal_parameters(const rdyn_chain* chain, double* pi);
/* Rigid-body reduction of a chain whose input joints are a subset of its joints, listed in any order (fixed joints, primitives_impl.h:74-83,
 * or joints left out of setInputJointsName): links joined by non-input joints move as one body, so the ten regressor columns of a
 * link f + 1 hanging from a non-input joint are a CONSTANT linear image of the ten columns of the link the body's input joint
 * carries:  Y(:, 10 f + p) = sum_a Y(:, 10 body_joint[f] + a) X[f][a][p]   (zero for links upstream of the first input joint).
 * The regressor -> Gram / R-factor entry points use it internally (they sweep the reduced chain and expand the small result);
 * exposed for callers that reduce the parameter vector themselves.  Returns the number of bodies (= input joints; the bodies are
 * numbered in CHAIN order whatever the order of the input list), 0 when the chain has no reduction (every joint an input joint, or
 * more than RDYN_MAX_SWEPT_JOINTS input joints), < 0 on a null chain.
 * body_joint[n_joints]: chain index of the input joint whose child link is the body's reference frame, -1 = rides on the base;
 * X[n_joints][10][10] row-major (a, p); pi_body[10 * bodies]: the merged nominal parameters.  HOST pointers, each may be NULL. */
int rdyn_chain_reduction(const rdyn_chain* chain, int32_t* body_joint, double* X, double* pi_body);

/* ---- batched evaluation (device pointers) ------------------------------------------------------------ */
typedef enum rdyn_layout
{
  /* x[s][e] -- every sample's record is contiguous and is exactly the memory image of the Eigen object the
   * reference returns for that sample (VectorXd, column-major MatrixXd / Matrix6Xd, 3x4 column-major
   * Affine3d::affine()).  Drop-in layout.  Kinematic / torque / inertia outputs that start on a 128-byte line (any
   * hipMalloc does) are written in whole lines through wave-private LDS (rdyn_record_stage.h) at about the rate of the
   * element-major layout; other alignments are served correctly by 8-byte stores at 2-3x the time. */
  RDYN_LAYOUT_SAMPLE_MAJOR = 0,
  /* x[e][s] -- one N-vector per record element (structure of arrays).  Fully coalesced on the GPU;
   * for the regressor this IS a column-major (n*N) x P matrix whose row index is j*N + s. */
  RDYN_LAYOUT_ELEMENT_MAJOR = 1
} rdyn_layout;

typedef struct rdyn_batch
{
  int64_t n_samples;
  const double* q;    /* n_active values per sample                           */
  const double* dq;   /* may be NULL where the call does not need velocities  */
  const double* ddq;  /* may be NULL where the call does not need accelerations */
  int32_t layout;     /* rdyn_layout of q/dq/ddq AND of every output of the call (the regressor excepted, below) */
  int32_t device;     /* HIP device ordinal, -1 = current device */
  void* stream;       /* hipStream_t; NULL = the default stream */
} rdyn_batch;

/* Regressor element (sample s, active-joint row j, parameter column p) is written to
 *   Y[s * stride_sample + j * stride_row + p * stride_col]        (strides in doubles). */
typedef struct rdyn_regressor_layout
{
  int64_t stride_sample, stride_row, stride_col;
} rdyn_regressor_layout;
/* presets (n = active joints, P = 10 * joints_number, N = samples):
 *   per-sample Eigen image  getRegressor()[s] column-major n x P : {n*P, 1, n}
 *   stacked column-major (N*n) x P, row = s*n + j                : {n,   1, N*n}
 *   element-major = column-major (n*N) x P, row = j*N + s        : {1,   N, n*N}   <- fastest */

/* getTransformation primitives.h:452 -> T_bt[12] per sample: column-major 3x4 [R | p] (= Affine3d::affine()).
 * T_links (optional, may be NULL): getTransformations primitives.h:454 -> links_number x 12 per sample. */
int rdyn_transformation(const rdyn_chain* chain, const rdyn_batch* batch, double* T_bt, double* T_links);
/* getJacobian primitives.h:455 -> 6 x n column-major per sample */
int rdyn_jacobian(const rdyn_chain* chain, const rdyn_batch* batch, double* J);
/* getJacobianLink primitives.h:456 / primitives_impl.h:951-979 -> 6 x n column-major per sample, referred to the origin
 * of chain link `link_index` (0 = base .. joints_number = tool; names via rdyn_chain_link_name).  As in the
 * reference only the first `up` input columns are filled (up = input joints upstream of the link, :965-972), the rest
 * are zero; with the default chain-ordered input list those are exactly the link's parent joints.  (The by-name getters getTransformationLink / getTwistLink are the record
 * `link_index` of rdyn_transformation's T_links / rdyn_twist's twists.)  Stateless: evaluated at the given q, whereas
 * the reference's getJacobianLink ignores its q and uses the frames of the previous call (primitives_impl.h:953). */
int rdyn_jacobian_link(const rdyn_chain* chain, const rdyn_batch* batch, int link_index, double* J);
/* getTwist primitives.h:457 (needs q, dq) -> links_number x 6 per sample, [lin; ang] per link.
 * getDTwist primitives.h:463 (needs q, dq, ddq) -> same shape; either output may be NULL. */
int rdyn_twist(const rdyn_chain* chain, const rdyn_batch* batch, double* twists, double* dtwists);
/* getDTwistLinearPart (needs q, ddq) primitives.h:468, getDTwistNonLinearPart (q, dq) :473, getDDTwist (q, dq, ddq, dddq)
 * :488 -> links_number x 6 per sample each; any output may be NULL; dddq (layout of q) only for ddtwists. */
int rdyn_twist_parts(const rdyn_chain* chain, const rdyn_batch* batch, const double* dddq, double* dtwists_linear,
                     double* dtwists_nonlinear, double* ddtwists);
/* getDDTwistLinearPart (needs q, dddq) primitives.h:476 / primitives_impl.h:1126-1154 and getDDTwistNonLinearPart
 * (q, dq, ddq) primitives.h:480 / primitives_impl.h:1156-1183 -> links_number x 6 per sample each; either may be NULL. */
int rdyn_jerk_parts(const rdyn_chain* chain, const rdyn_batch* batch, const double* dddq, double* ddtwists_linear,
                    double* ddtwists_nonlinear);
/* getWrench(q, Dq, DDq, ext_wrenches_in_link_frame) primitives.h:530 / primitives_impl.h:1225-1262 -> links_number x 6 per
 * sample: the wrench transmitted through every link ([force; torque], base-frame coordinates, referred to the link's
 * origin); getWrenchTool is the last record.  ext_wrenches as in rdyn_joint_torque_ext, NULL = none. */
int rdyn_wrench(const rdyn_chain* chain, const rdyn_batch* batch, const double* ext_wrenches, double* wrenches);
/* getJointTorque(q, Dq, DDq) primitives.h:540 -> n per sample */
int rdyn_joint_torque(const rdyn_chain* chain, const rdyn_batch* batch, double* tau);
/* getJointTorque(q, Dq, DDq, ext_wrenches_in_link_frame) primitives.h:539: ext_wrenches = links_number x 6 per sample
 * ([force; torque] applied TO each link, in the link's frame; record layout = batch->layout).  The reference's
 * twist-form transform of the wrench (primitives_impl.h:1255) is reproduced as it is. */
int rdyn_joint_torque_ext(const rdyn_chain* chain, const rdyn_batch* batch, const double* ext_wrenches, double* tau);
/* getJointTorqueNonLinearPart(q, Dq) primitives.h:541 (DDq = 0; batch->ddq ignored) */
int rdyn_joint_torque_nonlinear(const rdyn_chain* chain, const rdyn_batch* batch, double* tau);
/* getRegressor primitives.h:543 fused with getJointTorque: writes Y (dense, structural zeros included) and,
 * if tau != NULL, tau = Y * nominal parameters (== getJointTorque, identity verified to 1e-13). */
int rdyn_regressor(const rdyn_chain* chain, const rdyn_batch* batch, double* tau, double* Y, const rdyn_regressor_layout* y_layout);
/* getJointInertia primitives.h:547 -> n x n column-major per sample */
int rdyn_joint_inertia(const rdyn_chain* chain, const rdyn_batch* batch, double* M);

/* Every getter of a sample in ONE call (no reference counterpart: the reference caches what a call computed on the way, m_last_q,
 * primitives_impl.h:886, 985, 1088, so its harness rosdyn_speed_test.cpp:109-185 pays for the frames once per sample).  Outputs as the
 * single-purpose entry points write them, record layout = batch->layout; any of them may be NULL:
 *   T_links (links x 12), J (6 x n), twists, dtwists (links x 6), tau, tau_nonlinear (n), M (n x n), Y with y_layout (NULL layout: the
 *   per-sample Eigen image {n P, 1, n}).  Needs q, dq, ddq.
 * Chains of <= RDYN_MAX_SWEPT_JOINTS joints: ONE kernel launch, the seven sweeps side by side (made for small batches: a sample per call
 * costs a launch, not seven -- the C++ facade's evaluateAll reads q / Dq / DDq from and writes the record to pinned host memory, so a
 * sample costs that launch and a synchronisation); longer chains: the same results by the single-purpose launches on the stream. */
typedef struct rdyn_all_outputs
{
  double* T_links;
  double* J;
  double* twists;
  double* dtwists;
  double* tau;
  double* tau_nonlinear;
  double* M;
  double* Y;
  const rdyn_regressor_layout* y_layout;
} rdyn_all_outputs;
int rdyn_evaluate_all(const rdyn_chain* chain, const rdyn_batch* batch, const rdyn_all_outputs* out);

/* ---- batched local inverse kinematics.
 * computeLocalIk primitives.h:510 / primitives_impl.h:1398-1433 (weight == NULL) and computeWeigthedLocalIk
 * primitives.h:526 / primitives_impl.h:1436-1468 (weight = 6 HOST doubles), one pose per batch entry:
 *   repeat: e = getFrameDistance(T_target, getTransformation(sol)) (frame_distance.h:44-49); if |weight o e| < toll -> converged;
 *           dq = argmin 1/2 dq'(J'WJ)dq - (J'We)'dq  s.t.  q_min <= sol + dq <= q_max  (Eigen::solve_quadprog);  sol += dq.
 * batch->q = the seeds (n_active per pose, batch->layout); T_target = 12 per pose, the column-major 3x4 [R|p] record
 * rdyn_transformation writes, same layout; sol = like the seeds (may alias them).  The reference bounds the loop by
 * wall-clock time (max_time, default 5 ms); here by max_iterations QP updates.  Per-pose results (device int32, either
 * may be NULL): status 1 = converged (the reference's `true`), 0 = not within max_iterations (`false`), -1 = J'WJ not
 * positive definite (a Cholesky pivot <= 1e-10 trace: more than 6 input joints or a singular pose; the reference's
 * Cholesky-based QP is undefined there),
 * -2 = bounds infeasible (q_min > q_max), -3 = QP iteration guard; iterations = QP updates performed.  On a failure
 * sol holds the last iterate. */
int rdyn_local_ik(const rdyn_chain* chain, const rdyn_batch* batch, const double* T_target, const double* weight, double toll,
                  int max_iterations, double* sol, int32_t* status, int32_t* iterations);
/* The pose-error functions of frame_distance.h on n_pairs pairs of frames (12 doubles each: the column-major 3x4 [R|p]
 * record of rdyn_transformation; `layout` as in rdyn_batch) -> 6 per pair [translation; rotation], expressed in frame w:
 *   RDYN_FRAME_DISTANCE_AXIS_ANGLE  getFrameDistance        frame_distance.h:44-49    [p_a - p_b; -R_wa (angle axis)(R_wa' R_wb)]
 *   RDYN_FRAME_DISTANCE_QUAT        getFrameDistanceQuat    frame_distance.h:73-86    [p_a - p_b; -2 R_wa imag(q_ab)], q_ab.w >= 0
 *   RDYN_FRAME_DISTANCE_QUAT_JAC    getFrameDistanceQuatJac frame_distance.h:112-126  [p_b - p_a; -2 R_wa imag(q_ab)] (the
 *       translation is measured the other way round there) and, if jacobian != NULL, the 6 x 6 column-major
 *       [I 0; 0 R_wa (w I - skew(imag q_ab)) R_wa'] per pair. */
typedef enum rdyn_frame_distance_kind
{
  RDYN_FRAME_DISTANCE_AXIS_ANGLE = 0,
  RDYN_FRAME_DISTANCE_QUAT = 1,
  RDYN_FRAME_DISTANCE_QUAT_JAC = 2
} rdyn_frame_distance_kind;
int rdyn_frame_distance(int64_t n_pairs, const double* T_wa, const double* T_wb, int layout, int kind, double* distance,
                        double* jacobian, int device, void* stream);
/* The same iteration with a Levenberg term: damping^2 is added to the diagonal of J'WJ before the QP (damping = 0 is
 * rdyn_local_ik).  No counterpart in the reference; it is what makes the loop usable where the reference's QP is singular
 * (7-DOF arms: J'J is 7 x 7 of rank 6) and near singular poses. */
int rdyn_local_ik_damped(const rdyn_chain* chain, const rdyn_batch* batch, const double* T_target, const double* weight, double toll,
                         double damping, int max_iterations, double* sol, int32_t* status, int32_t* iterations);

/* ---- per-joint additive components: the extra regressor columns the identification step stacks next to
 * getRegressor.  Reference: FirstOrderPolynomialFriction friction_polynomial1.h:45-52
 * (columns [sign, omega]), SecondOrderPolynomialFriction friction_polynomial2.h:42-58 ([sign, omega, omega^2 sign]),
 * IdealSpring ideal_spring.h:64-70 ([q, 1]); getTorque = regressor row * parameters.  `joint` is the index of the
 * component's joint among the chain's ACTIVE joints (m_component_joint_number, base_component.h).
 * Constants follow friction_polynomial1.h:72-86: min_velocity < 1e-6 -> 1e-6, max_velocity <= 0 -> 1e6. */
typedef enum rdyn_component_type { RDYN_COMP_FRICTION1 = 0, RDYN_COMP_FRICTION2 = 1, RDYN_COMP_SPRING = 2 } rdyn_component_type;
typedef struct rdyn_component
{
  int32_t type;           /* rdyn_component_type */
  int32_t joint;          /* active-joint index */
  double min_velocity;    /* friction/constants/min_velocity */
  double max_velocity;    /* friction/constants/max_velocity */
  double parameters[3];   /* nominal: {coloumb, viscous} | {coloumb, first_order_viscous, second_order_viscous} | {elasticity, offset_effort} */
} rdyn_component;
/* Number of regressor columns of a component list (2, 3, 2 per type). */
int rdyn_components_columns(const rdyn_component* comps, int n_comps);
/* C(s, j, k) -> C[s*stride_sample + j*stride_row + k*stride_col] for k < rdyn_components_columns (dense: zero outside a
 * component's own joint row); pass Y + P*stride_col with Y's layout to append the columns to the inertial regressor.
 * tau_add (optional, layout of batch->q): += component torques.  C or tau_add may be NULL.  At most 30 components. */
int rdyn_components_regressor(const rdyn_component* comps, int n_comps, int n_active, const rdyn_batch* batch, double* C,
                              const rdyn_regressor_layout* c_layout, double* tau_add);

/* ---- mixed-chain batch (BASELINE.json configs[4]: 256 distinct 6-7-DOF chains x 4 096 samples) --------------
 * One launch per group of chains with equal joint count (and output-layout kind) evaluates rdyn_regressor for MANY
 * (chain, batch) items: grid = (ceil(max samples / workgroup), items); every workgroup reads its item's descriptor and its
 * chain's constants through scalar loads.  Items in the per-sample image or the stacked layout (rdyn_regressor_layout presets)
 * take the same LDS-staged whole-line kernels as rdyn_regressor; any other strides the strided one.  A plan freezes the items (device pointers, layouts) so that running it allocates and
 * copies nothing (graph-capturable).  All items must live on the same device (items[0].batch.device).  A plan holds the
 * device copies of its chains' constants: the chains must outlive the plan and must not be re-configured
 * (rdyn_chain_set_input_joints) while it exists. */
typedef struct rdyn_multi_item
{
  const rdyn_chain* chain;
  rdyn_batch batch;              /* stream field ignored */
  double* tau;                   /* may be NULL */
  double* Y;
  rdyn_regressor_layout y_layout;
} rdyn_multi_item;
typedef struct rdyn_multi_plan rdyn_multi_plan;
int rdyn_multi_plan_create(const rdyn_multi_item* items, int n_items, rdyn_multi_plan** out);
int rdyn_multi_plan_regressor(const rdyn_multi_plan* plan, void* stream);
void rdyn_multi_plan_destroy(rdyn_multi_plan* plan);

/* ---- normal equations of the stacked regressor (fp64 MFMA) ------------------------------------------------
 * No counterpart inside rosdyn_core: the identification step that stacked getRegressor rows and solved the
 * least-squares problem lived in the external rosdyn_identification (top-level README.md:15).  BASELINE.json's
 * north star asks for the Gram reduction on the matrix cores, so it is part of this boundary.
 *
 * rdyn_gram: A is any column-major rows x n_cols DEVICE matrix (leading dimension lda >= rows), b a DEVICE
 * vector of `rows` doubles or NULL.  Writes (accumulate == 0) or adds (accumulate != 0)
 *   G  = A^T A   n_cols x n_cols column-major, both triangles      (device)
 *   c  = A^T b   n_cols                                             (device, may be NULL)
 *   bb = b^T b   1                                                  (device, may be NULL)
 * with v_mfma_f64_16x16x4_f64; bitwise reproducible (fixed summation order, no atomics).
 * workspace: rdyn_gram_workspace_bytes(n_cols) bytes of device memory. */
size_t rdyn_gram_workspace_bytes(int n_cols);
int rdyn_gram(const double* A, int64_t rows, int64_t lda, int n_cols, const double* b, double* G, double* c, double* bb,
              int accumulate, void* workspace, size_t workspace_bytes, int device, void* stream);

/* rdyn_regressor_gram: G, c, bb of the stacked regressor A (N*n x P) of the batch and measured torques
 * tau_meas (same layout as batch->q) WITHOUT leaving the regressor in HBM: the batch is processed in chunks of
 * `chunk_samples` (0 = default 32768) whose element-major regressor image (chunk * n * (P + 1) doubles, reused
 * for every chunk, sized to stay resident in the 256 MiB Infinity Cache) is produced by the regressor kernel
 * and consumed by the Gram kernel.  Multi-GPU: every rank calls this on its shard and all-reduces
 * [G | c | bb] (P*P + P + 1 doubles) once -- rosdyn_amd/gram.py. */
/* The identification step in one call: normal equations of the stacked [Y | C] with the measured torque,
 *   G = [Y C]'[Y C]  ((P + K) x (P + K)),  c = [Y C]' tau_meas,  bb = tau_meas' tau_meas,
 * Y = getRegressor of the chain (P = 10 joints_number columns), C = the component columns of rdyn_components_regressor
 * (K = rdyn_components_columns; n_comps = 0: none).  Neither Y nor C is handed to the caller: they are produced chunk by
 * chunk into the workspace and reduced by the MFMA Gram kernel.  P + K <= 111.  Outputs / accumulate as rdyn_regressor_gram. */
size_t rdyn_identification_gram_workspace_bytes(const rdyn_chain* chain, const rdyn_component* comps, int n_comps);
int rdyn_identification_gram(const rdyn_chain* chain, const rdyn_component* comps, int n_comps, const rdyn_batch* batch,
                             const double* tau_meas, double* G, double* c, double* bb, int accumulate, void* workspace,
                             size_t workspace_bytes);
size_t rdyn_regressor_gram_workspace_bytes(const rdyn_chain* chain, int64_t chunk_samples);
int rdyn_regressor_gram(const rdyn_chain* chain, const rdyn_batch* batch, const double* tau_meas, double* G, double* c,
                        double* bb, int accumulate, int64_t chunk_samples, void* workspace, size_t workspace_bytes);

/* ---- BASELINE.json configs[3] inside the library (rdyn_multi_gpu.cpp): one process, the trajectory batch
 * sharded over the GPUs of a node, every GPU the fused regressor -> Gram of its shard, then ONE
 * ncclAllReduce(P*P + P + 2 doubles, ncclDouble, ncclSum) over RCCL / xGMI.  rdyn_multi_gpu_create initialises one communicator
 * (ncclCommInitAll) and one stream per device; RCCL is resolved at run time (the library named by RDYN_RCCL_PATH if that variable is
 * set, else librccl.so.1) -> RDYN_ERR_UNSUPPORTED if absent.
 * rdyn_regressor_gram_multi: batches[i] (device pointers on devices[i]; batch.device = devices[i] or -1), tau_meas[i] (may be NULL
 * as a whole or per shard), acc[i] = a device buffer of P*P + P + 2 doubles on devices[i].  The work runs on the CONTEXT's stream of
 * each device, ordered BEHIND everything already queued on batches[i].stream (NULL = that device's default stream) by an event, so
 * inputs still being produced there are safe; results are ordered for the caller by rdyn_multi_gpu_synchronize (or a device
 * synchronise).  Calls may be queued back to back without synchronising in between (the shard size is a kernel argument, nothing
 * host-side is re-used).  On completion
 * EVERY acc[i] holds the sums over all shards  [G = A'A (P*P, column-major) | c = A'tau_meas (P) | bb = tau_meas'tau_meas | count].
 * No reference counterpart (rosdyn_core has no multi-device code). */
typedef struct rdyn_multi_gpu rdyn_multi_gpu;
int rdyn_multi_gpu_create(const int* devices, int n_devices, rdyn_multi_gpu** out);
void rdyn_multi_gpu_destroy(rdyn_multi_gpu* ctx);
int rdyn_multi_gpu_device_count(const rdyn_multi_gpu* ctx);
int rdyn_multi_gpu_synchronize(rdyn_multi_gpu* ctx);
int rdyn_regressor_gram_multi(rdyn_multi_gpu* ctx, const rdyn_chain* chain, const rdyn_batch* batches, const double* const* tau_meas,
                              double* const* acc);
/* The same with accumulate != 0: acc[i] <- acc[i] + the sums over all shards of THIS call (a batch streamed through the devices in
 * pieces: the first piece with accumulate = 0, the others with 1; the accumulators hold the same totals on every device before and
 * after).  One ncclAllReduce per call either way.  The per-device part of both calls is issued by one host thread per device.
 * Failures: everything that can be refused (null pointers, devices, the chain's width) is checked BEFORE anything is queued; if a
 * device fails later (a launch or allocation error on one shard) the call returns that device's error WITHOUT issuing the collective,
 * and the other devices have already queued kernels that overwrite their outputs with their OWN shard's sums / factor: treat every
 * acc[i] / R1[i] of a failed call as undefined.  A collective that fails half-way aborts the communicators (and closes the RCCL group
 * it was issued in): the context then only accepts rdyn_multi_gpu_destroy; a new context can be created from the same thread. */
int rdyn_regressor_gram_multi_accumulate(rdyn_multi_gpu* ctx, const rdyn_chain* chain, const rdyn_batch* batches, const double* const* tau_meas,
                                         double* const* acc, int accumulate);
/* The R factor WITHOUT the normal equations over the devices of the context (the TSQR alternative):
 * device i computes the robust factor of its shard (rdyn_identification_tsqr with all its routes; comps may be NULL / n_comps 0:
 * rdyn_regressor_tsqr), ONE ncclAllGather moves the factors of the SWEPT chain (the reduced companion of a chain with joints that are
 * not input joints: at most 112 columns whatever the chain's own width -- the payload of the Gram all-reduce or less), every device
 * folds the same stack in the same fixed order and expands the result: on completion EVERY R1[i] (device i, n1 x n1 column-major,
 * n1 = 10 joints_number + K + 1) holds the factor of all shards, bitwise identical on all devices.  Every chain the single-device
 * entry point serves is served.  If a collective fails half-way the communicators are aborted and the context only accepts
 * rdyn_multi_gpu_destroy.  accumulate != 0: R1[i] <- factor of [previous R1[i] ; all shards].
 * Asynchronous like rdyn_regressor_gram_multi.  Both calls end by making batches[i].stream WAIT (on the device, by an event) for the
 * collective: work queued on the caller's stream afterwards sees the results; a host-side read still needs a synchronisation
 * (rdyn_multi_gpu_synchronize, or of that stream). */
int rdyn_identification_tsqr_multi(rdyn_multi_gpu* ctx, const rdyn_chain* chain, const rdyn_component* comps, int n_comps,
                                   const rdyn_batch* batches, const double* const* tau_meas, double* const* R1, int accumulate);
int rdyn_regressor_tsqr_multi(rdyn_multi_gpu* ctx, const rdyn_chain* chain, const rdyn_batch* batches, const double* const* tau_meas,
                              double* const* R1, int accumulate);

/* ---- tall-skinny QR (rdyn_tsqr.hip): the R factor of [A | b] WITHOUT forming A'A -- BASELINE.json configs[2] "regressor + TSQR".
 * The Gram route squares the condition number; this one does not.  Every wave folds row blocks into a running upper-triangular
 * factor by Householder reflections (R <- qr([R; block])), the factors are folded pairwise, level by level, in a fixed
 * order (bitwise reproducible).  Output R1, DEVICE, column-major n1 x n1 with n1 = n_cols + (b != NULL):
 *   R1 = [R d; 0 rho],  A = Q R,  d = Q'b,  rho = |A x_ls - b|   (diagonal signs are not normalised)
 * accumulate != 0: R1 <- factor of [previous R1 ; new rows] (chunked batches).  Multi-GPU: every rank all-gathers its R1 and folds
 * the stack with rdyn_tsqr_combine_host.  Solve with rdyn_solve_r_factor(R1, n1, n_cols, n_cols, d = R1 + n_cols * n1, ...).
 * rdyn_tsqr: any column-major rows x n_cols device matrix, n1 <= 112 (what rdyn_gram takes: a materialised [Y | C | tau_meas] of a
 * chain with fixed frames and friction columns has 90 - 110 columns).  rdyn_regressor_tsqr: the stacked regressor of the batch and
 * tau_meas (layout of batch->q), n1 = 10 joints_number + 1; chains of 1..10 INPUT joints in any order; joints that are not input
 * joints are folded away (the reduced chain of rdyn_chain_reduction is swept and the factor expanded by a small QR: up to
 * RDYN_MAX_JOINTS chain joints).  Up to 8 input joints the rows are generated in LDS by the regressor sweep and never stored (input
 * joints listed out of chain order: the kernels sweep in chain order and read q, Dq, DDq, tau_meas of every row through an index map
 * -- A'A, A'tau and R do not depend on the order of the rows inside a sample); 9..10 input joints: chunk images of 65 536 samples in
 * the workspace (65 536 x n x (10 n + 1) doubles: 0.43 GB at 9, 0.53 GB at 10 input joints -- per device in the multi-device form;
 * size buffers with rdyn_regressor_tsqr_workspace_bytes, never from a figure in a comment) factored by rdyn_tsqr's kernels.
 * rdyn_identification_tsqr: the same for the identification step's [Y | C | tau_meas] (C = the component columns of
 * rdyn_components_regressor -- friction_polynomial1.h:126, ideal_spring.h:64 -- K = rdyn_components_columns):
 * n1 = 10 joints_number + K + 1 <= 112 after the reduction, unknowns [inertial ; component] parameters.  The component columns
 * belong to input joints and pass through the reduction unchanged, so the reference's own chains (ur10 base_link -> tool0,
 * test.cpp:47-48; a Panda with its fixed flange and hand frames) are served with friction columns stacked beside getRegressor.
 * Routes, chosen by the shape and the batch size:
 *   Householder folds on the vector units -- in the registers of one wave where the factor fits them (rdyn_tsqr.hip: <= 64 columns of a
 *     matrix; swept chains of 2..7 joints, 2..6 with component columns; ~5x the time of rdyn_regressor_gram), with the factor packed in
 *     LDS beyond (rdyn_tsqr_wide.hip: up to 112 columns; a 7-joint arm with friction columns; slower, a dependent chain per column);
 *   from 4 096 samples (32 768 rows of a matrix) on, factors of <= 96 columns (rdyn_tsqr on a matrix: <= 112): preconditioned CholeskyQR with the heavy pass on the
 *     fp64 matrix cores (rdyn_cholqr.hip: a triangular T from the Gram matrix of a row subsample, W = T^-1 with nearly dependent
 *     pivots deferred, G2 = (A W)'(A W) over all rows by MFMA, R = chol(G2) T; the device accepts the result only if the measured
 *     error growth of A W and the conditioning of the equilibrated A W are small, runs a second round from R otherwise, and falls
 *     back to the Householder folds of all rows if that is not accepted either -- all inside the one asynchronous call; ~1.7x the
 *     time of rdyn_regressor_gram; rows of R at structurally dependent columns are exactly zero).
 * Every route returns R1 with R1'R1 = [A b]'[A b] to rounding and the small singular values to ~cond * eps.
 * The first call per (chain, device) uploads the chain's constants (a blocking copy): make it outside a stream capture. */
size_t rdyn_tsqr_workspace_bytes(int n_cols_with_rhs);
int rdyn_tsqr(const double* A, int64_t rows, int64_t lda, int n_cols, const double* b, double* R1, int accumulate, void* workspace,
              size_t workspace_bytes, int device, void* stream);
size_t rdyn_regressor_tsqr_workspace_bytes(const rdyn_chain* chain);
int rdyn_regressor_tsqr(const rdyn_chain* chain, const rdyn_batch* batch, const double* tau_meas, double* R1, int accumulate, void* workspace,
                        size_t workspace_bytes);
size_t rdyn_identification_tsqr_workspace_bytes(const rdyn_chain* chain, const rdyn_component* comps, int n_comps);
int rdyn_identification_tsqr(const rdyn_chain* chain, const rdyn_component* comps, int n_comps, const rdyn_batch* batch, const double* tau_meas,
                             double* R1, int accumulate, void* workspace, size_t workspace_bytes);
/* What the LAST rdyn_regressor_tsqr / rdyn_identification_tsqr call that used `workspace` did (the preconditioned route decides on the
 * device which of its stages vouches for the result; this reads the decision back).  Same chain, components and n_samples as that
 * call; waits for `stream`.  No reference counterpart. */
typedef struct rdyn_tsqr_report
{
  int32_t route;      /* 0: Householder folds (batches below 4 096 samples, shapes the other route does not serve): nothing else is
                         filled in; 1: preconditioned CholeskyQR */
  int32_t stage;      /* route 1: 0 = accepted after round 0, 1 = after round 1, 2 = the stand-by Householder factorisation ran */
  int32_t n_deferred; /* columns the last preconditioner did not use for elimination (structurally dependent or nearly so) */
  int32_t reserved;
  double gamma[2];    /* growth factor of the rounding of A W on the column norms of all rows, round 0 / 1 (0 = round not run);
                         accepted up to 1e4 */
  double rho[2];      /* |Re^-1|_F / sqrt(k) of the column-equilibrated A W (1 = orthogonal columns), round 0 / 1; accepted up to 4 */
} rdyn_tsqr_report;
int rdyn_tsqr_last_report(const rdyn_chain* chain, const rdyn_component* comps, int n_comps, int64_t n_samples, const void* workspace,
                          int device, void* stream, rdyn_tsqr_report* out);
/* the same for the last rdyn_tsqr call (n_cols_with_rhs = n_cols + (b != NULL), rows as in that call) */
int rdyn_tsqr_rows_last_report(int n_cols_with_rhs, int64_t rows, const void* workspace, int device, void* stream, rdyn_tsqr_report* out);
/* HOST: folds n_factors upper-triangular n x n factors (stacked, each column-major n x n) into one (Householder). */
int rdyn_tsqr_combine_host(const double* R_stack, int n_factors, int n, double* R_out);

/* ---- the small dense solves of the identification step (HOST pointers, column-major; rdyn_solve.cpp).  No counterpart inside
 * rosdyn_core (external rosdyn_identification, README.md:15).
 * rdyn_solve_normal_equations: minimum-norm least-squares solution x (n) of G x = c for the symmetric positive SEMI-definite
 *   n x n Gram G (the stacked regressor is structurally rank deficient: unobservable base-link parameters, fixed tail links):
 *   Jacobi eigen-decomposition, eigenvalues <= rtol * lambda_max dropped; *rank (may be NULL) = eigenvalues kept.
 * rdyn_gram_r_factor: rank-revealing R factor of A from its Gram: pivoted Cholesky, R'R = G[perm][:, perm]; R is n x n
 *   (rows >= *rank are zero), perm[n].  Squares the condition number (CholeskyQR): fine for cond(A) << 1e8.
 * rdyn_solve_r_factor: minimum-norm solution of min |R x - d| for a rows x n factor R (leading dimension ldr) as the TSQR path
 *   returns it (rdyn_regressor_tsqr below: condition number NOT squared), singular values <= rtol * sigma_max dropped. */
int rdyn_solve_normal_equations(const double* G, const double* c, int n, double rtol, double* x, int* rank);
int rdyn_gram_r_factor(const double* G, int n, double rtol, double* R, int32_t* perm, int* rank);
int rdyn_solve_r_factor(const double* R, int64_t ldr, int rows, int n, const double* d, double rtol, double* x, int* rank);

#ifdef __cplusplus
}
#endif
#endif /* RDYN_H */
