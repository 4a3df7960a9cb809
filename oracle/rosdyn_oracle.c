/*
 * rosdyn_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A scalar fp64 CPU restatement of the hot path of CNR-STIIMA-IRAS/rosdyn
 * (rosdyn_core), written operation by operation after the reference's Eigen
 * code: same frames (everything expressed in the base frame, reference point =
 * the link's own origin), same linear-first spatial vectors [lin; ang], same
 * dense 6x6 / 4x4 products, same 10 basis matrices for the regressor.
 *
 * PARITY UNPINNED (at the reference level): the reference cannot be built in
 * this image (it needs Eigen3, roscpp, urdfdom, eigen_matrix_utils -- none are
 * installed) and its own tests hold no golden vectors or numerical assertions
 * (rosdyn_core/test/test.cpp, rosdyn_speed_test.cpp).  The oracle is therefore
 * pinned only by this repository's own assets: an independent numpy
 * restatement (oracle/np_restatement.py -> tests/golden/), physics identities
 * and a symbolic 2R Lagrangian known-answer test (tests/test_oracle_*.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (librdyn_hip.so) never does.
 *
 * All reference citations are paths under /root/reference/rosdyn_core/include/rosdyn_core/ :
 *   sva.h  = spacevect_algebra.h
 *   impl.h = internal/primitives_impl.h
 *   urdf.h = urdf_parser.h
 *
 * Third-party arithmetic restated from published algorithms (sources absent):
 *   Eigen 3 (unpinned; CMakeLists.txt:31): Quaternion::toRotationMatrix, fixed-size products.
 *   urdfdom (ROS noetic system version, unpinned): the rpy->quaternion conversion is done by
 *   the Python front end (oracle/urdf_model.py); this file starts from the urdf::Pose.
 *
 * Statelessness: the reference caches on (q, Dq, DDq) with stale-cache hazards
 * (impl.h:886, 985, 1088, 1111).  The oracle always recomputes: it is the
 * stateless function of (chain, q, Dq, DDq) that a fresh reference Chain
 * evaluates on its first call.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAX_LINKS 33
#define ORC_REVOLUTE 0   /* primitives.h:67  enum Type {REVOLUTE, PRISMATIC, FIXED} */
#define ORC_PRISMATIC 1
#define ORC_FIXED 2

/* ------------------------------------------------------------------ small dense helpers */
typedef struct { double m[3][3]; } m3;
typedef struct { double v[3]; } v3;
typedef struct { double v[6]; } v6;      /* [lin(0..2); ang(3..5)]  sva.h:44-52 */
typedef struct { double m[6][6]; } m66;
typedef struct { double m[4][4]; } m4;   /* Eigen::Affine3d::matrix() */

static m3 m3_mul(const m3 a, const m3 b)
{
  m3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
    {
      double s = 0;
      for (int k = 0; k < 3; k++) s += a.m[i][k] * b.m[k][j];
      r.m[i][j] = s;
    }
  return r;
}
static m3 m3_T(const m3 a)
{
  m3 r;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i];
  return r;
}
static v3 m3_v(const m3 a, const v3 x)
{
  v3 r;
  for (int i = 0; i < 3; i++) r.v[i] = a.m[i][0] * x.v[0] + a.m[i][1] * x.v[1] + a.m[i][2] * x.v[2];
  return r;
}
static v3 cross(const v3 a, const v3 b)
{
  v3 r;
  r.v[0] = a.v[1] * b.v[2] - a.v[2] * b.v[1];
  r.v[1] = a.v[2] * b.v[0] - a.v[0] * b.v[2];
  r.v[2] = a.v[0] * b.v[1] - a.v[1] * b.v[0];
  return r;
}
static v3 v3_add(v3 a, v3 b) { v3 r = {{a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2]}}; return r; }
static v3 v3_sub(v3 a, v3 b) { v3 r = {{a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2]}}; return r; }
static v3 v3_scale(v3 a, double s) { v3 r = {{a.v[0] * s, a.v[1] * s, a.v[2] * s}}; return r; }
static v3 lin(const v6 x) { v3 r = {{x.v[0], x.v[1], x.v[2]}}; return r; }
static v3 ang(const v6 x) { v3 r = {{x.v[3], x.v[4], x.v[5]}}; return r; }
static v6 v6_from(v3 l, v3 a) { v6 r = {{l.v[0], l.v[1], l.v[2], a.v[0], a.v[1], a.v[2]}}; return r; }
static v6 v6_add(v6 a, v6 b) { v6 r; for (int i = 0; i < 6; i++) r.v[i] = a.v[i] + b.v[i]; return r; }
static v6 v6_scale(v6 a, double s) { v6 r; for (int i = 0; i < 6; i++) r.v[i] = a.v[i] * s; return r; }
static v6 v6_zero(void) { v6 r; memset(&r, 0, sizeof r); return r; }
static v6 m66_v(const m66* a, const v6 x)
{
  v6 r;
  for (int i = 0; i < 6; i++)
  {
    double s = 0;
    for (int k = 0; k < 6; k++) s += a->m[i][k] * x.v[k];
    r.v[i] = s;
  }
  return r;
}
static double v6_dot(v6 a, v6 b) { double s = 0; for (int i = 0; i < 6; i++) s += a.v[i] * b.v[i]; return s; }

/* ------------------------------------------------------------------ sva.h primitives */
/* sva.h:69-76 */
static m3 skew(const v3 a)
{
  m3 r = {{{0, -a.v[2], a.v[1]}, {a.v[2], 0, -a.v[0]}, {-a.v[1], a.v[0], 0}}};
  return r;
}
/* sva.h:88-93  twist x twist : ang = w1 x w2 ; lin = w1 x v2 + v1 x w2 */
static v6 spatialCrossProduct(const v6 a, const v6 b)
{
  return v6_from(v3_add(cross(ang(a), lin(b)), cross(lin(a), ang(b))), cross(ang(a), ang(b)));
}
/* sva.h:108-113 twist x* wrench : ang = w1 x t2 + v1 x f2 ; lin = w1 x f2 */
static v6 spatialDualCrossProduct(const v6 a, const v6 b)
{
  return v6_from(cross(ang(a), lin(b)), v3_add(cross(ang(a), ang(b)), cross(lin(a), lin(b))));
}
/* sva.h:129-133 twist: lin += ang x d */
static v6 spatialTranslation(const v6 t, const v3 d) { return v6_from(v3_add(lin(t), cross(ang(t), d)), ang(t)); }
/* sva.h:150-154 wrench: ang += lin x d */
static v6 spatialDualTranslation(const v6 w, const v3 d) { return v6_from(lin(w), v3_add(ang(w), cross(lin(w), d))); }
/* sva.h:172-175 */
static v6 spatialRotation(const v6 x, const m3 R) { return v6_from(m3_v(R, lin(x)), m3_v(R, ang(x))); }
/* sva.h:193-197 twist form: [R lin + (R ang) x t ; R ang] */
static v6 spatialTranformation(const v6 x, const m3 R, const v3 t)
{
  v3 Ra = m3_v(R, ang(x));
  return v6_from(v3_add(m3_v(R, lin(x)), cross(Ra, t)), Ra);
}
/* sva.h:232-239 */
static void computeSpatialInertiaMatrix(const m3 inertia, const v3 cog, double mass, m66* out)
{
  m3 cs = skew(cog), csT = m3_T(cs), cc = m3_mul(cs, csT);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
    {
      out->m[i][j] = mass * (i == j ? 1.0 : 0.0);
      out->m[i][3 + j] = mass * csT.m[i][j];
      out->m[3 + i][j] = mass * cs.m[i][j];
      out->m[3 + i][3 + j] = inertia.m[i][j] + mass * cc.m[i][j];
    }
}

/* Eigen::Quaterniond(w,x,y,z).toRotationMatrix()  (urdf.h:44-50, impl.h:317) -- Eigen's published formula, no normalisation */
static m3 quat_to_R(double x, double y, double z, double w)
{
  double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  double twx = tx * w, twy = ty * w, twz = tz * w;
  double txx = tx * x, txy = ty * x, txz = tz * x;
  double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  m3 R = {{{1 - (tyy + tzz), txy - twz, txz + twy},
           {txy + twz, 1 - (txx + tzz), tyz - twx},
           {txz - twy, tyz + twx, 1 - (txx + tyy)}}};
  return R;
}

/* ------------------------------------------------------------------ model objects */
typedef struct
{
  int type;                 /* after the mapping of impl.h:74-83 */
  m3 R_pj; v3 t_pj;         /* m_T_pj            impl.h:54 */
  v3 axis_in_j;             /* impl.h:55-59 */
  m3 skew_axis_in_j, square_skew_axis_in_j; /* impl.h:62-63 */
  v3 axis_in_p;             /* impl.h:69 */
  v6 screw_of_c_in_p;       /* impl.h:25-35 */
} orc_joint;

typedef struct
{
  double mass; v3 cog;
  m66 Inertia_cc;           /* impl.h:318 */
  m66 single_term[10];      /* impl.h:342-396 */
} orc_link;

typedef struct orc_chain
{
  int links_number, joints_number, active_joints_number; /* impl.h:638-639, 739 */
  orc_joint joints[ORC_MAX_LINKS];
  orc_link links[ORC_MAX_LINKS];       /* links[0] = base link */
  int active_joints[ORC_MAX_LINKS];    /* m_active_joints: chain index of input idx (impl.h:729) */
  double input_to_chain[ORC_MAX_LINKS][ORC_MAX_LINKS]; /* m_input_to_chain_joint nJ x nIn (impl.h:728) */
  v3 gravity;
} orc_chain;

/* urdf-model-level inputs (what urdfdom hands to the reference) */
typedef struct
{
  int urdf_type;            /* 0 revolute, 1 continuous, 2 prismatic, 3 fixed, 4 floating, 5 planar, 6 unknown */
  double xyz[3];            /* parent_to_joint_origin_transform.position */
  double quat[4];           /* .rotation x,y,z,w */
  double axis[3];
} orc_urdf_joint;

typedef struct
{
  int has_inertial;
  double mass;
  double xyz[3];            /* inertial->origin.position */
  double quat[4];           /* inertial->origin.rotation x,y,z,w */
  double ixx, ixy, ixz, iyy, iyz, izz;
} orc_urdf_link;

/* Joint::fromUrdf impl.h:50-83 + computeJacobian impl.h:25-35 */
static void joint_from_urdf(orc_joint* j, const orc_urdf_joint* u)
{
  j->R_pj = quat_to_R(u->quat[0], u->quat[1], u->quat[2], u->quat[3]); /* urdf.h:44-50 */
  for (int i = 0; i < 3; i++) j->t_pj.v[i] = u->xyz[i];
  for (int i = 0; i < 3; i++) j->axis_in_j.v[i] = u->axis[i];
  double nrm = sqrt(j->axis_in_j.v[0] * j->axis_in_j.v[0] + j->axis_in_j.v[1] * j->axis_in_j.v[1] + j->axis_in_j.v[2] * j->axis_in_j.v[2]);
  if (nrm > 0) /* impl.h:58-59 */
    for (int i = 0; i < 3; i++) j->axis_in_j.v[i] /= nrm;
  j->skew_axis_in_j = skew(j->axis_in_j);
  j->square_skew_axis_in_j = m3_mul(j->skew_axis_in_j, j->skew_axis_in_j);
  j->axis_in_p = m3_v(j->R_pj, j->axis_in_j);
  if (u->urdf_type == 0 || u->urdf_type == 1) j->type = ORC_REVOLUTE;   /* impl.h:74-77 */
  else if (u->urdf_type == 2) j->type = ORC_PRISMATIC;                  /* impl.h:78-81 */
  else j->type = ORC_FIXED;                                             /* impl.h:82-83 */
  j->screw_of_c_in_p = v6_zero();                                       /* impl.h:20 */
  v3 z = {{0, 0, 0}};
  if (j->type == ORC_REVOLUTE) j->screw_of_c_in_p = v6_from(z, j->axis_in_p);       /* impl.h:29 */
  else if (j->type == ORC_PRISMATIC) j->screw_of_c_in_p = v6_from(j->axis_in_p, z); /* impl.h:33 */
}

/* Link::fromUrdf impl.h:288-396 */
static void link_from_urdf(orc_link* l, const orc_urdf_link* u)
{
  m3 inertia;
  memset(&inertia, 0, sizeof inertia);
  l->mass = 0;
  memset(&l->cog, 0, sizeof l->cog);
  if (u->has_inertial)
  {
    l->mass = u->mass;
    inertia.m[0][0] = u->ixx; inertia.m[0][1] = u->ixy; inertia.m[0][2] = u->ixz;
    inertia.m[1][0] = u->ixy; inertia.m[1][1] = u->iyy; inertia.m[1][2] = u->iyz;
    inertia.m[2][0] = u->ixz; inertia.m[2][1] = u->iyz; inertia.m[2][2] = u->izz;
    for (int i = 0; i < 3; i++) l->cog.v[i] = u->xyz[i];
    m3 R = quat_to_R(u->quat[0], u->quat[1], u->quat[2], u->quat[3]);
    inertia = m3_mul(m3_mul(R, inertia), m3_T(R));                       /* impl.h:317 */
  }
  computeSpatialInertiaMatrix(inertia, l->cog, l->mass, &l->Inertia_cc); /* impl.h:318 / 325 */

  /* impl.h:342-396: the ten basis matrices are (re)written unconditionally, also for inertial-less links */
  for (int p = 0; p < 10; p++) memset(&l->single_term[p], 0, sizeof(m66));
  for (int i = 0; i < 3; i++) l->single_term[0].m[i][i] = 1.0;          /* mass */
  for (int k = 0; k < 3; k++)                                            /* mcx, mcy, mcz */
  {
    v3 e = {{0, 0, 0}};
    e.v[k] = 1;
    m3 s = skew(e), sT = m3_T(s);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
      {
        l->single_term[1 + k].m[i][3 + j] = sT.m[i][j];
        l->single_term[1 + k].m[3 + i][j] = s.m[i][j];
      }
  }
  l->single_term[4].m[3][3] = 1;                                   /* Ixx */
  l->single_term[5].m[3][4] = 1; l->single_term[5].m[4][3] = 1;    /* Ixy */
  l->single_term[6].m[3][5] = 1; l->single_term[6].m[5][3] = 1;    /* Ixz */
  l->single_term[7].m[4][4] = 1;                                   /* Iyy */
  l->single_term[8].m[4][5] = 1; l->single_term[8].m[5][4] = 1;    /* Iyz */
  l->single_term[9].m[5][5] = 1;                                   /* Izz */
}

/*
 * Build a chain from the ordered (base -> tool) joint / link lists the Python
 * front end extracted with the walk of Chain::init (impl.h:600-626).
 * input_chain_index[i] = chain joint index of input i (setInputJointsName, impl.h:724-731).
 */
orc_chain* orc_chain_create(int n_joints, const orc_urdf_joint* joints, const orc_urdf_link* links /* n_joints+1 */,
                            const double gravity[3], int n_inputs, const int* input_chain_index)
{
  if (n_joints + 1 > ORC_MAX_LINKS || n_joints < 0) return NULL;
  orc_chain* c = (orc_chain*)calloc(1, sizeof(orc_chain));
  c->joints_number = n_joints;
  c->links_number = n_joints + 1;
  for (int i = 0; i < n_joints; i++) joint_from_urdf(&c->joints[i], &joints[i]);
  for (int i = 0; i < n_joints + 1; i++) link_from_urdf(&c->links[i], &links[i]);
  for (int i = 0; i < 3; i++) c->gravity.v[i] = gravity[i];
  c->active_joints_number = n_inputs;
  for (int i = 0; i < n_inputs; i++)
  {
    c->active_joints[i] = input_chain_index[i];
    c->input_to_chain[input_chain_index[i]][i] = 1.0;
  }
  return c;
}
void orc_chain_destroy(orc_chain* c) { free(c); }
int orc_chain_links(const orc_chain* c) { return c->links_number; }
int orc_chain_joints(const orc_chain* c) { return c->joints_number; }
int orc_chain_active(const orc_chain* c) { return c->active_joints_number; }

/* ------------------------------------------------------------------ per-sample state (the reference's Chain members) */
typedef struct
{
  double sorted_q[ORC_MAX_LINKS], sorted_Dq[ORC_MAX_LINKS], sorted_DDq[ORC_MAX_LINKS], sorted_DDDq[ORC_MAX_LINKS];
  m4 T_bl[ORC_MAX_LINKS];
  v6 screws[ORC_MAX_LINKS], twists[ORC_MAX_LINKS], Dtwists[ORC_MAX_LINKS];
  v6 Dtw_lin[ORC_MAX_LINKS], Dtw_nonlin[ORC_MAX_LINKS], DDtwists[ORC_MAX_LINKS];
  v6 wrenches[ORC_MAX_LINKS], inertial_w[ORC_MAX_LINKS], gravity_w[ORC_MAX_LINKS];
  double wreg[ORC_MAX_LINKS][6][10];
} orc_state;

static m3 T_R(const m4* T) { m3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = T->m[i][j]; return r; }
static v3 T_p(const m4* T) { v3 r = {{T->m[0][3], T->m[1][3], T->m[2][3]}}; return r; }

/* m_sorted_x = m_input_to_chain_joint * x   (impl.h:865, 984, 1086) -- dense 0/1 mat-vec */
static void sort_in(const orc_chain* c, const double* x, double* sorted)
{
  for (int r = 0; r < c->joints_number; r++)
  {
    double s = 0;
    for (int k = 0; k < c->active_joints_number; k++) s += c->input_to_chain[r][k] * x[k];
    sorted[r] = s;
  }
}

/* Joint::getTransformation / computedTpc  impl.h:38-47, 219-227 */
static m4 joint_T_pc(const orc_joint* j, double q)
{
  m3 R = j->R_pj;
  v3 t = j->t_pj;
  if (j->type == ORC_REVOLUTE)
  {
    m3 R_jc;
    double s = sin(q), c1 = 1 - cos(q);
    for (int i = 0; i < 3; i++)
      for (int k = 0; k < 3; k++)
        R_jc.m[i][k] = (i == k ? 1.0 : 0.0) + s * j->skew_axis_in_j.m[i][k] + c1 * j->square_skew_axis_in_j.m[i][k]; /* impl.h:42 */
    R = m3_mul(j->R_pj, R_jc);                                                                                     /* impl.h:43 */
  }
  else if (j->type == ORC_PRISMATIC)
    t = v3_add(j->t_pj, v3_scale(j->axis_in_p, q));                                                                /* impl.h:46 */
  m4 T;
  memset(&T, 0, sizeof T);
  for (int i = 0; i < 3; i++) { for (int k = 0; k < 3; k++) T.m[i][k] = R.m[i][k]; T.m[i][3] = t.v[i]; }
  T.m[3][3] = 1;
  return T;
}

/* Chain::computeFrames impl.h:863-872  (4x4 matrix products) */
static void computeFrames(const orc_chain* c, orc_state* s, const double* q)
{
  sort_in(c, q, s->sorted_q);
  memset(&s->T_bl[0], 0, sizeof(m4));
  for (int i = 0; i < 4; i++) s->T_bl[0].m[i][i] = 1;
  for (int nl = 1; nl < c->links_number; nl++)
  {
    int nj = nl - 1;
    m4 Tpc = joint_T_pc(&c->joints[nj], s->sorted_q[nj]);
    for (int i = 0; i < 4; i++)
      for (int k = 0; k < 4; k++)
      {
        double a = 0;
        for (int m = 0; m < 4; m++) a += s->T_bl[nl - 1].m[i][m] * Tpc.m[m][k];
        s->T_bl[nl].m[i][k] = a;
      }
  }
}
/* Chain::computeScrews impl.h:874-882 : rotated by the PARENT link frame */
static void computeScrews(const orc_chain* c, orc_state* s)
{
  s->screws[0] = v6_zero();
  for (int nl = 1; nl < c->links_number; nl++)
    s->screws[nl] = spatialRotation(c->joints[nl - 1].screw_of_c_in_p, T_R(&s->T_bl[nl - 1]));
}
/* Chain::getTwist impl.h:981-1013 */
static void getTwist(const orc_chain* c, orc_state* s, const double* Dq)
{
  sort_in(c, Dq, s->sorted_Dq);
  s->twists[0] = v6_zero();
  for (int nl = 1; nl < c->links_number; nl++)
  {
    v3 d = v3_sub(T_p(&s->T_bl[nl]), T_p(&s->T_bl[nl - 1]));
    s->twists[nl] = v6_add(spatialTranslation(s->twists[nl - 1], d), v6_scale(s->screws[nl], s->sorted_Dq[nl - 1]));
  }
}
/* Chain::getDTwist impl.h:1082-1124 (direct branch 1113-1118) */
static void getDTwist(const orc_chain* c, orc_state* s, const double* DDq)
{
  sort_in(c, DDq, s->sorted_DDq);
  s->Dtwists[0] = v6_zero();
  for (int nl = 1; nl < c->links_number; nl++)
  {
    int nj = nl - 1;
    v3 d = v3_sub(T_p(&s->T_bl[nl]), T_p(&s->T_bl[nl - 1]));
    s->Dtwists[nl] = v6_add(v6_add(spatialTranslation(s->Dtwists[nl - 1], d),
                                   v6_scale(spatialCrossProduct(s->twists[nl], s->screws[nl]), s->sorted_Dq[nj])),
                            v6_scale(s->screws[nl], s->sorted_DDq[nj]));
  }
}
/* Chain::getDTwistLinearPart impl.h:1029-1061 */
static void getDTwistLinearPart(const orc_chain* c, orc_state* s, const double* DDq)
{
  sort_in(c, DDq, s->sorted_DDq);
  s->Dtw_lin[0] = v6_zero();
  for (int nl = 1; nl < c->links_number; nl++)
  {
    v3 d = v3_sub(T_p(&s->T_bl[nl]), T_p(&s->T_bl[nl - 1]));
    s->Dtw_lin[nl] = v6_add(spatialTranslation(s->Dtw_lin[nl - 1], d), v6_scale(s->screws[nl], s->sorted_DDq[nl - 1]));
  }
}
/* Chain::getDTwistNonLinearPart impl.h:1063-1080 */
static void getDTwistNonLinearPart(const orc_chain* c, orc_state* s)
{
  s->Dtw_nonlin[0] = v6_zero();
  for (int nl = 1; nl < c->links_number; nl++)
  {
    v3 d = v3_sub(T_p(&s->T_bl[nl]), T_p(&s->T_bl[nl - 1]));
    s->Dtw_nonlin[nl] = v6_add(spatialTranslation(s->Dtw_nonlin[nl - 1], d),
                               v6_scale(spatialCrossProduct(s->twists[nl], s->screws[nl]), s->sorted_Dq[nl - 1]));
  }
}
/* Chain::getDDTwist impl.h:1185-1223 (direct branch 1210-1219) */
static void getDDTwist(const orc_chain* c, orc_state* s, const double* DDDq)
{
  sort_in(c, DDDq, s->sorted_DDDq);
  s->DDtwists[0] = v6_zero();
  for (int nl = 1; nl < c->links_number; nl++)
  {
    int nj = nl - 1;
    v3 d = v3_sub(T_p(&s->T_bl[nl]), T_p(&s->T_bl[nl - 1]));
    v6 v_cross_s = spatialCrossProduct(s->twists[nl], s->screws[nl]);
    v6 r = spatialTranslation(s->DDtwists[nl - 1], d);
    r = v6_add(r, v6_scale(s->screws[nl], s->sorted_DDDq[nj]));
    r = v6_add(r, v6_scale(v_cross_s, s->sorted_DDq[nj]));
    r = v6_add(r, v6_scale(v6_add(spatialCrossProduct(s->Dtwists[nl], s->screws[nl]), spatialCrossProduct(s->twists[nl], v_cross_s)),
                           s->sorted_Dq[nj]));
    s->DDtwists[nl] = r;
  }
}
/* Chain::getWrench impl.h:1225-1262.  ext = L x 6 external wrenches in link frame (may be NULL = zeros) */
static void getWrench(const orc_chain* c, orc_state* s, const double* ext)
{
  int L = c->links_number;
  for (int nl = L - 1; nl >= 0; nl--)
  {
    m3 R = T_R(&s->T_bl[nl]), Rt = m3_T(R);
    if (nl == 0)
    {
      s->inertial_w[nl] = v6_zero();
      s->gravity_w[nl] = v6_zero();
    }
    else
    {
      const m66* I = &c->links[nl].Inertia_cc;
      v6 a_loc = spatialRotation(s->Dtwists[nl], Rt);
      v6 v_loc = spatialRotation(s->twists[nl], Rt);
      v6 v_loc2 = spatialRotation(s->twists[nl], Rt);
      v6 w = v6_add(m66_v(I, a_loc), spatialDualCrossProduct(v_loc, m66_v(I, v_loc2)));     /* impl.h:1240-1247 */
      s->inertial_w[nl] = spatialRotation(w, R);                                           /* impl.h:1248 */
      v3 mg = v3_scale(c->gravity, c->links[nl].mass);
      v3 gl = v3_scale(mg, -1.0);                                                          /* impl.h:1249 */
      v3 ga = v3_scale(cross(m3_v(R, c->links[nl].cog), mg), -1.0);                        /* impl.h:1250 */
      s->gravity_w[nl] = v6_from(gl, ga);
    }
    v6 e = v6_zero();
    if (ext) for (int i = 0; i < 6; i++) e.v[i] = -ext[nl * 6 + i];
    v6 r = v6_add(v6_add(spatialTranformation(e, R, T_p(&s->T_bl[nl])), s->inertial_w[nl]), s->gravity_w[nl]); /* impl.h:1255,1257 */
    if (nl < L - 1)
      r = v6_add(r, spatialDualTranslation(s->wrenches[nl + 1], v3_sub(T_p(&s->T_bl[nl]), T_p(&s->T_bl[nl + 1]))));
    s->wrenches[nl] = r;
  }
}

/* ------------------------------------------------------------------ public per-sample entry points */
/* getTransformations impl.h:908 : T_all = L x 12 (row-major 3x4 [R|p]) */
void orc_fk(const orc_chain* c, const double* q, double* T_all)
{
  orc_state s;
  computeFrames(c, &s, q);
  for (int l = 0; l < c->links_number; l++)
    for (int i = 0; i < 3; i++)
      for (int k = 0; k < 4; k++) T_all[l * 12 + i * 4 + k] = s.T_bl[l].m[i][k];
}
/* getJacobian impl.h:927-949 : J = 6 x n_active, column-major (Eigen::Matrix6Xd image) */
void orc_jacobian(const orc_chain* c, const double* q, double* J)
{
  orc_state s;
  computeFrames(c, &s, q);
  computeScrews(c, &s);
  int L = c->links_number;
  memset(J, 0, sizeof(double) * 6 * c->active_joints_number); /* impl.h:751-752 */
  for (int idx = 0; idx < c->active_joints_number; idx++)
  {
    int nj = c->active_joints[idx], nl = nj + 1;
    if (c->joints[nj].type != ORC_FIXED)
    {
      v6 col = spatialTranslation(s.screws[nl], v3_sub(T_p(&s.T_bl[L - 1]), T_p(&s.T_bl[nl])));
      for (int i = 0; i < 6; i++) J[idx * 6 + i] = col.v[i];
    }
  }
}
/* getJacobianLink impl.h:951-979, evaluated on the frames of q (the reference uses the frames of the previous call, :953).
 * m_parent_moveable_joints_of_link.at(link) (impl.h:798-827) lists the input positions of the active joints found while
 * walking the chain joints up to the one whose child is the link; only its SIZE is used (:970), the column index and the
 * joint both come from the loop counter (:972, :975). */
void orc_jacobian_link(const orc_chain* c, const double* q, const double* link_idx_as_double, double* J)
{
  orc_state s;
  computeFrames(c, &s, q);
  computeScrews(c, &s);
  int link_idx = (int)link_idx_as_double[0];
  int n_parent = 0; /* joints.size(), impl.h:808-821: chain joints 0 .. link_idx-1 that are active */
  for (int ijnt = 0; ijnt < link_idx; ijnt++)
    for (int k = 0; k < c->active_joints_number; k++)
      if (c->active_joints[k] == ijnt) { n_parent++; break; }
  memset(J, 0, sizeof(double) * 6 * c->active_joints_number); /* :969 */
  for (int idx = 0; idx < n_parent; idx++)
  {
    int nj = c->active_joints[idx], nl = nj + 1;
    if (c->joints[nj].type != ORC_FIXED)
    {
      v6 col = spatialTranslation(s.screws[nl], v3_sub(T_p(&s.T_bl[link_idx]), T_p(&s.T_bl[nl])));
      for (int i = 0; i < 6; i++) J[idx * 6 + i] = col.v[i];
    }
  }
}
/* getTwist impl.h:981 : twists = L x 6 */
void orc_twist(const orc_chain* c, const double* q, const double* Dq, double* twists)
{
  orc_state s;
  computeFrames(c, &s, q);
  computeScrews(c, &s);
  getTwist(c, &s, Dq);
  for (int l = 0; l < c->links_number; l++) for (int i = 0; i < 6; i++) twists[l * 6 + i] = s.twists[l].v[i];
}
/* getDTwist impl.h:1082 ; also the linear / non-linear split (impl.h:1029, 1063) when the pointers are non-NULL */
void orc_dtwist(const orc_chain* c, const double* q, const double* Dq, const double* DDq, double* dtw, double* dtw_lin, double* dtw_nonlin)
{
  orc_state s;
  computeFrames(c, &s, q);
  computeScrews(c, &s);
  getTwist(c, &s, Dq);
  getDTwist(c, &s, DDq);
  for (int l = 0; l < c->links_number; l++) for (int i = 0; i < 6; i++) dtw[l * 6 + i] = s.Dtwists[l].v[i];
  if (dtw_lin)
  {
    getDTwistLinearPart(c, &s, DDq);
    for (int l = 0; l < c->links_number; l++) for (int i = 0; i < 6; i++) dtw_lin[l * 6 + i] = s.Dtw_lin[l].v[i];
  }
  if (dtw_nonlin)
  {
    getDTwistNonLinearPart(c, &s);
    for (int l = 0; l < c->links_number; l++) for (int i = 0; i < 6; i++) dtw_nonlin[l * 6 + i] = s.Dtw_nonlin[l].v[i];
  }
}
/* getDDTwist impl.h:1185 */
void orc_ddtwist(const orc_chain* c, const double* q, const double* Dq, const double* DDq, const double* DDDq, double* ddtw)
{
  orc_state s;
  computeFrames(c, &s, q);
  computeScrews(c, &s);
  getTwist(c, &s, Dq);
  getDTwist(c, &s, DDq);
  getDDTwist(c, &s, DDDq);
  for (int l = 0; l < c->links_number; l++) for (int i = 0; i < 6; i++) ddtw[l * 6 + i] = s.DDtwists[l].v[i];
}
/* getDDTwistLinearPart impl.h:1126-1154 (jl) and getDDTwistNonLinearPart impl.h:1156-1183 (jn); either may be NULL */
void orc_ddtwist_parts(const orc_chain* c, const double* q, const double* Dq, const double* DDq, const double* DDDq, double* jl, double* jn)
{
  orc_state s;
  computeFrames(c, &s, q);
  computeScrews(c, &s);
  getTwist(c, &s, Dq);
  getDTwist(c, &s, DDq);
  sort_in(c, DDDq, s.sorted_DDDq);
  v6 lin_part[ORC_MAX_LINKS], nonlin_part[ORC_MAX_LINKS];
  lin_part[0] = v6_zero();
  nonlin_part[0] = v6_zero();
  for (int nl = 1; nl < c->links_number; nl++)
  {
    int nj = nl - 1;
    v3 d = v3_sub(T_p(&s.T_bl[nl]), T_p(&s.T_bl[nl - 1]));
    lin_part[nl] = v6_add(spatialTranslation(lin_part[nl - 1], d), v6_scale(s.screws[nl], s.sorted_DDDq[nj]));      /* :1148-1149 */
    v6 v_cross_s = spatialCrossProduct(s.twists[nl], s.screws[nl]);                                                   /* :1174 */
    v6 r = spatialTranslation(nonlin_part[nl - 1], d);                                                                /* :1175 */
    r = v6_add(r, v6_scale(v_cross_s, s.sorted_DDq[nj]));                                                             /* :1176 */
    r = v6_add(r, v6_scale(v6_add(spatialCrossProduct(s.Dtwists[nl], s.screws[nl]), spatialCrossProduct(s.twists[nl], v_cross_s)),
                           s.sorted_Dq[nj]));                                                                         /* :1177-1178 */
    nonlin_part[nl] = r;
  }
  for (int l = 0; l < c->links_number; l++)
    for (int i = 0; i < 6; i++)
    {
      if (jl) jl[l * 6 + i] = lin_part[l].v[i];
      if (jn) jn[l * 6 + i] = nonlin_part[l].v[i];
    }
}
/* getJointTorque impl.h:1264-1283 : tau = n_active ; wrenches (optional) = L x 6 */
void orc_joint_torque(const orc_chain* c, const double* q, const double* Dq, const double* DDq, const double* ext, double* tau, double* wrenches)
{
  orc_state s;
  computeFrames(c, &s, q);
  computeScrews(c, &s);
  getTwist(c, &s, Dq);
  getDTwist(c, &s, DDq);
  getWrench(c, &s, ext);
  double jt[ORC_MAX_LINKS];
  for (int nj = 0; nj < c->joints_number; nj++) jt[nj] = v6_dot(s.wrenches[nj + 1], s.screws[nj + 1]); /* impl.h:1270 */
  for (int k = 0; k < c->active_joints_number; k++)                                                   /* impl.h:1272 */
  {
    double a = 0;
    for (int r = 0; r < c->joints_number; r++) a += c->input_to_chain[r][k] * jt[r];
    tau[k] = a;
  }
  if (wrenches) for (int l = 0; l < c->links_number; l++) for (int i = 0; i < 6; i++) wrenches[l * 6 + i] = s.wrenches[l].v[i];
}
/* getRegressor impl.h:1295-1355 : Y = n_active x (10 nJ), column-major (Eigen::MatrixXd image) */
void orc_regressor(const orc_chain* c, const double* q, const double* Dq, const double* DDq, double* Y)
{
  orc_state* s = (orc_state*)malloc(sizeof(orc_state));
  int L = c->links_number, nJ = c->joints_number, P = 10 * nJ;
  computeFrames(c, s, q);
  computeScrews(c, s);
  getTwist(c, s, Dq);
  getDTwist(c, s, DDq);
  double* Yext = (double*)calloc((size_t)nJ * P, sizeof(double)); /* m_regressor_extended, zero-filled impl.h:690-691,718 */
  for (int nl = L - 1; nl > 0; nl--)
  {
    m3 R = T_R(&s->T_bl[nl]), Rt = m3_T(R);
    for (int p = 0; p < 10; p++)
    {
      const m66* E = &c->links[nl].single_term[p];
      v6 a_loc = spatialRotation(s->Dtwists[nl], Rt);
      v6 v_loc = spatialRotation(s->twists[nl], Rt);
      v6 col = spatialRotation(v6_add(m66_v(E, a_loc), spatialDualCrossProduct(v_loc, m66_v(E, v_loc))), R); /* impl.h:1326-1332 */
      for (int i = 0; i < 6; i++) s->wreg[nl][i][p] = col.v[i];
    }
    for (int i = 0; i < 3; i++) s->wreg[nl][i][0] -= c->gravity.v[i];                                       /* impl.h:1336 */
    for (int k = 0; k < 3; k++)                                                                              /* impl.h:1337-1339 */
    {
      v3 e = {{0, 0, 0}};
      e.v[k] = 1;
      v3 x = cross(m3_v(R, e), c->gravity);
      for (int i = 0; i < 3; i++) s->wreg[nl][3 + i][1 + k] -= x.v[i];
    }
    for (int p = 0; p < 10; p++)                                                                             /* impl.h:1341 */
    {
      double a = 0;
      for (int i = 0; i < 6; i++) a += s->screws[nl].v[i] * s->wreg[nl][i][p];
      Yext[(nl - 1) + (size_t)nJ * ((nl - 1) * 10 + p)] = a;
    }
    for (int nf = nl + 1; nf < L; nf++)                                                                      /* impl.h:1343-1347 */
      for (int p = 0; p < 10; p++)
      {
        v6 col;
        for (int i = 0; i < 6; i++) col.v[i] = s->wreg[nf][i][p];
        v6 t = spatialDualTranslation(col, v3_sub(T_p(&s->T_bl[nl]), T_p(&s->T_bl[nf])));
        Yext[(nl - 1) + (size_t)nJ * ((nf - 1) * 10 + p)] = v6_dot(s->screws[nl], t);
      }
  }
  /* result = (Yext^T * chain_to_input^T)^T   impl.h:1352 */
  int n = c->active_joints_number;
  for (int p = 0; p < P; p++)
    for (int k = 0; k < n; k++)
    {
      double a = 0;
      for (int r = 0; r < nJ; r++) a += Yext[r + (size_t)nJ * p] * c->input_to_chain[r][k];
      Y[k + (size_t)n * p] = a;
    }
  free(Yext);
  free(s);
}
/* getJointInertia impl.h:1357-1379 : M = n x n column-major */
void orc_joint_inertia(const orc_chain* c, const double* q, double* M)
{
  orc_state s;
  int nJ = c->joints_number, n = c->active_joints_number;
  computeFrames(c, &s, q);
  computeScrews(c, &s);
  double Mext[ORC_MAX_LINKS][ORC_MAX_LINKS];
  memset(Mext, 0, sizeof Mext);
  for (int nj = 0; nj < nJ; nj++)
  {
    v6 jac[ORC_MAX_LINKS];
    for (int i = 0; i < nJ; i++) jac[i] = v6_zero();
    m3 Rt = m3_T(T_R(&s.T_bl[nj + 1]));
    for (int ij = 0; ij <= nj; ij++)
    {
      int il = ij + 1;
      if (c->joints[ij].type != ORC_FIXED)
      {
        v6 t = spatialTranslation(s.screws[il], v3_sub(T_p(&s.T_bl[nj + 1]), T_p(&s.T_bl[il]))); /* impl.h:1371 */
        jac[ij] = spatialRotation(t, Rt);                                                         /* impl.h:1372 */
      }
    }
    const m66* I = &c->links[nj + 1].Inertia_cc;
    for (int a = 0; a < nJ; a++)                                                                  /* impl.h:1375 */
    {
      v6 Ia = m66_v(I, jac[a]); /* I * J(:,a) */
      for (int b = 0; b < nJ; b++) Mext[b][a] += v6_dot(jac[b], Ia);
    }
  }
  /* M = chain_to_input * Mext * input_to_chain  impl.h:1377 */
  for (int a = 0; a < n; a++)
    for (int b = 0; b < n; b++)
    {
      double acc = 0;
      for (int r = 0; r < nJ; r++)
        for (int t = 0; t < nJ; t++) acc += c->input_to_chain[r][a] * Mext[r][t] * c->input_to_chain[t][b];
      M[a + (size_t)n * b] = acc;
    }
}
/* Link::getNominalParameters impl.h:399-417 + Chain::getNominalParameters impl.h:1382-1391 */
void orc_nominal_parameters(const orc_chain* c, double* pi)
{
  for (int nl = c->links_number - 1; nl > 0; nl--)
  {
    const orc_link* l = &c->links[nl];
    double* o = pi + 10 * (nl - 1);
    o[0] = l->mass;
    for (int i = 0; i < 3; i++) o[1 + i] = l->cog.v[i] * l->mass;
    o[4] = l->Inertia_cc.m[3][3]; o[5] = l->Inertia_cc.m[3][4]; o[6] = l->Inertia_cc.m[3][5];
    o[7] = l->Inertia_cc.m[4][4]; o[8] = l->Inertia_cc.m[4][5];
    o[9] = l->Inertia_cc.m[5][5];
  }
}

/* ------------------------------------------------------------------ batched drivers (the caller's for-loop, speed.cpp:109-192) */
/* AoS inputs q[N][n]; tau[N][n]; Y[N][n*P] (per-sample column-major n x P).  threads<=1 -> serial. Returns threads used. */
int orc_batch_torque_regressor(const orc_chain* c, long N, const double* q, const double* Dq, const double* DDq,
                               double* tau, double* Y, int threads)
{
  int n = c->active_joints_number, P = 10 * c->joints_number;
  int used = 1;
#ifdef _OPENMP
  if (threads > 1) used = threads;
#pragma omp parallel for num_threads(used) schedule(static)
#endif
  for (long s = 0; s < N; s++)
  {
    if (tau) orc_joint_torque(c, q + s * n, Dq + s * n, DDq + s * n, NULL, tau + s * n, NULL);
    if (Y) orc_regressor(c, q + s * n, Dq + s * n, DDq + s * n, Y + (size_t)s * n * P);
  }
  return used;
}
int orc_has_openmp(void)
{
#ifdef _OPENMP
  return 1;
#else
  return 0;
#endif
}

/* ------------------------------------------------------------------ harness restatement (BASELINE.json configs[0])
 * rosdyn_core/test/rosdyn_speed_test.cpp:109-204: `ntrial` iterations; before EVERY timed call fresh q, Dq, DDq,
 * DDDq ~ U[-1,1] (Eigen setRandom there, a seeded splitmix64 stream here); mean microseconds per call for, in the
 * reference's order: pose, jacobian, twists, linear-acceleration twists, non-linear-acceleration twists,
 * acceleration twists, jerk twists, joint torque, joint inertia -- plus getRegressor (not timed by the reference).
 * Single thread.  out_us[10]. */
#include <time.h>
#include <stdint.h>
static uint64_t sm64_state;
static double sm64_pm1(void)
{
  uint64_t z = (sm64_state += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z ^= z >> 31;
  return (double)(z >> 11) * (1.0 / 4503599627370496.0) - 1.0;
}
static double now_us(void)
{
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}
double orc_speed_test(const orc_chain* c, int ntrial, uint64_t seed, double* out_us)
{
  int n = c->active_joints_number, L = c->links_number;
  double q[ORC_MAX_LINKS], Dq[ORC_MAX_LINKS], DDq[ORC_MAX_LINKS], DDDq[ORC_MAX_LINKS];
  double* buf = (double*)malloc(sizeof(double) * (size_t)(12 * L + 6 * n + 18 * L + n + n * n + 10 * n * c->joints_number + 64));
  double sink = 0;
  sm64_state = seed;
  for (int k = 0; k < 10; k++) out_us[k] = 0;
#define DRAW() for (int i = 0; i < n; i++) { q[i] = sm64_pm1(); Dq[i] = sm64_pm1(); DDq[i] = sm64_pm1(); DDDq[i] = sm64_pm1(); }
#define TIMED(slot, call) { DRAW(); double t0 = now_us(); call; out_us[slot] += now_us() - t0; sink += buf[0]; }
  for (int it = 0; it < ntrial; it++)
  {
    TIMED(0, orc_fk(c, q, buf));
    TIMED(1, orc_jacobian(c, q, buf));
    TIMED(2, orc_twist(c, q, Dq, buf));
    { DRAW(); orc_state s; double t0 = now_us(); computeFrames(c, &s, q); computeScrews(c, &s); getDTwistLinearPart(c, &s, DDq); out_us[3] += now_us() - t0; sink += s.Dtw_lin[L - 1].v[0]; }
    { DRAW(); orc_state s; double t0 = now_us(); computeFrames(c, &s, q); computeScrews(c, &s); getTwist(c, &s, Dq); getDTwistNonLinearPart(c, &s); out_us[4] += now_us() - t0; sink += s.Dtw_nonlin[L - 1].v[0]; }
    TIMED(5, orc_dtwist(c, q, Dq, DDq, buf, NULL, NULL));
    TIMED(6, orc_ddtwist(c, q, Dq, DDq, DDDq, buf));
    TIMED(7, orc_joint_torque(c, q, Dq, DDq, NULL, buf, NULL));
    TIMED(8, orc_joint_inertia(c, q, buf));
    TIMED(9, orc_regressor(c, q, Dq, DDq, buf));
  }
  for (int k = 0; k < 10; k++) out_us[k] /= ntrial;
  free(buf);
  return sink;
}
