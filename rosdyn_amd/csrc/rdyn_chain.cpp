// rdyn_chain.cpp -- chain ingest (host): urdf / flat description -> per-joint constants for the kernels.
//
// Follows the construction path of the reference:
//   Joint::fromUrdf            primitives_impl.h:50-149   (pose, axis normalisation, type map, limits)
//   Link::fromUrdf             primitives_impl.h:288-328  (inertia rotation, spatial inertia about the link origin)
//   Link::getNominalParameters primitives_impl.h:399-417
//   Chain::init                primitives_impl.h:580-703  (ordering, default input joints = moveable joints)
//   Chain::setInputJointsName  primitives_impl.h:705-737
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include <hip/hip_runtime.h>

#include "rdyn_chain.hpp"

static thread_local char g_err[512] = "";

void rdyn_set_error(const char* fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

extern "C" const char* rdyn_last_error(void) { return g_err; }

namespace
{
// Eigen::Quaterniond(w,x,y,z).toRotationMatrix() as used by urdf_parser.h:44-50 / primitives_impl.h:317; row-major out
void quat_to_R(const double q[4], double R[9])
{
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
void mat3_mul(const double* a, const double* b, double* r)
{
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
    {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += a[i * 3 + k] * b[k * 3 + j];
      r[i * 3 + j] = s;
    }
}
void skew3(const double* v, double* K)
{
  K[0] = 0;     K[1] = -v[2]; K[2] = v[1];
  K[3] = v[2];  K[4] = 0;     K[5] = -v[0];
  K[6] = -v[1]; K[7] = v[0];  K[8] = 0;
}
// Link::fromUrdf + Link::getNominalParameters (primitives_impl.h:288-328, 399-417): [m, m c, Ixx Ixy Ixz Iyy Iyz Izz about the link origin]
void link_parameters(const rdyn_link_desc& L, double pi[10], double* mass, double cog[3])
{
  double m = 0, cg[3] = {0, 0, 0}, I[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (L.has_inertial)
  {
    m = L.mass;
    for (int i = 0; i < 3; ++i) cg[i] = L.com_xyz[i];
    const double I0[9] = {L.ixx, L.ixy, L.ixz, L.ixy, L.iyy, L.iyz, L.ixz, L.iyz, L.izz};
    double Rc[9], RcT[9], tmp[9];
    quat_to_R(L.com_quat, Rc);
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) RcT[a * 3 + b] = Rc[b * 3 + a];
    mat3_mul(Rc, I0, tmp);
    mat3_mul(tmp, RcT, I);  // primitives_impl.h:317
  }
  double cs[9], csT[9], cc[9];
  skew3(cg, cs);
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) csT[a * 3 + b] = cs[b * 3 + a];
  mat3_mul(cs, csT, cc);  // spacevect_algebra.h:238
  double Io[9];
  for (int i = 0; i < 9; ++i) Io[i] = I[i] + m * cc[i];
  pi[0] = m;
  pi[1] = cg[0] * m;
  pi[2] = cg[1] * m;
  pi[3] = cg[2] * m;
  pi[4] = Io[0];
  pi[5] = Io[1];
  pi[6] = Io[2];
  pi[7] = Io[4];
  pi[8] = Io[5];
  pi[9] = Io[8];
  if (mass) *mass = m;
  if (cog)
    for (int i = 0; i < 3; ++i) cog[i] = cg[i];
}

int map_type(int urdf_type)
{
  // primitives_impl.h:74-83
  if (urdf_type == RDYN_URDF_REVOLUTE || urdf_type == RDYN_URDF_CONTINUOUS) return RDYN_REVOLUTE;
  if (urdf_type == RDYN_URDF_PRISMATIC) return RDYN_PRISMATIC;
  return RDYN_FIXED;
}

void free_device_copies(rdyn_chain* c)
{
  for (auto& kv : c->dev_const)
  {
    int prev = -1;
    if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(kv.first) == hipSuccess)
    {
      (void)hipFree(kv.second);
      (void)hipSetDevice(prev);
    }
  }
  c->dev_const.clear();
  for (auto& kv : c->dev_long)
  {
    int prev = -1;
    if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(kv.first) == hipSuccess)
    {
      (void)hipFree(kv.second);
      (void)hipSetDevice(prev);
    }
  }
  c->dev_long.clear();
  for (auto& kv : c->dev_expand)
  {
    int prev = -1;
    if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(kv.first) == hipSuccess)
    {
      (void)hipFree(kv.second);
      (void)hipSetDevice(prev);
    }
  }
  c->dev_expand.clear();
  if (c->reduced) free_device_copies(c->reduced.get());
  if (c->sorted) free_device_copies(c->sorted.get());
}

// inertial parameters [m, m c, Ixx Ixy Ixz Iyy Iyz Izz about the frame origin] of a body given in frame f, re-expressed in frame r,
// where x_r = R x_f + p:  m' = m;  (m c)' = R (m c) + m p;
// I' = R I R' + (2 p.(R mc)) 1 - (R mc) p' - p (R mc)' + m (|p|^2 1 - p p')      (parallel axes about the new origin)
void transform_parameters(const double R[9], const double p[3], const double in[10], double out[10])
{
  const double m = in[0];
  double h[3];
  for (int i = 0; i < 3; ++i) h[i] = R[i * 3] * in[1] + R[i * 3 + 1] * in[2] + R[i * 3 + 2] * in[3];
  const double I[9] = {in[4], in[5], in[6], in[5], in[7], in[8], in[6], in[8], in[9]};
  double RI[9], RIRt[9], Rt[9];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) Rt[a * 3 + b] = R[b * 3 + a];
  mat3_mul(R, I, RI);
  mat3_mul(RI, Rt, RIRt);
  const double hp = h[0] * p[0] + h[1] * p[1] + h[2] * p[2], pp = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
  double Io[9];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b)
      Io[a * 3 + b] = RIRt[a * 3 + b] + (a == b ? 2 * hp + m * pp : 0.0) - h[a] * p[b] - p[a] * h[b] - m * p[a] * p[b];
  out[0] = m;
  out[1] = h[0] + m * p[0];
  out[2] = h[1] + m * p[1];
  out[3] = h[2] + m * p[2];
  out[4] = Io[0];
  out[5] = Io[1];
  out[6] = Io[2];
  out[7] = Io[4];
  out[8] = Io[5];
  out[9] = Io[8];
}

// see rdyn_chain.hpp: the copy of a chain (host_joints / host_const / active already final) with its input joints renumbered in chain order
void build_sorted_view(rdyn_chain* c)
{
  c->sorted.reset();
  const int nj = c->n_joints(), n = c->n_active();
  c->row_input.resize(n);
  c->input_row.resize(n);
  bool monotonic = true;
  int rank = 0;
  for (int f = 0; f < nj; ++f)
  {
    const int k = c->host_joints[f].in_idx;
    if (k < 0) continue;
    monotonic = monotonic && k == rank;
    c->row_input[rank] = k;
    c->input_row[k] = rank;
    ++rank;
  }
  if (monotonic || c->long_chain()) return;
  std::unique_ptr<rdyn_chain> s(new rdyn_chain());
  s->joints = c->joints;
  s->links = c->links;
  s->moveable_names = c->moveable_names;
  memcpy(s->gravity, c->gravity, sizeof s->gravity);
  s->q_max = c->q_max;
  s->q_min = c->q_min;
  s->dq_max = c->dq_max;
  s->ddq_max = c->ddq_max;
  s->tau_max = c->tau_max;
  s->host_joints = c->host_joints;
  s->host_const = c->host_const;
  s->active.resize(n);
  for (int f = 0; f < nj; ++f)
  {
    const int k = c->host_joints[f].in_idx;
    if (k < 0) continue;
    s->host_joints[f].in_idx = s->host_const.j[f].in_idx = c->input_row[k];
    s->active[c->input_row[k]] = f;
  }
  s->row_input = c->row_input;
  s->input_row = c->input_row;
  c->sorted = std::move(s);
}

// see rdyn_chain.hpp: the reduced companion and the expansion blocks X_f
void build_reduced(rdyn_chain* c)
{
  free_device_copies(c);  // also the old companion's
  c->reduced.reset();
  c->red_of.clear();
  c->expand_X.clear();
  const int nj = c->n_joints(), n = c->n_active();
  c->red_chain.clear();
  if (n < 1 || n == nj) return;
  if (n > RDYN_MAX_SWEPT_JOINTS) return;  // the companion itself would be longer than what the kernels sweep
  // (the input joints may come in any order, primitives_impl.h:705-737: the companion keeps them in CHAIN order and carries the
  // input index of each as its in_idx -- the kernels read q, Dq, DDq and write their rows through that map)
  const std::vector<RdynJointConst>& HJ = c->host_joints;
  std::unique_ptr<rdyn_chain> r(new rdyn_chain());
  r->joints.resize(n);
  r->links.resize(n + 1);
  memset(r->joints.data(), 0, sizeof(rdyn_joint_desc) * n);
  memset(r->links.data(), 0, sizeof(rdyn_link_desc) * (n + 1));
  for (int i = 0; i < 3; ++i) r->gravity[i] = c->gravity[i];
  RdynChainConst& K = r->host_const;
  memset(&K, 0, sizeof K);
  K.n_joints = K.n_active = n;
  for (int i = 0; i < 3; ++i) K.g[i] = c->gravity[i];
  c->red_of.assign(nj, -1);
  c->expand_X.assign((size_t)nj * 100, 0.0);
  r->active.assign(n, 0);
  r->q_max.assign(n, 1e10);
  r->q_min.assign(n, -1e10);
  r->dq_max.assign(n, 1e10);
  r->ddq_max.assign(n, 1e11);
  r->tau_max.assign(n, 1e10);
  // frame of the current rigid body -> frame of the link being visited: x_body = Rc x_link + pc
  double Rc[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pc[3] = {0, 0, 0};
  int red = -1;  // reduced link the current body is (index of its joint in the reduced chain)
  for (int f = 0; f < nj; ++f)
  {
    const RdynJointConst& J = HJ[f];
    // compose the parent -> joint transform of chain joint f (R_pj, t_pj) onto the running transform
    double Rn[9], pn[3];
    mat3_mul(Rc, J.A, Rn);
    for (int i = 0; i < 3; ++i) pn[i] = pc[i] + Rc[i * 3] * J.t[0] + Rc[i * 3 + 1] * J.t[1] + Rc[i * 3 + 2] * J.t[2];
    if (J.in_idx >= 0)
    {
      // input joint: it becomes reduced joint red + 1 with everything since the previous input joint folded into its origin
      ++red;
      RdynJointConst& Q = K.j[red];
      double Ks[9], K2[9];
      skew3(J.u, Ks);
      mat3_mul(Ks, Ks, K2);
      memcpy(Q.A, Rn, sizeof Rn);
      mat3_mul(Rn, Ks, Q.B);
      mat3_mul(Rn, K2, Q.C);
      for (int i = 0; i < 3; ++i)
      {
        Q.t[i] = pn[i];
        Q.u[i] = J.u[i];
        Q.up[i] = Rn[i * 3] * J.u[0] + Rn[i * 3 + 1] * J.u[1] + Rn[i * 3 + 2] * J.u[2];
      }
      Q.type = J.type;
      Q.in_idx = J.in_idx;
      snprintf(r->joints[red].name, sizeof r->joints[red].name, "%s", c->joints[f].name);
      snprintf(r->links[red + 1].name, sizeof r->links[red + 1].name, "%s", c->links[f + 1].name);
      r->active[J.in_idx] = red;
      r->moveable_names.push_back(c->joints[f].name);
      c->red_chain.push_back(f);
      r->q_max[red] = c->q_max[f];
      r->q_min[red] = c->q_min[f];
      r->dq_max[red] = c->dq_max[f];
      r->ddq_max[red] = c->ddq_max[f];
      r->tau_max[red] = c->tau_max[f];
      // the child link of an input joint IS the new body's reference frame
      const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      memcpy(Rc, I3, sizeof I3);
      pc[0] = pc[1] = pc[2] = 0.0;
    }
    else
    {
      memcpy(Rc, Rn, sizeof Rn);
      memcpy(pc, pn, sizeof pn);
    }
    c->red_of[f] = red;
    if (red >= 0)
    {
      // X_f: columns = images of the ten unit parameter vectors of link f + 1 in the body frame; the body's own parameters add up
      double* X = c->expand_X.data() + (size_t)f * 100;
      for (int p = 0; p < 10; ++p)
      {
        double e[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, img[10];
        e[p] = 1.0;
        transform_parameters(Rc, pc, e, img);
        for (int a = 0; a < 10; ++a) X[a * 10 + p] = img[a];
      }
      double img[10];
      transform_parameters(Rc, pc, J.pi, img);
      for (int a = 0; a < 10; ++a) K.j[red].pi[a] += img[a];
    }
  }
  r->host_joints.assign(K.j, K.j + n);
  // the frames behind the last input joint: x_lastbody = tail_R x_tool + tail_t (the tool frame of the chain seen from the companion's)
  memcpy(c->tail_R, Rc, sizeof Rc);
  memcpy(c->tail_t, pc, sizeof pc);
  build_sorted_view(r.get());
  c->reduced = std::move(r);
}
}  // namespace

void rdyn_chain_finalize(rdyn_chain* c)
{
  const int nj = c->n_joints();
  RdynChainConst& H = c->host_const;
  memset(&H, 0, sizeof H);
  c->host_joints.assign(nj, RdynJointConst());
  std::vector<RdynJointConst>& HJ = c->host_joints;
  c->q_max.assign(nj, 0.0);
  c->q_min.assign(nj, 0.0);
  c->dq_max.assign(nj, 0.0);
  c->ddq_max.assign(nj, 0.0);
  c->tau_max.assign(nj, 0.0);
  for (int j = 0; j < nj; ++j)
  {
    const rdyn_joint_desc& d = c->joints[j];
    RdynJointConst& K = HJ[j];
    memset(&K, 0, sizeof K);
    double R[9], Ks[9], K2[9];
    quat_to_R(d.origin_quat, R);
    double u[3] = {d.axis[0], d.axis[1], d.axis[2]};
    const double nrm = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    if (nrm > 0)  // primitives_impl.h:58-59
      for (int i = 0; i < 3; ++i) u[i] /= nrm;
    skew3(u, Ks);
    mat3_mul(Ks, Ks, K2);
    memcpy(K.A, R, sizeof R);
    mat3_mul(R, Ks, K.B);
    mat3_mul(R, K2, K.C);
    for (int i = 0; i < 3; ++i)
    {
      K.t[i] = d.origin_xyz[i];
      K.u[i] = u[i];
      K.up[i] = R[i * 3 + 0] * u[0] + R[i * 3 + 1] * u[1] + R[i * 3 + 2] * u[2];
    }
    K.type = map_type(d.urdf_type);
    K.in_idx = -1;
    // child link nominal parameters (Link::fromUrdf 288-328, getNominalParameters 399-417)
    link_parameters(c->links[j + 1], K.pi, nullptr, nullptr);

    // limits (primitives_impl.h:85-143).  Where the reference leaves a member uninitialised
    // (m_Dq_max without <limit>), 1e10 is used.
    double qmax = 1e10, qmin = -1e10, dqmax = 1e10, taumax = 1e10;
    if (d.urdf_type == RDYN_URDF_PRISMATIC || d.urdf_type == RDYN_URDF_REVOLUTE)
    {
      if (d.has_limits)
      {
        qmax = d.upper;
        qmin = d.lower;
        if (qmax <= qmin)
        {
          qmax = 2 * M_PI;
          qmin = -2 * M_PI;
        }
        dqmax = d.velocity;
        if (dqmax <= 0.0) dqmax = 2 * M_PI;
        taumax = d.effort;
      }
    }
    else if (d.urdf_type == RDYN_URDF_CONTINUOUS)
    {
      if (d.has_limits)
      {
        dqmax = d.velocity;
        taumax = d.effort;
      }
    }
    c->q_max[j] = qmax;
    c->q_min[j] = qmin;
    c->dq_max[j] = dqmax;
    c->ddq_max[j] = 10.0 * dqmax;
    c->tau_max[j] = taumax;
  }
  for (int k = 0; k < c->n_active(); ++k) HJ[c->active[k]].in_idx = k;
  if (!c->long_chain())
  {
    // the flat copy the kernels read (a chain longer than what they sweep is served through its reduced companion only)
    H.n_joints = nj;
    H.n_active = c->n_active();
    for (int i = 0; i < 3; ++i) H.g[i] = c->gravity[i];
    for (int j = 0; j < nj; ++j) H.j[j] = HJ[j];
  }
  if (c->long_chain())
  {
    // the run-time-length kinematic kernels (rdyn_long_kin.hip) read the chain as it is
    RdynLongChainConst& Lc = c->host_long;
    memset(&Lc, 0, sizeof Lc);
    Lc.n_joints = nj;
    Lc.n_active = c->n_active();
    for (int i = 0; i < 3; ++i) Lc.g[i] = c->gravity[i];
    for (int j = 0; j < nj; ++j) Lc.j[j] = HJ[j];
  }
  build_reduced(c);
  build_sorted_view(c);
}

static int build_chain(std::vector<rdyn_joint_desc>& joints, std::vector<rdyn_link_desc>& links, const double gravity[3], rdyn_chain** out)
{
  if ((int)joints.size() > RDYN_MAX_JOINTS)
  {
    rdyn_set_error("chain has %d joints (fixed included); this build supports at most %d (of which at most %d input joints)", (int)joints.size(),
                   RDYN_MAX_JOINTS, RDYN_MAX_SWEPT_JOINTS);
    return RDYN_ERR_UNSUPPORTED;
  }
  rdyn_chain* c = new rdyn_chain();
  c->joints.swap(joints);
  c->links.swap(links);
  for (int i = 0; i < 3; ++i) c->gravity[i] = gravity ? gravity[i] : 0.0;  // default zero, primitives.h:346
  for (int j = 0; j < c->n_joints(); ++j)
    if (map_type(c->joints[j].urdf_type) != RDYN_FIXED)
    {
      c->moveable_names.push_back(c->joints[j].name);  // primitives_impl.h:634-635
      c->active.push_back(j);                          // setInputJointsName(m_moveable_joints_name), :700
    }
  rdyn_chain_finalize(c);
  *out = c;
  return RDYN_OK;
}

extern "C"
{

int rdyn_chain_from_urdf(const char* urdf_xml, const char* base_link, const char* tool_link, const double gravity[3], rdyn_chain** out)
{
  if (!urdf_xml || !base_link || !tool_link || !out)
  {
    rdyn_set_error("rdyn_chain_from_urdf: null argument");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  *out = nullptr;
  std::vector<rdyn_joint_desc> joints;
  std::vector<rdyn_link_desc> links;
  const int st = rdyn_urdf_extract_chain(urdf_xml, base_link, tool_link, joints, links);
  if (st != RDYN_OK) return st;
  return build_chain(joints, links, gravity, out);
}

int rdyn_chain_from_desc(const rdyn_chain_desc* desc, rdyn_chain** out)
{
  if (!desc || !out || desc->n_joints < 0 || !desc->links || (desc->n_joints > 0 && !desc->joints))
  {
    rdyn_set_error("rdyn_chain_from_desc: invalid description");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  *out = nullptr;
  std::vector<rdyn_joint_desc> joints(desc->joints, desc->joints + desc->n_joints);
  std::vector<rdyn_link_desc> links(desc->links, desc->links + desc->n_joints + 1);
  for (auto& j : joints) j.name[63] = 0;
  for (auto& l : links) l.name[63] = 0;
  return build_chain(joints, links, desc->gravity, out);
}

int rdyn_chain_clone(const rdyn_chain* chain, rdyn_chain** out)
{
  if (!chain || !out)
  {
    rdyn_set_error("rdyn_chain_clone: null argument");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  rdyn_chain* c = new rdyn_chain();
  c->joints = chain->joints;
  c->links = chain->links;
  c->moveable_names = chain->moveable_names;
  c->active = chain->active;
  memcpy(c->gravity, chain->gravity, sizeof c->gravity);
  rdyn_chain_finalize(c);
  *out = c;
  return RDYN_OK;
}

void rdyn_chain_destroy(rdyn_chain* chain)
{
  if (!chain) return;
  free_device_copies(chain);
  delete chain;
}

int rdyn_chain_links_number(const rdyn_chain* c) { return c ? c->n_joints() + 1 : -1; }
int rdyn_chain_joints_number(const rdyn_chain* c) { return c ? c->n_joints() : -1; }
int rdyn_chain_active_joints_number(const rdyn_chain* c) { return c ? c->n_active() : -1; }
int rdyn_chain_moveable_joints_number(const rdyn_chain* c) { return c ? (int)c->moveable_names.size() : -1; }
const char* rdyn_chain_link_name(const rdyn_chain* c, int i) { return (c && i >= 0 && i <= c->n_joints()) ? c->links[i].name : nullptr; }
const char* rdyn_chain_joint_name(const rdyn_chain* c, int i) { return (c && i >= 0 && i < c->n_joints()) ? c->joints[i].name : nullptr; }
const char* rdyn_chain_moveable_joint_name(const rdyn_chain* c, int i)
{
  return (c && i >= 0 && i < (int)c->moveable_names.size()) ? c->moveable_names[i].c_str() : nullptr;
}
const char* rdyn_chain_active_joint_name(const rdyn_chain* c, int i)
{
  return (c && i >= 0 && i < c->n_active()) ? c->joints[c->active[i]].name : nullptr;
}
int rdyn_chain_joint_type(const rdyn_chain* c, int i) { return (c && i >= 0 && i < c->n_joints()) ? c->host_joints[i].type : -1; }
int rdyn_chain_gravity(const rdyn_chain* c, double g[3])
{
  if (!c || !g) return RDYN_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < 3; ++i) g[i] = c->gravity[i];
  return RDYN_OK;
}

int rdyn_chain_set_input_joints(rdyn_chain* c, const char* const* names, int n_names)
{
  if (!c || n_names < 0 || (n_names > 0 && !names))
  {
    rdyn_set_error("rdyn_chain_set_input_joints: invalid argument");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  std::vector<int> act;
  for (int k = 0; k < n_names; ++k)
  {
    int found = -1;
    for (int j = 0; j < c->n_joints(); ++j)
      if (names[k] && !strcmp(names[k], c->joints[j].name)) found = j;
    if (found < 0)
    {
      rdyn_set_error("Joint named '%s' not found", names[k] ? names[k] : "(null)");  // primitives_impl.h:734
      return RDYN_ERR_JOINT_NOT_FOUND;
    }
    for (int e : act)
      if (e == found)
      {
        rdyn_set_error("Joint named '%s' listed twice", names[k]);
        return RDYN_ERR_INVALID_ARGUMENT;
      }
    act.push_back(found);
  }
  std::lock_guard<std::mutex> lk(c->mu);
  c->active.swap(act);
  rdyn_chain_finalize(c);
  free_device_copies(c);
  return RDYN_OK;
}

int rdyn_chain_joint_constants(const rdyn_chain* c, int i, double R_pj[9], double t_pj[3], double axis[3], double limits[5])
{
  if (!c || i < 0 || i >= c->n_joints())
  {
    rdyn_set_error("rdyn_chain_joint_constants: invalid argument");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  const RdynJointConst& K = c->host_joints[i];
  if (R_pj) memcpy(R_pj, K.A, sizeof K.A);
  if (t_pj) memcpy(t_pj, K.t, sizeof K.t);
  if (axis) memcpy(axis, K.u, sizeof K.u);
  if (limits)
  {
    limits[0] = c->q_max[i];
    limits[1] = c->q_min[i];
    limits[2] = c->dq_max[i];
    limits[3] = c->ddq_max[i];
    limits[4] = c->tau_max[i];
  }
  return RDYN_OK;
}

int rdyn_chain_link_parameters(const rdyn_chain* c, int i, double pi[10], double* mass, double cog[3])
{
  if (!c || i < 0 || i > c->n_joints())
  {
    rdyn_set_error("rdyn_chain_link_parameters: invalid argument");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  double tmp[10];
  link_parameters(c->links[i], pi ? pi : tmp, mass, cog);
  return RDYN_OK;
}

int rdyn_chain_limits(const rdyn_chain* c, double* q_max, double* q_min, double* dq_max, double* ddq_max, double* tau_max)
{
  if (!c) return RDYN_ERR_INVALID_ARGUMENT;
  for (int k = 0; k < c->n_active(); ++k)  // primitives_impl.h:768-776
  {
    const int j = c->active[k];
    if (q_max) q_max[k] = c->q_max[j];
    if (q_min) q_min[k] = c->q_min[j];
    if (dq_max) dq_max[k] = c->dq_max[j];
    if (ddq_max) ddq_max[k] = c->ddq_max[j];
    if (tau_max) tau_max[k] = c->tau_max[j];
  }
  return RDYN_OK;
}

int rdyn_chain_reduction(const rdyn_chain* c, int32_t* body_joint, double* X, double* pi_body)
{
  if (!c) return -1;
  if (!c->reduced) return 0;
  const int nj = c->n_joints(), nb = c->reduced->n_joints();
  for (int f = 0; f < nj; ++f)
  {
    if (body_joint) body_joint[f] = c->red_of[f] < 0 ? -1 : c->red_chain[c->red_of[f]];
    if (X) memcpy(X + (size_t)f * 100, c->expand_X.data() + (size_t)f * 100, sizeof(double) * 100);
  }
  for (int r = 0; pi_body && r < nb; ++r)
    for (int p = 0; p < 10; ++p) pi_body[10 * r + p] = c->reduced->host_joints[r].pi[p];
  return nb;
}

int rdyn_nominal_parameters(const rdyn_chain* c, double* pi)
{
  if (!c || !pi) return RDYN_ERR_INVALID_ARGUMENT;
  for (int j = 0; j < c->n_joints(); ++j)  // primitives_impl.h:1382-1391
    for (int p = 0; p < 10; ++p) pi[10 * j + p] = c->host_joints[j].pi[p];
  return RDYN_OK;
}

}  // extern "C"
