// Compiled by tests/test_facade.py with -Itests/mock_include: the Eigen-typed and urdfdom-typed branches of the facade
// (RDYN_FACADE_HAS_EIGEN, RDYN_FACADE_HAS_URDFDOM).  Host-only checks (no GPU call): a chain built from an in-memory urdf model
// through createChain(const urdf::ModelInterface&, ...) must equal the chain the library builds from the same robot as XML.
#include <cmath>
#include <cstdio>
#include "rosdyn_chain_facade.hpp"

#if !defined(RDYN_FACADE_HAS_EIGEN) || !defined(RDYN_FACADE_HAS_URDFDOM)
#error "compile with -Itests/mock_include"
#endif

static urdf::LinkSharedPtr add_link(urdf::ModelInterface& m, const char* name, double mass, double cx, double ixx)
{
  auto l = std::make_shared<urdf::Link>();
  l->name = name;
  if (mass > 0)
  {
    l->inertial = std::make_shared<urdf::Inertial>();
    l->inertial->mass = mass;
    l->inertial->origin.position.x = cx;
    l->inertial->ixx = ixx; l->inertial->iyy = 2 * ixx; l->inertial->izz = 3 * ixx;
  }
  m.links_[name] = l;
  return l;
}
static void add_joint(urdf::ModelInterface& m, const char* name, int type, urdf::LinkSharedPtr p, urdf::LinkSharedPtr c, double x, double z, double ax, double az)
{
  auto j = std::make_shared<urdf::Joint>();
  j->name = name;
  j->type = type;
  j->parent_link_name = p->name;
  j->child_link_name = c->name;
  j->parent_to_joint_origin_transform.position.x = x;
  j->parent_to_joint_origin_transform.position.z = z;
  j->axis.x = ax; j->axis.z = az;
  if (type != urdf::Joint::FIXED)
  {
    j->limits = std::make_shared<urdf::JointLimits>();
    j->limits->lower = -2; j->limits->upper = 2; j->limits->velocity = 1.5; j->limits->effort = 30;
  }
  c->parent_joint = j;
  c->setParent(p);
  p->child_joints.push_back(j);
  p->child_links.push_back(c);
  m.joints_[name] = j;
}

int main()
{
  urdf::ModelInterface m;
  auto b = add_link(m, "base", 0, 0, 0), l1 = add_link(m, "l1", 2.0, 0.1, 0.01), l2 = add_link(m, "l2", 1.5, 0.2, 0.02),
       l3 = add_link(m, "l3", 0.5, 0.05, 0.005), tool = add_link(m, "tool", 0, 0, 0);
  m.root_link_ = b;
  add_joint(m, "j1", urdf::Joint::REVOLUTE, b, l1, 0.0, 0.3, 0, 1);
  add_joint(m, "j2", urdf::Joint::PRISMATIC, l1, l2, 0.4, 0.0, 1, 0);
  add_joint(m, "j3", urdf::Joint::CONTINUOUS, l2, l3, 0.2, 0.1, 0, 2);   // axis not normalised
  add_joint(m, "jt", urdf::Joint::FIXED, l3, tool, 0.1, 0.0, 0, 0);
  const char* xml =
      "<robot name='r'><link name='base'/>"
      "<link name='l1'><inertial><origin xyz='0.1 0 0'/><mass value='2.0'/><inertia ixx='0.01' ixy='0' ixz='0' iyy='0.02' iyz='0' izz='0.03'/></inertial></link>"
      "<link name='l2'><inertial><origin xyz='0.2 0 0'/><mass value='1.5'/><inertia ixx='0.02' ixy='0' ixz='0' iyy='0.04' iyz='0' izz='0.06'/></inertial></link>"
      "<link name='l3'><inertial><origin xyz='0.05 0 0'/><mass value='0.5'/><inertia ixx='0.005' ixy='0' ixz='0' iyy='0.01' iyz='0' izz='0.015'/></inertial></link>"
      "<link name='tool'/>"
      "<joint name='j1' type='revolute'><parent link='base'/><child link='l1'/><origin xyz='0 0 0.3'/><axis xyz='0 0 1'/><limit lower='-2' upper='2' velocity='1.5' effort='30'/></joint>"
      "<joint name='j2' type='prismatic'><parent link='l1'/><child link='l2'/><origin xyz='0.4 0 0'/><axis xyz='1 0 0'/><limit lower='-2' upper='2' velocity='1.5' effort='30'/></joint>"
      "<joint name='j3' type='continuous'><parent link='l2'/><child link='l3'/><origin xyz='0.2 0 0.1'/><axis xyz='0 0 2'/><limit lower='-2' upper='2' velocity='1.5' effort='30'/></joint>"
      "<joint name='jt' type='fixed'><parent link='l3'/><child link='tool'/><origin xyz='0.1 0 0'/></joint></robot>";
  Eigen::Vector3d g;
  g(0) = 0; g(1) = 0; g(2) = -9.806;
  rosdyn::ChainPtr a = rosdyn::createChain(m, "base", "tool", g);                       // the reference's signature, primitives.h:566
  rosdyn::ChainPtr c = rosdyn::createChain(std::string(xml), "base", "tool", {0.0, 0.0, -9.806});
  int bad = 0;
  bad += a->getLinksNumber() != c->getLinksNumber() || a->getJointsNumber() != 4 || a->getActiveJointsNumber() != 3;
  bad += a->getActiveJointsName() != c->getActiveJointsName() || a->getLinksName() != c->getLinksName();
  const rosdyn::VectorXd pa = a->getNominalParameters(), pc = c->getNominalParameters();
  for (int i = 0; i < pa.rows(); ++i) bad += std::fabs(pa(i) - pc(i)) > 1e-15;
  for (int i = 0; i < 3; ++i) bad += a->getQMax()(i) != c->getQMax()(i) || a->getDQMax()(i) != c->getDQMax()(i) || a->getTauMax()(i) != c->getTauMax()(i);
  try { rosdyn::createChain(m, "nope", "tool", g); ++bad; } catch (const std::runtime_error& e) { bad += std::string(e.what()) != "Base link not found"; }
  try { rosdyn::createChain(m, "base", "nope", g); ++bad; } catch (const std::runtime_error& e) { bad += std::string(e.what()) != "Tool link not found"; }
  // getMultiplicity on Eigen types (host only): the continuous joint j3 has the reference's +-1e10 limits -> refused;
  // without it (input joints j1, j2) q has no other turn inside [-2, 2]: exactly one vector
  rosdyn::VectorXd q(3);
  q(0) = 0.1; q(1) = 0.2; q(2) = 0.3;
  try { a->getMultiplicity(q); ++bad; } catch (const std::invalid_argument&) {}
  bad += !a->setInputJointsName({"j1", "j2"});
  rosdyn::VectorXd q2(2);
  q2(0) = 0.1; q2(1) = 0.2;
  bad += a->getMultiplicity(q2).size() != 1;
  // the object tree (rosdyn::Link / rosdyn::Joint views, primitives.h:62-232): names, kinds, parent / child wiring, parameters
  const auto& links = c->getLinks();
  const auto& joints = c->getJoints();
  bad += links.size() != 5 || joints.size() != 4;
  bad += links[1]->getName() != "l1" || joints[1]->getName() != "j2" || joints[1]->getType() != rosdyn::Joint::PRISMATIC || !joints[3]->isFixed();
  bad += joints[2]->getParentLink() != links[2] || joints[2]->getChildLink() != links[3] || links[3]->getParentJoint() != joints[2];
  bad += links[0]->getParentJoint() != nullptr || links[4]->getChildrenJoints().size() != 0 || links[1]->getChildrenJoints()[0] != joints[1];
  bad += joints[0]->getQMax() != 2.0 || joints[0]->getDQMax() != 1.5 || joints[0]->getTauMax() != 30.0;
  for (int l = 1; l < 5; ++l)  // Chain::getNominalParameters = the links' parameters, base link excluded (primitives_impl.h:1382-1391)
  {
    const rosdyn::VectorXd p = links[(size_t)l]->getNominalParameters();
    for (int k = 0; k < 10; ++k) bad += p(k) != pc(10 * (l - 1) + k);
  }
  bad += links[2]->getMass() != 1.5 || links[2]->getCog()(0) != 0.2;
  {
    // spatial inertia about the link origin (spacevect_algebra.h:232-239): [m 1, m skew(c)^T; m skew(c), I + m skew(c) skew(c)^T]
    const rosdyn::Matrix66d& I = links[2]->getSpatialInertia();
    const double m2 = 1.5, cx = 0.2;
    bad += I(0, 0) != m2 || I(1, 1) != m2 || I(2, 2) != m2;
    bad += std::fabs(I(4, 2) - (-m2 * cx)) > 1e-15 || std::fabs(I(5, 1) - (m2 * cx)) > 1e-15 || std::fabs(I(2, 4) - I(4, 2)) > 0 || std::fabs(I(1, 5) - I(5, 1)) > 0;
    bad += std::fabs(I(3, 3) - 0.02) > 1e-15 || std::fabs(I(4, 4) - (0.04 + m2 * cx * cx)) > 1e-15 || std::fabs(I(5, 5) - (0.06 + m2 * cx * cx)) > 1e-15;
    bad += links[2]->getSpatialInertiaTerms().size() != 10;
  }
  {
    // Joint::getTransformation (primitives_impl.h:38-47): j1 turns about z at height 0.3; j2 slides along x from x = 0.4; j3's axis is normalised
    const rosdyn::Affine3d& T1 = joints[0]->getTransformation(0.5);
    bad += std::fabs(T1.matrix()(0, 0) - std::cos(0.5)) > 1e-15 || std::fabs(T1.matrix()(1, 0) - std::sin(0.5)) > 1e-15 || std::fabs(T1.matrix()(2, 3) - 0.3) > 0;
    const rosdyn::Affine3d& T2 = joints[1]->getTransformation(0.25);
    bad += std::fabs(T2.matrix()(0, 3) - 0.65) > 1e-15 || T2.matrix()(0, 0) != 1.0;
    const rosdyn::Vector6d& s3 = joints[2]->getScrew_of_child_in_parent();
    bad += s3(5) != 1.0 || s3(0) != 0.0;
    const rosdyn::Vector6d& s2 = joints[1]->getScrew_of_child_in_parent();
    bad += s2(0) != 1.0 || s2(3) != 0.0;
  }
  {
    // rigid-body reduction through the facade: the fixed tool joint folds "tool" into the body of j3 (chain index 2); X of a link that
    // is its body's own reference frame is the identity, the merged parameters of the last body include the (massless) tool link
    std::vector<int32_t> body;
    std::vector<double> X;
    rosdyn::VectorXd pib;
    bad += c->getBodyReduction(body, X, pib) != 3 || body.size() != 4 || body[0] != 0 || body[1] != 1 || body[2] != 2 || body[3] != 2;
    for (int a = 0; a < 10 && !bad; ++a)
      for (int p = 0; p < 10; ++p) bad += X[(size_t)2 * 100 + a * 10 + p] != (a == p ? 1.0 : 0.0);
    for (int k = 0; k < 10 && !bad; ++k) bad += std::fabs(pib(20 + k) - pc(20 + k)) > 1e-15;   // tool has no inertia: body 3 = link l3
  }
  {
    // the factor's report for a batch below the preconditioned route's threshold: route 0 (Householder folds), decided on the host
    double dummy = 0.0;
    const rdyn_tsqr_report rep = c->getTsqrReport(1000, &dummy);
    bad += rep.route != 0 || rep.stage != 0 || rep.n_deferred != 0;
  }
  std::printf("facade typed surface: %s (%u links, %u joints, %u active, %d parameters)\n", bad ? "MISMATCH" : "ok", a->getLinksNumber(),
              a->getJointsNumber(), a->getActiveJointsNumber(), (int)pa.rows());
  return bad ? 1 : 0;
}
