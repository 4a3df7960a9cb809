// TEST INFRASTRUCTURE ONLY -- a compile-and-run stand-in for the part of urdfdom_headers' <urdf_model/model.h> that
// rosdyn_chain_facade.hpp's createChain(const urdf::ModelInterface&, ...) overload touches (urdfdom is not installed in this image).
// Member names and shapes follow urdfdom_headers 1.0 (urdf_model/{pose,joint,link,model}.h); nothing here is product code and
// nothing of the reference is built with it.
#ifndef MOCK_URDF_MODEL_MODEL_H
#define MOCK_URDF_MODEL_MODEL_H
#include <map>
#include <memory>
#include <string>
#include <vector>
namespace urdf
{
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Rotation { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Vector3 position; Rotation rotation; };
struct JointLimits { double lower = 0, upper = 0, effort = 0, velocity = 0; };
struct Joint
{
  enum { UNKNOWN, REVOLUTE, CONTINUOUS, PRISMATIC, FLOATING, PLANAR, FIXED };
  std::string name;
  int type = UNKNOWN;
  Vector3 axis;
  std::string child_link_name, parent_link_name;
  Pose parent_to_joint_origin_transform;
  std::shared_ptr<JointLimits> limits;
};
struct Inertial { Pose origin; double mass = 0, ixx = 0, ixy = 0, ixz = 0, iyy = 0, iyz = 0, izz = 0; };
struct Link;
typedef std::shared_ptr<Link> LinkSharedPtr;
typedef std::shared_ptr<const Link> LinkConstSharedPtr;
typedef std::shared_ptr<Joint> JointSharedPtr;
struct Link
{
  std::string name;
  std::shared_ptr<Inertial> inertial;
  JointSharedPtr parent_joint;
  std::vector<JointSharedPtr> child_joints;
  std::vector<LinkSharedPtr> child_links;
  LinkSharedPtr getParent() const { return parent_link_.lock(); }
  void setParent(const LinkSharedPtr& p) { parent_link_ = p; }
private:
  std::weak_ptr<Link> parent_link_;
};
class ModelInterface
{
public:
  LinkConstSharedPtr getLink(const std::string& name) const
  {
    auto it = links_.find(name);
    return it == links_.end() ? LinkConstSharedPtr() : LinkConstSharedPtr(it->second);
  }
  std::map<std::string, LinkSharedPtr> links_;
  std::map<std::string, JointSharedPtr> joints_;
  LinkSharedPtr root_link_;
};
}  // namespace urdf
#endif
