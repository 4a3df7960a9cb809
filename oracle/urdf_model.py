"""URDF front end of the ORACLE -- test infrastructure, not product code.

Restates, independently of the product's C++ reader (rosdyn_amd/csrc/rdyn_urdf.cpp), what the
reference gets from urdfdom + its own tree walk:

* urdfdom semantics the reference silently relies on (third-party, source absent from
  /root/reference; ROS-noetic system urdfdom, version unpinned -- published behaviour restated):
  - ``rpy`` -> quaternion by the half-angle formula of ``urdf::Rotation::setFromRPY`` + normalisation;
  - missing ``<origin>`` -> identity pose; missing ``<axis>`` on a non-fixed, non-floating joint -> (1,0,0);
    fixed / floating joints keep axis (0,0,0);
  - missing ``<inertial>`` -> ``link->inertial == NULL``;
  - ``child_joints`` of a link are ordered by joint NAME (urdfdom builds the tree from a std::map).
* ``Link::fromUrdf`` recursion / ``Link::findChild`` (primitives_impl.h:276-286, 424-440)
* ``Chain::init`` tool->base walk (primitives_impl.h:600-636) and the default input order =
  moveable joints base->tool (primitives_impl.h:634-635, 700), ``setInputJointsName`` (705-737).
"""
import math
import xml.etree.ElementTree as ET

URDF_TYPES = {"revolute": 0, "continuous": 1, "prismatic": 2, "fixed": 3, "floating": 4, "planar": 5}


def _floats(text, n, default):
    if text is None:
        return list(default)
    vals = [float(t) for t in text.split()]
    if len(vals) != n:
        raise ValueError("expected %d numbers, got %r" % (n, text))
    return vals


def rpy_to_quat(roll, pitch, yaw):
    """urdf::Rotation::setFromRPY (urdfdom_headers pose.h): returns (x, y, z, w), normalised."""
    phi, the, psi = roll / 2.0, pitch / 2.0, yaw / 2.0
    x = math.sin(phi) * math.cos(the) * math.cos(psi) - math.cos(phi) * math.sin(the) * math.sin(psi)
    y = math.cos(phi) * math.sin(the) * math.cos(psi) + math.sin(phi) * math.cos(the) * math.sin(psi)
    z = math.cos(phi) * math.cos(the) * math.sin(psi) - math.sin(phi) * math.sin(the) * math.cos(psi)
    w = math.cos(phi) * math.cos(the) * math.cos(psi) + math.sin(phi) * math.sin(the) * math.sin(psi)
    s = math.sqrt(x * x + y * y + z * z + w * w)
    if s == 0.0:
        return (0.0, 0.0, 0.0, 1.0)
    return (x / s, y / s, z / s, w / s)


def _pose(elem):
    """-> (xyz, quaternion x,y,z,w, rpy); rpy is kept only for the independent numpy restatement."""
    if elem is None:
        return [0.0, 0.0, 0.0], (0.0, 0.0, 0.0, 1.0), [0.0, 0.0, 0.0]
    xyz = _floats(elem.get("xyz"), 3, (0.0, 0.0, 0.0))
    rpy = _floats(elem.get("rpy"), 3, (0.0, 0.0, 0.0))
    return xyz, rpy_to_quat(*rpy), rpy


class UJoint(object):
    def __init__(self, e):
        self.name = e.get("name")
        self.type_name = e.get("type")
        self.urdf_type = URDF_TYPES.get(self.type_name, 6)
        self.parent = e.find("parent").get("link")
        self.child = e.find("child").get("link")
        self.xyz, self.quat, self.rpy = _pose(e.find("origin"))
        ax = e.find("axis")
        if self.type_name in ("fixed", "floating"):
            self.axis = [0.0, 0.0, 0.0]
        elif ax is None:
            self.axis = [1.0, 0.0, 0.0]
        else:
            self.axis = _floats(ax.get("xyz"), 3, (1.0, 0.0, 0.0))
        lim = e.find("limit")
        self.limits = None
        if lim is not None:
            self.limits = dict(lower=float(lim.get("lower", 0.0)), upper=float(lim.get("upper", 0.0)),
                               effort=float(lim.get("effort", 0.0)), velocity=float(lim.get("velocity", 0.0)))


class ULink(object):
    def __init__(self, e):
        self.name = e.get("name")
        self.child_joints = []
        self.parent_joint = None
        ine = e.find("inertial")
        self.has_inertial = ine is not None
        self.mass = 0.0
        self.xyz, self.quat, self.rpy = [0.0, 0.0, 0.0], (0.0, 0.0, 0.0, 1.0), [0.0, 0.0, 0.0]
        self.inertia = [0.0] * 6
        if ine is not None:
            self.xyz, self.quat, self.rpy = _pose(ine.find("origin"))
            m = ine.find("mass")
            self.mass = float(m.get("value")) if m is not None else 0.0
            it = ine.find("inertia")
            if it is not None:
                self.inertia = [float(it.get(k, 0.0)) for k in ("ixx", "ixy", "ixz", "iyy", "iyz", "izz")]


class UModel(object):
    def __init__(self, xml_text):
        root = ET.fromstring(xml_text)
        self.links = {}
        for e in root.findall("link"):
            l = ULink(e)
            self.links[l.name] = l
        self.joints = {}
        for e in root.findall("joint"):
            j = UJoint(e)
            self.joints[j.name] = j
        for name in sorted(self.joints):  # std::map iteration order
            j = self.joints[name]
            self.links[j.parent].child_joints.append(j)
            self.links[j.child].parent_joint = j
        roots = [l for l in self.links.values() if l.parent_joint is None]
        if len(roots) != 1:
            raise ValueError("URDF must have exactly one root link")
        self.root = roots[0]

    # Link::findChild, primitives_impl.h:424-440 (depth-first, first match)
    def find_child(self, start, name):
        if start.name == name:
            return start
        for j in start.child_joints:
            c = self.links[j.child]
            if c.name == name:
                return c
            r = self.find_child(c, name)
            if r is not None:
                return r
        return None


class ChainSpec(object):
    """Ordered base->tool joints/links of one serial chain + the input map (what Chain::init produces)."""

    def __init__(self, model, base, tool, gravity=(0.0, 0.0, 0.0), input_joint_names=None):
        base_link = model.find_child(model.root, base)
        if base_link is None:
            raise RuntimeError("Base link not found")      # primitives_impl.h:603
        ee = model.find_child(base_link, tool)
        if ee is None:
            raise RuntimeError("Tool link not found")      # primitives_impl.h:610
        links, joints = [], []
        act = ee
        while True:                                         # primitives_impl.h:616-626
            links.insert(0, act)
            if act.name != base:
                joints.insert(0, act.parent_joint)
                act = model.links[act.parent_joint.parent]
            else:
                break
        self.links, self.joints = links, joints
        self.gravity = tuple(float(g) for g in gravity)
        self.link_names = [l.name for l in links]
        self.joint_names = [j.name for j in joints]
        self.moveable = [j.name for j in joints if j.urdf_type in (0, 1, 2)]   # !isFixed(), primitives_impl.h:634
        names = self.moveable if input_joint_names is None else list(input_joint_names)
        self.input_chain_index, self.active_names = [], []
        for nm in names:                                    # primitives_impl.h:724-737
            if nm in self.joint_names:
                self.input_chain_index.append(self.joint_names.index(nm))
                self.active_names.append(nm)
        self.n_joints = len(joints)
        self.n_links = len(links)
        self.n_active = len(self.input_chain_index)
        self.n_params = 10 * self.n_joints
        # position limits of the input joints (Joint::fromUrdf, primitives_impl.h:85-143; Chain, :768-776)
        self.q_min, self.q_max = [], []
        for c in self.input_chain_index:
            j = joints[c]
            lo, hi = -1e10, 1e10
            if j.urdf_type in (0, 2) and j.limits is not None:
                lo, hi = j.limits["lower"], j.limits["upper"]
                if hi <= lo:
                    lo, hi = -2 * math.pi, 2 * math.pi
            self.q_min.append(lo)
            self.q_max.append(hi)


def load(path_or_xml, base, tool, gravity=(0.0, 0.0, 0.0), input_joint_names=None):
    text = path_or_xml
    if "<robot" not in path_or_xml:
        with open(path_or_xml) as f:
            text = f.read()
    return ChainSpec(UModel(text), base, tool, gravity, input_joint_names)
