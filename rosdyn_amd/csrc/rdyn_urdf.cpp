// rdyn_urdf.cpp -- minimal URDF reader: robot_description XML -> ordered base->tool chain description.
//
// Replaces, for this path only, what the reference obtains from urdfdom (`urdf::Model`, un-vendored
// third party) plus its own tree recursion and walk:
//   Link::fromUrdf recursion      primitives_impl.h:276-286
//   Link::findChild               primitives_impl.h:424-440   (depth-first, first match)
//   Chain::init tool->base walk   primitives_impl.h:600-626
// urdfdom behaviour reproduced (published semantics; source not in /root/reference):
//   rpy -> quaternion (urdf::Rotation::setFromRPY half-angle formula + normalisation);
//   missing <origin> -> identity; missing <axis> on a joint that is neither fixed nor floating -> (1,0,0);
//   fixed/floating joints keep axis (0,0,0); missing <inertial> -> has_inertial = 0;
//   a link's child joints are ordered by joint name (std::map iteration in urdfdom's initTree).
// Only the subset of XML that URDF files use is understood: elements, attributes, comments,
// processing instructions, DOCTYPE, CDATA (skipped) and character data (ignored).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "rdyn_chain.hpp"

namespace
{

struct XmlNode
{
  std::string name;
  std::vector<std::pair<std::string, std::string>> attrs;
  std::vector<std::unique_ptr<XmlNode>> children;
  const std::string* attr(const char* k) const
  {
    for (auto& a : attrs)
      if (a.first == k) return &a.second;
    return nullptr;
  }
  const XmlNode* child(const char* k) const
  {
    for (auto& c : children)
      if (c->name == k) return c.get();
    return nullptr;
  }
};

struct XmlParser
{
  const char* p;
  const char* end;
  std::string err;

  explicit XmlParser(const char* text) : p(text), end(text + strlen(text)) {}
  bool starts(const char* s) const
  {
    size_t n = strlen(s);
    return (size_t)(end - p) >= n && memcmp(p, s, n) == 0;
  }
  bool skip_until(const char* s)
  {
    size_t n = strlen(s);
    while ((size_t)(end - p) >= n)
    {
      if (memcmp(p, s, n) == 0)
      {
        p += n;
        return true;
      }
      ++p;
    }
    err = std::string("unterminated construct, expected '") + s + "'";
    return false;
  }
  void skip_ws()
  {
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
  }
  static bool name_char(char c) { return isalnum((unsigned char)c) || c == '_' || c == '-' || c == ':' || c == '.'; }

  // skips text, comments, PIs, doctype, cdata; stops at '<' of an element start/end or at EOF
  bool skip_misc()
  {
    while (p < end)
    {
      if (*p != '<')
      {
        ++p;
        continue;
      }
      if (starts("<!--"))
      {
        if (!skip_until("-->")) return false;
      }
      else if (starts("<?"))
      {
        if (!skip_until("?>")) return false;
      }
      else if (starts("<![CDATA["))
      {
        if (!skip_until("]]>")) return false;
      }
      else if (starts("<!"))
      {
        if (!skip_until(">")) return false;
      }
      else
        return true;
    }
    return true;
  }

  static std::string unescape(const std::string& s)
  {
    if (s.find('&') == std::string::npos) return s;
    std::string o;
    for (size_t i = 0; i < s.size(); ++i)
    {
      if (s[i] == '&')
      {
        if (!s.compare(i, 4, "&lt;")) { o += '<'; i += 3; continue; }
        if (!s.compare(i, 4, "&gt;")) { o += '>'; i += 3; continue; }
        if (!s.compare(i, 5, "&amp;")) { o += '&'; i += 4; continue; }
        if (!s.compare(i, 6, "&quot;")) { o += '"'; i += 5; continue; }
        if (!s.compare(i, 6, "&apos;")) { o += '\''; i += 5; continue; }
      }
      o += s[i];
    }
    return o;
  }

  int depth = 0;  // nesting of parse_element: a robot_description never nests deeper than a few levels
  struct DepthGuard
  {
    int& d;
    explicit DepthGuard(int& x) : d(x) { ++d; }
    ~DepthGuard() { --d; }
  };
  std::unique_ptr<XmlNode> parse_element()
  {
    // at '<' of a start tag
    DepthGuard guard(depth);
    if (depth > 256)  // recursion bound: a hostile document must not overflow the host stack
    {
      err = "XML nested deeper than 256 elements";
      return nullptr;
    }
    ++p;
    std::unique_ptr<XmlNode> n(new XmlNode());
    const char* s = p;
    while (p < end && name_char(*p)) ++p;
    n->name.assign(s, p);
    if (n->name.empty())
    {
      err = "empty element name";
      return nullptr;
    }
    for (;;)
    {
      skip_ws();
      if (p >= end)
      {
        err = "unexpected end inside <" + n->name + ">";
        return nullptr;
      }
      if (*p == '/')
      {
        if (p + 1 < end && p[1] == '>')
        {
          p += 2;
          return n;
        }
        err = "malformed tag <" + n->name + ">";
        return nullptr;
      }
      if (*p == '>')
      {
        ++p;
        break;
      }
      const char* a = p;
      while (p < end && name_char(*p)) ++p;
      std::string key(a, p);
      skip_ws();
      if (key.empty() || p >= end || *p != '=')
      {
        err = "malformed attribute in <" + n->name + ">";
        return nullptr;
      }
      ++p;
      skip_ws();
      if (p >= end || (*p != '"' && *p != '\''))
      {
        err = "unquoted attribute value in <" + n->name + ">";
        return nullptr;
      }
      const char q = *p++;
      const char* v = p;
      while (p < end && *p != q) ++p;
      if (p >= end)
      {
        err = "unterminated attribute value in <" + n->name + ">";
        return nullptr;
      }
      n->attrs.emplace_back(key, unescape(std::string(v, p)));
      ++p;
    }
    // children until the matching end tag
    for (;;)
    {
      if (!skip_misc()) return nullptr;
      if (p >= end)
      {
        err = "missing </" + n->name + ">";
        return nullptr;
      }
      if (starts("</"))
      {
        p += 2;
        const char* e = p;
        while (p < end && name_char(*p)) ++p;
        if (std::string(e, p) != n->name)
        {
          err = "mismatched end tag </" + std::string(e, p) + "> for <" + n->name + ">";
          return nullptr;
        }
        skip_ws();
        if (p >= end || *p != '>')
        {
          err = "malformed end tag </" + n->name + ">";
          return nullptr;
        }
        ++p;
        return n;
      }
      auto c = parse_element();
      if (!c) return nullptr;
      n->children.push_back(std::move(c));
    }
  }

  std::unique_ptr<XmlNode> parse_document()
  {
    if (!skip_misc()) return nullptr;
    if (p >= end)
    {
      err = "no root element";
      return nullptr;
    }
    return parse_element();
  }
};

bool parse_doubles(const std::string* s, int n, double* out)
{
  if (!s) return false;
  const char* c = s->c_str();
  for (int i = 0; i < n; ++i)
  {
    char* e = nullptr;
    out[i] = strtod(c, &e);
    if (e == c) return false;
    c = e;
  }
  while (*c == ' ' || *c == '\t' || *c == '\n' || *c == '\r') ++c;
  return *c == 0;
}
bool parse_double(const std::string* s, double* out) { return parse_doubles(s, 1, out); }

// urdf::Rotation::setFromRPY (urdfdom_headers): x,y,z,w, normalised
void rpy_to_quat(const double rpy[3], double q[4])
{
  const double phi = rpy[0] / 2.0, the = rpy[1] / 2.0, psi = rpy[2] / 2.0;
  q[0] = sin(phi) * cos(the) * cos(psi) - cos(phi) * sin(the) * sin(psi);
  q[1] = cos(phi) * sin(the) * cos(psi) + sin(phi) * cos(the) * sin(psi);
  q[2] = cos(phi) * cos(the) * sin(psi) - sin(phi) * sin(the) * cos(psi);
  q[3] = cos(phi) * cos(the) * cos(psi) + sin(phi) * sin(the) * sin(psi);
  const double s = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (s == 0.0)
  {
    q[0] = q[1] = q[2] = 0.0;
    q[3] = 1.0;
  }
  else
    for (int i = 0; i < 4; ++i) q[i] /= s;
}

bool parse_pose(const XmlNode* origin, double xyz[3], double quat[4], std::string& err)
{
  xyz[0] = xyz[1] = xyz[2] = 0.0;
  quat[0] = quat[1] = quat[2] = 0.0;
  quat[3] = 1.0;
  if (!origin) return true;
  if (origin->attr("xyz") && !parse_doubles(origin->attr("xyz"), 3, xyz))
  {
    err = "malformed origin xyz";
    return false;
  }
  if (origin->attr("rpy"))
  {
    double rpy[3];
    if (!parse_doubles(origin->attr("rpy"), 3, rpy))
    {
      err = "malformed origin rpy";
      return false;
    }
    rpy_to_quat(rpy, quat);
  }
  return true;
}

struct TreeJoint
{
  rdyn_joint_desc d;
  std::string parent, child;
};
struct TreeLink
{
  rdyn_link_desc d;
  int parent_joint = -1;
  std::vector<int> child_joints;
};

void copy_name(char dst[64], const std::string& s)
{
  memset(dst, 0, 64);
  strncpy(dst, s.c_str(), 63);
}

}  // namespace

int rdyn_urdf_extract_chain(const char* xml, const char* base, const char* tool, std::vector<rdyn_joint_desc>& out_joints,
                            std::vector<rdyn_link_desc>& out_links)
{
  XmlParser P(xml);
  std::unique_ptr<XmlNode> doc = P.parse_document();
  if (!doc)
  {
    rdyn_set_error("URDF parse error: %s", P.err.c_str());
    return RDYN_ERR_URDF;
  }
  if (doc->name != "robot")
  {
    rdyn_set_error("URDF parse error: root element is <%s>, expected <robot>", doc->name.c_str());
    return RDYN_ERR_URDF;
  }
  std::string err;
  std::vector<TreeLink> links;
  std::map<std::string, int> link_index;
  std::map<std::string, TreeJoint> joints;  // name-ordered like urdfdom's joints_ map

  for (auto& e : doc->children)
  {
    if (e->name == "link")
    {
      const std::string* nm = e->attr("name");
      if (!nm)
      {
        rdyn_set_error("URDF parse error: <link> without name");
        return RDYN_ERR_URDF;
      }
      if (nm->size() > 63)  // the POD descriptor keeps 63 characters: a longer name could never be matched as base / tool
      {
        rdyn_set_error("URDF parse error: link name longer than 63 characters: '%.80s'", nm->c_str());
        return RDYN_ERR_URDF;
      }
      TreeLink L;
      memset(&L.d, 0, sizeof L.d);
      copy_name(L.d.name, *nm);
      L.d.com_quat[3] = 1.0;
      const XmlNode* in = e->child("inertial");
      if (in)
      {
        L.d.has_inertial = 1;
        if (!parse_pose(in->child("origin"), L.d.com_xyz, L.d.com_quat, err))
        {
          rdyn_set_error("URDF parse error: link '%s': %s", nm->c_str(), err.c_str());
          return RDYN_ERR_URDF;
        }
        const XmlNode* m = in->child("mass");
        if (!m || !parse_double(m->attr("value"), &L.d.mass))
        {
          rdyn_set_error("URDF parse error: link '%s': <inertial> needs <mass value=...>", nm->c_str());
          return RDYN_ERR_URDF;
        }
        const XmlNode* I = in->child("inertia");
        if (!I || !parse_double(I->attr("ixx"), &L.d.ixx) || !parse_double(I->attr("ixy"), &L.d.ixy) ||
            !parse_double(I->attr("ixz"), &L.d.ixz) || !parse_double(I->attr("iyy"), &L.d.iyy) ||
            !parse_double(I->attr("iyz"), &L.d.iyz) || !parse_double(I->attr("izz"), &L.d.izz))
        {
          rdyn_set_error("URDF parse error: link '%s': <inertial> needs <inertia ixx ixy ixz iyy iyz izz>", nm->c_str());
          return RDYN_ERR_URDF;
        }
      }
      if (link_index.count(*nm))
      {
        rdyn_set_error("URDF parse error: link '%s' is not unique", nm->c_str());
        return RDYN_ERR_URDF;
      }
      link_index[*nm] = (int)links.size();
      links.push_back(L);
    }
    else if (e->name == "joint")
    {
      const std::string* nm = e->attr("name");
      const std::string* ty = e->attr("type");
      const XmlNode* par = e->child("parent");
      const XmlNode* chi = e->child("child");
      if (!nm || !ty || !par || !chi || !par->attr("link") || !chi->attr("link"))
      {
        rdyn_set_error("URDF parse error: <joint> needs name, type, <parent link>, <child link>");
        return RDYN_ERR_URDF;
      }
      if (nm->size() > 63)
      {
        rdyn_set_error("URDF parse error: joint name longer than 63 characters: '%.80s'", nm->c_str());
        return RDYN_ERR_URDF;
      }
      TreeJoint J;
      memset(&J.d, 0, sizeof J.d);
      copy_name(J.d.name, *nm);
      J.parent = *par->attr("link");
      J.child = *chi->attr("link");
      if (*ty == "revolute") J.d.urdf_type = RDYN_URDF_REVOLUTE;
      else if (*ty == "continuous") J.d.urdf_type = RDYN_URDF_CONTINUOUS;
      else if (*ty == "prismatic") J.d.urdf_type = RDYN_URDF_PRISMATIC;
      else if (*ty == "floating") J.d.urdf_type = RDYN_URDF_FLOATING;
      else if (*ty == "planar") J.d.urdf_type = RDYN_URDF_PLANAR;
      else if (*ty == "fixed") J.d.urdf_type = RDYN_URDF_FIXED;
      else
      {
        rdyn_set_error("URDF parse error: joint '%s' has unknown type '%s'", nm->c_str(), ty->c_str());
        return RDYN_ERR_URDF;
      }
      if (!parse_pose(e->child("origin"), J.d.origin_xyz, J.d.origin_quat, err))
      {
        rdyn_set_error("URDF parse error: joint '%s': %s", nm->c_str(), err.c_str());
        return RDYN_ERR_URDF;
      }
      if (J.d.urdf_type != RDYN_URDF_FIXED && J.d.urdf_type != RDYN_URDF_FLOATING)
      {
        const XmlNode* ax = e->child("axis");
        if (!ax || !ax->attr("xyz"))
        {
          J.d.axis[0] = 1.0;  // urdfdom default
        }
        else if (!parse_doubles(ax->attr("xyz"), 3, J.d.axis))
        {
          rdyn_set_error("URDF parse error: joint '%s': malformed axis", nm->c_str());
          return RDYN_ERR_URDF;
        }
      }
      const XmlNode* lim = e->child("limit");
      if (lim)
      {
        J.d.has_limits = 1;
        if (lim->attr("lower")) parse_double(lim->attr("lower"), &J.d.lower);
        if (lim->attr("upper")) parse_double(lim->attr("upper"), &J.d.upper);
        if (lim->attr("velocity")) parse_double(lim->attr("velocity"), &J.d.velocity);
        if (lim->attr("effort")) parse_double(lim->attr("effort"), &J.d.effort);
      }
      if (joints.count(*nm))
      {
        rdyn_set_error("URDF parse error: joint '%s' is not unique", nm->c_str());
        return RDYN_ERR_URDF;
      }
      joints[*nm] = J;
    }
  }

  // tree (urdfdom initTree): joints in name order
  std::vector<const TreeJoint*> jlist;
  for (auto& kv : joints)
  {
    const TreeJoint& J = kv.second;
    auto ip = link_index.find(J.parent), ic = link_index.find(J.child);
    if (ip == link_index.end() || ic == link_index.end())
    {
      rdyn_set_error("URDF parse error: joint '%s' references an unknown link", kv.first.c_str());
      return RDYN_ERR_URDF;
    }
    if (links[ic->second].parent_joint >= 0)
    {
      rdyn_set_error("URDF parse error: link '%s' has two parent joints", J.child.c_str());
      return RDYN_ERR_URDF;
    }
    links[ic->second].parent_joint = (int)jlist.size();
    links[ip->second].child_joints.push_back((int)jlist.size());
    jlist.push_back(&J);
  }
  int root = -1, n_roots = 0;
  for (size_t i = 0; i < links.size(); ++i)
    if (links[i].parent_joint < 0)
    {
      root = (int)i;
      ++n_roots;
    }
  if (n_roots != 1)
  {
    rdyn_set_error("URDF parse error: expected exactly one root link, found %d", n_roots);
    return RDYN_ERR_URDF;
  }

  // Link::findChild (primitives_impl.h:424-440), iterative depth-first with the same visiting order
  auto find_child = [&](int start, const std::string& name) -> int {
    if (name == links[start].d.name) return start;
    // recursive semantics: for each child in order: check child, then recurse into it
    struct Rec
    {
      static int go(const std::vector<TreeLink>& L, const std::vector<const TreeJoint*>& J, const std::map<std::string, int>& idx, int at,
                    const std::string& nm, int depth)
      {
        if (depth > 4096) return -1;
        for (int cj : L[at].child_joints)
        {
          const int c = idx.find(J[cj]->child)->second;
          if (nm == L[c].d.name) return c;
          const int r = go(L, J, idx, c, nm, depth + 1);
          if (r >= 0) return r;
        }
        return -1;
      }
    };
    return Rec::go(links, jlist, link_index, start, name, 0);
  };

  const int ib = find_child(root, base);
  if (ib < 0)
  {
    rdyn_set_error("Base link not found");  // primitives_impl.h:603
    return RDYN_ERR_BASE_NOT_FOUND;
  }
  const int it = find_child(ib, tool);
  if (it < 0)
  {
    rdyn_set_error("Tool link not found");  // primitives_impl.h:610
    return RDYN_ERR_TOOL_NOT_FOUND;
  }
  // walk tool -> base (primitives_impl.h:616-626)
  out_joints.clear();
  out_links.clear();
  int act = it;
  for (;;)
  {
    out_links.insert(out_links.begin(), links[act].d);
    if (act != ib)
    {
      const TreeJoint* J = jlist[links[act].parent_joint];
      out_joints.insert(out_joints.begin(), J->d);
      act = link_index[J->parent];
    }
    else
      break;
  }
  return RDYN_OK;
}
