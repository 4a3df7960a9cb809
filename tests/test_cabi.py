"""CPU tests (no GPU): the C-ABI library loads, exports every symbol include/rdyn.h declares, and the
host-side chain ingest (own URDF reader, ordering, input map, nominal parameters, limits, errors) agrees
with the oracle's independent front end.  No compute call is made here."""
import os
import re

import numpy as np
import pytest

from conftest import FIXTURES, ROOT, golden_cases, load_golden


def test_library_exports_every_declared_symbol():
    from rosdyn_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "rdyn.h")).read()
    declared = set(re.findall(r"\b(rdyn_[a-z_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    l = _lib.lib()
    for name in declared:
        assert getattr(l, name) is not None


@pytest.mark.parametrize("name", golden_cases())
def test_ingest_matches_oracle_front_end(name):
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    g = load_golden(name)
    c = Chain(g["urdf_path"], g["base"], g["tool"], g["gravity"])
    if g["inputs"]:
        assert c.setInputJointsName(g["inputs"])
    o = OracleChain(g["urdf_path"], g["base"], g["tool"], g["gravity"], g["inputs"])
    assert c.getLinksName() == o.spec.link_names
    assert c.getJointsName() == o.spec.joint_names
    assert c.getMoveableJointNames() == o.spec.moveable
    assert c.getActiveJointsName() == o.spec.active_names
    assert (c.getLinksNumber(), c.getJointsNumber(), c.getActiveJointsNumber()) == (o.L, o.nJ, o.n)
    assert np.allclose(c.getGravity(), g["gravity"], atol=0)
    assert np.abs(c.getNominalParameters() - g["pi"]).max() <= 1e-15
    cl = c.clone()
    assert cl.getActiveJointsName() == c.getActiveJointsName()
    assert np.array_equal(cl.getNominalParameters(), c.getNominalParameters())


def test_errors_match_reference_messages():
    from rosdyn_amd import Chain, RdynError
    u = os.path.join(FIXTURES, "ur10_like.urdf")
    with pytest.raises(RdynError, match="Base link not found"):        # primitives_impl.h:603
        Chain(u, "nope", "tool0")
    with pytest.raises(RdynError, match="Tool link not found"):        # primitives_impl.h:610
        Chain(u, "base_link", "nope")
    with pytest.raises(RdynError, match="Tool link not found"):        # tool must descend from base (607)
        Chain(u, "wrist_1_link", "shoulder_link")
    with pytest.raises(RdynError, match="URDF parse error"):
        Chain("<robot name='x'><link name='a'></robot>", "a", "a")
    c = Chain(u, "base_link", "tool0")
    assert c.setInputJointsName(["bogus"]) is False                    # primitives_impl.h:732-736
    assert c.getActiveJointsNumber() == 6                              # unchanged


def test_limits_follow_reference_rules():
    from rosdyn_amd import Chain
    c = Chain(os.path.join(FIXTURES, "mixed_joints.urdf"), "world", "tip")
    assert c.getActiveJointsName() == ["yaw", "lift", "shoulder", "slide2", "wrist"]
    # continuous: +-1e10 (primitives_impl.h:137-138); DDq_max = 10 * Dq_max (119, 140)
    assert c.getQMax()[0] == 1e10 and c.getQMin()[0] == -1e10
    assert np.allclose(c.getDQMax(), [3.0, 1.0, 2.5, 0.5, 4.0])
    assert np.allclose(c.getDDQMax(), 10 * c.getDQMax())
    assert np.allclose(c.getTauMax(), [100, 500, 80, 50, 10])
    assert c.getJointTypes() == [2, 0, 1, 0, 2, 1, 2, 0]               # fixed, R, P, R, floating->F, P, planar->F, R


def test_too_many_joints_is_reported():
    from rosdyn_amd import Chain, RdynError
    def xml(nj):
        links = "".join("<link name='l%d'/>" % i for i in range(nj + 1))
        joints = "".join("<joint name='j%d' type='revolute'><parent link='l%d'/><child link='l%d'/><axis xyz='0 0 1'/>"
                         "<limit lower='-1' upper='1' effort='1' velocity='1'/></joint>" % (i, i, i + 1) for i in range(nj))
        return "<robot name='long'>%s%s</robot>" % (links, joints)
    with pytest.raises(RdynError, match="at most 32"):
        Chain(xml(33), "l0", "l33")
    # 12 moving joints: ingested (names, limits, parameters); regressor / torque / inertia are served by the run-time-length kernels (round 6,
    # tests/test_gpu_longkin.py), but the normal equations and the R factors stop at 111 columns (11 input joints) -- the workspace
    # queries say so before anything touches a device; with at most 10 of them as input joints every entry point serves the chain
    c = Chain(xml(12), "l0", "l12")
    from rosdyn_amd._lib import lib
    assert c.getJointsNumber() == 12 and lib().rdyn_regressor_tsqr_workspace_bytes(c._h) == 0 and lib().rdyn_regressor_gram_workspace_bytes(c._h, 0) == 0
    c.setInputJointsName(["j%d" % i for i in range(2, 9)])
    assert c.getActiveJointsNumber() == 7 and lib().rdyn_regressor_tsqr_workspace_bytes(c._h) > 0
    # 11 input joints: 110 + 1 columns are what the Gram kernel and the widest R factor hold -- both are served through chunk images of the
    # run-time-length regressor kernel (slabs + one chunk image of 32 768 samples x 11 rows x 111 columns; 65 536 for the factor)
    c11 = Chain(xml(11), "l0", "l11")
    w = lib().rdyn_regressor_gram_workspace_bytes(c11._h, 0)
    assert w >= 32768 * 11 * 111 * 8 and lib().rdyn_regressor_tsqr_workspace_bytes(c11._h) >= 65536 * 11 * 111 * 8
    assert lib().rdyn_regressor_gram_workspace_bytes(c11._h, 4096) < w   # the caller's chunk size sizes the image


def test_generated_chain_variants_ingest_identically():
    """The perturbed URDFs of the mixed-chain workload parse the same way in the product reader and the oracle front end."""
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.urdf_gen import mixed_chain_set
    specs = mixed_chain_set(FIXTURES, n_chains=6)
    pis = []
    for xml, base, tool in specs:
        c, o = Chain(xml, base, tool, (0, 0, -9.806)), OracleChain(xml, base, tool, (0, 0, -9.806))
        assert c.getJointsName() == o.spec.joint_names
        assert np.abs(c.getNominalParameters() - o.nominal_parameters()).max() <= 1e-15
        pis.append(c.getNominalParameters())
    assert not np.allclose(pis[0], pis[2]) and not np.allclose(pis[1], pis[3])   # variants differ


def test_chain_from_desc_equals_chain_from_urdf():
    """rdyn_chain_from_desc (for callers that already hold a parsed urdf::Model) builds the same chain as the XML path."""
    import ctypes as C
    from oracle import urdf_model
    from rosdyn_amd import Chain, _lib
    path = os.path.join(FIXTURES, "mixed_joints.urdf")
    spec = urdf_model.load(path, "world", "tip", (0.1, 0.2, -9.0))
    types = {0: 1, 1: 2, 2: 3, 3: 6, 4: 4, 5: 5, 6: 0}       # oracle urdf_type -> rdyn_urdf_joint_type
    nj = spec.n_joints
    joints = (_lib.JointDesc * nj)()
    links = (_lib.LinkDesc * (nj + 1))()
    for i, j in enumerate(spec.joints):
        joints[i].name = j.name.encode()
        joints[i].urdf_type = types[j.urdf_type]
        joints[i].origin_xyz[:] = j.xyz
        joints[i].origin_quat[:] = j.quat
        joints[i].axis[:] = j.axis
        if j.limits:
            joints[i].has_limits = 1
            joints[i].lower, joints[i].upper = j.limits["lower"], j.limits["upper"]
            joints[i].velocity, joints[i].effort = j.limits["velocity"], j.limits["effort"]
    for i, l in enumerate(spec.links):
        links[i].name = l.name.encode()
        links[i].has_inertial = int(l.has_inertial)
        links[i].mass = l.mass
        links[i].com_xyz[:] = l.xyz
        links[i].com_quat[:] = l.quat
        links[i].ixx, links[i].ixy, links[i].ixz, links[i].iyy, links[i].iyz, links[i].izz = l.inertia
    desc = _lib.ChainDesc(nj, joints, links, (C.c_double * 3)(0.1, 0.2, -9.0))
    h = C.c_void_p()
    _lib.check(_lib.lib().rdyn_chain_from_desc(C.byref(desc), C.byref(h)))
    a = Chain(None, None, None, _handle=h)
    b = Chain(path, "world", "tip", (0.1, 0.2, -9.0))
    assert a.getJointsName() == b.getJointsName() and a.getLinksName() == b.getLinksName()
    assert a.getActiveJointsName() == b.getActiveJointsName() and a.getJointTypes() == b.getJointTypes()
    assert np.array_equal(a.getNominalParameters(), b.getNominalParameters())
    assert np.array_equal(a.getQMax(), b.getQMax()) and np.array_equal(a.getTauMax(), b.getTauMax())
    assert np.array_equal(a.getGravity(), b.getGravity())


def _build_c_example(tmp_path, name="regressor_batch"):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / name)
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(root, "include"),
           os.path.join(root, "examples", name + ".c"), "-L" + os.path.join(root, "rosdyn_amd"), "-lrdyn_hip", "-L/opt/rocm/lib",
           "-lamdhip64", "-lm", "-Wl,-rpath," + os.path.join(root, "rosdyn_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_plain_c_and_the_c_example_links(tmp_path):
    """include/rdyn.h compiles as C99 (no C++-isms cross the ABI) and the examples link against the library."""
    _build_c_example(tmp_path)
    _build_c_example(tmp_path, "identify")


@pytest.mark.gpu
def test_c_example_runs(tmp_path):
    import subprocess
    exe = _build_c_example(tmp_path)
    r = subprocess.run([exe, os.path.join(FIXTURES, "ur10_like.urdf"), "base_link", "wrist_3_link", "50000"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "n = 6, P = 60" in r.stdout


@pytest.mark.gpu
def test_c_identification_example_runs(tmp_path):
    """examples/identify.c: the reference's benchmark chain in its public URDF form (9 joints, 6 input joints, P = 90) from plain C --
    torques -> fused Gram and R factor (400 000 samples: the preconditioned CholeskyQR route, reduced chain + expansion) -> parameters
    that reproduce the torques; both solves agree on the rank."""
    import subprocess
    exe = _build_c_example(tmp_path, "identify")
    r = subprocess.run([exe, os.path.join(FIXTURES, "ur10_public.urdf"), "base_link", "tool0", "400000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "n = 6, P = 90 (6 rigid bodies)" in r.stdout
    assert "with 6 friction components (12 columns)" in r.stdout      # round 4: friction identified too (rdyn_identification_tsqr)
    # the 7-joint arm with its fixed flange and hand frames, friction columns included
    r = subprocess.run([exe, os.path.join(FIXTURES, "panda_like.urdf"), "link0", "hand", "100000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "n = 7, P = 90 (7 rigid bodies)" in r.stdout and "with 7 friction components (14 columns)" in r.stdout


def test_hostile_urdf_is_rejected_not_crashed():
    """the XML reader recursed without a bound (2e5 nested elements overflowed the host stack) and names longer than the
    63 characters the POD descriptor keeps could never be matched.  Both are RDYN_ERR_URDF now."""
    import ctypes as C
    from rosdyn_amd._lib import lib
    l = lib()
    g = (C.c_double * 3)(0, 0, 0)
    h = C.c_void_p()
    deep = "<robot name='r'>" + "<a>" * 200000 + "</a>" * 200000 + "<link name='b'/></robot>"
    assert l.rdyn_chain_from_urdf(deep.encode(), b"b", b"b", g, C.byref(h)) == 4 and b"nested" in l.rdyn_last_error()
    long_name = "L" * 64
    xml = ("<robot name='r'><link name='b'/><link name='%s'/><joint name='j' type='revolute'><parent link='b'/><child link='%s'/>"
           "<axis xyz='0 0 1'/><limit lower='-1' upper='1' effort='1' velocity='1'/></joint></robot>" % (long_name, long_name))
    assert l.rdyn_chain_from_urdf(xml.encode(), b"b", long_name.encode(), g, C.byref(h)) == 4 and b"63 characters" in l.rdyn_last_error()
    ok63 = xml.replace(long_name, "L" * 63)
    assert l.rdyn_chain_from_urdf(ok63.encode(), b"b", ("L" * 63).encode(), g, C.byref(h)) == 0
    l.rdyn_chain_destroy(h)


def test_workspace_queries_reflect_what_the_tsqr_entry_points_support():
    """Host-only: the workspace queries answer 0 for the combinations the kernels are not instantiated for (no GPU touched)."""
    import ctypes as C
    from rosdyn_amd import Chain
    from rosdyn_amd._lib import lib
    from rosdyn_amd.components import FRICTION1, FRICTION2, ComponentSet
    ur6 = Chain(os.path.join(FIXTURES, "ur10_like.urdf"), "base_link", "wrist_3_link")
    ur7 = Chain(os.path.join(FIXTURES, "ur10_like.urdf"), "base_link", "tool0")          # 7 chain joints (fixed flange)
    mix8 = Chain(os.path.join(FIXTURES, "mixed_joints.urdf"), "world", "tip")            # 8 chain joints
    L = lib()
    assert L.rdyn_regressor_tsqr_workspace_bytes(ur6._h) > 0 and L.rdyn_regressor_tsqr_workspace_bytes(ur7._h) > 0
    # round 3: joints that are not input joints are folded away (the reduced chain is swept, the factor expanded): 8 chain joints with
    # 5 input joints are served; the limit is 7 INPUT joints
    assert L.rdyn_regressor_tsqr_workspace_bytes(mix8._h) > 0
    # same reduced chain; the chain with the fixed flange also carries the buffer of the expanded 71 x 71 factor
    assert L.rdyn_regressor_tsqr_workspace_bytes(ur7._h) >= L.rdyn_regressor_tsqr_workspace_bytes(ur6._h) + 71 * 71 * 8
    six = ComponentSet([dict(type=FRICTION1, joint=j, min_velocity=1e-3, max_velocity=5.0, parameters=[1.0, 1.0]) for j in range(6)], 6)
    arr = C.cast(six._arr, C.c_void_p)
    w6 = L.rdyn_identification_tsqr_workspace_bytes(ur6._h, arr, six.n_comps)
    assert w6 > 0                                                                        # one more 16-column slot: 80 x 80 factors
    # round 4: component columns ride through the reduction (they belong to input joints): fixed frames no longer exclude them
    assert L.rdyn_identification_tsqr_workspace_bytes(ur7._h, arr, six.n_comps) >= w6
    assert L.rdyn_identification_tsqr_workspace_bytes(ur6._h, None, 0) == L.rdyn_regressor_tsqr_workspace_bytes(ur6._h)
    many = ComponentSet([dict(type=FRICTION2, joint=j % 6, min_velocity=1e-3, max_velocity=5.0, parameters=[1.0, 1.0, 0.1]) for j in range(7)], 6)
    assert many.columns == 21                                                            # 61 + 21 > 80: beyond the extra 16-column slot ...
    assert L.rdyn_identification_tsqr_workspace_bytes(ur6._h, C.cast(many._arr, C.c_void_p), many.n_comps) > 0   # ... the LDS-resident folds take it
    toomany = ComponentSet([dict(type=FRICTION1, joint=j % 6, min_velocity=1e-3, max_velocity=5.0, parameters=[1.0, 1.0]) for j in range(26)], 6)
    assert toomany.columns == 52                                                         # 61 + 52 = 113 columns: nothing holds that factor
    assert L.rdyn_identification_tsqr_workspace_bytes(ur6._h, C.cast(toomany._arr, C.c_void_p), toomany.n_comps) == 0
    # a materialised matrix: up to 112 columns (right-hand side included)
    assert L.rdyn_tsqr_workspace_bytes(64) > 0 and L.rdyn_tsqr_workspace_bytes(112) > 0 and L.rdyn_tsqr_workspace_bytes(113) == 0


@pytest.mark.parametrize("name", ["ur10_tool0", "mixed"])
def test_joint_constants_and_link_parameters_rebuild_the_chain(name):
    """Host-only: the Joint / Link views behind Chain::getJoints() / getLinks() (rdyn_chain_joint_constants,
    rdyn_chain_link_parameters).  The product of Joint::getTransformation(q_j) over the chain is the oracle's T_bt(q), and the links'
    nominal parameters are Chain::getNominalParameters (base link excluded)."""
    import ctypes as C
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd._lib import check, lib
    urdf, base, tool = {"ur10_tool0": ("ur10_like.urdf", "base_link", "tool0"), "mixed": ("mixed_joints.urdf", "world", "tip")}[name]
    path = os.path.join(FIXTURES, urdf)
    chain, ref = Chain(path, base, tool), OracleChain(path, base, tool)
    nJ = chain.getJointsNumber()
    rng = np.random.default_rng(3)
    q = rng.uniform(-1, 1, size=(4, ref.n))
    T_ref = ref.fk(q)[:, -1]                                        # (4, 3, 4)
    names = chain.getJointsName()
    active = chain.getActiveJointsName()
    dbl = lambda k: (C.c_double * k)()
    for s in range(4):
        T = np.eye(4)
        for j in range(nJ):
            R, t, ax, lim = dbl(9), dbl(3), dbl(3), dbl(5)
            check(lib().rdyn_chain_joint_constants(chain._h, j, R, t, ax, lim))
            R, t, ax = np.array(R).reshape(3, 3), np.array(t), np.array(ax)
            ty = lib().rdyn_chain_joint_type(chain._h, j)
            qj = q[s, active.index(names[j])] if names[j] in active else 0.0
            K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
            Tj = np.eye(4)
            if ty == 0:      # RDYN_REVOLUTE
                Tj[:3, :3], Tj[:3, 3] = R @ (np.eye(3) + np.sin(qj) * K + (1 - np.cos(qj)) * K @ K), t
            elif ty == 1:    # RDYN_PRISMATIC
                Tj[:3, :3], Tj[:3, 3] = R, t + R @ ax * qj
            else:
                Tj[:3, :3], Tj[:3, 3] = R, t
            T = T @ Tj
        assert np.abs(T[:3] - T_ref[s]).max() < 1e-13
    pi = chain.getNominalParameters()
    for l in range(1, nJ + 1):
        p, m, cg = dbl(10), C.c_double(), dbl(3)
        check(lib().rdyn_chain_link_parameters(chain._h, l, p, C.byref(m), cg))
        assert np.array_equal(np.array(p), pi[10 * (l - 1):10 * l])
        assert p[0] == m.value and np.allclose(np.array(p)[1:4], m.value * np.array(cg), atol=1e-15)
    assert lib().rdyn_chain_link_parameters(chain._h, nJ + 1, None, None, None) != 0      # out of range
    assert lib().rdyn_chain_joint_constants(chain._h, -1, None, None, None, None) != 0


def test_tsqr_report_answers_without_a_device_where_it_can():
    """Host-only: a batch below the preconditioned route's threshold is reported as route 0 (Householder folds) without touching
    the workspace or the GPU; invalid arguments are refused."""
    import ctypes as C
    from rosdyn_amd import Chain
    from rosdyn_amd._lib import RdynTsqrReport, lib
    ur6 = Chain(os.path.join(FIXTURES, "ur10_like.urdf"), "base_link", "wrist_3_link")
    L = lib()
    rep = RdynTsqrReport()
    dummy = C.c_double(0.0)
    assert L.rdyn_tsqr_last_report(ur6._h, None, 0, 1000, C.byref(dummy), -1, None, C.byref(rep)) == 0
    assert (rep.route, rep.stage, rep.n_deferred) == (0, 0, 0)
    assert L.rdyn_tsqr_last_report(ur6._h, None, 0, 1000, None, -1, None, C.byref(rep)) != 0          # null workspace
    assert L.rdyn_tsqr_last_report(ur6._h, None, 0, -1, C.byref(dummy), -1, None, C.byref(rep)) != 0  # negative batch size
    assert L.rdyn_tsqr_last_report(None, None, 0, 1000, C.byref(dummy), -1, None, C.byref(rep)) != 0  # null chain


def test_chains_longer_than_the_kernels_sweep_are_ingested():
    """Host-only: 14 chain joints (ur10_public + five fixed frames, one mid-chain) -- names, types, nominal
    parameters and the rigid-body reduction are there; the limit is RDYN_MAX_JOINTS = 32 chain joints and 10 input joints."""
    import ctypes as C
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd._lib import RDYN_MAX_JOINTS, lib
    path = os.path.join(FIXTURES, "ur10_public_long.urdf")
    chain, ref = Chain(path, "base_link", "tcp", (0, 0, -9.806)), OracleChain(path, "base_link", "tcp", (0, 0, -9.806))
    assert chain.getJointsNumber() == 14 and chain.getActiveJointsNumber() == 6 and RDYN_MAX_JOINTS == 32
    assert np.abs(chain.getNominalParameters() - ref.nominal_parameters()).max() == 0.0
    L = lib()
    body = (C.c_int32 * 14)()
    assert L.rdyn_chain_reduction(chain._h, body, None, None) == 6      # six rigid bodies
    assert list(body)[:2] == [-1, 2] or list(body)[0] == -1             # the frame in front of the first input joint never moves
    assert L.rdyn_regressor_tsqr_workspace_bytes(chain._h) > 0 and L.rdyn_regressor_gram_workspace_bytes(chain._h, 0) > 0
    # 33 joints: refused at ingest
    links = "".join("<link name='l%d'/>" % i for i in range(34))
    joints = "".join("<joint name='j%d' type='fixed'><parent link='l%d'/><child link='l%d'/></joint>" % (i, i, i + 1) for i in range(33))
    h = C.c_void_p()
    g = (C.c_double * 3)(0, 0, 0)
    assert L.rdyn_chain_from_urdf(("<robot name='r'>%s%s</robot>" % (links, joints)).encode(), b"l0", b"l33", g, C.byref(h)) == 5
    assert b"at most 32" in L.rdyn_last_error()
