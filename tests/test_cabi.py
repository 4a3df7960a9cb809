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
    links = "".join("<link name='l%d'/>" % i for i in range(13))
    joints = "".join("<joint name='j%d' type='revolute'><parent link='l%d'/><child link='l%d'/><axis xyz='0 0 1'/>"
                     "<limit lower='-1' upper='1' effort='1' velocity='1'/></joint>" % (i, i, i + 1) for i in range(12))
    with pytest.raises(RdynError, match="at most"):
        Chain("<robot name='long'>%s%s</robot>" % (links, joints), "l0", "l12")


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
