"""Synthetic chain variants for the mixed-chain workload (BASELINE.json configs[4]):
distinct 6- / 7-DOF chains obtained by perturbing a base URDF by up to +-20 % (link offsets, masses, centres of
mass, inertia tensors) with a seeded generator.  Inertia tensors are scaled as a whole so they stay SPD."""
import xml.etree.ElementTree as ET

import numpy as np

from .samples import uniform_pm1


def perturbed_urdf(base_xml, seed, amount=0.2):
    root = ET.fromstring(base_xml)
    elems = [e for e in root.iter() if e.tag in ("origin", "mass", "inertia")]
    r = uniform_pm1(seed, (len(elems), 4)) * amount
    for e, k in zip(elems, r):
        if e.tag == "origin" and e.get("xyz"):
            xyz = np.array([float(t) for t in e.get("xyz").split()])
            e.set("xyz", " ".join(repr(float(v)) for v in xyz * (1.0 + k[:3])))
        elif e.tag == "mass":
            e.set("value", repr(float(float(e.get("value")) * (1.0 + k[0]))))
        elif e.tag == "inertia":
            for a in ("ixx", "ixy", "ixz", "iyy", "iyz", "izz"):
                e.set(a, repr(float(float(e.get(a, 0.0)) * (1.0 + k[0]))))
    return ET.tostring(root, encoding="unicode")


def mixed_chain_set(fixtures_dir, n_chains=256, seed=0x5EED0005):
    """[(urdf_xml, base, tool)] alternating the UR10-like 6-DOF (cut at wrist_3_link) and Panda-like 7-DOF (cut at link7)."""
    import os
    with open(os.path.join(fixtures_dir, "ur10_like.urdf")) as f:
        ur = f.read()
    with open(os.path.join(fixtures_dir, "panda_like.urdf")) as f:
        pa = f.read()
    out = []
    for i in range(n_chains):
        if i % 2 == 0:
            out.append((perturbed_urdf(ur, seed + 977 * i), "base_link", "wrist_3_link"))
        else:
            out.append((perturbed_urdf(pa, seed + 977 * i), "link0", "link7"))
    return out
