"""Seeded synthetic trajectory samples.

The reference harness draws every (q, Dq, DDq, DDDq) with ``Eigen::VectorXd::setRandom()`` --
i.i.d. uniform on [-1, 1], unseeded (rosdyn_core/test/rosdyn_speed_test.cpp:111-114).  This module
produces the same distribution from a seeded splitmix64 stream so that the GPU path, the CPU oracle
and the C++ harness (rosdyn_amd/csrc/rdyn_speed_test.cpp uses the same generator) see identical inputs.
"""
import numpy as np

_M64 = (1 << 64) - 1


def splitmix64(seed, count):
    """`count` successive outputs of splitmix64 started at `seed` (numpy uint64, vectorised)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, count + 1, dtype=np.uint64)
        z = np.uint64(seed & _M64) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform_pm1(seed, shape):
    """U[-1, 1) doubles with 53 random bits, C-order fill."""
    n = int(np.prod(shape))
    x = splitmix64(seed, n)
    return ((x >> np.uint64(11)).astype(np.float64) * (2.0 ** -52) - 1.0).reshape(shape)


def trajectory_batch(seed, n_samples, n_active, order=3):
    """Returns a tuple (q, Dq, DDq[, DDDq]) of (N, n) C-contiguous arrays; stream k uses seed + k."""
    return tuple(uniform_pm1(seed + 0x1000 * k, (n_samples, n_active)) for k in range(order))
