"""The in-library multi-device path (include/rdyn.h: rdyn_multi_gpu_*, rosdyn_amd/csrc/rdyn_multi_gpu.cpp) with
n_dev = 2, 4, 8 -- on ONE GPU.  The pool leases one GPU at a time and real RCCL refuses a communicator clique with repeated ordinals,
so until round 6 this code (one host thread and one stream per device, the grouped all-reduce / all-gather, the event ordering between
the context's and the callers' streams, abort on a partial group) had only ever run with one device.  Test infrastructure only:
  RDYN_TEST_ALIAS_DEVICES=1  rdyn_multi_gpu_create accepts {0, 0, ...}: n logical devices (own stream, events, workspace) on GPU 0
  RDYN_RCCL_PATH             tests/cpp/rccl_stub.hip built into tests/_build/librccl_stub.so: grouped all-reduce / all-gather as
                             device-side sums / copies across the logical ranks' buffers, ordered on their streams
Both are read when the library first loads RCCL, hence a child process per case (tests/_alias_multi_gpu.py holds the assertions: every
device bitwise equal, equal to the one-call result to 1e-11, accumulation, back-to-back calls without synchronisation, an empty shard,
a forced mid-group failure that must leave the RCCL group closed and a context that only accepts destroy)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "_build", "librccl_stub.so")


def _build_stub():
    src = os.path.join(ROOT, "tests", "cpp", "rccl_stub.hip")
    if os.path.exists(STUB) and os.path.getmtime(STUB) >= os.path.getmtime(src):
        return
    os.makedirs(os.path.dirname(STUB), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", STUB])


def test_stub_builds_and_exports_the_entry_points_the_library_resolves():
    """CPU: hipcc cross-compiles the stub; every symbol rdyn_multi_gpu.cpp dlsym()s is there."""
    import ctypes as C
    _build_stub()
    stub = C.CDLL(STUB)
    for name in ("ncclCommInitAll", "ncclCommDestroy", "ncclCommAbort", "ncclAllReduce", "ncclAllGather", "ncclGroupStart", "ncclGroupEnd",
                 "ncclGetErrorString", "rccl_stub_fail_at", "rccl_stub_group_depth", "rccl_stub_collectives"):
        getattr(stub, name)
    assert stub.ncclGroupStart() == 0 and stub.rccl_stub_group_depth() == 1 and stub.ncclGroupEnd() == 0 and stub.rccl_stub_group_depth() == 0
    assert stub.ncclGroupEnd() != 0   # nothing open


def test_repeated_ordinals_are_refused_without_the_test_hook():
    import ctypes as C
    from rosdyn_amd._lib import lib
    assert os.environ.get("RDYN_TEST_ALIAS_DEVICES") is None
    h = C.c_void_p()
    two_same = (C.c_int * 2)(0, 0)
    assert lib().rdyn_multi_gpu_create(two_same, 2, C.byref(h)) == 1 and b"distinct" in lib().rdyn_last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("n_dev", [2, 4, 8])
def test_multi_device_paths_on_logical_devices(n_dev):
    _build_stub()
    env = dict(os.environ, RDYN_TEST_ALIAS_DEVICES="1", RDYN_RCCL_PATH=STUB)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_alias_multi_gpu.py"), str(n_dev)], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, universal_newlines=True, timeout=900)
    tail = "\n".join(r.stdout.splitlines()[-25:])
    assert r.returncode == 0, tail
    assert "alias multi-gpu ok: n_dev = %d" % n_dev in r.stdout, tail
