"""Child process of tests/test_multi_gpu_alias.py: the in-library multi-device paths (rosdyn_amd/csrc/rdyn_multi_gpu.cpp) with
n_dev LOGICAL devices on ONE physical GPU.  Environment (set by the parent BEFORE this interpreter starts, so that the library's first
dlopen sees it): RDYN_TEST_ALIAS_DEVICES=1, RDYN_RCCL_PATH=tests/_build/librccl_stub.so (tests/cpp/rccl_stub.hip).
usage: python tests/_alias_multi_gpu.py <n_dev>; exits non-zero with the failing assertion."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FIXTURES = os.path.join(ROOT, "tests", "fixtures")
GRAV = (0.0, 0.0, -9.806)


def main(n_dev):
    import torch
    from rosdyn_amd import Chain
    from rosdyn_amd._lib import lib
    from rosdyn_amd.components import ComponentSet
    from rosdyn_amd.gram import MultiGpuGram

    assert os.environ.get("RDYN_TEST_ALIAS_DEVICES") == "1" and os.environ.get("RDYN_RCCL_PATH")
    stub = C.CDLL(os.environ["RDYN_RCCL_PATH"])
    stub.rccl_stub_collectives.restype = C.c_long
    stub.rccl_stub_fail_at.argtypes = [C.c_long]
    devs = [0] * n_dev
    gen = torch.Generator(device="cuda").manual_seed(600 + n_dev)

    def shards_for(n, sizes):
        out = []
        for N in sizes:
            base = torch.rand((4, N, n), dtype=torch.float64, device="cuda:0", generator=gen) * 2 - 1
            out.append(tuple(base[k].contiguous() for k in range(4)))
        return out

    def cat(shards):
        return tuple(torch.cat([s[k] for s in shards]) for k in range(4))

    # ---- 1. normal equations: ur10 base_link -> wrist_3_link (6 joints) and base_link -> tool0 (fixed tail: the reduced companion)
    for tool, P in (("wrist_3_link", 60), ("tool0", 70)):
        chain = Chain(os.path.join(FIXTURES, "ur10_like.urdf"), "base_link", tool, GRAV)
        sizes = [3000 + 117 * i for i in range(n_dev)]
        sizes[-1] = 0 if n_dev > 2 else sizes[-1]   # an EMPTY shard on the last device (a batch that does not fill the node)
        sh = shards_for(6, sizes)
        ctx = MultiGpuGram(devs)
        before = stub.rccl_stub_collectives()
        acc = ctx.regressor_gram(chain, sh)
        assert stub.rccl_stub_collectives() == before + 1, "ONE all-reduce per call"
        a0 = acc[0].cpu().numpy()
        for i in range(1, n_dev):
            assert np.array_equal(a0, acc[i].cpu().numpy()), "device %d holds different sums" % i
        G, c, bb = chain.getRegressorGram(*cat(sh))
        torch.cuda.synchronize()
        Gn = G.cpu().numpy()
        assert a0[P * P + P + 1] == sum(sizes), (a0[P * P + P + 1], sum(sizes))
        assert np.linalg.norm(a0[:P * P].reshape(P, P) - Gn) <= 1e-11 * np.linalg.norm(Gn)
        assert np.linalg.norm(a0[P * P:P * P + P] - c.cpu().numpy()) <= 1e-11 * np.linalg.norm(c.cpu().numpy())
        assert abs(a0[P * P + P] - float(bb.cpu()[0])) <= 1e-11 * float(bb.cpu()[0])
        # ---- 2. a batch streamed through the devices in pieces: overwrite, then accumulate
        sh2 = shards_for(6, [500 + 31 * i for i in range(n_dev)])
        acc = ctx.regressor_gram(chain, sh2, acc=acc, accumulate=True)
        a1 = acc[0].cpu().numpy()
        for i in range(1, n_dev):
            assert np.array_equal(a1, acc[i].cpu().numpy())
        G2, _, _ = chain.getRegressorGram(*cat(sh + sh2))
        torch.cuda.synchronize()
        assert a1[P * P + P + 1] == sum(sizes) + sum(s[0].shape[0] for s in sh2)
        assert np.linalg.norm(a1[:P * P].reshape(P, P) - G2.cpu().numpy()) <= 1e-11 * np.linalg.norm(G2.cpu().numpy())
        # ---- 3. two calls queued back to back, no synchronisation in between, inputs produced on torch's stream right before
        pend = []
        for rep in range(2):
            shx = shards_for(6, [700 + 13 * i + 101 * rep for i in range(n_dev)])
            pend.append((shx, ctx.regressor_gram(chain, shx, sync=False)))
        ctx.synchronize()
        for shx, accx in pend:
            Gx, _, _ = chain.getRegressorGram(*cat(shx))
            torch.cuda.synchronize()
            ax = accx[0].cpu().numpy()
            assert ax[P * P + P + 1] == sum(s[0].shape[0] for s in shx)
            assert np.linalg.norm(ax[:P * P].reshape(P, P) - Gx.cpu().numpy()) <= 1e-11 * np.linalg.norm(Gx.cpu().numpy())
            for i in range(1, n_dev):
                assert np.array_equal(ax, accx[i].cpu().numpy())
        del ctx

    # ---- 4. R factors: all-gather + fold + expand on every device
    for urdf, base, tool, n, with_comps in (("ur10_like.urdf", "base_link", "wrist_3_link", 6, False), ("ur10_public.urdf", "base_link", "tool0", 6, True),
                                            ("panda_like.urdf", "link0", "link7", 7, True)):
        chain = Chain(os.path.join(FIXTURES, urdf), base, tool, GRAV)
        comps = ComponentSet([dict(type=0, joint=j, min_velocity=1e-3, max_velocity=5.0, parameters=[0.4 + 0.1 * j, 1.0]) for j in range(n)], n) if with_comps else None
        sh = shards_for(n, [5000 + 211 * i for i in range(n_dev)])
        ctx = MultiGpuGram(devs)
        before = stub.rccl_stub_collectives()
        R = ctx.identification_tsqr(chain, sh, components=comps, sync=False)
        seen = (R[n_dev - 1].t() @ R[n_dev - 1]).clone()   # torch's stream, right behind the call: ordered by the library's event
        ctx.synchronize()
        torch.cuda.synchronize()
        assert stub.rccl_stub_collectives() == before + 1, "ONE all-gather per call"
        R0 = R[0].cpu().numpy()
        for i in range(1, n_dev):
            assert np.array_equal(R0, R[i].cpu().numpy()), "device %d holds a different factor" % i
        allq = cat(sh)
        if comps is not None:
            Gt = chain.getIdentificationGram(comps, *allq)
        else:
            Gt = chain.getRegressorGram(*allq)
        torch.cuda.synchronize()
        Gm, cm, bbm = (t.cpu().numpy() for t in Gt)
        n1 = R0.shape[0]
        full = np.zeros((n1, n1))
        full[:n1 - 1, :n1 - 1] = Gm
        full[:n1 - 1, n1 - 1] = full[n1 - 1, :n1 - 1] = cm
        full[n1 - 1, n1 - 1] = bbm[0]
        assert np.allclose(np.tril(R0, -1), 0.0)
        assert np.abs(R0.T @ R0 - full).max() <= 1e-10 * np.abs(full).max(), np.abs(R0.T @ R0 - full).max() / np.abs(full).max()
        assert np.abs(seen.cpu().numpy() - full).max() <= 1e-10 * np.abs(full).max()
        # accumulate: the same rows again -> twice the Gram matrix, identical on all devices
        R2 = ctx.identification_tsqr(chain, sh, components=comps, out=[r.clone() for r in R], accumulate=True)
        R20 = R2[0].cpu().numpy()
        for i in range(1, n_dev):
            assert np.array_equal(R20, R2[i].cpu().numpy())
        assert np.abs(R20.T @ R20 - 2 * full).max() <= 1e-10 * np.abs(full).max()
        del ctx

    # ---- 5. a collective that fails in the MIDDLE of its group: the call reports it, the group is closed (depth 0), the context only
    #         accepts destroy, and the SAME thread can create and use a new context (abort_collective left the group open)
    chain = Chain(os.path.join(FIXTURES, "ur10_like.urdf"), "base_link", "wrist_3_link", GRAV)
    sh = shards_for(6, [900 + 7 * i for i in range(n_dev)])
    for what in ("gram", "tsqr"):
        ctx = MultiGpuGram(devs)
        stub.rccl_stub_fail_at(2)           # the second rank's call inside the group
        try:
            (ctx.regressor_gram if what == "gram" else ctx.identification_tsqr)(chain, sh)
            raise AssertionError("the failing collective was not reported")
        except RuntimeError as e:
            assert "aborted" in str(e) and "destroy" in str(e), str(e)
        stub.rccl_stub_fail_at(0)
        assert stub.rccl_stub_group_depth() == 0, "the RCCL group was left open"
        try:
            ctx.regressor_gram(chain, sh)
            raise AssertionError("a broken context accepted work")
        except RuntimeError as e:
            assert "destroy" in str(e), str(e)
        del ctx
        ctx = MultiGpuGram(devs)            # same thread, right after
        acc = ctx.regressor_gram(chain, sh)
        G, _, _ = chain.getRegressorGram(*cat(sh))
        torch.cuda.synchronize()
        a0 = acc[0].cpu().numpy()
        assert np.linalg.norm(a0[:3600].reshape(60, 60) - G.cpu().numpy()) <= 1e-11 * np.linalg.norm(G.cpu().numpy())
        del ctx
    print("alias multi-gpu ok: n_dev = %d, %d collectives through the stub" % (n_dev, stub.rccl_stub_collectives()))


if __name__ == "__main__":
    main(int(sys.argv[1]))
