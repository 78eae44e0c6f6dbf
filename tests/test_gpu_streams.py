"""include/velo.h STREAM CONTRACT, the shared-stream variant: a caller that runs the ctx on ITS OWN stream
(velo_set_stream) needs no synchronisation at all between producing a device buffer and handing it to a *_dev
entry point -- the stream orders them.  The session-wide producer hook of tests/conftest.py (a safety net for
tests that build tensors on torch's stream while the ctx runs on a separate non-blocking one) is switched OFF
here, and nothing in these tests synchronises before a call: the library is exercised under an un-synchronised
caller."""
import numpy as np
import pytest
import torch

from veloslam_amd import capi
from tests.util_scene import make_workload

pytestmark = pytest.mark.gpu


@pytest.fixture()
def no_producer_hook():
    saved = capi._producer_sync
    capi.set_producer_sync(None)
    yield
    capi.set_producer_sync(saved)


def test_dev_entry_points_on_the_callers_stream_without_any_host_wait(oracle, no_producer_hook):
    wl = make_workload(map_points=150_000, n_frames=2)
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(device=dev)
    om = oracle.Map(*wl["map"], 1.0, 8)
    f = wl["frames"][0]
    s = f["sensor"]
    n = s["x"].size
    ox, oy, oz = oracle.compensate(s["x"], s["y"], s["z"], s["pkt"], f["table"])
    # pinned host copies: the H2D copies below are truly asynchronous, the kernels that follow really are
    # queued behind work that has not happened yet when the entry point is called
    pin = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory()   # noqa: E731
    hx, hy, hz = pin(s["x"]), pin(s["y"]), pin(s["z"])
    hpkt = pin(np.ascontiguousarray(s["pkt"], np.uint16).view(np.int16))
    htab = pin(np.ascontiguousarray(f["table"], np.float64).reshape(-1))
    hm = [pin(a) for a in wl["map"]]
    ctx = capi.Context(0, max_batch=2)
    try:
        with torch.cuda.stream(side):
            ctx.set_stream(side.cuda_stream)
            # a long-running kernel in front, so that every producer below is still pending at call time
            junk = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev)
            for _ in range(8):
                junk.normal_()
            mx, my, mz = (h.to(dev, non_blocking=True) for h in hm)
            ctx.map_reset_dev(mx.data_ptr(), my.data_ptr(), mz.data_ptr(), mx.numel(), 1.0, 8)
            dx, dy, dz = (h.to(dev, non_blocking=True) for h in (hx, hy, hz))
            dpkt = hpkt.to(dev, non_blocking=True)
            dtab = htab.to(dev, non_blocking=True)
            cx, cy, cz = (torch.empty(n, dtype=torch.float32, device=dev) for _ in range(3))
            ctx.compensate_dev(dx.data_ptr(), dy.data_ptr(), dz.data_ptr(), dpkt.data_ptr(), n, dtab.data_ptr(),
                               f["table"].shape[0], cx.data_ptr(), cy.data_ptr(), cz.data_ptr())
            ctx.frames_adopt_dev(cx.data_ptr(), cy.data_ptr(), cz.data_ptr(), [0, n])
            k = 8
            idx = torch.full((n, k), -7, dtype=torch.int32, device=dev)
            d2 = torch.zeros((n, k), dtype=torch.float32, device=dev)
            cnt = torch.zeros(n, dtype=torch.int32, device=dev)
            ctx.knn_dev(0, f["T0"], 1.0, k, idx.data_ptr(), d2.data_ptr(), cnt.data_ptr())
            res = ctx.icp_batch([f["T0"]], 6, 1.0)      # (fetches its result: waits for the stream)
            got = [t.cpu().numpy() for t in (cx, cy, cz, idx, d2, cnt)]
        assert np.array_equal(got[0], ox) and np.array_equal(got[1], oy) and np.array_equal(got[2], oz)
        oi, od, oc = om.knn(ox, oy, oz, f["T0"], 1.0, k)
        assert np.array_equal(got[3], oi) and np.array_equal(got[5], oc)
        assert np.array_equal(got[4].view(np.uint32), od.view(np.uint32))
        T_o, st, _ = om.icp(ox, oy, oz, f["T0"], 6, 1.0)
        assert all(int(res[0].iter[i].n_pairs) == int(st[i]["n_pairs"]) for i in range(6))
        assert np.allclose(np.array(list(res[0].T)), T_o, rtol=0, atol=1e-9)
    finally:
        ctx.close()
