"""umt_metric_partials_static: the allocation-free twin of umt_metric_partials (include/umetrack_hip.h).

Called through the C ABI (ctypes), compared bit for bit with the workspace entry point on two calls in a
row, the second with a smaller input (so the static array still holds the first call's rows), then on an
input at the size limit's edge and one beyond it (clear error, no launch).  The file sorts last so that a
failure here cannot hide another test from `pytest -x`.
"""
import ctypes

import numpy as np
import pytest
import torch

from umetrack_amd import _native as nat

pytestmark = pytest.mark.gpu


def _inputs(n_rec, T, seed, dev):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    gt = (torch.randn(n_rec, 2, T, 21, 3, generator=g) * 60).to(dev)
    pred = gt + torch.randn(n_rec, 2, T, 21, 3, generator=g).to(dev) * 9
    valid = (torch.rand(n_rec, 2, T, generator=g) > 0.15).to(dev)
    return gt.contiguous(), pred.contiguous(), valid.contiguous()


def _with_workspace(L, gt, pred, valid, dev):
    n_rec, _, T = valid.shape
    out = torch.full((nat.METRIC_WORDS,), -1.0, dtype=torch.float64, device=dev)
    nbytes = L.umt_metric_workspace_bytes(n_rec, T)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    nat.check(L.umt_metric_partials(nat.ptr(gt), nat.ptr(pred), nat.ptr(valid.view(torch.uint8)), n_rec, T,
                                    nat.ptr(out), nat.ptr(ws), nbytes, nat.stream_ptr(dev)), "umt_metric_partials")
    return out


def _static(L, gt, pred, valid, out, dev):
    n_rec, _, T = valid.shape
    return L.umt_metric_partials_static(nat.ptr(gt), nat.ptr(pred), nat.ptr(valid.view(torch.uint8)), n_rec, T,
                                        nat.ptr(out), nat.stream_ptr(dev))


def test_static_entry_point_matches_workspace_entry_point():
    dev = nat.require_gpu()
    L = nat.lib()
    with torch.cuda.device(dev):
        # first call: 40 x 2 x 67 = 5360 hand-frames = 21 workgroups; second, smaller: 3 x 2 x 11 = 66 = 1 workgroup
        for n_rec, T, seed in ((40, 67, 1), (3, 11, 2)):
            gt, pred, valid = _inputs(n_rec, T, seed, dev)
            want = _with_workspace(L, gt, pred, valid, dev)
            got = torch.full((nat.METRIC_WORDS,), -2.0, dtype=torch.float64, device=dev)
            # a first call resolves the static array's address and loads the code object; the property is about
            # every call after that.  Device-level free memory (hipMemGetInfo) moves with ANY hipMalloc/hipFree in the
            # process, the library's included; torch's own counter would only see torch's caching allocator.
            nat.check(_static(L, gt, pred, valid, got, dev), "umt_metric_partials_static")
            got.fill_(-2.0)
            torch.cuda.synchronize()
            free_before, torch_before = torch.cuda.mem_get_info(dev)[0], torch.cuda.memory_allocated(dev)
            nat.check(_static(L, gt, pred, valid, got, dev), "umt_metric_partials_static")
            torch.cuda.synchronize()
            assert torch.cuda.mem_get_info(dev)[0] == free_before and torch.cuda.memory_allocated(dev) == torch_before
            assert torch.equal(got, want), (n_rec, T, (got - want).abs().max().item())
            # and against float64 numpy for the sums a reader can check by eye
            e = (gt.double() - pred.double()).norm(dim=-1).mean(-1)[valid]
            assert got[1].item() == float(valid.sum().item())
            np.testing.assert_allclose(got[0].item(), float(e.sum().item()), rtol=1e-12)
            assert got[6].item() == float(valid.numel())


def test_static_entry_point_limit():
    dev = nat.require_gpu()
    L = nat.lib()
    lim = nat.METRIC_STATIC_MAX_HAND_FRAMES
    with torch.cuda.device(dev):
        # exactly at the limit: valid mask all False keeps the test cheap (no landmark is read for invalid rows)
        n_rec, T = lim // 2 // 64, 64
        assert n_rec * 2 * T == lim
        gt = torch.zeros(n_rec, 2, T, 21, 3, device=dev)
        valid = torch.zeros(n_rec, 2, T, dtype=torch.bool, device=dev)
        valid[5, 1, 7] = True
        out = torch.empty(nat.METRIC_WORDS, dtype=torch.float64, device=dev)
        nat.check(_static(L, gt, gt, valid, out, dev), "umt_metric_partials_static")
        torch.cuda.synchronize()
        assert out[1].item() == 1.0 and out[6].item() == float(lim) and out[0].item() == 0.0
        assert torch.equal(out, _with_workspace(L, gt, gt, valid, dev))
        # one recording more: refused with a clear error, nothing launched, output untouched
        out.fill_(-7.0)
        rc = L.umt_metric_partials_static(nat.ptr(gt), nat.ptr(gt), nat.ptr(valid.view(torch.uint8)), n_rec + 1, T,
                                          nat.ptr(out), nat.stream_ptr(dev))
        assert rc == 5
        msg = L.umt_last_error().decode()
        assert "static scratch limit" in msg and str(lim) in msg
        torch.cuda.synchronize()
        assert bool((out == -7.0).all())
    assert ctypes.c_int(rc).value != 0
