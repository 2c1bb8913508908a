"""sppm_intersect_static -- the entry point that allocates nothing -- against sppm_intersect, through the C ABI:
two calls in a row on the same GPU, the second with a smaller input (fewer rays AND a smaller scene), so that
whatever the first call left in the library's static device array must not leak into the second.
Bit-exact ids and hit records, and the hit counter equals the number of ids >= 0.  Collected last on purpose."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import capi, scene, sppm

pytestmark = pytest.mark.gpu


def _rays(n, seed):
    rng = np.random.default_rng(seed)
    org = rng.uniform([1.0, 0.0, 0.0], [99.0, 81.6, 170.0], size=(n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.sqrt((d * d).sum(1, keepdims=True))
    return np.ascontiguousarray(np.hstack([org, d]))


def _existing(table, rays):
    import torch
    ctx = sppm.SppmContext(sppm.make_config(8, 8), table)
    ids, hit = ctx.intersect(torch.from_numpy(rays))
    out = ids.cpu().numpy(), hit.cpu().numpy()
    ctx.check(); ctx.close()
    return out


def _static(table, rays):
    import torch
    L = capi.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    d_rays = torch.from_numpy(rays).to(dev)
    n = rays.shape[0]
    ids = torch.full((max(n, 1),), -7, dtype=torch.int32, device=dev)
    hit = torch.full((max(n, 1), 7), -7.0, dtype=torch.float64, device=dev)
    n_hit = torch.full((1,), 12345, dtype=torch.int64, device=dev)
    arr = sppm._sphere_array(table)
    stream = torch.cuda.current_stream(dev).cuda_stream
    capi.check(L.sppm_intersect_static(arr, len(arr), d_rays.data_ptr(), n, ids.data_ptr(), hit.data_ptr(), n_hit.data_ptr(), C.c_void_p(stream)))
    torch.cuda.synchronize()
    return ids[:n].cpu().numpy(), hit[:n].cpu().numpy(), int(n_hit.item())


def test_static_entry_point_equals_sppm_intersect_twice():
    big, small = list(scene.CORNELL_SPHERES), list(scene.CORNELL_SPHERES)[:7]  # the second scene lacks spheres 7..9
    for table, n, seed in ((big, 200_000, 1), (small, 1_777, 2), (big, 0, 3)):
        rays = _rays(n, seed)
        got_id, got_hit, got_n = _static(table, rays)
        if n == 0:
            assert got_n == 0
            continue
        ref_id, ref_hit = _existing(table, rays)
        assert np.array_equal(got_id, ref_id)
        assert np.array_equal(got_hit.view(np.uint64), ref_hit.view(np.uint64))
        assert got_n == int((ref_id >= 0).sum()) and got_n > 0


def test_static_entry_point_allocates_nothing_and_handles_300_spheres():
    """A scene near the upper end of the static table (300 of 512 spheres: eight upload launches) against sppm_intersect,
    which also takes the brute-force path up to 512 spheres, so equal-distance ties resolve alike; and the device's free
    memory (hipMemGetInfo) is the same before and after a call: the entry point allocates nothing."""
    import torch
    table = scene.synthetic_spheres(300 - 7)  # (the generator adds the seven spheres of the box)
    assert 256 < len(table) <= 512
    rays = _rays(50_000, 5)
    ref_id, ref_hit = _existing(table, rays)
    _static(table, rays[:64])  # a first call on this device (it resolves the array's address)
    L = capi.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    d_rays = torch.from_numpy(rays).to(dev)
    n = rays.shape[0]
    ids = torch.full((n,), -7, dtype=torch.int32, device=dev)
    hit = torch.full((n, 7), -7.0, dtype=torch.float64, device=dev)
    arr = sppm._sphere_array(table)
    stream = torch.cuda.current_stream(dev).cuda_stream
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(dev)[0]
    capi.check(L.sppm_intersect_static(arr, len(arr), d_rays.data_ptr(), n, ids.data_ptr(), hit.data_ptr(), None, C.c_void_p(stream)))
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(dev)[0]
    assert free1 == free0, f"free device memory changed by {free0 - free1} bytes across the call"
    assert np.array_equal(ids.cpu().numpy(), ref_id)
    assert np.array_equal(hit.cpu().numpy().view(np.uint64), ref_hit.view(np.uint64))


def test_static_entry_point_rejects_large_scenes():
    import torch
    L = capi.lib()
    table = scene.synthetic_spheres(600)
    arr = sppm._sphere_array(table)
    rays = torch.zeros((4, 6), dtype=torch.float64, device="cuda")
    ids = torch.zeros((4,), dtype=torch.int32, device="cuda")
    hit = torch.zeros((4, 7), dtype=torch.float64, device="cuda")
    rc = L.sppm_intersect_static(arr, len(arr), rays.data_ptr(), 4, ids.data_ptr(), hit.data_ptr(), None, None)
    assert rc == -1 and b"512" in L.sppm_last_error()
