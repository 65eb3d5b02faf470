"""
wgridder-style degridding (BASELINE configs[4]; africanus/gridding/wgridder/im2vis.py:63-99).  The reference's
arithmetic is ducc0 (absent: parity unpinned); what its tests pin is the accuracy contract
(africanus/gridding/wgridder/tests/test_wgridder.py:18-113): relative l2 error <= epsilon against the direct transform.
Same recipe here (their shapes, field of view and uvw scaling), for the image -> visibility direction, with the direct
transform evaluated by the CPU oracle's im_to_vis on the image's pixels as point sources.
"""
import numpy as np
import pytest

import oracle
from codex_africanus_amd.gridding.wgridder import model

pytestmark = pytest.mark.gpu
LIGHTSPEED = 2.99792458e8


def _l2error(a, b):
    return np.sqrt(np.sum(np.abs(a - b) ** 2) / np.maximum(np.sum(np.abs(a) ** 2), np.sum(np.abs(b) ** 2)))


def _explicit_degridder(uvw, freq, image, cell, celly, apply_w=True):
    """vis[r, c] = sum_xy image[x, y] / n exp(-2 pi i f/c (u x + v y - w (n - 1))): the adjoint of explicit_gridder
    (test_wgridder.py:18-46), through the oracle's direct transform (sources = pixels, w negated)."""
    nx, ny = image.shape
    x, y = np.meshgrid(*[-ss / 2 + np.arange(ss) for ss in (nx, ny)], indexing="ij")
    x, y = x * cell, y * celly
    lm = np.stack([x.ravel(), y.ravel()], axis=1)
    if apply_w:
        n = np.sqrt(1.0 - x ** 2 - y ** 2)
        img = (image / n).ravel()
        uvw2 = uvw * np.array([1.0, 1.0, -1.0])
        src = np.broadcast_to(img[:, None, None], (img.size, freq.size, 1)).copy()
        return oracle.im_to_vis(src, uvw2, lm, freq, omp=True)[:, :, 0]
    ph = (uvw[:, None, 0, None] * lm[None, None, :, 0] + uvw[:, None, 1, None] * lm[None, None, :, 1]) \
        * (freq / LIGHTSPEED)[None, :, None]
    return (np.exp(-2j * np.pi * ph) * image.ravel()[None, None, :]).sum(axis=2)


def _case(nx, ny, fov, nrow, nchan, nband, seed=420):
    rng = np.random.default_rng(seed)
    cell = fov * np.pi / 180 / nx
    f0 = 1e9
    freq = f0 + np.arange(nchan) * (f0 / nchan)
    uvw = (rng.random((nrow, 3)) - 0.5) / (cell * freq[-1] / LIGHTSPEED)
    step = nchan // nband
    freq_bin_idx = np.arange(0, nchan, step)
    freq_bin_counts = np.append(freq_bin_idx, nchan)[1:] - np.append(freq_bin_idx, nchan)[:-1]
    image = rng.standard_normal((freq_bin_idx.size, nx, ny))
    return cell, freq, uvw, freq_bin_idx, freq_bin_counts, image


@pytest.mark.parametrize("ny", (18, 64))
@pytest.mark.parametrize("nchan, nband", [(1, 1), (7, 1), (7, 3)])
@pytest.mark.parametrize("epsilon", (1e-3, 1e-4, 1e-6))
def test_model_meets_the_accuracy_contract(ny, nchan, nband, epsilon):
    """test_wgridder.py:49-113 transposed to the degridder: 16 x ny pixels, 5 degree field, 1000 rows"""
    nx, fov, nrow = 16, 5.0, 1000
    cell, freq, uvw, fbi, fbc, image = _case(nx, ny, fov, nrow, nchan, nband)
    vis = model(uvw, freq, image, fbi, fbc, cell, epsilon=epsilon)
    assert vis.shape == (nrow, nchan) and vis.dtype == np.complex128
    ref = np.zeros((nrow, nchan), dtype=np.complex128)
    for b in range(fbi.size):
        ind = slice(fbi[b], fbi[b] + fbc[b])
        ref[:, ind] = _explicit_degridder(uvw, freq[ind], image[b], cell, cell)
    assert _l2error(vis, ref) <= epsilon


def test_model_weights_flags_wide_field_and_no_wstacking():
    nx, ny, nrow, nchan = 24, 20, 700, 4
    cell, freq, uvw, fbi, fbc, image = _case(nx, ny, 25.0, nrow, nchan, 2, seed=7)      # 25 degrees: a strong w term
    celly = cell * 1.3
    rng = np.random.default_rng(3)
    wgt = rng.random((nrow, nchan))
    flag = (rng.random((nrow, nchan)) > 0.2).astype(np.uint8)                # != 0: process
    ref = np.zeros((nrow, nchan), dtype=np.complex128)
    for b in range(2):
        ind = slice(fbi[b], fbi[b] + fbc[b])
        ref[:, ind] = _explicit_degridder(uvw, freq[ind], image[b], cell, celly)
    vis = model(uvw, freq, image, fbi, fbc, cell, weights=wgt, flag=flag, celly=celly, epsilon=1e-7)
    assert np.all(vis[flag == 0] == 0)
    assert _l2error(vis, ref * wgt * (flag != 0)) <= 1e-7
    # without w-stacking: w and n are ignored (ducc0's do_wstacking=False)
    flat = model(uvw, freq, image, fbi, fbc, cell, celly=celly, epsilon=1e-6, do_wstacking=False)
    ref0 = np.zeros_like(ref)
    for b in range(2):
        ind = slice(fbi[b], fbi[b] + fbc[b])
        ref0[:, ind] = _explicit_degridder(uvw, freq[ind], image[b], cell, celly, apply_w=False)
    assert _l2error(flat, ref0) <= 1e-6
    assert _l2error(flat, ref) > 1e-3          # and the w term matters at this field of view
    # float32 images give complex64 (result_type(image, complex64))
    assert model(uvw, freq, image.astype(np.float32), fbi, fbc, cell, epsilon=1e-4).dtype == np.complex64
    # row chunks carry bin starts that do not begin at zero (im2vis.py:33)
    again = model(uvw, freq, image, fbi + 5, fbc, cell, weights=wgt, flag=flag, celly=celly, epsilon=1e-7)
    np.testing.assert_array_equal(again, vis)


def test_model_device_resident_matches_host():
    import torch
    cell, freq, uvw, fbi, fbc, image = _case(20, 32, 3.5, 777, 4, 2, seed=11)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    host = model(uvw, freq, image, fbi, fbc, cell, epsilon=1e-5)
    dev = model(t(uvw), t(freq), t(image), fbi, fbc, cell, epsilon=1e-5)
    np.testing.assert_array_equal(dev.cpu().numpy(), host)
    with pytest.raises(ValueError, match="horizon"):
        model(uvw, freq, image, fbi, fbc, 0.2)
    with pytest.raises(ValueError, match="one entry per band"):
        model(uvw, freq, image, fbi[:1], fbc, cell)
    with pytest.raises(ValueError, match="must be even"):      # as ducc0
        model(uvw, freq, image[:, :19], fbi, fbc, cell)


def _wide_case(nx, ny, nrow, nchan, seed):
    cell, freq, uvw, fbi, fbc, image = _case(nx, ny, 25.0, nrow, nchan, 1, seed=seed)
    rng = np.random.default_rng(seed + 1)
    wgt = rng.random((nrow, nchan))
    flag = (rng.random((nrow, nchan)) > 0.15).astype(np.uint8)
    uvw[5] = 0.0                                            # a row in the middle of every axis
    uvw[6, :2] *= -1.0
    return cell, freq, uvw, fbi, fbc, image, wgt, flag


@pytest.mark.parametrize("nx, ny, epsilon", [(64, 48, 1e-7), (34, 70, 1e-4), (48, 48, 1e-10)])
def test_sorted_tile_path_meets_the_contract_and_matches_the_gather_path(nx, ny, epsilon):
    """Calls of >= 65536 visibilities sort them by (uv tile, w-plane) and stage the tiles through LDS; smaller calls
    gather from memory.  Same planes, same taps: the two agree far inside epsilon, and both meet the contract.  Odd
    sizes, non-square pixels, flags and weights, tiles that wrap around the grid's edges."""
    nrow, nchan = 3000, 24                                  # 72000 visibilities
    cell, freq, uvw, fbi, fbc, image, wgt, flag = _wide_case(nx, ny, nrow, nchan, seed=nx)
    celly = cell * 0.8
    vis = model(uvw, freq, image, fbi, fbc, cell, weights=wgt, flag=flag, celly=celly, epsilon=epsilon)
    ref = _explicit_degridder(uvw, freq, image[0], cell, celly) * wgt * (flag != 0)
    assert np.all(vis[flag == 0] == 0)
    assert _l2error(vis, ref) <= epsilon
    # the gather path on the same planes: the w range of a call fixes the planes, so keep the two extreme rows in
    # every piece
    lo, hi = np.argmin(uvw[:, 2]), np.argmax(uvw[:, 2])
    for a in range(0, nrow, 1000):
        rows = np.unique(np.concatenate([np.arange(a, a + 1000), [lo, hi]]))
        part = model(uvw[rows], freq, image, fbi, fbc, cell, weights=wgt[rows], flag=flag[rows], celly=celly,
                     epsilon=epsilon)
        assert np.abs(part - vis[rows]).max() <= 1e-12 * np.abs(vis).max()


def test_sorted_tile_path_in_plane_batches(monkeypatch):
    """A workspace that holds fewer planes than the call needs: the planes are worked through in batches, one pass of
    the sorted visibilities per batch.  Same result as with every plane resident, to rounding."""
    from codex_africanus_amd.gridding.wgridder import im2vis
    cell, freq, uvw, fbi, fbc, image, wgt, flag = _wide_case(40, 40, 3000, 24, seed=5)
    full = model(uvw, freq, image, fbi, fbc, cell, flag=flag, epsilon=1e-6)
    from codex_africanus_amd import _lib
    nu = int(_lib.load().af_wgrid_padded(40))
    monkeypatch.setattr(im2vis, "PLANE_BUDGET", 3 * nu * nu * 16)
    batched = model(uvw, freq, image, fbi, fbc, cell, flag=flag, epsilon=1e-6)
    assert np.abs(batched - full).max() <= 1e-13 * np.abs(full).max()
    # the gridding direction in the same plane batches (one exact sort and one ring pass per batch)
    from codex_africanus_amd.gridding.wgridder import dirty
    rng = np.random.default_rng(6)
    ms = rng.standard_normal((3000, 24)) + 1j * rng.standard_normal((3000, 24))
    monkeypatch.undo()
    img_full = dirty(uvw, freq, ms, fbi, fbc, 40, 40, cell, flag=flag, epsilon=1e-6)
    monkeypatch.setattr(im2vis, "PLANE_BUDGET", 3 * nu * nu * 16)
    img_batched = dirty(uvw, freq, ms, fbi, fbc, 40, 40, cell, flag=flag, epsilon=1e-6)
    assert np.abs(img_batched - img_full).max() <= 1e-12 * np.abs(img_full).max()
    flat = model(uvw, freq, image, fbi, fbc, cell, epsilon=1e-6, do_wstacking=False)
    assert _l2error(flat, _explicit_degridder(uvw, freq, image[0], cell, cell, apply_w=False)) <= 1e-6


# ---------------------------------------------------------------------------------------------------------------
# the adjoint direction: dirty, residual, hessian (africanus/gridding/wgridder/{vis2im,im2residim,hessian}.py)

def _explicit_gridder(uvw, freq, ms, wgt, nx, ny, cell, celly, apply_w=True, mask=None):
    """test_wgridder.py:18-46 through the oracle's vis_to_im (sources = pixels, w negated, summed over channels)."""
    x, y = np.meshgrid(*[-ss / 2 + np.arange(ss) for ss in (nx, ny)], indexing="ij")
    x, y = x * cell, y * celly
    v = ms if wgt is None else ms * wgt
    if mask is not None:
        v = v * (mask != 0)
    if apply_w:
        lm = np.stack([x.ravel(), y.ravel()], axis=1)
        n = np.sqrt(1.0 - x ** 2 - y ** 2)
        im = oracle.vis_to_im(v[:, :, None].astype(np.complex128), uvw * np.array([1.0, 1.0, -1.0]), lm, freq,
                              np.zeros(v.shape + (1,), dtype=np.uint8), omp=True)
        return im[:, :, 0].sum(axis=1).reshape(nx, ny) / n
    ph = (uvw[:, None, 0, None] * x.ravel()[None, None, :] + uvw[:, None, 1, None] * y.ravel()[None, None, :]) \
        * (freq / LIGHTSPEED)[None, :, None]
    return (np.exp(2j * np.pi * ph) * v[:, :, None]).real.sum(axis=(0, 1)).reshape(nx, ny)


@pytest.mark.parametrize("ny", (18, 64))
@pytest.mark.parametrize("nchan, nband", [(1, 1), (7, 1), (7, 3)])
@pytest.mark.parametrize("precision, epsilon", [("single", 1e-3), ("single", 1e-4), ("double", 1e-3), ("double", 1e-4),
                                                ("double", 1e-7)])
def test_dirty_meets_the_accuracy_contract(ny, nchan, nband, precision, epsilon):
    """test_wgridder.py:48-108 (test_gridder): 16 x ny pixels, 5 degree field, 1000 rows, weights"""
    from codex_africanus_amd.gridding.wgridder import dirty
    nx, fov, nrow = 16, 5.0, 1000
    cell, freq, uvw, fbi, fbc, _ = _case(nx, ny, fov, nrow, nchan, nband)
    rng = np.random.default_rng(40)
    ms = rng.standard_normal((nrow, nchan)) + 1j * rng.standard_normal((nrow, nchan))
    wgt = rng.random((nrow, nchan))
    ctype, rtype = (np.complex64, np.float32) if precision == "single" else (np.complex128, np.float64)
    img = dirty(uvw, freq, ms.astype(ctype), fbi, fbc, nx, ny, cell, weights=wgt.astype(rtype), epsilon=epsilon)
    assert img.shape == (fbi.size, nx, ny) and img.dtype == rtype
    ms_used = ms.astype(ctype).astype(np.complex128)
    for b in range(fbi.size):
        ind = slice(fbi[b], fbi[b] + fbc[b])
        ref = _explicit_gridder(uvw, freq[ind], ms_used[:, ind], wgt.astype(rtype).astype(np.float64)[:, ind], nx, ny, cell, cell)
        assert _l2error(img[b], ref) <= max(epsilon, 3e-7 if precision == "single" else 0)


@pytest.mark.parametrize("nrow, nchan, tiled", [(600, 4, False), (6000, 24, True)])
def test_dirty_is_the_adjoint_of_model(nrow, nchan, tiled):
    """test_wgridder.py:111-188: <R x, v> == <x, R^H v> -- here to rounding, because both directions use the same
    planes and taps; with flags, weights applied on one side each, two bands, non-square pixels; the small call takes
    the per-visibility kernels, the large one the sorted tile kernels."""
    from codex_africanus_amd.gridding.wgridder import dirty
    nx, ny, nband = 30, 50, 2
    cell, freq, uvw, fbi, fbc, image = _case(nx, ny, 12.0, nrow, nchan, nband, seed=2)
    assert (nrow * (nchan // nband) >= 65536) == tiled
    celly = cell * 1.2
    rng = np.random.default_rng(9)
    ms = rng.standard_normal((nrow, nchan)) + 1j * rng.standard_normal((nrow, nchan))
    wgt = rng.random((nrow, nchan))
    flag = (rng.random((nrow, nchan)) > 0.1).astype(np.uint8)
    for eps in (1e-4, 1e-9):
        vis = model(uvw, freq, image, fbi, fbc, cell, weights=wgt, flag=flag, celly=celly, epsilon=eps)
        img = dirty(uvw, freq, ms, fbi, fbc, nx, ny, cell, weights=wgt, flag=flag, celly=celly, epsilon=eps)
        lhs = np.vdot(ms, vis).real           # Re <v, R x>
        rhs = np.sum(image * img)             # <R^H v, x>
        assert abs(lhs - rhs) <= 1e-11 * max(abs(lhs), np.abs(image).sum() * np.abs(img).max())
    # and the contract for the large call
    if tiled:
        ref = _explicit_gridder(uvw, freq[:fbc[0]], ms[:, :fbc[0]], wgt[:, :fbc[0]], nx, ny, cell, celly, mask=flag[:, :fbc[0]])
        assert _l2error(img[0], ref) <= 1e-9


@pytest.mark.parametrize("nrow, nchan, tiled", [(700, 4, False), (9000, 16, True)])
def test_float32_planes_for_single_precision_images_and_on_request(nrow, nchan, tiled):
    """float32 w-planes (float32 FFTs, fp64 sums; csrc/af_wgridder.hip): taken by float32 images -- the reference's
    single-precision call, whose tests ask l2 <= max(epsilon, 3e-7) and adjointness to 1e-4
    (test_wgridder.py:55-108,125-188) -- and by ``plane_precision("single")``; only for epsilon >= 1e-5 (a finer request
    keeps fp64 planes whatever the mode); float64 images keep fp64 planes and adjointness to rounding by default.
    Small call = the per-visibility kernel, large call = the sorted tile kernel."""
    from codex_africanus_amd.gridding.wgridder import dirty, plane_precision
    nx, ny, nband = 32, 48, 2
    cell, freq, uvw, fbi, fbc, image = _case(nx, ny, 8.0, nrow, nchan, nband, seed=11)
    assert (nrow * (nchan // nband) >= 65536) == tiled
    ref = np.zeros((nrow, nchan), dtype=np.complex128)
    for b in range(fbi.size):
        ind = slice(fbi[b], fbi[b] + fbc[b])
        ref[:, ind] = _explicit_degridder(uvw, freq[ind], image[b], cell, cell)
    base = model(uvw, freq, image, fbi, fbc, cell, epsilon=1e-5)
    with plane_precision("single"):
        opt = model(uvw, freq, image, fbi, fbc, cell, epsilon=1e-5)
        fine = model(uvw, freq, image, fbi, fbc, cell, epsilon=1e-7)        # W = 9: fp64 planes whatever the mode
    again = model(uvw, freq, image, fbi, fbc, cell, epsilon=1e-5)           # the mode ended with the block
    assert _l2error(base, ref) <= 1e-5 and _l2error(opt, ref) <= 1e-5 and _l2error(fine, ref) <= 1e-7
    assert np.array_equal(again, base) and not np.array_equal(opt, base)
    assert 1e-9 < _l2error(opt, base) < 3e-6                               # float32 planes were used, and cost ~1e-7..1e-6
    assert np.array_equal(fine, model(uvw, freq, image, fbi, fbc, cell, epsilon=1e-7))
    # a float32 image: complex64 result, float32 planes, the single-precision contract
    img32 = image.astype(np.float32)
    v32 = model(uvw, freq, img32, fbi, fbc, cell, epsilon=1e-5)
    assert v32.dtype == np.complex64
    ref32 = np.zeros_like(ref)
    for b in range(fbi.size):
        ind = slice(fbi[b], fbi[b] + fbc[b])
        ref32[:, ind] = _explicit_degridder(uvw, freq[ind], img32[b].astype(np.float64), cell, cell)
    assert _l2error(v32, ref32) <= max(1e-5, 3e-7)
    # adjointness: to rounding for the default, to single precision with float32 planes (reference tolerances 1e-12 / 1e-4)
    rng = np.random.default_rng(3)
    ms = rng.standard_normal((nrow, nchan)) + 1j * rng.standard_normal((nrow, nchan))
    img = dirty(uvw, freq, ms, fbi, fbc, nx, ny, cell, epsilon=1e-5)
    rhs = np.sum(image * img)
    for vis, tol in ((base, 1e-11), (opt, 1e-5)):
        lhs = np.vdot(ms, vis).real
        assert abs(lhs - rhs) <= tol * max(abs(lhs), np.abs(image).sum() * np.abs(img).max()), tol


def test_the_three_visibility_sorts_give_the_same_visibilities(monkeypatch):
    """The sorted tile path orders the (row, chan) visibilities by (tile, w bucket) on the device: two levels with every
    atomic in LDS (default), rank + place with global atomics (AFHIP_WGRID_SORT1=1), count + scatter (=0).  The order
    inside a bucket differs, the result of a visibility does not depend on it: bit-equal outputs; both directions."""
    from codex_africanus_amd.gridding.wgridder import dirty
    nx, ny, nrow, nchan = 96, 64, 9000, 16
    cell, freq, uvw, fbi, fbc, image = _case(nx, ny, 6.0, nrow, nchan, 2, seed=21)
    rng = np.random.default_rng(4)
    flag = (rng.random((nrow, nchan)) > 0.05).astype(np.uint8)
    ms = rng.standard_normal((nrow, nchan)) + 1j * rng.standard_normal((nrow, nchan))
    out = {}
    for mode in ("2", "1", "0"):
        monkeypatch.setenv("AFHIP_WGRID_SORT1", mode)
        out[mode] = (model(uvw, freq, image, fbi, fbc, cell, flag=flag, epsilon=1e-6),
                     dirty(uvw, freq, ms, fbi, fbc, nx, ny, cell, flag=flag, epsilon=1e-6))
    for mode in ("1", "0"):
        assert np.array_equal(out[mode][0], out["2"][0])
        # the adjoint adds the visibilities of a cell in sorted order: equal to rounding, not to the bit
        assert np.abs(out[mode][1] - out["2"][1]).max() <= 1e-12 * np.abs(out["2"][1]).max()


def test_residual_and_hessian_compose_model_and_dirty():
    """test_wgridder.py:191-354: residual = dirty(vis - model(image)) with the weights on the imaging side only;
    hessian = dirty(model(image)); results in the image's dtype; torch inputs give torch outputs."""
    import torch
    from codex_africanus_amd.gridding.wgridder import dirty, residual, hessian
    nx, ny, nrow, nchan, nband = 20, 32, 900, 5, 2
    cell, freq, uvw, fbi, fbc, image = _case(nx, ny, 3.5, nrow, nchan, nband, seed=4)
    rng = np.random.default_rng(5)
    ms = rng.standard_normal((nrow, nchan)) + 1j * rng.standard_normal((nrow, nchan))
    wgt = rng.random((nrow, nchan))
    flag = (rng.random((nrow, nchan)) > 0.1).astype(np.uint8)
    mv = model(uvw, freq, image, fbi, fbc, cell, flag=flag, epsilon=1e-7)
    res = residual(uvw, freq, image, ms, fbi, fbc, cell, weights=wgt, flag=flag, epsilon=1e-7)
    want = dirty(uvw, freq, ms - mv, fbi, fbc, nx, ny, cell, weights=wgt, flag=flag, epsilon=1e-7)
    assert res.shape == image.shape and res.dtype == image.dtype
    assert np.abs(res - want).max() <= 1e-12 * np.abs(want).max()
    hes = hessian(uvw, freq, image, fbi, fbc, cell, weights=wgt, flag=flag, epsilon=1e-7)
    want = dirty(uvw, freq, mv, fbi, fbc, nx, ny, cell, weights=wgt, flag=flag, epsilon=1e-7)
    assert np.abs(hes - want).max() <= 1e-12 * np.abs(want).max()
    # the Hessian is symmetric positive semi-definite: <y, H x> == <H y, x>, <x, H x> >= 0
    other = rng.standard_normal(image.shape)
    h2 = hessian(uvw, freq, other, fbi, fbc, cell, weights=wgt, flag=flag, epsilon=1e-7)
    assert abs(np.sum(other * hes) - np.sum(h2 * image)) <= 1e-10 * abs(np.sum(other * hes))
    assert np.sum(image * hes) >= 0
    f32 = residual(uvw, freq, image.astype(np.float32), ms, fbi, fbc, cell, weights=wgt, flag=flag, epsilon=1e-4)
    assert f32.dtype == np.float32
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dev = hessian(t(uvw), t(freq), t(image), fbi, fbc, cell, weights=t(wgt), flag=t(flag), epsilon=1e-7)
    assert isinstance(dev, torch.Tensor) and np.abs(dev.cpu().numpy() - hes).max() <= 1e-12 * np.abs(hes).max()
    with pytest.raises(ValueError, match="incorrect type"):
        dirty(uvw, freq, ms.real, fbi, fbc, nx, ny, cell)
    # no rows: a zero image
    assert not dirty(uvw[:0], freq, ms[:0], fbi, fbc, nx, ny, cell).any()


def test_hundreds_of_w_planes_are_sorted_per_batch():
    """A wide field with long w: more first planes than one exact sort has buckets for (256), so the gridding direction
    sorts and grids per batch of planes; the other direction sorts once with plane buckets.  Contract and adjointness."""
    from codex_africanus_amd import _lib
    from codex_africanus_amd.gridding.wgridder import dirty
    nx, ny, nrow, nchan = 32, 32, 6000, 12
    cell, freq, uvw, fbi, fbc, image = _case(nx, ny, 50.0, nrow, nchan, 1, seed=8)
    uvw[:, 2] *= 20.0          # (the planes cover |w| only: the w fold)
    eps = 1e-6
    emax = 2 * (nx / 2 * cell) ** 2
    wl = np.abs(uvw[:, 2]).max() * freq.max() / LIGHTSPEED
    assert _lib.load().af_wgrid_planes(-wl, wl, emax / (np.sqrt(1 - emax) + 1), 8, 1) > 300
    rng = np.random.default_rng(3)
    ms = rng.standard_normal((nrow, nchan)) + 1j * rng.standard_normal((nrow, nchan))
    vis = model(uvw, freq, image, fbi, fbc, cell, epsilon=eps)
    assert _l2error(vis, _explicit_degridder(uvw, freq, image[0], cell, cell)) <= eps
    img = dirty(uvw, freq, ms, fbi, fbc, nx, ny, cell, epsilon=eps)
    assert _l2error(img[0], _explicit_gridder(uvw, freq, ms, None, nx, ny, cell, cell)) <= eps
    lhs, rhs = np.vdot(ms, vis).real, np.sum(image * img)
    assert abs(lhs - rhs) <= 1e-11 * np.abs(image).sum() * np.abs(img).max()


def test_concurrent_threads_do_not_share_fft_scratch():
    """hipFFT plans used to be cached per (device, n, batch) and shared by every host thread, each
    on its own stream -- two dask workers transforming row chunks of the same shape on one device then ran ONE plan's
    work buffer concurrently (rocFFT needs it for multi-kernel lengths: nu = 2 nx >= 4100).  Plans are per stream now:
    six threads run model + dirty of the same 2052-pixel-wide shape at once and must reproduce the serial results
    (model bit for bit: gathers and transforms are deterministic; dirty to rounding: its small-call spreading uses
    atomics whose order varies; shared scratch shows up as corrupted planes, orders of magnitude above either)."""
    from concurrent.futures import ThreadPoolExecutor
    from codex_africanus_amd.gridding.wgridder import dirty
    nx, ny, nrow, nchan = 2052, 40, 400, 2
    cell, freq, uvw, fbi, fbc, _ = _case(nx, ny, 2.0, nrow, nchan, 1, seed=11)
    rng = np.random.default_rng(5)
    images = [rng.standard_normal((1, nx, ny)) for _ in range(6)]
    viss = [rng.standard_normal((nrow, nchan)) + 1j * rng.standard_normal((nrow, nchan)) for _ in range(6)]

    def one(k):
        v = model(uvw, freq, images[k], fbi, fbc, cell, epsilon=1e-6)
        d = dirty(uvw, freq, viss[k], fbi, fbc, nx, ny, cell, epsilon=1e-6)
        return v, d

    serial = [one(k) for k in range(6)]
    for _ in range(3):
        with ThreadPoolExecutor(6) as ex:
            got = list(ex.map(one, range(6)))
        for (v0, d0), (v1, d1) in zip(serial, got):
            assert np.array_equal(v0, v1)
            assert np.abs(d0 - d1).max() <= 1e-11 * np.abs(d0).max()


def test_model_with_caller_supplied_w_bounds_and_epsilon_floor():
    """VERDICT r2 item 9 / ADVICE r2: a device-resident call with ``w_bounds`` needs no host read-back of the w range
    and gives the same visibilities (a superset of the range only adds planes: still within epsilon); an epsilon
    below the float64 floor raises instead of being clamped to 16 taps."""
    import torch
    nx, ny, nrow, nchan = 32, 32, 500, 3
    cell, freq, uvw, fbi, fbc, image = _case(nx, ny, 10.0, nrow, nchan, 1, seed=9)
    ref = model(uvw, freq, image, fbi, fbc, cell, epsilon=1e-6)
    dev = torch.device("cuda:0")
    got = model(torch.from_numpy(uvw).to(dev), freq, torch.from_numpy(image).to(dev), fbi, fbc, cell, epsilon=1e-6,
                w_bounds=(uvw[:, 2].min(), uvw[:, 2].max()))
    assert np.array_equal(got.cpu().numpy(), ref)
    wide = model(uvw, freq, image, fbi, fbc, cell, epsilon=1e-6, w_bounds=(1.5 * uvw[:, 2].min(), 1.5 * uvw[:, 2].max()))
    assert _l2error(wide, ref) <= 2e-6
    with pytest.raises(ValueError):
        model(uvw, freq, image, fbi, fbc, cell, epsilon=1e-16)
    with pytest.raises(ValueError):
        model(uvw, freq, image, fbi, fbc, cell, w_bounds=(1.0, -1.0))


@pytest.mark.parametrize("nx, ny", [(16, 512), (64, 512), (512, 512), (512, 20), (16, 1024), (2048, 16), (16, 2048),
                                    (1024, 2048), (16, 8192), (8192, 16)])
def test_fused_fill_and_first_transform_equals_the_hipfft_route(nx, ny, monkeypatch):
    """rows of 512, 1024, 2048, 8192 (and 4096: tests/test_gpu_full_size.py) image cells -- radix 8 throughout, or a radix-2 / radix-4
    first pass: the fill pass and the transform along v in
    one kernel (wg_fill_fft_rows: two half-length Stockham transforms of the row, no zero ever stored).  Same visibilities
    as wg_fill_rows + hipFFT to rounding, and the accuracy contract against the direct transform on a sparse image."""
    nrow, nchan = 3000, 3
    cell, freq, uvw, fbi, fbc, image = _case(nx, ny, 3.0 * nx / max(nx, ny, 512), nrow, nchan, 1, seed=11)   # <= 3 degrees across the longer side
    image[0][np.random.default_rng(1).random((nx, ny)) < 1.0 - 800.0 / (nx * ny)] = 0.0   # sparse: the direct transform stays cheap
    monkeypatch.setenv("AFHIP_WGRID_FFT1", "0")          # (512 rows: the second transform takes the same kernel,
    monkeypatch.setenv("AFHIP_WGRID_FFT2", "0")          #  its rows read from a compact transposition)
    ref = model(uvw, freq, image, fbi, fbc, cell, celly=cell * 0.9, epsilon=1e-7)
    monkeypatch.delenv("AFHIP_WGRID_FFT1")
    monkeypatch.delenv("AFHIP_WGRID_FFT2")
    vis = model(uvw, freq, image, fbi, fbc, cell, celly=cell * 0.9, epsilon=1e-7)
    assert np.abs(vis - ref).max() <= 1e-12 * np.abs(ref).max()
    nz = np.nonzero(image[0])
    x, y = (nz[0] - nx / 2) * cell, (nz[1] - ny / 2) * cell * 0.9
    n = np.sqrt(1.0 - x * x - y * y)
    src = np.broadcast_to((image[0][nz] / n)[:, None, None], (x.size, nchan, 1)).copy()
    direct = oracle.im_to_vis(src, uvw * np.array([1.0, 1.0, -1.0]), np.stack([x, y], 1), freq, omp=True)[:, :, 0]
    assert _l2error(vis, direct) <= 1e-7


def test_own_row_transforms_with_float32_planes(monkeypatch):
    """float32 planes of a 512 x 512 image: the own row transforms compute in fp64 and round once on their way out, so they
    sit CLOSER to the fp64-plane result than hipFFT's float32 transforms do; both meet the contract at epsilon = 1e-5."""
    from codex_africanus_amd.gridding.wgridder import plane_precision
    nx = ny = 512
    nrow, nchan = 3000, 2
    cell, freq, uvw, fbi, fbc, image = _case(nx, ny, 3.0, nrow, nchan, 1, seed=5)
    image[0][np.random.default_rng(2).random((nx, ny)) < 0.97] = 0.0
    base = model(uvw, freq, image, fbi, fbc, cell, epsilon=1e-5)
    with plane_precision("single"):
        own = model(uvw, freq, image, fbi, fbc, cell, epsilon=1e-5)
        monkeypatch.setenv("AFHIP_WGRID_FFT1", "0")
        monkeypatch.setenv("AFHIP_WGRID_FFT2", "0")
        lib = model(uvw, freq, image, fbi, fbc, cell, epsilon=1e-5)
    assert not np.array_equal(own, base) and not np.array_equal(own, lib)
    assert _l2error(own, base) <= _l2error(lib, base) < 3e-6
    nz = np.nonzero(image[0])
    x, y = (nz[0] - nx / 2) * cell, (nz[1] - ny / 2) * cell
    n = np.sqrt(1.0 - x * x - y * y)
    src = np.broadcast_to((image[0][nz] / n)[:, None, None], (x.size, nchan, 1)).copy()
    direct = oracle.im_to_vis(src, uvw * np.array([1.0, 1.0, -1.0]), np.stack([x, y], 1), freq, omp=True)[:, :, 0]
    assert _l2error(own, direct) <= 1e-5 and _l2error(lib, direct) <= 1e-5


@pytest.mark.parametrize("nrow", (900, 40000))          # the gather kernel / the tile kernel (>= 65536 visibilities)
def test_w_fold_one_sign_both_signs_and_the_mirror_property(nrow):
    """Visibilities with w < 0 are evaluated at the mirrored point and conjugated (real image: V(-u,-v,-w) = conj V),
    so the planes cover the range of |w| only.  All-negative, all-positive and mixed w meet the accuracy contract; the
    mirrored call returns the conjugate BIT FOR BIT (same planes, same taps); the plane count of a symmetric range is
    that of its positive half."""
    from codex_africanus_amd import _lib
    nx, ny, nchan = 24, 20, 3
    cell, freq, uvw, fbi, fbc, image = _case(nx, ny, 20.0, nrow, nchan, 1, seed=31)
    sample = np.random.default_rng(5).choice(nrow, 600, replace=False)
    for sign in (-1.0, 1.0, 0.0):
        u = uvw.copy()
        if sign:
            u[:, 2] = sign * (np.abs(u[:, 2]) + 0.1 * np.abs(u[:, 2]).max())       # one sign, away from zero
        vis = model(u, freq, image, fbi, fbc, cell, epsilon=1e-6)
        ref = _explicit_degridder(u[sample], freq, image[0], cell, cell)
        assert _l2error(vis[sample], ref) <= 1e-6, sign
        mirrored = model(-u, freq, image, fbi, fbc, cell, epsilon=1e-6)
        assert np.array_equal(mirrored, np.conj(vis)), sign
    lib = _lib.load()
    assert lib.af_wgrid_planes(-5e3, 5e3, 1e-3, 7, 1) == lib.af_wgrid_planes(0.0, 5e3, 1e-3, 7, 1)
    assert lib.af_wgrid_planes(-5e3, -2e3, 1e-3, 7, 1) == lib.af_wgrid_planes(2e3, 5e3, 1e-3, 7, 1)
    assert lib.af_wgrid_planes(-5e3, 5e3, 1e-3, 7, 1) < lib.af_wgrid_planes(0.0, 1e4, 1e-3, 7, 1)


def test_many_caller_streams_share_nothing_and_overflow_gracefully():
    """The sort of a large image -> vis call runs on a library-owned side stream per caller stream; past 64 caller streams
    the least recently used entry that no call is using is synchronised and destroyed (round 6; until then such a call lost
    the overlap for good), and only when every entry is in use does a call sort on the caller's stream.  70 torch streams,
    the same bits from each -- twice, so that evicted entries come back."""
    import torch
    nx, ny, nrow, nchan = 24, 20, 30000, 3            # 90000 visibilities: the tile kernel, hence the sorted path
    cell, freq, uvw, fbi, fbc, image = _case(nx, ny, 20.0, nrow, nchan, 1, seed=77)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    d = [t(uvw), t(freq), t(image)]
    ref = model(*d, fbi, fbc, cell, epsilon=1e-6).cpu().numpy()
    streams = [torch.cuda.Stream() for _ in range(70)]
    outs = []
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            outs.append(model(*d, fbi, fbc, cell, epsilon=1e-6))
    for s in streams[:10] + streams[60:]:                 # entries evicted above are made again, others evicted in turn
        with torch.cuda.stream(s):
            outs.append(model(*d, fbi, fbc, cell, epsilon=1e-6))
    torch.cuda.synchronize()
    for o in outs:
        assert np.array_equal(o.cpu().numpy(), ref)
