"""GPU tests of the chunk contract: chunked == unchunked for the reference's
chunkings (africanus/rime/tests/test_predict.py:20-31, africanus/dft/tests/test_dft.py:218-250),
streams=True and False.  The dask front-ends themselves run in tests/test_gpu_dask_conda.py (tests/dask_cases.py, under an
interpreter that has dask) and, without dask, through the block-contract emulator in tests/test_gpu_blocks.py."""
import numpy as np
import pytest
from numpy.testing import assert_array_equal, assert_array_almost_equal

from codex_africanus_amd import chunked, rime, dft

pytestmark = pytest.mark.gpu

CHUNKS = {"source": (2, 3, 4, 2, 2, 2, 2, 2, 2), "time": (2, 1, 1), "row": (4, 4, 2), "chan": (3, 2)}
CORR = {"c1": (1,), "c2": (2,), "c22": (2, 2)}
DDE = {"ddecoh": (True, True, True), "dde": (True, False, True), "coh": (False, True, False)}
DIE = {"diebv": (True, True, True), "die": (True, False, True), "bv": (False, True, False)}


@pytest.mark.parametrize("ck", list(CORR))
@pytest.mark.parametrize("dk", list(DDE))
@pytest.mark.parametrize("gk", list(DIE))
@pytest.mark.parametrize("streams", [True, False])
def test_chunked_predict_vis_equals_unchunked(g2, ck, dk, gk, streams):
    get = lambda k: g2["%s_%s" % (ck, k)]
    ti, a1, a2 = g2["time_idx"], g2["ant1"], g2["ant2"]
    a1j, blj, a2j = DDE[dk]
    g1j, bvis, g2j = DIE[gk]
    args = (get("a1") if a1j else None, get("bl") if blj else None, get("a2") if a2j else None,
            get("g1") if g1j else None, get("bv") if bvis else None, get("g2") if g2j else None)
    out = chunked.predict_vis(ti, a1, a2, *args, streams=streams, chunks=CHUNKS)
    ref = g2["%s_%s_%s_vis" % (ck, dk, gk)]
    assert out.shape == ref.shape
    assert_array_almost_equal(out, ref, decimal=12)
    if streams and not (g1j or bvis):
        # the serial chain adds sources in the reference's ascending order: bit-identical
        assert_array_equal(out, ref)


def test_chunked_phase_delay_exact(g1):
    out = chunked.phase_delay(g1["lm"], g1["uvw"], g1["frequency"], chunks={"source": 3, "row": 10, "chan": 2})
    assert_array_equal(out, rime.phase_delay(g1["lm"], g1["uvw"], g1["frequency"]))


def test_chunked_im_to_vis(g3):
    """rows chunked as in africanus/dft/tests/test_dft.py:240-243; row chunking is exact."""
    out = chunked.im_to_vis(g3["img_r4"], g3["uvw"], g3["lm"], g3["frequency"], chunks={"row": 7})
    assert_array_equal(out, dft.im_to_vis(g3["img_r4"], g3["uvw"], g3["lm"], g3["frequency"]))
    out2 = chunked.im_to_vis(g3["img_r70"], g3["uvw"], g3["lm"], g3["frequency70"],
                             chunks={"row": 20, "chan": 26})
    assert_array_almost_equal(out2, g3["vis_r70_fourier"], decimal=11)


def test_chunked_vis_to_im():
    from conftest import load_golden
    g6 = load_golden("g6_vis_to_im.npz")
    ref = g6["im70"]
    out = chunked.vis_to_im(g6["vis70"], g6["uvw300"], g6["lm"], g6["frequency70"], g6["flags70"],
                            chunks={"row": 77, "chan": 26})
    # same bound as the unchunked parity test: 1e-11 relative to sum_r |vis|
    assert np.abs(out - ref).max() <= 1e-11 * np.abs(g6["vis70"]).sum(axis=0).max()


def test_numpy_result_downloaded_in_overlapped_row_chunks(monkeypatch):
    """the numpy-in / numpy-out im_to_vis produces its result in row chunks whose downloads (copy
    stream, behind an event) overlap the next chunk's transform.  Same bits as the one-piece call, ragged last chunk,
    and dtype=complex64 converted on the device before the copy (= numpy's rounding of the complex128 result)."""
    from codex_africanus_amd import dft
    from codex_africanus_amd.testing import synthetic_inputs, real_image
    d = synthetic_inputs(seed=12, nrow=70001, nchan=64, nsrc=30, nant=7)
    image = real_image(d)
    monkeypatch.setenv("AFHIP_D2H_PIPELINE", "0")
    whole = dft.im_to_vis(image, d["uvw"], d["lm"], d["frequency"])
    monkeypatch.delenv("AFHIP_D2H_PIPELINE")
    monkeypatch.setenv("AFHIP_D2H_CHUNK_MB", "16")           # 70001 rows x 4 KB = 287 MB: 18 chunks of 4096 rows
    piped = dft.im_to_vis(image, d["uvw"], d["lm"], d["frequency"])
    assert piped.dtype == np.complex128 and np.array_equal(piped, whole)
    c64 = dft.im_to_vis(image, d["uvw"], d["lm"], d["frequency"], dtype=np.complex64)
    assert c64.dtype == np.complex64 and np.array_equal(c64, whole.astype(np.complex64))
    cplx = dft.im_to_vis(image * (1 + 0.5j), d["uvw"], d["lm"], d["frequency"])
    monkeypatch.setenv("AFHIP_D2H_PIPELINE", "0")
    assert np.array_equal(cplx, dft.im_to_vis(image * (1 + 0.5j), d["uvw"], d["lm"], d["frequency"]))
