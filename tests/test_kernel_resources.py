"""
Build-time invariant of the direct-transform kernels: their record-group refreshes are issued from
inline asm and retired by hand-counted ``s_waitcnt vmcnt(N)``; a compiler spill to scratch inside
those loops would add vector-memory operations the counts do not know about.  Every instantiation
must therefore compile with zero scratch and zero spilled registers (CPU-only check: hipcc
cross-compiles gfx950 without a GPU).
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "codex_africanus_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
SGPR_SPILL_OK = ("af_fused_predict.hip", "af_degridder.hip", "af_calibration.hip", "af_wgridder.hip")


@pytest.mark.parametrize("source, kernels, min_seen, reg_cap", [
    ("af_im_to_vis.hip", ("dft_recurrence_dpp_kernel", "dft_recurrence_dpp4_kernel"), 8, 256),
    ("af_vis_to_im.hip", ("v2i_recurrence_kernel",), 8, 256),
    ("af_wsclean_predict.hip", ("wsc_recurrence_kernel",), 5, 512),
    # MFMA-accumulator kernels: the 64-channel tile is one wave per SIMD by design (256 AGPRs of
    # accumulators + arch VGPRs); its global -> LDS copies are asm-issued and hand-waited as well
    ("af_im_to_vis_mfma.hip", ("dft_mfma_kernel",), 3, 512),
    ("af_vis_to_im_mfma.hip", ("v2i_mfma_kernel",), 3, 512),
    # kernels tuned to a register budget (no counted waits, but a spill would silently cost the occupancy their
    # design rests on): fused predict 8-wave (256 at 2 waves/SIMD), cooperative degridder, calibration consumers
    ("af_fused_predict.hip", ("fused_predict_kernelILb0ELb0E", "fused_predict_kernelILb1ELb0E",
                              "fused_predict_kernelILb0ELb1E", "fused_predict_kernelILb1ELb1E"), 12, 256),
    ("af_degridder.hip", ("degrid_coop_kernel", "degrid_kernel"), 4, 256),
    # round 3.  predict_vis (row block, chan tile): asm-issued LDS copies retired by counted vmcnt waits; the default
    # instantiations (one cell per lane) must leave two 512-lane workgroups per CU: <= 128 registers
    ("af_predict_vis.hip", ("predict_vis_tile_kernelIdiLi4ELb1ELb1ELi4ELi512ELi1E", "predict_vis_tile_kernelIflLi4ELb1ELb1E",
                            "predict_vis_tile_kernelIdlLi2ELb0ELb0E"), 3, 128),
    # ... and its streamed form (every load of the loop an asm-issued LDS copy, counted waits): a spill would add
    # vector-memory operations to the count; one 512-lane workgroup per CU = two waves per SIMD: <= 256 registers
    ("af_predict_vis.hip", ("predict_vis_stream_kernel",), 16, 256),
    # single-precision transforms: accumulator quads + phasor arrays fit two waves per SIMD
    ("af_im_to_vis_f32.hip", ("dft_f32_kernelILi16E", "dft_f32_kernelILi15E", "v2i_f32_kernelILi32E"), 10, 256),
    ("af_calibration.hip", ("calib_kernel",), 16, 256),
    # round 6.  The single-precision lane-per-row fused predict: 12 waves at <= 168 registers (it holds <= 128), no scratch
    ("af_fused_predict_c64.hip", ("fused_rows_c64_kernel",), 20, 168),
    # the wgridder's tile pass at configs[4]'s kernel width, both plane precisions: four waves per SIMD (<= 128 registers)
    # WITHOUT scratch -- an edit that changed nothing but the shape of the code around its loop once made the compiler aim
    # the fp64 instantiation at five waves with 112 bytes of scratch: 7.0 -> 19.9 ms
    ("af_wgridder.hip", ("wg_degrid_tilesILi7E",), 2, 128),
])
def test_counted_wait_kernels_have_no_scratch(tmp_path, source, kernels, min_seen, reg_cap):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
           "--cuda-device-only", "-c", os.path.join(CSRC, source), "-o", str(tmp_path / "k.o"),
           "-Rpass-analysis=kernel-resource-usage"]
    text = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    blocks = re.split(r"remark: [^\n]*Function Name: ", text)[1:]
    seen = 0
    for b in blocks:
        name = b.split()[0]
        if not any(k in name for k in kernels):
            continue
        seen += 1
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        vspill = int(re.search(r"VGPRs Spill: (\d+)", b).group(1))
        sspill = int(re.search(r"SGPRs Spill: (\d+)", b).group(1))
        vgprs = int(re.search(r"VGPRs: (\d+)", b).group(1))
        agprs = int(re.search(r"AGPRs: (\d+)", b).group(1))
        # SGPR spills go to VGPR lanes, not memory: they matter only for the counted-wait kernels' tight loops
        # the 12-wave fused predict with Gaussian shapes on the grouped plan <FEED, GAUSS=true, NP, WS=true, ST=0, GR=true>
        # holds 168 registers and keeps six values (28 bytes) in scratch; no counted waits there
        spill_ok = 32 if re.search(r"fused_predict_kernelILb[01]ELb1ELi\d+ELb1ELi0ELb1E", name) else 0
        assert scratch <= spill_ok and vspill <= spill_ok // 4 and (sspill == 0 or source in SGPR_SPILL_OK), \
            (name, scratch, vspill, sspill)
        assert vgprs + agprs <= reg_cap, (name, vgprs, agprs)   # 256: two waves per SIMD
    assert seen >= min_seen, "expected the template instantiations, found %d" % seen


def test_wave_specialised_fused_kernel_fits_three_waves_per_simd(tmp_path):
    """The 12-wave fused predict (8 accumulating + 4 sampling waves) needs <= 168 registers and no scratch in the
    variants without Gaussian shapes; the Gaussian ones (rolled source loop) may keep a few spilled values outside
    the hot loop."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
           "--cuda-device-only", "-c", os.path.join(CSRC, "af_fused_predict.hip"), "-o", str(tmp_path / "k.o"),
           "-Rpass-analysis=kernel-resource-usage"]
    text = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    blocks = re.split(r"remark: [^\n]*Function Name: ", text)[1:]
    seen = 0
    for b in blocks:
        name = b.split()[0]
        # <FEED, GAUSS=false, NP, WS=true, ST, GR>
        m = re.search(r"fused_predict_kernelILb([01])ELb0ELi(\d+)ELb1ELi(\d+)ELb([01])E", name)
        if not m:
            continue
        seen += 1
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        vgprs = int(re.search(r"VGPRs: (\d+)", b).group(1))
        agprs = int(re.search(r"AGPRs: (\d+)", b).group(1))
        assert scratch == 0 and vgprs + agprs <= 168, (name, scratch, vgprs, agprs)
    assert seen == 16, seen     # (NP in {64, 128, run-time} + the unrolled ST = 8 at NP = 64) x FEED x grouped / row plan


def test_gemm_form_of_the_fused_predict_fits_three_waves_per_simd(tmp_path):
    """The 12-wave GEMM-form kernel (8 matrix waves with up to five 16 x 16 tiles x three accumulators, 4 sampling waves
    with four rounds of gathers in flight) lives on 168 registers; the default (3M) instantiations may keep at most a
    couple of values in scratch outside the matrix loop (the 64-antenna one keeps none)."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
           "--cuda-device-only", "-c", os.path.join(CSRC, "af_fused_gemm.hip"), "-o", str(tmp_path / "k.o"),
           "-Rpass-analysis=kernel-resource-usage"]
    text = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    blocks = re.split(r"remark: [^\n]*Function Name: ", text)[1:]
    seen = 0
    for b in blocks:
        name = b.split()[0]
        # template <bool FEED, bool RECT, int NBR, int NBC, int ST, bool FOURM>
        m = re.search(r"fused_gemm3_kernelILb([01])ELb([01])ELi(\d+)ELi(\d+)ELi(\d+)ELb([01])E", name)
        if not m:
            continue
        seen += 1
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        vgprs = int(re.search(r"VGPRs: (\d+)", b).group(1))
        agprs = int(re.search(r"AGPRs: (\d+)", b).group(1))
        lds = int(re.search(r"LDS Size \[bytes/block\]: (\d+)", b).group(1))
        assert vgprs + agprs <= 168 and scratch <= 32 and lds == 0, (name, scratch, vgprs, agprs, lds)   # LDS is dynamic
        if m.group(1) == "0" and (m.group(3) == "8"):
            assert scratch == 0, (name, scratch)          # BASELINE configs[2]'s instantiation (DIAG<8>) and the RECT super-tile
    # (8 DIAG antenna-block counts + the 8 x 4 RECT super-tile + the 8 x 8 RECT super-tile in the four-product form) x FEED
    assert seen == 20, seen


def test_single_precision_gemm_form_fits_four_waves_per_simd(tmp_path):
    """The 16-wave single-precision GEMM-form kernel (8 matrix + 8 sampling waves, csrc/af_fused_gemm_c64.hip) lives on 128
    registers with nothing in scratch: DIAG of 1 .. 8 blocks (3M form), RECT 8 x 4 (four products), RECT 8 x 8 (NEG form),
    each with and without feed rotation."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
           "--cuda-device-only", "-c", os.path.join(CSRC, "af_fused_gemm_c64.hip"), "-o", str(tmp_path / "k.o"),
           "-Rpass-analysis=kernel-resource-usage"]
    text = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    blocks = re.split(r"remark: [^\n]*Function Name: ", text)[1:]
    seen = 0
    for b in blocks:
        name = b.split()[0]
        if "fused_gemm_c64_kernel" not in name:
            continue
        seen += 1
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        vgprs = int(re.search(r"VGPRs: (\d+)", b).group(1))
        agprs = int(re.search(r"AGPRs: (\d+)", b).group(1))
        assert scratch == 0 and vgprs + agprs <= 128, (name, scratch, vgprs, agprs)
    assert seen == 20, seen
