"""Fused predict with per-antenna beam-cube DDEs (BASELINE configs[2]): lane-per-row and GEMM forms."""
import ctypes
import os
import time

import numpy as np

from .common import (FP32_PEAK_TFLOPS, FP64_PEAK_TFLOPS, L2_PEAK_GBS, NUMBA_CALIBRATION, parallel_rows as _parallel_rows,
                     sized_cpu_sample, threads_available as _threads)


def gemm_mfma_per_source(nant, single=False):
    """Matrix instructions (16 x 16 x 4) the GEMM form issues per (timestep, channel, source): the tiling of
    csrc/af_fused_gemm.hip / af_fused_gemm_c64.hip restated -- super-blocks of 8 blocks of 8 antennas; a diagonal super-block
    of n blocks: n (n + 1) / 2 tiles in the 3M form (1.5 instructions per tile and source); a pair of super-blocks: the
    column super-block's n blocks x 8 block rows, in the four-product form (2.0) when n > 4 (one 8 x 8 super-tile; single
    precision: always 2.0), else in the 3M form on 8 x 4 super-tiles.  tools/summarize_round.py holds the bench line's
    figure against SQ_INSTS_MFMA of the PMC pass."""
    nb = (nant + 7) // 8
    sizes = [min(8, nb - 8 * i) for i in range((nb + 7) // 8)]
    n = 0.0
    for i, si in enumerate(sizes):
        n += si * (si + 1) / 2 * 1.5
        for sj in sizes[i + 1:]:
            n += 8 * sj * (2.0 if (sj > 4 or single) else 1.5)
    return n


class FusedDde(object):
    """BASELINE configs[2]: 64 antennas, 2016 baselines per timestep, beam cube 257 x 257 x 33
    x 2 x 2 complex128, parallactic angles U(0, pi/6), pointing errors 1e-3 N(0,1), antenna scaling 1 +- 1e-3;
    brightness = flat-spectrum coherency matrices of the synthetic sky.  The reference chain it replaces:
    phase_delay -> einsum -> beam_cube_dde -> predict_vis (africanus/rime/examples/predict.py:404-525)."""
    NANT, LW, MH, NUD = 64, 257, 257, 33
    FORCE_ANTENNAS = None

    def __init__(self, args, rank, dev, lib, _lib, t):
        import torch
        from codex_africanus_amd.testing import synthetic_inputs
        self.args, self._lib = args, _lib
        self.NANT = self.FORCE_ANTENNAS or int(getattr(args, "antennas", 64))     # 64 unless --antennas says otherwise
        nrow, nchan, nsrc, nant = args.rows, args.chans, args.sources, self.NANT
        d = synthetic_inputs(seed=args.seed, nrow=16, nchan=nchan, nsrc=nsrc, nant=64)
        rng = np.random.default_rng(1000 + args.seed + rank)
        uvw = np.empty((nrow, 3))
        uvw[:, 0] = rng.uniform(-4000, 4000, nrow)
        uvw[:, 1] = rng.uniform(-4000, 4000, nrow)
        uvw[:, 2] = rng.uniform(-400, 400, nrow)
        lm, freq = d["lm"], d["frequency"]
        a1, a2 = np.triu_indices(nant, 1)
        nbl = a1.shape[0]
        ntime = -(-nrow // nbl)
        ant1 = np.tile(a1, ntime)[:nrow].astype(np.int32)
        ant2 = np.tile(a2, ntime)[:nrow].astype(np.int32)
        time_index = np.repeat(np.arange(ntime, dtype=np.int64), nbl)[:nrow]
        self.antennas = args.workload.startswith("fused_dde_ant") or getattr(args, "uvw", "random") == "antennas"
        if self.antennas:
            # a Measurement Set's uvw: per-(time, antenna) coordinates, baselines are their differences (same extent as
            # the per-row recipe: |u|, |v| <= 4000 m, |w| <= 400 m)
            xyz = rng.uniform(-1, 1, (ntime, nant, 3)) * np.array([2000.0, 2000.0, 200.0])
            uvw = xyz[time_index, ant1] - xyz[time_index, ant2]
        g = np.linspace(-1, 1, self.LW)
        ll, mm = np.meshgrid(g, g, indexing="ij")
        pattern = np.exp(-(ll**2 + mm**2) / 0.5) * np.exp(1j * (0.3 * ll + 0.2 * mm))
        gains = (1 + 0.02 * np.arange(self.NUD))[:, None] * np.array([1.0, 0.05j, -0.04j, 0.95])[None, :]
        beam = (pattern[:, :, None, None] * gains[None, None]).reshape(self.LW, self.MH, self.NUD, 2, 2)
        extents = np.array([[-0.06, 0.06], [-0.06, 0.06]])
        beam_freq_map = np.linspace(freq[0], freq[-1], self.NUD)
        pa = rng.uniform(0, np.pi / 6, (ntime, nant))
        if args.pa == "common":
            pa = np.linspace(0, np.pi / 6, ntime)[:, None] + 1e-3 * rng.standard_normal((ntime, nant))
        pe = 1e-3 * rng.standard_normal((ntime, nant, nchan, 2))
        asc = 1.0 + 1e-3 * rng.standard_normal((nant, nchan, 2))
        X = np.ascontiguousarray(np.broadcast_to(d["brightness"][:, None, :], (nsrc, nchan, 4))).reshape(nsrc, nchan, 2, 2)
        # the plan (host side, once per row layout): 2 x 2 blocks of baselines that share their antennas' Jones terms
        # (AFHIP_FUSED_GROUPS=0: plain row ranges, for A/B runs)
        n_items, n_groups = ctypes.c_int64(0), ctypes.c_int64(0)
        tip = time_index.ctypes.data_as(ctypes.c_void_p)
        pa1, pa2 = ant1.ctypes.data_as(ctypes.c_void_p), ant2.ctypes.data_as(ctypes.c_void_p)
        if os.environ.get("AFHIP_FUSED_GROUPS", "1") != "0":
            _lib.call("af_fused_plan_groups", tip, pa1, pa2, nrow, nant, None, 0, ctypes.byref(n_items), None, 0,
                      ctypes.byref(n_groups))
            items = np.zeros((n_items.value, 4), dtype=np.int32)
            groups = np.zeros((n_groups.value, 8), dtype=np.int32)
            _lib.call("af_fused_plan_groups", tip, pa1, pa2, nrow, nant, items.ctypes.data_as(ctypes.c_void_p),
                      n_items.value, ctypes.byref(n_items), groups.ctypes.data_as(ctypes.c_void_p), n_groups.value,
                      ctypes.byref(n_groups))
        else:
            groups = None
            _lib.call("af_fused_plan_rows", tip, nrow, None, 0, ctypes.byref(n_items))
            items = np.zeros((n_items.value, 4), dtype=np.int32)
            _lib.call("af_fused_plan_rows", tip, nrow, items.ctypes.data_as(ctypes.c_void_p), n_items.value,
                      ctypes.byref(n_items))
        self.n_items, self.ntime, self.nbl = n_items.value, ntime, nbl
        if self.antennas:
            nap = 8 * ((nant + 7) // 8)
            au, rm = np.zeros((ntime, nant, 3)), np.zeros((ntime, nap, nap), np.int32)
            res, ok = ctypes.c_double(), ctypes.c_int()
            HP = lambda x: x.ctypes.data_as(ctypes.c_void_p)
            _lib.call("af_fused_plan_antennas", tip, pa1, pa2, HP(uvw), nrow, nant, 1e-10, ntime, HP(au), HP(rm),
                      ctypes.byref(res), ctypes.byref(ok))
            if not ok.value:
                raise SystemExit("fused_dde_ant: the synthetic uvw did not decompose (residual %g m)" % res.value)
            self.plan_residual = res.value
            self.d_au, self.d_rm = t(au), t(rm)
        self.dv = dict(items=t(items), groups=None if groups is None else t(groups), a1=t(ant1), a2=t(ant2), X=t(X), beam=t(beam), ext=t(extents),
                       fmap=t(beam_freq_map), pa=t(pa), pe=t(pe), asc=t(asc), lm=t(lm), uvw=t(uvw), freq=t(freq))
        self.ws_bytes = int(lib.af_fused_predict_workspace_bytes(nsrc, nchan, self.LW, self.MH, self.NUD))
        self.d_ws = torch.empty(max(self.ws_bytes, 256), dtype=torch.uint8, device=dev)
        self.h = dict(time_index=time_index, ant1=ant1, ant2=ant2, X=X, beam=beam, extents=extents,
                      beam_freq_map=beam_freq_map, pa=pa, pe=pe, asc=asc, lm=lm, uvw=uvw, freq=freq)
        self.ncorr = 4
        self.label = ("fused predict with per-antenna beam-cube DDEs, %d antennas (BASELINE configs[2]%s), "
                      "parallactic angles %s" % (nant, "" if nant == 64 else " at another array size", args.pa))
        if self.antennas:
            self.label += "; antenna-decomposable uvw (Measurement-Set geometry): GEMM form on the fp64 matrix cores"

    def predict(self, d_vis, stream, P):
        a, v = self.args, self.dv
        if self.antennas:
            self._lib.call("af_fused_predict_antennas_c128", P(self.d_au), P(self.d_rm), self.ntime, a.rows, P(v["lm"]),
                           P(v["freq"]), P(v["X"]), a.sources, a.chans, P(v["beam"]), self.LW, self.MH, self.NUD, P(v["ext"]),
                           P(v["fmap"]), P(v["pa"]), self.ntime, self.NANT, P(v["pe"]), P(v["asc"]), None,
                           self._lib.CONVENTION["fourier"], P(d_vis), P(self.d_ws), self.ws_bytes, stream)
            return
        self._lib.call("af_fused_predict_c128", P(v["items"]), self.n_items, P(v["a1"]), P(v["a2"]),
                       None if v["groups"] is None else P(v["groups"]), a.rows,
                       P(v["lm"]), P(v["uvw"]), P(v["freq"]), P(v["X"]), a.sources, a.chans, P(v["beam"]), self.LW,
                       self.MH, self.NUD, P(v["ext"]), P(v["fmap"]), P(v["pa"]), self.ntime, self.NANT, P(v["pe"]),
                       P(v["asc"]), None, None, self._lib.CONVENTION["fourier"], P(d_vis), P(self.d_ws),
                       self.ws_bytes, stream)

    def front_end_check(self, d_vis, rank, world, dev):
        """The row-shard front-end a multi-GPU job goes through -- sharding.fused_predict_shard with this rank's rows
        and timesteps (bounds given) -- on the arrays of the benchmark: its visibilities must equal the direct C-ABI
        call's (d_vis) in every bit.  One extra predict before the timed region."""
        import torch
        from codex_africanus_amd import sharding
        v, a = self.dv, self.args
        ti = torch.from_numpy(self.h["time_index"]).to(dev)
        vis, _, bounds = sharding.fused_predict_shard(
            rank, world, ti, v["a1"], v["a2"], v["lm"], v["uvw"], v["freq"], v["X"], v["beam"], v["ext"], v["fmap"],
            v["pa"], v["pe"], v["asc"], bounds=(rank * a.rows, (rank + 1) * a.rows))
        if os.environ.get("AFHIP_BENCH_FORCE_MISMATCH"):       # exercises the forensics below (tests): one value off by one ulp
            vis = vis.clone()
            torch.view_as_real(vis).reshape(-1).view(torch.int64)[12345] += 1
        same = bool(torch.equal(vis.reshape(d_vis.shape), d_vis))
        if not same:
            raise SystemExit(self._front_end_forensics(vis.reshape(d_vis.shape), d_vis, ti, rank, world, dev))
        return "sharding.fused_predict_shard(rank %d of %d, rows %s) == the direct C-ABI call (%s): bit-equal" % (
            rank, world, bounds, "af_fused_predict_antennas_c128" if self.antennas else "af_fused_predict_c128")

    def _front_end_forensics(self, vis, d_vis, ti, rank, world, dev):
        """A front-end result that differs from the direct call's: everything needed to tell host planner from scratch
        aliasing from kernel -- digests of both calls' plan arrays, a THIRD and FOURTH opinion (the
        direct call and the front-end once more, each into a fresh buffer), where the bits differ ((timestep, channel)
        workgroups, 8 x 8-antenna tiles, correlations), NaN or value.  Written to gpurun_out/front_end_mismatch_*.json
        (+ .npz with the differing cells) and returned as the failure text."""
        import ctypes
        import hashlib
        import json
        import torch
        from codex_africanus_amd import sharding
        from codex_africanus_amd.rime import fused
        v, a = self.dv, self.args
        torch.cuda.synchronize(dev)
        dig = lambda x: hashlib.blake2b(np.ascontiguousarray(x.detach().cpu().numpy() if hasattr(x, "detach") else x)
                                        .view(np.uint8).reshape(-1), digest_size=8).hexdigest()
        rep = {"rank": rank, "world": world, "pid": os.getpid(), "time": time.time()}
        A_, B_ = torch.view_as_real(vis), torch.view_as_real(d_vis)
        bad = ((A_ != B_) | torch.isnan(A_) | torch.isnan(B_)).any(-1)
        idx = bad.nonzero()
        rows = idx[:, 0]
        rep["cells"] = int(idx.shape[0])
        rep["nan_front_end"], rep["nan_direct"] = int(torch.isnan(A_).sum()), int(torch.isnan(B_).sum())
        rep["largest_difference"] = float(torch.nan_to_num(A_ - B_).abs().max())
        steps = (rows // self.nbl)
        rep["timesteps"] = sorted(set(steps.tolist()))[:64]
        rep["channels"] = sorted(set(idx[:, 1].tolist()))
        rep["correlations"] = sorted(set(idx[:, 2].tolist()))
        wg = torch.stack([steps, idx[:, 1]], 1).unique(dim=0, return_counts=True)
        rep["workgroups_t_chan_cells"] = [[int(x[0]), int(x[1]), int(c)] for x, c in zip(wg[0][:64], wg[1][:64])]
        a1 = torch.from_numpy(self.h["ant1"]).to(dev)[rows].long()
        a2 = torch.from_numpy(self.h["ant2"]).to(dev)[rows].long()
        tl = torch.stack([a1 // 8, a2 // 8], 1).unique(dim=0, return_counts=True)
        rep["tiles_p8_q8_cells"] = [[int(x[0]), int(x[1]), int(c)] for x, c in zip(tl[0][:64], tl[1][:64])]
        rep["antennas_p"] = sorted(set(a1.tolist()))[:64]
        rep["antennas_q"] = sorted(set(a2.tolist()))[:64]
        # the two calls' plan arrays: the bench's own (made on the host at start-up) against the front-end's cached plan
        plan = fused.cached_plan(ti, v["a1"], v["a2"], self.NANT, uvw=v["uvw"])
        rep["plan"] = {"decomposable": bool(plan.decomposable), "fill": plan.fill, "residual": plan.residual}
        if plan.decomposable and self.antennas:
            rep["plan"].update(
                ant_uvw_front_end=dig(plan.ant_uvw), ant_uvw_direct=dig(self.d_au), rowmap_front_end=dig(plan.rowmap),
                rowmap_direct=dig(self.d_rm),
                ant_uvw_device_copy=[dig(t_) for k_, t_ in plan._dev.items() if k_[0] == id(plan.ant_uvw)],
                rowmap_device_copy=[dig(t_) for k_, t_ in plan._dev.items() if k_[0] == id(plan.rowmap)])
        rep["inputs"] = {k_: dig(v[k_]) for k_ in ("lm", "uvw", "freq", "X", "pa", "pe", "asc", "beam", "ext", "fmap")}
        rep["inputs_host"] = {k_: dig(self.h[k2]) for k_, k2 in (("lm", "lm"), ("uvw", "uvw"), ("freq", "freq"), ("X", "X"), ("pa", "pa"),
                                                                ("pe", "pe"), ("asc", "asc"), ("beam", "beam"))}
        # third and fourth opinions
        P = lambda x: ctypes.c_void_p(x.data_ptr())
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        C = torch.full_like(d_vis, complex(float("nan"), 0.0))
        self.predict(C, stream, P)
        D, _, _ = sharding.fused_predict_shard(rank, world, ti, v["a1"], v["a2"], v["lm"], v["uvw"], v["freq"], v["X"], v["beam"], v["ext"],
                                               v["fmap"], v["pa"], v["pe"], v["asc"], bounds=(rank * a.rows, (rank + 1) * a.rows))
        D = D.reshape(d_vis.shape)
        torch.cuda.synchronize(dev)
        eq = lambda x, y: bool(torch.equal(x, y))
        rep["agreement"] = {"front_end==direct": eq(vis, d_vis), "direct==direct_again": eq(d_vis, C), "front_end==front_end_again": eq(vis, D),
                            "direct_again==front_end_again": eq(C, D), "front_end==direct_again": eq(vis, C),
                            "direct==front_end_again": eq(d_vis, D)}
        rep["digests"] = {"front_end": dig(vis), "direct": dig(d_vis), "direct_again": dig(C), "front_end_again": dig(D)}
        out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out_dir, exist_ok=True)
        tag = "%d_%d_%d" % (int(rep["time"]), os.getpid(), rank)
        n = min(int(idx.shape[0]), 200000)
        np.savez_compressed(os.path.join(out_dir, "front_end_mismatch_%s.npz" % tag), idx=idx[:n].cpu().numpy(),
                            front_end=vis[bad][:n].cpu().numpy(), direct=d_vis[bad][:n].cpu().numpy(),
                            direct_again=C[bad][:n].cpu().numpy())
        with open(os.path.join(out_dir, "front_end_mismatch_%s.json" % tag), "w") as f:
            json.dump(rep, f, indent=1)
        return ("rank %d: sharding.fused_predict_shard differs from the C-ABI call in %d cells (NaNs %d / %d, largest difference "
                "%.3e); forensics: %s" % (rank, rep["cells"], rep["nan_front_end"], rep["nan_direct"], rep["largest_difference"],
                                          json.dumps(rep)))

    def _chain(self, rows, dde=None, tinv=None):
        """The reference chain on `rows` (only their timesteps' Jones terms are built)."""
        import oracle
        h = self.h
        if dde is None:
            tsel, tinv = np.unique(h["time_index"][rows], return_inverse=True)
            dde = oracle.beam_cube_dde(h["beam"], h["extents"], h["beam_freq_map"], h["lm"], h["pa"][tsel],
                                       h["pe"][tsel], h["asc"], h["freq"])
        phase = oracle.phase_delay(h["lm"], h["uvw"][rows], h["freq"])
        coh = np.einsum("srf,sfij->srfij", phase, h["X"])
        return oracle.predict_vis(tinv, h["ant1"][rows], h["ant2"][rows], dde, coh, dde, None, None, None)

    def reference_rows(self, rows):
        # the oracle's beam terms cost ~2.4 s per timestep on one host thread (1000 sources x 64 antennas x 64 channels):
        # check 10 rows of each of THREE timesteps (first, middle, last) instead of rows spread over all of them
        a = self.args
        picks = []
        for t in sorted({0, self.ntime // 2, self.ntime - 1}):
            lo, hi = t * self.nbl, min((t + 1) * self.nbl, a.rows)
            if hi > lo:
                picks.append(np.linspace(lo, hi - 1, min(10, hi - lo)).astype(np.int64))
        rows = np.unique(np.concatenate(picks))
        return self._chain(rows).reshape(len(rows), a.chans, 4), rows

    def roofline(self, kernel_s):
        a = self.args
        nrow, nchan, nsrc = a.rows, a.chans, a.sources
        # 64 B written per vis + uvw and indices 36 B/row + the beam cube (with |.|: 24 B per complex)
        # + parangles / pointing errors / scaling + brightness; ~150 flop per (row, chan, src): phasor 8 +
        # E X E^H 112 + 4 complex MACs 32 (SURVEY's count, kept so that rounds compare)
        alg_bytes = (nrow * nchan * 64 + nrow * 36 + self.LW * self.MH * self.NUD * 4 * 24
                     + self.ntime * self.NANT * (8 + nchan * 16) + self.NANT * nchan * 16 + nsrc * nchan * 64)
        # what the kernel actually issues (counted in the ISA of the unrolled, grouped, wave-specialised instantiation:
        # tools/count_fused_isa.sh): 63 fp64 VALU instructions per (row, chan, src) in the accumulating waves + 344
        # per 512 Jones terms in the sampling waves (one term per 31.5 units at 64 antennas); an fp64 instruction
        # occupies its SIMD for 4 cycles, so the pipe's capacity is 256 CU x 4 SIMD x 16 lanes x clock lane-instructions/s
        units = float(nrow) * nchan * nsrc
        terms = float(nsrc) * self.ntime * self.NANT * nchan
        if self.antennas:
            # the GEMM form: 8 complex MACs = 64 flop per (row, chan, source) of needed output; executed: 36 of the 64
            # (af_fused_gemm_slots / 64 beyond 64 antennas) 16 x 16 tiles of M per (timestep, channel), the complex product in the three-product (3M) form: 3 MFMA
            # 16x16x4 (2048 flop each) per tile and source PAIR = 1.5 per (tile, source).  (Until round 4 this line
            # counted 2 per (tile, source), the removed four-product kernel's schedule: VERDICT r4 "weak" 3.)
            # (36 tiles x 1.5 at 64 antennas; 128 antennas: two diagonal super-blocks in the 3M form + one 8 x 8 super-tile in
            # the four-product form)
            self.mfma_per_launch = gemm_mfma_per_source(self.NANT) * nsrc * self.ntime * nchan    # = SQ_INSTS_MFMA of the PMC pass
            mfma_flops = self.mfma_per_launch * 2048
            executed = {"mfma_instructions": self.mfma_per_launch, "mfma_flop_per_unit": mfma_flops / units,
                        "mfma_tflops": mfma_flops / kernel_s / 1e12,
                        "mfma_pipe_occupancy_at_2.4GHz": mfma_flops / kernel_s / 1e12 / FP64_PEAK_TFLOPS,
                        "note": "matrix-core flops actually issued (upper block triangle incl. the diagonal blocks' lower "
                                "halves and baselines a short last timestep lacks; 3M complex product) against the "
                                "78.6 TFLOP/s fp64 pipe; mfma_instructions must equal SQ_INSTS_MFMA of the PMC pass "
                                "(tools/summarize_round.py asserts it)"}
            return dict(kernel="fused_gemm3_kernel", bound="mfma", alg_flops=units * 64.0, alg_bytes=float(alg_bytes),
                        channels_in_kernel=nchan, executed=executed,
                        note="antenna-decomposable uvw: V(t, nu) = G H^H, M = N = 128, K = 2 nsrc per (timestep, channel) on "
                             "v_mfma_f64_16x16x4; 64 flop per (row, chan, src) (8 complex MACs) against the fp64 pipe")
        fp64_lane_instr = 63.0 * units + (344.0 * 64 / 512) * terms
        cap = 256 * 4 * 16 * 2.4e9
        executed = {"fp64_instructions_per_unit": fp64_lane_instr / units, "flop_equivalent_per_unit": 2 * fp64_lane_instr / units,
                    "fp64_pipe_occupancy_at_2.4GHz": fp64_lane_instr / kernel_s / cap,
                    "note": "fraction of the fp64 pipe's issue slots (4 cycles per wave instruction) the kernel fills at the "
                            "nominal 2.4 GHz; the chip holds ~2.03 GHz under this all-VALU fp64 mix, i.e. x 1.18 at the "
                            "clock it runs at"}
        return dict(kernel="fused_predict_kernel", bound="mfma", alg_flops=float(nrow) * nchan * nsrc * 150.0,
                    alg_bytes=float(alg_bytes), channels_in_kernel=nchan, executed=executed,
                    note="fp64 VALU bound (same 78.6 TFLOP/s fp64 pipe as the matrix path): 2x2 complex Jones "
                         "algebra per (row, chan, src), 150 flop (SURVEY 8(d))")

    def cpu_baseline(self, min_seconds):
        """One timestep of the workload through the oracle chain: beam_cube_dde for the timestep's 64 antennas
        (single thread, as the reference's numba kernel), then phase_delay -> einsum -> predict_vis on a row
        sample spread over the host threads (dask row chunks in the reference); the row part is scaled to the
        timestep's 2016 rows, so the Jones terms are amortised as in the full job."""
        import oracle
        h, a = self.h, self.args
        threads = min(_threads(), 64)
        rows_t = np.arange(min(self.nbl, a.rows))
        oracle.beam_cube_dde(h["beam"], h["extents"], h["beam_freq_map"], h["lm"][:8], h["pa"][:1], h["pe"][:1],
                             h["asc"], h["freq"])                                        # warm-up, discarded
        t0 = time.perf_counter()
        dde = oracle.beam_cube_dde(h["beam"], h["extents"], h["beam_freq_map"], h["lm"], h["pa"][:1], h["pe"][:1],
                                   h["asc"], h["freq"])
        t_beam = time.perf_counter() - t0
        tinv = np.zeros(len(rows_t), dtype=np.int64)
        self._chain(rows_t[:2], dde, tinv[:2])                                           # warm-up, discarded
        n1 = min(16, len(rows_t))
        t0 = time.perf_counter()
        self._chain(rows_t[:n1], dde, tinv[:n1])
        per_row = (time.perf_counter() - t0) / n1
        per_thread = int(max(2, min(32, 0.3 * min_seconds / per_row)))                   # coh: 4 MB per row
        n = min(len(rows_t), per_thread * threads)
        dt = _parallel_rows(lambda lo, hi: self._chain(rows_t[lo:hi], dde, tinv[lo:hi]) if hi > lo else None, n, threads)
        t_step = t_beam + dt * len(rows_t) / n
        return {
            "value": len(rows_t) * a.chans / t_step / 1e6, "unit": "Mvis/s", "cores": threads, "kind": "port",
            "sample": "oracle chain beam_cube_dde -> phase_delay -> einsum -> predict_vis (C restatements of "
                      "africanus/rime/fast_beam_cubes.py:57-240, phase.py:20-63, predict.py:193-252) for ONE "
                      "timestep (%d rows x %d chan x %d src, %d antennas): beam terms %.2f s on 1 thread + %d rows on "
                      "%d threads in %.2f s scaled to the timestep's rows" % (len(rows_t), a.chans, a.sources, self.NANT,
                                                                             t_beam, n, threads, dt),
            "single_thread_value": a.chans / (per_row + t_beam / len(rows_t)) / 1e6,
            "probe_rows": n1, "sample_rows": n, "sample_seconds": dt,
        }



class FusedDdeAnt128(FusedDde):
    """The GEMM form on a 128-antenna array (8128 baselines per timestep, the same 1e6 x 64 x 1000 counts): M is cut into
    super-tiles -- two 64-antenna diagonal super-blocks and two 8 x 4-block rectangles per (timestep, channel) (round 5;
    the reference's sum has no antenna limit, africanus/rime/predict.py:199-212)."""
    FORCE_ANTENNAS = 128


class FusedDdeAntC64(FusedDde):
    """BASELINE configs[2]'s counts on a Measurement Set's uvw in SINGLE PRECISION: every input float32 / complex64,
    complex64 out (af_fused_predict_antennas_c64: v_mfma_f32_16x16x4_f32 on float operand panels, float32 beam planes,
    the antenna phasors in double) -- the precision in which the reference runs this chain for such callers
    (africanus/util/type_inference.py:24-26).  No chi^2 in its step (the chi^2 entry is complex128).  Errors are against
    the float64 chain on the same float32 values."""
    vis_dtype, chi2 = "complex64", False

    def __init__(self, args, rank, dev, lib, _lib, t):
        import torch
        FusedDde.__init__(self, args, rank, dev, lib, _lib, t)
        h = self.h
        f, c = np.float32, np.complex64
        # a Measurement Set written in single precision: float32 differences of the antenna coordinates
        ti, a1, a2 = h["time_index"], h["ant1"], h["ant2"]
        xyz = (np.random.default_rng(3000 + args.seed + rank).uniform(-1, 1, (self.ntime, self.NANT, 3)) *
               np.array([2000.0, 2000.0, 200.0])).astype(f)
        uvw = xyz[ti, a1] - xyz[ti, a2]
        for k, dt in (("lm", f), ("freq", f), ("extents", f), ("beam_freq_map", f), ("pa", f), ("pe", f), ("asc", f), ("X", c), ("beam", c)):
            h[k] = h[k].astype(dt)
        h["uvw"] = uvw
        nap = 8 * ((self.NANT + 7) // 8)
        au, rm = np.zeros((self.ntime, self.NANT, 3)), np.zeros((self.ntime, nap, nap), np.int32)
        res, ok = ctypes.c_double(), ctypes.c_int()
        HP = lambda x: x.ctypes.data_as(ctypes.c_void_p)
        uvw64 = uvw.astype(np.float64)
        tol = float(np.abs(uvw64).max()) * 2.0 ** -22
        _lib.call("af_fused_plan_antennas", HP(np.ascontiguousarray(ti, np.int64)), HP(a1), HP(a2), HP(uvw64), args.rows, self.NANT, tol,
                  self.ntime, HP(au), HP(rm), ctypes.byref(res), ctypes.byref(ok))
        if not ok.value:
            raise SystemExit("fused_dde_ant_c64: the float32 uvw did not decompose (residual %g m, tolerance %g)" % (res.value, tol))
        self.plan_residual = res.value
        self.d_au, self.d_rm = t(au), t(rm)
        self.dv = dict(X=t(h["X"]), beam=t(h["beam"]), ext=t(h["extents"]), fmap=t(h["beam_freq_map"]), pa=t(h["pa"]), pe=t(h["pe"]),
                       asc=t(h["asc"]), lm=t(h["lm"]), uvw=t(uvw), freq=t(h["freq"]), a1=t(a1), a2=t(a2))
        self.ws_bytes = int(lib.af_fused_predict_c64_workspace_bytes(args.sources, args.chans, self.LW, self.MH, self.NUD))
        self.d_ws = torch.empty(max(self.ws_bytes, 256), dtype=torch.uint8, device=dev)
        self.label = ("fused predict with per-antenna beam-cube DDEs, %d antennas, Measurement-Set uvw, SINGLE precision "
                      "(float32 / complex64 in, complex64 out): GEMM form on the fp32 matrix cores" % self.NANT)

    def predict(self, d_vis, stream, P):
        a, v = self.args, self.dv
        self._lib.call("af_fused_predict_antennas_c64", P(self.d_au), P(self.d_rm), self.ntime, a.rows, P(v["lm"]), P(v["freq"]),
                       P(v["X"]), a.sources, a.chans, P(v["beam"]), self.LW, self.MH, self.NUD, P(v["ext"]), P(v["fmap"]), P(v["pa"]),
                       self.ntime, self.NANT, P(v["pe"]), P(v["asc"]), None, self._lib.CONVENTION["fourier"], P(d_vis), P(self.d_ws),
                       self.ws_bytes, stream)

    def front_end_check(self, d_vis, rank, world, dev):
        """rime.fused_predict_vis on the same single-precision tensors must take the same route: same bits"""
        import torch
        from codex_africanus_amd import rime
        v = self.dv
        ti = torch.from_numpy(self.h["time_index"]).to(dev)
        vis = rime.fused_predict_vis(ti, v["a1"], v["a2"], v["lm"], v["uvw"], v["freq"], v["X"], v["beam"], v["ext"], v["fmap"], v["pa"],
                                     v["pe"], v["asc"])
        if vis.dtype != torch.complex64 or not bool(torch.equal(vis.reshape(d_vis.shape), d_vis)):
            raise SystemExit("rank %d: rime.fused_predict_vis on single-precision tensors differs from af_fused_predict_antennas_c64" % rank)
        return "rime.fused_predict_vis(float32 / complex64 tensors) == af_fused_predict_antennas_c64: bit-equal, complex64"

    def _chain(self, rows, dde=None, tinv=None):
        """The reference chain in float64 on the promoted single-precision values."""
        import oracle
        h = self.h
        p = lambda a: a.astype(np.complex128 if np.iscomplexobj(a) else np.float64)
        if dde is None:
            tsel, tinv = np.unique(h["time_index"][rows], return_inverse=True)
            dde = oracle.beam_cube_dde(p(h["beam"]), p(h["extents"]), p(h["beam_freq_map"]), p(h["lm"]), p(h["pa"][tsel]),
                                       p(h["pe"][tsel]), p(h["asc"]), p(h["freq"]))
        phase = oracle.phase_delay(p(h["lm"]), p(h["uvw"][rows]), p(h["freq"]))
        coh = np.einsum("srf,sfij->srfij", phase, p(h["X"]))
        return oracle.predict_vis(tinv, h["ant1"][rows], h["ant2"][rows], dde, coh, dde, None, None, None)

    def roofline(self, kernel_s):
        r = FusedDde.roofline(self, kernel_s)
        a = self.args
        units = float(a.rows) * a.chans * a.sources
        # 3M form on the diagonal super-blocks, four products on the rectangles (v_mfma_f32_16x16x4_f32, 2048 flop each)
        self.mfma_per_launch = gemm_mfma_per_source(self.NANT, single=True) * a.sources * self.ntime * a.chans
        r.update(kernel="fused_gemm_c64_kernel", peak_tflops=FP32_PEAK_TFLOPS,
                 executed={"mfma_instructions": self.mfma_per_launch, "mfma_tflops": self.mfma_per_launch * 2048 / kernel_s / 1e12},
                 note="single precision: 64 flop per (row, chan, src) (8 complex MACs) against the fp32 matrix peak (157.3 TFLOP/s); "
                      "v_mfma_f32_16x16x4_f32, 3M form on diagonal super-blocks")
        return r


class FusedDdeC64(FusedDdeAntC64):
    """BASELINE configs[2] as it draws its uvw (per row: nothing decomposes) in SINGLE PRECISION: every input float32 /
    complex64, complex64 out (af_fused_predict_c64: the lane-per-row kernel with packed float32 Jones algebra, float32 beam
    planes and sampler, phases in double).  No chi^2 in its step.  Errors are against the float64 chain on the same float32
    values."""

    def __init__(self, args, rank, dev, lib, _lib, t):
        import torch
        FusedDde.__init__(self, args, rank, dev, lib, _lib, t)
        h = self.h
        f, c = np.float32, np.complex64
        for k, dt in (("lm", f), ("freq", f), ("extents", f), ("beam_freq_map", f), ("pa", f), ("pe", f), ("asc", f), ("uvw", f),
                      ("X", c), ("beam", c)):
            h[k] = h[k].astype(dt)
        keep = {k: self.dv[k] for k in ("items", "groups", "a1", "a2")}
        self.dv = dict(keep, X=t(h["X"]), beam=t(h["beam"]), ext=t(h["extents"]), fmap=t(h["beam_freq_map"]), pa=t(h["pa"]),
                       pe=t(h["pe"]), asc=t(h["asc"]), lm=t(h["lm"]), uvw=t(h["uvw"]), freq=t(h["freq"]))
        self.ws_bytes = int(lib.af_fused_predict_c64_workspace_bytes(args.sources, args.chans, self.LW, self.MH, self.NUD))
        self.d_ws = torch.empty(max(self.ws_bytes, 256), dtype=torch.uint8, device=dev)
        self.label = ("fused predict with per-antenna beam-cube DDEs, %d antennas, uvw drawn per row (BASELINE configs[2]), SINGLE "
                      "precision (float32 / complex64 in, complex64 out): lane-per-row kernel, packed float32" % self.NANT)

    def predict(self, d_vis, stream, P):
        a, v = self.args, self.dv
        self._lib.call("af_fused_predict_c64", P(v["items"]), self.n_items, P(v["a1"]), P(v["a2"]),
                       None if v["groups"] is None else P(v["groups"]), a.rows, P(v["lm"]), P(v["uvw"]), P(v["freq"]), P(v["X"]),
                       a.sources, a.chans, P(v["beam"]), self.LW, self.MH, self.NUD, P(v["ext"]), P(v["fmap"]), P(v["pa"]),
                       self.ntime, self.NANT, P(v["pe"]), P(v["asc"]), None, None, self._lib.CONVENTION["fourier"], P(d_vis),
                       P(self.d_ws), self.ws_bytes, stream)

    def front_end_check(self, d_vis, rank, world, dev):
        """rime.fused_predict_vis on the same single-precision tensors must take the same route: same bits"""
        import torch
        from codex_africanus_amd import rime
        v = self.dv
        ti = torch.from_numpy(self.h["time_index"]).to(dev)
        vis = rime.fused_predict_vis(ti, v["a1"], v["a2"], v["lm"], v["uvw"], v["freq"], v["X"], v["beam"], v["ext"], v["fmap"], v["pa"],
                                     v["pe"], v["asc"])
        if vis.dtype != torch.complex64 or not bool(torch.equal(vis.reshape(d_vis.shape), d_vis)):
            raise SystemExit("rank %d: rime.fused_predict_vis on single-precision tensors differs from af_fused_predict_c64" % rank)
        return "rime.fused_predict_vis(float32 / complex64 tensors) == af_fused_predict_c64: bit-equal, complex64"

    def roofline(self, kernel_s):
        r = FusedDde.roofline(self, kernel_s)
        a = self.args
        units = float(a.rows) * a.chans * a.sources
        r.update(kernel="fused_rows_c64_kernel", peak_tflops=FP32_PEAK_TFLOPS,
                 executed={"packed_f32_instructions_per_unit": 24.0, "fp64_instructions_per_unit": 6.0,
                           "note": "accumulating waves, per (row, chan, src): 24 packed float32 instructions (M = G E^H, acc += y M) + "
                                   "9 float32 + 6 fp64 for the phasor"},
                 note="single precision: 150 flop per (row, chan, src) (SURVEY 8(d)'s count, as the fp64 row kernel's line) against "
                      "the fp32 vector peak (157.3 TFLOP/s, packed)")
        return r
